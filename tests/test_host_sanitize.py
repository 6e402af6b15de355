"""CPU: the host layer (JSON reader, validation, reductions, K0 set-up paths) under AddressSanitizer + UBSan.

tests/sanitize_host.c links host_*.c with stubs in place of the GPU engine, validates every golden input of its
kind, runs the driver entry point up to the (failing) engine call, and then feeds a few thousand byte-level
mutations of those inputs through the same paths.  Any heap error, leak or undefined behaviour aborts the binary."""
import glob
import os
import subprocess

import pytest

from helpers import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "phyly_amd", "csrc")
EX = os.path.join(GOLDEN, "examples")


@pytest.fixture(scope="module")
def binary(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("san") / "sanitize_host")
    srcs = [os.path.join(ROOT, "tests", "sanitize_host.c")] + sorted(glob.glob(os.path.join(CSRC, "host_*.c")))
    cmd = ["gcc", "-O1", "-g", "-std=c11", "-D_GNU_SOURCE", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-o", out] + srcs + ["-lquadmath", "-lm", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build unavailable: " + r.stderr[-300:])
    return out


def _files(kind):
    pats = {"ll": ["*/in.json", "*/ll/in*.json"], "deriv": ["*/deriv/in.json", "JC.long.branch/*.json"],
            "marginal": ["*/marginal/in.json"], "dwell": ["*/dwell/*/in.json", "BEAST.MarkovJumps/MarkovRewardsC/in.json"],
            "trans": ["*/trans/*/in*.json", "BEAST.MarkovJumps/MarkovJumpsC/in*.json", "BEAST.MarkovJumps/MarkovMarginalRate/in*.json"],
            "em_update": ["*/em-update/*/in.json"], "hess": ["*/hess/*/in.json"]}[kind]
    out = []
    for p in pats:
        out += sorted(glob.glob(os.path.join(EX, p)))
    return out


@pytest.mark.parametrize("kind", ["ll", "deriv", "marginal", "dwell", "trans", "em_update", "hess"])
def test_host_layer_under_sanitizers(binary, kind):
    files = _files(kind)
    assert files
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    for fake in ("0", "1"):       # 0: the engine fails (error paths); 1: a stand-in engine succeeds (table-writing paths)
        r = subprocess.run([binary, kind] + files, capture_output=True, env=dict(env, FAKE_ENGINE=fake), timeout=600)
        out, err = r.stdout.decode("utf-8", "replace"), r.stderr.decode("utf-8", "replace")   # diagnostics echo mutated bytes
        assert r.returncode == 0, (fake, out[-500:], err[-3000:])
        last = out.strip().splitlines()[-1].split()
        assert last[0] == "files" and int(last[1]) == len(files)
        assert int(last[3]) == len(files), "every golden input must validate: " + out[-300:]


@pytest.mark.parametrize("kind,seed", [("ll", 11), ("deriv", 22), ("marginal", 33)])
def test_random_queries_under_sanitizers(binary, tmp_path, kind, seed):
    """the seeded random queries of tests/test_gpu_differential.py (multifurcating trees, duplicate selections,
    weighted aggregations, both observation forms) through the drivers with the stand-in engine"""
    import json
    import random
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_differential import random_model
    rng = random.Random(seed)
    files = []
    for i in range(30):
        f = tmp_path / ("q%03d.json" % i)
        f.write_text(json.dumps(random_model(rng, kind)))
        files.append(str(f))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               FAKE_ENGINE="1")
    for devices in (None, "0,1,2"):      # one engine; three engines (site blocks, threads, partial-sum reduction)
        env.pop("ARBPLF_DEVICES", None)
        if devices:
            env["ARBPLF_DEVICES"] = devices
        r = subprocess.run([binary, kind] + files, capture_output=True, env=env, timeout=600)
        out, err = r.stdout.decode("utf-8", "replace"), r.stderr.decode("utf-8", "replace")
        assert r.returncode == 0, (devices, out[-500:], err[-3000:])


def test_compaction_of_probability_arrays_under_sanitizers(binary, tmp_path):
    """the 256-entry definition table of parse_probability_array (phyly_amd/csrc/host_model.c): exactly 256 distinct
    rows, 257 (dense fallback), 0.0 / -0.0 rows, one site, and the dense layout forced by ARBPLF_COMPACT_DENSE=0"""
    import json
    import random
    rng = random.Random(5)
    files = []
    for i, nrows in enumerate([1, 4, 16, 17, 255, 256, 257, 400]):
        n_nodes, k = 4, 3
        S = -(-nrows // n_nodes)
        rows = [[round(rng.random(), 6) + 0.01 for _ in range(k)] for _ in range(S * n_nodes)]
        for r in range(nrows, S * n_nodes):
            rows[r] = rows[0]
        pa = [rows[s * n_nodes:(s + 1) * n_nodes] for s in range(S)]
        if i == 2:
            pa[0][1] = [0.0, 1.0, 0.0]
            pa[0][2] = [-0.0, 1.0, -0.0]
        x = {"model_and_data": {"edges": [[0, 1], [0, 2], [0, 3]], "edge_rate_coefficients": [0.1, 0.2, 0.3],
                                "rate_matrix": [[0, 1, 1], [1, 0, 1], [1, 1, 0]], "probability_array": pa},
             "site_reduction": {"aggregation": "sum"}}
        f = tmp_path / ("c%d.json" % i)
        f.write_text(json.dumps(x))
        files.append(str(f))
    for dense in (None, "0"):
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
                   FAKE_ENGINE="1")
        env.pop("ARBPLF_COMPACT_DENSE", None)
        if dense:
            env["ARBPLF_COMPACT_DENSE"] = dense
        for kind in ("ll", "deriv", "marginal"):
            r = subprocess.run([binary, kind] + files, capture_output=True, env=env, timeout=600)
            out, err = r.stdout.decode("utf-8", "replace"), r.stderr.decode("utf-8", "replace")
            assert r.returncode == 0, (dense, kind, out[-500:], err[-3000:])
            last = out.strip().splitlines()[-1].split()
            assert int(last[3]) == len(files), out[-300:]
