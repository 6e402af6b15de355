"""CPU: the traversal-program builder and the host replays of the device interpreters
(phyly_amd/csrc/plk_program.h) under AddressSanitizer + UBSan.

The engine refuses to launch a kernel whose program, LDS image or register-stack use the replay rejects
(PLK_E_ARG); this test runs the same builder + replays over
  * 6000 seeded random trees of every shape (chains, stars, caterpillars, complete binary / ternary, random
    multifurcating, 2..3500 nodes, data on internal nodes, 1..256 character definitions) with negative controls,
  * the trees and definition counts of the seeded random JSON queries of tests/test_gpu_differential.py as the host
    layer hands them to the engine once probability arrays are compacted to codes (every distinct row a definition).
"""
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "phyly_amd", "csrc")


@pytest.fixture(scope="module")
def binary(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("prog") / "progcheck")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
           "-I", CSRC, "-o", out, os.path.join(ROOT, "tests", "progcheck_main.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return out


ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")


def test_random_trees_every_variant(binary):
    r = subprocess.run([binary], capture_output=True, text=True, env=ENV, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    tag, trees, ops = r.stdout.split()
    assert tag == "ok" and int(trees) >= 6000 and int(ops) > 100000


def test_differential_generator_inputs(binary, tmp_path):
    from test_gpu_differential import random_model
    lines = []
    for kind, seed in (("ll", 11), ("deriv", 22), ("marginal", 33), ("ll", 77), ("ll", 99)):
        rng = random.Random(seed)
        for _ in range(70):
            md = random_model(rng, kind)["model_and_data"]
            edges = md["edges"]
            n = len(edges) + 1
            if "probability_array" in md:
                rows = {tuple(r) for site in md["probability_array"] for r in site}
                nchar = len(rows)
                has = [int(any(any(v != 1 for v in site[a]) for site in md["probability_array"])) for a in range(n)]
            else:
                defs = md["character_definitions"]
                nchar = len(defs)
                has = [int(any(any(v != 1 for v in defs[site[a]]) for site in md["character_data"])) for a in range(n)]
            if nchar > 256:
                continue
            lines.append("%d %d %s | %s" % (n, nchar, " ".join("%d %d" % (a, b) for a, b in edges), " ".join(map(str, has))))
    f = tmp_path / "trees.txt"
    f.write_text("\n".join(lines) + "\n")
    r = subprocess.run([binary, str(f)], capture_output=True, text=True, env=ENV, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.split()[:2] == ["ok", str(len(lines))]
