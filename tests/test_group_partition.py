"""CPU: the multi-GPU layer of the C-ABI (plk_group_*, phyly_amd/csrc/host_group.c): contiguous site blocks, one
host thread per engine, per-site outputs at global positions, {hi, lo} partial sums added in engine order.

tests/group_check.c links host_group.c with a stand-in engine whose outputs depend on exactly the data block it
received; 220 cases (1..4097 sites, 1..9 engines incl. more engines than sites, codes and dense observations, with
and without weights, masks) compare a group of G engines with a single engine: per-site outputs bit for bit, sums
to 1e-15.  Run under AddressSanitizer + UBSan and under ThreadSanitizer (the engines of a group run concurrently).
The same partition is exercised on a real GPU with ARBPLF_DEVICES=0,0 in tests/test_gpu_group.py."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_partition_and_reduction(tmp_path, san):
    out = str(tmp_path / "group_check")
    cmd = ["gcc", "-O1", "-g", "-std=c11", "-D_GNU_SOURCE", "-fsanitize=" + san, "-fno-omit-frame-pointer",
           "-I", os.path.join(ROOT, "include"), "-o", out, os.path.join(ROOT, "tests", "group_check.c"),
           os.path.join(ROOT, "phyly_amd", "csrc", "host_group.c"), "-lpthread", "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build unavailable: " + r.stderr[-300:])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1",
               TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([out], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    assert r.stdout.split() == ["ok", "220"]
