"""CPU: the C-ABI library loads and exports every symbol include/*.h declares (no
compute calls), fails loudly without a GPU, and the N > 1 site-sharding path
(phyly_amd.shard) reproduces the unsharded sums under torch.distributed/gloo
with world_size 2."""
import ctypes
import json
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    for hdr in ("plk.h", "arbplf.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"\b((?:plk|arbplf)_[a-z0-9_]+)\s*\(", text):
            names.add(m.group(1))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    from phyly_amd import engine
    lib = engine.load_library()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libarbplf_amd.so does not export %s" % n


def test_no_cpu_fallback_fails_loudly():
    """without a GPU (this container) engine creation must fail with a message"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from phyly_amd import engine
    with pytest.raises(engine.EngineError) as e:
        engine.Engine(0)
    assert "no CPU fallback" in str(e.value)
    import arbplf
    good = open(os.path.join(ROOT, "tests/golden/examples/bpp.phyl/ll/in.json")).read()
    with pytest.raises(RuntimeError):
        arbplf.arbplf_ll(good)
    exe = os.path.join(ROOT, "phyly_amd", "csrc", "arbplf-ll")
    p = subprocess.run([exe], input=good.encode(), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode != 0 and p.stdout == b"" and b"no CPU fallback" in p.stderr


def test_product_does_not_reference_oracle():
    """nothing under phyly_amd/ (or the shims at the root) imports or links oracle/"""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "phyly_amd")):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", "Makefile")):
                text = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"\boracle\b|liborc|plf_core", text):
                    bad.append(os.path.join(base, f))
    assert not bad, bad
    assert "oracle" not in open(os.path.join(ROOT, "arbplf.py")).read()


def test_shard_ranges_partition():
    from phyly_amd.shard import shard_range
    for S in (1, 7, 8, 1000, 10_000_000):
        for world in (1, 2, 3, 8):
            got = [shard_range(S, r, world) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == S
            for (a0, a1), (b0, b1) in zip(got, got[1:]):
                assert a1 == b0 and a0 <= a1
            assert max(b - a for a, b in got) == -(-S // world)


_WORKER = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import torch.distributed as dist
from phyly_amd import synth, shard
from oracle import arbplf_oracle as O
from helpers import oracle_model
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
wl = synth.Workload(T=12, k=4, tree="yule", model="gtr_g4", seed=5)
S = 1001
m, w = oracle_model(O, wl, wl.simulate(1))

def local(s0, s1):
    codes = wl.simulate(s1 - s0, site0=s0)                 # this rank's block of the alignment
    ll, _ = O.site_ll(m, w, codes=np.ascontiguousarray(codes.T), defs=wl.defs)
    wts = 1.0 + 0.001 * np.arange(s0, s1)
    tot = np.sum(ll.astype(np.longdouble) * wts)
    hi = float(tot)
    return np.array([[hi, float(tot - hi)]])

total = shard.sharded_sum(local, S, rank, world)
if rank == 0:
    print(json.dumps({"total": float(total[0])}))
dist.barrier()
dist.destroy_process_group()
"""


def test_site_sharding_gloo_world2(tmp_path, oracle):
    """two ranks, each evaluating its block of the same synthetic alignment, one all-reduce"""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % {"root": ROOT})
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e.decode()[-2000:]
    got = json.loads(outs[0][0].decode().strip().splitlines()[-1])["total"]
    # unsharded reference
    from phyly_amd import synth
    from helpers import oracle_model
    wl = synth.Workload(T=12, k=4, tree="yule", model="gtr_g4", seed=5)
    m, w = oracle_model(oracle, wl, wl.simulate(1))
    codes = wl.simulate(1001)
    ll, _ = oracle.site_ll(m, w, codes=np.ascontiguousarray(codes.T), defs=wl.defs)
    want = float(np.sum(ll.astype(np.longdouble) * (1.0 + 0.001 * np.arange(1001))))
    assert abs(got - want) <= 1e-13 * abs(want)


_WORKER_EMPTY = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from phyly_amd import shard
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
S = 2

def local(s0, s1):                       # [3][2] double-double partials of this block
    x = np.arange(s0, s1, dtype=np.float64) + 1.0
    return np.stack([[x.sum(), 1e-20 * len(x)], [(x * x).sum(), 0.0], [len(x), 0.0]])

tot = shard.sharded_sum(local, S, rank, world)
tot2 = shard.sharded_sum(local, S, rank, world, shape=(3, 2))
if rank == world - 1:
    print(json.dumps({"tot": tot.tolist(), "tot2": tot2.tolist(), "block": list(shard.shard_range(S, rank, world))}))
dist.barrier()
dist.destroy_process_group()
"""


def test_rank_without_sites_contributes_zeros_gloo_world3(tmp_path):
    """more ranks than sites (S = 2, three ranks): the empty rank takes part in the all-reduce with zeros"""
    script = tmp_path / "worker_empty.py"
    script.write_text(_WORKER_EMPTY % {"root": ROOT})
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(3):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e.decode()[-2000:]
    got = json.loads(outs[2][0].decode().strip().splitlines()[-1])
    assert got["block"] == [2, 2]                                     # the reporting rank is the empty one
    assert got["tot"] == [3.0, 5.0, 2.0] and got["tot2"] == got["tot"]
