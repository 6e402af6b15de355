"""GPU: N > 1 RANKS on the HIP path.

Two (or three) fresh child processes, one rank each, every rank creating its own Engine on the box's one GPU,
evaluating its shard.shard_range block of the same synthetic alignment with the HIP kernels (ll sum, site-summed
edge gradient, site-summed marginals) and reducing the {hi, lo} partial sums with one all-reduce.  RCCL refuses two
ranks on one device, so the collective runs over gloo here; the payload (16 B ... 2Nk doubles) does not care.  The
result must equal one engine on the whole alignment to 1e-13.  The reference's only cross-site step is this
reduction (src/ndaccum.c:198-254 after the site loops of src/arbplfll.c:139-170, src/arbplfderiv.c:329-356,
src/arbplfmarginal.c:237-256)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from phyly_amd import synth, shard
from phyly_amd.engine import Engine
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
cfg, S = %(cfg)d, %(S)d
wl = synth.Workload(cfg)
eng = None

def engine_for(s0, s1):
    global eng
    eng = Engine(0)
    wl.setup_engine(eng)
    eng.set_patterns_codes(wl.simulate(s1 - s0, site0=s0), wl.defs)
    eng.set_site_weights(1.0 + 1e-6 * np.arange(s0, s1, dtype=np.float64))

def ll_part(s0, s1):
    if eng is None: engine_for(s0, s1)
    _, (hi, lo) = eng.ll(per_site=False)
    return np.array([[hi, lo]])

def deriv_part(s0, s1):
    return eng.deriv(per_site=False)[1]

def marg_part(s0, s1):
    return eng.marginal(per_site=False)[1]

out = {}
out["ll"] = shard.sharded_sum(ll_part, S, rank, world).tolist()
out["deriv"] = shard.sharded_sum(deriv_part, S, rank, world).tolist()
out["marginal"] = shard.sharded_sum(marg_part, S, rank, world).tolist()
s0, s1 = shard.shard_range(S, rank, world)
out["block"] = [s0, s1]
out["kernel"] = eng.info(0) if eng is not None else -1
if rank == 0:
    print(json.dumps(out))
dist.barrier()
dist.destroy_process_group()
"""


def _run_ranks(tmp_path, world, cfg, S):
    script = tmp_path / ("worker_%d_%d.py" % (world, S))
    script.write_text(_WORKER % {"root": ROOT, "cfg": cfg, "S": S})
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e.decode()[-3000:]
    return json.loads(outs[0][0].decode().strip().splitlines()[-1])


def _one_engine(cfg, S):
    from phyly_amd import synth
    from phyly_amd.engine import Engine
    wl = synth.Workload(cfg)
    eng = Engine(0)
    wl.setup_engine(eng)
    eng.set_patterns_codes(wl.simulate(S), wl.defs)
    eng.set_site_weights(1.0 + 1e-6 * np.arange(S, dtype=np.float64))
    _, (hi, lo) = eng.ll(per_site=False)
    d = eng.deriv(per_site=False)[1]
    m = eng.marginal(per_site=False)[1]
    eng.close()
    return hi + lo, d[:, 0] + d[:, 1], m[..., 0] + m[..., 1]


def _close(got, want, scale=None):
    got, want = np.asarray(got, float), np.asarray(want, float)
    sc = np.maximum(np.abs(want), np.max(np.abs(want)) * 1e-3 if scale is None else scale)
    return float(np.max(np.abs(got - want) / sc))


def test_two_ranks_shard_cfg3_on_the_hip_path(tmp_path):
    """BASELINE config 3 (GTR+G4, 100 taxa), 200k sites cut into two blocks, two processes, two engines"""
    S = 200_000
    got = _run_ranks(tmp_path, 2, 3, S)
    assert got["block"] == [0, S // 2] and got["kernel"] == 1          # the fused k = 4 kernel ran on the rank
    ll, d, m = _one_engine(3, S)
    assert abs(got["ll"][0] - ll) <= 1e-13 * abs(ll)
    assert _close(got["deriv"], d) <= 1e-13
    assert _close(got["marginal"], m, scale=float(S)) <= 1e-13


def test_three_ranks_with_an_empty_one(tmp_path):
    """more ranks than sites: the empty rank contributes zeros instead of raising"""
    S = 2
    got = _run_ranks(tmp_path, 3, 2, S)
    ll, d, m = _one_engine(2, S)
    assert abs(got["ll"][0] - ll) <= 1e-13 * abs(ll)
    assert _close(got["deriv"], d) <= 1e-12
    assert _close(got["marginal"], m, scale=float(S)) <= 1e-13


_RCCL_WORKER = r"""
import sys
sys.path.insert(0, %(root)r)
import torch
from phyly_amd import synth
from phyly_amd.engine import Engine
wl = synth.Workload(2)
eng = Engine(0)
wl.setup_engine(eng)
eng.set_patterns_codes(wl.simulate(5000), wl.defs)
_, (hi, lo) = eng.ll(per_site=False)
eng.comm_init(1, 0, Engine.comm_unique_id())
red = torch.zeros(2, dtype=torch.float64, device="cuda:0")
eng.set_stream(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    eng.update_edge_rates(wl.edge_rates_csr)
    eng.ll_async(sum_device_ptr=red.data_ptr())
    eng.allreduce_sum_async(red.data_ptr(), 2)
torch.cuda.synchronize()
assert red.tolist() == [hi, lo], (red.tolist(), hi, lo)
eng.comm_destroy()
eng.close()
print("RCCL_OK")
"""


def test_engine_rccl_reduction_world_size_one(tmp_path):
    """plk_comm_* / plk_allreduce_sum_async (include/plk.h): the engine loads the process's RCCL and queues the reduction
    on its own stream.  One rank is all the one-GPU box allows (RCCL refuses two ranks on a device): the sum over one
    rank is the input, queued behind an ll evaluation whose {hi, lo} it reduces in place.  In a fresh process, as a
    rank of a real job is (RCCL's own initialisation does not always succeed late in a long-lived test process)."""
    script = tmp_path / "rccl_worker.py"
    script.write_text(_RCCL_WORKER % {"root": ROOT})
    p = subprocess.run([sys.executable, str(script)], env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0 and b"RCCL_OK" in p.stdout, p.stderr.decode()[-3000:]
