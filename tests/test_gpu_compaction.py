"""GPU: `probability_array` inputs re-expressed as character codes + definitions by the host layer (default) against
the same inputs kept dense (ARBPLF_COMPACT_DENSE=0) and against the oracle.

Compaction is bitwise on the observation rows, so both layouts describe the same model; they run on different
kernels (compact: fused / vector / matrix-core kernels with tip tables; dense: the generic kernels), so agreement
is a cross-check of the two kernel families on reference-style inputs.  Edge cases of the 256-entry definition
table: exactly 256 distinct rows (compact), 257 (falls back to dense), 0.0 vs -0.0 rows, all-zero rows."""
import glob
import json
import os
import random

import numpy as np
import pytest

from helpers import GOLDEN

pytestmark = pytest.mark.gpu
EX = os.path.join(GOLDEN, "examples")


def _fns():
    import arbplf
    return {"ll": arbplf.arbplf_ll, "deriv": arbplf.arbplf_deriv, "marginal": arbplf.arbplf_marginal,
            "dwell": arbplf.arbplf_dwell, "trans": arbplf.arbplf_trans, "em_update": arbplf.arbplf_em_update,
            "hess": arbplf.arbplf_hess}


def _both(monkeypatch, kind, s):
    fn = _fns()[kind]
    monkeypatch.delenv("ARBPLF_COMPACT_DENSE", raising=False)
    compact = json.loads(fn(s))
    monkeypatch.setenv("ARBPLF_COMPACT_DENSE", "0")
    dense = json.loads(fn(s))
    monkeypatch.delenv("ARBPLF_COMPACT_DENSE", raising=False)
    return compact, dense


def _close(a, b, rel=1e-12, floor=1.0):
    assert a["columns"] == b["columns"] and len(a["data"]) == len(b["data"])
    scale = max([abs(r[-1]) for r in b["data"]] + [floor])
    for x, y in zip(a["data"], b["data"]):
        assert x[:-1] == y[:-1]
        assert abs(x[-1] - y[-1]) <= rel * max(abs(y[-1]), 1e-2 * scale) + 1e-300, (x, y)


GOLDEN_PA = []
for _kind, _pats in (("ll", ["*/in.json", "*/ll/in*.json"]), ("deriv", ["*/deriv/in.json"]), ("marginal", ["*/marginal/in.json"]),
                     ("dwell", ["*/dwell/*/in.json"]), ("trans", ["*/trans/*/in*.json"]), ("em_update", ["*/em-update/*/in.json"]),
                     ("hess", ["*/hess/*/in.json"])):
    for _p in _pats:
        for _f in sorted(glob.glob(os.path.join(EX, _p))):
            with open(_f) as _fh:
                if "probability_array" in _fh.read():
                    GOLDEN_PA.append((_kind, _f))


@pytest.mark.parametrize("kind,path", GOLDEN_PA, ids=[k + ":" + os.path.relpath(p, EX) for k, p in GOLDEN_PA])
def test_golden_probability_arrays_compact_equals_dense(monkeypatch, kind, path):
    with open(path) as f:
        s = f.read()
    compact, dense = _both(monkeypatch, kind, s)
    _close(compact, dense, rel=1e-11 if kind in ("hess", "em_update") else 1e-12)


@pytest.mark.parametrize("kind", ["ll", "deriv", "marginal"])
def test_random_probability_arrays_compact_equals_dense_and_oracle(monkeypatch, oracle, kind):
    from test_gpu_differential import random_model, _check
    orc = {"ll": oracle.arbplf_ll, "deriv": oracle.arbplf_deriv, "marginal": oracle.arbplf_marginal}[kind]
    rng = random.Random({"ll": 511, "deriv": 522, "marginal": 533}[kind])
    done = 0
    while done < 25:
        x = random_model(rng, kind)
        if "probability_array" not in x["model_and_data"]:
            continue
        s = json.dumps(x)
        want = json.loads(orc(s))
        if any(not np.isfinite(r[-1]) for r in want["data"]):
            continue
        compact, dense = _both(monkeypatch, kind, s)
        _check(kind, compact, want)
        _check(kind, dense, want)
        done += 1


def _model(n_rows_distinct, k=4, signed_zero=False, seed=3):
    """star tree with 3 leaves; sites whose rows are pairwise distinct until n_rows_distinct rows exist"""
    rng = np.random.default_rng(seed)
    n_nodes = 4
    S = -(-n_rows_distinct // n_nodes)
    rows = np.round(rng.random((S * n_nodes, k)) * 0.9 + 0.05, 6)
    rows[n_rows_distinct:] = rows[0]                 # repeat a row so that exactly n_rows_distinct differ
    pa = rows.reshape(S, n_nodes, k).tolist()
    if signed_zero:
        pa[0][1] = [0.0, 1.0, 0.0, 1.0][:k]
        pa[0][2] = [-0.0, 1.0, -0.0, 1.0][:k]
    Q = (rng.random((k, k)) + 0.1).tolist()
    return {"model_and_data": {"edges": [[0, 1], [0, 2], [0, 3]], "edge_rate_coefficients": [0.1, 0.2, 0.3],
                               "rate_matrix": Q, "probability_array": pa, "root_prior": "equilibrium_distribution"}}


@pytest.mark.parametrize("nrows", [16, 17, 255, 256, 257, 300])
@pytest.mark.parametrize("kind", ["ll", "deriv", "marginal"])
def test_definition_table_limits(monkeypatch, oracle, kind, nrows):
    """16 / 17 rows: 4-bit vs 8-bit staged codes; 256: the last compact case; 257+: dense fallback"""
    from test_gpu_differential import _check
    orc = {"ll": oracle.arbplf_ll, "deriv": oracle.arbplf_deriv, "marginal": oracle.arbplf_marginal}[kind]
    s = json.dumps(_model(nrows))
    want = json.loads(orc(s))
    compact, dense = _both(monkeypatch, kind, s)
    _check(kind, compact, want)
    _check(kind, dense, want)


def test_signed_zero_rows_are_distinct_definitions_with_one_meaning(monkeypatch, oracle):
    from test_gpu_differential import _check
    s = json.dumps(_model(9, signed_zero=True))
    for kind in ("ll", "marginal"):
        orc = {"ll": oracle.arbplf_ll, "marginal": oracle.arbplf_marginal}[kind]
        compact, dense = _both(monkeypatch, kind, s.replace("-0.0", "-0.0"))
        _check(kind, compact, json.loads(orc(s)))
        _check(kind, dense, json.loads(orc(s)))


def test_many_definitions_large_lds_image(monkeypatch, oracle):
    """a 60-leaf tree with 40 distinct rows: the fused kernel's LDS image (tip tables of 61 slots x 40 definitions
    = 78 KB + code rows) is above the 64 KB a workgroup gets without asking"""
    from test_gpu_differential import _check
    rng = np.random.default_rng(12)
    T = 60
    edges = []
    nxt, live = T, list(range(T))
    while len(live) > 1:
        a, b = live.pop(int(rng.integers(len(live)))), live.pop(int(rng.integers(len(live))))
        edges += [[nxt, a], [nxt, b]]
        live.append(nxt)
        nxt += 1
    N = nxt
    amb = np.round(rng.random((35, 4)) * 0.9 + 0.05, 4).tolist()
    rows = [[1.0 if j == i else 0.0 for j in range(4)] for i in range(4)] + [[1.0] * 4] + amb
    pa = [[rows[int(rng.integers(40))] if a < T else [1.0] * 4 for a in range(N)] for _ in range(6)]
    x = {"model_and_data": {"edges": edges, "edge_rate_coefficients": (rng.random(N - 1) * 0.2 + 0.01).tolist(),
                            "rate_matrix": [[0, 1, 2, 1], [1, 0, 1, 2], [2, 1, 0, 1], [1, 2, 1, 0]], "probability_array": pa,
                            "rate_divisor": "equilibrium_exit_rate", "root_prior": "uniform_distribution"}}
    s = json.dumps(x)
    for kind in ("ll", "deriv"):
        orc = {"ll": oracle.arbplf_ll, "deriv": oracle.arbplf_deriv}[kind]
        compact, dense = _both(monkeypatch, kind, s)
        want = json.loads(orc(s))
        _check(kind, compact, want)
        _check(kind, dense, want)
