"""GPU: several engines in one process behind the JSON drivers and the plk_group C-ABI (include/plk.h).

The GPU box has one device, so the device list names it twice or three times: ARBPLF_DEVICES=0,0,0 gives three engines
(own streams, own buffers, concurrent host threads) that each take a contiguous block of the site patterns.  Results
must equal the single-engine results: per-site values bit for bit (the same kernels on the same sites), aggregated
values to 1e-14 relative (partial sums added in engine order instead of one device-wide reduction)."""
import ctypes
import json
import random

import numpy as np
import pytest

from phyly_amd import synth
from phyly_amd.engine import load_library

pytestmark = pytest.mark.gpu


def _fns():
    import arbplf
    return {"ll": arbplf.arbplf_ll, "deriv": arbplf.arbplf_deriv, "marginal": arbplf.arbplf_marginal,
            "dwell": arbplf.arbplf_dwell, "em_update": arbplf.arbplf_em_update, "hess": arbplf.arbplf_hess}


def _query(kind, wl, S, agg):
    codes = wl.simulate(S)
    x = {"model_and_data": wl.json_model(codes)}
    if agg:
        rng = random.Random(S)
        x["site_reduction"] = {"aggregation": [round(rng.uniform(-1, 2), 3) for _ in range(S)]} if agg == "weights" else {"aggregation": agg}
    if kind == "dwell":
        x["state_reduction"] = {"aggregation": "sum"}
    return json.dumps(x)


@pytest.mark.parametrize("kind,agg", [("ll", None), ("ll", "sum"), ("deriv", None), ("deriv", "weights"), ("marginal", "avg"),
                                      ("marginal", None), ("dwell", "sum"), ("em_update", "sum"), ("hess", "sum")])
@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
def test_drivers_with_several_engines(monkeypatch, kind, agg, devices):
    wl = synth.Workload(T=9, k=4, tree="yule", model="gtr_g4", seed=21)
    S = 301 if kind != "hess" else 40
    s = _query(kind, wl, S, agg)
    fn = _fns()[kind]
    monkeypatch.delenv("ARBPLF_DEVICES", raising=False)
    one = json.loads(fn(s))
    monkeypatch.setenv("ARBPLF_DEVICES", devices)
    many = json.loads(fn(s))
    monkeypatch.delenv("ARBPLF_DEVICES", raising=False)
    assert one["columns"] == many["columns"] and len(one["data"]) == len(many["data"])
    for a, b in zip(one["data"], many["data"]):
        assert a[:-1] == b[:-1]
        if agg is None:
            assert a[-1] == b[-1], (a, b)
        else:
            assert abs(a[-1] - b[-1]) <= 1e-14 * max(abs(a[-1]), 1e-3), (a, b)


def test_more_engines_than_sites(monkeypatch):
    wl = synth.Workload(T=6, k=4, tree="yule", model="hky85", seed=4)
    s = _query("ll", wl, 2, "sum")
    import arbplf
    one = json.loads(arbplf.arbplf_ll(s))
    monkeypatch.setenv("ARBPLF_DEVICES", "0,0,0,0")
    many = json.loads(arbplf.arbplf_ll(s))
    monkeypatch.delenv("ARBPLF_DEVICES", raising=False)
    assert abs(one["data"][0][0] - many["data"][0][0]) <= 1e-14 * abs(one["data"][0][0])


def test_group_cabi_amino_acids_and_blocks():
    """the C-ABI itself, k = 20: ll / deriv sums of a 3-engine group against one engine, and the block map"""
    lib = load_library()
    vp, ci, cl = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
    lib.plk_group_create.argtypes = [ctypes.POINTER(vp), ci, vp]
    lib.plk_group_destroy.argtypes = [vp]
    lib.plk_group_destroy.restype = None
    lib.plk_group_last_error.argtypes = [vp]
    lib.plk_group_last_error.restype = ctypes.c_char_p
    lib.plk_group_block.argtypes = [vp, ci, ctypes.POINTER(cl), ctypes.POINTER(cl)]
    lib.plk_group_set_tree.argtypes = [vp, ci, vp, vp, vp]
    lib.plk_group_set_model.argtypes = [vp, ci, ci, vp, vp, vp, vp, vp, ci, vp]
    lib.plk_group_set_patterns_codes.argtypes = [vp, cl, vp, ci, vp]
    lib.plk_group_set_site_weights.argtypes = [vp, vp]
    lib.plk_group_ll.argtypes = [vp, vp, vp]
    lib.plk_group_deriv.argtypes = [vp, vp, vp, vp]
    wl = synth.Workload(T=14, k=20, tree="yule", model="aa20", seed=8)
    k0 = wl.prepare()
    S = 1000
    codes = np.ascontiguousarray(wl.simulate(S))
    w = np.linspace(0.5, 1.5, S)
    res = {}
    for G in (1, 3):
        g = vp()
        dev = (ci * G)(*([0] * G))
        assert lib.plk_group_create(ctypes.byref(g), G, dev) == 0
        ip, ix, pre = (np.ascontiguousarray(a, dtype=np.int32) for a in (wl.indptr, wl.indices, wl.preorder))
        P = lambda a: a.ctypes.data_as(vp)
        assert lib.plk_group_set_tree(g, wl.N, P(ip), P(ix), P(pre)) == 0
        Qn, Ql, er = (np.ascontiguousarray(a, dtype=np.float64) for a in (k0["Qn"], k0["Qn_lo"], wl.edge_rates_csr))
        cr, cp, pi = (np.ascontiguousarray(a, dtype=np.float64) for a in (k0["cat_rates"], k0["cat_prior"], k0["pi"]))
        assert lib.plk_group_set_model(g, wl.k, k0["C"], P(Qn), P(Ql), P(er), P(cr), P(cp), 4, P(pi)) == 0
        defs = np.ascontiguousarray(wl.defs, dtype=np.float64)
        assert lib.plk_group_set_patterns_codes(g, S, P(codes), wl.nchar, P(defs)) == 0, lib.plk_group_last_error(g)
        assert lib.plk_group_set_site_weights(g, P(w)) == 0
        ll, s2 = np.zeros(S), np.zeros(2)
        assert lib.plk_group_ll(g, P(ll), P(s2)) == 0, lib.plk_group_last_error(g)
        ds = np.zeros((wl.E, 2))
        assert lib.plk_group_deriv(g, None, None, P(ds)) == 0, lib.plk_group_last_error(g)
        res[G] = (ll, s2.sum(), ds.sum(axis=1))
        if G == 3:
            a, b = cl(), cl()
            blocks = []
            for i in range(G):
                assert lib.plk_group_block(g, i, ctypes.byref(a), ctypes.byref(b)) == 0
                blocks.append((a.value, b.value))
            assert blocks == [(0, 334), (334, 668), (668, 1000)]
        lib.plk_group_destroy(g)
    assert np.array_equal(res[1][0], res[3][0])
    assert abs(res[1][1] - res[3][1]) <= 1e-14 * abs(res[1][1])
    assert np.max(np.abs(res[1][2] - res[3][2])) <= 1e-13 * np.max(np.abs(res[1][2]))
