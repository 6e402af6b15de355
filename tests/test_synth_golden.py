"""Independent anchors above k = 4: tests/golden/synth/*.json hold per-site log
likelihoods, edge derivatives and marginals of the BASELINE configurations 2-5 at reduced site counts, computed by
tools/make_synth_fixtures.py with mpmath at 50 digits (eigendecomposition / Taylor transition matrices, incomplete
gamma functions, forward-backward vectors cross-checked by edge substitution) -- nothing shared with the oracle or
the product but the JSON input -- plus closed-form answers of the equal-rates model at k = 20 and 61.

CPU: the oracle against them (1e-13 relative).  GPU: the product through the JSON API (1e-12)."""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN
from phyly_amd import synth

SYNTH = os.path.join(GOLDEN, "synth")


def _inputs(cfg):
    with open(os.path.join(SYNTH, "cfg%d.json" % cfg)) as f:
        fx = json.load(f)
    wl = synth.Workload(cfg)
    codes = np.array(fx["codes"], dtype=np.uint8)
    regenerated = wl.simulate(fx["sites"])
    regenerated[1, 3] = wl.k
    regenerated[5, 7 % fx["sites"]] = wl.k
    assert np.array_equal(codes, regenerated), "the synthetic generator no longer reproduces the fixture's sites"
    md = wl.json_model(codes)
    q_ll = json.dumps({"model_and_data": md})
    q_d = json.dumps({"model_and_data": md, "edge_reduction": {"selection": fx["deriv_edges_user_order"]}})
    q_m = json.dumps({"model_and_data": md, "node_reduction": {"selection": fx["marginal_nodes"]}})
    return fx, q_ll, q_d, q_m


def _check(fx, ll, dv, mv, rel):
    S, ne, nn = fx["sites"], len(fx["deriv_edges_user_order"]), len(fx["marginal_nodes"])
    got_ll = np.array([r[-1] for r in ll["data"]])
    want_ll = np.array(fx["ll"])
    assert np.max(np.abs(got_ll - want_ll) / np.maximum(1.0, np.abs(want_ll))) <= rel
    got_d = np.array([r[-1] for r in dv["data"]]).reshape(S, ne)
    want_d = np.array(fx["deriv"])
    scale = np.max(np.abs(want_d), axis=1, keepdims=True)
    assert np.max(np.abs(got_d - want_d) / np.maximum(np.abs(want_d), scale)) <= rel * 10
    k = len(fx["marginal"][0][0])
    got_m = np.array([r[-1] for r in mv["data"]]).reshape(S, nn, k)
    assert np.max(np.abs(got_m - np.array(fx["marginal"]))) <= rel * 10


@pytest.mark.parametrize("cfg", [2, 3, 4, 5])
def test_oracle_matches_mpmath_fixtures(oracle, cfg):
    fx, q_ll, q_d, q_m = _inputs(cfg)
    _check(fx, json.loads(oracle.arbplf_ll(q_ll)), json.loads(oracle.arbplf_deriv(q_d)), json.loads(oracle.arbplf_marginal(q_m)), 1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [2, 3, 4, 5])
def test_product_matches_mpmath_fixtures(cfg):
    import arbplf
    fx, q_ll, q_d, q_m = _inputs(cfg)
    _check(fx, json.loads(arbplf.arbplf_ll(q_ll)), json.loads(arbplf.arbplf_deriv(q_d)), json.loads(arbplf.arbplf_marginal(q_m)), 1e-12)


def _closed_form_queries(case):
    k, mu, (t1, t2) = case["k"], case["mu"], case["t"]
    Q = [[0.0 if i == j else mu for j in range(k)] for i in range(k)]
    md = {"edges": [[2, 0], [2, 1]], "edge_rate_coefficients": [t1, t2], "rate_matrix": Q,
          "root_prior": "uniform_distribution",
          "character_definitions": np.eye(k).tolist() + [[1.0] * k],
          "character_data": [[3, 3, k], [3, k - 1, k]]}
    return json.dumps({"model_and_data": md}), json.dumps({"model_and_data": md, "edge_reduction": {"selection": [0]}})


def _closed_form_check(case, ll, dv, rel, fp64_site_arithmetic=False):
    got = [r[-1] for r in ll["data"]]
    assert abs(got[0] - case["ll_same"]) <= rel * abs(case["ll_same"])
    assert abs(got[1] - case["ll_diff"]) <= rel * abs(case["ll_diff"])
    d = [r[-1] for r in dv["data"]]
    # long branches: the derivative is e^{-k mu t} ~ 1e-17 .. 1e-23 times O(1), i.e. what is left after the
    # transition probabilities cancel to 1/k; 1e-10 relative on such a value is ~1e-33 absolute
    drel = 10 * rel if case["t"][0] < 10 else 1e-10
    # the product's per-site arithmetic is fp64: the derivative is a sum over root states of terms of size
    # k mu e^{-k mu t1} / k that cancel to first order; what survives is e^{-k mu t2} times smaller (1e-14 at k = 61,
    # t2 = 40), so the result can only be as good as 2^-53 of the cancelling terms
    floor = 4e-16 * case["k"] * case["mu"] * np.exp(-case["k"] * case["mu"] * case["t"][0]) if fp64_site_arithmetic else 0.0
    assert abs(d[0] - case["dll_dt1_same"]) <= drel * abs(case["dll_dt1_same"]) + floor + 1e-300
    assert abs(d[1] - case["dll_dt1_diff"]) <= drel * abs(case["dll_dt1_diff"]) + floor + 1e-300


with open(os.path.join(SYNTH, "closed_form_equal_rates.json")) as _f:
    CLOSED = json.load(_f)


@pytest.mark.parametrize("case", CLOSED, ids=["k%d-t%g" % (c["k"], c["t"][0]) for c in CLOSED])
def test_oracle_equal_rates_closed_form(oracle, case):
    """p_same = 1/k + (k-1)/k e^{-k mu t}, p_diff = 1/k - 1/k e^{-k mu t}: the long-branch case has e^{-k mu t} ~ 1e-17
    (k = 20) and 1e-23 (k = 61), where the derivative is pure cancellation"""
    q_ll, q_d = _closed_form_queries(case)
    _closed_form_check(case, json.loads(oracle.arbplf_ll(q_ll)), json.loads(oracle.arbplf_deriv(q_d)), 1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CLOSED, ids=["k%d-t%g" % (c["k"], c["t"][0]) for c in CLOSED])
def test_product_equal_rates_closed_form(case):
    import arbplf
    q_ll, q_d = _closed_form_queries(case)
    _closed_form_check(case, json.loads(arbplf.arbplf_ll(q_ll)), json.loads(arbplf.arbplf_deriv(q_d)), 1e-12, fp64_site_arithmetic=True)
