"""CPU: pin the oracle (oracle/) against every ll / deriv / marginal golden vector the
reference holds under examples/ (copied as data into tests/golden/examples), and
against the known-answer values quoted in its READMEs and test_scripts.

This is what makes the oracle trustworthy as the checker of the GPU path."""
import glob
import json
import math
import os

import numpy as np
import pytest

from helpers import GOLDEN, load_json

EX = os.path.join(GOLDEN, "examples")


def _run(fn, path):
    with open(path) as f:
        return json.loads(fn(f.read()))


def _compare(got, exp, rel=1e-14, abs_floor=1e-30):
    assert got["columns"] == exp["columns"]
    assert len(got["data"]) == len(exp["data"])
    for a, b in zip(got["data"], exp["data"]):
        assert a[:-1] == b[:-1]
        assert abs(a[-1] - b[-1]) <= rel * abs(b[-1]) + abs_floor, (a, b)


LL_DIRS = sorted(d for d in glob.glob(os.path.join(EX, "BEAST.*")) if os.path.exists(os.path.join(d, "in.json")))
LL_DIRS += [os.path.join(EX, p) for p in ("BEAST.AncestralState/ll", "Felsenstein.2004.fig.16.4/ll",
                                          "JC.long.branch/ll", "bpp.phyl/ll")]


@pytest.mark.parametrize("d", LL_DIRS, ids=[os.path.relpath(d, EX) for d in LL_DIRS])
def test_ll_golden(oracle, d):
    _compare(_run(oracle.arbplf_ll, os.path.join(d, "in.json")), load_json(os.path.join(d, "out.json")))


def test_ll_readme_values(oracle):
    got = _run(oracle.arbplf_ll, os.path.join(EX, "Felsenstein.2004.fig.16.4/ll/in2.json"))
    assert [r[1] for r in got["data"]] == pytest.approx([0.0, -11.297288182875496, -12.390132492111672], rel=1e-15, abs=1e-30)
    for name, want in (("GeLL.test.likelihood", -2616.073919844292), ("GeLL.driver.DNA", -2616.0735881244163)):
        got = _run(oracle.arbplf_ll, os.path.join(EX, name, "in.json"))
        assert got["columns"] == ["value"]
        assert got["data"][0][0] == pytest.approx(want, rel=1e-15)


@pytest.mark.parametrize("d", ["Felsenstein.2004.fig.16.4/deriv", "bpp.phyl/deriv", "JC.long.branch/deriv"])
def test_deriv_golden(oracle, d):
    _compare(_run(oracle.arbplf_deriv, os.path.join(EX, d, "in.json")), load_json(os.path.join(EX, d, "out.json")))


@pytest.mark.parametrize("f,want", [("jc29.same", -6.4467380574161446e-17), ("jc29.diff", 2.1489126858053815e-17),
                                    ("jc30.same", -1.6993417021166355e-17), ("jc30.diff", 5.6644723403887852e-18),
                                    ("jc600.same", 0.0)])
def test_jc_long_branch_readme(oracle, f, want):
    path = os.path.join(EX, "JC.long.branch", f + ".json")
    got = _run(oracle.arbplf_deriv, path)
    assert abs(got["data"][0][2] - want) <= 1e-14 * abs(want) + 1e-30
    # the closed forms quoted in the README
    t = float(f[2:4]) if f[2:4].isdigit() and not f.startswith("jc600") else 600.0
    if t < 100:
        closed = -4 / (math.exp(4 * t / 3) + 3) if f.endswith("same") else 4 / (3 * math.exp(4 * t / 3) - 3)
        assert abs(got["data"][0][2] - closed) <= 1e-13 * abs(closed)
    ll = _run(oracle.arbplf_ll, path)
    assert ll["data"][0][1] == pytest.approx(-2.7725887222397811, rel=1e-15)


@pytest.mark.parametrize("d", ["Felsenstein.2004.fig.16.4/marginal", "BEAST.AncestralState/marginal",
                               "JC.long.branch/marginal"])
def test_marginal_golden(oracle, d):
    _compare(_run(oracle.arbplf_marginal, os.path.join(EX, d, "in.json")), load_json(os.path.join(EX, d, "out.json")))


def test_gamma_discretization_known_values(oracle):
    """test_scripts/test_gamma_discretization.py of the reference: Gamma(0.5, 4 categories)"""
    r, p = oracle.gamma_mixture(3, 4, 0.5)
    np.testing.assert_allclose(r, [0.0333877533835995, 0.251915917593438, 0.820268481973649, 2.89442784704931], rtol=1e-14)
    np.testing.assert_allclose(p, [0.25] * 4, rtol=0)
    assert abs(float(np.dot(r, p)) - 1.0) < 1e-15
    # +I: rates scaled by 1/(1-p), invariable category appended last
    r2, p2 = oracle.gamma_mixture(3, 4, 0.5, 0.3)
    np.testing.assert_allclose(r2[:4], r / 0.7, rtol=1e-14)
    assert r2[4] == 0.0 and p2[4] == 0.3
    np.testing.assert_allclose(p2[:4], [0.175] * 4, rtol=1e-15)
    # tiny shape: all mass near zero except the last class
    r3, _ = oracle.gamma_mixture(3, 4, 1e-6)
    np.testing.assert_allclose(r3, [0, 0, 0, 4], atol=1e-4)
    # median variant is normalised to mean 1
    r4, _ = oracle.gamma_mixture(4, 4, 0.7)
    assert abs(r4.mean() - 1.0) < 1e-15 and np.all(np.diff(r4) > 0)


def test_three_precisions_agree(oracle):
    """double port (timed baseline), long double and binary128 evaluators agree"""
    md = load_json(os.path.join(EX, "BEAST.GTRGI", "in.json"))["model_and_data"]
    m = oracle.parse_model(md)
    w = oracle.prepare(m)
    a, _ = oracle.site_ll(m, w, B=m.B, precise=0)
    b, _ = oracle.site_ll(m, w, B=m.B, precise=1)
    c, _ = oracle.site_ll(m, w, B=m.B, precise=2)
    assert np.max(np.abs(a - b) / np.abs(b)) < 1e-14
    assert np.max(np.abs(c - b) / np.abs(b)) < 1e-15
    d1 = oracle.site_deriv(m, w, m.B[:8], precise=1)
    d2 = oracle.site_deriv(m, w, m.B[:8], precise=2)
    assert np.max(np.abs(d1 - d2)) <= 1e-13 * np.max(np.abs(d2))


def test_path_closed_forms(oracle):
    """test_scripts/test_path_exponential_absorbing.py: 2-state absorbing path,
    ll = log(1 - e^-T), deriv = 1/expm1(T), marginal = [e^-u, -expm1(-u)]"""
    rates = [0.3, 0.5, 0.7]
    T = sum(rates)
    x = {"model_and_data": {
        "edges": [[0, 1], [1, 2], [2, 3]],
        "edge_rate_coefficients": rates,
        "rate_matrix": [[0, 1], [0, 0]],
        "probability_array": [[[1, 0], [1, 1], [1, 1], [0, 1]]]}}
    ll = json.loads(oracle.arbplf_ll(json.dumps(x)))["data"][0][1]
    assert ll == pytest.approx(math.log(-math.expm1(-T)), rel=1e-15)
    d = json.loads(oracle.arbplf_deriv(json.dumps(dict(x, edge_reduction={"aggregation": "avg"}))))
    assert d["data"][0][1] == pytest.approx(1 / math.expm1(T), rel=1e-14)
    # unconditional marginals with no observation at the far end
    y = json.loads(json.dumps(x))
    y["model_and_data"]["probability_array"] = [[[1, 0], [1, 1], [1, 1], [1, 1]]]
    mg = json.loads(oracle.arbplf_marginal(json.dumps(y)))["data"]
    u = rates[0]
    got = {(r[1], r[2]): r[3] for r in mg}
    assert got[(1, 0)] == pytest.approx(math.exp(-u), rel=1e-15)
    assert got[(1, 1)] == pytest.approx(-math.expm1(-u), rel=1e-15)


# ---- dwell / trans / em-update: every golden the reference holds ----
def _pairs(kind_dirs):
    out = []
    for d in kind_dirs:
        for inp in sorted(glob.glob(os.path.join(EX, d, "in*.json"))):
            out.append((inp, os.path.join(os.path.dirname(inp), "out" + os.path.basename(inp)[2:])))
    return out


DWELL = _pairs(["Felsenstein.2004.fig.16.4/dwell/adenine", "Felsenstein.2004.fig.16.4/dwell/pyrimidines",
                "BEAST.MarkovJumps/MarkovRewardsC"])
TRANS = _pairs(["Felsenstein.2004.fig.16.4/trans/A.to.C.only", "Felsenstein.2004.fig.16.4/trans/all.types",
                "Felsenstein.2004.fig.16.4/trans/transversions.only", "BEAST.MarkovJumps/MarkovJumpsC",
                "BEAST.MarkovJumps/MarkovMarginalRate"])
EM = _pairs(["Felsenstein.2004.fig.16.4/em-update/with.full.data", "Felsenstein.2004.fig.16.4/em-update/with.leaf.data",
             "Felsenstein.2004.fig.16.4/em-update/with.no.data"])


@pytest.mark.parametrize("inp,outp", DWELL, ids=[os.path.relpath(p[0], EX) for p in DWELL])
def test_dwell_golden(oracle, inp, outp):
    _compare(_run(oracle.arbplf_dwell, inp), load_json(outp), rel=2e-15)


@pytest.mark.parametrize("inp,outp", TRANS, ids=[os.path.relpath(p[0], EX) for p in TRANS])
def test_trans_golden(oracle, inp, outp):
    _compare(_run(oracle.arbplf_trans, inp), load_json(outp), rel=2e-15)


@pytest.mark.parametrize("inp,outp", EM, ids=[os.path.relpath(p[0], EX) for p in EM])
def test_em_update_golden(oracle, inp, outp):
    _compare(_run(oracle.arbplf_em_update, inp), load_json(outp), rel=2e-15)


def test_markov_marginal_rate_known_answer(oracle):
    """examples/BEAST.MarkovJumps/MarkovMarginalRate/README.md: the expectation is 12/199"""
    got = _run(oracle.arbplf_trans, os.path.join(EX, "BEAST.MarkovJumps/MarkovMarginalRate/in.json"))
    assert got["data"][0][-1] == pytest.approx(12 / 199, rel=1e-15)


def test_dwell_rewards_sum_to_one(oracle):
    """examples/BEAST.MarkovJumps/MarkovRewardsC: a reward of 1 for every state gives 1 on every branch"""
    got = _run(oracle.arbplf_dwell, os.path.join(EX, "BEAST.MarkovJumps/MarkovRewardsC/in.json"))
    assert all(abs(r[-1] - 1.0) <= 1e-15 for r in got["data"])


HESS = _pairs(["Felsenstein.2004.fig.16.4/hess/with.full.data", "Felsenstein.2004.fig.16.4/hess/with.leaf.data",
               "Felsenstein.2004.fig.16.4/hess/with.no.data"])


@pytest.mark.parametrize("inp,outp", HESS, ids=[os.path.relpath(p[0], EX) for p in HESS])
def test_hess_golden(oracle, inp, outp):
    _compare(_run(oracle.arbplf_hess, inp), load_json(outp), rel=2e-15)


def test_hess_is_derivative_of_deriv(oracle):
    """the Hessian row of edge j is the central finite difference of the oracle's own gradient (8-digit check)"""
    inp = os.path.join(EX, "Felsenstein.2004.fig.16.4/hess/with.leaf.data/in.json")
    x = load_json(inp)
    H = np.array([r[2] for r in _run(oracle.arbplf_hess, inp)["data"]]).reshape(7, 7)
    rates = x["model_and_data"]["edge_rate_coefficients"]
    for j in (0, 3, 6):
        h = 1e-6 * rates[j]
        g = []
        for sgn in (+1, -1):
            y = json.loads(json.dumps(x))
            y["model_and_data"]["edge_rate_coefficients"][j] = rates[j] + sgn * h
            y["edge_reduction"] = {}
            g.append(np.array([r[-1] for r in json.loads(oracle.arbplf_deriv(json.dumps(y)))["data"]]))
        fd = (g[0] - g[1]) / (2 * h)
        assert np.max(np.abs(fd - H[j]) / np.maximum(np.abs(H[j]), 1.0)) < 1e-6
