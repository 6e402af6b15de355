"""GPU: cases restated from the reference's test_scripts/ (same inputs, same
expectations), run through the product's JSON string API.

Where the reference asserts bit-for-bit equality (it returns correctly rounded
values) this fp64 build asserts agreement to 1e-13 relative."""
import copy
import json
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _f(kind):
    import arbplf
    return {"ll": arbplf.arbplf_ll, "deriv": arbplf.arbplf_deriv, "marginal": arbplf.arbplf_marginal}[kind]


def run(kind, d):
    return json.loads(_f(kind)(json.dumps(d)))


def values(out):
    return np.array([r[-1] for r in out["data"]])


def close(a, b, rel=1e-13, floor=1e-300):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape
    assert np.all(np.abs(a - b) <= rel * np.maximum(np.abs(a), np.abs(b)) + floor), (a, b)


# ----------------------------------------------------------------- test_ll.py
LL_IN = {"model_and_data": {
    "edges": [[5, 0], [5, 1], [5, 6], [6, 2], [6, 7], [7, 3], [7, 4]],
    "edge_rate_coefficients": [0.01, 0.2, 0.15, 0.3, 0.05, 0.3, 0.02],
    "rate_matrix": [[0, .3, .4, .5], [.3, 0, .3, .3], [.3, .6, 0, .3], [.3, .3, .3, 0]],
    "probability_array": [
        [[1, 0, 0, 0], [0, 1, 0, 0], [0, 1, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0.25, 0.25, 0.25, 0.25], [1, 1, 1, 1], [1, 1, 1, 1]],
        [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 1, 0, 0], [0, 0, 1, 0], [0.25, 0.25, 0.25, 0.25], [1, 1, 1, 1], [1, 1, 1, 1]]]}}


def test_ll_selection_and_aggregation_consistency():
    base = run("ll", LL_IN)
    assert base["columns"] == ["site", "value"] and [r[0] for r in base["data"]] == [0, 1]
    lls = values(base)
    assert run("ll", dict(LL_IN, site_reduction={})) == base
    rev = run("ll", dict(LL_IN, site_reduction={"selection": [1, 0]}))
    assert [r[0] for r in rev["data"]] == [1, 0]
    close(values(rev), lls[::-1])
    for site in (0, 1):
        only = run("ll", dict(LL_IN, site_reduction={"selection": [site], "aggregation": "only"}))
        assert only["columns"] == ["value"]
        for agg in ("sum", "avg"):
            close(values(run("ll", dict(LL_IN, site_reduction={"selection": [site], "aggregation": agg}))), values(only))
        close(values(only), [lls[site]])
    close(values(run("ll", dict(LL_IN, site_reduction={"aggregation": "sum"}))), [lls.sum()])
    close(values(run("ll", dict(LL_IN, site_reduction={"aggregation": "avg"}))), [lls.mean()])
    close(values(run("ll", dict(LL_IN, site_reduction={"aggregation": [2.5, -1]}))), [2.5 * lls[0] - lls[1]])
    dup = run("ll", dict(LL_IN, site_reduction={"selection": [1, 1, 0], "aggregation": [1, 2, 3]}))
    close(values(dup), [3 * lls[1] + 3 * lls[0]])


# ----------------------------------------------------------------- test_ll_deriv.py
def test_deriv_aggregation_selection_and_edge_permutation():
    x = copy.deepcopy(LL_IN)
    full = run("deriv", x)
    assert full["columns"] == ["site", "edge", "value"] and len(full["data"]) == 14
    D = values(full).reshape(2, 7)
    close(values(run("deriv", dict(x, site_reduction={"aggregation": "sum"}))), D.sum(axis=0))
    close(values(run("deriv", dict(x, edge_reduction={"aggregation": "avg"}))), D.mean(axis=1))
    close(values(run("deriv", dict(x, site_reduction={"aggregation": "sum"}, edge_reduction={"aggregation": "sum"}))), [D.sum()])
    sel = run("deriv", dict(x, edge_reduction={"selection": [4, 4, 0]}))
    assert [r[1] for r in sel["data"]] == [4, 4, 0, 4, 4, 0]
    close(values(sel).reshape(2, 3), D[:, [4, 4, 0]])
    w = run("deriv", dict(x, site_reduction={"aggregation": [1, -1]}, edge_reduction={"selection": [1, 2, 1], "aggregation": [1, 1, -1]}))
    close(values(w), [(D[0] - D[1])[2]], rel=1e-12)
    # permuting the edge list permutes the derivatives (test_ll_deriv.py:332-389)
    perm = [3, 0, 6, 1, 5, 2, 4]
    y = copy.deepcopy(x)
    y["model_and_data"]["edges"] = [x["model_and_data"]["edges"][i] for i in perm]
    y["model_and_data"]["edge_rate_coefficients"] = [x["model_and_data"]["edge_rate_coefficients"][i] for i in perm]
    Dp = values(run("deriv", y)).reshape(2, 7)
    close(Dp, D[:, perm], rel=1e-12)


def test_deriv_matches_finite_differences():
    """test_ll_deriv_accuracy.py: gradient vs central differences, rtol 1e-3, over mixtures x root priors"""
    base = copy.deepcopy(LL_IN)
    base["site_reduction"] = {"aggregation": "sum"}
    for mix in (None, {"rates": [0.5, 2.0], "prior": [0.25, 0.75]}):
        for root in (None, "equilibrium_distribution", "uniform_distribution", [0.1, 0.2, 0.3, 0.4]):
            x = copy.deepcopy(base)
            if mix:
                x["model_and_data"]["rate_mixture"] = mix
            if root:
                x["model_and_data"]["root_prior"] = root
            g = values(run("deriv", x))
            for e in range(7):
                h = 1e-6
                lo, hi = copy.deepcopy(x), copy.deepcopy(x)
                lo["model_and_data"]["edge_rate_coefficients"][e] -= h
                hi["model_and_data"]["edge_rate_coefficients"][e] += h
                fd = (values(run("ll", hi))[0] - values(run("ll", lo))[0]) / (2 * h)
                assert abs(fd - g[e]) <= 1e-3 * max(1.0, abs(g[e]))


# ----------------------------------------------------------------- closed forms
def test_pure_birth_path_exact_values():
    """test_marginal_no_change.py: ll exactly -2 / -3, marginals exactly 1.0 / 0.0"""
    d = {"model_and_data": {"edges": [[0, 1], [1, 2]], "edge_rate_coefficients": [1, 1],
                            "rate_matrix": [[0, 1], [0, 0]], "probability_array": [[[1, 0], [1, 1], [1, 0]]]},
         "site_reduction": {"aggregation": "only"}}
    mg = run("marginal", d)
    assert mg["columns"] == ["node", "state", "value"]
    close(values(mg), [1, 0, 1, 0, 1, 0], rel=0, floor=1e-15)
    close(values(run("ll", d)), [-2.0], rel=1e-15)
    d["model_and_data"]["edge_rate_coefficients"] = [1, 2]
    close(values(run("ll", d)), [-3.0], rel=1e-15)
    close(values(run("marginal", d)), [1, 0, 1, 0, 1, 0], rel=0, floor=1e-15)


def test_exponential_absorbing_path():
    """test_path_exponential_absorbing.py: ll = log(1-e^-T), d ll / d rate = 1/expm1(T), marginals [e^-u, -expm1(-u)]"""
    rates = [0.3, 0.5, 0.7]
    T = sum(rates)
    x = {"model_and_data": {"edges": [[0, 1], [1, 2], [2, 3]], "edge_rate_coefficients": rates,
                            "rate_matrix": [[0, 1], [0, 0]],
                            "probability_array": [[[1, 0], [1, 1], [1, 1], [0, 1]]]}}
    close(values(run("ll", x)), [math.log(-math.expm1(-T))], rel=1e-13)
    close(values(run("deriv", x)), [1 / math.expm1(T)] * 3, rel=1e-12)
    y = copy.deepcopy(x)
    y["model_and_data"]["probability_array"] = [[[1, 0], [1, 1], [1, 1], [1, 1]]]
    mg = {(r[1], r[2]): r[3] for r in run("marginal", y)["data"]}
    u = 0.0
    for node, r in zip((1, 2, 3), rates):
        u += r
        close([mg[(node, 0)], mg[(node, 1)]], [math.exp(-u), -math.expm1(-u)], rel=1e-13)


# ----------------------------------------------------------------- invariances
TWO_STATE = {"model_and_data": {
    "edges": [[0, 1], [0, 2], [0, 3]], "edge_rate_coefficients": [0.28, 0.11, 0.59],
    "rate_matrix": [[0, 3], [1, 0]],
    "probability_array": [[[0.25, 0.75], [1, 0], [0, 1], [1, 0]], [[0.25, 0.75], [0, 1], [0, 1], [0, 1]],
                          [[0.25, 0.75], [1, 1], [1, 0], [0, 1]]]},
    "site_reduction": {"aggregation": [1, 2.5, 0.5]}}


@pytest.mark.parametrize("kind", ["ll", "deriv", "marginal"])
def test_rate_divisor_invariances(kind):
    """test_rate_divisor.py: Q/100 == 3Q/300; the diagonal is ignored; equilibrium_exit_rate == 1.5 here"""
    a = copy.deepcopy(TWO_STATE)
    a["model_and_data"]["rate_divisor"] = 100
    b = copy.deepcopy(TWO_STATE)
    b["model_and_data"]["rate_matrix"] = [[0, 9], [3, 0]]
    b["model_and_data"]["rate_divisor"] = 300
    close(values(run(kind, a)), values(run(kind, b)))
    c = copy.deepcopy(a)
    c["model_and_data"]["rate_matrix"] = [[42, 3], [1, 42]]
    close(values(run(kind, a)), values(run(kind, c)))
    d = copy.deepcopy(TWO_STATE)
    d["model_and_data"]["rate_divisor"] = 1.5          # pi = (1/4, 3/4): exit rate 1/4*3 + 3/4*1
    e = copy.deepcopy(TWO_STATE)
    e["model_and_data"]["rate_matrix"] = [[0, 9], [3, 0]]
    e["model_and_data"]["rate_divisor"] = "equilibrium_exit_rate"
    # 3Q with its own exit rate 4.5 == Q with divisor 1.5
    close(values(run(kind, d)), values(run(kind, e)))


@pytest.mark.parametrize("kind", ["ll", "deriv", "marginal"])
def test_root_prior_equivalences(kind):
    """test_root_prior.py: a prior folded into the root's probability row == the same explicit root_prior"""
    a = copy.deepcopy(TWO_STATE)                       # 0.25 / 0.75 sits in the root rows
    b = copy.deepcopy(TWO_STATE)
    for site in b["model_and_data"]["probability_array"]:
        site[0] = [1, 1]
    b["model_and_data"]["root_prior"] = [0.25, 0.75]
    c = copy.deepcopy(b)
    c["model_and_data"]["root_prior"] = "equilibrium_distribution"
    va = values(run(kind, a))
    close(va, values(run(kind, b)))
    close(va, values(run(kind, c)), rel=1e-12)
    u = copy.deepcopy(b)
    u["model_and_data"]["root_prior"] = "uniform_distribution"
    v = copy.deepcopy(b)
    v["model_and_data"]["root_prior"] = [0.5, 0.5]
    close(values(run(kind, u)), values(run(kind, v)))


@pytest.mark.parametrize("kind", ["ll", "deriv"])
def test_site_weights_only_their_sums_matter(kind):
    """test_site_weights.py: duplicated sites with split weights == one site with the summed weight"""
    a = copy.deepcopy(TWO_STATE)
    b = copy.deepcopy(TWO_STATE)
    pa = b["model_and_data"]["probability_array"]
    b["model_and_data"]["probability_array"] = [pa[0], pa[1], pa[1], pa[2], pa[0]]
    b["site_reduction"] = {"aggregation": [0.25, 2.0, 0.5, 0.5, 0.75]}
    close(values(run(kind, a)), values(run(kind, b)), rel=1e-12)


def test_gamma_mixture_equals_explicit_rates():
    """test_gamma_discretization.py: Gamma(0.5, 4) == custom rates with uniform prior; +I 0.3; tiny shape"""
    rates = [0.0333877533835995, 0.251915917593438, 0.820268481973649, 2.89442784704931]
    g = copy.deepcopy(LL_IN)
    g["model_and_data"]["gamma_rate_mixture"] = {"gamma_shape": 0.5, "gamma_categories": 4}
    c = copy.deepcopy(LL_IN)
    c["model_and_data"]["rate_mixture"] = {"rates": rates, "prior": "uniform_distribution"}
    for kind in ("ll", "deriv", "marginal"):
        close(values(run(kind, g)), values(run(kind, c)), rel=1e-7, floor=1e-12)
    gi = copy.deepcopy(LL_IN)
    gi["model_and_data"]["gamma_rate_mixture"] = {"gamma_shape": 0.5, "gamma_categories": 4, "invariable_prior": 0.3}
    ci = copy.deepcopy(LL_IN)
    ci["model_and_data"]["rate_mixture"] = {"rates": [r / 0.7 for r in rates] + [0], "prior": [0.175] * 4 + [0.3]}
    close(values(run("ll", gi)), values(run("ll", ci)), rel=1e-7)
    t = copy.deepcopy(LL_IN)
    t["model_and_data"]["gamma_rate_mixture"] = {"gamma_shape": 1e-6, "gamma_categories": 4}
    e = copy.deepcopy(LL_IN)
    e["model_and_data"]["rate_mixture"] = {"rates": [0, 4], "prior": [0.75, 0.25]}
    close(values(run("ll", t)), values(run("ll", e)), rel=1e-4)


def test_mixture_equals_block_diagonal_model():
    """test_rate_mixture_vs_block.py / test_marginal.py:216-263: a 2-category mixture of a 4-state model ==
    one 8-state block-diagonal model whose root prior carries the category weights"""
    md = LL_IN["model_and_data"]
    Q = np.array(md["rate_matrix"])
    r, p = [0.5, 2.0], [0.3, 0.7]
    m = copy.deepcopy(LL_IN)
    m["model_and_data"]["rate_mixture"] = {"rates": r, "prior": p}
    for site in m["model_and_data"]["probability_array"]:
        site[5] = [1, 1, 1, 1]
    m["model_and_data"]["root_prior"] = [0.25] * 4
    b = copy.deepcopy(m)
    del b["model_and_data"]["rate_mixture"]
    Q8 = np.zeros((8, 8))
    Q8[:4, :4] = Q * r[0]
    Q8[4:, 4:] = Q * r[1]
    b["model_and_data"]["rate_matrix"] = Q8.tolist()
    b["model_and_data"]["probability_array"] = [[row + row for row in site] for site in m["model_and_data"]["probability_array"]]
    b["model_and_data"]["root_prior"] = [0.25 * p[0]] * 4 + [0.25 * p[1]] * 4
    close(values(run("ll", m)), values(run("ll", b)), rel=1e-12)
    mm = values(run("marginal", dict(m, site_reduction={"aggregation": "sum"}))).reshape(8, 4)
    mb = values(run("marginal", dict(b, site_reduction={"aggregation": "sum"}))).reshape(8, 8)
    close(mm, mb[:, :4] + mb[:, 4:], rel=1e-11, floor=1e-14)


def test_marginal_via_likelihood_identity():
    """test_marginal.py:129-160: P(node a in state i | data) = L(data, a=i) / L(data)"""
    x = copy.deepcopy(LL_IN)
    x["model_and_data"]["root_prior"] = "uniform_distribution"
    mg = {(r[0], r[1], r[2]): r[3] for r in run("marginal", x)["data"]}
    ll = values(run("ll", x))
    for node in (5, 6, 7, 2):
        for state in range(4):
            y = copy.deepcopy(x)
            for s in range(2):
                row = y["model_and_data"]["probability_array"][s][node]
                y["model_and_data"]["probability_array"][s][node] = [row[j] if j == state else 0 for j in range(4)]
            try:
                lly = values(run("ll", y))
            except RuntimeError:
                continue            # likelihood exactly zero for this constraint: refused
            for s in range(2):
                close([mg[(s, node, state)]], [math.exp(lly[s] - ll[s])], rel=1e-11)


def test_zero_rate_category_and_degenerate_mixtures():
    """test_marginal.py:162-178: a rate-0 category gives P = I exactly; degenerate mixtures agree"""
    x = copy.deepcopy(LL_IN)
    x["model_and_data"]["rate_mixture"] = {"rates": [1.0], "prior": [1.0]}
    for kind in ("ll", "marginal"):
        close(values(run(kind, x)), values(run(kind, LL_IN)))
    y = copy.deepcopy(LL_IN)
    y["model_and_data"]["rate_mixture"] = {"rates": [1.0, 1.0, 1.0], "prior": [0.2, 0.3, 0.5]}
    close(values(run("ll", y)), values(run("ll", LL_IN)), rel=1e-13)
    z = copy.deepcopy(LL_IN)          # all data identical at every leaf: survives a rate-0 category
    for site in z["model_and_data"]["probability_array"]:
        for n in range(5):
            site[n] = [1, 0, 0, 0]
    z["model_and_data"]["rate_mixture"] = {"rates": [0.0, 1.0], "prior": [0.5, 0.5]}
    out = run("marginal", dict(z, site_reduction={"selection": [0]}, node_reduction={"selection": [6]}))
    assert abs(sum(values(out)) - 1.0) < 1e-13


def test_out_of_scope_functions_raise():
    import arbplf
    for name in ("arbplf_hess", "arbplf_dwell", "arbplf_trans", "arbplf_em_update", "arbplf_newton_refine"):
        with pytest.raises(RuntimeError):
            getattr(arbplf, name)("{}")


def test_output_format_is_what_jansson_prints():
    import arbplf
    s = arbplf.arbplf_ll(json.dumps(dict(LL_IN, site_reduction={"aggregation": "sum"})))
    assert s.startswith('{"columns": ["value"], "data": [[') and s.endswith("]]}")
    s2 = arbplf.arbplf_marginal(json.dumps({
        "model_and_data": {"edges": [[0, 1]], "edge_rate_coefficients": [1], "rate_matrix": [[0, 1], [0, 0]],
                           "probability_array": [[[1, 0], [1, 0]]]}}))
    assert '[0, 0, 0, 1.0]' in s2 and '[0, 0, 1, 0.0]' in s2     # integral doubles keep ".0"
