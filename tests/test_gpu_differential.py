"""GPU: differential tests of the whole product path (JSON string API -> host C -> HIP
engine) against the oracle on seeded random inputs: random rooted trees (with
multifurcations and data on internal nodes), k in {2..8}, every root-prior /
rate-divisor / rate-mixture form, both observation forms, and random
selections / aggregations (duplicates, weights of either sign, avg, only).

Tolerances as in BASELINE.md: ll |d| <= 1e-12*max(1,|ll|); deriv / marginal
|d| <= 1e-12*max(|value|, row scale) with an absolute floor of 1e-14 times the
largest value in the table (cancelling weights)."""
import json
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def random_tree(rng, n_nodes):
    """random rooted tree on nodes 0..n-1 with shuffled labels; edges parent->child in random order"""
    labels = list(range(n_nodes))
    rng.shuffle(labels)
    edges = []
    for i in range(1, n_nodes):
        parent = rng.randrange(0, i) if rng.random() < 0.7 else max(0, i - 1 - rng.randrange(0, min(i, 3)))
        edges.append([labels[parent], labels[i]])
    rng.shuffle(edges)
    return edges


def random_model(rng, kind):
    n_nodes = rng.randrange(2, 14)
    k = rng.choice([2, 3, 4, 4, 4, 5, 8])
    S = rng.randrange(1, 9)
    edges = random_tree(rng, n_nodes)
    md = {"edges": edges,
          "edge_rate_coefficients": [rng.choice([0.0, 0.01, 0.1, 0.5, 1.0, 2.5]) * rng.random() for _ in edges]}
    Q = [[0 if i == j else rng.choice([0, 0.2, 1, 1.5, 3]) * rng.random() + (0.05 if rng.random() < 0.8 else 0)
          for j in range(k)] for i in range(k)]
    for i in range(k):
        Q[i][i] = rng.choice([0, 7.5])          # diagonal must be ignored
        if sum(Q[i][j] for j in range(k) if j != i) == 0:
            Q[i][(i + 1) % k] = 0.3             # keep the chain irreducible enough for pi
    md["rate_matrix"] = Q
    if rng.random() < 0.5:
        nchar = k + 1 + rng.randrange(0, 3)
        defs = [[1.0 if j == c else 0.0 for j in range(k)] for c in range(k)] + [[1.0] * k]
        while len(defs) < nchar:
            defs.append([rng.choice([0, 1, 0.5]) for _ in range(k)])
            if sum(defs[-1]) == 0:
                defs[-1][0] = 1
        md["character_definitions"] = defs
        md["character_data"] = [[rng.randrange(0, nchar) if rng.random() < 0.6 else k for _ in range(n_nodes)]
                                for _ in range(S)]
    else:
        pa = []
        for _ in range(S):
            site = []
            for _ in range(n_nodes):
                r = rng.random()
                if r < 0.4:
                    site.append([1] * k)
                elif r < 0.8:
                    row = [0] * k
                    row[rng.randrange(k)] = 1
                    site.append(row)
                else:
                    site.append([round(rng.random(), 3) + 0.01 for _ in range(k)])
            pa.append(site)
        md["probability_array"] = pa
    r = rng.random()
    if r < 0.3:
        md["rate_divisor"] = "equilibrium_exit_rate"
    elif r < 0.6:
        md["rate_divisor"] = rng.choice([0.5, 3, 100])
    r = rng.random()
    if r < 0.25:
        md["root_prior"] = "equilibrium_distribution"
    elif r < 0.5:
        md["root_prior"] = "uniform_distribution"
    elif r < 0.75:
        w = [rng.random() + 0.01 for _ in range(k)]
        md["root_prior"] = [x / sum(w) for x in w]
    r = rng.random()
    if r < 0.2:
        md["gamma_rate_mixture"] = {"gamma_shape": rng.choice([0.3, 1.0, 2.5]), "gamma_categories": rng.randrange(1, 5)}
        if rng.random() < 0.5:
            md["gamma_rate_mixture"]["invariable_prior"] = 0.2
    elif r < 0.35:
        md["normalized_median_gamma_rate_mixture"] = {"gamma_shape": rng.choice([0.5, 1.7]), "gamma_categories": rng.randrange(2, 5)}
    elif r < 0.5:
        n = rng.randrange(1, 4)
        md["rate_mixture"] = {"rates": [rng.choice([0, 0.5, 1, 2.0]) for _ in range(n)],
                              "prior": "uniform_distribution" if rng.random() < 0.4 else [1.0 / n] * n}
        if sum(md["rate_mixture"]["rates"]) == 0:
            md["rate_mixture"]["rates"][0] = 1.0
    x = {"model_and_data": md}

    def reduction(n):
        r = rng.random()
        if r < 0.25:
            return None
        red = {}
        if rng.random() < 0.6:
            red["selection"] = [rng.randrange(n) for _ in range(rng.randrange(1, n + 3))]
        m = len(red.get("selection", range(n)))
        r = rng.random()
        if r < 0.25:
            red["aggregation"] = "sum"
        elif r < 0.45:
            red["aggregation"] = "avg"
        elif r < 0.7:
            red["aggregation"] = [round(rng.uniform(-2, 3), 3) for _ in range(m)]
        elif r < 0.8 and m == 1:
            red["aggregation"] = "only"
        return red

    axes = {"ll": [("site_reduction", S)],
            "deriv": [("site_reduction", S), ("edge_reduction", n_nodes - 1)],
            "marginal": [("site_reduction", S), ("node_reduction", n_nodes), ("state_reduction", k)]}[kind]
    for name, n in axes:
        red = reduction(n)
        if red is not None:
            x[name] = red
    return x


def _check(kind, got, want):
    assert got["columns"] == want["columns"]
    assert len(got["data"]) == len(want["data"])
    table_scale = max([abs(r[-1]) for r in want["data"]] + [0.0])
    for a, b in zip(got["data"], want["data"]):
        assert a[:-1] == b[:-1]
        if kind == "ll":
            tol = 1e-12 * max(1.0, abs(b[-1]))
        else:
            tol = 1e-12 * abs(b[-1]) + 1e-14 * max(table_scale, 1.0 if kind == "marginal" else 0.0) + 1e-300
        assert abs(a[-1] - b[-1]) <= tol, (a, b)


@pytest.mark.parametrize("kind", ["ll", "deriv", "marginal"])
@pytest.mark.parametrize("form", ["character_data", "probability_array"])
def test_duplicate_sites_pattern_compression(oracle, kind, form):
    """the host layer evaluates each distinct site pattern once: an alignment made of many
    copies of a few patterns must give the same table as the oracle, with and without aggregation"""
    import arbplf
    prod = {"ll": arbplf.arbplf_ll, "deriv": arbplf.arbplf_deriv, "marginal": arbplf.arbplf_marginal}[kind]
    orc = {"ll": oracle.arbplf_ll, "deriv": oracle.arbplf_deriv, "marginal": oracle.arbplf_marginal}[kind]
    rng = random.Random(99)
    base = None
    while base is None:
        x = random_model(rng, kind)
        md = x["model_and_data"]
        if (form in md) and len(md[form]) >= 3:
            probe = {"model_and_data": dict(md, **{form: md[form][:3]})}
            if all(np.isfinite(r[-1]) for r in json.loads(oracle.arbplf_ll(json.dumps(probe)))["data"]):
                base = x
    md = base["model_and_data"]
    pats = md[form][:3]
    order = [rng.randrange(3) for _ in range(40)]
    md[form] = [pats[i] for i in order]
    for red in (None, {"aggregation": "sum"}, {"selection": [5, 5, 17, 30, 2], "aggregation": [1.5, -0.5, 2, 1, 1]},
                {"selection": [39, 0, 7, 7]}):
        y = {k: v for k, v in base.items() if k != "site_reduction"}
        if red is not None:
            y["site_reduction"] = red
        s = json.dumps(y)
        want = json.loads(orc(s))
        if any(not np.isfinite(r[-1]) for r in want["data"]):
            pytest.skip("zero-likelihood pattern drawn")
        _check(kind, json.loads(prod(s)), want)


@pytest.mark.parametrize("kind", ["ll", "deriv", "marginal"])
def test_random_inputs_match_oracle(oracle, kind):
    import arbplf
    prod = {"ll": arbplf.arbplf_ll, "deriv": arbplf.arbplf_deriv, "marginal": arbplf.arbplf_marginal}[kind]
    orc = {"ll": oracle.arbplf_ll, "deriv": oracle.arbplf_deriv, "marginal": oracle.arbplf_marginal}[kind]
    rng = random.Random({"ll": 11, "deriv": 22, "marginal": 33}[kind])
    done = skipped = 0
    for case in range(46):
        x = random_model(rng, kind)
        s = json.dumps(x)
        want = json.loads(orc(s))
        vals = [r[-1] for r in want["data"]]
        if any(not np.isfinite(v) for v in vals):
            # a selected site has likelihood 0: the reference never terminates; the product must refuse
            with pytest.raises(RuntimeError):
                prod(s)
            skipped += 1
            continue
        got = json.loads(prod(s))
        _check(kind, got, want)
        done += 1
    assert done >= 25, (done, skipped)


@pytest.mark.parametrize("k", [9, 12, 16, 17, 20, 25, 32, 33, 48, 64])
def test_medium_and_large_state_spaces(oracle, k):
    """state counts that exercise every padded width of the register-resident vector kernel (K = 16, 20, 32) and of
    the matrix-core kernels (T = 1..4 row tiles), through the JSON API with compact character data"""
    import arbplf
    rng = random.Random(1000 + k)
    n_nodes = 9
    edges = random_tree(rng, n_nodes)
    Q = [[0 if i == j else rng.random() + 0.05 for j in range(k)] for i in range(k)]
    nchar = k + 2
    defs = [[1.0 if j == c else 0.0 for j in range(k)] for c in range(k)] + [[1.0] * k] + \
           [[1.0 if j % 3 == 0 else 0.0 for j in range(k)]]
    S = 12 if k < 48 else 6
    data = [[rng.randrange(nchar) if rng.random() < 0.7 else k for _ in range(n_nodes)] for _ in range(S)]
    md = {"edges": edges, "edge_rate_coefficients": [rng.random() * 0.5 + 0.01 for _ in edges], "rate_matrix": Q,
          "rate_divisor": "equilibrium_exit_rate", "root_prior": "equilibrium_distribution",
          "character_definitions": defs, "character_data": data,
          "rate_mixture": {"rates": [0.3, 1.7], "prior": [0.4, 0.6]}}
    for kind, prod, orc in (("ll", arbplf.arbplf_ll, oracle.arbplf_ll), ("deriv", arbplf.arbplf_deriv, oracle.arbplf_deriv),
                            ("marginal", arbplf.arbplf_marginal, oracle.arbplf_marginal)):
        s = json.dumps({"model_and_data": md})
        want = json.loads(orc(s))
        if any(not np.isfinite(r[-1]) for r in want["data"]):
            pytest.skip("zero-likelihood site drawn")
        _check(kind, json.loads(prod(s)), want)
    # a derivative query for a few edges only (edge mask)
    full = json.dumps({"model_and_data": md})
    few = json.dumps({"model_and_data": md, "edge_reduction": {"selection": [0, 3, 3, n_nodes - 2]}})
    _check("deriv", json.loads(arbplf.arbplf_deriv(few)), json.loads(oracle.arbplf_deriv(few)))
    if k > 20:
        # matrix-core kernels: the same two queries through the arbplf-deriv executable with the node-visit up pass
        # (k_up_nodes_mfma, ARBPLF_UP_NODES=1; the calls above took the default one-edge-at-a-time pass).  Random
        # trees here have multifurcations, unary nodes and data at internal nodes, and there are two rate categories
        import os
        import subprocess
        exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "phyly_amd", "csrc", "arbplf-deriv")
        for q in (full, few):
            r = subprocess.run([exe], input=q, capture_output=True, text=True, timeout=300, env=dict(os.environ, ARBPLF_UP_NODES="1"))
            assert r.returncode == 0, r.stderr
            _check("deriv", json.loads(r.stdout), json.loads(oracle.arbplf_deriv(q)))
