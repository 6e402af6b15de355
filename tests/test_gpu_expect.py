"""GPU parity of the expectation queries: arbplf-dwell, arbplf-trans,
arbplf-em-update and the engine entry point behind them (plk_edge_expect), against

  * every golden in/out pair the reference holds for these commands (tests/golden/examples),
  * the binary128 oracle on seeded synthetic workloads and random small models,
  * size-independent identities at larger sizes: dwell proportions sum to 1 over states,
    and d ll / d edge_rate = E[rate-weighted transitions] - E[rate-weighted exits].

Tolerance: |d| <= 1e-12 * max(|expected|, scale), scale = the largest magnitude in the same
table / site row (as for deriv and marginal)."""
import copy
import glob
import json
import os
import random

import numpy as np
import pytest

from helpers import GOLDEN, load_json, oracle_model

pytestmark = pytest.mark.gpu
EX = os.path.join(GOLDEN, "examples")


def _pairs(dirs):
    out = []
    for d in dirs:
        for inp in sorted(glob.glob(os.path.join(EX, d, "in*.json"))):
            out.append((inp, os.path.join(os.path.dirname(inp), "out" + os.path.basename(inp)[2:])))
    return out


CASES = [("dwell", p) for p in _pairs(["Felsenstein.2004.fig.16.4/dwell/adenine", "Felsenstein.2004.fig.16.4/dwell/pyrimidines",
                                       "BEAST.MarkovJumps/MarkovRewardsC"])]
CASES += [("trans", p) for p in _pairs(["Felsenstein.2004.fig.16.4/trans/A.to.C.only", "Felsenstein.2004.fig.16.4/trans/all.types",
                                        "Felsenstein.2004.fig.16.4/trans/transversions.only", "BEAST.MarkovJumps/MarkovJumpsC",
                                        "BEAST.MarkovJumps/MarkovMarginalRate"])]
CASES += [("em_update", p) for p in _pairs(["Felsenstein.2004.fig.16.4/em-update/with.full.data",
                                            "Felsenstein.2004.fig.16.4/em-update/with.leaf.data",
                                            "Felsenstein.2004.fig.16.4/em-update/with.no.data"])]


def _prod(kind):
    import arbplf
    return {"ll": arbplf.arbplf_ll, "deriv": arbplf.arbplf_deriv, "dwell": arbplf.arbplf_dwell,
            "trans": arbplf.arbplf_trans, "em_update": arbplf.arbplf_em_update}[kind]


def _orc(oracle, kind):
    return {"dwell": oracle.arbplf_dwell, "trans": oracle.arbplf_trans, "em_update": oracle.arbplf_em_update}[kind]


def _check_table(got, want, rel=1e-12):
    assert got["columns"] == want["columns"]
    assert len(got["data"]) == len(want["data"])
    scale = max([abs(r[-1]) for r in want["data"]] + [0.0])
    for a, b in zip(got["data"], want["data"]):
        assert a[:-1] == b[:-1]
        assert abs(a[-1] - b[-1]) <= rel * abs(b[-1]) + 1e-14 * scale + 1e-300, (a, b)


@pytest.mark.parametrize("kind,pair", CASES, ids=[os.path.relpath(c[1][0], EX) for c in CASES])
def test_reference_goldens(kind, pair):
    inp, outp = pair
    with open(inp) as f:
        got = json.loads(_prod(kind)(f.read()))
    _check_table(got, load_json(outp))


def test_cli_dwell_trans_em(tmp_path):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for exe, d in (("arbplf-dwell", "Felsenstein.2004.fig.16.4/dwell/adenine"),
                   ("arbplf-trans", "Felsenstein.2004.fig.16.4/trans/all.types"),
                   ("arbplf-em-update", "Felsenstein.2004.fig.16.4/em-update/with.leaf.data")):
        with open(os.path.join(EX, d, "in.json")) as f:
            r = subprocess.run([os.path.join(root, "phyly_amd", "csrc", exe)], stdin=f, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        _check_table(json.loads(r.stdout), load_json(os.path.join(EX, d, "out.json")))
    r = subprocess.run([os.path.join(root, "phyly_amd", "csrc", "arbplf-em-update")], input='{"model_and_data": {}}',
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout == ""


# ------------------------------------------------------------------ engine level
@pytest.fixture(scope="module")
def eng():
    from phyly_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _row_err(got, want):
    scale = np.max(np.abs(want), axis=1, keepdims=True)
    return np.max(np.abs(got - want) / np.maximum(np.abs(want), np.maximum(scale, 1e-300)))


def _directions(w, rng):
    k = w.k
    out = []
    L = np.zeros((k, k)); L[rng.integers(k), rng.integers(k)] = 1.0
    out.append(("unit", L, 0))
    out.append(("diag", np.diag(rng.uniform(-1, 2, k)), 0))
    out.append(("dense", rng.uniform(-1, 1, (k, k)), 1))
    out.append(("offdiag", rng.uniform(0, 1, (k, k)) * (1 - np.eye(k)), 2))
    return out


@pytest.mark.parametrize("cfg,S,nreq", [(2, 200, 98), (3, 120, 198), (4, 24, 12), (5, 8, 2)])
def test_edge_expect_matches_oracle(eng, oracle, cfg, S, nreq):
    """plk_edge_expect (fused k=4 path, MFMA path) against the oracle's Frechet matrices + site evaluator"""
    from phyly_amd import synth
    w = synth.Workload(cfg)
    w.setup_engine(eng)
    codes = w.simulate(S)
    m, ow = oracle_model(oracle, w, codes)
    B = w.defs[codes.T]
    eng.set_patterns_codes(codes, w.defs)
    eng.set_site_weights(None)
    rng = np.random.default_rng(1000 + cfg)
    mask = np.zeros(w.E, dtype=np.int32)
    mask[rng.choice(w.E, size=nreq, replace=False)] = 1
    sel = mask.astype(bool)
    precise = 2 if w.k <= 4 else 1
    for name, L, coef in _directions(w, rng):
        F = oracle.frechet(m, ow, L, 1.0, False, mask, precise=precise)
        want = oracle.site_edge_expect(m, ow, B, F, coef, mask, precise=precise)
        got, sums = eng.edge_expect(L, coef, edge_mask=mask)
        assert _row_err(got[:, sel], want[:, sel]) <= 1e-12, (name, coef)
        assert np.all(got[:, ~sel] == 0.0)
        ref = want.astype(np.longdouble).sum(axis=0).astype(float)
        tot = sums[:, 0] + sums[:, 1]
        assert np.max(np.abs(tot[sel] - ref[sel])) <= 1e-12 * np.max(np.abs(ref[sel])), name


@pytest.mark.parametrize("cfg,S,nL", [(3, 300, 2), (3, 257, 5), (2, 64, 4), (4, 16, 3)])
def test_edge_expect_multi_equals_single_calls(eng, cfg, S, nL):
    """several directions per pass (k = 4: up to four share one down / up pass; more are split; other k: one pass
    each) must give exactly what the single-direction entry point gives"""
    from phyly_amd import synth, engine as E
    w = synth.Workload(cfg)
    w.setup_engine(eng)
    codes = w.random_codes(S, seed=cfg)
    eng.set_patterns_codes(codes, w.defs)
    wts = np.linspace(0.5, 1.5, S)
    eng.set_site_weights(wts)
    rng = np.random.default_rng(5 + nL)
    Ls = rng.uniform(-1, 1, (nL, w.k, w.k))
    mask = (rng.random(w.E) < 0.7).astype(np.int32)
    got, gsum = eng.edge_expect_multi(Ls, E.COEF_PRIOR_RATE_EDGE, edge_mask=mask)
    for m in range(nL):
        one, osum = eng.edge_expect(Ls[m], E.COEF_PRIOR_RATE_EDGE, edge_mask=mask)
        assert np.array_equal(got[:, m, :], one)
        assert np.array_equal(gsum[m], osum)
    eng.set_site_weights(None)


@pytest.mark.parametrize("cfg,nreq", [(3, 198), (4, 6), (5, 1)])
def test_frechet_matrices_match_oracle(eng, oracle, cfg, nreq):
    """the device double-double block exponential against the binary128 one, entry by entry"""
    from phyly_amd import synth, engine as E
    w = synth.Workload(cfg)
    w.setup_engine(eng)
    codes = w.simulate(4)
    m, ow = oracle_model(oracle, w, codes)
    rng = np.random.default_rng(77 + cfg)
    mask = np.zeros(w.E, dtype=np.int32)
    mask[rng.choice(w.E, size=nreq, replace=False)] = 1
    k = w.k
    L = rng.uniform(-1, 1, (k, k))
    for coef in (E.COEF_PRIOR, E.COEF_PRIOR_RATE_EDGE, E.COEF_PRIOR_RATE):
        got = eng.frechet_matrices(L, coef)
        want = oracle.frechet(m, ow, L, 1.0, False, mask, precise=1).reshape(ow["C"], w.E, k, k).copy()
        for c in range(ow["C"]):
            for e in range(w.E):
                if coef == E.COEF_PRIOR_RATE_EDGE:
                    want[c, e] *= ow["cat_rates"][c] * m.edge_rates_csr[e]
                elif coef == E.COEF_PRIOR_RATE:
                    want[c, e] *= ow["cat_rates"][c]
        sel = mask.astype(bool)
        scale = np.max(np.abs(want[:, sel]), axis=(2, 3), keepdims=True)
        assert np.max(np.abs(got[:, sel] - want[:, sel]) / np.maximum(scale, 1e-300)) <= 1e-14


@pytest.mark.parametrize("cfg,S", [(3, 200_000), (2, 300_000), (5, 4096)])
def test_identities_at_scale(eng, cfg, S):
    """size-independent properties on a large batch:
       sum_s dwell(s) = 1 on every edge (sum_s F(e_s e_s^T) = F(I) = P), and
       d ll / d rate_e = E[transitions on e] - E[exit-rate dwell on e], both rate-weighted
       (Q = offdiag(Q) - diag(exit), F is linear in the direction, F(Q) = Q exp(Q))."""
    from phyly_amd import synth, engine as E
    w = synth.Workload(cfg)
    w.setup_engine(eng)
    codes = w.simulate(S)
    eng.set_patterns_codes(codes, w.defs)
    eng.set_site_weights(None)
    k = w.k
    ones, _ = eng.edge_expect(np.eye(k), E.COEF_PRIOR, want_sums=False)
    assert np.max(np.abs(ones - 1.0)) <= 1e-12
    Qn = w.prepare()["Qn"]
    tr, _ = eng.edge_expect(Qn * (1 - np.eye(k)), E.COEF_PRIOR_RATE, want_sums=False)
    dw, _ = eng.edge_expect(-np.diag(np.diag(Qn)), E.COEF_PRIOR_RATE, want_sums=False)
    d, _ = eng.deriv(want_sums=False)
    scale = np.maximum(np.max(np.abs(tr), axis=1, keepdims=True), np.max(np.abs(dw), axis=1, keepdims=True))
    assert np.max(np.abs((tr - dw) - d) / scale) <= 1e-11


# ------------------------------------------------------------------ JSON level, random models
def _random_query(rng, kind):
    from test_gpu_differential import random_model
    x = random_model(rng, "deriv")          # model + optional site / edge reductions
    md = x["model_and_data"]
    k = len(md["rate_matrix"])

    def red(n):
        r = rng.random()
        if r < 0.3:
            return None
        out = {}
        if rng.random() < 0.6:
            out["selection"] = [rng.randrange(n) for _ in range(rng.randrange(1, n + 2))]
        m = len(out.get("selection", range(n)))
        r = rng.random()
        if r < 0.3:
            out["aggregation"] = "sum"
        elif r < 0.45:
            out["aggregation"] = "avg"
        elif r < 0.75:
            out["aggregation"] = [round(rng.uniform(-1, 2), 3) for _ in range(m)]
        return out

    if kind == "dwell":
        r = red(k)
        if r is not None:
            x["state_reduction"] = r
    elif kind == "trans":
        r = rng.random()
        if r < 0.25:
            pass
        elif r < 0.4:
            x["trans_reduction"] = {"aggregation": rng.choice(["sum", "avg"])}
        else:
            pairs = [[rng.randrange(k), rng.randrange(k)] for _ in range(rng.randrange(1, 6))]
            tr = {"selection": pairs}
            r = rng.random()
            if r < 0.3:
                tr["aggregation"] = "sum"
            elif r < 0.6:
                tr["aggregation"] = [round(rng.uniform(-1, 2), 3) for _ in pairs]
            x["trans_reduction"] = tr
    else:
        x.pop("edge_reduction", None)
        sr = x.get("site_reduction") or {}
        if "aggregation" not in sr:
            sr["aggregation"] = "sum"
        x["site_reduction"] = sr
    return x


@pytest.mark.parametrize("kind", ["dwell", "trans", "em_update"])
def test_random_inputs_match_oracle(oracle, kind):
    rng = random.Random({"dwell": 44, "trans": 55, "em_update": 66}[kind])
    done = 0
    ncases = {"dwell": 20, "trans": 16, "em_update": 20}[kind]      # trans: one binary128 Frechet build per state pair in the oracle
    for case in range(ncases):
        x = _random_query(rng, kind)
        s = json.dumps(x)
        want = json.loads(_orc(oracle, kind)(s))
        if any(not np.isfinite(r[-1]) for r in want["data"]):
            with pytest.raises(RuntimeError):
                _prod(kind)(s)
            continue
        got = json.loads(_prod(kind)(s))
        if kind == "em_update":
            # a ratio of two sums: where the denominator cancels to ~0 the ratio is ill conditioned; compare
            # rows whose oracle value is on the scale of the input rates
            rates = x["model_and_data"]["edge_rate_coefficients"]
            assert got["columns"] == want["columns"] and len(got["data"]) == len(want["data"])
            for a, b, r in zip(got["data"], want["data"], rates):
                assert a[0] == b[0]
                assert abs(a[1] - b[1]) <= 1e-10 * max(abs(b[1]), r) + 1e-300, (a, b)
        else:
            _check_table(got, want)
        done += 1
    assert done >= (2 * ncases) // 5


def test_em_update_increases_likelihood():
    """EM monotonicity on small random star trees (the property the reference checks in
    test_scripts/test_em_monotonicity.py), through the product's own ll and em-update"""
    rng = np.random.default_rng(1234)
    for _ in range(8):
        n = int(rng.integers(2, 5))
        Q = float(rng.exponential()) * np.exp(rng.standard_normal((n, n)))
        np.fill_diagonal(Q, 0)
        S = int(rng.integers(1, 5))
        arr = np.exp(rng.standard_normal((S, 4, n)))
        md = {"edges": [[0, 1], [0, 2], [0, 3]], "edge_rate_coefficients": np.exp(rng.standard_normal(3)).tolist(),
              "rate_matrix": Q.tolist(), "probability_array": arr.tolist()}
        if rng.integers(2):
            c = int(rng.integers(1, 5))
            p = np.exp(rng.standard_normal(c))
            md["rate_mixture"] = {"rates": np.exp(rng.standard_normal(c)).tolist(), "prior": (p / p.sum()).tolist()}
        q = {"model_and_data": md, "site_reduction": {"aggregation": "sum"}}
        ll0 = json.loads(_prod("ll")(json.dumps(q)))["data"][0][0]
        new = [r[1] for r in json.loads(_prod("em_update")(json.dumps(q)))["data"]]
        q2 = copy.deepcopy(q)
        q2["model_and_data"]["edge_rate_coefficients"] = new
        ll1 = json.loads(_prod("ll")(json.dumps(q2)))["data"][0][0]
        assert ll1 >= ll0 - 1e-12 * abs(ll0)
