"""GPU parity of plk_deriv / plk_marginal (C-ABI) against the binary128 oracle on
seeded synthetic workloads.

Tolerance: |d| <= 1e-12 * max(|expected|, scale) where scale is the largest
magnitude in the same site row."""
import numpy as np
import pytest

from helpers import oracle_model

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from phyly_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _dense_from_codes(w, codes):
    return w.defs[codes.T]          # [S][N][k]


def _row_err(got, want):
    got = got.reshape(got.shape[0], -1)
    want = want.reshape(want.shape[0], -1)
    scale = np.max(np.abs(want), axis=1, keepdims=True)
    return np.max(np.abs(got - want) / np.maximum(np.abs(want), np.maximum(scale, 1e-300)))


@pytest.mark.parametrize("cfg,S", [(2, 300), (3, 200), (4, 40), (5, 24)])
def test_deriv_matches_oracle(eng, oracle, cfg, S):
    from phyly_amd import synth
    w = synth.Workload(cfg)
    w.setup_engine(eng)
    codes = w.simulate(S)
    m, ow = oracle_model(oracle, w, codes)
    want = oracle.site_deriv(m, ow, _dense_from_codes(w, codes), precise=2 if w.k <= 4 else 1)
    eng.set_patterns_codes(codes, w.defs)
    wts = np.linspace(0.25, 1.75, S)
    eng.set_site_weights(wts)
    got, sums = eng.deriv()
    assert _row_err(got, want) <= 1e-12
    ref = (want.astype(np.longdouble) * wts[:, None].astype(np.longdouble)).sum(axis=0)
    tot = sums[:, 0] + sums[:, 1]
    assert np.max(np.abs(tot - ref.astype(float)) / np.max(np.abs(ref))) <= 1e-12


def test_deriv_edge_mask_and_chunking(eng, oracle):
    from phyly_amd import synth, engine as E
    w = synth.Workload(3)
    w.setup_engine(eng)
    S = 500
    codes = w.random_codes(S, seed=9)
    m, ow = oracle_model(oracle, w, codes)
    want = oracle.site_deriv(m, ow, _dense_from_codes(w, codes), precise=2)
    eng.set_patterns_codes(codes, w.defs)
    mask = np.zeros(w.E, dtype=np.int32)
    mask[::3] = 1
    eng.set_option(E.OPT_SITE_CHUNK, 128)          # forces 4 chunks
    got, sums = eng.deriv(edge_mask=mask)
    eng.set_option(E.OPT_SITE_CHUNK, 0)
    sel = mask.astype(bool)
    assert _row_err(got[:, sel], want[:, sel]) <= 1e-12
    assert np.all(got[:, ~sel] == 0.0)
    ref = want.astype(np.longdouble).sum(axis=0).astype(float)
    tot = sums[:, 0] + sums[:, 1]
    assert np.max(np.abs(tot[sel] - ref[sel]) / np.max(np.abs(ref))) <= 1e-12


@pytest.mark.parametrize("cfg,S", [(2, 300), (3, 200), (4, 40), (5, 24)])
def test_marginal_matches_oracle(eng, oracle, cfg, S):
    from phyly_amd import synth
    w = synth.Workload(cfg)
    w.setup_engine(eng)
    codes = w.simulate(S)
    m, ow = oracle_model(oracle, w, codes)
    want = oracle.site_marginal(m, ow, _dense_from_codes(w, codes), precise=1)
    eng.set_patterns_codes(codes, w.defs)
    got, sums = eng.marginal()
    assert np.max(np.abs(got - want)) <= 1e-12          # probabilities: absolute = relative to 1
    np.testing.assert_allclose(got.sum(axis=2), 1.0, rtol=0, atol=1e-12)
    tot = sums[..., 0] + sums[..., 1]
    assert np.max(np.abs(tot - want.sum(axis=0))) <= 1e-12 * S


@pytest.mark.parametrize("cfg,S", [(2, 1000), (3, 777), (4, 333), (5, 100), (5, 7)])
@pytest.mark.parametrize("masked", [False, True])
def test_site_summed_marginal_is_reduced_where_it_is_produced(eng, cfg, S, masked):
    """Site sums without per-site output (src/arbplfmarginal.c:237-256 accumulates as it goes): the up pass leaves per-wave
    weighted sums, the N k planes are not written.  Same sums as the two-stage path (planes + k_wsum_rows, taken when
    per-site values are asked for as well) to 1e-14 of the total site weight, with site weights, a node mask and a
    ragged last wave."""
    from phyly_amd import synth
    w = synth.Workload(cfg)
    w.setup_engine(eng)
    codes = w.random_codes(S, seed=cfg) if cfg != 3 else w.simulate(S)
    eng.set_patterns_codes(codes, w.defs)
    wts = np.linspace(0.5, 1.5, S)
    eng.set_site_weights(wts)
    mask = None
    if masked:
        mask = (np.arange(w.N) % 3 != 1).astype(np.int32)
    per_site, two_stage = eng.marginal(node_mask=mask)
    _, fused = eng.marginal(node_mask=mask, per_site=False)
    eng.set_site_weights(None)
    a, b = two_stage[..., 0] + two_stage[..., 1], fused[..., 0] + fused[..., 1]
    assert np.max(np.abs(a - b)) <= 1e-14 * wts.sum()
    want = (per_site * wts[:, None, None]).sum(axis=0)
    assert np.max(np.abs(b - want)) <= 1e-13 * wts.sum()
    if masked:
        assert np.all(b[mask == 0] == 0.0)


def test_marginal_node_mask_dense(eng, oracle):
    """dense observations with data on internal nodes; only some nodes requested"""
    from phyly_amd import synth
    w = synth.Workload(2)
    w.setup_engine(eng)
    S = 64
    rng = np.random.default_rng(11)
    B = rng.random((S, w.N, w.k)) + 0.05
    B[:, 1::2, :] = 1.0
    m, ow = oracle_model(oracle, w, w.simulate(1))
    want = oracle.site_marginal(m, ow, B, precise=1)
    wantd = oracle.site_deriv(m, ow, B, precise=2)
    eng.set_patterns_dense(np.ascontiguousarray(B.transpose(1, 2, 0)))
    mask = np.zeros(w.N, dtype=np.int32)
    mask[[0, 5, w.N - 1, w.N // 2]] = 1
    got, _ = eng.marginal(node_mask=mask)
    sel = mask.astype(bool)
    assert np.max(np.abs(got[:, sel] - want[:, sel])) <= 1e-12
    assert np.all(got[:, ~sel] == 0.0)
    gotd, _ = eng.deriv()
    assert _row_err(gotd, wantd) <= 1e-12


def test_deep_tree_rescaling_k4(eng, oracle):
    """1500 taxa: site likelihoods ~ e^-2000 are far below the double range; the k = 4 down/up kernels rescale node
    vectors by exact powers of two and combine categories at a common exponent (binary128 oracle as the checker)"""
    from phyly_amd import synth, engine as E
    w = synth.Workload(T=1500, k=4, tree="yule", model="gtr_g4", seed=78)
    w.setup_engine(eng)
    S = 48
    codes = w.simulate(S)
    m, ow = oracle_model(oracle, w, codes)
    B = _dense_from_codes(w, codes)
    eng.set_patterns_codes(codes, w.defs)
    eng.set_site_weights(None)
    ll, _ = eng.ll()
    assert ll.min() < -1000.0 and np.sum(ll < -745.0) >= 10    # exp(ll) underflows for many of the sites
    want = oracle.site_deriv(m, ow, B, precise=2)
    got, _ = eng.deriv()
    assert np.all(np.isfinite(got))
    assert _row_err(got, want) <= 1e-12
    wantm = oracle.site_marginal(m, ow, B, precise=2)
    gotm, _ = eng.marginal()
    assert np.max(np.abs(gotm - wantm)) <= 1e-12
    L = np.diag([1.0, 2.0, 3.0, 4.0])
    F = oracle.frechet(m, ow, L, 1.0, False, None, precise=2)
    wantx = oracle.site_edge_expect(m, ow, B, F, 0, None, precise=2)
    gotx, _ = eng.edge_expect(L, E.COEF_PRIOR)
    assert _row_err(gotx, wantx) <= 1e-12


def test_deep_tree_rescaling_generic_and_mfma(eng, oracle):
    """the same property for the other two kernel families: the vector kernels (dense observations, k = 4,
    1200 taxa) and the matrix-core kernels (k = 20, 400 taxa: site likelihoods ~ 20^-400)"""
    from phyly_amd import synth
    w = synth.Workload(T=1200, k=4, tree="yule", model="hky85", seed=79)
    w.setup_engine(eng)
    S = 24
    codes = w.simulate(S)
    m, ow = oracle_model(oracle, w, codes)
    B = _dense_from_codes(w, codes)
    eng.set_patterns_dense(np.ascontiguousarray(B.transpose(1, 2, 0)))
    eng.set_site_weights(None)
    ll, _ = eng.ll()
    assert np.sum(ll < -745.0) >= 10
    got, _ = eng.deriv()
    assert np.all(np.isfinite(got))
    assert _row_err(got, oracle.site_deriv(m, ow, B, precise=2)) <= 1e-12
    gotm, _ = eng.marginal()
    assert np.max(np.abs(gotm - oracle.site_marginal(m, ow, B, precise=2))) <= 1e-12

    w = synth.Workload(T=400, k=20, tree="yule", model="aa20", seed=80)
    w.setup_engine(eng)
    S = 8
    codes = w.random_codes(S, seed=3, missing_frac=0.0)      # unrelated states at the tips: ll ~ -3 per taxon
    m, ow = oracle_model(oracle, w, codes)
    B = _dense_from_codes(w, codes)
    eng.set_patterns_codes(codes, w.defs)
    ll, _ = eng.ll()
    assert ll.max() < -745.0
    got, _ = eng.deriv()
    assert np.all(np.isfinite(got))
    assert _row_err(got, oracle.site_deriv(m, ow, B, precise=2)) <= 1e-12
    gotm, _ = eng.marginal()
    assert np.max(np.abs(gotm - oracle.site_marginal(m, ow, B, precise=2))) <= 1e-12


def test_node_visit_up_pass_equals_edge_up_pass(eng, oracle):
    """matrix-core kernels (BASELINE config 5, codon), derivative queries: k_up_nodes_mfma over stored edge vectors
    (PLK_OPT_UP_NODES = 1) against the one-edge-at-a-time k_up_mfma (the default) and the oracle; all edges and a sparse
    edge mask, several site chunks.  Trees with multifurcations, unary nodes and data at internal nodes:
    test_gpu_differential.py::test_medium_and_large_state_spaces."""
    from phyly_amd import synth, engine as E
    w = synth.Workload(5)
    w.setup_engine(eng)
    S = 300
    codes = w.simulate(S)
    eng.set_patterns_codes(codes, w.defs)
    eng.set_site_weights(None)
    mask = np.zeros(w.E, dtype=np.int32)
    mask[[1, 7, w.E // 2, w.E - 1]] = 1
    out = {}
    for nodes in (1, 0):
        eng.set_option(E.OPT_UP_NODES, nodes)
        eng.set_option(E.OPT_SITE_CHUNK, 128)
        out[nodes] = (eng.deriv()[0], eng.deriv(edge_mask=mask)[0])
    eng.set_option(E.OPT_UP_NODES, 2)          # the default: k = 4 node visits on, matrix-core node visits off
    eng.set_option(E.OPT_SITE_CHUNK, 0)
    for q in (0, 1):
        scale = np.max(np.abs(out[0][q]), axis=1, keepdims=True)
        assert np.max(np.abs(out[1][q] - out[0][q]) / np.maximum(scale, 1e-300)) <= 1e-13
    sel = mask.astype(bool)
    assert np.all(out[1][1][:, ~sel] == 0.0)
    # with a mask other children continue in registers and the messages are multiplied in another order: rounding only
    assert np.max(np.abs(out[1][1][:, sel] - out[1][0][:, sel]) / np.max(np.abs(out[1][0][:, sel]), axis=1, keepdims=True)) <= 1e-14
    assert np.max(np.abs(out[0][1][:, sel] - out[0][0][:, sel])) == 0.0
    nq = 12
    m, ow = oracle_model(oracle, w, codes[:, :nq])
    want = oracle.site_deriv(m, ow, _dense_from_codes(w, codes[:, :nq]), precise=1)
    assert _row_err(out[1][0][:nq], want) <= 1e-12


@pytest.mark.parametrize("cfg", [2, 3])
def test_k4_node_visit_up_pass_equals_edge_up_pass(eng, oracle, cfg):
    """k = 4 derivative queries: k_up4_nodes (default, PLK_OPT_UP_NODES bit 1) against k_up4 (bit cleared) and the
    oracle; all edges and a sparse edge mask, several site chunks"""
    from phyly_amd import synth, engine as E
    w = synth.Workload(cfg)
    w.setup_engine(eng)
    S = 700
    codes = w.random_codes(S, seed=cfg)
    eng.set_patterns_codes(codes, w.defs)
    eng.set_site_weights(None)
    mask = np.zeros(w.E, dtype=np.int32)
    mask[[0, 5, w.E // 2, w.E - 1]] = 1
    out = {}
    for nodes in (2, 0):
        eng.set_option(E.OPT_UP_NODES, nodes)
        eng.set_option(E.OPT_SITE_CHUNK, 256)
        out[nodes] = (eng.deriv()[0], eng.deriv(edge_mask=mask)[0])
    eng.set_option(E.OPT_UP_NODES, 2)
    eng.set_option(E.OPT_SITE_CHUNK, 0)
    for q in (0, 1):
        scale = np.max(np.abs(out[0][q]), axis=1, keepdims=True)
        assert np.max(np.abs(out[2][q] - out[0][q]) / np.maximum(scale, 1e-300)) <= 1e-13
    sel = mask.astype(bool)
    assert np.all(out[2][1][:, ~sel] == 0.0)
    m, ow = oracle_model(oracle, w, codes[:, :60])
    want = oracle.site_deriv(m, ow, _dense_from_codes(w, codes[:, :60]), precise=2)
    assert _row_err(out[2][0][:60], want) <= 1e-12


@pytest.mark.parametrize("cfg,S", [(3, 300), (4, 64), (5, 32)])
def test_derivative_after_an_ll_evaluation_at_new_rates(eng, cfg, S):
    """an ll evaluation runs K1 without dP = r Q P (nothing in it reads dP); the first derivative query afterwards makes it
    from the stored double-double P (k_dP_dd, ensure_dP in plk_engine.hip).  Both orders of the two queries must give the
    same gradient bit for bit, for the LDS form of K1 (k = 4, 20) and the tiled one (k = 61)."""
    from phyly_amd import synth
    w = synth.Workload(cfg)
    w.setup_engine(eng)
    eng.set_patterns_codes(w.simulate(S), w.defs)
    r = w.edge_rates_csr * 1.25
    eng.update_edge_rates(r)
    d_first, _ = eng.deriv()                      # K1 with dP
    eng.update_edge_rates(r * 1.0)                # same values, marks the model dirty again
    ll, _ = eng.ll()                              # K1 without dP
    d_after, _ = eng.deriv()                      # dP on demand
    assert np.all(np.isfinite(ll))
    assert np.array_equal(d_first, d_after)
    eng.update_edge_rates(w.edge_rates_csr)
