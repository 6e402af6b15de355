"""GPU parity tests of the ll hot path (plk_* C-ABI) against the CPU oracle.

Tolerance: per-site |ll_gpu - ll_oracle| <= 1e-12 * max(1, |ll_oracle|)
(the fp64 tolerance BASELINE.json's north_star states)."""
import numpy as np
import pytest

from helpers import oracle_site_ll, oracle_model, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-12


@pytest.fixture(scope="module")
def eng():
    from phyly_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _workload(cfg):
    from phyly_amd import synth
    return synth.Workload(cfg)


@pytest.mark.parametrize("cfg", [2, 3, 4, 5])
def test_transition_matrices_match_oracle(eng, oracle, cfg):
    w = _workload(cfg)
    w.setup_engine(eng)
    P = eng.transition_matrices()
    codes = w.simulate(4)
    m, ow = oracle_model(oracle, w, codes)
    assert P.shape == ow["P"].shape
    # entries of P are probabilities: absolute tolerance relative to 1
    assert np.max(np.abs(P - ow["P"])) <= 4e-16
    np.testing.assert_allclose(P.sum(axis=-1), 1.0, rtol=0, atol=1e-14)


@pytest.mark.parametrize("cfg,S", [(2, 3000), (3, 3000), (4, 700), (5, 300)])
@pytest.mark.parametrize("kind", ["simulated", "random"])
def test_site_ll_matches_oracle(eng, oracle, cfg, S, kind):
    from phyly_amd import engine as E
    w = _workload(cfg)
    w.setup_engine(eng)
    codes = w.simulate(S) if kind == "simulated" else w.random_codes(S, seed=cfg)
    want = oracle_site_ll(oracle, w, codes)
    eng.set_option(E.OPT_FORCE_GENERIC, 0)
    eng.set_patterns_codes(codes, w.defs)
    got, (hi, lo) = eng.ll()
    kernel = eng.info(E.INFO_LL_KERNEL)
    assert kernel == (1 if w.k == 4 else 4 if w.k <= 32 else 3)   # assembly interpreter / register-resident vector / fp64 MFMA
    if kernel == 4:
        eng.set_option(E.OPT_MFMA, 2)                # the matrix-core kernel must agree as well
        got3, _ = eng.ll()
        assert eng.info(E.INFO_LL_KERNEL) == 3
        assert rel_err(got3, want) <= TOL
        eng.set_option(E.OPT_MFMA, 1)
    assert rel_err(got, want) <= TOL
    assert abs((hi + lo) - float(np.sum(want.astype(np.longdouble)))) <= TOL * abs(np.sum(want))
    # the generic (HBM-slot) traversal must agree too
    eng.set_option(E.OPT_FORCE_GENERIC, 1)
    got2, _ = eng.ll()
    assert eng.info(E.INFO_LL_KERNEL) == 2
    assert rel_err(got2, want) <= TOL
    eng.set_option(E.OPT_FORCE_GENERIC, 0)


def test_ragged_sizes_and_weights(eng, oracle):
    """S not a multiple of the 256-site tile; weighted sums; S = 1."""
    w = _workload(3)
    w.setup_engine(eng)
    for S in (1, 63, 257, 1000):
        codes = w.simulate(S)
        want = oracle_site_ll(oracle, w, codes)
        eng.set_patterns_codes(codes, w.defs)
        wts = np.linspace(0.5, 2.0, S)
        eng.set_site_weights(wts)
        got, (hi, lo) = eng.ll()
        assert rel_err(got, want) <= TOL
        ref = float(np.sum(want.astype(np.longdouble) * wts.astype(np.longdouble)))
        assert abs((hi + lo) - ref) <= TOL * max(1.0, abs(ref))


def test_dense_patterns_generic_kernel(eng, oracle):
    """probability_array-style dense observations B[N][k][S] with data on internal nodes."""
    w = _workload(2)
    w.setup_engine(eng)
    S = 200
    rng = np.random.default_rng(5)
    B = rng.random((S, w.N, w.k))            # oracle layout [S][N][k]
    B[:, ::3, :] = 1.0                        # some nodes unobserved
    m, ow = oracle_model(oracle, w, w.simulate(1))
    want, _ = oracle.site_ll(m, ow, B=B)
    eng.set_patterns_dense(np.ascontiguousarray(B.transpose(1, 2, 0)))
    got, _ = eng.ll()
    assert rel_err(got, want) <= TOL


def test_deep_scaling_no_underflow(eng, oracle):
    """A 400-taxon caterpillar-ish random tree with random data: site likelihoods
    around 1e-250..1e-400 must not underflow (exact power-of-two rescaling)."""
    from phyly_amd import synth
    w = synth.Workload(T=600, k=4, tree="yule", model="gtr_g4", seed=77)
    w.setup_engine(eng)
    codes = w.random_codes(500, seed=3, missing_frac=0.0)
    want = oracle_site_ll(oracle, w, codes)
    assert want.min() < -700          # below the fp64 underflow threshold in linear space
    eng.set_patterns_codes(codes, w.defs)
    got, _ = eng.ll()
    assert np.all(np.isfinite(got))
    assert rel_err(got, want) <= TOL


@pytest.mark.parametrize("ns", [1, 2])
def test_fused_assembly_variants(eng, oracle, ns):
    """the 256-site-tile interpreters under PLK_OPT_FUSED_SITES_PER_LANE = 1 (the assembly interpreter) and = 2 (the C++
    interpreter with two sites per lane, 512-site tiles), ragged sizes, a caterpillar-like tree that needs the 8-slot stack,
    more than 16 character definitions (unpacked codes)"""
    from phyly_amd import synth, engine as E
    eng.set_option(E.OPT_FUSED_NS, ns)
    try:
        for cfg, sizes in ((3, (1, 255, 513, 1500)), (2, (700,))):
            w = _workload(cfg)
            w.setup_engine(eng)
            for S in sizes:
                codes = w.random_codes(S, seed=S)
                want = oracle_site_ll(oracle, w, codes)
                eng.set_patterns_codes(codes, w.defs)
                eng.set_site_weights(None)
                got, (hi, lo) = eng.ll()
                assert eng.info(E.INFO_LL_KERNEL) == 1
                assert rel_err(got, want) <= TOL, (cfg, S)
                assert abs((hi + lo) - float(np.sum(want.astype(np.longdouble)))) <= TOL * abs(np.sum(want))
        w = synth.Workload(T=300, k=4, tree="yule", model="gtr_g4", seed=5)
        w.setup_engine(eng)
        codes = w.simulate(777)
        eng.set_patterns_codes(codes, w.defs)
        got, _ = eng.ll()
        assert rel_err(got, oracle_site_ll(oracle, w, codes)) <= TOL
        # 20 character definitions: codes are staged one per byte
        w = _workload(2)
        w.setup_engine(eng)
        rng = np.random.default_rng(8)
        defs = np.vstack([np.eye(4), np.ones((1, 4)), rng.integers(0, 2, (15, 4)).astype(float)])
        defs[defs.sum(axis=1) == 0] = 1.0
        codes = rng.integers(0, 20, (w.N, 900)).astype(np.uint8)
        m, ow = oracle_model(oracle, w, codes[:, :1] % 5)
        want, _ = oracle.site_ll(m, ow, codes=np.ascontiguousarray(codes.T), defs=defs)
        eng.set_patterns_codes(codes, defs)
        got, _ = eng.ll()
        assert eng.info(E.INFO_LL_KERNEL) == 1
        assert rel_err(got, want) <= TOL
    finally:
        eng.set_option(E.OPT_FUSED_NS, 0)


@pytest.mark.parametrize("T,k,S", [(40, 20, 700), (23, 12, 300)])
def test_vector_kernel_pair_tables_equal_the_plain_program(oracle, T, k, S):
    """k_ll_vec on the pair-table program (two-leaf subtrees as rows of L2-resident tables built in double-double,
    PLK_OPT_PAIR_TABLES on, the default) against the same kernel on the plain program and against the oracle, with
    missing data and ambiguity at the leaves."""
    from phyly_amd import engine as E, synth
    model = "aa20" if k == 20 else None
    wl = synth.Workload(T=T, k=k, tree="yule", model=model, seed=31) if model else None
    if wl is None:
        wl = synth.Workload(T=T, k=20, tree="yule", model="aa20", seed=32)
        # a 12-state model: the top-left block of the 20-state one
        wl.k = k
        wl.Q = [row[:k] for row in wl.Q[:k]]
        wl.defs = np.vstack([np.eye(k), np.ones((1, k))])
        wl.nchar = k + 1
        wl.k0 = None
        wl._cum = None
    eng = E.Engine(0)
    wl.setup_engine(eng)
    codes = wl.random_codes(S, seed=5, missing_frac=0.1)
    out = {}
    for opt in (1, 0):
        eng.set_option(E.OPT_PAIR_TABLES, opt)
        eng.set_patterns_codes(codes, wl.defs)
        ll, s = eng.ll()
        assert eng.info(E.INFO_LL_KERNEL) == 4
        assert (eng.info(E.INFO_PAIR_TABLES) > 0) == (opt == 1)
        out[opt] = ll
    eng.close()
    want = oracle_site_ll(oracle, wl, codes)
    for opt in (1, 0):
        assert np.max(np.abs(out[opt] - want) / np.maximum(1.0, np.abs(want))) <= 1e-12, opt


def test_matrix_core_kernel_with_two_site_groups_per_wave(oracle):
    """PLK_OPT_MFMA_NS2 = 1 (opt-in): k_ll_mfma_ns2, 128 sites per workgroup, against the default kernel and the oracle"""
    from phyly_amd import engine as E, synth
    wl = synth.Workload(5)
    eng = E.Engine(0)
    wl.setup_engine(eng)
    codes = wl.random_codes(333, seed=8, missing_frac=0.1)
    out = {}
    for opt in (0, 1):
        eng.set_option(E.OPT_MFMA_NS2, opt)
        eng.set_patterns_codes(codes, wl.defs)
        ll, _ = eng.ll()
        assert eng.info(E.INFO_LL_KERNEL) == 3
        out[opt] = ll
    eng.close()
    want = oracle_site_ll(oracle, wl, codes)
    assert np.array_equal(out[0], out[1])
    assert np.max(np.abs(out[1] - want) / np.maximum(1.0, np.abs(want))) <= 1e-12
