"""GPU: k_ll_fused4_cn (two or four rate categories per pass of the traversal program) against the oracle and against
the one-category-per-pass kernel, on everything that changes its code path: 4 and 8 categories, 4-bit and 8-bit staged
codes, internal nodes with data (pseudo tip slot), stack pushes, rescaling on a deep tree, weighted sums, ragged
tile sizes; and that the engine picks it exactly when it applies."""
import numpy as np
import pytest

from helpers import oracle_site_ll
from phyly_amd import synth
from phyly_amd import engine as E

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = E.Engine(0)
    yield e
    e.close()


def _both(eng, codes, defs, w=None):
    """variant 2: four categories per pass, 4: two per pass, 1: one per pass (PLK_INFO_LL_VARIANT numbering)"""
    out = {}
    for variant, opt in ((2, 4), (4, 2), (1, 0)):
        eng.set_option(E.OPT_FUSED_C4, opt)
        eng.set_patterns_codes(codes, defs)
        eng.set_site_weights(w)
        ll, s = eng.ll()
        assert eng.info(E.INFO_LL_KERNEL) == 1 and eng.info(E.INFO_LL_VARIANT) == variant
        out[variant] = (ll, s[0] + s[1])
    eng.set_option(E.OPT_FUSED_C4, 2)
    eng.set_site_weights(None)
    return out


@pytest.mark.parametrize("T,S", [(100, 5000), (37, 257), (12, 1), (64, 1023)])
def test_four_categories_match_oracle_and_single_category_kernel(eng, oracle, T, S):
    wl = synth.Workload(T=T, k=4, tree="yule", model="gtr_g4", seed=100 + T)
    wl.setup_engine(eng)
    codes = wl.simulate(S)
    w = np.linspace(0.25, 1.75, S)
    out = _both(eng, codes, wl.defs, w)
    want = oracle_site_ll(oracle, wl, codes)
    for variant in (1, 2, 4):
        ll, tot = out[variant]
        assert np.max(np.abs(ll - want) / np.maximum(1.0, np.abs(want))) <= 1e-12
        assert abs(tot - float(np.sum(want.astype(np.longdouble) * w))) <= 1e-12 * abs(tot)
    assert np.max(np.abs(out[1][0] - out[2][0])) <= 1e-13 * np.max(np.abs(want))
    assert np.max(np.abs(out[1][0] - out[4][0])) <= 1e-13 * np.max(np.abs(want))


def test_eight_categories_wide_codes_and_node_data(eng, oracle):
    """8 categories (two groups of four), 21 character definitions (8-bit staged codes), ambiguity rows at leaves and
    data on internal nodes (the pseudo tip slot)"""
    wl = synth.Workload(T=40, k=4, tree="yule", model="gtr_g4", seed=77)
    wl.mixture = dict(gamma_shape=0.7, gamma_categories=8)
    wl.k0 = None
    wl._cum = None
    wl.setup_engine(eng)
    S = 700
    codes = wl.simulate(S)
    rng = np.random.default_rng(3)
    extra = np.round(rng.random((16, 4)) * 0.9 + 0.05, 3)
    defs = np.vstack([wl.defs, extra])                      # 21 definitions
    amb = rng.random(codes.shape) < 0.15
    codes = np.where(amb, rng.integers(5, 21, size=codes.shape), codes).astype(np.uint8)
    out = _both(eng, codes, defs)
    md = wl.json_model(codes[:, :1])
    md["character_definitions"] = defs.tolist()
    m = oracle.parse_model(md)
    ow = oracle.prepare(m)
    want, _ = oracle.site_ll(m, ow, codes=np.ascontiguousarray(codes.T), defs=defs, precise=1)
    for variant in (1, 2, 4):
        assert np.max(np.abs(out[variant][0] - want) / np.maximum(1.0, np.abs(want))) <= 1e-12


def test_deep_tree_rescaling(eng, oracle):
    wl = synth.Workload(T=700, k=4, tree="yule", model="gtr_g4", seed=9)
    wl.setup_engine(eng)
    codes = wl.simulate(300)
    eng.set_patterns_codes(codes, wl.defs)
    ll, _ = eng.ll()
    assert eng.info(E.INFO_LL_VARIANT) in (1, 4)      # two per pass only while two tip tables of 701 slots fit in LDS
    want = oracle_site_ll(oracle, wl, codes)
    assert np.min(want) < -745
    assert np.max(np.abs(ll - want) / np.abs(want)) <= 1e-12


def test_variant_selection(eng):
    for model, variant in (("gtr_g4", 4), ("hky85", 1)):          # C = 4 -> two per pass (default); C = 1 -> one per pass
        wl = synth.Workload(T=20, k=4, tree="yule", model=model, seed=1)
        wl.setup_engine(eng)
        eng.set_patterns_codes(wl.simulate(100), wl.defs)
        eng.ll()
        assert eng.info(E.INFO_LL_VARIANT) == variant
