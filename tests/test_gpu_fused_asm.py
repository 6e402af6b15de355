"""GPU: k_ll_fused4_asm (k = 4 assembly interpreter, VGPR stack for <= 4 slots, AGPR stack up to 8) against the oracle
and against the C++ interpreter (PLK_OPT_FUSED_ASM = 0) on everything that changes its code path: 1, 4, 5 and 8 rate
categories, 4-bit and 8-bit staged codes, internal nodes with data (pseudo tip slot), stack pushes, rescaling on a
deep tree, weighted sums, ragged tile sizes."""
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


def _both(eng, codes, defs, w=None, pt=True):
    """PLK_INFO_LL_VARIANT numbering: 6 pair-table interpreter with two sites per lane (the default where it applies;
    PLK_OPT_PAIR_TABLES 1 = widest tile that fits, 5 = 1024-site tiles), 5 pair tables with one site per lane
    (options 2, 3), 1 assembly interpreter over 256-site tiles (option 0), 3 C++ interpreter.
    -> {(variant, option): (per-site ll, weighted sum)}"""
    out = {}
    for variant, asm, pairs in ((6, 1, 1), (6, 1, 5), (5, 1, 2), (5, 1, 3), (1, 1, 0), (3, 0, 0)):
        if variant >= 5 and not pt:
            continue
        eng.set_option(E.OPT_FUSED_ASM, asm)
        eng.set_option(E.OPT_PAIR_TABLES, pairs)
        eng.set_patterns_codes(codes, defs)
        eng.set_site_weights(w)
        ll, s = eng.ll()
        assert eng.info(E.INFO_LL_KERNEL) == 1 and eng.info(E.INFO_LL_VARIANT) == variant, (variant, pairs, eng.info(E.INFO_LL_VARIANT))
        out[(variant, pairs)] = (ll, s[0] + s[1])
    eng.set_option(E.OPT_FUSED_ASM, 1)
    eng.set_option(E.OPT_PAIR_TABLES, 1)
    eng.set_site_weights(None)
    return out


@pytest.mark.parametrize("T,S,cats", [(100, 5000, 4), (37, 257, 5), (12, 1, 1), (64, 1023, 8)])
def test_matches_oracle_and_cpp_interpreter(eng, oracle, T, S, cats):
    wl = synth.Workload(T=T, k=4, tree="yule", model="gtr_g4", seed=100 + T)
    wl.mixture = dict(gamma_shape=0.5, gamma_categories=cats)
    wl.k0 = None
    wl._cum = None
    wl.setup_engine(eng)
    codes = wl.simulate(S)
    w = np.linspace(0.25, 1.75, S)
    out = _both(eng, codes, wl.defs, w)
    want = oracle_site_ll(oracle, wl, codes)
    for key, (ll, tot) in out.items():
        assert np.max(np.abs(ll - want) / np.maximum(1.0, np.abs(want))) <= 1e-12, key
        assert abs(tot - float(np.sum(want.astype(np.longdouble) * w))) <= 1e-12 * abs(tot), key
    assert np.max(np.abs(out[(1, 0)][0] - out[(3, 0)][0])) <= 1e-13 * np.max(np.abs(want))
    for key in out:
        assert np.max(np.abs(out[key][0] - out[(1, 0)][0])) <= 1e-13 * np.max(np.abs(want)), key
    assert np.array_equal(out[(6, 1)][0], out[(5, 2)][0])          # the same tables, the same arithmetic per site
    if T >= 12:
        assert eng.info(E.INFO_PAIR_TABLES) == 0          # the last evaluation ran without them
        eng.set_patterns_codes(codes, wl.defs)
        eng.ll()
        assert eng.info(E.INFO_LL_VARIANT) == 6 and eng.info(E.INFO_PAIR_TABLES) >= 1


def test_wide_codes_and_node_data(eng, oracle):
    """21 character definitions (8-bit staged codes), ambiguity rows at leaves and data on internal nodes (the pseudo
    tip slot)"""
    wl = synth.Workload(T=40, k=4, tree="yule", model="gtr_g4", seed=77)
    wl.setup_engine(eng)
    S = 700
    codes = wl.simulate(S)
    rng = np.random.default_rng(3)
    extra = np.round(rng.random((16, 4)) * 0.9 + 0.05, 3)
    defs = np.vstack([wl.defs, extra])                      # 21 definitions
    amb = rng.random(codes.shape) < 0.15
    codes = np.where(amb, rng.integers(5, 21, size=codes.shape), codes).astype(np.uint8)
    out = _both(eng, codes, defs, pt=False)                 # 21 definitions: beyond the pair tables' 16
    md = wl.json_model(codes[:, :1])
    md["character_definitions"] = defs.tolist()
    m = oracle.parse_model(md)
    ow = oracle.prepare(m)
    want, _ = oracle.site_ll(m, ow, codes=np.ascontiguousarray(codes.T), defs=defs, precise=1)
    for key in ((1, 0), (3, 0)):
        assert np.max(np.abs(out[key][0] - want) / np.maximum(1.0, np.abs(want))) <= 1e-12


@pytest.mark.parametrize("T,variant", [(430, 1), (700, 0)])
def test_deep_tree_rescaling(eng, oracle, T, variant):
    """430 taxa: the largest tip tables the assembly interpreter's LDS budget takes (rescaling ops every few levels);
    700 taxa: beyond it (variant 0: the general kernel takes over), and site likelihoods fall below the smallest double"""
    wl = synth.Workload(T=T, k=4, tree="yule", model="gtr_g4", seed=9)
    wl.setup_engine(eng)
    codes = wl.simulate(300)
    eng.set_patterns_codes(codes, wl.defs)
    ll, _ = eng.ll()
    assert eng.info(E.INFO_LL_VARIANT) == variant
    want = oracle_site_ll(oracle, wl, codes)
    if T == 700:
        assert np.min(want) < -745
    assert np.max(np.abs(ll - want) / np.abs(want)) <= 1e-12
