"""GPU: the BASELINE configurations at their FULL sizes (10M sites for cfg3, 1M for cfg2 / cfg4, 500k for cfg5),
checked through size-independent properties, since the oracle cannot be run at these sizes:

  * linearity: an alignment made of R copies of a block of patterns has R times the block's log likelihood
    (the block itself is checked against the oracle site by site elsewhere), in double-double sums;
  * sum_s dwell(all states) = S on every edge;
  * sum_s d ll_s / d rate_e = sum_s E[transitions] - sum_s E[exit-rate dwell], edge by edge;
  * the site-chunked down / up passes (the stored vectors of 10M sites do not fit in HBM at once) agree with
    an unchunked run on a prefix."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from phyly_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("cfg", [3, 2, 4, 5])
def test_full_size_properties(eng, cfg):
    from phyly_amd import synth, engine as E
    w = synth.Workload(cfg)
    w.setup_engine(eng)
    S = w.default_S
    B = 8192
    block = w.simulate(B)
    reps, rem = divmod(S, B)
    codes = np.ascontiguousarray(np.concatenate([np.tile(block, (1, reps)), block[:, :rem]], axis=1))
    assert codes.shape[1] == S
    eng.set_patterns_codes(block, w.defs)
    eng.set_site_weights(None)
    ll_block, _ = eng.ll()
    ref = float(np.sum(ll_block.astype(np.longdouble)) * reps + np.sum(ll_block[:rem].astype(np.longdouble)))
    eng.set_patterns_codes(codes, w.defs)
    del codes
    _, (hi, lo) = eng.ll(per_site=False)
    assert abs((hi + lo) - ref) <= 1e-12 * abs(ref)

    k = w.k
    _, ones = eng.edge_expect(np.eye(k), E.COEF_PRIOR, per_site=False)
    tot = ones[:, 0] + ones[:, 1]
    assert np.max(np.abs(tot - S)) <= 1e-11 * S
    Qn = w.prepare()["Qn"]
    _, both = eng.edge_expect_multi(np.stack([-np.diag(np.diag(Qn)), Qn * (1 - np.eye(k))]), E.COEF_PRIOR_RATE, per_site=False)
    dw, tr = both[0, :, 0] + both[0, :, 1], both[1, :, 0] + both[1, :, 1]
    _, d = eng.deriv(per_site=False)
    dv = d[:, 0] + d[:, 1]
    assert np.max(np.abs((tr - dw) - dv) / np.maximum(np.abs(tr), np.abs(dw))) <= 1e-10

    # chunking: the first 3 * 8192 sites with forced chunks of 1024 sites against one chunk
    eng.set_patterns_codes(np.ascontiguousarray(np.tile(block, (1, 3))), w.defs)
    one, _ = eng.deriv(want_sums=False)
    eng.set_option(E.OPT_SITE_CHUNK, 1024)
    many, _ = eng.deriv(want_sums=False)
    eng.set_option(E.OPT_SITE_CHUNK, 0)
    assert np.array_equal(one, many)


def test_cli_with_binary_alignment_file(eng, tmp_path):
    """the whole operator path at scale: arbplf-ll reads 1M sites x 99 nodes from a character_data_file, compresses
    identical patterns, uploads, evaluates, aggregates -- and must agree with the engine driven directly"""
    import json
    import os
    import subprocess
    from phyly_amd import synth
    w = synth.Workload(2)
    w.setup_engine(eng)
    S = 1_000_000
    block = w.simulate(200_000)
    codes = np.ascontiguousarray(np.tile(block, (1, 5)))            # every pattern occurs (at least) five times
    eng.set_patterns_codes(codes, w.defs)
    eng.set_site_weights(None)
    _, (hi, lo) = eng.ll(per_site=False)
    f = tmp_path / "aln.u8"
    np.ascontiguousarray(codes.T).tofile(f)                          # [site][node]
    md = w.json_model(codes[:, :1])
    del md["character_data"]
    md["character_data_file"] = str(f)
    q = {"model_and_data": md, "site_reduction": {"aggregation": "sum"}}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([os.path.join(root, "phyly_amd", "csrc", "arbplf-ll")], input=json.dumps(q),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-500:]
    out = json.loads(r.stdout)
    assert out["columns"] == ["value"]
    assert abs(out["data"][0][0] - (hi + lo)) <= 1e-12 * abs(hi + lo)
