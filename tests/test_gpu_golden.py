"""GPU: the product path (JSON string API -> host C -> HIP engine) against the
reference's own golden input/output files (tests/golden/examples, copied as data
from the reference's examples/ directory).

Tolerances: log likelihoods |d| <= 1e-12 * max(1, |expected|); derivatives and
marginals |d| <= 1e-12 * max(|expected|, row scale) + 1e-300."""
import glob
import json
import os
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN, load_json

pytestmark = pytest.mark.gpu
EX = os.path.join(GOLDEN, "examples")
ROOT = os.path.dirname(GOLDEN.rstrip("/")).rsplit("/tests", 1)[0]


def _run(kind, path):
    import arbplf
    fn = {"ll": arbplf.arbplf_ll, "deriv": arbplf.arbplf_deriv, "marginal": arbplf.arbplf_marginal}[kind]
    with open(path) as f:
        return json.loads(fn(f.read()))


def _compare(got, exp, rel=1e-12, floor=1.0, abs_floor=1e-300):
    assert got["columns"] == exp["columns"]
    assert len(got["data"]) == len(exp["data"])
    scale = max([abs(r[-1]) for r in exp["data"]] + [0.0])
    for a, b in zip(got["data"], exp["data"]):
        assert a[:-1] == b[:-1]
        tol = rel * max(abs(b[-1]), floor if floor else 0.0) + abs_floor
        assert abs(a[-1] - b[-1]) <= tol, (a, b, scale)


LL_DIRS = sorted(d for d in glob.glob(os.path.join(EX, "BEAST.*")) if os.path.exists(os.path.join(d, "in.json")))
LL_DIRS += [os.path.join(EX, p) for p in ("BEAST.AncestralState/ll", "Felsenstein.2004.fig.16.4/ll",
                                          "JC.long.branch/ll", "bpp.phyl/ll")]


@pytest.mark.parametrize("d", LL_DIRS, ids=[os.path.relpath(d, EX) for d in LL_DIRS])
def test_ll_golden(d):
    _compare(_run("ll", os.path.join(d, "in.json")), load_json(os.path.join(d, "out.json")))


def test_ll_felsenstein_three_sites():
    got = _run("ll", os.path.join(EX, "Felsenstein.2004.fig.16.4/ll/in2.json"))
    want = [0.0, -11.297288182875496, -12.390132492111672]   # ll/README.md of the reference
    assert got["columns"] == ["site", "value"]
    for row, w in zip(got["data"], want):
        assert abs(row[1] - w) <= 1e-12 * max(1.0, abs(w))


@pytest.mark.parametrize("name,want", [("GeLL.test.likelihood", -2616.073919844292),
                                       ("GeLL.driver.DNA", -2616.0735881244163)])
def test_ll_gell_readme_values(name, want):
    got = _run("ll", os.path.join(EX, name, "in.json"))
    assert got["columns"] == ["value"]
    assert abs(got["data"][0][0] - want) <= 1e-12 * abs(want)


@pytest.mark.parametrize("d", ["Felsenstein.2004.fig.16.4/deriv", "bpp.phyl/deriv", "JC.long.branch/deriv"])
def test_deriv_golden(d):
    # relative tolerance on every entry, including JC.long.branch's 3.5e-12 derivative
    # (dP = r Q exp(Qrt) is formed in double-double on the device, so no cancellation)
    # absolute floor 1e-25: entries whose exact value is 0 by a stationarity identity (pi Q = 0)
    _compare(_run("deriv", os.path.join(EX, d, "in.json")), load_json(os.path.join(EX, d, "out.json")),
             floor=0.0, abs_floor=1e-25)


@pytest.mark.parametrize("f,want", [("jc29.same", -6.4467380574161446e-17), ("jc29.diff", 2.1489126858053815e-17),
                                    ("jc30.same", -1.6993417021166355e-17), ("jc30.diff", 5.6644723403887852e-18),
                                    ("jc600.same", 0.0)])
def test_deriv_jc_long_branch_readme(f, want):
    path = os.path.join(EX, "JC.long.branch", f + ".json")
    got = _run("deriv", path)
    # absolute floor 1e-30: jc600's true derivative (-4/(e^800+3)) is below double-double resolution
    assert abs(got["data"][0][2] - want) <= 1e-12 * abs(want) + 1e-30
    ll = _run("ll", path)
    assert abs(ll["data"][0][1] - (-2.7725887222397811)) <= 1e-12 * 2.78


@pytest.mark.parametrize("d", ["Felsenstein.2004.fig.16.4/marginal", "BEAST.AncestralState/marginal",
                               "JC.long.branch/marginal"])
def test_marginal_golden(d):
    _compare(_run("marginal", os.path.join(EX, d, "in.json")), load_json(os.path.join(EX, d, "out.json")), floor=1.0)


def test_cli_matches_golden():
    """the arbplf-ll / -deriv / -marginal executables: stdin -> stdout, exit status 0"""
    csrc = os.path.join(os.path.dirname(GOLDEN), "..", "phyly_amd", "csrc")
    for kind in ("ll", "deriv", "marginal"):
        d = os.path.join(EX, "Felsenstein.2004.fig.16.4", kind)
        exe = os.path.abspath(os.path.join(csrc, "arbplf-" + kind))
        with open(os.path.join(d, "in.json"), "rb") as f:
            p = subprocess.run([exe], stdin=f, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert p.returncode == 0, p.stderr.decode()
        assert p.stdout.endswith(b"\n") and p.stdout.count(b"\n") == 1
        _compare(json.loads(p.stdout), load_json(os.path.join(d, "out.json")), floor=0.0 if kind == "deriv" else 1.0,
                 abs_floor=1e-25)
    # failure: nothing on stdout, nonzero status
    p = subprocess.run([os.path.abspath(os.path.join(csrc, "arbplf-ll"))], input=b'{"model_and_data": {}}',
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0 and p.stdout == b"" and p.stderr


def test_character_data_file_equals_inline(tmp_path):
    """the binary side channel gives the same tables as the same alignment written inline"""
    import arbplf
    rng = np.random.default_rng(12)
    N, k, S = 9, 4, 3000
    edges = [[8, 0], [8, 7], [7, 1], [7, 6], [6, 2], [6, 5], [5, 3], [5, 4]]
    codes = rng.integers(0, k + 1, (S, N)).astype(np.uint8)
    codes[:, 5:] = k                                     # internal nodes unobserved
    md = {"edges": edges, "edge_rate_coefficients": [0.1, 0.2, 0.05, 0.3, 0.15, 0.02, 0.25, 0.4],
          "rate_matrix": [[0, 1, 2, 1], [1, 0, 1, 2], [2, 1, 0, 1], [1, 2, 1, 0]],
          "rate_divisor": "equilibrium_exit_rate", "root_prior": "equilibrium_distribution",
          "gamma_rate_mixture": {"gamma_shape": 0.7, "gamma_categories": 3},
          "character_definitions": np.vstack([np.eye(k), np.ones((1, k))]).tolist()}
    f = tmp_path / "aln.u8"
    f.write_bytes(codes.tobytes())
    for fn, extra in ((arbplf.arbplf_ll, {"site_reduction": {"aggregation": "sum"}}),
                      (arbplf.arbplf_ll, {"site_reduction": {"selection": [0, 17, 2999]}}),
                      (arbplf.arbplf_deriv, {"site_reduction": {"aggregation": "avg"}}),
                      (arbplf.arbplf_em_update, {"site_reduction": {"aggregation": "sum"}})):
        a = dict(extra, model_and_data=dict(md, character_data=codes.tolist()))
        b = dict(extra, model_and_data=dict(md, character_data_file=str(f)))
        assert json.loads(fn(json.dumps(a))) == json.loads(fn(json.dumps(b)))
