"""
oracle/arbplf_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU oracle for the arbplf-ll / arbplf-deriv / arbplf-marginal path of
argriffing/phyly.  The JSON boundary (validation, tree construction,
reductions, output table) is restated here in Python; the numerics live in
oracle/plf_core.c (binary128 for the site-agnostic part, long double / double
for the per-site part).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; nothing under phyly_amd/ does.

Reference files restated (paths relative to the reference repo root):
  src/parsemodel.c:786-913   validate_model_and_data
  src/parsereduction.c:20-195 validate_column_reduction
  src/csr_graph.c:29-46,103-229 CSR tree, edge map, BFS order
  src/reduction.c:25-118, src/ndaccum.c:198-437 aggregation + output table
  src/arbplfll.c, src/arbplfderiv.c, src/arbplfmarginal.c  drivers
  src/arbplf.c:209-250       Python string API + RuntimeError on failure

Parity pinning: tests/test_oracle_golden.py compares this oracle with every
ll/deriv/marginal golden vector of the reference's examples/ directory.
"""
import ctypes
import json
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile oracle/plf_core.c -> oracle/liborc.so (gcc, OpenMP, quadmath)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liborc.so"])


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liborc.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        for name in ("orc_prepare", "orc_ll", "orc_deriv", "orc_marginal", "orc_gamma_mixture",
                     "orc_frechet", "orc_edge_expect", "orc_hess"):
            getattr(_LIB, name).restype = ctypes.c_int
    return _LIB


class OracleError(Exception):
    pass


def _fail(msg):
    raise OracleError(msg)


# ---------------------------------------------------------------- JSON types
def _is_int(x):
    return isinstance(x, int) and not isinstance(x, bool)


def _is_num(x):
    return (isinstance(x, (int, float))) and not isinstance(x, bool)


def _exists(x):
    return x is not None


def _strict_keys(obj, required, optional, what):
    if not isinstance(obj, dict):
        _fail("%s: expected an object" % what)
    for key in required:
        if key not in obj:
            _fail("%s: missing key %s" % (what, key))
    for key in obj:
        if key not in required and key not in optional:
            _fail("%s: unexpected key %s" % (what, key))


def _nonneg_array(x, n, what):
    if not isinstance(x, list):
        _fail("%s: not an array" % what)
    if len(x) != n:
        _fail("%s: unexpected array length (actual: %d desired: %d)" % (what, len(x), n))
    for v in x:
        if not _is_num(v):
            _fail("%s: not a number" % what)
        if v < 0:
            _fail("%s: array entries must be nonnegative" % what)
    return [float(v) for v in x]


# ---------------------------------------------------------------- model
class Model:
    pass


def _build_tree(m, edges):
    """src/parsemodel.c:211-368 + src/csr_graph.c"""
    if not isinstance(edges, list):
        _fail("edges: not an array")
    E = len(edges)
    N = E + 1
    indeg = [0] * N
    outdeg = [0] * N
    pairs = []
    for e in edges:
        if not (isinstance(e, list) and len(e) == 2 and _is_int(e[0]) and _is_int(e[1])):
            _fail("edges: each edge must be an array of two integers")
        a, b = e
        for idx in (a, b):
            if idx < 0 or idx >= N:
                _fail("edges: node indices out of range")
        if a == b:
            _fail("edges: edges cannot be loops")
        outdeg[a] += 1
        indeg[b] += 1
        pairs.append((a, b))
    roots = [i for i in range(N) if indeg[i] == 0]
    if len(roots) != 1:
        _fail("edges: exactly one node should have in-degree 0")
    if any(d > 1 for d in indeg):
        _fail("edges: the in-degree of each node must be 0 or 1")
    if any(indeg[i] + outdeg[i] < 1 for i in range(N)):
        _fail("edges: a node is not an endpoint of any edge")
    root = roots[0]
    indptr = [0] * (N + 1)
    for i in range(N):
        indptr[i + 1] = indptr[i] + outdeg[i]
    fill = [0] * N
    indices = [-1] * E
    order = [0] * E  # user edge -> csr index
    for i, (a, b) in enumerate(pairs):
        pos = indptr[a] + fill[a]
        indices[pos] = b
        order[i] = pos
        fill[a] += 1
    # BFS by levels (csr_graph_get_tree_topo_sort)
    visited = [False] * N
    visited[root] = True
    pre = []
    frontier = [root]
    while frontier:
        nxt = []
        for a in frontier:
            pre.append(a)
            for j in range(indptr[a], indptr[a + 1]):
                b = indices[j]
                if visited[b]:
                    _fail("edges: topo sort failed")
                visited[b] = True
                nxt.append(b)
        frontier = nxt
    if len(pre) != N:
        _fail("edges: the topo sort does not reach every node")
    m.N, m.E, m.root = N, E, root
    m.indptr = np.array(indptr, dtype=np.int32)
    m.indices = np.array(indices, dtype=np.int32)
    m.preorder = np.array(pre, dtype=np.int32)
    m.order = order


def parse_model(md):
    """src/parsemodel.c:786-913"""
    m = Model()
    _strict_keys(md, ["edges", "edge_rate_coefficients", "rate_matrix"],
                 ["probability_array", "character_definitions", "character_data",
                  "rate_divisor", "root_prior", "rate_mixture", "gamma_rate_mixture",
                  "normalized_median_gamma_rate_mixture"], "model_and_data")
    g = md.get
    mixtures = sum(1 for key in ("rate_mixture", "gamma_rate_mixture",
                                 "normalized_median_gamma_rate_mixture") if _exists(g(key)))
    if mixtures > 1:
        _fail("conflicting rate mixture options")
    if _exists(g("probability_array")) and _exists(g("character_data")):
        _fail("probability_array and character_data are mutually exclusive")
    if _exists(g("probability_array")) and _exists(g("character_definitions")):
        _fail("probability_array and character_definitions are mutually exclusive")

    _build_tree(m, md["edges"])
    coeffs = _nonneg_array(md["edge_rate_coefficients"], m.E, "edge_rate_coefficients")
    m.edge_rates_csr = np.zeros(m.E)
    for i, pos in enumerate(m.order):
        m.edge_rates_csr[pos] = coeffs[i]

    rm = md["rate_matrix"]
    if not isinstance(rm, list):
        _fail("rate_matrix: not an array")
    k = len(rm)
    for row in rm:
        if not isinstance(row, list):
            _fail("rate_matrix: this row is not an array")
        if len(row) != k:
            _fail("rate_matrix: row length mismatch")
        for v in row:
            if not _is_num(v):
                _fail("rate_matrix: not a number")
            if v < 0:
                _fail("rate_matrix: entries must be nonnegative")
    m.k = k
    m.rate_matrix = np.array(rm, dtype=np.float64).reshape(k, k)

    if _exists(g("probability_array")):
        pa = md["probability_array"]
        if not isinstance(pa, list):
            _fail("probability_array: expected an array")
        for site in pa:
            if not isinstance(site, list):
                _fail("probability_array: expected an array")
            if len(site) != m.N:
                _fail("probability_array: failed to match the number of nodes")
            for row in site:
                _nonneg_array(row, k, "probability_array")
        m.S = len(pa)
        m.B = np.array(pa, dtype=np.float64).reshape(m.S, m.N, k)
    elif _exists(g("character_data")):
        cd, defs = md["character_data"], g("character_definitions")
        if not isinstance(cd, list):
            _fail("character_data: expected an array")
        if not isinstance(defs, list):
            _fail("character_definitions: expected an array")
        dd = [_nonneg_array(row, k, "character_definitions") for row in defs]
        nchar = len(dd)
        for site in cd:
            if not isinstance(site, list):
                _fail("character_data: expected an array")
            if len(site) != m.N:
                _fail("character_data: failed to match the number of nodes")
            for c in site:
                if not _is_int(c):
                    _fail("character_data: character indices must be integers")
                if c < 0 or c >= nchar:
                    _fail("character_data: character index out of range")
        m.S = len(cd)
        dd = np.array(dd, dtype=np.float64).reshape(nchar, k)
        cd = np.array(cd, dtype=np.int64).reshape(m.S, m.N)
        m.B = dd[cd] if m.S else np.zeros((0, m.N, k))
    else:
        _fail("either probability_array or character_data must be specified")

    # rate divisor (src/parsemodel.c:83-126)
    m.use_eq_divisor, m.divisor = 0, 1.0
    rd = g("rate_divisor")
    if _exists(rd):
        if isinstance(rd, str):
            if rd != "equilibrium_exit_rate":
                _fail("rate_divisor: bad string")
            m.use_eq_divisor = 1
        elif _is_num(rd):
            if rd <= 0:
                _fail("rate_divisor: must be positive")
            m.divisor = float(rd)
        else:
            _fail("rate_divisor: bad type")

    # root prior (src/parsemodel.c:130-188); modes as src/model.h:16-21
    rp = g("root_prior")
    m.root_custom = None
    if not _exists(rp):
        m.root_mode = 1
    elif isinstance(rp, str):
        if rp == "equilibrium_distribution":
            m.root_mode = 4
        elif rp == "uniform_distribution":
            m.root_mode = 3
        else:
            _fail("root_prior: bad string")
    else:
        m.root_mode = 2
        m.root_custom = _nonneg_array(rp, k, "root_prior")

    # rate mixtures (src/parsemodel.c:632-783)
    m.mix_mode, m.mix_n = 0, 1
    m.mix_rates = m.mix_prior = None
    m.gamma_shape, m.pinv = 1.0, 0.0
    gm = g("gamma_rate_mixture")
    gmed = g("normalized_median_gamma_rate_mixture")
    if _exists(gm) or _exists(gmed):
        spec = gm if _exists(gm) else gmed
        m.mix_mode = 3 if _exists(gm) else 4
        _strict_keys(spec, ["gamma_shape", "gamma_categories"], ["invariable_prior"], "gamma mixture")
        ip = spec.get("invariable_prior")
        if _exists(ip):
            if not _is_num(ip):
                _fail("invariable_prior: not a number")
            m.pinv = float(ip)
        if not _is_num(spec["gamma_shape"]):
            _fail("gamma_shape: not a number")
        m.gamma_shape = float(spec["gamma_shape"])
        if not _is_int(spec["gamma_categories"]):
            _fail("gamma_categories: not an integer")
        m.mix_n = spec["gamma_categories"]
        # unpinned by the reference: reject nonsensical values
        if m.mix_n <= 0 or not (m.gamma_shape > 0) or not (0 <= m.pinv < 1):
            _fail("gamma mixture: parameter out of range")
    elif _exists(g("rate_mixture")):
        spec = g("rate_mixture")
        _strict_keys(spec, ["rates", "prior"], [], "rate_mixture")
        if not isinstance(spec["rates"], list):
            _fail("rate_mixture: rates is not an array")
        n = len(spec["rates"])
        m.mix_rates = _nonneg_array(spec["rates"], n, "rate_mixture rates")
        m.mix_n = n
        pr = spec["prior"]
        if isinstance(pr, str):
            if pr != "uniform_distribution":
                _fail("rate_mixture: bad prior string")
            m.mix_mode = 2
            m.mix_prior = [0.0] * n
        elif isinstance(pr, list):
            m.mix_prior = _nonneg_array(pr, n, "rate_mixture prior")
            m.mix_mode = 1
        else:
            _fail("rate_mixture: prior must be an array or a string")  # unpinned; rejected
        if n == 0:
            _fail("rate_mixture: empty mixture")
    return m


# ---------------------------------------------------------------- reductions
AGG_NONE, AGG_AVG, AGG_SUM, AGG_WEIGHTED, AGG_ONLY = range(5)


class Reduction:
    pass


def parse_reduction(root, n, name):
    """src/parsereduction.c:162-195"""
    r = Reduction()
    sel = agg = None
    if root is not None:
        _strict_keys(root, [], ["selection", "aggregation"], name + "_reduction")
        # absent keys only; an explicit null is an error in the reference
        if "selection" in root:
            sel = root["selection"]
            if not isinstance(sel, list):
                _fail("%s selection: should be an array" % name)
        if "aggregation" in root:
            agg = root["aggregation"]
            if agg is None:
                _fail("%s aggregation: bad value" % name)
    if sel is None:
        r.selection = list(range(n))
    else:
        for v in sel:
            if not _is_int(v):
                _fail("%s selection: not an integer" % name)
            if v < 0 or v >= n:
                _fail("%s selection: out of range" % name)
        r.selection = list(sel)
    r.weights = None
    if agg is None:
        r.mode = AGG_NONE
    elif isinstance(agg, str):
        if agg == "sum":
            r.mode = AGG_SUM
        elif agg == "avg":
            r.mode = AGG_AVG
        elif agg == "only":
            if len(r.selection) != 1:
                _fail("%s aggregation only: selection length must be 1" % name)
            r.mode = AGG_ONLY
        else:
            _fail("%s aggregation: bad string" % name)
    elif isinstance(agg, list):
        if len(agg) != len(r.selection):
            _fail("%s aggregation: wrong number of weights" % name)
        for v in agg:
            if not _is_num(v):
                _fail("%s aggregation: weights should be numeric" % name)
        r.mode = AGG_WEIGHTED
        r.weights = [float(v) for v in agg]
    else:
        _fail("%s aggregation: bad type" % name)
    r.n = n
    return r


def _axis_weights(r):
    """src/reduction.c:25-118 -> (weights[n] longdouble, divisor)"""
    w = np.zeros(r.n, dtype=np.longdouble)
    div = np.longdouble(1)
    if r.mode == AGG_WEIGHTED:
        for idx, wt in zip(r.selection, r.weights):
            w[idx] += np.longdouble(wt)
    elif r.mode in (AGG_SUM, AGG_AVG):
        for idx in r.selection:
            w[idx] += 1
        if r.mode == AGG_AVG:
            div = np.longdouble(len(r.selection))
    elif r.mode == AGG_ONLY:
        w[r.selection[0]] = 1
    return w, div


def _table(values, reductions, names):
    """src/ndaccum.c:198-254 (accumulate) + :382-437 (table).
    values: ndarray over the full axes (unrequested cells may hold anything).
    A reduction with a `components` attribute (names, index lists) prints those
    component indices instead of its own index (src/ndaccum.c:355-366, :401-409)."""
    arr = np.asarray(values, dtype=np.longdouble)
    keep_names = []
    for ax, (r, name) in enumerate(zip(reductions, names)):
        if r.mode == AGG_NONE:
            comp = getattr(r, "components", None)
            if comp is not None:
                keep_names.extend(comp[0])
            else:
                keep_names.append(name)
            continue
        w, div = _axis_weights(r)
        arr = np.moveaxis(arr, ax, -1)
        used = np.nonzero(w)[0]
        # only requested (selected) cells take part; others may be NaN
        red = np.zeros(arr.shape[:-1], dtype=np.longdouble)
        for idx in used:
            red = red + arr[..., idx] * w[idx] / div
        arr = np.moveaxis(red[..., None], -1, ax)
    rows = []

    def rec(ax, prefix, index):
        if ax == len(reductions):
            v = float(arr[tuple(index)])
            if v == 0.0:
                v = 0.0  # -0.0 -> 0.0 (src/util.c:44-48)
            rows.append(prefix + [v])
            return
        r = reductions[ax]
        if r.mode != AGG_NONE:
            rec(ax + 1, prefix, index + [0])
        else:
            comp = getattr(r, "components", None)
            for idx in r.selection:
                label = [idx] if comp is None else [c[idx] for c in comp[1]]
                rec(ax + 1, prefix + label, index + [idx])

    rec(0, [], [])
    return {"columns": keep_names + ["value"], "data": rows}


# ---------------------------------------------------------------- numerics
def _dptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _iptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


def prepare(m):
    """-> dict(C, cat_rates, cat_prior, pi, Qn, P) via orc_prepare (binary128)."""
    lib = _lib()
    k, E = m.k, m.E
    Cmax = m.mix_n + 1
    cat_rates = np.zeros(Cmax)
    cat_prior = np.zeros(Cmax)
    pi = np.zeros(k)
    Qn = np.zeros((k, k))
    P = np.zeros((Cmax, max(E, 1), k, k))
    need_pi = 1 if (m.root_mode == 4 or m.use_eq_divisor) else 0
    mr = np.array(m.mix_rates if m.mix_rates is not None else [0.0], dtype=np.float64)
    mp = np.array(m.mix_prior if m.mix_prior is not None else [0.0], dtype=np.float64)
    rmat = np.ascontiguousarray(m.rate_matrix)
    er = np.ascontiguousarray(m.edge_rates_csr)
    # P is laid out [C][E][k][k] with the true C; allocate exactly after the call
    Pflat = np.zeros(Cmax * E * k * k)
    Pq = np.zeros(Cmax * E * k * k * 2)   # binary128 copies, 16 bytes per entry
    Qq = np.zeros(k * k * 2)
    C = lib.orc_prepare(ctypes.c_int(k), _dptr(rmat),
                        ctypes.c_int(m.use_eq_divisor), ctypes.c_double(m.divisor), ctypes.c_int(need_pi),
                        ctypes.c_int(m.mix_mode), ctypes.c_int(m.mix_n), _dptr(mr), _dptr(mp),
                        ctypes.c_double(m.gamma_shape), ctypes.c_double(m.pinv),
                        ctypes.c_int(E), _dptr(er),
                        _dptr(cat_rates), _dptr(cat_prior), _dptr(pi), _dptr(Qn), _dptr(Pflat),
                        _dptr(Qq), _dptr(Pq))
    if C < 0:
        _fail("orc_prepare failed")
    P = Pflat[:C * E * k * k].reshape(C, E, k, k).copy()
    root_w = np.zeros(k)
    if m.root_mode == 2:
        root_w = np.array(m.root_custom, dtype=np.float64)
    elif m.root_mode == 4:
        root_w = pi.copy()
    return dict(C=C, cat_rates=cat_rates[:C].copy(), cat_prior=cat_prior[:C].copy(),
                pi=pi, Qn=Qn, P=P, root_w=root_w, Pq=Pq[:C * E * k * k * 2].copy(), Qq=Qq)


def site_ll(m, w, B=None, codes=None, defs=None, precise=1, nthreads=0):
    """Per-site log likelihoods for dense B[S][N][k] or codes[S][N]+defs.
    precise: 0 double port (timed baseline), 1 long double, 2 binary128."""
    precise = int(precise)
    lib = _lib()
    if B is not None:
        B = np.ascontiguousarray(B, dtype=np.float64)
        S = B.shape[0]
        bp, cp, dp = _dptr(B), None, None
    else:
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        defs = np.ascontiguousarray(defs, dtype=np.float64)
        S = codes.shape[0]
        bp = None
        cp = codes.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
        dp = _dptr(defs)
    out = np.zeros(S)
    P = w["Pq"] if precise == 2 else np.ascontiguousarray(w["P"])
    used = lib.orc_ll(ctypes.c_int(m.N), ctypes.c_int(m.E), ctypes.c_int(m.k), ctypes.c_int(w["C"]),
                      _iptr(m.indptr), _iptr(m.indices), _iptr(m.preorder),
                      _dptr(P), _dptr(w["cat_prior"]),
                      ctypes.c_int(m.root_mode), _dptr(w["root_w"]),
                      ctypes.c_long(S), bp, cp, dp,
                      ctypes.c_int(precise), ctypes.c_int(nthreads), _dptr(out))
    return out, used


def site_ll_timed(m, w, codes, defs, precise=0, nthreads=0):
    """site_ll for compact codes with only the C call inside the clock (bench.py's cpu_baseline leg):
    -> (ll, threads used, seconds).  The output array is touched before the clock starts."""
    import time
    lib = _lib()
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    defs = np.ascontiguousarray(defs, dtype=np.float64)
    S = codes.shape[0]
    out = np.empty(S)
    out.fill(0.0)
    P = w["Pq"] if int(precise) == 2 else np.ascontiguousarray(w["P"])
    args = (ctypes.c_int(m.N), ctypes.c_int(m.E), ctypes.c_int(m.k), ctypes.c_int(w["C"]),
            _iptr(m.indptr), _iptr(m.indices), _iptr(m.preorder), _dptr(P), _dptr(w["cat_prior"]),
            ctypes.c_int(m.root_mode), _dptr(w["root_w"]), ctypes.c_long(S), None,
            codes.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), _dptr(defs),
            ctypes.c_int(int(precise)), ctypes.c_int(nthreads), _dptr(out))
    t0 = time.perf_counter()
    used = lib.orc_ll(*args)
    dt = time.perf_counter() - t0
    return out, used, dt


def site_deriv(m, w, B, edge_requested=None, nthreads=0, precise=2):
    lib = _lib()
    B = np.ascontiguousarray(B, dtype=np.float64)
    S = B.shape[0]
    out = np.zeros((S, m.E))
    P = w["Pq"] if precise == 2 else np.ascontiguousarray(w["P"])
    Qn = w["Qq"] if precise == 2 else np.ascontiguousarray(w["Qn"])
    req = None
    if edge_requested is not None:
        edge_requested = np.ascontiguousarray(edge_requested, dtype=np.int32)
        req = _iptr(edge_requested)
    lib.orc_deriv(ctypes.c_int(m.N), ctypes.c_int(m.E), ctypes.c_int(m.k), ctypes.c_int(w["C"]),
                  _iptr(m.indptr), _iptr(m.indices), _iptr(m.preorder),
                  _dptr(P), _dptr(Qn), _dptr(w["cat_prior"]), _dptr(w["cat_rates"]),
                  ctypes.c_int(m.root_mode), _dptr(w["root_w"]),
                  ctypes.c_long(S), _dptr(B), None, None, req, ctypes.c_int(precise),
                  ctypes.c_int(nthreads), _dptr(out))
    return out


def site_marginal(m, w, B, nthreads=0, precise=2):
    lib = _lib()
    B = np.ascontiguousarray(B, dtype=np.float64)
    S = B.shape[0]
    out = np.zeros((S, m.N, m.k))
    P = w["Pq"] if precise == 2 else np.ascontiguousarray(w["P"])
    lib.orc_marginal(ctypes.c_int(m.N), ctypes.c_int(m.E), ctypes.c_int(m.k), ctypes.c_int(w["C"]),
                     _iptr(m.indptr), _iptr(m.indices), _iptr(m.preorder),
                     _dptr(P), _dptr(w["cat_prior"]),
                     ctypes.c_int(m.root_mode), _dptr(w["root_w"]),
                     ctypes.c_long(S), _dptr(B), None, None, ctypes.c_int(precise),
                     ctypes.c_int(nthreads), _dptr(out))
    return out


def gamma_mixture(mode, ncat, shape, pinv=0.0):
    """(rates, prior) of gamma_rate_mixture (mode 3) / normalized median (mode 4)."""
    lib = _lib()
    r = np.zeros(ncat + 1)
    p = np.zeros(ncat + 1)
    C = lib.orc_gamma_mixture(ctypes.c_int(mode), ctypes.c_int(ncat), ctypes.c_double(shape),
                              ctypes.c_double(pinv), _dptr(r), _dptr(p))
    return r[:C], p[:C]


# ---------------------------------------------------------------- drivers
def _load(s):
    def bad_const(name):
        raise ValueError("invalid JSON constant " + name)
    try:
        return json.loads(s, parse_constant=bad_const)
    except ValueError as e:
        _fail("error on json: %s" % e)


def _selected_sites(r_site):
    return sorted(set(r_site.selection))


def _red(root, key, n, name):
    if key in root:
        if root[key] is None:
            _fail(key + ": null")
        return parse_reduction(root[key], n, name)
    return parse_reduction(None, n, name)


def run_ll(root):
    """src/arbplfll.c:250-323"""
    _strict_keys(root, ["model_and_data"], ["site_reduction"], "top level")
    m = parse_model(root["model_and_data"])
    r_site = _red(root, "site_reduction", m.S, "site")
    w = prepare(m)
    vals = np.full(m.S, np.nan)
    sel = _selected_sites(r_site)
    if sel:
        ll, _ = site_ll(m, w, B=m.B[sel])
        vals[sel] = ll
    return _table(vals, [r_site], ["site"])


def run_deriv(root):
    """src/arbplfderiv.c:445-531"""
    _strict_keys(root, ["model_and_data"], ["site_reduction", "edge_reduction"], "top level")
    m = parse_model(root["model_and_data"])
    r_site = _red(root, "site_reduction", m.S, "site")
    r_edge = _red(root, "edge_reduction", m.E, "edge")
    w = prepare(m)
    vals = np.full((m.S, m.E), np.nan)
    sel = _selected_sites(r_site)
    req = np.zeros(m.E, dtype=np.int32)
    for user_edge in set(r_edge.selection):
        req[m.order[user_edge]] = 1
    if sel:
        d = site_deriv(m, w, m.B[sel], req)  # CSR edge order
        user = np.full((len(sel), m.E), np.nan)
        for user_edge, pos in enumerate(m.order):
            if req[pos]:
                user[:, user_edge] = d[:, pos]
        vals[sel] = user
    return _table(vals, [r_site, r_edge], ["site", "edge"])


def run_marginal(root):
    """src/arbplfmarginal.c:348-446"""
    _strict_keys(root, ["model_and_data"],
                 ["site_reduction", "node_reduction", "state_reduction"], "top level")
    m = parse_model(root["model_and_data"])
    r_site = _red(root, "site_reduction", m.S, "site")
    r_node = _red(root, "node_reduction", m.N, "node")
    r_state = _red(root, "state_reduction", m.k, "state")
    w = prepare(m)
    vals = np.full((m.S, m.N, m.k), np.nan)
    sel = _selected_sites(r_site)
    if sel:
        vals[sel] = site_marginal(m, w, m.B[sel])
    return _table(vals, [r_site, r_node, r_state], ["site", "node", "state"])


# ------------------------------------------------ dwell / trans / em-update
def frechet(m, w, Lw, divisor=1.0, mul_by_Q=False, edge_requested=None, precise=2):
    """Frechet matrices [C][E][k][k] (binary128 when precise == 2) for the direction
    L = Lw / divisor (entrywise * Qn when mul_by_Q); src/util.c:501-548."""
    lib = _lib()
    k, E, C = m.k, m.E, w["C"]
    Lw = np.asarray(Lw, dtype=np.longdouble).reshape(k, k)
    hi = np.ascontiguousarray(Lw.astype(np.float64))
    lo = np.ascontiguousarray((Lw - hi.astype(np.longdouble)).astype(np.float64))
    Fq = np.zeros(C * max(E, 1) * k * k * 2)
    Fd = np.zeros(C * max(E, 1) * k * k)
    req = None
    if edge_requested is not None:
        edge_requested = np.ascontiguousarray(edge_requested, dtype=np.int32)
        req = _iptr(edge_requested)
    er = np.ascontiguousarray(m.edge_rates_csr, dtype=np.float64)
    lib.orc_frechet(ctypes.c_int(k), ctypes.c_int(C), ctypes.c_int(E), _dptr(w["Qq"]),
                    _dptr(w["cat_rates"]), _dptr(er), _dptr(hi), _dptr(lo),
                    ctypes.c_double(float(divisor)), ctypes.c_int(1 if mul_by_Q else 0), req,
                    _dptr(Fq), _dptr(Fd))
    return Fq if precise == 2 else Fd


def site_edge_expect(m, w, B, F, coef_mode, edge_requested=None, nthreads=0, precise=2):
    """[S][E] (CSR edge order) conditional edge expectations for Frechet matrices F."""
    lib = _lib()
    B = np.ascontiguousarray(B, dtype=np.float64)
    S = B.shape[0]
    out = np.zeros((S, max(m.E, 1)))
    P = w["Pq"] if precise == 2 else np.ascontiguousarray(w["P"])
    req = None
    if edge_requested is not None:
        edge_requested = np.ascontiguousarray(edge_requested, dtype=np.int32)
        req = _iptr(edge_requested)
    er = np.ascontiguousarray(m.edge_rates_csr, dtype=np.float64)
    lib.orc_edge_expect(ctypes.c_int(m.N), ctypes.c_int(m.E), ctypes.c_int(m.k), ctypes.c_int(w["C"]),
                        _iptr(m.indptr), _iptr(m.indices), _iptr(m.preorder),
                        _dptr(P), _dptr(F), _dptr(w["cat_prior"]), _dptr(w["cat_rates"]), _dptr(er),
                        ctypes.c_int(coef_mode),
                        ctypes.c_int(m.root_mode), _dptr(w["root_w"]),
                        ctypes.c_long(S), _dptr(B), None, None, req, ctypes.c_int(precise),
                        ctypes.c_int(nthreads), _dptr(out))
    return out[:, :m.E]


def parse_pair_reduction(root, k, name):
    """src/parsereduction.c:205-392 validate_column_pair_reduction"""
    r = Reduction()
    sel = agg = None
    if root is not None:
        _strict_keys(root, [], ["selection", "aggregation"], name + "_reduction")
        sel = root.get("selection")        # an explicit null counts as absent (:13-16, :327)
        agg = root.get("aggregation")
    r.weights = None
    if sel is not None:
        if not isinstance(sel, list):
            _fail("%s selection: should be an array" % name)
        first, second = [], []
        for pair in sel:
            if not (isinstance(pair, list) and len(pair) == 2 and all(_is_int(v) for v in pair)):
                _fail("%s selection: expected [int, int]" % name)
            if not (0 <= pair[0] < k and 0 <= pair[1] < k):
                _fail("%s selection: out of range" % name)
            first.append(pair[0])
            second.append(pair[1])
        r.selection = list(range(len(sel)))
        # the aggregation is validated as for a plain column reduction (:333)
        d = {"selection": list(range(len(sel)))}
        if "aggregation" in root:
            d["aggregation"] = root["aggregation"]
        tmp = parse_reduction(d, max(len(sel), 1), name)
        r.mode, r.weights = tmp.mode, tmp.weights
    else:
        if agg is None:
            r.mode = AGG_NONE
        elif agg == "sum":
            r.mode = AGG_SUM
        elif agg == "avg":
            r.mode = AGG_AVG
        else:
            _fail("%s reduction (no selection): only sum or avg" % name)
        first = [a for a in range(k) for b in range(k) if a != b]
        second = [b for a in range(k) for b in range(k) if a != b]
        r.selection = list(range(len(first)))
    r.n = len(r.selection)
    r.first, r.second = first, second
    r.components = (["first_state", "second_state"], [first, second])
    return r


def _edge_request(m, r_edge):
    req = np.zeros(max(m.E, 1), dtype=np.int32)
    for user_edge in set(r_edge.selection):
        req[m.order[user_edge]] = 1
    return req


def _to_user_edges(m, d, req):
    user = np.full((d.shape[0], m.E), np.nan)
    for user_edge, pos in enumerate(m.order):
        if req[pos]:
            user[:, user_edge] = d[:, pos]
    return user


def run_dwell(root):
    """src/arbplfdwell.c:303-610"""
    _strict_keys(root, ["model_and_data"], ["site_reduction", "edge_reduction", "state_reduction"], "top level")
    m = parse_model(root["model_and_data"])
    r_site = _red(root, "site_reduction", m.S, "site")
    r_edge = _red(root, "edge_reduction", m.E, "edge")
    r_state = _red(root, "state_reduction", m.k, "state")
    w = prepare(m)
    sel = _selected_sites(r_site)
    req = _edge_request(m, r_edge)
    k = m.k
    if r_state.mode == AGG_NONE:
        vals = np.full((m.S, m.E, k), np.nan)
        for state in sorted(set(r_state.selection)):
            Lw = np.zeros((k, k))
            Lw[state, state] = 1
            F = frechet(m, w, Lw, 1.0, False, req)
            if sel:
                vals[sel, :, state] = _to_user_edges(m, site_edge_expect(m, w, m.B[sel], F, 0, req), req)
        return _table(vals, [r_site, r_edge, r_state], ["site", "edge", "state"])
    sw, div = _axis_weights(r_state)
    F = frechet(m, w, np.diag(sw), div, False, req)
    vals = np.full((m.S, m.E), np.nan)
    if sel:
        vals[sel] = _to_user_edges(m, site_edge_expect(m, w, m.B[sel], F, 0, req), req)
    return _table(vals, [r_site, r_edge], ["site", "edge"])


def run_trans(root):
    """src/arbplftrans.c:346-660"""
    _strict_keys(root, ["model_and_data"], ["site_reduction", "edge_reduction", "trans_reduction"], "top level")
    m = parse_model(root["model_and_data"])
    r_site = _red(root, "site_reduction", m.S, "site")
    r_edge = _red(root, "edge_reduction", m.E, "edge")
    tr = root.get("trans_reduction") if "trans_reduction" in root else None
    if "trans_reduction" in root and tr is None:
        _fail("trans_reduction: null")
    r_trans = parse_pair_reduction(tr, m.k, "trans")
    w = prepare(m)
    sel = _selected_sites(r_site)
    req = _edge_request(m, r_edge)
    k = m.k
    if r_trans.mode == AGG_NONE:
        vals = np.full((m.S, m.E, max(r_trans.n, 1)), np.nan)
        for t in r_trans.selection:
            Lw = np.zeros((k, k))
            Lw[r_trans.first[t], r_trans.second[t]] = 1
            F = frechet(m, w, Lw, 1.0, True, req)
            if sel:
                vals[sel, :, t] = _to_user_edges(m, site_edge_expect(m, w, m.B[sel], F, 1, req), req)
        return _table(vals, [r_site, r_edge, r_trans], ["site", "edge", "trans"])
    tw, div = _axis_weights(r_trans)
    Lw = np.zeros((k, k), dtype=np.longdouble)
    for t in range(r_trans.n):
        Lw[r_trans.first[t], r_trans.second[t]] += tw[t]
    F = frechet(m, w, Lw, div, True, req)
    vals = np.full((m.S, m.E), np.nan)
    if sel:
        vals[sel] = _to_user_edges(m, site_edge_expect(m, w, m.B[sel], F, 1, req), req)
    return _table(vals, [r_site, r_edge], ["site", "edge"])


def run_em_update(root):
    """src/arbplfem.c:397-588: one EM update of the edge rate coefficients."""
    _strict_keys(root, ["model_and_data"], ["site_reduction"], "top level")
    m = parse_model(root["model_and_data"])
    r_site = _red(root, "site_reduction", m.S, "site")
    if r_site.mode == AGG_NONE:
        _fail("aggregation over sites is required")
    w = prepare(m)
    k = m.k
    sel = _selected_sites(r_site)
    sw, div = _axis_weights(r_site)
    Fd = frechet(m, w, -np.eye(k), 1.0, True)
    Ft = frechet(m, w, 1.0 - np.eye(k), 1.0, True)
    dwell = np.zeros(m.E, dtype=np.longdouble)
    trans = np.zeros(m.E, dtype=np.longdouble)
    if sel:
        dv = site_edge_expect(m, w, m.B[sel], Fd, 2).astype(np.longdouble)
        tv = site_edge_expect(m, w, m.B[sel], Ft, 2).astype(np.longdouble)
        for i, site in enumerate(sel):
            dwell += dv[i] * sw[site] / div
            trans += tv[i] * sw[site] / div
    rows = []
    for user_edge, pos in enumerate(m.order):
        if trans[pos] == 0:
            v = 0.0
        else:
            v = float(trans[pos] / dwell[pos] * np.longdouble(m.edge_rates_csr[pos]))
        rows.append([user_edge, v])
    return {"columns": ["edge", "value"], "data": rows}


def site_hess(m, w, B, nthreads=0, precise=2):
    """[S][E][E] (CSR edge order) Hessians of the site log likelihoods; NaN blocks for infeasible sites."""
    lib = _lib()
    B = np.ascontiguousarray(B, dtype=np.float64)
    S = B.shape[0]
    out = np.zeros((S, m.E, m.E))
    P = w["Pq"] if precise == 2 else np.ascontiguousarray(w["P"])
    Qn = w["Qq"] if precise == 2 else np.ascontiguousarray(w["Qn"])
    lib.orc_hess(ctypes.c_int(m.N), ctypes.c_int(m.E), ctypes.c_int(m.k), ctypes.c_int(w["C"]),
                 _iptr(m.indptr), _iptr(m.indices), _iptr(m.preorder),
                 _dptr(P), _dptr(Qn), _dptr(w["cat_prior"]), _dptr(w["cat_rates"]),
                 ctypes.c_int(m.root_mode), _dptr(w["root_w"]),
                 ctypes.c_long(S), _dptr(B), None, None, ctypes.c_int(precise),
                 ctypes.c_int(nthreads), _dptr(out))
    return out


def run_hess(root):
    """src/arbplfhess.c:1163-1206 (_parse_second_order) + :1279-1343 (hess_query)"""
    _strict_keys(root, ["model_and_data", "site_reduction"], [], "top level")
    m = parse_model(root["model_and_data"])
    if root["site_reduction"] is None:
        _fail("site_reduction: null")
    r_site = parse_reduction(root["site_reduction"], m.S, "site")
    if r_site.mode == AGG_NONE:
        _fail("aggregation over sites is required")
    w = prepare(m)
    sel = _selected_sites(r_site)
    sw, div = _axis_weights(r_site)
    H = np.zeros((m.E, m.E), dtype=np.longdouble)
    if sel and m.E:
        hs = site_hess(m, w, m.B[sel]).astype(np.longdouble)
        for i, site in enumerate(sel):
            H += hs[i] * sw[site] / div
    rows = []
    for first in range(m.E):
        for second in range(m.E):
            v = float(H[m.order[first], m.order[second]])
            rows.append([first, second, 0.0 if v == 0.0 else v])
    return {"columns": ["first_edge", "second_edge", "value"], "data": rows}


def _string_api(fn, s):
    """src/arbplf.c:209-250: str -> str, RuntimeError on any failure."""
    try:
        root = _load(s)
        if not isinstance(root, dict):
            _fail("top level: expected an object")
        out = fn(root)
        for row in out["data"]:
            if not math.isfinite(row[-1]):
                # the reference would never terminate / cannot print nan; see DESIGN.md
                pass
        return json.dumps(out)
    except OracleError:
        raise RuntimeError("arbplf likelihood error")


def arbplf_ll(s):
    return _string_api(run_ll, s)


def arbplf_deriv(s):
    return _string_api(run_deriv, s)


def arbplf_marginal(s):
    return _string_api(run_marginal, s)


def arbplf_dwell(s):
    return _string_api(run_dwell, s)


def arbplf_trans(s):
    return _string_api(run_trans, s)


def arbplf_em_update(s):
    return _string_api(run_em_update, s)


def arbplf_hess(s):
    return _string_api(run_hess, s)
