"""
phyly_amd.synth -- seeded synthetic workloads for the BASELINE.json configs.

Harness code (bench.py / tests), not part of the likelihood path: it only
builds inputs.  Trees, branch lengths, models and simulated site patterns
follow SURVEY.md section 8(d).  The site simulator is a counter-based hash
(splitmix64 of (seed, site, node)) evaluated with 64-bit integer arithmetic
that is identical in numpy and in torch, so the CPU checker and the GPU see
the same sites for any site range without generating the whole alignment.
"""
import ctypes
import math
import random

import numpy as np

from . import engine as _engine

# ---------------------------------------------------------------- configs
CONFIGS = {
    2: dict(name="HKY85 k=4 T=50 balanced", T=50, k=4, C=1, S=1_000_000, tree="balanced", model="hky85"),
    3: dict(name="GTR+G4 k=4 T=100", T=100, k=4, C=4, S=10_000_000, tree="yule", model="gtr_g4"),
    4: dict(name="Poisson-AA k=20 T=200", T=200, k=20, C=1, S=1_000_000, tree="yule", model="aa20"),
    5: dict(name="codon k=61 T=64", T=64, k=61, C=1, S=500_000, tree="yule", model="codon61"),
}
SEED0 = 20250355


def make_tree(T, kind, seed):
    """-> edges [[parent, child], ...]; leaves 0..T-1, internals T..2T-2, root 2T-2."""
    rng = random.Random(seed)
    edges = []
    nxt = T
    if kind == "balanced":
        level = list(range(T))
        while len(level) > 1:
            up = []
            for i in range(0, len(level) - 1, 2):
                edges.append([nxt, level[i]])
                edges.append([nxt, level[i + 1]])
                up.append(nxt)
                nxt += 1
            if len(level) % 2:
                up.append(level[-1])
            level = up
    else:
        active = list(range(T))
        while len(active) > 1:
            i = rng.randrange(len(active))
            a = active.pop(i)
            j = rng.randrange(len(active))
            b = active.pop(j)
            edges.append([nxt, a])
            edges.append([nxt, b])
            active.append(nxt)
            nxt += 1
    assert nxt == 2 * T - 1
    return edges


def branch_lengths(E, seed):
    rng = random.Random(seed + 7919)
    return [min(1.0, max(1e-4, rng.expovariate(10.0))) for _ in range(E)]


_CODONS = None


def _sense_codons():
    global _CODONS
    if _CODONS is None:
        nts = "TCAG"
        aa = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"
        out = []
        idx = 0
        for a in nts:
            for b in nts:
                for c in nts:
                    if aa[idx] != "*":
                        out.append((a + b + c, aa[idx]))
                    idx += 1
        _CODONS = out
    return _CODONS


def rate_matrix(model):
    """raw rate matrix (list of lists) + mixture description for the JSON / K0 layer."""
    pi4 = [0.1, 0.2, 0.3, 0.4]
    transitions = {(0, 2), (2, 0), (1, 3), (3, 1)}  # state order A C G T style: 0<->2, 1<->3
    if model == "hky85":
        Q = [[0.0 if i == j else pi4[j] * (2.0 if (i, j) in transitions else 1.0) for j in range(4)] for i in range(4)]
        return Q, None
    if model == "gtr_g4":
        ex = {(0, 1): 1.0, (0, 2): 2.0, (0, 3): 0.5, (1, 2): 1.5, (1, 3): 3.0, (2, 3): 1.0}
        Q = [[0.0 if i == j else pi4[j] * ex[(min(i, j), max(i, j))] for j in range(4)] for i in range(4)]
        return Q, dict(gamma_shape=0.5, gamma_categories=4)
    if model == "aa20":
        Q = [[0.0 if i == j else 1.0 for j in range(20)] for i in range(20)]
        return Q, None
    if model == "codon61":
        cod = _sense_codons()
        n = len(cod)
        assert n == 61
        purines = set("AG")
        Q = [[0.0] * n for _ in range(n)]
        for i, (ci, ai) in enumerate(cod):
            for j, (cj, aj) in enumerate(cod):
                diff = [p for p in range(3) if ci[p] != cj[p]]
                if len(diff) != 1:
                    continue
                x, y = ci[diff[0]], cj[diff[0]]
                ts = (x in purines) == (y in purines)
                Q[i][j] = (1.0 / 61.0) * (2.0 if ts else 1.0) * (1.0 if ai == aj else 0.2)
        return Q, None
    raise ValueError(model)


def csr_from_edges(edges):
    """CSR out-adjacency, BFS preorder and user-edge -> CSR-edge map, as the host
    layer builds them (reference: src/csr_graph.c)."""
    E = len(edges)
    N = E + 1
    outdeg = [0] * N
    indeg = [0] * N
    for a, b in edges:
        outdeg[a] += 1
        indeg[b] += 1
    root = [i for i in range(N) if indeg[i] == 0][0]
    indptr = [0] * (N + 1)
    for i in range(N):
        indptr[i + 1] = indptr[i] + outdeg[i]
    fill = [0] * N
    indices = [0] * E
    order = [0] * E
    for i, (a, b) in enumerate(edges):
        pos = indptr[a] + fill[a]
        indices[pos] = b
        order[i] = pos
        fill[a] += 1
    pre = []
    frontier = [root]
    while frontier:
        nxt = []
        for a in frontier:
            pre.append(a)
            nxt.extend(indices[indptr[a]:indptr[a + 1]])
        frontier = nxt
    return (np.array(indptr, np.int32), np.array(indices, np.int32), np.array(pre, np.int32), order)


class _K0Mixture(ctypes.Structure):
    _fields_ = [("mode", ctypes.c_int), ("n", ctypes.c_int),
                ("rates", ctypes.POINTER(ctypes.c_double)), ("prior", ctypes.POINTER(ctypes.c_double)),
                ("gamma_shape", ctypes.c_double), ("invariable_prior", ctypes.c_double)]


def k0_prepare(Q, mixture, use_eq_divisor=True, divisor=1.0, need_pi=True):
    """Product host K0 (phyly_amd/csrc/host_k0.c) through ctypes."""
    lib = _engine.load_library()
    Q = np.ascontiguousarray(Q, dtype=np.float64)
    k = Q.shape[0]
    mix = _K0Mixture()
    if mixture is None:
        mix.mode, mix.n = 1, 1
    else:
        mix.mode = 4
        mix.n = int(mixture["gamma_categories"])
        mix.gamma_shape = float(mixture["gamma_shape"])
        mix.invariable_prior = float(mixture.get("invariable_prior", 0.0))
    lib.arbplf_k0_category_count.argtypes = [ctypes.POINTER(_K0Mixture)]
    C = lib.arbplf_k0_category_count(ctypes.byref(mix))
    rates = np.zeros(C)
    prior = np.zeros(C)
    pi = np.zeros(k)
    Qn = np.zeros((k, k))
    Qn_lo = np.zeros((k, k))
    dp = ctypes.POINTER(ctypes.c_double)
    lib.arbplf_k0_prepare.argtypes = [ctypes.c_int, dp, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                      ctypes.POINTER(_K0Mixture), dp, dp, dp, dp, dp]
    rc = lib.arbplf_k0_prepare(k, Q.ctypes.data_as(dp), int(use_eq_divisor), float(divisor), int(need_pi),
                               ctypes.byref(mix), rates.ctypes.data_as(dp), prior.ctypes.data_as(dp),
                               pi.ctypes.data_as(dp), Qn.ctypes.data_as(dp), Qn_lo.ctypes.data_as(dp))
    if rc != C:
        raise RuntimeError("arbplf_k0_prepare failed")
    return dict(C=C, cat_rates=rates, cat_prior=prior, pi=pi, Qn=Qn, Qn_lo=Qn_lo)


# ---------------------------------------------------------------- simulation
def _i64(v):
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


_GOLD = _i64(0x9E3779B97F4A7C15)
_M1 = _i64(0xBF58476D1CE4E5B9)
_M2 = _i64(0x94D049BB133111EB)


def _lsr(x, n):
    """logical shift right of int64 arrays / tensors"""
    return (x >> n) & ((1 << (64 - n)) - 1)


def _splitmix(x):
    z = x + _GOLD
    z = (z ^ _lsr(z, 30)) * _M1
    z = (z ^ _lsr(z, 27)) * _M2
    return z ^ _lsr(z, 31)


def _expm_host(A):
    from scipy.linalg import expm
    return expm(A)


class Workload:
    """Tree + model + simulator for one BASELINE config (or a custom one)."""

    def __init__(self, cfg_id=None, T=None, k=None, tree=None, model=None, seed=None):
        if cfg_id is not None:
            c = CONFIGS[cfg_id]
            T, k, tree, model = c["T"], c["k"], c["tree"], c["model"]
            seed = SEED0 + cfg_id if seed is None else seed
            self.default_S = c["S"]
            self.name = c["name"]
        else:
            self.default_S = 1024
            self.name = "custom %s T=%d" % (model, T)
        self.T, self.k, self.seed = T, k, seed
        self.edges = make_tree(T, tree, seed)
        self.E = len(self.edges)
        self.N = self.E + 1
        self.edge_rates = branch_lengths(self.E, seed)
        self.Q, self.mixture = rate_matrix(model)
        self.indptr, self.indices, self.preorder, self.order = csr_from_edges(self.edges)
        self.edge_rates_csr = np.zeros(self.E)
        for i, pos in enumerate(self.order):
            self.edge_rates_csr[pos] = self.edge_rates[i]
        self.k0 = None
        self._cum = None
        # character definitions: identity + all-ones "missing" (code k)
        self.defs = np.vstack([np.eye(k), np.ones((1, k))])
        self.nchar = k + 1

    def prepare(self):
        """K0 through the product host layer."""
        if self.k0 is None:
            self.k0 = k0_prepare(self.Q, self.mixture, True, 1.0, True)
        return self.k0

    # -- simulator tables (host, float64): cumulative rows of P per (cat, edge)
    def _tables(self):
        if self._cum is None:
            k0 = self.prepare()
            C = k0["C"]
            cum = np.zeros((C, self.E, self.k, self.k))
            for c in range(C):
                for e in range(self.E):
                    P = _expm_host(k0["Qn"] * (k0["cat_rates"][c] * self.edge_rates_csr[e]))
                    P = np.maximum(P, 0)
                    cum[c, e] = np.cumsum(P / P.sum(axis=1, keepdims=True), axis=1)
            self._cum = cum
            self._cum_pi = np.cumsum(k0["pi"] / k0["pi"].sum())
        return self._cum, self._cum_pi

    def simulate(self, S, site0=0, device=None):
        """codes[N][S] uint8: leaves observed (0..k-1), internal nodes = k (missing).
        device=None -> numpy on the host; else a torch device (tensor returned)."""
        cum, cum_pi = self._tables()
        C = cum.shape[0]
        if device is None:
            xp = np
            sites = np.arange(site0, site0 + S, dtype=np.int64)
            tocum = lambda a: a
            zeros = lambda: np.zeros(S, dtype=np.int64)
        else:
            import torch
            xp = torch
            sites = torch.arange(site0, site0 + S, dtype=torch.int64, device=device)
            tocum = lambda a: torch.as_tensor(a, device=device)
            zeros = lambda: torch.zeros(S, dtype=torch.int64, device=device)
        seedmix = _i64(self.seed * 0x2545F4914F6CDD1D)

        def uniform(node):
            z = _splitmix(_splitmix(sites ^ seedmix) + _i64(node * 0x9E3779B97F4A7C15))
            top = _lsr(z, 11)
            if device is None:
                return top.astype(np.float64) * (1.0 / 9007199254740992.0)
            return top.to(xp.float64) * (1.0 / 9007199254740992.0)

        def draw(u, cumrows):
            # cumrows: [S][k] ; state = number of thresholds (first k-1) that u reaches
            st = zeros()
            for j in range(self.k - 1):
                st = st + (u >= cumrows[:, j])
            return st

        cat = _lsr(_splitmix(sites ^ _i64(seedmix + 12345)), 33) % C
        state = [None] * self.N
        root = int(self.preorder[0])
        cpi = tocum(cum_pi)
        u = uniform(root)
        st = zeros()
        for j in range(self.k - 1):
            st = st + (u >= cpi[j])
        state[root] = st
        cumt = tocum(cum)  # [C][E][k][k]
        for a in self.preorder:
            a = int(a)
            for idx in range(self.indptr[a], self.indptr[a + 1]):
                b = int(self.indices[idx])
                rows = cumt[cat, idx, state[a]]  # [S][k]
                state[b] = draw(uniform(b), rows)
            if self.indptr[a + 1] > self.indptr[a] and a != root:
                state[a] = None  # free memory early
        if device is None:
            codes = np.full((self.N, S), self.k, dtype=np.uint8)
            for n in range(self.N):
                if self.indptr[n + 1] == self.indptr[n]:
                    codes[n] = state[n].astype(np.uint8)
        else:
            import torch
            codes = torch.full((self.N, S), self.k, dtype=torch.uint8, device=device)
            for n in range(self.N):
                if self.indptr[n + 1] == self.indptr[n]:
                    codes[n] = state[n].to(torch.uint8)
        return codes

    def random_codes(self, S, seed=1, missing_frac=0.05):
        """uniformly random leaf codes with some missing data (stress test, host)."""
        rng = np.random.default_rng(seed)
        codes = np.full((self.N, S), self.k, dtype=np.uint8)
        for n in range(self.N):
            if self.indptr[n + 1] == self.indptr[n]:
                c = rng.integers(0, self.k, size=S)
                c[rng.random(S) < missing_frac] = self.k
                codes[n] = c
        return codes

    def setup_engine(self, eng):
        k0 = self.prepare()
        eng.set_tree(self.indptr, self.indices, self.preorder)
        eng.set_model(k0["Qn"], self.edge_rates_csr, k0["cat_rates"], k0["cat_prior"],
                      _engine.ROOT_EQUILIBRIUM, k0["pi"], Qn_lo=k0["Qn_lo"])

    def json_model(self, codes_host):
        """model_and_data dict for the JSON boundary (codes_host: [N][S] numpy)."""
        md = {
            "edges": self.edges,
            "edge_rate_coefficients": self.edge_rates,
            "rate_matrix": self.Q,
            "rate_divisor": "equilibrium_exit_rate",
            "root_prior": "equilibrium_distribution",
            "character_definitions": self.defs.tolist(),
            "character_data": np.ascontiguousarray(codes_host.T).astype(int).tolist(),
        }
        if self.mixture is not None:
            md["gamma_rate_mixture"] = dict(self.mixture)
        return md

    # algorithmic bytes / flops per site
    def algorithmic(self):
        C = self.prepare()["C"]
        I = self.N - self.T
        A_ll = 2 * (I - 1) * C * self.k * 8 + self.N + 8
        W_ll = C * self.E * (2 * self.k * self.k + self.k) + 2 * C * self.k
        A_deriv = 6 * (I - 1) * C * self.k * 8 + self.N      # site-aggregated gradient: no per-site output
        return dict(A_ll=A_ll, W_ll=W_ll, compulsory=self.N + 8, A_deriv=A_deriv)
