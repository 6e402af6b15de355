#!/usr/bin/env python3
"""Golden fixtures above k = 4: the BASELINE configurations 2-5 at
reduced site counts, evaluated from their JSON form by an implementation that shares nothing with the oracle or
the product but the input: mpmath at 50 digits, transition matrices through the symmetric eigendecomposition of
the (reversible) rate matrix instead of scaling and squaring, Gamma categories through mpmath's incomplete
gamma function and bisection, derivatives both by the forward/backward vectors and, as a self-check, by
substituting r Q P on one edge at a time (the reference's formulation, src/arbplfderiv.c:112-207).

Also writes the closed-form known answers of the equal-rates k-state model for k = 20 and 61
(p_ii = 1/k + (k-1)/k e^{-k mu t}, p_ij = 1/k - 1/k e^{-k mu t}): two-leaf and three-leaf trees whose site
likelihoods are elementary functions.

Run here (CPU container) once: python tools/make_synth_fixtures.py ; commits tests/golden/synth/*.json.
mpmath 1.3.0 is importable in this image; it is not needed at test time.
"""
import json
import os
import sys
import time

import numpy as np
from mpmath import mp, mpf, matrix

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "synth")
mp.dps = 50


def matmul(X, Y, k):
    return [[sum(X[i][l] * Y[l][j] for l in range(k)) for j in range(k)] for i in range(k)]


def stationary(Q, k):
    """pi Q = 0, sum pi = 1 (diagonal of the input ignored)"""
    A = matrix(k, k)
    for i in range(k):
        for j in range(k):
            if i != j:
                A[i, j] = mpf(Q[i][j])
        A[i, i] = -sum(mpf(Q[i][j]) for j in range(k) if j != i)
    M = A.T.copy()
    for j in range(k):
        M[k - 1, j] = 1
    b = matrix(k, 1)
    b[k - 1] = 1
    pi = mp.lu_solve(M, b)
    return A, [pi[i] for i in range(k)]


def gamma_mean_rates(shape, n):
    """Yang 1994 mean discretisation of Gamma(shape, rate = shape) into n equiprobable categories"""
    a = mpf(shape)
    cdf = lambda x: mp.gammainc(a, 0, a * x, regularized=True)
    cuts = [mpf(0)]
    for i in range(1, n):
        target = mpf(i) / n
        lo, hi = mpf(0), mpf(1)
        while cdf(hi) < target:
            hi *= 2
        for _ in range(200):
            mid = (lo + hi) / 2
            if cdf(mid) < target:
                lo = mid
            else:
                hi = mid
        cuts.append((lo + hi) / 2)
    cdf1 = lambda x: mp.gammainc(a + 1, 0, a * x, regularized=True)
    rates = []
    for i in range(n):
        upper = mpf(1) if i == n - 1 else cdf1(cuts[i + 1])
        rates.append(n * (upper - cdf1(cuts[i])))
    return rates


def expm_taylor(A, k, s):
    """exp(A s) by scaling to norm <= 1/16 and 40 Taylor terms, then squaring (any matrix; 50-digit arithmetic)"""
    X = [[A[i, j] * s for j in range(k)] for i in range(k)]
    norm = max(sum(abs(x) for x in row) for row in X)
    q = 0
    while norm > mpf(1) / 16:
        norm /= 2
        q += 1
    X = [[x / 2 ** q for x in row] for row in X]
    out = [[(mpf(1) if i == j else mpf(0)) + X[i][j] for j in range(k)] for i in range(k)]
    term = X
    for n in range(2, 41):
        term = [[x / n for x in row] for row in matmul(term, X, k)]
        out = [[out[i][j] + term[i][j] for j in range(k)] for i in range(k)]
    for _ in range(q):
        out = matmul(out, out, k)
    return out


def transition_matrices(A, pi, k, scales):
    """exp(A s) for each s: through the symmetric eigendecomposition of D^1/2 A D^-1/2 when the matrix satisfies
    detailed balance exactly (as the binary numbers it is given in), by Taylor series otherwise"""
    reversible = all(abs(pi[i] * A[i, j] - pi[j] * A[j, i]) < mpf(10) ** -40 for i in range(k) for j in range(k))
    if not reversible:
        return [expm_taylor(A, k, s) for s in scales]
    sq = [mp.sqrt(p) for p in pi]
    Sm = matrix(k, k)
    for i in range(k):
        for j in range(k):
            Sm[i, j] = sq[i] * A[i, j] / sq[j]
    Sm = (Sm + Sm.T) / 2
    lam, V = mp.eigsy(Sm)
    out = []
    for s in scales:
        ex = [mp.exp(lam[m] * s) for m in range(k)]
        P = [[None] * k for _ in range(k)]
        for i in range(k):
            vi = [V[i, m] * ex[m] for m in range(k)]
            for j in range(k):
                acc = mpf(0)
                for m in range(k):
                    acc += vi[m] * V[j, m]
                P[i][j] = acc * sq[j] / sq[i]
        out.append(P)
    return out


def evaluate(md, deriv_edges, marg_nodes):
    edges = md["edges"]
    E, N = len(edges), len(edges) + 1
    k = len(md["rate_matrix"])
    A, pi = stationary(md["rate_matrix"], k)
    exit_rate = sum(pi[i] * -A[i, i] for i in range(k))
    assert md["rate_divisor"] == "equilibrium_exit_rate" and md["root_prior"] == "equilibrium_distribution"
    if "gamma_rate_mixture" in md:
        g = md["gamma_rate_mixture"]
        rates = gamma_mean_rates(g["gamma_shape"], g["gamma_categories"])
        prior = [mpf(1) / len(rates)] * len(rates)
    else:
        rates, prior = [mpf(1)], [mpf(1)]
    An = A / exit_rate
    t = [mpf(x) for x in md["edge_rate_coefficients"]]
    children = [[] for _ in range(N)]
    parent = [-1] * N
    for e, (a, b) in enumerate(edges):
        children[a].append((e, b))
        parent[b] = a
    root = [a for a in range(N) if parent[a] < 0][0]
    order = [root]
    for a in order:
        order += [b for _, b in children[a]]
    defs = [[mpf(v) for v in row] for row in md["character_definitions"]]
    data = md["character_data"]
    S = len(data)
    C = len(rates)
    P = [transition_matrices(An, pi, k, [rates[c] * t[e] for e in range(E)]) for c in range(C)]
    Anl = [[An[i, j] for j in range(k)] for i in range(k)]
    dP = {}
    for c in range(C):
        for e in deriv_edges:
            QP = matmul(Anl, P[c][e], k)
            dP[c, e] = [[rates[c] * QP[i][j] for j in range(k)] for i in range(k)]

    def down(site, c, subst=None):
        """partial vectors L[a]; subst = (edge, matrix) replaces P on that edge"""
        L = [None] * N
        M = [None] * E
        for a in reversed(order):
            v = list(defs[data[site][a]])
            for e, b in children[a]:
                Pm = subst[1] if subst and subst[0] == e else P[c][e]
                Lb = L[b]
                nz = [j for j in range(k) if Lb[j] != 0]
                msg = [sum(Pm[i][j] * Lb[j] for j in nz) for i in range(k)]
                M[e] = msg
                v = [v[i] * msg[i] for i in range(k)]
            L[a] = v
        return L, M

    ll, dv, mv = [], [], []
    for s in range(S):
        lh = mpf(0)
        Ls, Ms = [], []
        for c in range(C):
            L, M = down(s, c)
            Ls.append(L)
            Ms.append(M)
            lh += prior[c] * sum(pi[i] * L[root][i] for i in range(k))
        ll.append(mp.log(lh))
        # forward (outside) vectors
        dsite = {e: mpf(0) for e in deriv_edges}
        msite = {a: [mpf(0)] * k for a in marg_nodes}
        for c in range(C):
            L, M = Ls[c], Ms[c]
            F = [None] * N
            F[root] = list(pi)
            for a in order:
                base = [F[a][i] * defs[data[s][a]][i] for i in range(k)]
                for e, b in children[a]:
                    fe = list(base)
                    for e2, _ in children[a]:
                        if e2 != e:
                            fe = [fe[i] * M[e2][i] for i in range(k)]
                    if e in dsite:
                        y = [sum(dP[c, e][i][j] * L[b][j] for j in range(k)) for i in range(k)]
                        dsite[e] += prior[c] * sum(fe[i] * y[i] for i in range(k))
                    F[b] = [sum(P[c][e][i][j] * fe[i] for i in range(k)) for j in range(k)]
            for a in marg_nodes:
                msite[a] = [msite[a][i] + prior[c] * F[a][i] * L[a][i] for i in range(k)]
        dv.append([dsite[e] / lh for e in deriv_edges])
        mv.append([[x / lh for x in msite[a]] for a in marg_nodes])
        if s == 0 and deriv_edges:
            # self-check: the reference's formulation, r Q P substituted on one edge
            e = deriv_edges[0]
            acc = mpf(0)
            for c in range(C):
                L, _ = down(s, c, subst=(e, dP[c, e]))
                acc += prior[c] * sum(pi[i] * L[root][i] for i in range(k))
            assert abs(acc / lh - dv[0][0]) <= mpf(10) ** -35 * max(1, abs(dv[0][0])), "forward/backward != substitution"
            for a_i, a in enumerate(marg_nodes):
                assert abs(sum(mv[0][a_i]) - 1) < mpf(10) ** -35
    return ll, dv, mv


def closed_form_cases():
    """equal-rates model, all off-diagonal rates mu: p_same = 1/k + (k-1)/k e^{-k mu t}, p_diff = 1/k - 1/k e^{-k mu t}
    in units where rate_divisor = 1.  Two leaves joined at the root (uniform prior): P(x, y) = 1/k * p(t1 + t2)."""
    cases = []
    for k in (20, 61):
        mu = mpf("0.013")
        for (t1, t2) in ((mpf("0.7"), mpf("1.9")), (mpf("25"), mpf("40"))):
            e = mp.exp(-k * mu * (t1 + t2))
            same = mp.log((mpf(1) / k + mpf(k - 1) / k * e) / k)
            diff = mp.log((mpf(1) / k - e / k) / k)
            # d/dt1 of the log likelihoods
            dsame = (-mu * (k - 1) * e) / (mpf(1) / k + mpf(k - 1) / k * e)
            ddiff = (mu * e) / (mpf(1) / k - e / k)
            cases.append(dict(k=k, mu=float(mu), t=[float(t1), float(t2)], ll_same=float(same), ll_diff=float(diff),
                              dll_dt1_same=float(dsame), dll_dt1_diff=float(ddiff)))
    return cases


def main():
    from phyly_amd import synth
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "closed_form_equal_rates.json"), "w") as f:
        json.dump(closed_form_cases(), f, indent=1)
    sizes = {2: 64, 3: 64, 4: 24, 5: 12}
    for cfg in (2, 3, 4, 5):
        t0 = time.time()
        wl = synth.Workload(cfg)
        S = sizes[cfg]
        codes = wl.simulate(S)
        # a few missing leaves so that the constant-vector shortcut is exercised
        codes[1, 3] = wl.k
        codes[5, 7 % S] = wl.k
        md = wl.json_model(codes)
        E = wl.E
        deriv_edges = sorted({0, 1, E // 3, E // 2, E - 2, E - 1})
        internal = [a for a in range(wl.N) if a >= wl.T]
        marg_nodes = [internal[0], internal[len(internal) // 2], internal[-1]]
        ll, dv, mv = evaluate(md, deriv_edges, marg_nodes)
        out = dict(config=cfg, sites=S, codes=codes.astype(int).tolist(), deriv_edges_user_order=deriv_edges,
                   marginal_nodes=marg_nodes, ll=[float(x) for x in ll],
                   deriv=[[float(x) for x in row] for row in dv],
                   marginal=[[[float(x) for x in node] for node in site] for site in mv],
                   note="mpmath %d digits, tools/make_synth_fixtures.py" % mp.dps)
        with open(os.path.join(OUT, "cfg%d.json" % cfg), "w") as f:
            json.dump(out, f)
        print("cfg%d: %d sites, %.0f s, ll[0] = %s" % (cfg, S, time.time() - t0, mp.nstr(ll[0], 20)), flush=True)


if __name__ == "__main__":
    main()
