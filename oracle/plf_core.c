/*
 * oracle/plf_core.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the numerical core of argriffing/phyly's
 * arbplf-ll / arbplf-deriv / arbplf-marginal hot path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * Nothing under phyly_amd/ links, imports or executes it.
 *
 * The reference evaluates everything in Arb ball arithmetic and re-runs at
 * doubling precision until every output rounds to a unique double
 * (src/arbplfll.c:206-224).  This restatement has no interval arithmetic;
 * instead the site-agnostic part ("K0/K1": rate mixture, equilibrium,
 * divisor, P = exp(Q r t)) is evaluated in IEEE binary128 (__float128) and
 * rounded once, and the per-site part is compiled three times: in binary128
 * on the unrounded binary128 P (the checker for deriv/marginal, where
 * cancellation makes rounded-P arithmetic lose digits), in long double (the
 * ll checker: 64-bit mantissa, no underflow for any realistic tree) and in
 * double (the timed "port" CPU baseline, with exact power-of-two rescaling).
 *
 * Parity pinning: tests/test_oracle_golden.py checks this file against every
 * ll / deriv / marginal golden vector under the reference's examples/
 * (copied as data into tests/golden/examples/).
 *
 * Each function cites the reference file:line it follows
 * (paths relative to the reference repository root).
 */
#include <math.h>
#include <quadmath.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef __float128 q128;

/* ------------------------------------------------------------------ */
/* K1: matrix exponential in binary128                                  */
/* follows src/cross_site_ws.c:151-168 (arb_mat_exp of Q*rate*t)        */
/* ------------------------------------------------------------------ */

static void q_matmul(int k, const q128 *A, const q128 *B, q128 *C)
{
    for (int i = 0; i < k; i++)
        for (int j = 0; j < k; j++) {
            q128 acc = 0;
            for (int l = 0; l < k; l++) acc += A[i * k + l] * B[l * k + j];
            C[i * k + j] = acc;
        }
}

/* exp(A) by scaling-and-squaring with a Taylor series, all in binary128. */
static void q_expm(int k, const q128 *A, q128 *out)
{
    size_t n = (size_t)k * k;
    q128 *X = malloc(n * sizeof(q128));
    q128 *T = malloc(n * sizeof(q128));
    q128 *W = malloc(n * sizeof(q128));
    q128 norm = 0;
    for (int i = 0; i < k; i++) {
        q128 s = 0;
        for (int j = 0; j < k; j++) s += fabsq(A[i * k + j]);
        if (s > norm) norm = s;
    }
    int sq = 0;
    while (norm > 0.25Q) { norm *= 0.5Q; sq++; }
    q128 scale = scalbnq(1.0Q, -sq);
    for (size_t i = 0; i < n; i++) X[i] = A[i] * scale;
    /* out = I + X + X^2/2! + ... */
    for (size_t i = 0; i < n; i++) out[i] = 0;
    for (int i = 0; i < k; i++) out[i * k + i] = 1;
    memcpy(T, X, n * sizeof(q128));
    for (int term = 1; term < 200; term++) {
        q128 tn = 0;
        for (size_t i = 0; i < n; i++) { out[i] += T[i]; q128 a = fabsq(T[i]); if (a > tn) tn = a; }
        if (tn < 1e-45Q) break;
        q_matmul(k, T, X, W);
        q128 inv = 1.0Q / (q128)(term + 1);
        for (size_t i = 0; i < n; i++) T[i] = W[i] * inv;
    }
    for (int s = 0; s < sq; s++) {
        q_matmul(k, out, out, W);
        memcpy(out, W, n * sizeof(q128));
    }
    free(X); free(T); free(W);
}

/* ------------------------------------------------------------------ */
/* K0: equilibrium distribution                                         */
/* follows src/equilibrium.c:21-88: solve [Q^T e; e^T 0][pi;l] = 1      */
/* (diagonal of Q ignored, exit rates on the diagonal)                  */
/* ------------------------------------------------------------------ */

static int q_equilibrium(int k, const q128 *Q, q128 *pi)
{
    int n = k + 1;
    q128 *R = calloc((size_t)n * (n + 1), sizeof(q128)); /* augmented */
    for (int i = 0; i < k; i++) {
        q128 exit = 0;
        for (int j = 0; j < k; j++) if (i != j) exit += Q[i * k + j];
        for (int j = 0; j < k; j++) if (i != j) R[i * (n + 1) + j] = Q[j * k + i];
        R[i * (n + 1) + i] = -exit;
        R[i * (n + 1) + k] = 1;
        R[k * (n + 1) + i] = 1;
    }
    for (int i = 0; i < n; i++) R[i * (n + 1) + n] = 1;
    int ok = 1;
    for (int c = 0; c < n; c++) {
        int p = c; q128 best = fabsq(R[c * (n + 1) + c]);
        for (int r = c + 1; r < n; r++) {
            q128 v = fabsq(R[r * (n + 1) + c]);
            if (v > best) { best = v; p = r; }
        }
        if (best == 0) { ok = 0; break; }
        if (p != c)
            for (int j = 0; j <= n; j++) {
                q128 t = R[c * (n + 1) + j]; R[c * (n + 1) + j] = R[p * (n + 1) + j]; R[p * (n + 1) + j] = t;
            }
        for (int r = 0; r < n; r++) {
            if (r == c) continue;
            q128 f = R[r * (n + 1) + c] / R[c * (n + 1) + c];
            if (f == 0) continue;
            for (int j = c; j <= n; j++) R[r * (n + 1) + j] -= f * R[c * (n + 1) + j];
        }
    }
    if (ok)
        for (int i = 0; i < k; i++) pi[i] = R[i * (n + 1) + n] / R[i * (n + 1) + i];
    else
        for (int i = 0; i < k; i++) pi[i] = nanq("");
    free(R);
    return ok;
}

/* ------------------------------------------------------------------ */
/* K0: discretised gamma rates                                          */
/* follows src/gamma_discretization.c:209-370 (unit-scale quantiles,   */
/* regularised lower incomplete gamma) and src/rate_mixture.c:167-229   */
/* ------------------------------------------------------------------ */

/* P(s, x) regularised lower incomplete gamma; all-positive series. */
static q128 q_gamma_p(q128 s, q128 x)
{
    if (!(x > 0)) return 0;
    if (isinfq(x)) return 1;
    q128 lead = s * logq(x) - x - lgammaq(s + 1);
    q128 term = 1, sum = 1;
    for (int n = 1; n < 100000; n++) {
        term *= x / (s + n);
        sum += term;
        if (term < sum * 1e-40Q) break;
    }
    q128 r = expq(lead) * sum;
    if (r > 1) r = 1;
    return r;
}

/* quantile q with P(s, q) = p, bisection on log2(q) then on q */
static q128 q_gamma_quantile(q128 s, q128 p)
{
    q128 lo = -16000, hi = 16000;
    if (q_gamma_p(s, scalbnq(1.0Q, (int)lo)) >= p) return 0; /* below range */
    for (int it = 0; it < 200; it++) {
        q128 mid = 0.5Q * (lo + hi);
        q128 x = exp2q(mid);
        if (q_gamma_p(s, x) < p) lo = mid; else hi = mid;
    }
    q128 a = exp2q(lo), b = exp2q(hi);
    for (int it = 0; it < 200; it++) {
        q128 m = 0.5Q * (a + b);
        if (m <= a || m >= b) break;
        if (q_gamma_p(s, m) < p) a = m; else b = m;
    }
    return 0.5Q * (a + b);
}

/* mode 3: Yang-1994 mean rates (gamma_rates, :299-332)
 * mode 4: normalised medians (normalized_median_gamma_rates, :335-370) */
static void q_gamma_rates(int mode, int n, q128 shape, q128 *rates)
{
    if (mode == 3) {
        q128 *e = malloc((n + 1) * sizeof(q128));
        e[0] = 0; e[n] = 1;
        for (int k = 1; k < n; k++) {
            q128 qt = q_gamma_quantile(shape, (q128)k / (q128)n);
            e[k] = q_gamma_p(shape + 1, qt);
        }
        for (int k = 0; k < n; k++) rates[k] = (e[k + 1] - e[k]) * n;
        free(e);
    } else {
        q128 tot = 0;
        for (int k = 0; k < n; k++) {
            rates[k] = q_gamma_quantile(shape, (q128)(2 * k + 1) / (q128)(2 * n));
            tot += rates[k];
        }
        for (int k = 0; k < n; k++) rates[k] = rates[k] / tot * n;
    }
}

/* exported for tests/test_gamma: rates and priors of a gamma mixture */
int orc_gamma_mixture(int mode, int ncat, double shape, double pinv,
                      double *rates_out, double *prior_out)
{
    int C = ncat + (pinv != 0 ? 1 : 0);
    q128 *r = malloc(C * sizeof(q128));
    q128 p = pinv, q = 1 - p;
    q_gamma_rates(mode, ncat, (q128)shape, r);
    for (int i = 0; i < ncat; i++) { rates_out[i] = (double)(r[i] / q); prior_out[i] = (double)(q / ncat); }
    if (pinv != 0) { rates_out[ncat] = 0; prior_out[ncat] = pinv; }
    free(r);
    return C;
}

/*
 * orc_prepare: the whole cross-site workspace update,
 * src/cross_site_ws.c:200-242 in the reference's order of operations.
 *
 * mix_mode: 0 none, 1 custom(prior array), 2 custom(uniform prior),
 *           3 gamma mean, 4 gamma median (src/rate_mixture.h:11-17)
 * use_eq_divisor: "equilibrium_exit_rate" (src/parsemodel.c:83-126)
 * need_pi: root prior is equilibrium, or divisor uses it (src/model.c:119-123)
 * edge_rates_csr[E]: edge_rate_coefficients permuted to CSR order
 *                    (src/cross_site_ws.c:101-106)
 * Outputs (doubles, rounded once from binary128):
 *   cat_rates[C], cat_prior[C], pi[k], Qn[k*k] (normalised, with diagonal),
 *   P[C][E][k][k]
 * returns the category count C, or -1 on failure.
 */
int orc_prepare(int k, const double *rate_matrix,
                int use_eq_divisor, double divisor_value, int need_pi,
                int mix_mode, int mix_n, const double *mix_rates, const double *mix_prior,
                double gamma_shape, double pinv,
                int E, const double *edge_rates_csr,
                double *cat_rates, double *cat_prior, double *pi_out,
                double *Qn_out, double *P_out,
                void *Qn_q_out, void *P_q_out /* binary128 copies, may be NULL */)
{
    int C;
    q128 *rates, expect = 1;
    /* rate mixture, src/rate_mixture.c:288-339 */
    if (mix_mode == 0) {
        C = 1; rates = malloc(sizeof(q128)); rates[0] = 1; cat_prior[0] = 1;
    } else if (mix_mode == 1 || mix_mode == 2) {
        C = mix_n; rates = malloc(C * sizeof(q128));
        expect = 0;
        for (int i = 0; i < C; i++) {
            rates[i] = mix_rates[i];
            if (mix_mode == 1) { cat_prior[i] = mix_prior[i]; expect += (q128)mix_rates[i] * (q128)mix_prior[i]; }
            else { cat_prior[i] = (double)(1.0Q / (q128)C); expect += (q128)mix_rates[i]; }
        }
        if (mix_mode == 2) expect /= C;
    } else {
        C = mix_n + (pinv != 0 ? 1 : 0);
        rates = calloc(C, sizeof(q128));
        q128 p = pinv, q = 1 - p;
        q_gamma_rates(mix_mode, mix_n, (q128)gamma_shape, rates);
        for (int i = 0; i < mix_n; i++) { rates[i] /= q; cat_prior[i] = (double)(q / mix_n); }
        if (pinv != 0) { rates[mix_n] = 0; cat_prior[mix_n] = pinv; }
        expect = 1;
    }
    for (int i = 0; i < C; i++) cat_rates[i] = (double)rates[i];

    size_t kk = (size_t)k * k;
    q128 *Q = malloc(kk * sizeof(q128));
    for (size_t i = 0; i < kk; i++) Q[i] = rate_matrix[i];
    q128 *pi = malloc(k * sizeof(q128));
    for (int i = 0; i < k; i++) pi[i] = 0;
    if (need_pi) q_equilibrium(k, Q, pi);
    for (int i = 0; i < k; i++) { Q[i * k + i] = 0; pi_out[i] = (double)pi[i]; }
    /* divisor, src/cross_site_ws.c:175-191 */
    q128 divisor;
    if (use_eq_divisor) {
        divisor = 0;
        for (int i = 0; i < k; i++) {
            q128 rs = 0;
            for (int j = 0; j < k; j++) rs += Q[i * k + j];
            divisor += rs * pi[i];
        }
        divisor *= expect;
    } else divisor = divisor_value;
    for (size_t i = 0; i < kk; i++) Q[i] /= divisor;
    for (int i = 0; i < k; i++) {
        q128 rs = 0;
        for (int j = 0; j < k; j++) if (j != i) rs += Q[i * k + j];
        Q[i * k + i] = -rs;
    }
    for (size_t i = 0; i < kk; i++) Qn_out[i] = (double)Q[i];
    if (Qn_q_out) memcpy(Qn_q_out, Q, kk * sizeof(q128));
    /* transition matrices */
    q128 *A = malloc(kk * sizeof(q128)), *X = malloc(kk * sizeof(q128));
    for (int c = 0; c < C; c++)
        for (int e = 0; e < E; e++) {
            q128 s = rates[c] * (q128)edge_rates_csr[e];
            for (size_t i = 0; i < kk; i++) A[i] = Q[i] * s;
            q_expm(k, A, X);
            double *dst = P_out + ((size_t)c * E + e) * kk;
            for (size_t i = 0; i < kk; i++) dst[i] = (double)X[i];
            if (P_q_out) memcpy((q128 *)P_q_out + ((size_t)c * E + e) * kk, X, kk * sizeof(q128));
        }
    free(A); free(X); free(Q); free(pi); free(rates);
    return C;
}

/* ------------------------------------------------------------------ */
/* per-site evaluators, compiled for two scalar types                   */
/* ------------------------------------------------------------------ */

typedef struct {
    int N, E, k, C;
    const int *indptr;   /* N+1, CSR out-adjacency (src/csr_graph.h:17-24) */
    const int *indices;  /* E */
    const int *preorder; /* N, BFS order (src/csr_graph.c:103-177) */
    const void *P;       /* [C][E][k][k], double or binary128 per build */
    const void *Qn;      /* [k][k] */
    const double *cat_prior, *cat_rates;
    int root_mode;       /* 1 none, 2 custom, 3 uniform, 4 equilibrium (src/model.h:16-21) */
    const double *root_w;/* k: custom distribution or pi */
    /* observations: dense B[S][N][k], or codes[S][N] + defs[nchar][k] */
    const double *B; const uint8_t *codes; const double *defs;
} orc_problem;

#define REAL q128
#define PT q128
#define SUF(x) x##_q
#define RESCALE 0
#define LOGF logq
#include "plf_site.inc"
#undef REAL
#undef PT
#undef SUF
#undef RESCALE
#undef LOGF

#define REAL long double
#define PT double
#define SUF(x) x##_ld
#define RESCALE 0
#define LOGF logl
#include "plf_site.inc"
#undef REAL
#undef PT
#undef SUF
#undef RESCALE
#undef LOGF

#define REAL double
#define PT double
#define SUF(x) x##_d
#define RESCALE 1
#define LOGF log
#include "plf_site.inc"
#undef REAL
#undef PT
#undef SUF
#undef RESCALE
#undef LOGF

static void fill_problem(orc_problem *p, int N, int E, int k, int C,
        const int *indptr, const int *indices, const int *preorder,
        const void *P, const void *Qn, const double *cat_prior, const double *cat_rates,
        int root_mode, const double *root_w,
        const double *B, const uint8_t *codes, const double *defs)
{
    p->N = N; p->E = E; p->k = k; p->C = C;
    p->indptr = indptr; p->indices = indices; p->preorder = preorder;
    p->P = P; p->Qn = Qn; p->cat_prior = cat_prior; p->cat_rates = cat_rates;
    p->root_mode = root_mode; p->root_w = root_w;
    p->B = B; p->codes = codes; p->defs = defs;
}

/*
 * Per-site log likelihoods; src/arbplfll.c:110-177.
 * precise != 0: long double checker; precise == 0: double port (timed).
 * nthreads <= 0: all cores.
 */
int orc_ll(int N, int E, int k, int C,
        const int *indptr, const int *indices, const int *preorder,
        const void *P, const double *cat_prior,
        int root_mode, const double *root_w,
        long S, const double *B, const uint8_t *codes, const double *defs,
        int precise /* 0 double port, 1 long double, 2 binary128 (P is binary128) */,
        int nthreads, double *ll_out)
{
    orc_problem pr;
    fill_problem(&pr, N, E, k, C, indptr, indices, preorder, P, NULL, cat_prior, NULL,
                 root_mode, root_w, B, codes, defs);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    int used = 1;
#pragma omp parallel
    {
#ifdef _OPENMP
#pragma omp single
        used = omp_get_num_threads();
#endif
        void *ws = precise == 2 ? site_ws_alloc_q(&pr) : precise ? site_ws_alloc_ld(&pr) : site_ws_alloc_d(&pr);
#pragma omp for schedule(static)
        for (long s = 0; s < S; s++)
            ll_out[s] = precise == 2 ? site_ll_q(&pr, ws, s) : precise ? site_ll_ld(&pr, ws, s) : site_ll_d(&pr, ws, s);
        free(ws);
    }
    return used;
}

/* d ll_s / d edge_rate_coefficient, CSR edge order; src/arbplfderiv.c:112-371 */
int orc_deriv(int N, int E, int k, int C,
        const int *indptr, const int *indices, const int *preorder,
        const void *P, const void *Qn, const double *cat_prior, const double *cat_rates,
        int root_mode, const double *root_w,
        long S, const double *B, const uint8_t *codes, const double *defs,
        const int *edge_requested, int precise /* 1 long double, 2 binary128 */,
        int nthreads, double *deriv_out /* [S][E] */)
{
    orc_problem pr;
    fill_problem(&pr, N, E, k, C, indptr, indices, preorder, P, Qn, cat_prior, cat_rates,
                 root_mode, root_w, B, codes, defs);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    int used = 1;
#pragma omp parallel
    {
#ifdef _OPENMP
#pragma omp single
        used = omp_get_num_threads();
#endif
        void *ws = precise == 2 ? site_ws_alloc_q(&pr) : site_ws_alloc_ld(&pr);
#pragma omp for schedule(static)
        for (long s = 0; s < S; s++) {
            if (precise == 2) site_deriv_q(&pr, ws, s, edge_requested, deriv_out + (size_t)s * E);
            else site_deriv_ld(&pr, ws, s, edge_requested, deriv_out + (size_t)s * E);
        }
        free(ws);
    }
    return used;
}

/* marginal state distributions [S][N][k]; src/arbplfmarginal.c:111-264 */
int orc_marginal(int N, int E, int k, int C,
        const int *indptr, const int *indices, const int *preorder,
        const void *P, const double *cat_prior,
        int root_mode, const double *root_w,
        long S, const double *B, const uint8_t *codes, const double *defs,
        int precise /* 1 long double, 2 binary128 */,
        int nthreads, double *marg_out /* [S][N][k] */)
{
    orc_problem pr;
    fill_problem(&pr, N, E, k, C, indptr, indices, preorder, P, NULL, cat_prior, NULL,
                 root_mode, root_w, B, codes, defs);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    int used = 1;
#pragma omp parallel
    {
#ifdef _OPENMP
#pragma omp single
        used = omp_get_num_threads();
#endif
        void *ws = precise == 2 ? site_ws_alloc_q(&pr) : site_ws_alloc_ld(&pr);
#pragma omp for schedule(static)
        for (long s = 0; s < S; s++) {
            if (precise == 2) site_marginal_q(&pr, ws, s, marg_out + (size_t)s * N * k);
            else site_marginal_ld(&pr, ws, s, marg_out + (size_t)s * N * k);
        }
        free(ws);
    }
    return used;
}

/*
 * Frechet-derivative matrices of the matrix exponential, src/util.c:501-548
 * (_arb_mat_exp_frechet): F[c][e] = top-right k x k block of
 * exp([[A, L], [0, A]]) with A = Qn * cat_rate_c * edge_rate_e.
 * The direction L = (Lw_hi + Lw_lo) / divisor, multiplied entrywise by Qn when
 * mul_by_Q is set:
 *   dwell, one state s     Lw = e_s e_s^T                (src/arbplfdwell.c:117-157)
 *   dwell, aggregated      Lw = diag(state weights)      (src/arbplfdwell.c:159-204)
 *   trans, one pair (i,j)  Lw = e_i e_j^T, mul_by_Q      (src/arbplftrans.c:116-160)
 *   trans, aggregated      Lw = summed pair weights, mul_by_Q (src/arbplftrans.c:162-224)
 *   em-update              Lw = -I resp. 1 - I, mul_by_Q (src/arbplfem.c:100-158)
 * Outputs: binary128 F_q_out and/or rounded F_out, [C][E][k][k]; unrequested
 * edges are left zero.
 */
int orc_frechet(int k, int C, int E, const void *Qn_q, const double *cat_rates,
                const double *edge_rates_csr, const double *Lw_hi, const double *Lw_lo,
                double divisor, int mul_by_Q, const int *edge_requested,
                void *F_q_out, double *F_out)
{
    const q128 *Q = Qn_q;
    int n = 2 * k;
    size_t kk = (size_t)k * k;
    q128 *L = malloc(kk * sizeof(q128));
    q128 *M = calloc((size_t)n * n, sizeof(q128)), *X = malloc((size_t)n * n * sizeof(q128));
    for (size_t i = 0; i < kk; i++) {
        L[i] = ((q128)Lw_hi[i] + (q128)(Lw_lo ? Lw_lo[i] : 0.0));
        if (mul_by_Q) L[i] *= Q[i];
        L[i] /= (q128)divisor;
    }
    for (int c = 0; c < C; c++)
        for (int e = 0; e < E; e++) {
            q128 *dq = F_q_out ? (q128 *)F_q_out + ((size_t)c * E + e) * kk : NULL;
            double *dd = F_out ? F_out + ((size_t)c * E + e) * kk : NULL;
            if (edge_requested && !edge_requested[e]) {
                for (size_t i = 0; i < kk; i++) { if (dq) dq[i] = 0; if (dd) dd[i] = 0; }
                continue;
            }
            q128 s = (q128)cat_rates[c] * (q128)edge_rates_csr[e];
            for (int i = 0; i < k; i++)
                for (int j = 0; j < k; j++) {
                    M[(size_t)i * n + j] = Q[i * k + j] * s;
                    M[(size_t)(k + i) * n + k + j] = Q[i * k + j] * s;
                    M[(size_t)i * n + k + j] = L[i * k + j];
                    M[(size_t)(k + i) * n + j] = 0;
                }
            q_expm(n, M, X);
            for (int i = 0; i < k; i++)
                for (int j = 0; j < k; j++) {
                    q128 v = X[(size_t)i * n + k + j];
                    if (dq) dq[i * k + j] = v;
                    if (dd) dd[i * k + j] = (double)v;
                }
        }
    free(L); free(M); free(X);
    return 0;
}

/* conditional edge expectations [S][E] (CSR edge order); see site_edge_expect in plf_site.inc */
int orc_edge_expect(int N, int E, int k, int C,
        const int *indptr, const int *indices, const int *preorder,
        const void *P, const void *F, const double *cat_prior, const double *cat_rates,
        const double *edge_rates_csr, int coef_mode,
        int root_mode, const double *root_w,
        long S, const double *B, const uint8_t *codes, const double *defs,
        const int *edge_requested, int precise /* 1 long double (double P, F), 2 binary128 */,
        int nthreads, double *out /* [S][E] */)
{
    orc_problem pr;
    fill_problem(&pr, N, E, k, C, indptr, indices, preorder, P, NULL, cat_prior, cat_rates,
                 root_mode, root_w, B, codes, defs);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    int used = 1;
#pragma omp parallel
    {
#ifdef _OPENMP
#pragma omp single
        used = omp_get_num_threads();
#endif
        void *ws = precise == 2 ? site_ws_alloc_q(&pr) : site_ws_alloc_ld(&pr);
#pragma omp for schedule(static)
        for (long s = 0; s < S; s++) {
            if (precise == 2) site_edge_expect_q(&pr, ws, s, edge_requested, F, coef_mode, edge_rates_csr, out + (size_t)s * E);
            else site_edge_expect_ld(&pr, ws, s, edge_requested, F, coef_mode, edge_rates_csr, out + (size_t)s * E);
        }
        free(ws);
    }
    return used;
}

/* per-site Hessians of the log likelihood [S][E][E] (CSR edge order); src/arbplfhess.c:503-760.
 * Returns the number of infeasible sites (their blocks are NaN). */
int orc_hess(int N, int E, int k, int C,
        const int *indptr, const int *indices, const int *preorder,
        const void *P, const void *Qn, const double *cat_prior, const double *cat_rates,
        int root_mode, const double *root_w,
        long S, const double *B, const uint8_t *codes, const double *defs,
        int precise /* 1 long double, 2 binary128 */, int nthreads, double *out)
{
    orc_problem pr;
    fill_problem(&pr, N, E, k, C, indptr, indices, preorder, P, Qn, cat_prior, cat_rates,
                 root_mode, root_w, B, codes, defs);
    if (E < 1) return 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    int bad = 0;
#pragma omp parallel reduction(+:bad)
    {
        void *ws = precise == 2 ? site_ws_alloc_q(&pr) : site_ws_alloc_ld(&pr);
        void *H = malloc((size_t)E * E * sizeof(q128)), *g = malloc((size_t)E * sizeof(q128));
#pragma omp for schedule(static)
        for (long s = 0; s < S; s++) {
            int rc = precise == 2 ? site_hess_q(&pr, ws, s, H, g, out + (size_t)s * E * E)
                                  : site_hess_ld(&pr, ws, s, H, g, out + (size_t)s * E * E);
            if (rc) {
                bad++;
                for (size_t i = 0; i < (size_t)E * E; i++) out[(size_t)s * E * E + i] = NAN;
            }
        }
        free(H); free(g); free(ws);
    }
    return bad;
}
