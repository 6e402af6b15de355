/*
 * host_k0.c -- site-agnostic model preparation on the host ("K0").
 *
 * Product code (host C).  Turns the user's raw rate matrix, rate divisor, root
 * prior and rate-mixture description into what the device engine consumes
 * (plk_set_model): the normalised rate matrix Qn, category rates and priors,
 * and the stationary distribution.  Evaluated in IEEE binary128 (__float128,
 * libquadmath) and rounded once; the normalised matrix is handed over as a
 * double-double (hi, lo) pair so that its rows sum to zero to ~1e-32.
 * P = exp(Qn r t) itself is computed on the GPU.
 *
 * Follows the order of operations of the reference
 * (paths relative to argriffing/phyly):
 *   src/cross_site_ws.c:200-242   cross_site_ws_update_with_edge_rates
 *   src/cross_site_ws.c:175-191   _update_rate_divisor
 *   src/equilibrium.c:21-88       stationary distribution via a bordered solve
 *   src/rate_mixture.c:92-137,167-229,288-339   mixture expectation / gamma / summary
 *   src/gamma_discretization.c:209-370          quantiles and mean / median rates
 */
#include <math.h>
#include <quadmath.h>
#include <stdlib.h>
#include <string.h>

#include "host_k0.h"

typedef __float128 ld;

/* regularised lower incomplete gamma P(s, x), all-positive series:
 * P = x^s e^-x / Gamma(s+1) * sum_{n>=0} x^n / ((s+1)...(s+n)) */
static ld gamma_p(ld s, ld x)
{
    if (!(x > 0)) return 0;
    if (isinfq(x)) return 1;
    ld lead = s * logq(x) - x - lgammaq(s + 1);
    ld term = 1, sum = 1;
    for (int n = 1; n < 200000; n++) {
        term *= x / (s + n);
        sum += term;
        if (term < sum * 1e-36Q) break;
    }
    ld r = expq(lead) * sum;
    return r > 1 ? 1 : r;
}

/* unit-scale quantile: P(s, q) = p.  Bisection on log2 q, then on q. */
static ld gamma_quantile(ld s, ld p)
{
    ld lo = -16000, hi = 16000;
    if (gamma_p(s, ldexpq(1.0Q, (int)lo)) >= p) return 0;
    for (int it = 0; it < 200; it++) {
        ld mid = 0.5Q * (lo + hi);
        if (gamma_p(s, exp2q(mid)) < p) lo = mid; else hi = mid;
    }
    ld a = exp2q(lo), b = exp2q(hi);
    for (int it = 0; it < 200; it++) {
        ld m = 0.5Q * (a + b);
        if (m <= a || m >= b) break;
        if (gamma_p(s, m) < p) a = m; else b = m;
    }
    return 0.5Q * (a + b);
}

static void gamma_rates(int mode, int n, ld shape, ld *rates)
{
    if (mode == K0_MIX_GAMMA) {
        /* Yang 1994 category means: n * (P(s+1, q_{j+1}) - P(s+1, q_j)) */
        ld prev = 0;
        for (int j = 0; j < n; j++) {
            ld next = 1;
            if (j + 1 < n) next = gamma_p(shape + 1, gamma_quantile(shape, (ld)(j + 1) / (ld)n));
            rates[j] = (next - prev) * n;
            prev = next;
        }
    } else {
        ld tot = 0;
        for (int j = 0; j < n; j++) {
            rates[j] = gamma_quantile(shape, (ld)(2 * j + 1) / (ld)(2 * n));
            tot += rates[j];
        }
        for (int j = 0; j < n; j++) rates[j] = rates[j] / tot * n;
    }
}

/* stationary distribution: [Q^T e; e^T 0][pi; l] = 1, diagonal of Q ignored */
static int equilibrium(int k, const ld *Q, ld *pi)
{
    int n = k + 1, w = n + 1;
    ld *R = calloc((size_t)n * w, sizeof(ld));
    if (!R) return -1;
    for (int i = 0; i < k; i++) {
        ld exitr = 0;
        for (int j = 0; j < k; j++) if (j != i) { exitr += Q[i * k + j]; R[i * w + j] = Q[j * k + i]; }
        R[i * w + i] = -exitr;
        R[i * w + k] = 1;
        R[k * w + i] = 1;
    }
    for (int i = 0; i < n; i++) R[i * w + n] = 1;
    int ok = 0;
    for (int c = 0; c < n; c++) {
        int p = c;
        ld best = fabsq(R[c * w + c]);
        for (int r = c + 1; r < n; r++) if (fabsq(R[r * w + c]) > best) { best = fabsq(R[r * w + c]); p = r; }
        if (best == 0) { ok = -1; break; }
        if (p != c) for (int j = 0; j < w; j++) { ld t = R[c * w + j]; R[c * w + j] = R[p * w + j]; R[p * w + j] = t; }
        for (int r = 0; r < n; r++) {
            if (r == c) continue;
            ld f = R[r * w + c] / R[c * w + c];
            if (f != 0) for (int j = c; j < w; j++) R[r * w + j] -= f * R[c * w + j];
        }
    }
    for (int i = 0; i < k; i++) pi[i] = ok ? (ld)NAN : R[i * w + n] / R[i * w + i];
    free(R);
    return ok;
}

int arbplf_k0_category_count(const k0_mixture *mix)
{
    if (mix->mode == K0_MIX_NONE) return 1;
    if (mix->mode == K0_MIX_CUSTOM || mix->mode == K0_MIX_UNIFORM) return mix->n;
    return mix->n + (mix->invariable_prior != 0 ? 1 : 0);
}

int arbplf_k0_prepare(int k, const double *rate_matrix,
                      int use_equilibrium_divisor, double divisor_value, int need_equilibrium,
                      const k0_mixture *mix,
                      double *cat_rates, double *cat_prior, double *pi_out, double *Qn_out, double *Qn_lo_out)
{
    const int C = arbplf_k0_category_count(mix);
    ld expect = 1;
    ld *rates = calloc((size_t)C, sizeof(ld));
    if (!rates) return -1;
    if (mix->mode == K0_MIX_NONE) {
        rates[0] = 1; cat_prior[0] = 1;
    } else if (mix->mode == K0_MIX_CUSTOM) {
        expect = 0;
        for (int i = 0; i < C; i++) { rates[i] = mix->rates[i]; cat_prior[i] = mix->prior[i]; expect += (ld)mix->rates[i] * (ld)mix->prior[i]; }
    } else if (mix->mode == K0_MIX_UNIFORM) {
        expect = 0;
        for (int i = 0; i < C; i++) { rates[i] = mix->rates[i]; cat_prior[i] = (double)(1.0Q / (ld)C); expect += (ld)mix->rates[i]; }
        expect /= C;
    } else {
        ld p = mix->invariable_prior, q = 1 - p;
        gamma_rates(mix->mode, mix->n, (ld)mix->gamma_shape, rates);
        for (int i = 0; i < mix->n; i++) { rates[i] /= q; cat_prior[i] = (double)(q / mix->n); }
        if (mix->invariable_prior != 0) { rates[mix->n] = 0; cat_prior[mix->n] = mix->invariable_prior; }
        expect = 1;
    }
    for (int i = 0; i < C; i++) cat_rates[i] = (double)rates[i];
    free(rates);

    const size_t kk = (size_t)k * k;
    ld *Q = malloc(kk * sizeof(ld)), *pi = calloc((size_t)k, sizeof(ld));
    if (!Q || !pi) { free(Q); free(pi); return -1; }
    for (size_t i = 0; i < kk; i++) Q[i] = rate_matrix[i];
    if (need_equilibrium) equilibrium(k, Q, pi);
    for (int i = 0; i < k; i++) { Q[i * k + i] = 0; pi_out[i] = (double)pi[i]; }
    ld divisor = divisor_value;
    if (use_equilibrium_divisor) {
        divisor = 0;
        for (int i = 0; i < k; i++) {
            ld rs = 0;
            for (int j = 0; j < k; j++) rs += Q[i * k + j];
            divisor += rs * pi[i];
        }
        divisor *= expect;
    }
    for (int i = 0; i < k; i++) {
        ld rs = 0;
        for (int j = 0; j < k; j++) if (j != i) { Q[i * k + j] /= divisor; rs += Q[i * k + j]; }
        Q[i * k + i] = -rs;
    }
    for (size_t i = 0; i < kk; i++) {
        Qn_out[i] = (double)Q[i];
        if (Qn_lo_out) Qn_lo_out[i] = (double)(Q[i] - (ld)Qn_out[i]);
    }
    free(Q); free(pi);
    return C;
}
