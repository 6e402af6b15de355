/*
 * tests/group_check.c -- CPU check of the multi-GPU site partition and reduction (phyly_amd/csrc/host_group.c) with a
 * stand-in engine whose outputs are data-dependent functions of exactly the block it was given, so that a wrong
 * offset, a lost tail block, a weight slice that does not follow its sites or a reduction that skips an engine
 * changes the result.  Built by tests/test_group_partition.py with AddressSanitizer/UBSan and with ThreadSanitizer.
 *
 * The stand-in "likelihoods": ll_s = -sum_a (1 + code[a][s]) * (a + 1) / 64; deriv_{s,e} = ll_s * (e + 1) for
 * requested edges; marginal_{s,a,j} = ll_s + a + j / 8; expectations as deriv scaled by (1 + form index);
 * Hessian entry (i, j) = sum_s w_s ll_s (i + 2 j).  Sums are returned as exact {hi, lo} pairs (long double).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "plk.h"

struct plk_engine { int device, N, E, k; long S; double *ll, *w; char err[64]; };

const char *plk_create_error(void) { return "stand-in"; }
int plk_create(plk_engine **out, int device)
{
    if (device >= 100) { *out = NULL; return PLK_E_DEVICE; }
    plk_engine *h = calloc(1, sizeof(*h));
    h->device = device;
    *out = h;
    return PLK_OK;
}
void plk_destroy(plk_engine *h) { if (h) { free(h->ll); free(h->w); free(h); } }
const char *plk_last_error(const plk_engine *h) { return h->err; }
int plk_set_tree(plk_engine *h, int N, const int *a, const int *b, const int *c) { (void)a; (void)b; (void)c; h->N = N; h->E = N - 1; return PLK_OK; }
int plk_set_model(plk_engine *h, int k, int C, const double *a, const double *b, const double *c, const double *d,
                  const double *e, int m, const double *f) { (void)C; (void)a; (void)b; (void)c; (void)d; (void)e; (void)m; (void)f; h->k = k; return PLK_OK; }
int plk_update_edge_rates(plk_engine *h, const double *r) { (void)h; return r ? PLK_OK : PLK_E_ARG; }
int plk_set_patterns_codes(plk_engine *h, long S, const uint8_t *codes, int where, int nchar, const double *defs)
{
    (void)where; (void)defs;
    free(h->ll); free(h->w); h->w = NULL;
    h->ll = malloc((size_t)S * sizeof(double));
    h->S = S;
    for (long s = 0; s < S; s++) {
        double v = 0;
        for (int a = 0; a < h->N; a++) {
            if (codes[(size_t)a * S + s] >= nchar) { snprintf(h->err, sizeof h->err, "bad code"); return PLK_E_ARG; }
            v -= (1.0 + codes[(size_t)a * S + s]) * (a + 1) / 64.0;
        }
        h->ll[s] = v;
    }
    return PLK_OK;
}
int plk_set_patterns_dense(plk_engine *h, long S, const double *B, int where)
{
    (void)where;
    free(h->ll); free(h->w); h->w = NULL;
    h->ll = malloc((size_t)S * sizeof(double));
    h->S = S;
    for (long s = 0; s < S; s++) {
        double v = 0;
        for (int r = 0; r < h->N * h->k; r++) v -= B[(size_t)r * S + s] * (r + 1) / 64.0;
        h->ll[s] = v;
    }
    return PLK_OK;
}
int plk_set_site_weights(plk_engine *h, const double *w, int where)
{
    (void)where;
    free(h->w); h->w = NULL;
    if (w) { h->w = malloc((size_t)h->S * sizeof(double)); memcpy(h->w, w, (size_t)h->S * sizeof(double)); }
    return PLK_OK;
}
static void put(double *out, long double v) { out[0] = (double)v; out[1] = (double)(v - (long double)out[0]); }
int plk_ll(plk_engine *h, double *site, int where, double *sum)
{
    (void)where;
    long double acc = 0;
    for (long s = 0; s < h->S; s++) { if (site) site[s] = h->ll[s]; acc += (long double)(h->w ? h->w[s] : 1.0) * h->ll[s]; }
    if (sum) put(sum, acc);
    return PLK_OK;
}
int plk_edge_expect_multi(plk_engine *h, int nL, const double *Lh, const double *Ll, int coef, const int *mask, double *site, double *sums)
{
    (void)Lh; (void)Ll; (void)coef;
    for (int m = 0; m < nL; m++)
        for (int e = 0; e < h->E; e++) {
            long double acc = 0;
            for (long s = 0; s < h->S; s++) {
                const double v = (!mask || mask[e]) ? h->ll[s] * (e + 1) * (1 + m) : 0.0;
                if (site) site[((size_t)s * nL + m) * h->E + e] = v;
                acc += (long double)(h->w ? h->w[s] : 1.0) * v;
            }
            if (sums) put(sums + 2 * ((size_t)m * h->E + e), acc);
        }
    return PLK_OK;
}
int plk_deriv(plk_engine *h, const int *mask, double *site, double *sums) { return plk_edge_expect_multi(h, 1, NULL, NULL, 0, mask, site, sums); }
int plk_marginal(plk_engine *h, const int *mask, double *site, double *sums)
{
    for (int a = 0; a < h->N; a++)
        for (int j = 0; j < h->k; j++) {
            long double acc = 0;
            for (long s = 0; s < h->S; s++) {
                const double v = (!mask || mask[a]) ? h->ll[s] + a + j / 8.0 : 0.0;
                if (site) site[((size_t)s * h->N + a) * h->k + j] = v;
                acc += (long double)(h->w ? h->w[s] : 1.0) * v;
            }
            if (sums) put(sums + 2 * ((size_t)a * h->k + j), acc);
        }
    return PLK_OK;
}
int plk_hess(plk_engine *h, double *out)
{
    for (int i = 0; i < h->E; i++)
        for (int j = 0; j < h->E; j++) {
            long double acc = 0;
            for (long s = 0; s < h->S; s++) acc += (long double)(h->w ? h->w[s] : 1.0) * h->ll[s] * (i + 2 * j);
            put(out + 2 * ((size_t)i * h->E + j), acc);
        }
    return PLK_OK;
}

/* ------------------------------------------------------------------------------------------------------------ */
static unsigned long long rng_state = 88172645463325252ULL;
static unsigned rnd(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (unsigned)(rng_state >> 32); }

#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return 1; } } while (0)

static int close_dd(const double *a, const double *b, size_t n, const char *what)
{
    for (size_t r = 0; r < n; r++) {
        const long double x = (long double)a[2 * r] + a[2 * r + 1], y = (long double)b[2 * r] + b[2 * r + 1];
        if (fabsl(x - y) > 1e-15L * fmaxl(1.0L, fabsl(y)))   /* partial sums are added in long double: 2^-64 of the largest partial */ { fprintf(stderr, "%s: sum %zu differs: %.20Lg vs %.20Lg\n", what, r, x, y); return 1; }
    }
    return 0;
}

static int one_case(int N, int k, long S, int G, int dense, int weighted)
{
    const int E = N - 1, nchar = 7;
    int dev[64];
    for (int i = 0; i < G; i++) dev[i] = i % 3;
    plk_group *g1 = NULL, *gG = NULL;
    const int one = 0;
    CHECK(plk_group_create(&g1, 1, &one) == 0 && plk_group_create(&gG, G, dev) == 0, "create");
    CHECK(plk_group_size(gG) == G, "size");
    int *ip = calloc(N + 1, sizeof(int)), *ix = calloc(N, sizeof(int)), *pre = calloc(N, sizeof(int));
    uint8_t *codes = malloc((size_t)N * S);
    double *B = malloc((size_t)N * k * S * sizeof(double)), *w = malloc((size_t)S * sizeof(double));
    double defs[7 * 64] = {0}, Q[64 * 64] = {0}, er[64] = {0}, cr[1] = {1}, cp[1] = {1};
    for (size_t i = 0; i < (size_t)N * S; i++) codes[i] = (uint8_t)(rnd() % nchar);
    for (size_t i = 0; i < (size_t)N * k * S; i++) B[i] = (rnd() % 1000) / 1000.0;
    for (long s = 0; s < S; s++) w[s] = (int)(rnd() % 2001 - 1000) / 500.0;
    int *mask = calloc(N + 1, sizeof(int));
    for (int i = 0; i < N; i++) mask[i] = rnd() % 3 != 0;
    int rc = 1;
    plk_group *gs[2] = {g1, gG};
    double *site[2][4], *sums[2][5];
    const size_t row[4] = {1, (size_t)E, (size_t)N * k, (size_t)3 * E}, ns[5] = {1, (size_t)E, (size_t)N * k, (size_t)3 * E, (size_t)E * E};
    for (int v = 0; v < 2; v++) {
        plk_group *g = gs[v];
        CHECK(plk_group_set_tree(g, N, ip, ix, pre) == 0, "tree");
        CHECK(plk_group_set_model(g, k, 1, Q, NULL, er, cr, cp, PLK_ROOT_NONE, NULL) == 0, "model");
        if (dense) CHECK(plk_group_set_patterns_dense(g, S, B) == 0, "dense: %s", plk_group_last_error(g));
        else CHECK(plk_group_set_patterns_codes(g, S, codes, nchar, defs) == 0, "codes: %s", plk_group_last_error(g));
        CHECK(plk_group_set_site_weights(g, weighted ? w : NULL) == 0, "weights");
        for (int q = 0; q < 4; q++) site[v][q] = calloc(row[q] * (size_t)S + 1, sizeof(double));
        for (int q = 0; q < 5; q++) sums[v][q] = calloc(ns[q] * 2 + 2, sizeof(double));
        CHECK(plk_group_ll(g, site[v][0], sums[v][0]) == 0, "ll");
        CHECK(plk_group_deriv(g, mask, site[v][1], sums[v][1]) == 0, "deriv");
        CHECK(plk_group_marginal(g, mask, site[v][2], sums[v][2]) == 0, "marginal");
        CHECK(plk_group_edge_expect_multi(g, 3, Q, NULL, 0, mask, site[v][3], sums[v][3]) == 0, "expect");
        CHECK(plk_group_hess(g, sums[v][4]) == 0, "hess");
    }
    /* blocks: contiguous, cover [0, S), sizes ceil(S / G) except the tail */
    long prev = 0;
    for (int i = 0; i < G; i++) {
        long a, b;
        CHECK(plk_group_block(gG, i, &a, &b) == 0 && a == prev && b >= a && b - a <= (S + G - 1) / G, "block %d", i);
        prev = b;
    }
    CHECK(prev == S, "blocks do not cover the sites");
    for (int q = 0; q < 4; q++)
        CHECK(!memcmp(site[0][q], site[1][q], row[q] * (size_t)S * sizeof(double)), "per-site output %d differs (N %d S %ld G %d)", q, N, S, G);
    const char *names[5] = {"ll", "deriv", "marginal", "expect", "hess"};
    for (int q = 0; q < 5; q++) if (close_dd(sums[1][q], sums[0][q], ns[q], names[q])) goto done;
    /* a pattern code out of range in the LAST block must fail the whole call and name the engine */
    if (!dense && S >= G) {
        codes[(size_t)(N - 1) * S + (S - 1)] = (uint8_t)nchar;
        CHECK(plk_group_set_patterns_codes(gG, S, codes, nchar, defs) == PLK_E_ARG, "bad code accepted");
        CHECK(strstr(plk_group_last_error(gG), "engine") != NULL, "error text: %s", plk_group_last_error(gG));
        CHECK(plk_group_ll(gG, NULL, sums[1][0]) == PLK_E_ARG, "query after a failed upload");
    }
    rc = 0;
done:
    for (int v = 0; v < 2; v++) { for (int q = 0; q < 4; q++) free(site[v][q]); for (int q = 0; q < 5; q++) free(sums[v][q]); }
    free(ip); free(ix); free(pre); free(codes); free(B); free(w); free(mask);
    plk_group_destroy(g1); plk_group_destroy(gG);
    return rc;
}

int main(void)
{
    int cases = 0;
    const long Ss[] = {1, 2, 3, 7, 8, 9, 63, 64, 65, 1000, 4097};
    for (size_t si = 0; si < sizeof Ss / sizeof *Ss; si++)
        for (int G = 1; G <= 9; G += (G < 4 ? 1 : 5))
            for (int dense = 0; dense < 2; dense++)
                for (int weighted = 0; weighted < 2; weighted++) {
                    if (one_case(2 + (int)(rnd() % 9), dense ? 3 : 4, Ss[si], G, dense, weighted)) {
                        fprintf(stderr, "FAILED: S %ld G %d dense %d weighted %d\n", Ss[si], G, dense, weighted);
                        return 1;
                    }
                    cases++;
                }
    int bad[2] = {0, 100};
    plk_group *g = NULL;
    if (plk_group_create(&g, 2, bad) == 0 || g) { fprintf(stderr, "a missing device must fail the group\n"); return 1; }
    printf("ok %d\n", cases);
    return 0;
}
