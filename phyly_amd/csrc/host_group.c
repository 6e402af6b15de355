/*
 * host_group.c -- several GPUs in one process: plk_group_* of include/plk.h (host C over plk_engine handles).
 *
 * Replaces nothing in the reference one for one: its site loops are sequential (src/arbplfll.c:139-170,
 * src/arbplfderiv.c:329-356, src/arbplfmarginal.c:237-256) and meet only in nd_accum_accumulate
 * (src/ndaccum.c:198-254).  Here the loop is cut into contiguous site blocks, one engine (GPU) per block, one host
 * thread per engine while a query runs; per-site results land at their global positions, aggregated results are the
 * engines' {hi, lo} partial sums added in engine order in long double (deterministic, no atomics, no collective).
 * No HIP in this file: it is exercised on the CPU with the stand-in engine of tests/sanitize_host.c.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "plk.h"

#define GROUP_MAX 64

struct plk_group {
    int G;
    plk_engine *eng[GROUP_MAX];
    int device[GROUP_MAX];
    int N, E, k;
    long S, s0[GROUP_MAX + 1];       /* block of engine i: [s0[i], s0[i+1]) */
    int have_patterns;
    char err[512];
};

static int fail(plk_group *g, int rc, const char *msg)
{
    snprintf(g->err, sizeof g->err, "%s", msg);
    return rc;
}

int plk_group_create(plk_group **out, int ndev, const int *devices)
{
    if (!out) return PLK_E_ARG;
    *out = NULL;
    if (ndev < 1 || ndev > GROUP_MAX || !devices) return PLK_E_ARG;
    plk_group *g = calloc(1, sizeof(*g));
    if (!g) return PLK_E_NOMEM;
    g->G = ndev;
    for (int i = 0; i < ndev; i++) {
        g->device[i] = devices[i];
        int rc = plk_create(&g->eng[i], devices[i]);
        if (rc) {
            for (int j = 0; j < i; j++) plk_destroy(g->eng[j]);
            free(g);
            return rc;               /* plk_create_error() has the text */
        }
    }
    *out = g;
    return PLK_OK;
}

void plk_group_destroy(plk_group *g)
{
    if (!g) return;
    for (int i = 0; i < g->G; i++) plk_destroy(g->eng[i]);
    free(g);
}

const char *plk_group_last_error(const plk_group *g) { return g ? g->err : "null group"; }
int plk_group_size(const plk_group *g) { return g ? g->G : 0; }
plk_engine *plk_group_engine(plk_group *g, int i) { return g && i >= 0 && i < g->G ? g->eng[i] : NULL; }

int plk_group_block(const plk_group *g, int i, long *s0, long *s1)
{
    if (!g || i < 0 || i >= g->G || !g->have_patterns) return PLK_E_ARG;
    if (s0) *s0 = g->s0[i];
    if (s1) *s1 = g->s0[i + 1];
    return PLK_OK;
}

/* ---- one host thread per engine ------------------------------------------------------------------------- */
typedef int (*job_fn)(plk_group *g, int i, void *ctx);
typedef struct { plk_group *g; int i; job_fn fn; void *ctx; int rc; } job;

static void *job_main(void *p)
{
    job *j = p;
    j->rc = j->fn(j->g, j->i, j->ctx);
    return NULL;
}

/* runs fn for every engine (skip_empty: only engines whose block is not empty); first failure wins */
static int for_each(plk_group *g, job_fn fn, void *ctx, int skip_empty)
{
    job jobs[GROUP_MAX];
    pthread_t th[GROUP_MAX];
    int started[GROUP_MAX];
    int rc = PLK_OK;
    for (int i = 0; i < g->G; i++) {
        jobs[i].g = g; jobs[i].i = i; jobs[i].fn = fn; jobs[i].ctx = ctx; jobs[i].rc = PLK_OK;
        started[i] = 0;
    }
    for (int i = 1; i < g->G; i++) {
        if (skip_empty && g->s0[i + 1] == g->s0[i]) continue;
        if (pthread_create(&th[i], NULL, job_main, &jobs[i]) == 0) started[i] = 1;
        else job_main(&jobs[i]);     /* no thread: run it here */
    }
    if (!(skip_empty && g->s0[1] == g->s0[0])) job_main(&jobs[0]);
    for (int i = 1; i < g->G; i++) if (started[i]) pthread_join(th[i], NULL);
    for (int i = 0; i < g->G; i++)
        if (jobs[i].rc && !rc) {
            rc = jobs[i].rc;
            snprintf(g->err, sizeof g->err, "engine %d (device %d): %s", i, g->device[i], plk_last_error(g->eng[i]));
        }
    return rc;
}

/* ---- broadcast calls ------------------------------------------------------------------------------------ */
typedef struct { int N; const int *indptr, *indices, *preorder; } tree_ctx;
static int job_tree(plk_group *g, int i, void *p) { tree_ctx *c = p; return plk_set_tree(g->eng[i], c->N, c->indptr, c->indices, c->preorder); }

int plk_group_set_tree(plk_group *g, int N, const int *indptr, const int *indices, const int *preorder)
{
    if (!g) return PLK_E_ARG;
    tree_ctx c = {N, indptr, indices, preorder};
    g->have_patterns = 0;
    int rc = for_each(g, job_tree, &c, 0);
    if (!rc) { g->N = N; g->E = N - 1; g->k = 0; }
    return rc;
}

typedef struct { int k, C, root_mode; const double *Qn, *Qn_lo, *er, *cr, *cp, *rw; } model_ctx;
static int job_model(plk_group *g, int i, void *p)
{
    model_ctx *c = p;
    return plk_set_model(g->eng[i], c->k, c->C, c->Qn, c->Qn_lo, c->er, c->cr, c->cp, c->root_mode, c->rw);
}

int plk_group_set_model(plk_group *g, int k, int C, const double *Qn, const double *Qn_lo, const double *edge_rates_csr,
                        const double *cat_rates, const double *cat_prior, int root_mode, const double *root_w)
{
    if (!g) return PLK_E_ARG;
    model_ctx c = {k, C, root_mode, Qn, Qn_lo, edge_rates_csr, cat_rates, cat_prior, root_w};
    if (g->k != k) g->have_patterns = 0;
    int rc = for_each(g, job_model, &c, 0);
    if (!rc) g->k = k;
    return rc;
}

static int job_rates(plk_group *g, int i, void *p) { return plk_update_edge_rates(g->eng[i], p); }
int plk_group_update_edge_rates(plk_group *g, const double *edge_rates_csr)
{
    if (!g || !edge_rates_csr) return PLK_E_ARG;
    return for_each(g, job_rates, (void *)edge_rates_csr, 0);
}

/* ---- patterns: contiguous blocks of ceil(S / G) sites ----------------------------------------------------- */
static void set_blocks(plk_group *g, long S)
{
    const long per = (S + g->G - 1) / g->G;
    g->S = S;
    for (int i = 0; i <= g->G; i++) {
        long v = (long)i * per;
        g->s0[i] = v < S ? v : S;
    }
}

typedef struct { const uint8_t *codes; int nchar; const double *defs; const double *B; } pat_ctx;

static int job_codes(plk_group *g, int i, void *p)
{
    pat_ctx *c = p;
    const long a = g->s0[i], n = g->s0[i + 1] - a;
    if (g->G == 1) return plk_set_patterns_codes(g->eng[i], n, c->codes, PLK_HOST, c->nchar, c->defs);
    uint8_t *blk = malloc((size_t)g->N * (size_t)n);
    if (!blk) return PLK_E_NOMEM;
    for (int r = 0; r < g->N; r++) memcpy(blk + (size_t)r * n, c->codes + (size_t)r * g->S + a, (size_t)n);
    int rc = plk_set_patterns_codes(g->eng[i], n, blk, PLK_HOST, c->nchar, c->defs);
    free(blk);
    return rc;
}

int plk_group_set_patterns_codes(plk_group *g, long S, const uint8_t *codes, int nchar, const double *defs)
{
    if (!g) return PLK_E_ARG;
    if (S < 1 || !codes || !defs || g->k == 0) return fail(g, PLK_E_ARG, "plk_group_set_patterns_codes: bad arguments or call order");
    set_blocks(g, S);
    pat_ctx c = {codes, nchar, defs, NULL};
    int rc = for_each(g, job_codes, &c, 1);
    g->have_patterns = !rc;
    return rc;
}

static int job_dense(plk_group *g, int i, void *p)
{
    pat_ctx *c = p;
    const long a = g->s0[i], n = g->s0[i + 1] - a;
    if (g->G == 1) return plk_set_patterns_dense(g->eng[i], n, c->B, PLK_HOST);
    const size_t rows = (size_t)g->N * g->k;
    double *blk = malloc(rows * (size_t)n * sizeof(double));
    if (!blk) return PLK_E_NOMEM;
    for (size_t r = 0; r < rows; r++) memcpy(blk + r * (size_t)n, c->B + r * (size_t)g->S + a, (size_t)n * sizeof(double));
    int rc = plk_set_patterns_dense(g->eng[i], n, blk, PLK_HOST);
    free(blk);
    return rc;
}

int plk_group_set_patterns_dense(plk_group *g, long S, const double *B)
{
    if (!g) return PLK_E_ARG;
    if (S < 1 || !B || g->k == 0) return fail(g, PLK_E_ARG, "plk_group_set_patterns_dense: bad arguments or call order");
    set_blocks(g, S);
    pat_ctx c = {NULL, 0, NULL, B};
    int rc = for_each(g, job_dense, &c, 1);
    g->have_patterns = !rc;
    return rc;
}

static int job_weights(plk_group *g, int i, void *p)
{
    const double *w = p;
    return plk_set_site_weights(g->eng[i], w ? w + g->s0[i] : NULL, PLK_HOST);
}

int plk_group_set_site_weights(plk_group *g, const double *w)
{
    if (!g) return PLK_E_ARG;
    if (!g->have_patterns) return fail(g, PLK_E_ARG, "plk_group_set_site_weights: set the patterns first");
    return for_each(g, job_weights, (void *)w, 1);
}

/* ---- queries -------------------------------------------------------------------------------------------- */
/* every engine writes its per-site rows at its global offset and its partial sums into its own slice of `part`;
 * the slices are then added in engine order */
typedef struct {
    int kind;                      /* 0 ll, 1 deriv, 2 marginal, 3 edge expectations, 4 hess */
    const int *mask;
    double *site_out;              /* global buffer or NULL */
    size_t site_row;               /* doubles per site in site_out */
    double *part;                  /* [G][nsum][2] or NULL */
    size_t nsum;
    int nL, coef_mode;
    const double *L_hi, *L_lo;
} q_ctx;

static int job_query(plk_group *g, int i, void *p)
{
    q_ctx *c = p;
    double *site = c->site_out ? c->site_out + (size_t)g->s0[i] * c->site_row : NULL;
    double *sums = c->part ? c->part + (size_t)i * c->nsum * 2 : NULL;
    switch (c->kind) {
    case 0: return plk_ll(g->eng[i], site, PLK_HOST, sums);
    case 1: return plk_deriv(g->eng[i], c->mask, site, sums);
    case 2: return plk_marginal(g->eng[i], c->mask, site, sums);
    case 3: return plk_edge_expect_multi(g->eng[i], c->nL, c->L_hi, c->L_lo, c->coef_mode, c->mask, site, sums);
    default: return plk_hess(g->eng[i], sums);
    }
}

static int run_query(plk_group *g, q_ctx *c, double *sums_out)
{
    if (!g->have_patterns) return fail(g, PLK_E_ARG, "plk_group: tree, model and patterns must be set");
    if (sums_out && g->G == 1) {         /* one engine: its sums are the result, bit for bit */
        c->part = sums_out;
        return for_each(g, job_query, c, 1);
    }
    if (sums_out) {
        c->part = calloc((size_t)g->G * c->nsum * 2 + 2, sizeof(double));
        if (!c->part) return fail(g, PLK_E_NOMEM, "plk_group: out of host memory");
    }
    int rc = for_each(g, job_query, c, 1);
    if (!rc && sums_out) {
        for (size_t r = 0; r < c->nsum; r++) {
            long double acc = 0;
            for (int i = 0; i < g->G; i++) {
                if (g->s0[i + 1] == g->s0[i]) continue;
                acc += (long double)c->part[((size_t)i * c->nsum + r) * 2];
                acc += (long double)c->part[((size_t)i * c->nsum + r) * 2 + 1];
            }
            const double hi = (double)acc;
            sums_out[2 * r] = hi;
            sums_out[2 * r + 1] = (double)(acc - (long double)hi);
        }
    }
    if (sums_out) free(c->part);
    return rc;
}

int plk_group_ll(plk_group *g, double *site_ll_out, double *sum_out)
{
    if (!g) return PLK_E_ARG;
    q_ctx c = {0};
    c.kind = 0; c.site_out = site_ll_out; c.site_row = 1; c.nsum = 1;
    return run_query(g, &c, sum_out);
}

int plk_group_deriv(plk_group *g, const int *edge_mask, double *site_edge_out, double *edge_sums_out)
{
    if (!g) return PLK_E_ARG;
    q_ctx c = {0};
    c.kind = 1; c.mask = edge_mask; c.site_out = site_edge_out; c.site_row = (size_t)g->E; c.nsum = (size_t)g->E;
    return run_query(g, &c, edge_sums_out);
}

int plk_group_marginal(plk_group *g, const int *node_mask, double *site_out, double *sums_out)
{
    if (!g) return PLK_E_ARG;
    q_ctx c = {0};
    c.kind = 2; c.mask = node_mask; c.site_out = site_out; c.site_row = (size_t)g->N * g->k; c.nsum = (size_t)g->N * g->k;
    return run_query(g, &c, sums_out);
}

int plk_group_edge_expect_multi(plk_group *g, int nL, const double *L_hi, const double *L_lo, int coef_mode,
                                const int *edge_mask, double *site_out, double *sums_out)
{
    if (!g) return PLK_E_ARG;
    if (nL < 1) return fail(g, PLK_E_ARG, "plk_group_edge_expect_multi: bad direction count");
    q_ctx c = {0};
    c.kind = 3; c.mask = edge_mask; c.site_out = site_out; c.site_row = (size_t)nL * g->E; c.nsum = (size_t)nL * g->E;
    c.nL = nL; c.L_hi = L_hi; c.L_lo = L_lo; c.coef_mode = coef_mode;
    return run_query(g, &c, sums_out);
}

int plk_group_hess(plk_group *g, double *hess_sums_out)
{
    if (!g || !hess_sums_out) return PLK_E_ARG;
    q_ctx c = {0};
    c.kind = 4; c.nsum = (size_t)g->E * g->E;
    return run_query(g, &c, hess_sums_out);
}
