/*
 * host_drivers.c -- the three query drivers behind include/arbplf.h (host C).
 *
 * Each driver follows the reference's _parse -> _query -> table sequence
 * (src/arbplfll.c:250-323, src/arbplfderiv.c:445-531, src/arbplfmarginal.c:348-446)
 * with the precision-doubling loop and the per-site Arb evaluation replaced by
 * one pass of the GPU engine (include/plk.h).  Selection / aggregation
 * semantics are those of src/reduction.c:25-118 and src/ndaccum.c:198-437:
 * only selected sites are uploaded and evaluated, aggregated site axes are
 * reduced on the device (double-double), other axes on the host (long double).
 */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "arbplf.h"
#include "plk.h"
#include "host_json.h"
#include "host_k0.h"
#include "host_model.h"

static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static plk_group *g_grp = NULL;
static char g_grp_spec[256];

/* The engines of this process: one per entry of ARBPLF_DEVICES (comma separated device ids; a device may be listed
 * twice), else the single device ARBPLF_DEVICE, else device 0.  The site patterns of a query are cut into one
 * contiguous block per engine (plk_group_*, host_group.c).  The group is cached across queries and rebuilt when the
 * environment names another device list. */
static plk_group *get_group(void)
{
    const char *list = getenv("ARBPLF_DEVICES"), *one = getenv("ARBPLF_DEVICE");
    const char *spec = list && *list ? list : (one && *one ? one : "0");
    if (g_grp && strncmp(spec, g_grp_spec, sizeof g_grp_spec)) { plk_group_destroy(g_grp); g_grp = NULL; }
    if (!g_grp) {
        int dev[64], n = 0;
        const char *p = spec;
        while (*p && n < 64) {
            char *end;
            long v = strtol(p, &end, 10);
            if (end == p || v < 0 || v > 1023) { fprintf(stderr, "error: ARBPLF_DEVICES: expected a comma separated list of device ids\n"); return NULL; }
            dev[n++] = (int)v;
            p = end;
            while (*p == ',' || *p == ' ') p++;
        }
        if (n == 0 || *p) { fprintf(stderr, "error: ARBPLF_DEVICES: expected 1 to 64 device ids\n"); return NULL; }
        if (plk_group_create(&g_grp, n, dev)) {
            fprintf(stderr, "error: %s\n", plk_create_error());
            g_grp = NULL;
            return NULL;
        }
        snprintf(g_grp_spec, sizeof g_grp_spec, "%s", spec);
    }
    return g_grp;
}

void arbplf_shutdown(void)
{
    pthread_mutex_lock(&g_mu);
    if (g_grp) { plk_group_destroy(g_grp); g_grp = NULL; }
    pthread_mutex_unlock(&g_mu);
}

/* ------------------------------------------------------------------ */
typedef struct {
    host_model m;
    host_reduction r_site, r_a, r_b;   /* site + up to two further axes */
    int *pair_first, *pair_second;     /* trans_reduction: state pairs parallel to r_b.selection */
    /* prepared model */
    int C;
    double *cat_rates, *cat_prior, *pi, *Qn, *Qn_lo;
    /* selected unique sites */
    long U;
    long *usites;      /* unique selected sites, ascending */
    long *site_to_u;   /* S entries, -1 when not selected */
    long double *w_site, div_site;  /* per-site aggregation weights (S) */
    plk_group *eng;    /* the engines of this query: one site block per device */
} query;

static void query_init(query *q)
{
    memset(q, 0, sizeof(*q));
    host_model_init(&q->m);
    host_reduction_init(&q->r_site);
    host_reduction_init(&q->r_a);
    host_reduction_init(&q->r_b);
}

static void query_clear(query *q)
{
    host_model_clear(&q->m);
    host_reduction_clear(&q->r_site);
    host_reduction_clear(&q->r_a);
    host_reduction_clear(&q->r_b);
    free(q->cat_rates); free(q->cat_prior); free(q->pi); free(q->Qn); free(q->Qn_lo);
    free(q->usites); free(q->site_to_u); free(q->w_site);
    free(q->pair_first); free(q->pair_second);
}

#define ENG(q, call) do { if (call) { fprintf(stderr, "error: %s\n", plk_group_last_error((q)->eng)); return -1; } } while (0)

/* K0 + engine set-up + upload of the selected sites; returns 0 / -1 */
static int query_prepare(query *q)
{
    host_model *m = &q->m;
    const int k = m->k, N = m->N;
    const long S = m->S;
    /* selected sites */
    q->site_to_u = malloc((size_t)(S + 1) * sizeof(long));
    q->usites = malloc((size_t)(S + 1) * sizeof(long));
    q->w_site = malloc((size_t)(S + 1) * sizeof(long double));
    if (!q->site_to_u || !q->usites || !q->w_site) return -1;
    for (long s = 0; s < S; s++) q->site_to_u[s] = -1;
    for (int i = 0; i < q->r_site.selection_len; i++) q->site_to_u[q->r_site.selection[i]] = 0;
    /* Pattern compression: selected sites with identical observation rows share one
     * evaluated pattern; usites[] holds one representative site per pattern, site_to_u[] maps every
     * selected site to its pattern.  Aggregation weights are summed per pattern further down.  This is
     * what users of the reference do by hand (the BEAST examples ship pre-compressed patterns + weights). */
    q->U = 0;
    {
        const size_t row_bytes = m->codes8 ? (size_t)N : (size_t)N * k * sizeof(double);
        const unsigned char *rows = m->codes8 ? (const unsigned char *)m->codes8 : (const unsigned char *)m->prob;
        long nsel = 0;
        for (long s = 0; s < S; s++) if (q->site_to_u[s] == 0) nsel++;
        size_t tsize = 16;
        while (tsize < (size_t)nsel * 2) tsize <<= 1;
        long *table = malloc(tsize * sizeof(long));
        if (!table) return -1;
        for (size_t i = 0; i < tsize; i++) table[i] = -1;
        for (long s = 0; s < S; s++) {
            if (q->site_to_u[s] != 0) continue;
            const unsigned char *row = rows + (size_t)s * row_bytes;
            unsigned long long hsh = 1469598103934665603ULL;          /* FNV-1a */
            for (size_t b = 0; b < row_bytes; b++) { hsh ^= row[b]; hsh *= 1099511628211ULL; }
            size_t pos = (size_t)hsh & (tsize - 1);
            long found = -1;
            while (table[pos] >= 0) {
                const long p = table[pos];
                if (!memcmp(rows + (size_t)q->usites[p] * row_bytes, row, row_bytes)) { found = p; break; }
                pos = (pos + 1) & (tsize - 1);
            }
            if (found < 0) { found = q->U; table[pos] = found; q->usites[q->U++] = s; }
            q->site_to_u[s] = found;
        }
        free(table);
    }
    q->div_site = 1;
    if (q->r_site.agg_mode != AGG_NONE) host_reduction_weights(&q->r_site, q->w_site, &q->div_site);
    if (q->U == 0) return 0;

    /* K0 (host, long double) */
    const int need_pi = (m->root_mode == HM_ROOT_EQUILIBRIUM) || m->use_equilibrium_divisor;
    q->C = arbplf_k0_category_count(&m->mix);
    q->cat_rates = malloc(q->C * sizeof(double));
    q->cat_prior = malloc(q->C * sizeof(double));
    q->pi = calloc(k, sizeof(double));
    q->Qn = malloc((size_t)k * k * sizeof(double));
    q->Qn_lo = malloc((size_t)k * k * sizeof(double));
    if (!q->cat_rates || !q->cat_prior || !q->pi || !q->Qn || !q->Qn_lo) return -1;
    if (arbplf_k0_prepare(k, m->rate_matrix, m->use_equilibrium_divisor, m->rate_divisor, need_pi, &m->mix,
                          q->cat_rates, q->cat_prior, q->pi, q->Qn, q->Qn_lo) != q->C) {
        fprintf(stderr, "error: model preparation failed\n");
        return -1;
    }
    for (int i = 0; i < k * k; i++)
        if (!isfinite(q->Qn[i])) { fprintf(stderr, "error: the normalised rate matrix is not finite (zero rate divisor or singular equilibrium system)\n"); return -1; }

    q->eng = get_group();
    if (!q->eng) return -1;
    ENG(q, plk_group_set_tree(q->eng, N, m->indptr, m->indices, m->preorder));
    const double *root_w = m->root_mode == HM_ROOT_CUSTOM ? m->root_custom : (m->root_mode == HM_ROOT_EQUILIBRIUM ? q->pi : NULL);
    ENG(q, plk_group_set_model(q->eng, k, q->C, q->Qn, q->Qn_lo, m->edge_rates_csr, q->cat_rates, q->cat_prior, m->root_mode, root_w));

    /* observations of the selected sites, device layout (site fastest) */
    const long U = q->U;
    if (m->codes8) {
        uint8_t *codes = malloc((size_t)N * U);
        if (!codes) return -1;
        for (long u = 0; u < U; u++) {
            const uint8_t *src = m->codes8 + (size_t)q->usites[u] * N;
            for (int a = 0; a < N; a++) codes[(size_t)a * U + u] = src[a];
        }
        int rc = plk_group_set_patterns_codes(q->eng, U, codes, m->nchar, m->defs);
        free(codes);
        ENG(q, rc);
    } else {
        double *B = malloc((size_t)N * k * U * sizeof(double));
        if (!B) return -1;
        for (long u = 0; u < U; u++) {
            const double *src = m->prob + (size_t)q->usites[u] * N * k;
            for (int a = 0; a < N; a++)
                for (int j = 0; j < k; j++) B[((size_t)a * k + j) * U + u] = src[(size_t)a * k + j];
        }
        int rc = plk_group_set_patterns_dense(q->eng, U, B);
        free(B);
        ENG(q, rc);
    }
    if (q->r_site.agg_mode != AGG_NONE) {
        double *w = malloc((size_t)U * sizeof(double));
        if (!w) return -1;
        long double *wp = calloc((size_t)U, sizeof(long double));
        if (!wp) { free(w); return -1; }
        for (long s = 0; s < S; s++) if (q->site_to_u[s] >= 0) wp[q->site_to_u[s]] += q->w_site[s];
        for (long u = 0; u < U; u++) w[u] = (double)wp[u];
        free(wp);
        int rc = plk_group_set_site_weights(q->eng, w);
        free(w);
        ENG(q, rc);
    }
    return 0;
}

static double clean(long double v)
{
    double d = (double)v;
    if (d == 0.0) d = 0.0; /* -0.0 -> 0.0 (src/util.c:44-48) */
    return d;
}

static void table_begin(jbuf *b, const char *const *names, const host_reduction *const *reds, int ndim)
{
    jbuf_puts(b, "{\"columns\": [");
    for (int i = 0; i < ndim; i++)
        if (reds[i]->agg_mode == AGG_NONE) { jbuf_puts(b, "\""); jbuf_puts(b, names[i]); jbuf_puts(b, "\", "); }
    jbuf_puts(b, "\"value\"], \"data\": [");
}

static int check_finite(double v, const char *what)
{
    if (isfinite(v)) return 0;
    fprintf(stderr, "error: %s is not finite (a selected site has likelihood zero, or the computation overflowed); "
                    "the reference does not terminate on such input\n", what);
    return -1;
}

/* _parse of the drivers: kind 0 ll, 1 deriv, 2 marginal, 3 dwell, 4 trans, 5 em-update
 * (src/arbplfll.c:250-288, src/arbplfderiv.c:445-493, src/arbplfmarginal.c:348-405,
 *  src/arbplfdwell.c:507-566, src/arbplftrans.c:553-614, src/arbplfem.c:505-545) */
static int query_parse(query *q, int kind, const jval *root)
{
    static const char *const req[] = {"model_and_data", NULL};
    static const char *const all_ll[] = {"model_and_data", "site_reduction", NULL};
    static const char *const all_deriv[] = {"model_and_data", "site_reduction", "edge_reduction", NULL};
    static const char *const all_marg[] = {"model_and_data", "site_reduction", "node_reduction", "state_reduction", NULL};
    static const char *const all_dwell[] = {"model_and_data", "site_reduction", "edge_reduction", "state_reduction", NULL};
    static const char *const all_trans[] = {"model_and_data", "site_reduction", "edge_reduction", "trans_reduction", NULL};
    const char *const *allowed = kind == 0 ? all_ll : kind == 1 ? all_deriv : kind == 2 ? all_marg :
                                 kind == 3 ? all_dwell : kind == 4 ? all_trans : all_ll;
    if (host_check_keys(root, req, allowed, "input")) return -1;
    if (host_model_parse(&q->m, j_get(root, "model_and_data"))) return -1;
    if (host_reduction_parse(&q->r_site, (int)q->m.S, "site", j_get(root, "site_reduction"))) return -1;
    if (kind == 1 && host_reduction_parse(&q->r_a, q->m.E, "edge", j_get(root, "edge_reduction"))) return -1;
    if (kind == 2 && host_reduction_parse(&q->r_a, q->m.N, "node", j_get(root, "node_reduction"))) return -1;
    if (kind == 2 && host_reduction_parse(&q->r_b, q->m.k, "state", j_get(root, "state_reduction"))) return -1;
    if ((kind == 3 || kind == 4) && host_reduction_parse(&q->r_a, q->m.E, "edge", j_get(root, "edge_reduction"))) return -1;
    if (kind == 3 && host_reduction_parse(&q->r_b, q->m.k, "state", j_get(root, "state_reduction"))) return -1;
    if (kind == 4 && host_pair_reduction_parse(&q->r_b, &q->pair_first, &q->pair_second, q->m.k, "trans",
                                               j_get(root, "trans_reduction"))) return -1;
    if (kind == 6 && !j_get(root, "site_reduction")) { fprintf(stderr, "error: site_reduction is required\n"); return -1; }
    if ((kind == 5 || kind == 6) && q->r_site.agg_mode == AGG_NONE) { fprintf(stderr, "error: aggregation over sites is required\n"); return -1; }
    return 0;
}

/* ------------------------------------------------------------------ ll */
static int run_ll(const jval *root, jbuf *out)
{
    query q;
    int rc = -1;
    double *site_ll = NULL;
    query_init(&q);
    if (query_parse(&q, 0, root)) goto done;
    if (query_prepare(&q)) goto done;
    const char *names[] = {"site"};
    const host_reduction *reds[] = {&q.r_site};
    table_begin(out, names, reds, 1);
    if (q.r_site.agg_mode != AGG_NONE) {
        double sum[2] = {0, 0};
        if (q.U > 0 && plk_group_ll(q.eng, NULL, sum)) { fprintf(stderr, "error: %s\n", plk_group_last_error(q.eng)); goto done; }
        double v = clean(((long double)sum[0] + (long double)sum[1]) / q.div_site);
        if (check_finite(v, "the aggregated log likelihood")) goto done;
        jbuf_puts(out, "[");
        jbuf_real(out, v);
        jbuf_puts(out, "]");
    } else {
        site_ll = malloc((size_t)(q.U + 1) * sizeof(double));
        if (!site_ll) goto done;
        if (q.U > 0 && plk_group_ll(q.eng, site_ll, NULL)) { fprintf(stderr, "error: %s\n", plk_group_last_error(q.eng)); goto done; }
        for (int i = 0; i < q.r_site.selection_len; i++) {
            int s = q.r_site.selection[i];
            double v = clean(site_ll[q.site_to_u[s]]);
            if (check_finite(v, "a site log likelihood")) goto done;
            if (i) jbuf_puts(out, ", ");
            jbuf_puts(out, "["); jbuf_int(out, s); jbuf_puts(out, ", "); jbuf_real(out, v); jbuf_puts(out, "]");
        }
    }
    jbuf_puts(out, "]}");
    rc = 0;
done:
    free(site_ll);
    query_clear(&q);
    return rc;
}

/* ------------------------------------------------------------------ deriv */
static int run_deriv(const jval *root, jbuf *out)
{
    query q;
    int rc = -1;
    int *mask = NULL;
    double *vals = NULL, *sums = NULL;
    long double *w_edge = NULL;
    query_init(&q);
    if (query_parse(&q, 1, root)) goto done;
    if (query_prepare(&q)) goto done;
    const int E = q.m.E;
    const host_reduction *re = &q.r_a;
    mask = calloc(E + 1, sizeof(int));
    w_edge = malloc((size_t)(E + 1) * sizeof(long double));
    if (!mask || !w_edge) goto done;
    for (int i = 0; i < re->selection_len; i++) mask[q.m.edge_order[re->selection[i]]] = 1;
    long double div_edge = 1;
    if (re->agg_mode != AGG_NONE) host_reduction_weights(re, w_edge, &div_edge);
    const int site_agg = q.r_site.agg_mode != AGG_NONE, edge_agg = re->agg_mode != AGG_NONE;
    if (site_agg) {
        sums = calloc((size_t)E * 2 + 2, sizeof(double));
        if (!sums) goto done;
        if (q.U > 0 && re->selection_len > 0 && plk_group_deriv(q.eng, mask, NULL, sums)) { fprintf(stderr, "error: %s\n", plk_group_last_error(q.eng)); goto done; }
    } else {
        vals = calloc((size_t)q.U * E + 1, sizeof(double));
        if (!vals) goto done;
        if (q.U > 0 && re->selection_len > 0 && plk_group_deriv(q.eng, mask, vals, NULL)) { fprintf(stderr, "error: %s\n", plk_group_last_error(q.eng)); goto done; }
    }
    const char *names[] = {"site", "edge"};
    const host_reduction *reds[] = {&q.r_site, re};
    table_begin(out, names, reds, 2);
    int first = 1;
    const int nsite_rows = site_agg ? 1 : q.r_site.selection_len;
    for (int si = 0; si < nsite_rows; si++) {
        const int s = site_agg ? -1 : q.r_site.selection[si];
        const long u = site_agg ? -1 : q.site_to_u[s];
        const int nedge_rows = edge_agg ? 1 : re->selection_len;
        for (int ei = 0; ei < nedge_rows; ei++) {
            long double v = 0;
            if (edge_agg) {
                for (int ue = 0; ue < E; ue++) {
                    if (w_edge[ue] == 0) continue;
                    const int ce = q.m.edge_order[ue];
                    long double x = site_agg ? ((long double)sums[2 * ce] + (long double)sums[2 * ce + 1]) / q.div_site
                                             : (long double)vals[(size_t)u * E + ce];
                    v += x * w_edge[ue] / div_edge;
                }
            } else {
                const int ce = q.m.edge_order[re->selection[ei]];
                v = site_agg ? ((long double)sums[2 * ce] + (long double)sums[2 * ce + 1]) / q.div_site
                             : (long double)vals[(size_t)u * E + ce];
            }
            double d = clean(v);
            if (check_finite(d, "a log likelihood derivative")) goto done;
            if (!first) jbuf_puts(out, ", ");
            first = 0;
            jbuf_puts(out, "[");
            if (!site_agg) { jbuf_int(out, s); jbuf_puts(out, ", "); }
            if (!edge_agg) { jbuf_int(out, re->selection[ei]); jbuf_puts(out, ", "); }
            jbuf_real(out, d);
            jbuf_puts(out, "]");
        }
    }
    jbuf_puts(out, "]}");
    rc = 0;
done:
    free(mask); free(vals); free(sums); free(w_edge);
    query_clear(&q);
    return rc;
}

/* ------------------------------------------------------------------ marginal */
static int run_marginal(const jval *root, jbuf *out)
{
    query q;
    int rc = -1;
    int *mask = NULL;
    double *vals = NULL, *sums = NULL;
    long double *w_node = NULL, *w_state = NULL;
    query_init(&q);
    if (query_parse(&q, 2, root)) goto done;
    if (query_prepare(&q)) goto done;
    const int N = q.m.N, k = q.m.k;
    const host_reduction *rn = &q.r_a, *rs = &q.r_b;
    mask = calloc(N + 1, sizeof(int));
    w_node = malloc((size_t)(N + 1) * sizeof(long double));
    w_state = malloc((size_t)(k + 1) * sizeof(long double));
    if (!mask || !w_node || !w_state) goto done;
    for (int i = 0; i < rn->selection_len; i++) mask[rn->selection[i]] = 1;
    long double div_node = 1, div_state = 1;
    const int site_agg = q.r_site.agg_mode != AGG_NONE, node_agg = rn->agg_mode != AGG_NONE, state_agg = rs->agg_mode != AGG_NONE;
    if (node_agg) host_reduction_weights(rn, w_node, &div_node);
    if (state_agg) host_reduction_weights(rs, w_state, &div_state);
    const int need = q.U > 0 && rn->selection_len > 0 && rs->selection_len > 0;
    if (site_agg) {
        sums = calloc((size_t)N * k * 2 + 2, sizeof(double));
        if (!sums) goto done;
        if (need && plk_group_marginal(q.eng, mask, NULL, sums)) { fprintf(stderr, "error: %s\n", plk_group_last_error(q.eng)); goto done; }
    } else {
        vals = calloc((size_t)q.U * N * k + 1, sizeof(double));
        if (!vals) goto done;
        if (need && plk_group_marginal(q.eng, mask, vals, NULL)) { fprintf(stderr, "error: %s\n", plk_group_last_error(q.eng)); goto done; }
    }
    const char *names[] = {"site", "node", "state"};
    const host_reduction *reds[] = {&q.r_site, rn, rs};
    table_begin(out, names, reds, 3);
    int first = 1;
    const int nsite_rows = site_agg ? 1 : q.r_site.selection_len;
    const int nnode_rows = node_agg ? 1 : rn->selection_len;
    const int nstate_rows = state_agg ? 1 : rs->selection_len;
    for (int si = 0; si < nsite_rows; si++) {
        const int s = site_agg ? -1 : q.r_site.selection[si];
        const long u = site_agg ? -1 : q.site_to_u[s];
        for (int ni = 0; ni < nnode_rows; ni++)
            for (int ti = 0; ti < nstate_rows; ti++) {
                long double v = 0;
                const int a_lo = node_agg ? 0 : rn->selection[ni], a_hi = node_agg ? N : a_lo + 1;
                const int j_lo = state_agg ? 0 : rs->selection[ti], j_hi = state_agg ? k : j_lo + 1;
                for (int a = a_lo; a < a_hi; a++) {
                    const long double wa = node_agg ? w_node[a] / div_node : 1;
                    if (node_agg && w_node[a] == 0) continue;
                    for (int j = j_lo; j < j_hi; j++) {
                        if (state_agg && w_state[j] == 0) continue;
                        const long double wj = state_agg ? w_state[j] / div_state : 1;
                        const size_t cell = (size_t)a * k + j;
                        long double x = site_agg ? ((long double)sums[2 * cell] + (long double)sums[2 * cell + 1]) / q.div_site
                                                 : (long double)vals[(size_t)u * N * k + cell];
                        v += x * wa * wj;
                    }
                }
                double d = clean(v);
                if (check_finite(d, "a marginal probability")) goto done;
                if (!first) jbuf_puts(out, ", ");
                first = 0;
                jbuf_puts(out, "[");
                if (!site_agg) { jbuf_int(out, s); jbuf_puts(out, ", "); }
                if (!node_agg) { jbuf_int(out, rn->selection[ni]); jbuf_puts(out, ", "); }
                if (!state_agg) { jbuf_int(out, rs->selection[ti]); jbuf_puts(out, ", "); }
                jbuf_real(out, d);
                jbuf_puts(out, "]");
            }
    }
    jbuf_puts(out, "]}");
    rc = 0;
done:
    free(mask); free(vals); free(sums); free(w_node); free(w_state);
    query_clear(&q);
    return rc;
}

/* ------------------------------------------------------------------ dwell / trans */
/* split a long double into an unevaluated (hi, lo) pair of doubles */
static void split_ld(long double v, double *hi, double *lo)
{
    *hi = (double)v;
    *lo = isfinite(*hi) ? (double)(v - (long double)*hi) : 0.0;
}

/*
 * arbplf-dwell (kind 3, src/arbplfdwell.c:303-505) and arbplf-trans (kind 4, src/arbplftrans.c:346-551).
 * The third axis (states / state pairs) is either aggregated -- then its weights go into the one
 * direction matrix L ("linear algebra trick", src/arbplfdwell.c:159-204, src/arbplftrans.c:162-224) --
 * or listed, one engine pass per selected state / pair (src/arbplfdwell.c:117-157, src/arbplftrans.c:116-160).
 */
static int run_edge_expect(const jval *root, jbuf *out, int kind)
{
    query q;
    int rc = -1;
    int *mask = NULL;
    double *vals_all = NULL, *sums_all = NULL, *Lall_hi = NULL, *Lall_lo = NULL, *Lhi = NULL, *Llo = NULL;
    long double *w_edge = NULL, *w_third = NULL, *Lacc = NULL;
    int npass = 0;
    query_init(&q);
    if (query_parse(&q, kind, root)) goto done;
    if (query_prepare(&q)) goto done;
    const int E = q.m.E, k = q.m.k;
    const host_reduction *re = &q.r_a, *rt = &q.r_b;
    const int site_agg = q.r_site.agg_mode != AGG_NONE, edge_agg = re->agg_mode != AGG_NONE, third_agg = rt->agg_mode != AGG_NONE;
    const int coef = kind == 3 ? PLK_COEF_PRIOR : PLK_COEF_PRIOR_RATE_EDGE;
    mask = calloc(E + 1, sizeof(int));
    w_edge = malloc((size_t)(E + 1) * sizeof(long double));
    w_third = malloc((size_t)(rt->n + 1) * sizeof(long double));
    Lhi = calloc((size_t)k * k + 1, sizeof(double));
    Llo = calloc((size_t)k * k + 1, sizeof(double));
    Lacc = calloc((size_t)k * k + 1, sizeof(long double));
    if (!mask || !w_edge || !w_third || !Lhi || !Llo || !Lacc) goto done;
    for (int i = 0; i < re->selection_len; i++) mask[q.m.edge_order[re->selection[i]]] = 1;
    long double div_edge = 1, div_third = 1;
    if (edge_agg) host_reduction_weights(re, w_edge, &div_edge);
    if (third_agg) host_reduction_weights(rt, w_third, &div_third);
    npass = third_agg ? 1 : rt->selection_len;
    Lall_hi = calloc((size_t)npass * k * k + 1, sizeof(double));
    Lall_lo = calloc((size_t)npass * k * k + 1, sizeof(double));
    if (!Lall_hi || !Lall_lo) goto done;
    for (int t = 0; t < npass; t++) {
        /* direction matrix; for trans its entries are multiplied by the normalised rates
         * (src/arbplftrans.c:137-138, :196), hi/lo parts of Qn included */
        for (int i = 0; i < k * k; i++) Lacc[i] = 0;
        if (third_agg) {
            for (int x = 0; x < rt->n; x++) {
                if (w_third[x] == 0) continue;
                const int a = kind == 3 ? x : q.pair_first[x], b = kind == 3 ? x : q.pair_second[x];
                Lacc[a * k + b] += w_third[x];
            }
        } else {
            const int x = rt->selection[t];
            const int a = kind == 3 ? x : q.pair_first[x], b = kind == 3 ? x : q.pair_second[x];
            Lacc[a * k + b] = 1;
        }
        if (q.U > 0)
            for (int i = 0; i < k * k; i++) {
                long double v = Lacc[i];
                if (kind == 4) v *= (long double)q.Qn[i] + (long double)q.Qn_lo[i];
                v /= div_third;
                split_ld(v, &Lhi[i], &Llo[i]);
            }
        memcpy(Lall_hi + (size_t)t * k * k, Lhi, (size_t)k * k * sizeof(double));
        memcpy(Lall_lo + (size_t)t * k * k, Llo, (size_t)k * k * sizeof(double));
    }
    /* one engine call for all directions: the k = 4 kernels carry up to four of them per pass */
    if (site_agg) sums_all = calloc((size_t)npass * E * 2 + 2, sizeof(double));
    else vals_all = calloc((size_t)q.U * npass * E + 1, sizeof(double));
    if (!sums_all && !vals_all) goto done;
    if (q.U > 0 && re->selection_len > 0 && npass > 0 &&
        plk_group_edge_expect_multi(q.eng, npass, Lall_hi, Lall_lo, coef, mask, vals_all, sums_all)) {
        fprintf(stderr, "error: %s\n", plk_group_last_error(q.eng)); goto done;
    }
#define EXP_SUM(t, ce) (((long double)sums_all[((size_t)(t) * E + (ce)) * 2] + (long double)sums_all[((size_t)(t) * E + (ce)) * 2 + 1]) / q.div_site)
#define EXP_VAL(t, u, ce) ((long double)vals_all[((size_t)(u) * npass + (t)) * E + (ce)])
    /* table header; the trans axis prints its two component indices (src/ndaccum.c:355-366, :401-409) */
    jbuf_puts(out, "{\"columns\": [");
    if (!site_agg) jbuf_puts(out, "\"site\", ");
    if (!edge_agg) jbuf_puts(out, "\"edge\", ");
    if (!third_agg) jbuf_puts(out, kind == 3 ? "\"state\", " : "\"first_state\", \"second_state\", ");
    jbuf_puts(out, "\"value\"], \"data\": [");
    int first = 1;
    const int nsite_rows = site_agg ? 1 : q.r_site.selection_len;
    const int nedge_rows = edge_agg ? 1 : re->selection_len;
    for (int si = 0; si < nsite_rows; si++) {
        const int s = site_agg ? -1 : q.r_site.selection[si];
        const long u = site_agg ? -1 : q.site_to_u[s];
        for (int ei = 0; ei < nedge_rows; ei++) {
            for (int t = 0; t < npass; t++) {
                long double v = 0;
                if (edge_agg) {
                    for (int ue = 0; ue < E; ue++) {
                        if (w_edge[ue] == 0) continue;
                        const int ce = q.m.edge_order[ue];
                        long double x = site_agg ? EXP_SUM(t, ce) : EXP_VAL(t, u, ce);
                        v += x * w_edge[ue] / div_edge;
                    }
                } else {
                    const int ce = q.m.edge_order[re->selection[ei]];
                    v = site_agg ? EXP_SUM(t, ce) : EXP_VAL(t, u, ce);
                }
                double d = clean(v);
                if (check_finite(d, kind == 3 ? "a dwell expectation" : "a transition count expectation")) goto done;
                if (!first) jbuf_puts(out, ", ");
                first = 0;
                jbuf_puts(out, "[");
                if (!site_agg) { jbuf_int(out, s); jbuf_puts(out, ", "); }
                if (!edge_agg) { jbuf_int(out, re->selection[ei]); jbuf_puts(out, ", "); }
                if (!third_agg) {
                    const int x = rt->selection[t];
                    if (kind == 3) { jbuf_int(out, x); jbuf_puts(out, ", "); }
                    else { jbuf_int(out, q.pair_first[x]); jbuf_puts(out, ", "); jbuf_int(out, q.pair_second[x]); jbuf_puts(out, ", "); }
                }
                jbuf_real(out, d);
                jbuf_puts(out, "]");
            }
        }
    }
    jbuf_puts(out, "]}");
    rc = 0;
done:
#undef EXP_SUM
#undef EXP_VAL
    free(vals_all); free(sums_all); free(Lall_hi); free(Lall_lo);
    free(mask); free(w_edge); free(w_third); free(Lhi); free(Llo); free(Lacc);
    query_clear(&q);
    return rc;
}

static int run_dwell(const jval *root, jbuf *out) { return run_edge_expect(root, out, 3); }
static int run_trans(const jval *root, jbuf *out) { return run_edge_expect(root, out, 4); }

/* ------------------------------------------------------------------ em-update */
/* arbplf-em-update (src/arbplfem.c:397-503): edge_rate_e * E[transitions on e] / E[rate-weighted dwell on e],
 * both accumulated over the weighted sites; exactly 0 where the expected transition count is 0. */
static int run_em_update(const jval *root, jbuf *out)
{
    query q;
    int rc = -1;
    double *Lhi = NULL, *Llo = NULL, *dw = NULL, *tr = NULL, *both = NULL;
    query_init(&q);
    if (query_parse(&q, 5, root)) goto done;
    if (query_prepare(&q)) goto done;
    const int E = q.m.E, k = q.m.k;
    Lhi = calloc((size_t)k * k * 2 + 1, sizeof(double));
    Llo = calloc((size_t)k * k * 2 + 1, sizeof(double));
    dw = calloc((size_t)E * 2 + 2, sizeof(double));
    tr = calloc((size_t)E * 2 + 2, sizeof(double));
    if (!Lhi || !Llo || !dw || !tr) goto done;
    if (q.U > 0 && E > 0) {
        /* L_dwell: exit rates on the diagonal; L_trans: rates off the diagonal (src/arbplfem.c:118-129) */
        for (int i = 0; i < k; i++)
            for (int j = 0; j < k; j++) {
                Lhi[i * k + j] = i == j ? -q.Qn[i * k + j] : 0.0;
                Llo[i * k + j] = i == j ? -q.Qn_lo[i * k + j] : 0.0;
            }
        for (int i = 0; i < k; i++)
            for (int j = 0; j < k; j++) {
                Lhi[k * k + i * k + j] = i != j ? q.Qn[i * k + j] : 0.0;
                Llo[k * k + i * k + j] = i != j ? q.Qn_lo[i * k + j] : 0.0;
            }
        /* both expectations in one call: [2][E][2] sums (dwell first) */
        both = calloc((size_t)E * 4 + 4, sizeof(double));
        if (!both) goto done;
        if (plk_group_edge_expect_multi(q.eng, 2, Lhi, Llo, PLK_COEF_PRIOR_RATE, NULL, NULL, both)) { fprintf(stderr, "error: %s\n", plk_group_last_error(q.eng)); goto done; }
        memcpy(dw, both, (size_t)E * 2 * sizeof(double));
        memcpy(tr, both + (size_t)E * 2, (size_t)E * 2 * sizeof(double));
    }
    jbuf_puts(out, "{\"columns\": [\"edge\", \"value\"], \"data\": [");
    for (int ue = 0; ue < E; ue++) {
        const int ce = q.m.edge_order[ue];
        const long double t = (long double)tr[2 * ce] + (long double)tr[2 * ce + 1];
        const long double d = (long double)dw[2 * ce] + (long double)dw[2 * ce + 1];
        double v = 0.0;
        if (t != 0) v = clean(t / d * (long double)q.m.edge_rates_csr[ce]);
        if (check_finite(v, "an updated edge rate coefficient")) goto done;
        if (ue) jbuf_puts(out, ", ");
        jbuf_puts(out, "["); jbuf_int(out, ue); jbuf_puts(out, ", "); jbuf_real(out, v); jbuf_puts(out, "]");
    }
    jbuf_puts(out, "]}");
    rc = 0;
done:
    free(Lhi); free(Llo); free(dw); free(tr); free(both);
    query_clear(&q);
    return rc;
}

/* ------------------------------------------------------------------ hess */
/* arbplf-hess (src/arbplfhess.c:1163-1206 _parse_second_order, :1279-1343 hess_query): the E x E matrix of
 * second derivatives of the site-aggregated log likelihood, all user edges in order on both axes */
static int run_hess(const jval *root, jbuf *out)
{
    query q;
    int rc = -1;
    double *hs = NULL;
    query_init(&q);
    if (query_parse(&q, 6, root)) goto done;
    if (query_prepare(&q)) goto done;
    const int E = q.m.E;
    hs = calloc((size_t)E * E * 2 + 2, sizeof(double));
    if (!hs) goto done;
    if (q.U > 0 && E > 0 && plk_group_hess(q.eng, hs)) { fprintf(stderr, "error: %s\n", plk_group_last_error(q.eng)); goto done; }
    jbuf_puts(out, "{\"columns\": [\"first_edge\", \"second_edge\", \"value\"], \"data\": [");
    for (int a = 0; a < E; a++)
        for (int b = 0; b < E; b++) {
            const size_t pos = (size_t)q.m.edge_order[a] * E + q.m.edge_order[b];
            double v = clean(((long double)hs[2 * pos] + (long double)hs[2 * pos + 1]) / q.div_site);
            if (check_finite(v, "a second derivative of the log likelihood")) goto done;
            if (a || b) jbuf_puts(out, ", ");
            jbuf_puts(out, "["); jbuf_int(out, a); jbuf_puts(out, ", "); jbuf_int(out, b); jbuf_puts(out, ", ");
            jbuf_real(out, v); jbuf_puts(out, "]");
        }
    jbuf_puts(out, "]}");
    rc = 0;
done:
    free(hs);
    query_clear(&q);
    return rc;
}

/* ------------------------------------------------------------------ string API */
static char *string_hom(int (*run)(const jval *, jbuf *), void *userdata, const char *s_in, int *retcode)
{
    char err[256];
    char *s_out = NULL;
    int rc = -1;
    if (retcode) *retcode = -1;
    if (userdata) { fprintf(stderr, "internal error: unexpected userdata\n"); return NULL; }
    if (!s_in) { fprintf(stderr, "error: null input\n"); return NULL; }
    json_doc *doc = json_doc_parse(s_in, err, sizeof err);
    if (!doc) { fprintf(stderr, "%s\n", err); return NULL; }
    jbuf b;
    jbuf_init(&b);
    pthread_mutex_lock(&g_mu);
    rc = run(json_doc_root(doc), &b);
    pthread_mutex_unlock(&g_mu);
    json_doc_free(doc);
    if (rc == 0) {
        s_out = jbuf_take(&b);
        if (!s_out) { fprintf(stderr, "error: failed to dump the json object to a string\n"); rc = -1; }
    } else {
        free(jbuf_take(&b));
    }
    if (retcode) *retcode = rc;
    return s_out;
}

char *arbplf_ll_string(void *userdata, const char *s_in, int *retcode) { return string_hom(run_ll, userdata, s_in, retcode); }
char *arbplf_deriv_string(void *userdata, const char *s_in, int *retcode) { return string_hom(run_deriv, userdata, s_in, retcode); }
char *arbplf_marginal_string(void *userdata, const char *s_in, int *retcode) { return string_hom(run_marginal, userdata, s_in, retcode); }
char *arbplf_dwell_string(void *userdata, const char *s_in, int *retcode) { return string_hom(run_dwell, userdata, s_in, retcode); }
char *arbplf_trans_string(void *userdata, const char *s_in, int *retcode) { return string_hom(run_trans, userdata, s_in, retcode); }
char *arbplf_em_update_string(void *userdata, const char *s_in, int *retcode) { return string_hom(run_em_update, userdata, s_in, retcode); }
char *arbplf_hess_string(void *userdata, const char *s_in, int *retcode) { return string_hom(run_hess, userdata, s_in, retcode); }

/* Host-only validation (JSON grammar, model, reductions); no GPU is touched.
 * what: "ll", "deriv", "marginal", "dwell", "trans", "em_update" or "hess".  Returns 0 when the input would be accepted. */
int arbplf_validate_string(const char *what, const char *s_in)
{
    char err[256];
    int kind = !strcmp(what, "ll") ? 0 : !strcmp(what, "deriv") ? 1 : !strcmp(what, "marginal") ? 2 :
               !strcmp(what, "dwell") ? 3 : !strcmp(what, "trans") ? 4 : !strcmp(what, "em_update") ? 5 : !strcmp(what, "hess") ? 6 : -1;
    if (kind < 0 || !s_in) return -1;
    json_doc *doc = json_doc_parse(s_in, err, sizeof err);
    if (!doc) { fprintf(stderr, "%s\n", err); return -1; }
    query q;
    query_init(&q);
    int rc = query_parse(&q, kind, json_doc_root(doc));
    query_clear(&q);
    json_doc_free(doc);
    return rc;
}

/* src/runjson.c:88-147: read all of stdin, run, print one line */
int arbplf_run_stdin(char *(*f)(void *, const char *, int *))
{
    size_t cap = 1 << 16, n = 0;
    char *buf = malloc(cap);
    if (!buf) { fprintf(stderr, "failed to read string from stdin\n"); return -1; }
    while (1) {
        if (n + 4096 > cap) {
            cap *= 2;
            char *nb = realloc(buf, cap);
            if (!nb) { free(buf); fprintf(stderr, "failed to read string from stdin\n"); return -1; }
            buf = nb;
        }
        size_t got = fread(buf + n, 1, cap - n - 1, stdin);
        n += got;
        if (got == 0) break;
    }
    buf[n] = 0;
    int retcode = 0;
    char *s_out = f(NULL, buf, &retcode);
    free(buf);
    if (s_out) { puts(s_out); free(s_out); }
    arbplf_shutdown();
    return retcode;
}
