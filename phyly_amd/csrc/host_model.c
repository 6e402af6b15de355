/*
 * host_model.c -- JSON -> host_model validation (host C, product code).
 *
 * Reproduces which inputs the reference accepts and rejects
 * (src/parsemodel.c, src/parsereduction.c, src/csr_graph.c); the diagnostics on
 * stderr are this build's own wording, keyed by the JSON field at fault (stderr
 * text is not part of the drop-in contract: only the exit status is).  Behaviours the reference leaves unpinned
 *  are rejected: non-positive gamma_categories / gamma_shape,
 * invariable_prior outside [0, 1), a rate_mixture prior that is neither a
 * string nor an array, an empty alignment.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "host_model.h"

static int exists(const jval *v) { return v && !j_is_null(v); }

#define FAILF(...) do { fprintf(stderr, __VA_ARGS__); return -1; } while (0)

int host_check_keys(const jval *obj, const char *const *required, const char *const *allowed, const char *what)
{
    if (!j_is_object(obj)) FAILF("error: %s: expected an object\n", what);
    for (size_t i = 0; i < j_len(obj); i++) {
        const char *key = obj->u.c.keys[i];
        int ok = 0;
        for (const char *const *a = allowed; *a; a++) if (!strcmp(*a, key)) ok = 1;
        if (!ok) FAILF("error: %s: unexpected key \"%s\"\n", what, key);
    }
    for (const char *const *r = required; *r; r++)
        if (!j_get(obj, *r)) FAILF("error: %s: missing key \"%s\"\n", what, *r);
    return 0;
}

/* src/parsemodel.c:26-75 */
static int nonneg_array(double *dest, int n, const jval *a, const char *what)
{
    if (!j_is_array(a)) FAILF("error: %s must be an array of numbers\n", what);
    if ((int)j_len(a) != n) FAILF("error: %s has %d entries where %d are needed\n", what, (int)j_len(a), n);
    for (int i = 0; i < n; i++) {
        const jval *x = j_at(a, i);
        if (!j_is_number(x)) FAILF("error: %s holds something that is not a number\n", what);
        double d = j_number(x);
        if (d < 0) FAILF("error: %s holds a negative number\n", what);
        dest[i] = d;
    }
    return 0;
}

void host_model_init(host_model *m) { memset(m, 0, sizeof(*m)); m->rate_divisor = 1.0; m->mix.mode = K0_MIX_NONE; m->mix.n = 1; }

void host_model_clear(host_model *m)
{
    free(m->indptr); free(m->indices); free(m->preorder); free(m->edge_order); free(m->csr_to_user);
    free(m->edge_rates_user); free(m->edge_rates_csr); free(m->rate_matrix); free(m->root_custom);
    free(m->mix_rates); free(m->mix_prior); free(m->prob); free(m->codes8); free(m->defs);
    memset(m, 0, sizeof(*m));
}

/* src/parsemodel.c:211-368 + src/csr_graph.c:103-229 */
static int parse_edges(host_model *m, const jval *edges)
{
    if (!j_is_array(edges)) FAILF("error: edges: a list of [parent, child] pairs is required\n");
    const int E = (int)j_len(edges), N = E + 1;
    /* An empty edge list is rejected by the reference as well: node_count = edge_count + 1 = 1 (src/parsemodel.c:230-231)
     * and node 0 then fails "is not an endpoint of any edge" (src/parsemodel.c:306-318), so there is no single-node
     * tree to evaluate.  Same outcome here, said up front. */
    if (E == 0) FAILF("error: edges: the list is empty: node 0 is not an endpoint of any edge\n");
    int rc = -1;
    int *indeg = calloc(N, sizeof(int)), *outdeg = calloc(N, sizeof(int));
    int *pa = malloc((E + 1) * sizeof(int)), *pb = malloc((E + 1) * sizeof(int));
    int *fill = NULL, *visited = NULL;
    if (!indeg || !outdeg || !pa || !pb) goto done;
    for (int i = 0; i < E; i++) {
        const jval *e = j_at(edges, i);
        if (!j_is_array(e) || j_len(e) != 2 || !j_is_int(j_at(e, 0)) || !j_is_int(j_at(e, 1))) {
            fprintf(stderr, "error: edges: every entry must be a pair [parent, child] of integers\n"); goto done;
        }
        long long a = j_at(e, 0)->u.i, b = j_at(e, 1)->u.i;
        if (a < 0 || a >= N || b < 0 || b >= N) {
            fprintf(stderr, "error: edges: with E edges the node ids are 0..E\n"); goto done;
        }
        if (a == b) { fprintf(stderr, "error: edges: an edge joins a node to itself\n"); goto done; }
        pa[i] = (int)a; pb[i] = (int)b;
        outdeg[a]++; indeg[b]++;
    }
    int root = -1, roots = 0;
    for (int i = 0; i < N; i++) if (!indeg[i]) { root = i; roots++; }
    if (roots != 1) { fprintf(stderr, "error: edges: the tree needs exactly one root (a node that is nobody's child)\n"); goto done; }
    for (int i = 0; i < N; i++) if (indeg[i] > 1) { fprintf(stderr, "error: edges: a node has two parents\n"); goto done; }
    for (int i = 0; i < N; i++) if (indeg[i] + outdeg[i] < 1) { fprintf(stderr, "error: edges: node %d does not occur in any edge\n", i); goto done; }
    m->N = N; m->E = E; m->root = root;
    m->indptr = malloc((N + 1) * sizeof(int));
    m->indices = malloc((E + 1) * sizeof(int));
    m->preorder = malloc(N * sizeof(int));
    m->edge_order = malloc((E + 1) * sizeof(int));
    m->csr_to_user = malloc((E + 1) * sizeof(int));
    fill = calloc(N, sizeof(int));
    visited = calloc(N, sizeof(int));
    if (!m->indptr || !m->indices || !m->preorder || !m->edge_order || !m->csr_to_user || !fill || !visited) goto done;
    m->indptr[0] = 0;
    for (int i = 0; i < N; i++) m->indptr[i + 1] = m->indptr[i] + outdeg[i];
    for (int i = 0; i < E; i++) {
        int pos = m->indptr[pa[i]] + fill[pa[i]]++;
        m->indices[pos] = pb[i];
        m->edge_order[i] = pos;
        m->csr_to_user[pos] = i;
    }
    /* BFS by levels from the root; must reach every node exactly once */
    int npre = 0, head = 0;
    m->preorder[npre++] = root; visited[root] = 1;
    while (head < npre) {
        int a = m->preorder[head++];
        for (int j = m->indptr[a]; j < m->indptr[a + 1]; j++) {
            int b = m->indices[j];
            if (visited[b]) { fprintf(stderr, "error: edges: node %d is reached twice from the root\n", b); goto done; }
            visited[b] = 1;
            m->preorder[npre++] = b;
        }
    }
    if (npre != N) { fprintf(stderr, "error: edges: only %d of the %d nodes hang below the root\n", npre, N); goto done; }
    rc = 0;
done:
    free(indeg); free(outdeg); free(pa); free(pb); free(fill); free(visited);
    return rc;
}

static int parse_rate_matrix(host_model *m, const jval *rm)
{
    if (!j_is_array(rm)) FAILF("error: rate_matrix: a square list of rows is required\n");
    const int k = (int)j_len(rm);
    m->k = k;
    m->rate_matrix = malloc((size_t)(k ? k : 1) * (k ? k : 1) * sizeof(double));
    if (!m->rate_matrix) return -1;
    for (int i = 0; i < k; i++) {
        const jval *row = j_at(rm, i);
        if (!j_is_array(row)) FAILF("error: rate_matrix: a row is not a list\n");
        if ((int)j_len(row) != k) FAILF("error: rate_matrix: a row has %d entries, the matrix has %d rows\n", (int)j_len(row), k);
        for (int j = 0; j < k; j++) {
            const jval *y = j_at(row, j);
            if (!j_is_number(y)) FAILF("error: rate_matrix: an entry is not a number\n");
            double d = j_number(y);
            if (d < 0) FAILF("error: rate_matrix: negative rate (the diagonal is ignored but must not be negative either)\n");
            m->rate_matrix[(size_t)i * k + j] = d;
        }
    }
    if (k < 1) FAILF("error: rate_matrix: at least one state is required\n");
    return 0;
}

static int parse_probability_array(host_model *m, const jval *pa)
{
    const char *name = "probability_array";
    if (!j_is_array(pa)) FAILF("error: %s: a list per site, each with one row per node, is required\n", name);
    const long S = (long)j_len(pa);
    const int N = m->N, k = m->k;
    m->S = S;
    m->prob = malloc(((size_t)S * N * k + 1) * sizeof(double));
    if (!m->prob) return -1;
    for (long s = 0; s < S; s++) {
        const jval *x = j_at(pa, s);
        if (!j_is_array(x)) FAILF("error: %s: a list per site, each with one row per node, is required\n", name);
        if ((int)j_len(x) != N) FAILF("error: %s: a site lists %d nodes, the tree has %d\n", name, (int)j_len(x), N);
        for (int a = 0; a < N; a++)
            if (nonneg_array(m->prob + ((size_t)s * N + a) * k, k, j_at(x, a), name)) return -1;
    }
    /* Compact form ("character_data fast path"): real probability arrays consist of a handful of
     * distinct rows (one-hot states, all-ones "missing", a few ambiguity sets).  When there are at most 256 of them,
     * the array is re-expressed as character codes + definitions -- bitwise the same observation vectors -- so that
     * the compact device layout (1 byte per node and site), the tip tables and the fused kernels apply.  Rows are
     * compared bit for bit (0.0 and -0.0 are different rows with the same meaning).  On by default;
     * ARBPLF_COMPACT_DENSE=0 keeps the dense layout (and the dense generic kernels) for comparison. */
    const char *compact_env = getenv("ARBPLF_COMPACT_DENSE");
    if (S * (long)N > 0 && !(compact_env && compact_env[0] == '0')) {
        const size_t rows = (size_t)S * N, rb = (size_t)k * sizeof(double);
        uint8_t *codes = malloc(rows + 1);
        double *defs = malloc(256 * rb + 1);
        int table[1024];
        int nchar = 0, ok = codes && defs;
        for (int i = 0; i < 1024; i++) table[i] = -1;
        for (size_t r = 0; ok && r < rows; r++) {
            const unsigned char *row = (const unsigned char *)(m->prob + r * k);
            unsigned long long hsh = 1469598103934665603ULL;
            for (size_t b = 0; b < rb; b++) { hsh ^= row[b]; hsh *= 1099511628211ULL; }
            size_t pos = (size_t)hsh & 1023;
            int found = -1;
            while (table[pos] >= 0) {
                if (!memcmp(defs + (size_t)table[pos] * k, row, rb)) { found = table[pos]; break; }
                pos = (pos + 1) & 1023;
            }
            if (found < 0) {
                if (nchar == 256) { ok = 0; break; }
                memcpy(defs + (size_t)nchar * k, row, rb);
                table[pos] = found = nchar++;
            }
            codes[r] = (uint8_t)found;
        }
        if (ok) {
            free(m->prob); m->prob = NULL;
            m->codes8 = codes; m->defs = defs; m->nchar = nchar;
        } else { free(codes); free(defs); }
    }
    return 0;
}

/* src/parsemodel.c:515-628 */
static int parse_character_data(host_model *m, const jval *cd, const jval *defs)
{
    const char *name = "character_data";
    if (!exists(defs)) fprintf(stderr, "error: %s needs 'character_definitions' next to it\n", name);
    if (!j_is_array(cd)) FAILF("error: %s: a list per site of one character index per node is required\n", name);
    if (!j_is_array(defs)) FAILF("error: %s: 'character_definitions' must be a list of rows, one per character\n", name);
    const long S = (long)j_len(cd);
    const int N = m->N, k = m->k, nchar = (int)j_len(defs);
    m->S = S; m->nchar = nchar;
    m->defs = malloc(((size_t)nchar * k + 1) * sizeof(double));
    if (!m->defs) return -1;
    for (int c = 0; c < nchar; c++)
        if (nonneg_array(m->defs + (size_t)c * k, k, j_at(defs, c), name)) return -1;
    const int compact = nchar <= 256;
    if (compact) m->codes8 = malloc((size_t)S * N + 1);
    else m->prob = malloc(((size_t)S * N * k + 1) * sizeof(double));
    if (!m->codes8 && !m->prob) return -1;
    for (long s = 0; s < S; s++) {
        const jval *x = j_at(cd, s);
        if (!j_is_array(x)) FAILF("error: %s: a list per site, each with one row per node, is required\n", name);
        if ((int)j_len(x) != N) FAILF("error: %s: a site lists %d nodes, the tree has %d\n", name, (int)j_len(x), N);
        for (int a = 0; a < N; a++) {
            const jval *y = j_at(x, a);
            if (!j_is_int(y)) FAILF("error: %s: a character index is not an integer\n", name);
            long long c = y->u.i;
            if (c < 0) FAILF("error: %s: negative character index\n", name);
            if (c >= nchar) FAILF("error: %s: character index beyond the %d definitions\n", name, nchar);
            if (compact) m->codes8[(size_t)s * N + a] = (uint8_t)c;
            else memcpy(m->prob + ((size_t)s * N + a) * k, m->defs + (size_t)c * k, k * sizeof(double));
        }
    }
    if (!compact) { free(m->defs); m->defs = NULL; m->nchar = 0; }
    return 0;
}

/*
 * Binary side channel for alignments that JSON cannot carry (an extension, the reference has no
 * counterpart): "character_data_file" names a raw file of S * N bytes, the same [site][node] character indices that
 * "character_data" would list, used together with "character_definitions" (at most 256 of them).
 */
static int parse_character_data_file(host_model *m, const jval *path, const jval *defs)
{
    const char *name = "character_data_file";
    if (!j_is_string(path)) FAILF("%s: expected the path of a raw byte file\n", name);
    if (!j_is_array(defs)) FAILF("error: %s: 'character_definitions' must be a list of rows, one per character\n", name);
    const int N = m->N, k = m->k, nchar = (int)j_len(defs);
    if (nchar < 1 || nchar > 256) FAILF("%s: between 1 and 256 character definitions are required\n", name);
    m->nchar = nchar;
    m->defs = malloc(((size_t)nchar * k + 1) * sizeof(double));
    if (!m->defs) return -1;
    for (int c = 0; c < nchar; c++)
        if (nonneg_array(m->defs + (size_t)c * k, k, j_at(defs, c), name)) return -1;
    FILE *f = fopen(path->u.s, "rb");
    if (!f) FAILF("%s: cannot open '%s'\n", name, path->u.s);
    if (fseek(f, 0, SEEK_END)) { fclose(f); FAILF("%s: cannot seek in '%s'\n", name, path->u.s); }
    const long bytes = ftell(f);
    rewind(f);
    if (bytes < 0 || bytes % N) { fclose(f); FAILF("%s: the size of '%s' (%ld bytes) is not a multiple of the node count (%d)\n", name, path->u.s, bytes, N); }
    m->S = bytes / N;
    m->codes8 = malloc((size_t)bytes + 1);
    if (!m->codes8) { fclose(f); return -1; }
    const size_t got = fread(m->codes8, 1, (size_t)bytes, f);
    fclose(f);
    if (got != (size_t)bytes) FAILF("%s: short read from '%s'\n", name, path->u.s);
    for (long i = 0; i < bytes; i++)
        if (m->codes8[i] >= nchar) FAILF("error: %s: character index beyond the %d definitions\n", name, nchar);
    return 0;
}

static int parse_rate_divisor(host_model *m, const jval *rd)
{
    const char *msg = "error: rate_divisor: give a number greater than zero or \"equilibrium_exit_rate\"\n";
    if (!exists(rd)) return 0;
    if (j_is_string(rd)) {
        if (strcmp(rd->u.s, "equilibrium_exit_rate")) FAILF("%s", msg);
        m->use_equilibrium_divisor = 1;
    } else if (j_is_number(rd)) {
        double d = j_number(rd);
        if (!(d > 0)) FAILF("%s", msg);
        m->rate_divisor = d;
    } else FAILF("%s", msg);
    return 0;
}

static int parse_root_prior(host_model *m, const jval *rp)
{
    const char *msg = "error: root_prior: give one weight per state, \"equilibrium_distribution\" or \"uniform_distribution\"\n";
    if (!exists(rp)) { m->root_mode = HM_ROOT_NONE; return 0; }
    if (j_is_string(rp)) {
        if (!strcmp(rp->u.s, "equilibrium_distribution")) m->root_mode = HM_ROOT_EQUILIBRIUM;
        else if (!strcmp(rp->u.s, "uniform_distribution")) m->root_mode = HM_ROOT_UNIFORM;
        else FAILF("%s", msg);
        return 0;
    }
    m->root_mode = HM_ROOT_CUSTOM;
    m->root_custom = malloc((size_t)m->k * sizeof(double));
    if (!m->root_custom) return -1;
    if (nonneg_array(m->root_custom, m->k, rp, "root_prior")) FAILF("%s", msg);
    return 0;
}

/* src/parsemodel.c:632-713 */
static int parse_rate_mixture(host_model *m, const jval *rm)
{
    static const char *const req[] = {"rates", "prior", NULL};
    if (host_check_keys(rm, req, req, "rate_mixture")) return -1;
    const jval *rates = j_get(rm, "rates"), *prior = j_get(rm, "prior");
    if (!j_is_array(rates)) FAILF("error: rate_mixture: 'rates' must be a list of numbers\n");
    const int n = (int)j_len(rates);
    if (n < 1) FAILF("error: rate_mixture: at least one category is required\n");
    m->mix_rates = malloc(n * sizeof(double));
    m->mix_prior = calloc(n, sizeof(double));
    if (!m->mix_rates || !m->mix_prior) return -1;
    if (nonneg_array(m->mix_rates, n, rates, "rate_mixture")) FAILF("error: rate_mixture: 'rates' is not usable\n");
    if (j_is_string(prior)) {
        if (strcmp(prior->u.s, "uniform_distribution"))
            FAILF("error: rate_mixture: 'prior' is one weight per category or \"uniform_distribution\"\n");
        m->mix.mode = K0_MIX_UNIFORM;
    } else if (j_is_array(prior)) {
        if (nonneg_array(m->mix_prior, n, prior, "rate_mixture")) FAILF("error: rate_mixture: 'prior' is not usable\n");
        m->mix.mode = K0_MIX_CUSTOM;
    } else FAILF("error: rate_mixture: 'prior' is one weight per category or \"uniform_distribution\"\n");
    m->mix.n = n; m->mix.rates = m->mix_rates; m->mix.prior = m->mix_prior;
    return 0;
}

/* src/parsemodel.c:716-783 */
static int parse_gamma_mixture(host_model *m, const jval *g, int mode)
{
    static const char *const req[] = {"gamma_shape", "gamma_categories", NULL};
    static const char *const all[] = {"gamma_shape", "gamma_categories", "invariable_prior", NULL};
    if (host_check_keys(g, req, all, "gamma rate mixture")) return -1;
    const jval *ip = j_get(g, "invariable_prior"), *gs = j_get(g, "gamma_shape"), *gc = j_get(g, "gamma_categories");
    m->mix.mode = mode;
    m->mix.invariable_prior = 0;
    if (exists(ip)) {
        if (!j_is_number(ip)) FAILF("error: invariable_prior must be a number\n");
        m->mix.invariable_prior = j_number(ip);
    }
    if (!j_is_number(gs)) FAILF("error: gamma_shape must be a number\n");
    m->mix.gamma_shape = j_number(gs);
    if (!j_is_int(gc)) FAILF("error: gamma_categories must be an integer\n");
    if (gc->u.i < 1 || gc->u.i > 1000000) FAILF("error: gamma_categories must be at least 1\n");
    m->mix.n = (int)gc->u.i;
    if (!(m->mix.gamma_shape > 0)) FAILF("error: gamma_shape must be greater than zero\n");
    if (!(m->mix.invariable_prior >= 0 && m->mix.invariable_prior < 1)) FAILF("error: invariable_prior must lie in [0, 1)\n");
    return 0;
}

/* src/parsemodel.c:786-913 */
int host_model_parse(host_model *m, const jval *root)
{
    static const char *const req[] = {"edges", "edge_rate_coefficients", "rate_matrix", NULL};
    static const char *const all[] = {"edges", "edge_rate_coefficients", "rate_matrix", "probability_array",
        "character_definitions", "character_data", "character_data_file", "rate_divisor", "root_prior", "rate_mixture",
        "gamma_rate_mixture", "normalized_median_gamma_rate_mixture", NULL};
    if (host_check_keys(root, req, all, "model_and_data")) return -1;
    const jval *pa = j_get(root, "probability_array"), *cdefs = j_get(root, "character_definitions");
    const jval *cdata = j_get(root, "character_data"), *rmix = j_get(root, "rate_mixture");
    const jval *cfile = j_get(root, "character_data_file");
    const jval *gmix = j_get(root, "gamma_rate_mixture"), *gmed = j_get(root, "normalized_median_gamma_rate_mixture");
    if (exists(rmix) + exists(gmix) + exists(gmed) > 1) FAILF("error: more than one of rate_mixture / gamma_rate_mixture / normalized_median_gamma_rate_mixture given\n");
    if (exists(pa) && exists(cdata)) FAILF("error: give either 'probability_array' or 'character_data', not both\n");
    if (exists(pa) && exists(cdefs)) FAILF("error: 'character_definitions' belongs with 'character_data', not with 'probability_array'\n");
    if (exists(cfile) && (exists(pa) || exists(cdata))) FAILF("error: 'character_data_file' excludes 'probability_array' and 'character_data'\n");

    if (parse_edges(m, j_get(root, "edges"))) return -1;
    m->edge_rates_user = malloc((size_t)(m->E + 1) * sizeof(double));
    m->edge_rates_csr = malloc((size_t)(m->E + 1) * sizeof(double));
    if (!m->edge_rates_user || !m->edge_rates_csr) return -1;
    if (nonneg_array(m->edge_rates_user, m->E, j_get(root, "edge_rate_coefficients"), "edge_rate_coefficients")) return -1;
    for (int i = 0; i < m->E; i++) m->edge_rates_csr[m->edge_order[i]] = m->edge_rates_user[i];
    if (parse_rate_matrix(m, j_get(root, "rate_matrix"))) return -1;
    if (exists(pa)) { if (parse_probability_array(m, pa)) return -1; }
    else if (exists(cdata)) { if (parse_character_data(m, cdata, cdefs)) return -1; }
    else if (exists(cfile)) { if (parse_character_data_file(m, cfile, cdefs)) return -1; }
    else FAILF("error: no observations: 'probability_array' or 'character_data' is required\n");
    if (parse_rate_divisor(m, j_get(root, "rate_divisor"))) return -1;
    if (parse_root_prior(m, j_get(root, "root_prior"))) return -1;
    if (exists(gmix)) { if (parse_gamma_mixture(m, gmix, K0_MIX_GAMMA)) return -1; }
    else if (exists(gmed)) { if (parse_gamma_mixture(m, gmed, K0_MIX_GAMMA_MEDIAN)) return -1; }
    else if (exists(rmix)) { if (parse_rate_mixture(m, rmix)) return -1; }
    else { m->mix.mode = K0_MIX_NONE; m->mix.n = 1; }
    return 0;
}

/* ---------------------------------------------------------------- reductions */
void host_reduction_init(host_reduction *r) { memset(r, 0, sizeof(*r)); }
void host_reduction_clear(host_reduction *r) { free(r->selection); free(r->weights); memset(r, 0, sizeof(*r)); }

/* src/parsereduction.c:20-195 */
/* src/parsereduction.c:84-159 _validate_column_aggregation */
static int reduction_parse_aggregation(host_reduction *r, const char *name, const jval *agg)
{
    if (!agg) r->agg_mode = AGG_NONE;
    else if (j_is_string(agg)) {
        if (!strcmp(agg->u.s, "sum")) r->agg_mode = AGG_SUM;
        else if (!strcmp(agg->u.s, "avg")) r->agg_mode = AGG_AVG;
        else if (!strcmp(agg->u.s, "only")) {
            if (r->selection_len != 1) FAILF("error: %s_reduction: \"only\" needs a selection of exactly one index, not %d\n", name, r->selection_len);
            r->agg_mode = AGG_ONLY;
        } else FAILF("error: %s_reduction: aggregation is \"sum\", \"avg\", \"only\" or a list of weights\n", name);
    } else if (j_is_array(agg)) {
        r->agg_mode = AGG_WEIGHTED_SUM;
        if ((int)j_len(agg) != r->selection_len) FAILF("error: %s_reduction: one weight per selected %s is required\n", name, name);
        r->weights = malloc((size_t)(r->selection_len + 1) * sizeof(double));
        if (!r->weights) return -1;
        for (int i = 0; i < r->selection_len; i++) {
            const jval *x = j_at(agg, i);
            if (!j_is_number(x)) FAILF("error: %s_reduction: a weight is not a number\n", name);
            r->weights[i] = j_number(x);
        }
    } else FAILF("error: %s_reduction: aggregation is neither a string nor a list of weights\n", name);
    if (r->agg_mode == AGG_AVG && r->selection_len == 0) FAILF("error: %s_reduction: \"avg\" over nothing\n", name);
    return 0;
}

int host_reduction_parse(host_reduction *r, int n, const char *name, const jval *root)
{
    static const char *const none[] = {NULL};
    static const char *const all[] = {"selection", "aggregation", NULL};
    const jval *sel = NULL, *agg = NULL;
    r->n = n;
    if (root) {
        if (host_check_keys(root, none, all, "reduction")) return -1;
        sel = j_get(root, "selection");
        agg = j_get(root, "aggregation");
    }
    if (!sel) {
        r->selection_len = n;
        r->selection = malloc((size_t)(n + 1) * sizeof(int));
        if (!r->selection) return -1;
        for (int i = 0; i < n; i++) r->selection[i] = i;
    } else {
        if (!j_is_array(sel)) FAILF("error: %s_reduction: selection must be a list\n", name);
        r->selection_len = (int)j_len(sel);
        r->selection = malloc((size_t)(r->selection_len + 1) * sizeof(int));
        if (!r->selection) return -1;
        for (int i = 0; i < r->selection_len; i++) {
            const jval *x = j_at(sel, i);
            if (!j_is_int(x)) FAILF("error: %s_reduction: a selected index is not an integer\n", name);
            if (x->u.i < 0) FAILF("error: %s_reduction: negative index in the selection\n", name);
            if (x->u.i >= n) FAILF("error: %s_reduction: selected index beyond the last %s\n", name, name);
            r->selection[i] = (int)x->u.i;
        }
    }
    return reduction_parse_aggregation(r, name, agg);
}

/* src/parsereduction.c:205-392 validate_column_pair_reduction: a selection of [first, second] state
 * pairs; without a selection every ordered pair of distinct states is selected and only
 * "sum" / "avg" (or no aggregation) are allowed.  r->selection[i] = i; *first / *second are
 * malloc'd arrays of r->selection_len entries. */
int host_pair_reduction_parse(host_reduction *r, int **first, int **second, int k, const char *name, const jval *root)
{
    static const char *const none[] = {NULL};
    static const char *const all[] = {"selection", "aggregation", NULL};
    const jval *sel = NULL, *agg = NULL;
    *first = *second = NULL;
    if (root) {
        if (host_check_keys(root, none, all, "reduction")) return -1;
        sel = j_get(root, "selection");
        agg = j_get(root, "aggregation");
        if (j_is_null(sel)) sel = NULL;      /* _exists(): an explicit null counts as absent (:13-16) */
    }
    if (sel) {
        if (!j_is_array(sel)) FAILF("error: %s_reduction: selection must be a list\n", name);
        const int n = (int)j_len(sel);
        r->n = n;
        r->selection_len = n;
        r->selection = malloc((size_t)(n + 1) * sizeof(int));
        *first = malloc((size_t)(n + 1) * sizeof(int));
        *second = malloc((size_t)(n + 1) * sizeof(int));
        if (!r->selection || !*first || !*second) return -1;
        for (int i = 0; i < n; i++) {
            const jval *p = j_at(sel, i);
            if (!j_is_array(p) || j_len(p) != 2 || !j_is_int(j_at(p, 0)) || !j_is_int(j_at(p, 1)))
                FAILF("error: %s_reduction: every selected entry is a pair [from, to] of integers\n", name);
            const long a = j_at(p, 0)->u.i, b = j_at(p, 1)->u.i;
            if (a < 0 || a >= k || b < 0 || b >= k)
                FAILF("error: %s_reduction: a state of a selected pair is out of range\n", name);
            (*first)[i] = (int)a; (*second)[i] = (int)b;
            r->selection[i] = i;
        }
        return reduction_parse_aggregation(r, name, agg);
    }
    if (!agg || j_is_null(agg)) r->agg_mode = AGG_NONE;
    else if (j_is_string(agg) && !strcmp(agg->u.s, "sum")) r->agg_mode = AGG_SUM;
    else if (j_is_string(agg) && !strcmp(agg->u.s, "avg")) r->agg_mode = AGG_AVG;
    else FAILF("error: %s_reduction: without a selection only \"sum\" and \"avg\" make sense\n", name);
    const int n = k * (k - 1);
    r->n = n;
    r->selection_len = n;
    r->selection = malloc((size_t)(n + 1) * sizeof(int));
    *first = malloc((size_t)(n + 1) * sizeof(int));
    *second = malloc((size_t)(n + 1) * sizeof(int));
    if (!r->selection || !*first || !*second) return -1;
    int i = 0;
    for (int a = 0; a < k; a++)
        for (int b = 0; b < k; b++)
            if (a != b) { r->selection[i] = i; (*first)[i] = a; (*second)[i] = b; i++; }
    if (r->agg_mode == AGG_AVG && n == 0) FAILF("error: %s_reduction: \"avg\" over nothing\n", name);
    return 0;
}

/* src/reduction.c:25-118 */
void host_reduction_weights(const host_reduction *r, long double *w, long double *divisor)
{
    for (int i = 0; i < r->n; i++) w[i] = 0;
    *divisor = 1;
    if (r->agg_mode == AGG_WEIGHTED_SUM) {
        for (int i = 0; i < r->selection_len; i++) w[r->selection[i]] += (long double)r->weights[i];
    } else if (r->agg_mode == AGG_SUM || r->agg_mode == AGG_AVG) {
        for (int i = 0; i < r->selection_len; i++) w[r->selection[i]] += 1;
        if (r->agg_mode == AGG_AVG) *divisor = r->selection_len;
    } else if (r->agg_mode == AGG_ONLY) {
        w[r->selection[0]] = 1;
    }
}
