/*
 * tests/sanitize_host.c -- AddressSanitizer / UBSan driver for the host layer (JSON reader, model and reduction
 * validation, K0), built and run by tests/test_host_sanitize.py on the CPU.  The GPU engine is replaced by stubs
 * that fail: only arbplf_validate_string() and the error paths of the drivers are exercised.
 *
 * usage: sanitize_host <kind> <file>...      every file is validated as <kind>; then byte-level mutations of each
 *                                            file are fed to the validator and to the driver entry point.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "arbplf.h"
#include "plk.h"

/* ---- engine stand-in: no GPU in the sanitizer build.  With FAKE_ENGINE=1 in the environment the entry points
 * succeed and fill their outputs with constants of the documented shapes, so that the drivers' table code runs under
 * the sanitizers; otherwise plk_create fails and the drivers' error paths run. ---- */
struct plk_engine { int N, E, k, C; long S; };
static int fake_on(void) { const char *e = getenv("FAKE_ENGINE"); return e && *e == '1'; }
const char *plk_create_error(void) { return "stub: no engine in the sanitizer build"; }
int plk_create(plk_engine **out, int device) { (void)device; if (!fake_on()) { *out = NULL; return PLK_E_DEVICE; } *out = calloc(1, sizeof(struct plk_engine)); return *out ? PLK_OK : PLK_E_NOMEM; }
void plk_destroy(plk_engine *h) { free(h); }
int plk_update_edge_rates(plk_engine *h, const double *r) { (void)h; return r ? PLK_OK : PLK_E_ARG; }
const char *plk_last_error(const plk_engine *h) { (void)h; return "stub"; }
int plk_set_tree(plk_engine *h, int N, const int *a, const int *b, const int *c) { (void)a; (void)b; (void)c; h->N = N; h->E = N - 1; return PLK_OK; }
int plk_set_model(plk_engine *h, int k, int C, const double *a, const double *b, const double *c, const double *d,
                  const double *e, int m, const double *f) { (void)a; (void)b; (void)c; (void)d; (void)e; (void)m; (void)f; h->k = k; h->C = C; return PLK_OK; }
int plk_set_patterns_codes(plk_engine *h, long S, const uint8_t *c, int w, int n, const double *d)
{
    (void)w;
    unsigned long long acc = 0;                                   /* read every byte the driver promised */
    for (long i = 0; i < S * h->N; i++) acc += c[i];
    for (int i = 0; i < n * h->k; i++) acc += (unsigned long long)d[i];
    h->S = S;
    return acc == ~0ULL ? PLK_E_ARG : PLK_OK;
}
int plk_set_patterns_dense(plk_engine *h, long S, const double *B, int w)
{
    (void)w;
    double acc = 0;
    for (long i = 0; i < S * h->N * h->k; i++) acc += B[i];
    h->S = S;
    return acc < 0 ? PLK_E_ARG : PLK_OK;
}
int plk_set_site_weights(plk_engine *h, const double *w, int where) { (void)where; double a = 0; if (w) for (long i = 0; i < h->S; i++) a += w[i]; return a != a ? PLK_E_ARG : PLK_OK; }
int plk_ll(plk_engine *h, double *o, int w, double *s) { (void)w; if (o) for (long i = 0; i < h->S; i++) o[i] = -1.5; if (s) { s[0] = -1.5 * h->S; s[1] = 0; } return PLK_OK; }
int plk_deriv(plk_engine *h, const int *m, double *o, double *s)
{
    double a = 0; if (m) for (int i = 0; i < h->E; i++) a += m[i];
    if (o) for (long i = 0; i < h->S * h->E; i++) o[i] = 0.25;
    if (s) for (int i = 0; i < 2 * h->E; i++) s[i] = 0.5;
    return a < 0 ? PLK_E_ARG : PLK_OK;
}
int plk_marginal(plk_engine *h, const int *m, double *o, double *s)
{
    double a = 0; if (m) for (int i = 0; i < h->N; i++) a += m[i];
    if (o) for (long i = 0; i < h->S * h->N * h->k; i++) o[i] = 0.125;
    if (s) for (int i = 0; i < 2 * h->N * h->k; i++) s[i] = 0.5;
    return a < 0 ? PLK_E_ARG : PLK_OK;
}
int plk_edge_expect_multi(plk_engine *h, int n, const double *a, const double *b, int c, const int *m, double *o, double *s)
{
    double acc = 0; (void)c;
    for (int i = 0; i < n * h->k * h->k; i++) acc += a[i] + (b ? b[i] : 0);
    if (m) for (int i = 0; i < h->E; i++) acc += m[i];
    if (o) for (long i = 0; i < h->S * n * h->E; i++) o[i] = 0.75;
    if (s) for (int i = 0; i < 2 * n * h->E; i++) s[i] = 0.5;
    return acc != acc ? PLK_E_ARG : PLK_OK;
}
int plk_hess(plk_engine *h, double *o) { for (int i = 0; i < 2 * h->E * h->E; i++) o[i] = -2.0; return PLK_OK; }

static char *slurp(const char *path, size_t *n)
{
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    rewind(f);
    char *buf = malloc((size_t)sz + 1);
    if (buf && fread(buf, 1, (size_t)sz, f) != (size_t)sz) { free(buf); buf = NULL; }
    fclose(f);
    if (buf) { buf[sz] = 0; *n = (size_t)sz; }
    return buf;
}

static unsigned long long rng_state = 88172645463325252ULL;
static unsigned rnd(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (unsigned)(rng_state >> 11); }

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const char *kind = argv[1];
    char *(*driver)(void *, const char *, int *) =
        !strcmp(kind, "ll") ? arbplf_ll_string : !strcmp(kind, "deriv") ? arbplf_deriv_string :
        !strcmp(kind, "marginal") ? arbplf_marginal_string : !strcmp(kind, "dwell") ? arbplf_dwell_string :
        !strcmp(kind, "trans") ? arbplf_trans_string : !strcmp(kind, "em_update") ? arbplf_em_update_string : arbplf_hess_string;
    int accepted = 0, files = 0;
    long mutants = 0;
    for (int a = 2; a < argc; a++) {
        size_t n = 0;
        char *doc = slurp(argv[a], &n);
        if (!doc) { fprintf(stdout, "cannot read %s\n", argv[a]); return 3; }
        files++;
        if (arbplf_validate_string(kind, doc) == 0) accepted++;
        /* the driver itself must fail cleanly at the engine (stub) without leaking */
        int rc = 0;
        char *out = driver(NULL, doc, &rc);
        free(out);
        const int nm = n > 20000 ? 40 : 300;
        char *mut = malloc(n + 8);
        for (int m = 0; m < nm; m++) {
            memcpy(mut, doc, n + 1);
            size_t len = n;
            const int edits = 1 + (int)(rnd() % 3);
            for (int e = 0; e < edits && len > 0; e++) {
                const size_t pos = rnd() % len;
                switch (rnd() % 5) {
                case 0: mut[pos] = (char)(rnd() % 256); break;                       /* random byte */
                case 1: mut[pos] = "[]{},:\"0-9.eE tfn"[rnd() % 18]; break;          /* structural byte */
                case 2: memmove(mut + pos, mut + pos + 1, len - pos); len--; break;  /* delete */
                case 3: mut[len = pos] = 0; break;                                   /* truncate */
                default: if (pos + 1 < len) { char t = mut[pos]; mut[pos] = mut[pos + 1]; mut[pos + 1] = t; } break;
                }
            }
            mut[len] = 0;
            (void)arbplf_validate_string(kind, mut);
            if (m % 10 == 0) { out = driver(NULL, mut, &rc); free(out); }
            mutants++;
        }
        free(mut);
        free(doc);
    }
    arbplf_shutdown();
    fprintf(stdout, "files %d accepted %d mutants %ld\n", files, accepted, mutants);
    return 0;
}
