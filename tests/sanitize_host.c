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

/* ---- engine stubs: no GPU in the sanitizer build ---- */
const char *plk_create_error(void) { return "stub: no engine in the sanitizer build"; }
int plk_create(plk_engine **out, int device) { (void)device; *out = NULL; return PLK_E_DEVICE; }
void plk_destroy(plk_engine *h) { (void)h; }
const char *plk_last_error(const plk_engine *h) { (void)h; return "stub"; }
int plk_set_tree(plk_engine *h, int N, const int *a, const int *b, const int *c) { (void)h; (void)N; (void)a; (void)b; (void)c; return PLK_E_DEVICE; }
int plk_set_model(plk_engine *h, int k, int C, const double *a, const double *b, const double *c, const double *d,
                  const double *e, int m, const double *f) { (void)h; (void)k; (void)C; (void)a; (void)b; (void)c; (void)d; (void)e; (void)m; (void)f; return PLK_E_DEVICE; }
int plk_set_patterns_codes(plk_engine *h, long S, const uint8_t *c, int w, int n, const double *d) { (void)h; (void)S; (void)c; (void)w; (void)n; (void)d; return PLK_E_DEVICE; }
int plk_set_patterns_dense(plk_engine *h, long S, const double *B, int w) { (void)h; (void)S; (void)B; (void)w; return PLK_E_DEVICE; }
int plk_set_site_weights(plk_engine *h, const double *w, int where) { (void)h; (void)w; (void)where; return PLK_E_DEVICE; }
int plk_ll(plk_engine *h, double *o, int w, double *s) { (void)h; (void)o; (void)w; (void)s; return PLK_E_DEVICE; }
int plk_deriv(plk_engine *h, const int *m, double *o, double *s) { (void)h; (void)m; (void)o; (void)s; return PLK_E_DEVICE; }
int plk_marginal(plk_engine *h, const int *m, double *o, double *s) { (void)h; (void)m; (void)o; (void)s; return PLK_E_DEVICE; }
int plk_edge_expect_multi(plk_engine *h, int n, const double *a, const double *b, int c, const int *m, double *o, double *s)
{ (void)h; (void)n; (void)a; (void)b; (void)c; (void)m; (void)o; (void)s; return PLK_E_DEVICE; }
int plk_hess(plk_engine *h, double *o) { (void)h; (void)o; return PLK_E_DEVICE; }

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
