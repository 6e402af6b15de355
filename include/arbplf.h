/*
 * arbplf.h -- the operator-level C-ABI of the drop-in (libarbplf_amd.so).
 *
 * The reference exposes each query as a json_hom_fn_t
 *     json_t *arbplf_ll_run(void *userdata, json_t *root, int *retcode);
 * (src/arbplfll.h:10, src/arbplfderiv.h, src/arbplfmarginal.h; type in
 * src/runjson.h:32-33) and derives from it the string form
 *     char *(*string_hom_fn_t)(void *userdata, const char *s_in, int *retcode);
 * (src/runjson.h:28-29, jsonwrap in src/runjson.c:10-66), which is what its CLI
 * mains (src/arbplf-ll.c:4-15) and its Python module (src/arbplf.c:209-250) call.
 * jansson is not a dependency of this build, so the string form is the boundary:
 * the three functions below have exactly the string_hom_fn_t signature.
 *
 * Contract (as in the reference): `userdata` must be NULL; `s_in` is one JSON
 * document; on success *retcode = 0 and the return value is a malloc'd JSON
 * string the caller frees with free(); on failure *retcode != 0, NULL is
 * returned and a diagnostic has been written to stderr.  Never uses errno.
 * Unlike the reference (flint_cleanup at the end of every call) the functions
 * are thread-safe; calls are serialised on one cached GPU engine.
 *
 * The likelihood work runs on the GPU selected by the environment variable
 * ARBPLF_DEVICE (default 0).  There is no CPU path: without a usable MI355X
 * the call fails with a diagnostic.
 */
#ifndef ARBPLF_H
#define ARBPLF_H

#ifdef __cplusplus
extern "C" {
#endif

char *arbplf_ll_string(void *userdata, const char *s_in, int *retcode);
char *arbplf_deriv_string(void *userdata, const char *s_in, int *retcode);
char *arbplf_marginal_string(void *userdata, const char *s_in, int *retcode);
/* the expectation queries that reuse the same down / up passes
 * (src/arbplfdwell.h, src/arbplftrans.h, src/arbplfem.h: arbplf_dwell_run, arbplf_trans_run,
 * arbplf_em_update_run) */
char *arbplf_dwell_string(void *userdata, const char *s_in, int *retcode);
char *arbplf_trans_string(void *userdata, const char *s_in, int *retcode);
char *arbplf_em_update_string(void *userdata, const char *s_in, int *retcode);
/* the Hessian of the log likelihood (hess_query behind arbplf_second_order_run,
 * src/arbplfhess.c:1279-1343, :1736-1771), fp64 and uncertified */
char *arbplf_hess_string(void *userdata, const char *s_in, int *retcode);

/* Host-only validation of an input document (JSON grammar, model_and_data,
 * reductions) exactly as the corresponding query would perform it, without
 * touching the GPU.  what = "ll" | "deriv" | "marginal" | "dwell" | "trans" | "em_update" | "hess".
 * 0 = accepted. */
int arbplf_validate_string(const char *what, const char *s_in);

/* stdin -> stdout filter used by the CLI mains (run_string_script,
 * src/runjson.c:118-147): returns the process exit status */
int arbplf_run_stdin(char *(*f)(void *, const char *, int *));

/* release the cached engine (optional; e.g. before unloading the library) */
void arbplf_shutdown(void);

#ifdef __cplusplus
}
#endif
#endif
