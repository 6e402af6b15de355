/*
 * host_model.h -- host-side model-and-data structures and JSON validation.
 *
 * Mirrors the reference's model_and_data_struct (src/model.h:76-89) and
 * validate_model_and_data (src/parsemodel.c:786-913), and its column
 * reductions (src/reduction.h:16-22, src/parsereduction.c:162-195).
 */
#ifndef HOST_MODEL_H
#define HOST_MODEL_H

#include <stdint.h>
#include "host_json.h"
#include "host_k0.h"

#ifdef __cplusplus
extern "C" {
#endif

/* root prior modes: same numbering as plk.h / src/model.h:16-21 */
enum { HM_ROOT_NONE = 1, HM_ROOT_CUSTOM = 2, HM_ROOT_UNIFORM = 3, HM_ROOT_EQUILIBRIUM = 4 };

typedef struct {
    /* tree (src/csr_graph.h:17-24, :44-50; src/model.h:62-67) */
    int N, E, root;
    int *indptr, *indices, *preorder;
    int *edge_order;          /* user edge -> CSR edge */
    int *csr_to_user;         /* CSR edge -> user edge */
    double *edge_rates_user;  /* as given */
    double *edge_rates_csr;
    /* substitution model */
    int k;
    double *rate_matrix;      /* k*k raw */
    int use_equilibrium_divisor;
    double rate_divisor;
    int root_mode;
    double *root_custom;      /* k or NULL */
    k0_mixture mix;
    double *mix_rates, *mix_prior; /* owned storage behind mix */
    /* observations: exactly one of the two forms */
    long S;
    double *prob;             /* [S][N][k] (probability_array) or NULL */
    uint8_t *codes8;          /* [S][N] when nchar <= 256 */
    int nchar;
    double *defs;             /* [nchar][k] */
} host_model;

void host_model_init(host_model *m);
void host_model_clear(host_model *m);
/* 0 on success; on failure prints a diagnostic to stderr and returns -1 */
int host_model_parse(host_model *m, const jval *model_and_data);

/* aggregation modes: src/reduction.h:6-10 */
enum { AGG_NONE = 0, AGG_AVG = 1, AGG_SUM = 2, AGG_WEIGHTED_SUM = 3, AGG_ONLY = 4 };

typedef struct {
    int n;               /* axis length */
    int *selection;
    int selection_len;
    double *weights;     /* parallel to selection, AGG_WEIGHTED_SUM only */
    int agg_mode;
} host_reduction;

void host_reduction_init(host_reduction *r);
void host_reduction_clear(host_reduction *r);
/* root may be NULL (key absent) */
int host_reduction_parse(host_reduction *r, int n, const char *name, const jval *root);
/* selection of [first, second] state pairs (trans_reduction); caller frees *first / *second */
int host_pair_reduction_parse(host_reduction *r, int **first, int **second, int k, const char *name, const jval *root);
/* per-index aggregation weight (long double) and divisor; weights must hold n entries */
void host_reduction_weights(const host_reduction *r, long double *weights, long double *divisor);

/* strict object check: every key must be in allowed[]; required[] must be present */
int host_check_keys(const jval *obj, const char *const *required, const char *const *allowed, const char *what);

#ifdef __cplusplus
}
#endif
#endif
