/*
 * plk.h -- C-ABI of the MI355X pruning-likelihood engine (libarbplf_amd.so).
 *
 * This is the device boundary of the drop-in: everything above it is host C
 * (model parsing, tree, rate mixtures, reductions, JSON), everything below it
 * is hand-written HIP for gfx950.  The entry points replace, one for one, the
 * pieces of the reference (argriffing/phyly, paths relative to its root) that
 * its three query drivers call between "model parsed" and "table written":
 *
 *   plk_set_tree        <- csr_graph_struct + navigation preorder
 *                          (src/csr_graph.h:17-24, src/model.h:62-67)
 *   plk_set_model       <- cross_site_ws_update: normalised Q, category
 *                          rates/priors, P[c][e] = exp(Q r_c t_e)
 *                          (src/cross_site_ws.c:151-168, :200-242)
 *   plk_set_patterns_*  <- pmat_struct observation likelihoods
 *                          (src/model.h:41-47, src/parsemodel.c:459-628)
 *   plk_ll              <- ll _nd_accum_update: site loop x category loop x
 *                          evaluate_site_lhood + log + site aggregation
 *                          (src/arbplfll.c:110-177, src/evaluate_site_lhood.c:7-63)
 *   plk_deriv           <- deriv _nd_accum_update + evaluate_site_derivatives
 *                          (src/arbplfderiv.c:112-371)
 *   plk_marginal        <- marginal _nd_accum_update + evaluate_site_forward +
 *                          evaluate_site_marginal_unnormalized
 *                          (src/arbplfmarginal.c:111-264)
 *
 * Conventions: every function returns 0 on success and a nonzero PLK_E_* code
 * on failure (plk_last_error() gives the text).  The caller owns all buffers
 * it passes; the engine owns its device memory.  Calls are synchronous.  One
 * engine may be used from one thread at a time; distinct engines are
 * independent (one per GPU / per process rank).  There is no CPU fallback: if
 * no gfx950 device is usable, plk_create fails.
 *
 * Edge indices at this boundary are CSR edge indices (position in
 * `indices`), as in the reference's cross_site_ws; the host layer maps user
 * edge order <-> CSR order (src/csr_graph.c:29-46).
 */
#ifndef PLK_H
#define PLK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct plk_engine plk_engine;

enum {
    PLK_OK = 0,
    PLK_E_DEVICE = 1,    /* no usable GPU / HIP runtime error */
    PLK_E_ARG = 2,       /* invalid argument or call order */
    PLK_E_NOMEM = 3,     /* host or device allocation failed */
    PLK_E_UNSUPPORTED = 4
};

/* root prior modes, numbered as enum root_prior_mode in src/model.h:16-21 */
enum {
    PLK_ROOT_NONE = 1,        /* plain sum over root states */
    PLK_ROOT_CUSTOM = 2,      /* root_w = user distribution */
    PLK_ROOT_UNIFORM = 3,     /* 1/k */
    PLK_ROOT_EQUILIBRIUM = 4  /* root_w = stationary distribution */
};

/* where a caller buffer lives */
enum { PLK_HOST = 0, PLK_DEVICE = 1 };

/* Create an engine on HIP device `device` (0-based).  Fails loudly
 * (PLK_E_DEVICE) when the HIP runtime reports no device. */
int plk_create(plk_engine **out, int device);
void plk_destroy(plk_engine *h);
const char *plk_last_error(const plk_engine *h);
/* text for failures of plk_create itself (no engine yet) */
const char *plk_create_error(void);

/* Tree in CSR out-adjacency form; `preorder` is the BFS order from the root
 * (preorder[0] = root), all host arrays.  N nodes, E = N-1 edges. */
int plk_set_tree(plk_engine *h, int N, const int *indptr, const int *indices,
                 const int *preorder);

/* Model: k states, C rate categories.
 * Qn[k*k] row-major: rate matrix already divided by the rate divisor, with
 * its diagonal set to minus the row sums (src/cross_site_ws.c:217-232).
 * Qn_lo[k*k] (may be NULL = zeros): low-order words, Qn + Qn_lo being the
 * normalised matrix to ~106 bits; the reference normalises in exact ball
 * arithmetic, and d/dt exp(Qt) = Q exp(Qt) at long branches is only as accurate
 * as the zero row sums of Q (examples/JC.long.branch of the reference).
 * edge_rates_csr[E], cat_rates[C], cat_prior[C], root_w[k] (ignored for
 * NONE/UNIFORM).  Runs the device exp(Q r t) kernel for all (c, e). */
int plk_set_model(plk_engine *h, int k, int C, const double *Qn, const double *Qn_lo,
                  const double *edge_rates_csr,
                  const double *cat_rates, const double *cat_prior,
                  int root_mode, const double *root_w);

/* Recompute P for new edge rates only (branch-length optimisation loops). */
int plk_update_edge_rates(plk_engine *h, const double *edge_rates_csr);

/* Observations, compact form: codes[N][S] (site index fastest, one byte per
 * node per site) + definitions defs[nchar][k] (host).  `where` says whether
 * `codes` is a host or a device pointer; the engine copies it either way. */
int plk_set_patterns_codes(plk_engine *h, long S, const uint8_t *codes, int where,
                           int nchar, const double *defs);

/* Observations, dense form: B[N][k][S] doubles (site index fastest). */
int plk_set_patterns_dense(plk_engine *h, long S, const double *B, int where);

/* Optional per-site weights for the aggregated outputs (NULL = all 1).
 * A weight of exactly 0 still evaluates the site; selection is the caller's. */
int plk_set_site_weights(plk_engine *h, const double *w, int where);

/*
 * Log likelihoods.  site_ll_out: NULL or S doubles (host/device per `where`).
 * sum_out: NULL or 2 doubles {hi, lo}: sum_s w_s * ll_s as an unevaluated
 * double-double (hi + lo), summed in a fixed order (deterministic).
 */
int plk_ll(plk_engine *h, double *site_ll_out, int where, double *sum_out);

/*
 * The same evaluation queued on the engine's stream without waiting for it (optimisation loops, several GPUs):
 * site_ll_dev (NULL or S doubles) and sum_dev (NULL or 2 doubles {hi, lo}) are DEVICE buffers of the caller; nothing is
 * copied to the host.  plk_sync() -- or any synchronous call on the engine -- waits for the queued work.
 * plk_set_stream() makes the engine issue its work on a HIP stream of the caller (hipStream_t passed as void *; NULL =
 * the engine's own stream again), so that the caller's collectives and events are ordered with the engine's kernels.
 */
int plk_ll_async(plk_engine *h, double *site_ll_dev, double *sum_dev);
int plk_sync(plk_engine *h);
int plk_set_stream(plk_engine *h, void *hip_stream);

/*
 * Edge-rate derivatives d ll_s / d edge_rate_coefficient_e (CSR edge order).
 * edge_mask: NULL (all) or E ints, nonzero = requested.
 * site_edge_out: NULL or [S][E] doubles, host; unrequested edges get 0.
 * edge_sums_out: NULL or [E][2] double-double sums of w_s * d_{s,e}.
 */
int plk_deriv(plk_engine *h, const int *edge_mask,
              double *site_edge_out, double *edge_sums_out);

/*
 * Marginal state distributions.  node_mask: NULL (all) or N ints.
 * site_out: NULL or [S][N][k] doubles, host (unrequested nodes get 0).
 * sums_out: NULL or [N][k][2] double-double sums over sites of w_s * m_{s,a,i}.
 */
int plk_marginal(plk_engine *h, const int *node_mask,
                 double *site_out, double *sums_out);

/*
 * Conditional edge expectations: the shared core of arbplf-dwell, arbplf-trans
 * and arbplf-em-update.  For a direction matrix L (k x k, given as
 * an unevaluated sum L_hi + L_lo; L_lo may be NULL) the engine forms, per category c
 * and edge e, the Frechet derivative of the matrix exponential
 *     F_{c,e} = int_0^1 exp(s Qn u) L exp(s Qn (1-u)) du,   s = cat_rate_c * edge_rate_e
 * (the top-right block of exp([[s Qn, L], [0, s Qn]]), src/util.c:501-548) and returns
 *     x_{s,e} = sum_c prior_c * coef_{c,e} * fe_{c,e}^T F_{c,e} L_{c,b} / lhood_s
 * (src/evaluate_site_frechet.c:5-42 and the category loops of src/arbplfdwell.c:205-300,
 * src/arbplftrans.c:233-330, src/arbplfem.c:258-390), with coef_{c,e} =
 *   PLK_COEF_PRIOR            1                         (dwell: L = e_s e_s^T or diag(weights))
 *   PLK_COEF_PRIOR_RATE_EDGE  cat_rate_c * edge_rate_e  (trans: L = weights o Qn)
 *   PLK_COEF_PRIOR_RATE       cat_rate_c                (em-update numerators / denominators)
 * Outputs as for plk_deriv (CSR edge order, unrequested edges 0).
 */
enum { PLK_COEF_PRIOR = 0, PLK_COEF_PRIOR_RATE_EDGE = 1, PLK_COEF_PRIOR_RATE = 2 };
int plk_edge_expect(plk_engine *h, const double *L_hi, const double *L_lo, int coef_mode,
                    const int *edge_mask, double *site_edge_out, double *edge_sums_out);
/* nL direction matrices in one call (L_hi / L_lo: [nL][k][k]); site_out: NULL or [S][nL][E], sums_out: NULL or
 * [nL][E][2].  The k = 4 kernels evaluate up to four directions per pass (shared down pass and forward vectors):
 * em-update's two expectations, dwell per state, trans per state pair. */
int plk_edge_expect_multi(plk_engine *h, int nL, const double *L_hi, const double *L_lo, int coef_mode,
                          const int *edge_mask, double *site_out, double *sums_out);
/* the scaled Frechet matrices coef_{c,e} * F_{c,e} themselves, [C][E][k][k] host (tests) */
int plk_get_frechet_matrices(plk_engine *h, const double *L_hi, const double *L_lo, int coef_mode,
                             double *F_out);

/*
 * Fit the edge rate coefficients by maximum likelihood with the patterns, weights and tree resident on the
 * device.  The reference has no driver for this: its old-examples/opt.py and
 * old-examples/gell_dna_opt.py drive scipy's L-BFGS-B with arbplf_ll / arbplf_deriv through the Python API,
 * and test_scripts/test_em_monotonicity.py:95-130 iterates arbplf_em_update; this entry point is that loop.
 * method PLK_FIT_EM: r_e <- r_e * E[transitions on e] / E[rate-weighted dwell on e] (src/arbplfem.c:397-503);
 * method PLK_FIT_LBFGS: L-BFGS on log r_e, gradient from the deriv pass.  Edges with rate 0 or outside
 * edge_mask stay fixed.  Stops after max_iter iterations or when the gain in the weighted log likelihood is
 * <= ftol * max(1, |ll|).  rates_inout: E doubles, CSR order (also left set in the engine);
 * ll_trace: NULL or max_iter + 1 doubles (ll before the first and after every iteration).
 */
enum { PLK_FIT_EM = 0, PLK_FIT_LBFGS = 1 };
int plk_fit_edge_rates(plk_engine *h, int method, int max_iter, double ftol, const int *edge_mask,
                       double *rates_inout, double *ll_trace, int *iters_out, long *evals_out);

/*
 * Hessian of sum_s w_s ll_s with respect to the edge rate coefficients (SURVEY.md 8f-4, fp64 and
 * uncertified): replaces _recompute_second_order of src/arbplfhess.c:503-760 with its helpers
 * evaluate_site_derivatives (:343-437) and _lhood_hess_to_ll_hess (:455-493).
 * hess_sums_out: [E][E][2] double-double entries, CSR edge order, symmetric.
 */
int plk_hess(plk_engine *h, double *hess_sums_out);

/*
 * One process per GPU: the reduction step of the site-sharded path on RCCL, issued from the engine.
 * Sites are independent given (tree, Q, rates); every rank evaluates its block of site patterns and the only exchange
 * is the sum of the aggregated outputs (src/ndaccum.c:198-254 is the reference's only cross-site step): 2 doubles for
 * ll, 2E for edge gradients, 2Nk for site-summed marginals, always {hi, lo} words.  The engine loads the RCCL the
 * process already has (librccl.so.1; no link-time dependency) and queues ncclAllReduce on its own stream, behind the
 * kernels that produce the sums -- no framework call per step.
 *   plk_comm_unique_id   rank 0 makes the 128-byte id; the caller hands it to the other ranks (torch.distributed, MPI, a file)
 *   plk_comm_init        collective: every rank calls it with the same id
 *   plk_allreduce_sum_async  in-place sum over ranks of `count` doubles in DEVICE memory, queued on the engine's stream
 */
int plk_comm_available(void);                 /* 1 when RCCL can be loaded in this process (local, not a collective) */
int plk_comm_unique_id(unsigned char id_out[128]);
int plk_comm_init(plk_engine *h, int nranks, int rank, const unsigned char id[128]);
int plk_allreduce_sum_async(plk_engine *h, double *dev, long count);
int plk_comm_destroy(plk_engine *h);

/* Introspection for tests and profiling. */
int plk_get_transition_matrices(plk_engine *h, double *P_out /* [C][E][k][k] host */);
int plk_get_info(plk_engine *h, int what, long *out);
enum {
    PLK_INFO_LL_KERNEL = 0,       /* 0 = none yet, 1 = fused register-stack (k=4), 2 = generic vector, 3 = fp64 MFMA,
                                     4 = register-resident vector kernel (9 <= k <= 32) */
    PLK_INFO_STACK_SLOTS = 1,     /* register-stack slots the tree needs */
    PLK_INFO_PROGRAM_OPS = 2,     /* ops in the traversal program */
    PLK_INFO_LAST_LL_KERNEL_NS = 3, /* HIP-event time of the last ll traversal kernel */
    PLK_INFO_LAST_LL_TOTAL_NS = 4,  /* HIP-event time of the last whole plk_ll device work */
    PLK_INFO_LL_KERNEL_NS_SUM = 5,  /* HIP-event time of the traversal kernels of all ll evaluations since this item was */
    PLK_INFO_LL_KERNEL_COUNT = 6,   /* last read, and their number (reading waits for queued evaluations, then resets) */
    PLK_INFO_LL_VARIANT = 7,        /* k = 4 tile kernel of the last evaluation: 1 assembly interpreter, 3 C++ interpreter,
                                       5 assembly interpreter with pair tables, 6 the same with two sites per lane, 0 another kernel */
    PLK_INFO_PAIR_TABLES = 8,       /* two-leaf subtrees the last k = 4 evaluation read from tables */
    PLK_INFO_LL_EXEC_FLOPS = 9      /* fp64 flops per site the last ll traversal kernel executed (all categories): 2k^2 - k per
                                       matrix-vector product it ran (k padded to 16 rows on the matrix cores), k per elementwise
                                       multiply (leaf rows, stack pops); table look-ups, moves and rescaling count nothing */
};

/* force the generic (HBM-resident partials) traversal even where the fused
 * kernel applies; used by tests and by bench.py --kernel generic */
int plk_set_option(plk_engine *h, int option, long value);
enum { PLK_OPT_FORCE_GENERIC = 0, PLK_OPT_SITE_CHUNK = 1, PLK_OPT_FUSED_SITES_PER_LANE = 2 /* 0 auto, 1, 2 */,
       PLK_OPT_FUSED_ASM = 3 /* 1 (default): assembly interpreter loop where applicable, 0: C++ loop */,
       PLK_OPT_MFMA = 4 /* 1 (default): register-resident vector kernel for 9 <= k <= 32, fp64 matrix-core kernel for
                           33 <= k <= 64; 2: matrix-core kernel for all of 9 <= k <= 64; 0: generic vector kernel */,
       PLK_OPT_UP_NODES = 5 /* node-visit up pass for derivative queries, a bit per kernel family (default 2):
                               bit 1 (2): k = 4 kernels (k_up4_nodes: a third fewer HBM bytes than k_up4);
                               bit 2 (4): k = 4 kernels WITHOUT the table rebuild of nodes whose two children are leaves or
                               two-leaf nodes (by default their vectors do not go through HBM in either pass);
                               bit 3 (8): k = 4 kernels WITHOUT finishing two-leaf nodes inside their parent's visit (done for models with one
                               rate category: with four in flight the visit has no registers left for it);
                               bit 0 (1): matrix-core kernels, 21 <= k <= 64 or all of 9 <= k <= 64 under PLK_OPT_MFMA = 2
                               (k_up_nodes_mfma: fewer HBM bytes, same speed as measured in round 2).
                               Marginal queries always take the one-edge-at-a-time passes.  ARBPLF_UP_NODES in the
                               environment sets the initial value. */,
       PLK_OPT_PAIR_TABLES = 6 /* k = 4 trees within 4 stack slots and 16 character definitions: two-leaf subtrees as table
                               look-ups, grid-stride tile loop, one workgroup per CU.  1 (default): the interpreter with
                               two sites per lane (k_ll_fused4_v4) on 1536-site tiles, or 1024-site tiles when the
                               tables leave less LDS, else one site per lane; 5 / 6: two sites per lane on 1024 / 1536
                               sites; 2 / 3: one site per lane (k_ll_fused4_asm_pt) on 1024 / 512 sites; 0: the round-2
                               interpreter over 256-site tiles, no pair tables */,
       PLK_OPT_VEC_REG_STACK = 7 /* 1 (default): the vector kernels (9 <= k <= 20) keep the three busiest stack slots in
                               registers (2 waves per SIMD); 0: every waiting vector goes through HBM slots (round 2) */,
       PLK_OPT_MFMA_NS2 = 8 /* 1: the matrix-core ll kernel for 33 <= k <= 64 gives a wave two groups of 16 sites (every staged
                               matrix and every fragment read serve 128 sites of a workgroup); 0 (default): one group.  Measured
                               equal at BASELINE config 5 */ };

/* ------------------------------------------------------------------------------------------------------------
 * Several GPUs in one process: a group of engines, one per listed device, behind the same calls.
 *
 * Sites are independent given (tree, Q, rates); the only cross-site operations of the reference are the axis
 * reductions of nd_accum_accumulate (src/ndaccum.c:198-254) at the end of the site loops (src/arbplfll.c:139-170,
 * src/arbplfderiv.c:329-356, src/arbplfmarginal.c:237-256).  A group therefore gives engine i the contiguous block
 * [i*ceil(S/G), min(S, (i+1)*ceil(S/G))) of the site patterns (and of the weights), runs the engines concurrently
 * (one host thread each), returns per-site outputs at their global positions and adds the {hi, lo} partial sums of
 * the engines in engine order (long double): ll (2 doubles), edge sums (2E), node/state sums (2Nk), Hessian (2EE).
 * Deterministic for a given device list.  Every engine computes the same transition matrices itself (cheaper than
 * moving them).  The same device may be listed more than once (two engines then share it).
 * The host drivers (arbplf-ll and friends) build their group from the environment: ARBPLF_DEVICES=0,1,2,...
 * (default: ARBPLF_DEVICE or device 0).  Host buffers only; every call is synchronous.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct plk_group plk_group;

int plk_group_create(plk_group **out, int ndev, const int *devices);
void plk_group_destroy(plk_group *g);
const char *plk_group_last_error(const plk_group *g);
int plk_group_size(const plk_group *g);
plk_engine *plk_group_engine(plk_group *g, int i);           /* for plk_set_option / plk_get_info */
/* site block of engine i as set by the last plk_group_set_patterns_*: [*s0, *s1) */
int plk_group_block(const plk_group *g, int i, long *s0, long *s1);

int plk_group_set_tree(plk_group *g, int N, const int *indptr, const int *indices, const int *preorder);
int plk_group_set_model(plk_group *g, int k, int C, const double *Qn, const double *Qn_lo, const double *edge_rates_csr,
                        const double *cat_rates, const double *cat_prior, int root_mode, const double *root_w);
int plk_group_update_edge_rates(plk_group *g, const double *edge_rates_csr);
int plk_group_set_patterns_codes(plk_group *g, long S, const uint8_t *codes /* [N][S] host */, int nchar, const double *defs);
int plk_group_set_patterns_dense(plk_group *g, long S, const double *B /* [N][k][S] host */);
int plk_group_set_site_weights(plk_group *g, const double *w /* [S] host or NULL */);
int plk_group_ll(plk_group *g, double *site_ll_out, double *sum_out);
int plk_group_deriv(plk_group *g, const int *edge_mask, double *site_edge_out, double *edge_sums_out);
int plk_group_marginal(plk_group *g, const int *node_mask, double *site_out, double *sums_out);
int plk_group_edge_expect_multi(plk_group *g, int nL, const double *L_hi, const double *L_lo, int coef_mode,
                                const int *edge_mask, double *site_out, double *sums_out);
int plk_group_hess(plk_group *g, double *hess_sums_out);

#ifdef __cplusplus
}
#endif
#endif
