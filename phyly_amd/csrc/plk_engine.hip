/*
 * plk_engine.hip -- MI355X (gfx950) pruning-likelihood engine behind include/plk.h.
 *
 * Kernels (all fp64; see DESIGN.md for layouts and rooflines).  In this file:
 *   k_expm_dd<F>     P[c][e] = exp(Qn * r_c * t_e) in double-double arithmetic, one workgroup per (c, e);
 *                    replaces arb_mat_exp in src/cross_site_ws.c:151-168.  F = true: the 2k x 2k block
 *                    exponential whose top-right block is the Frechet matrix (src/util.c:501-548).
 *   k_build_*        P gathered into traversal-program order, tip tables P_e * defs[code]
 *   k_ll_generic<K>  post-order traversal program for any k <= 64, stack slots in HBM
 *   k_down_store<K>, k_up<K>   down pass with stored vectors + BFS up pass (deriv, marginal, edge
 *                    expectations, Hessian rows) for dense observations and small k
 *   k_wsum_rows, k_dd_final, k_gram   deterministic double-double reductions over sites
 *                    (src/ndaccum.c:198-254 for aggregated site axes)
 * and in the included headers:
 *   plk_fused4_asm.h   k_ll_fused4_asm: the k = 4 headline kernel, interpreter in CDNA4 assembly
 *                      (src/arbplfll.c:139-170 x src/evaluate_site_lhood.c:21-57 x src/util.c:242-301)
 *   plk_fused4.h       the same program interpreted by a C++ loop (deeper stacks, two sites per lane option)
 *   plk_updown4.h      k_down_fused4 / k_down_store4 / k_up4: k = 4 deriv, marginal, expectations
 *                      (src/evaluate_site_forward.c:32-105, src/arbplfderiv.c:112-371,
 *                      src/arbplfmarginal.c:111-264, src/evaluate_site_frechet.c:5-42)
 *   plk_vec.h          k_ll_vec: 9 <= k <= 32 on the vector pipe
 *   plk_mfma.h, plk_mfma_updown.h   fp64 matrix-core kernels for k up to 64
 * Host side of the engine: model / pattern set-up, program builder (Sethi-Ullman ordered post-order),
 * plk_ll / plk_deriv / plk_marginal / plk_edge_expect(_multi) / plk_fit_edge_rates / plk_hess.
 *
 * There is no CPU path in this file: every entry point needs a HIP device.
 */
#include <hip/hip_runtime.h>
#include <algorithm>
#include <utility>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "plk.h"
#include "plk_dd.h"
#include "plk_program.h"     /* traversal program: opcodes, builder, device formats and their checkers (host only) */

#define PLK_MAX_K 64
#define PLK_MAX_C 64

struct plk_engine {
    int device = 0;
    hipStream_t stream = nullptr, own_stream = nullptr;   /* stream in use; the engine's own one */
    bool info_pending = false;           /* event times of the last ll evaluation not read yet */
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    /* HIP-event pairs around the traversal kernel of the last PLK_EV_RING ll evaluations (read lazily, so that
     * queued evaluations are not serialised by the measurement) */
    hipEvent_t evk[64][2] = {};
    bool evk_pending[64] = {};
    int evk_next = 0;
    long evk_sum_ns = 0, evk_count = 0;
    std::string err;

    /* tree */
    int N = 0, E = 0;
    std::vector<int> indptr, indices, preorder, b_to_idx, idx_to_a;
    int *d_indptr = nullptr, *d_indices = nullptr, *d_preorder = nullptr;

    /* model */
    int k = 0, C = 0, K = 0, root_mode = 0;
    std::vector<double> Qn, edge_rates, cat_rates, cat_prior, root_w;
    double *d_Qn = nullptr, *d_edge_rates = nullptr, *d_cat_rates = nullptr, *d_cat_prior = nullptr;
    double *d_root_w = nullptr;          /* K, zero padded; the weights of the root dot */
    dd *d_Pdd = nullptr;                 /* [C][E][k*k] unrounded */
    double *d_P = nullptr, *d_dP = nullptr; /* [C][E][k][k] rounded; dP = r_c Qn P */
    bool dp_valid = false;                  /* d_dP holds dP of the current P (K1 skips it for ll evaluations: ensure_dP) */
    dd *d_scratch = nullptr;
    size_t pdd_cap = 0, p_cap = 0, dp_cap = 0, scratch_cap = 0;
    bool model_dirty = true;

    /* patterns */
    long S = 0, Spad = 0;
    int pat_mode = 0;                    /* 0 none, 1 codes, 2 dense */
    uint8_t *d_codes = nullptr;          /* [N][Spad] */
    int nchar = 0;
    std::vector<double> defs;            /* [nchar][k] host */
    std::vector<char> node_observed;     /* N: every site's code at the node is a single observed state (definition = one 1.0) */
    double *d_defs = nullptr;            /* [nchar][K] zero padded */
    double *d_B = nullptr;               /* [N][k][S] */
    std::vector<char> node_has_data;     /* internal nodes whose observations are not all-ones */
    double *d_w = nullptr;

    /* traversal program (plk_program.h) */
    /* prog_dirty: the program must be rebuilt (tree / patterns / options changed); fmt_dirty: its device formats are
     * not uploaded for kernel kind fmt_kind; tables_dirty: the P-dependent tables (matrix stream, tip tables, A
     * fragments) are older than P */
    bool prog_dirty = true, fmt_dirty = true, tables_dirty = true;
    long fmt_kind = 0;
    PlkProgram pg;
    std::vector<plk_op2> &ops = pg.ops;
    std::vector<int> &op_edge = pg.op_edge;          /* CSR edge per op or -1 */
    std::vector<int> &tip_edge = pg.tip_edge;        /* CSR edge per tip slot */
    std::vector<char> &scale_node = pg.scale_node;   /* N: node vectors rescaled here (every >= 16 accumulated edges) */
    std::vector<int> &obs_nodes = pg.obs_nodes;      /* nodes whose codes the fused kernels stage */
    int &slots_needed = pg.slots_needed;
    PlkFused fu;                         /* fused k = 4 formats of the program */
    PlkFusedPT fpt;                      /* ... and of the pair-table interpreter (k_ll_fused4_asm_pt) */
    bool fmt_pt = false;                 /* the uploaded kind-1 formats are the pair-table ones */
    unsigned *d_words_pt = nullptr;
    PlkFusedV4 fv4;                      /* 64-bit ops of the two-sites-per-lane interpreter */
    PlkVecPT vpt;                        /* the vector ll kernel's program on pair tables */
    bool vec_pt = false;
    int pt_kind = 0, pt_tile_sites = 0;  /* 1: k_ll_fused4_asm_pt, 2: k_ll_fused4_v4; sites per tile of the uploaded formats */
    unsigned v4_tip_base = 0;            /* LDS address of the table image = static LDS of the kernel */
    int *d_row_nodes = nullptr, *d_tabs = nullptr;   /* [2][rows] staged-row nodes; [4][ntab] unit, edge, leaf edges of a pair */
    int num_cus = 256;
    int2 *d_ops = nullptr;
    int4 *d_fops = nullptr;              /* fused-kernel program (C++ interpreter) */
    unsigned *d_words = nullptr;         /* fused-kernel program (assembly interpreter) */
    std::vector<int> &mat_edge = fu.mat_edge;        /* CSR edge per compact matrix of the fused stream */
    int *d_op_edge = nullptr, *d_tip_edge = nullptr, *d_obs_nodes = nullptr, *d_mat_edge = nullptr, *d_edge_slot = nullptr;
    double *d_PS = nullptr;              /* [C][nops][K*K] transposed: PS[j*K+i] = P[i][j] */
    double *d_tip = nullptr;             /* [C][ntips][nchar][4] */
    double *d_frag = nullptr; size_t frag_cap = 0;   /* MFMA A fragments */
    double *d_root_wd = nullptr;         /* root weights, distributed layout */
    int4 *d_mops = nullptr;              /* MFMA program (observation ops chained for the value prefetch) */
    double *d_exL = nullptr, *d_exF = nullptr;   /* edge expectations: directions, scaled Frechet matrices */
    int *d_exmask = nullptr;
    dd *d_exscr = nullptr;
    size_t exL_cap = 0, exF_cap = 0, exmask_cap = 0, exscr_cap = 0;
    int *d_u4pack = nullptr;             /* k = 4 down / up passes: the call's integer tables, one block */
    double *d_u4tip = nullptr;           /* ... and its tip / edge-form tip tables */
    size_t u4pack_cap = 0, u4tip_cap = 0;
    double *d_stage = nullptr; size_t stage_cap = 0;   /* transposed per-site outputs on their way to the host */
    double *d_uvmat = nullptr; size_t uvmat_cap = 0;   /* vector down / up passes: PT and the up-pass matrix stream */
    int mfma_first_mv = -1, mfma_first_slot = -1, mfma_first_row = 0, vec_second_row = 0;
    size_t ps_cap = 0, tip_cap = 0;

    /* workspaces */
    double *d_slots = nullptr; size_t slots_cap = 0;
    double *d_site_ll = nullptr; size_t site_ll_cap = 0;
    dd *d_partial = nullptr; size_t partial_cap = 0;
    double *d_work = nullptr; size_t work_cap = 0;   /* deriv / marginal workspace */

    /* options / info */
    long opt_force_generic = 0, opt_site_chunk = 0, opt_fused_ns = 0, opt_fused_asm = 1, opt_mfma = 1, opt_up_nodes = 2, opt_pair_tables = 1, opt_vec_reg_stack = 1, opt_mfma_ns2 = 0;
    void *comm = nullptr;                /* ncclComm_t of the one-process-per-GPU reduction step */
    int comm_ranks = 0;
    long info_ll_kernel = 0, info_ll_kernel_ns = 0, info_ll_total_ns = 0, info_ll_variant = 0, info_ll_exec_flops = 0;
};

static std::string g_create_error;

/* RCCL is taken from the process at run time (the framework's copy when there is one): no link-time dependency, and a
 * process that never calls plk_comm_* never touches it */
struct PlkNcclId { char internal[128]; };
struct PlkRccl {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, /* ncclUniqueId by value: 128 bytes */ PlkNcclId, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string err;
};
static PlkRccl g_rccl;
static std::mutex g_rccl_mu;

static bool rccl_load()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return true;
    const char *names[] = {"librccl.so.1", "librccl.so"};
    void *lib = nullptr;
    for (const char *n : names) if (!lib) lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);     /* the copy already in the process */
    for (const char *n : names) if (!lib) lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!lib) { g_rccl.err = std::string("RCCL is not available: ") + (dlerror() ? dlerror() : "librccl.so.1 not found"); return false; }
    g_rccl.GetUniqueId = reinterpret_cast<int (*)(void *)>(dlsym(lib, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<int (*)(void **, int, PlkNcclId, int)>(dlsym(lib, "ncclCommInitRank"));
    g_rccl.AllReduce = reinterpret_cast<int (*)(const void *, void *, size_t, int, int, void *, hipStream_t)>(dlsym(lib, "ncclAllReduce"));
    g_rccl.CommDestroy = reinterpret_cast<int (*)(void *)>(dlsym(lib, "ncclCommDestroy"));
    g_rccl.GetErrorString = reinterpret_cast<const char *(*)(int)>(dlsym(lib, "ncclGetErrorString"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy) { g_rccl.err = "RCCL: entry points missing"; return false; }
    g_rccl.lib = lib;
    return true;
}

static std::string rccl_text(int rc) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : ("code " + std::to_string(rc)); }



/* Live engine handles.  Every entry point refuses a handle that plk_create did not return or that plk_destroy has
 * already taken (PLK_E_ARG instead of a use after free); plk_destroy of such a handle is a no-op.  Once the process
 * is exiting (atexit) handles are only forgotten, not torn down: the HIP runtime's own exit handlers may already
 * have run by the time a late destructor (a garbage-collected binding object, a static of the caller) gets here. */
static std::mutex g_live_mu;
static std::set<const plk_engine *> g_live;
static bool g_exiting = false, g_atexit_set = false;

static bool plk_live(const plk_engine *h)
{
    if (!h) return false;
    std::lock_guard<std::mutex> lk(g_live_mu);
    return !g_exiting && g_live.count(h) != 0;
}

/* HIPCHKC: the same inside a function that owns per-call device memory through a local `cleanup` lambda */
#define HIPCHKC(h, call)                                                           \
    do {                                                                           \
        hipError_t e_ = (call);                                                    \
        if (e_ != hipSuccess) {                                                    \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);         \
            cleanup();                                                             \
            return PLK_E_DEVICE;                                                   \
        }                                                                          \
    } while (0)
#define HIPCHK(h, call)                                                            \
    do {                                                                           \
        hipError_t e_ = (call);                                                    \
        if (e_ != hipSuccess) {                                                    \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);         \
            return PLK_E_DEVICE;                                                   \
        }                                                                          \
    } while (0)

template <typename T>
static int dev_alloc(plk_engine *h, T **p, size_t n)
{
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    if (n == 0) n = 1;
    hipError_t e = hipMalloc((void **)p, n * sizeof(T));
    if (e != hipSuccess) {
        h->err = std::string("hipMalloc: ") + hipGetErrorString(e);
        *p = nullptr;
        return PLK_E_NOMEM;
    }
    return PLK_OK;
}

template <typename T>
static int dev_upload(plk_engine *h, T **p, const T *src, size_t n)
{
    int rc = dev_alloc(h, p, n);
    if (rc) return rc;
    if (n) HIPCHK(h, hipMemcpy(*p, src, n * sizeof(T), hipMemcpyHostToDevice));
    return PLK_OK;
}

template <typename T>
static int dev_reserve(plk_engine *h, T **p, size_t *cap, size_t n)
{
    if (*p && *cap >= n) return PLK_OK;
    int rc = dev_alloc(h, p, n);
    *cap = rc ? 0 : n;
    return rc;
}

/* ====================================================================== */
/* K1: P = exp(Qn * r_c * t_e) in double-double                            */
/* ====================================================================== */

__device__ static inline dd dd_div_d(dd x, double y)
{
    double q1 = x.hi / y;
    dd p = dd_two_prod(q1, y);
    dd r = dd_add(x, dd_make(-p.hi, -p.lo));
    double q2 = r.hi / y;
    return dd_quick_two_sum(q1, q2);
}

__device__ static void dd_matmul_block(int k, const dd *A, const dd *B, dd *Cm)
{
    int kk = k * k;
    for (int idx = threadIdx.x; idx < kk; idx += blockDim.x) {
        int i = idx / k, j = idx - i * k;
        /* compensated dot product (Ogita, Rump & Oishi's Dot2 on double-double operands): the high words are summed with
         * an exact two-sum, every rounding error and the cross terms go to one low accumulator, one renormalisation at the
         * end -- 12 flops per term instead of the 28 of dd_add(dd_mul); the result carries ~2^-104 relative to
         * sum |a||b|, the same class as before (K1 for k = 61: 0.99 -> see profiles/r03_exp_codon_kernel_variants.json) */
        double sh = 0.0, sl = 0.0;
        for (int l = 0; l < k; l++) {
            const dd a = A[i * k + l], b = B[l * k + j];
            const double p = a.hi * b.hi;
            double e = fma(a.hi, b.hi, -p);
            e = fma(a.hi, b.lo, e);
            e = fma(a.lo, b.hi, e);
            const double t = sh + p, bb = t - sh;
            sl += ((sh - (t - bb)) + (p - bb)) + e;
            sh = t;
        }
        Cm[idx] = dd_quick_two_sum(sh, sl);
    }
}

/* The same product for matrices that live in global scratch (k > 26: codon models, every Frechet block matrix from
 * k = 14 up).  One thread per entry (above) sends two 16-byte loads per term through the CU's vector memory path, which
 * is what bounded K1 at k = 61.  Here a wave owns R rows x 64 columns of the result: it copies its R rows of A into its
 * own LDS strip once, then per term l every lane loads its B[l][j] (coalesced) and reads the R values A[i0 + r][l] from
 * the LDS strip (one address for the whole wave: a broadcast) -- 1 vector-memory load per R terms instead of 2 per term.
 * The sums run over l in the same order with the same Dot2 update, so the result is the one dd_matmul_block gives. */
#define EXPM_TILE_ROWS 4
template <int R>
__device__ static void dd_matmul_tiled(int k, const dd *A, const dd *B, dd *Cm, dd *strips)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    dd *arow = strips + (size_t)wave * EXPM_TILE_ROWS * k;
    const int ngroups = (k + R - 1) / R, nchunks = (k + 63) / 64;
    for (int item = wave; item < ngroups * nchunks; item += nw) {
        const int g = item / nchunks, ch = item - g * nchunks;
        const int i0 = g * R, j = ch * 64 + lane;
        __builtin_amdgcn_wave_barrier();                 /* the strip's previous reads are done (same wave: in order) */
#pragma unroll
        for (int r = 0; r < R; r++)
            for (int t = lane; t < k; t += 64) arow[r * k + t] = i0 + r < k ? A[(size_t)(i0 + r) * k + t] : dd_make(0.0, 0.0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const bool act = j < k;
        const dd *bcol = B + (act ? j : 0);
        double sh[R], sl[R];
#pragma unroll
        for (int r = 0; r < R; r++) { sh[r] = 0.0; sl[r] = 0.0; }
#pragma unroll 2
        for (int l = 0; l < k; l++) {
            const dd b = bcol[(size_t)l * k];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const dd a = arow[r * k + l];
                const double p = a.hi * b.hi;
                double e = fma(a.hi, b.hi, -p);
                e = fma(a.hi, b.lo, e);
                e = fma(a.lo, b.hi, e);
                const double t = sh[r] + p, bb = t - sh[r];
                sl[r] += ((sh[r] - (t - bb)) + (p - bb)) + e;
                sh[r] = t;
            }
        }
        if (act) {
#pragma unroll
            for (int r = 0; r < R; r++) if (i0 + r < k) Cm[(size_t)(i0 + r) * k + j] = dd_quick_two_sum(sh[r], sl[r]);
        }
    }
}

/* strips == nullptr: the buffers are in LDS (small k), one thread per entry */
__device__ static void dd_matmul(int k, const dd *A, const dd *B, dd *Cm, dd *strips)
{
    if (!strips) { dd_matmul_block(k, A, B, Cm); return; }
    const int nw = blockDim.x >> 6;
    if (((k + 3) / 4) * ((k + 63) / 64) >= nw) dd_matmul_tiled<4>(k, A, B, Cm, strips);
    else dd_matmul_tiled<2>(k, A, B, Cm, strips);
}

/* dP = r_c Qn P, rounded, every entry a double-double sum over l in order (src/util.c:338-345 is what reads it);
 * strips as for dd_matmul: null = one thread per entry, else a wave owns 4 rows x 64 columns and keeps its rows of Qn in LDS */
__device__ static void dd_rate_product(int k, const double *__restrict__ Qn /* [2][k*k] */, const dd *O, double rc, double *dPout, dd *strips)
{
    const int kk = k * k;
    if (!strips) {
        for (int idx = threadIdx.x; idx < kk; idx += blockDim.x) {
            const int i = idx / k, j = idx - i * k;
            dd acc = dd_make(0.0, 0.0);
            for (int l = 0; l < k; l++) acc = dd_add(acc, dd_mul(O[l * k + j], dd_make(Qn[i * k + l], Qn[kk + i * k + l])));
            dPout[idx] = dd_mul_d(acc, rc).hi;
        }
        return;
    }
    constexpr int R = EXPM_TILE_ROWS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    dd *arow = strips + (size_t)wave * R * k;
    const int ngroups = (k + R - 1) / R, nchunks = (k + 63) / 64;
    for (int item = wave; item < ngroups * nchunks; item += nw) {
        const int g = item / nchunks, ch = item - g * nchunks;
        const int i0 = g * R, j = ch * 64 + lane;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < R; r++)
            for (int t = lane; t < k; t += 64)
                arow[r * k + t] = i0 + r < k ? dd_make(Qn[(i0 + r) * k + t], Qn[kk + (i0 + r) * k + t]) : dd_make(0.0, 0.0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const bool act = j < k;
        const dd *bcol = O + (act ? j : 0);
        dd acc[R];
#pragma unroll
        for (int r = 0; r < R; r++) acc[r] = dd_make(0.0, 0.0);
        for (int l = 0; l < k; l++) {
            const dd b = bcol[(size_t)l * k];
#pragma unroll
            for (int r = 0; r < R; r++) acc[r] = dd_add(acc[r], dd_mul(b, arow[r * k + l]));
        }
        if (act) {
#pragma unroll
            for (int r = 0; r < R; r++) if (i0 + r < k) dPout[(size_t)(i0 + r) * k + j] = dd_mul_d(acc[r], rc).hi;
        }
    }
}

/* dynamic LDS of a k_expm_dd launch on kdim x kdim matrices: the six buffers, or one strip of rows per wave */
static inline size_t expm_lds_bytes(size_t kdim, int threads, int *use_lds)
{
    const size_t buffers = 6 * kdim * kdim * sizeof(dd);
    *use_lds = buffers <= 64 * 1024;
    return *use_lds ? buffers : (size_t)(threads / 64) * EXPM_TILE_ROWS * kdim * sizeof(dd);
}

#define EXPM_TERMS 16     /* with |A| <= 2^-5 after scaling: 2^-80 / 16! < 1e-37, below double-double resolution */
/* The degree-16 Taylor polynomial is evaluated by Paterson-Stockmeyer: powers A^2, A^3, A^4 (three products), then
 * Horner in A^4 over the four cubic blocks B_i = c_4i I + c_4i+1 A + c_4i+2 A^2 + c_4i+3 A^3 (three more products) --
 * 6 double-double matrix products instead of 15, before the squarings.  Six k x k buffers per matrix instead of four. */
#define EXPM_BUFFERS 6

/*
 * FRECHET = false: P = exp(s Qn), s = r_c t_e; outputs unrounded / rounded P and dP = r_c Qn P.
 * FRECHET = true: the 2k x 2k block matrix [[s Qn, L], [0, s Qn]] is exponentiated instead and the
 * top-right block -- the Frechet derivative of exp at s Qn in direction L, i.e.
 * int_0^1 exp(s Qn u) L exp(s Qn (1-u)) du (src/util.c:501-548) -- is written to Fout, scaled by
 * coef = 1 (PLK_COEF_PRIOR), s (PLK_COEF_PRIOR_RATE_EDGE) or r_c (PLK_COEF_PRIOR_RATE).
 */
/* k = 4, fused kernels: K1 also writes what the traversal reads, so that an evaluation with new edge rates is K1 + the
 * traversal kernel + the reduction and nothing else: P_e (transposed) into its matrix of the stream for an internal
 * edge, P_e * defs[code] (double-double, constant rows exact) into its tip slot for a leaf edge */
struct ExpmPost {
    const int *edge_slot;      /* [2][E]: matrix index of the edge or -1 | tip slot of the edge or -1; null: no post */
    int nmat1, ntips1, nchar;  /* matrices / tip slots per category (incl. the spare / pseudo one), definitions */
    const double *defs;        /* [nchar][4] */
    double *PS, *tip;
    int skip_dP;               /* an ll evaluation does not read dP: k_dP_dd makes it when a derivative asks (ensure_dP) */
};

template <bool FRECHET>
__global__ __launch_bounds__(1024) void k_expm_dd(int ks, int E, const double *__restrict__ Qn /* [2][ks*ks]: hi then lo */,
                          const double *__restrict__ edge_rates,
                          const double *__restrict__ cat_rates,
                          dd *__restrict__ Pdd, double *__restrict__ P, double *__restrict__ dP,
                          dd *gscratch, int use_lds,
                          const double *__restrict__ Lm /* [2][ks*ks] */, int coef_mode,
                          const int *__restrict__ edge_mask, double *__restrict__ Fout, ExpmPost post)
{
    extern __shared__ double smem_raw[];
    __shared__ double s_row[2 * PLK_MAX_K];
    __shared__ int s_sq;
    __shared__ dd s_coef[EXPM_TERMS + 1];           /* 1 / n! in double-double */
    const int ce = blockIdx.x;
    const int c = ce / E, e = ce - c * E;
    const int k = FRECHET ? 2 * ks : ks;
    const int kk = k * k, kks = ks * ks;
    if (FRECHET && edge_mask && !edge_mask[e]) {
        for (int idx = threadIdx.x; idx < kks; idx += blockDim.x) Fout[(size_t)ce * kks + idx] = 0.0;
        return;
    }
    dd *base = use_lds ? reinterpret_cast<dd *>(smem_raw) : gscratch + (size_t)ce * EXPM_BUFFERS * kk;
    dd *strips = use_lds ? nullptr : reinterpret_cast<dd *>(smem_raw);       /* global buffers: row strips of the tiled product */
    dd *X = base, *X2 = base + kk, *O = base + 2 * kk, *W = base + 3 * kk, *X3 = base + 4 * kk, *X4 = base + 5 * kk;

    const dd s = dd_two_prod(cat_rates[c], edge_rates[e]);
    for (int idx = threadIdx.x; idx < kk; idx += blockDim.x) {
        if (!FRECHET) {
            X[idx] = dd_mul(s, dd_make(Qn[idx], Qn[kk + idx]));
        } else {
            const int i = idx / k, j = idx - i * k;
            const int bi = i >= ks, bj = j >= ks;
            const int q = (i - bi * ks) * ks + (j - bj * ks);
            dd v = dd_make(0.0, 0.0);
            if (bi == bj) v = dd_mul(s, dd_make(Qn[q], Qn[kks + q]));
            else if (!bi) v = dd_make(Lm[q], Lm[kks + q]);
            X[idx] = v;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < k) {
        double r = 0;
        for (int j = 0; j < k; j++) r += fabs(X[threadIdx.x * k + j].hi);
        s_row[threadIdx.x] = r;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double norm = 0;
        for (int i = 0; i < k; i++) norm = fmax(norm, s_row[i]);
        int sq = 0;
        while (norm > 0.03125) { norm *= 0.5; sq++; }
        s_sq = sq;
        dd cn = dd_make(1.0, 0.0);
        s_coef[0] = cn;
        for (int n = 1; n <= EXPM_TERMS; n++) { cn = dd_div_d(cn, (double)n); s_coef[n] = cn; }
    }
    __syncthreads();
    const int sq = s_sq;
    for (int idx = threadIdx.x; idx < kk; idx += blockDim.x) X[idx] = dd_ldexp(X[idx], -sq);
    __syncthreads();
    dd_matmul(k, X, X, X2, strips);
    __syncthreads();
    dd_matmul(k, X2, X, X3, strips);
    dd_matmul(k, X2, X2, X4, strips);
    __syncthreads();
    /* cubic block i of the polynomial at entry idx */
    auto block = [&](int ib, int idx) -> dd {
        const int i = idx / k, j = idx - i * k;
        dd v = dd_add(dd_mul(s_coef[4 * ib + 1], X[idx]), dd_add(dd_mul(s_coef[4 * ib + 2], X2[idx]), dd_mul(s_coef[4 * ib + 3], X3[idx])));
        if (i == j) v = dd_add(v, s_coef[4 * ib]);
        return v;
    };
    for (int idx = threadIdx.x; idx < kk; idx += blockDim.x) O[idx] = dd_add(block(3, idx), dd_mul(s_coef[16], X4[idx]));
    __syncthreads();
    for (int ib = 2; ib >= 0; ib--) {
        dd_matmul(k, X4, O, W, strips);
        __syncthreads();
        for (int idx = threadIdx.x; idx < kk; idx += blockDim.x) O[idx] = dd_add(block(ib, idx), W[idx]);
        __syncthreads();
    }
    for (int q = 0; q < sq; q++) {
        dd_matmul(k, O, O, W, strips);
        __syncthreads();
        for (int idx = threadIdx.x; idx < kk; idx += blockDim.x) O[idx] = W[idx];
        __syncthreads();
    }
    const double rc = cat_rates[c];
    if (FRECHET) {
        for (int idx = threadIdx.x; idx < kks; idx += blockDim.x) {
            const int i = idx / ks, j = idx - i * ks;
            dd v = O[i * k + ks + j];
            if (coef_mode == PLK_COEF_PRIOR_RATE_EDGE) v = dd_mul(v, s);
            else if (coef_mode == PLK_COEF_PRIOR_RATE) v = dd_mul_d(v, rc);
            Fout[(size_t)ce * kks + idx] = v.hi;
        }
        return;
    }
    /* outputs: unrounded P, rounded P, rounded dP = r_c * Qn * P */
    for (int idx = threadIdx.x; idx < kk; idx += blockDim.x) {
        dd v = O[idx];
        if (v.hi < 0) v = dd_make(0.0, 0.0);
        Pdd[(size_t)ce * kk + idx] = v;
        P[(size_t)ce * kk + idx] = v.hi;
    }
    if (!post.skip_dP) {
        /* from the clamped matrix, as k_dP_dd computes it later from Pdd: one definition of dP whichever kernel wrote it */
        __syncthreads();
        dd_rate_product(k, Qn, Pdd + (size_t)ce * kk, rc, dP + (size_t)ce * kk, strips);
    }
    if (!FRECHET && post.edge_slot && k == 4) {
        const int mi = post.edge_slot[e], t = post.edge_slot[E + e];
        if (mi >= 0)
            for (int idx = threadIdx.x; idx < 16; idx += blockDim.x) {
                const int j = idx >> 2, i = idx & 3;
                const double v = O[i * 4 + j].hi;
                const size_t slot = (size_t)c * post.nmat1 + mi;
                post.PS[slot * 16 + idx] = v < 0 ? 0.0 : v;
            }
        if (t >= 0)
            for (int idx = threadIdx.x; idx < post.nchar * 4; idx += blockDim.x) {
                const int code = idx >> 2, i = idx & 3;
                const double *d = post.defs + code * 4;
                double out;
                if (d[0] == d[1] && d[0] == d[2] && d[0] == d[3]) out = d[0];      /* src/util.c:276-283 */
                else {
                    dd acc = dd_make(0.0, 0.0);
                    for (int j = 0; j < 4; j++) {
                        dd pij = O[i * 4 + j];
                        if (pij.hi < 0) pij = dd_make(0.0, 0.0);
                        acc = dd_add(acc, dd_mul_d(pij, d[j]));
                    }
                    out = acc.hi;
                }
                post.tip[(((size_t)c * post.ntips1 + t) * post.nchar + code) * 4 + i] = out;
            }
    }
}

/* dP = r_c Qn P for every (category, edge) from the stored double-double P: what K1 skips for an ll evaluation */
__global__ __launch_bounds__(1024) void k_dP_dd(int k, int E, const double *__restrict__ Qn, const double *__restrict__ cat_rates,
                                                const dd *__restrict__ Pdd, double *__restrict__ dP, int tiled)
{
    extern __shared__ double smem_raw[];
    const int ce = blockIdx.x, c = ce / E;
    const size_t kk = (size_t)k * k;
    dd_rate_product(k, Qn, Pdd + ce * kk, cat_rates[c], dP + ce * kk, tiled ? reinterpret_cast<dd *>(smem_raw) : nullptr);
}

/* PS[c][pc][j*K + i] = P[c][edge(pc)][i][j], zero padded to K (transposed so that
 * the column needed for one input state is contiguous for scalar loads) */
__global__ void k_build_stream(int k, int K, int E, int nops, const int *__restrict__ op_edge,
                               const double *__restrict__ P, double *__restrict__ PS)
{
    const int pc = blockIdx.x, c = blockIdx.y;
    const int e = op_edge[pc];
    double *dst = PS + ((size_t)c * nops + pc) * K * K;
    for (int idx = threadIdx.x; idx < K * K; idx += blockDim.x) {
        int j = idx / K, i = idx - j * K;
        double v = 0.0;
        if (e >= 0 && i < k && j < k) v = P[((size_t)c * E + e) * k * k + i * k + j];
        dst[idx] = v;
    }
}

/* tip[c][t][code][i] = sum_j P[c][edge(t)][i][j] * defs[code][j] in dd (k = 4).
 * An exactly constant definition row (e.g. all-ones "missing") maps to itself,
 * which is what the reference's exact shortcut produces (src/util.c:276-283). */
__global__ void k_build_tip(int E, int ntips, int nchar, const int *__restrict__ tip_edge,
                            const dd *__restrict__ Pdd, const double *__restrict__ defs /* [nchar][4] */,
                            double *__restrict__ tip)
{
    const int t = blockIdx.x, c = blockIdx.y;
    const int e = tip_edge[t];
    const dd *Pm = Pdd + ((size_t)c * E + (e < 0 ? 0 : e)) * 16;
    for (int idx = threadIdx.x; idx < nchar * 4; idx += blockDim.x) {
        int code = idx >> 2, i = idx & 3;
        const double *d = defs + code * 4;
        double out;
        if (e < 0) {
            out = d[i];                          /* pseudo slot: the definition itself */
        } else if (d[0] == d[1] && d[0] == d[2] && d[0] == d[3]) {
            out = d[0];
        } else {
            dd acc = dd_make(0.0, 0.0);
            for (int j = 0; j < 4; j++) acc = dd_add(acc, dd_mul_d(Pm[i * 4 + j], d[j]));
            out = acc.hi;
        }
        tip[(((size_t)c * ntips + t) * nchar + code) * 4 + i] = out;
    }
}

/* Tables of the pair-table interpreter (plk_fused4_asm.h: k_ll_fused4_asm_pt), one block per (table, category), from
 * the unrounded P in double-double.  tab[4][ntab]: first unit | edge (leaf edge of a single table, edge above the cherry
 * of a pair table, -1 pseudo) | pair: the two leaf edges, else -1.
 *   single  out[code][i]         = (P_e defs[code])_i
 *   pair    out[cb*nchar+cc][i]  = (P_a (P_b defs[cb] o P_c defs[cc]))_i    (src/evaluate_site_lhood.c:36-56 for a cherry)
 *   pseudo  out[code][i]         = defs[code][i]
 * A constant vector maps to itself exactly under a stochastic matrix, as the reference's shortcut does (src/util.c:276-283). */
__global__ void k_build_tables_pt(int E, int ntab, int units, int nchar, const int *__restrict__ tab,
                                  const dd *__restrict__ Pdd, const double *__restrict__ defs /* [nchar][4] */,
                                  double *__restrict__ tip)
{
    const int t = blockIdx.x, c = blockIdx.y;
    const int unit = tab[t], e = tab[ntab + t], eb = tab[2 * ntab + t], ec = tab[3 * ntab + t];
    double *out = tip + ((size_t)c * units + unit) * nchar * 4;
    auto leaf = [&](const dd *Pm, const double *d, dd (&v)[4]) {
        if (d[0] == d[1] && d[0] == d[2] && d[0] == d[3]) { for (int i = 0; i < 4; i++) v[i] = dd_make(d[0], 0.0); return; }
        for (int i = 0; i < 4; i++) {
            dd acc = dd_make(0.0, 0.0);
            for (int j = 0; j < 4; j++) acc = dd_add(acc, dd_mul_d(Pm[i * 4 + j], d[j]));
            v[i] = acc;
        }
    };
    if (eb < 0) {
        const dd *Pm = Pdd + ((size_t)c * E + (e < 0 ? 0 : e)) * 16;
        for (int code = threadIdx.x; code < nchar; code += blockDim.x) {
            const double *d = defs + code * 4;
            dd v[4];
            if (e < 0) { for (int i = 0; i < 4; i++) out[code * 4 + i] = d[i]; continue; }
            leaf(Pm, d, v);
            for (int i = 0; i < 4; i++) out[code * 4 + i] = v[i].hi;
        }
        return;
    }
    const dd *Pa = Pdd + ((size_t)c * E + e) * 16, *Pb = Pdd + ((size_t)c * E + eb) * 16, *Pc = Pdd + ((size_t)c * E + ec) * 16;
    for (int comb = threadIdx.x; comb < nchar * nchar; comb += blockDim.x) {
        const int cb = comb / nchar, cc = comb - cb * nchar;
        dd vb[4], vc[4], pr[4];
        leaf(Pb, defs + cb * 4, vb);
        leaf(Pc, defs + cc * 4, vc);
        bool cst = true;
        for (int j = 0; j < 4; j++) { pr[j] = dd_mul(vb[j], vc[j]); cst = cst && pr[j].hi == pr[0].hi && pr[j].lo == pr[0].lo; }
        for (int i = 0; i < 4; i++) {
            dd acc = pr[0];
            if (!cst) {
                acc = dd_make(0.0, 0.0);
                for (int j = 0; j < 4; j++) acc = dd_add(acc, dd_mul(Pa[i * 4 + j], pr[j]));
            }
            out[comb * 4 + i] = acc.hi;
        }
    }
}

/* flags[n] bit 0: some site's code at node n is not an all-ones definition; bit 1: some code is >= nchar
 * (the kernels index LDS and the tip tables with the codes: an out-of-range code must never reach them); bit 2: some
 * code's definition is not a single 1.0 among zeros (code_trivial[ch]: bit 0 all-ones definition, bit 1 single state) */
__global__ void k_node_flags(long S, long Spad, const uint8_t *__restrict__ codes, int nchar,
                             const int *__restrict__ code_trivial, int *__restrict__ flags)
{
    /* 16 codes per lane and iteration (rows are 1024-byte padded with code 0 beyond S;
     * the tail is masked), grid-stride over the row */
    const int n = blockIdx.y;
    const uint4 *row = reinterpret_cast<const uint4 *>(codes + (size_t)n * Spad);
    const long nvec = (S + 15) / 16;
    int bad = 0;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (long)gridDim.x * blockDim.x) {
        const uint4 q = row[v];
        const unsigned wds[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const long s = v * 16 + j;
            const int ch = (wds[j >> 2] >> ((j & 3) * 8)) & 0xff;
            if (s < S) { const int ct = ch >= nchar ? 0 : code_trivial[ch]; bad |= ch >= nchar ? 2 : ((ct & 1) ? 0 : 1) | ((ct & 2) ? 0 : 4); }
        }
    }
    const int wbad = (__any(bad & 1) ? 1 : 0) | (__any(bad & 2) ? 2 : 0) | (__any(bad & 4) ? 4 : 0);
    if (wbad && (threadIdx.x & 63) == 0) atomicOr(&flags[n], wbad);
}

/* ====================================================================== */
/* deterministic double-double reductions                                  */
/* ====================================================================== */

__device__ static inline dd dd_wave_sum(dd v)
{
    for (int off = 32; off > 0; off >>= 1) {
        dd o;
        o.hi = __shfl_down(v.hi, off, 64);
        o.lo = __shfl_down(v.lo, off, 64);
        v = dd_add(v, o);
    }
    return v;
}

/* block-wide dd sum, result valid in thread 0; blockDim.x multiple of 64, <= 1024 */
__device__ static inline dd dd_block_sum(dd v)
{
    __shared__ double sh_hi[16], sh_lo[16];
    v = dd_wave_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) { sh_hi[wave] = v.hi; sh_lo[wave] = v.lo; }
    __syncthreads();
    dd r = dd_make(0.0, 0.0);
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; i++) r = dd_add(r, dd_make(sh_hi[i], sh_lo[i]));
    }
    return r;
}

/* out[row] = sum over blocks of partial[row][block]; one block per row */
__global__ void k_dd_final(int nblocks, const dd *__restrict__ partial, dd *__restrict__ out)
{
    const int row = blockIdx.x;
    dd acc = dd_make(0.0, 0.0);
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x) acc = dd_add(acc, partial[(size_t)row * nblocks + b]);
    dd r = dd_block_sum(acc);
    if (threadIdx.x == 0) out[row] = r;
}

/* out[b] = sum of the b-th of gridDim.x contiguous slices of partial[0 .. n) */
__global__ void k_dd_slices(int n, const dd *__restrict__ partial, dd *__restrict__ out)
{
    const int per = (n + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    dd acc = dd_make(0.0, 0.0);
    for (int b = lo + threadIdx.x; b < hi; b += blockDim.x) acc = dd_add(acc, partial[b]);
    dd r = dd_block_sum(acc);
    if (threadIdx.x == 0) out[blockIdx.x] = r;
}

/* partial[row][block] = sum_{s in block's range} w[s] * X[row][s] */
__global__ void k_wsum_rows(long S, long row_stride, const double *__restrict__ X,
                            const double *__restrict__ w, int nblocks, dd *__restrict__ partial)
{
    const int row = blockIdx.y;
    const long per = (S + nblocks - 1) / nblocks;
    const long lo = (long)blockIdx.x * per;
    const long hi = lo + per < S ? lo + per : S;
    dd acc = dd_make(0.0, 0.0);
    for (long s = lo + threadIdx.x; s < hi; s += blockDim.x) {
        double x = X[(size_t)row * row_stride + s];
        dd t = w ? dd_two_prod(w[s], x) : dd_make(x, 0.0);
        acc = dd_add(acc, t);
    }
    dd r = dd_block_sum(acc);
    if (threadIdx.x == 0) partial[(size_t)row * nblocks + blockIdx.x] = r;
}

/* ---- site sums taken where the values are produced (site-aggregated marginals; src/arbplfmarginal.c:237-256) ----
 * Cross-lane sums by data-parallel-primitive moves: every lane of a row of 16 ends with the row's sum (quad swaps,
 * half-row mirror, row mirror), then row 3 collects the four rows (row broadcasts): lane 63 holds the sum of the 64
 * lanes.  All lanes must be active; lanes without a site contribute 0.  Plain fp64 within the wave (a pairwise tree of
 * 64 terms), double-double across waves afterwards (k_wsum_rows / k_dd_final). */
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_fetch(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double x)
{
    x += dpp_fetch<0xB1, 0xf>(x);      /* quad_perm [1,0,3,2] */
    x += dpp_fetch<0x4E, 0xf>(x);      /* quad_perm [2,3,0,1] */
    x += dpp_fetch<0x141, 0xf>(x);     /* row_half_mirror */
    x += dpp_fetch<0x140, 0xf>(x);     /* row_mirror */
    return x;
}
__device__ __forceinline__ double wave64_sum_lane63(double x)
{
    x = row16_sum(x);
    x += dpp_fetch<0x142, 0xa>(x);     /* row_bcast15 into rows 1 and 3 */
    x += dpp_fetch<0x143, 0xc>(x);     /* row_bcast31 into rows 2 and 3 */
    return x;
}

#include "plk_fused4.h"
#include "plk_fused4_asm.h"
#include "plk_fused4_v4.h"
#include "plk_mfma.h"
#include "plk_mfma_updown.h"
#include "plk_vec.h"
#include "plk_updown_vec.h"
#include "plk_updown4.h"

/* ====================================================================== */
/* K2+K3 generic: any k <= K, stack slots in HBM                           */
/* ====================================================================== */

struct GenArgs {
    long S, Spad;
    int k, C, nops, nchar, pat_mode, root_mode;
    const int2 *ops;
    const double *PS;        /* [C][nops][K*K] transposed, zero padded */
    const uint8_t *codes;    /* [N][Spad] */
    const double *defs;      /* [nchar][K] zero padded */
    const double *B;         /* [N][k][S] */
    const double *cat_prior, *root_w, *w;
    double *slots;           /* [nslots][k][S] */
    double *site_ll;
    dd *partial;
};

#define GEN_BLOCK 64

template <int K>
__device__ static inline void gen_load_obs(const GenArgs &a, int node, long sc, int tid, double (*xs)[GEN_BLOCK])
{
    if (a.pat_mode == 1) {
        const int ch = a.codes[(size_t)node * a.Spad + sc];
        const double *dv = a.defs + (size_t)ch * K;
        for (int j = 0; j < a.k; j++) xs[j][tid] = dv[j];
    } else {
        const double *bp = a.B + (size_t)node * a.k * a.S + sc;
        for (int j = 0; j < a.k; j++) xs[j][tid] = bp[(size_t)j * a.S];
    }
}

template <int K>
__global__ __launch_bounds__(GEN_BLOCK) void k_ll_generic(GenArgs a)
{
    __shared__ double xs[K][GEN_BLOCK];
    const int tid = threadIdx.x;
    const long s = (long)blockIdx.x * GEN_BLOCK + tid;
    const bool valid = s < a.S;
    const long sc = valid ? s : a.S - 1;

    double sum = 0.0;
    int Eexp = 0;
    bool have = false;

    for (int c = 0; c < a.C; c++) {
        double cur[K];
#pragma unroll
        for (int i = 0; i < K; i++) cur[i] = 1.0;
        int esc = 0;
        const double *PSc = a.PS + (size_t)c * a.nops * K * K;
        for (int pc = 0; pc < a.nops; pc++) {
            int2 op;
            op.x = as_uniform(reinterpret_cast<const int *>(a.ops))[2 * pc];
            op.y = as_uniform(reinterpret_cast<const int *>(a.ops))[2 * pc + 1];
            const int code = op.x & 0xff;
            if (code == OP_MATVEC || code == OP_TIP_SET || code == OP_TIP_MUL) {
                if (code == OP_MATVEC) {
#pragma unroll
                    for (int j = 0; j < K; j++) xs[j][tid] = cur[j];
                } else {
                    gen_load_obs<K>(a, op.y, sc, tid, xs);
                }
                const double *M = PSc + (size_t)pc * K * K;
                double acc[K];
#pragma unroll
                for (int i = 0; i < K; i++) acc[i] = 0.0;
                for (int j = 0; j < a.k; j++) {
                    const double x = xs[j][tid];
                    const double *col = M + j * K;
#pragma unroll
                    for (int i = 0; i < K; i++) acc[i] = fma(col[i], x, acc[i]);
                }
                if (code == OP_TIP_MUL) {
#pragma unroll
                    for (int i = 0; i < K; i++) cur[i] *= acc[i];
                } else {
#pragma unroll
                    for (int i = 0; i < K; i++) cur[i] = acc[i];
                }
            } else if (code == OP_PUSH) {
                double *sp = a.slots + (size_t)op.y * a.k * a.S + sc;
#pragma unroll
                for (int i = 0; i < K; i++)
                    if (i < a.k && valid) sp[(size_t)i * a.S] = cur[i];
            } else if (code == OP_POPMUL) {
                const double *sp = a.slots + (size_t)op.y * a.k * a.S + sc;
#pragma unroll
                for (int i = 0; i < K; i++)
                    if (i < a.k) cur[i] *= valid ? sp[(size_t)i * a.S] : 1.0;
            } else if (code == OP_NODE_MUL) {
                gen_load_obs<K>(a, op.y, sc, tid, xs);
#pragma unroll
                for (int i = 0; i < K; i++)
                    if (i < a.k) cur[i] *= xs[i][tid];
            } else if (code == OP_SCALE) {
                double m = 0.0;
#pragma unroll
                for (int i = 0; i < K; i++) m = fmax(m, cur[i]);
                const int e = frexp_exp(m);
#pragma unroll
                for (int i = 0; i < K; i++) cur[i] = ldexp(cur[i], -e);
                esc += e;
            }
        }
        double lh = 0.0;
        if (a.root_mode == PLK_ROOT_NONE || a.root_mode == PLK_ROOT_UNIFORM) {
#pragma unroll
            for (int i = 0; i < K; i++)
                if (i < a.k) lh += cur[i];
            if (a.root_mode == PLK_ROOT_UNIFORM) lh /= (double)a.k;
        } else {
#pragma unroll
            for (int i = 0; i < K; i++) lh = fma(a.root_w[i], cur[i], lh);
        }
        const double term = a.cat_prior[c] * lh;
        if (term != 0.0) {
            if (!have) { sum = term; Eexp = esc; have = true; }
            else if (esc > Eexp) { sum = ldexp(sum, Eexp - esc) + term; Eexp = esc; }
            else sum += ldexp(term, esc - Eexp);
        }
    }
    const double ll = have ? log(sum) + (double)Eexp * 0.6931471805599453094 : -INFINITY;
    if (valid && a.site_ll) a.site_ll[s] = ll;
    if (a.partial) {
        dd v = dd_make(0.0, 0.0);
        if (valid) v = a.w ? dd_two_prod(a.w[s], ll) : dd_make(ll, 0.0);
        dd r = dd_block_sum(v);
        if (tid == 0) a.partial[blockIdx.x] = r;
    }
}

/* ====================================================================== */
/* K2 with stored vectors + K4/K5/K6 (up pass): deriv and marginal         */
/* ====================================================================== */

struct UpArgs {
    long S, Spad;        /* full pattern extent (row pitch of codes / B) */
    long s0, n;          /* chunk [s0, s0+n) */
    int N, E, k, C, nchar, pat_mode, root_mode;
    int dzero;           /* 1: the edge-form matrices have zero row sums (dP): exact zero for constant inputs */
    const int *indptr, *indices, *preorder;
    const int *node_has_data;  /* N */
    const double *PT;    /* [C][E][K*K] transposed P:  PT[j*K+i] = P[i][j]  */
    const double *PN;    /* [C][E][K*K] plain P:       PN[i*K+j] = P[i][j]  */
    const double *DT;    /* [C][E][K*K] transposed dP: DT[j*K+i] = dP[i][j] */
    /* second-order passes (plk_hess): edge mod_edge uses dP in the role of P and d2P = r^2 Q Q P in the
     * role of dP; DN = plain dP, D2T = transposed d2P; LHdiv = likelihoods of the unmodified model */
    int mod_edge;        /* -1: none */
    const double *DN, *D2T, *LHdiv;
    /* exact power-of-two rescaling (deriv / marginal / edge expectations; off for the Hessian passes):
     * node_scale[N] = slot or -1 (null = no rescaling), SC[(slot*C + c)][n] = 2^-e, CW[C][n] = 2^(X_c - Xmax) */
    const int *node_scale;
    double *SC, *CW, *XC;
    double *XM;          /* [n] or null: the common exponent Xmax of the site (LH and the edge forms are at 2^Xmax) */
    const double *XMdiv; /* [n] or null: Xmax of the model LHdiv belongs to (second-order passes) */
    const uint8_t *codes;
    const double *defs;  /* [nchar][K] */
    const double *B;     /* [N][k][S] */
    const double *cat_prior, *root_w;
    const int *edge_mask, *node_mask;   /* device, may be null = all */
    double *EV;          /* [E][C][k][n] */
    double *LN;          /* [N][C][k][n] (internal nodes only are written) */
    double *FN;          /* [N][C][k][n] */
    double *LH;          /* [n] site likelihood */
    double *DV;          /* [E][n] derivative values */
    double *MV;          /* [N][k][n] marginal values */
};

template <int K>
__device__ static inline void up_load_obs_reg(const UpArgs &a, int node, long sg, double (&out)[K])
{
    if (a.pat_mode == 1) {
        const int ch = a.codes[(size_t)node * a.Spad + sg];
        const double *dv = a.defs + (size_t)ch * K;
#pragma unroll
        for (int i = 0; i < K; i++) out[i] = dv[i];
    } else {
        const double *bp = a.B + (size_t)node * a.k * a.S + sg;
#pragma unroll
        for (int i = 0; i < K; i++) out[i] = i < a.k ? bp[(size_t)i * a.S] : 0.0;
    }
}

template <int K>
__device__ static inline void up_stage_obs(const UpArgs &a, int node, long sg, int tid, double (*xs)[GEN_BLOCK])
{
    if (a.pat_mode == 1) {
        const int ch = a.codes[(size_t)node * a.Spad + sg];
        const double *dv = a.defs + (size_t)ch * K;
        for (int j = 0; j < a.k; j++) xs[j][tid] = dv[j];
    } else {
        const double *bp = a.B + (size_t)node * a.k * a.S + sg;
        for (int j = 0; j < a.k; j++) xs[j][tid] = bp[(size_t)j * a.S];
    }
}

/* acc[i] = sum_j M[j*K+i] * xs[j]  (M uniform, zero padded).
 * CONST_MODE mirrors the reference's exact shortcut for a constant input column
 * (src/util.c:276-283, :338-345; src/arb_mat_extras.c:36-51): a stochastic matrix maps
 * it to itself (1), a rate matrix with zero row sums maps it to zero (2). */
template <int K, int CONST_MODE>
__device__ static inline void up_matvec(const double *M, int k, double (*xs)[GEN_BLOCK], int tid, double (&acc)[K])
{
#pragma unroll
    for (int i = 0; i < K; i++) acc[i] = 0.0;
    const double x0 = xs[0][tid];
    bool is_const = true;
    for (int j = 0; j < k; j++) {
        const double x = xs[j][tid];
        is_const = is_const && (x == x0);
        const double *col = M + j * K;
#pragma unroll
        for (int i = 0; i < K; i++) acc[i] = fma(col[i], x, acc[i]);
    }
    if (CONST_MODE != 0 && is_const) {
#pragma unroll
        for (int i = 0; i < K; i++) acc[i] = (CONST_MODE == 1 && i < k) ? x0 : 0.0;
    }
}

/* down pass storing every edge vector and internal node vector (no rescaling) */
template <int K>
__global__ __launch_bounds__(GEN_BLOCK) void k_down_store(UpArgs a)
{
    __shared__ double xs[K][GEN_BLOCK];
    const int tid = threadIdx.x;
    const long sl = (long)blockIdx.x * GEN_BLOCK + tid;   /* local site */
    const bool valid = sl < a.n;
    const long slc = valid ? sl : a.n - 1;
    const long sg = a.s0 + slc;
    const size_t n = (size_t)a.n;
    double lh_total = 0.0;
    int xmax = INT_MIN, xall = INT_MIN;
    for (int c = 0; c < a.C; c++) {
        double lh_c = 0.0;
        int X = 0;
        for (int u = a.N - 1; u >= 0; u--) {
            const int nd = as_uniform(a.preorder)[u];
            const int start = as_uniform(a.indptr)[nd], stop = as_uniform(a.indptr)[nd + 1];
            if (start == stop) continue;
            double acc[K];
            if (as_uniform(a.node_has_data)[nd]) up_load_obs_reg<K>(a, nd, sg, acc);
            else {
#pragma unroll
                for (int i = 0; i < K; i++) acc[i] = 1.0;
            }
            for (int idx = start; idx < stop; idx++) {
                const int b = as_uniform(a.indices)[idx];
                const bool b_leaf = as_uniform(a.indptr)[b] == as_uniform(a.indptr)[b + 1];
                if (b_leaf) up_stage_obs<K>(a, b, sg, tid, xs);
                else {
                    const double *lb = a.LN + ((size_t)b * a.C + c) * a.k * n + slc;
                    for (int j = 0; j < a.k; j++) xs[j][tid] = lb[(size_t)j * n];
                }
                double m[K];
                if (idx == a.mod_edge) up_matvec<K, 2>(a.DT + ((size_t)c * a.E + idx) * K * K, a.k, xs, tid, m);
                else up_matvec<K, 1>(a.PT + ((size_t)c * a.E + idx) * K * K, a.k, xs, tid, m);
                /* leaf-edge vectors are not stored: the up pass recomputes them (k^2 flops vs 2k doubles of HBM) */
                double *ev = a.EV + ((size_t)idx * a.C + c) * a.k * n + slc;
#pragma unroll
                for (int i = 0; i < K; i++) {
                    if (!b_leaf && i < a.k && valid) ev[(size_t)i * n] = m[i];
                    acc[i] *= m[i];
                }
            }
            const int slot = a.node_scale ? as_uniform(a.node_scale)[nd] : -1;
            if (slot >= 0) {
                double mx = 0.0;       /* |.|: the vectors of an edge-modified model (plk_hess) are not sign definite */
#pragma unroll
                for (int i = 0; i < K; i++) mx = fmax(mx, fabs(acc[i]));
                double sc = 1.0;
                if (mx > 0x1p-1000 && mx < 0x1p+1000) {
                    const int e = ilogb(mx);
                    sc = ldexp(1.0, -e);
#pragma unroll
                    for (int i = 0; i < K; i++) acc[i] *= sc;
                    X += e;
                }
                if (valid) a.SC[((size_t)slot * a.C + c) * n + sl] = sc;
            }
            double *ln = a.LN + ((size_t)nd * a.C + c) * a.k * n + slc;
#pragma unroll
            for (int i = 0; i < K; i++)
                if (i < a.k && valid) ln[(size_t)i * n] = acc[i];
            if (u == 0) {
                if (a.root_mode == PLK_ROOT_NONE || a.root_mode == PLK_ROOT_UNIFORM) {
#pragma unroll
                    for (int i = 0; i < K; i++)
                        if (i < a.k) lh_c += acc[i];
                    if (a.root_mode == PLK_ROOT_UNIFORM) lh_c /= (double)a.k;
                } else {
#pragma unroll
                    for (int i = 0; i < K; i++) lh_c = fma(as_uniform(a.root_w)[i], acc[i], lh_c);
                }
            }
        }
        if (!a.node_scale) { lh_total = fma(as_uniform(a.cat_prior)[c], lh_c, lh_total); continue; }
        lh_c *= as_uniform(a.cat_prior)[c];
        if (lh_c != 0.0 && X > xmax) xmax = X;
        if (X > xall) xall = X;
        if (valid) { a.XC[(size_t)c * n + sl] = (double)X; a.CW[(size_t)c * n + sl] = lh_c; }
    }
    if (a.node_scale && valid) {
        /* combine the categories at the largest exponent: LH = sum_c prior_c lh_c 2^(X_c - Xmax).  When every term is
         * zero (an edge-modified model of plk_hess at a site that does not depend on the edge) the exponent of the
         * vectors themselves is kept, so that the division by the unmodified likelihood stays finite */
        if (xmax == INT_MIN) xmax = xall == INT_MIN ? 0 : xall;
        if (a.XM) a.XM[sl] = (double)xmax;
        for (int c = 0; c < a.C; c++) {
            const double w = ldexp(1.0, (int)a.XC[(size_t)c * n + sl] - xmax);
            lh_total = fma(a.CW[(size_t)c * n + sl], w, lh_total);
            a.CW[(size_t)c * n + sl] = w;
        }
    }
    if (valid) a.LH[sl] = lh_total;
}

/*
 * up pass in BFS order.  For every edge e = (a -> b):
 *   fe   = F_a o B_a o prod_{siblings j != e} Ev_j          (src/evaluate_site_forward.c:69-94)
 *   d_e  = sum_c prior_c * fe . (dP_{c,e} L_b) / lhood       (dP = r_c Q P; equals the reference's
 *          rate * rootward recomputation with Q Ev_e, src/arbplfderiv.c:112-207, :312-342)
 *   F_b  = P_e^T fe                                          (src/evaluate_site_forward.c:97-100)
 *   m_b  = sum_c prior_c * F_b o L_b / lhood                 (src/arbplfmarginal.c:206-234)
 */
template <int K, bool DERIV, bool MARG>
__global__ __launch_bounds__(GEN_BLOCK) void k_up(UpArgs a)
{
    __shared__ double xs[K][GEN_BLOCK];
    const int tid = threadIdx.x;
    const long sl = (long)blockIdx.x * GEN_BLOCK + tid;
    const bool valid = sl < a.n;
    const long slc = valid ? sl : a.n - 1;
    const long sg = a.s0 + slc;
    const size_t n = (size_t)a.n;
    /* second-order passes divide by the likelihood of the unmodified model, which sits at its own exponent */
    double inv = 1.0 / (a.LHdiv ? a.LHdiv[slc] : a.LH[slc]);
    if (a.LHdiv && a.XM && a.XMdiv) inv = ldexp(inv, (int)(a.XM[slc] - a.XMdiv[slc]));
    const int root = as_uniform(a.preorder)[0];

    /* root: forward vector = root prior weights; its marginal */
    {
        double macc[K];
#pragma unroll
        for (int i = 0; i < K; i++) macc[i] = 0.0;
        for (int c = 0; c < a.C; c++) {
            double *fr = a.FN + ((size_t)root * a.C + c) * a.k * n + slc;
            const double *lr = a.LN + ((size_t)root * a.C + c) * a.k * n + slc;
#pragma unroll
            for (int i = 0; i < K; i++) {
                if (i < a.k) {
                    if (valid) fr[(size_t)i * n] = as_uniform(a.root_w)[i];
                    if (MARG) macc[i] = fma(as_uniform(a.cat_prior)[c] * (a.CW ? a.CW[(size_t)c * n + slc] : 1.0) * as_uniform(a.root_w)[i], lr[(size_t)i * n], macc[i]);
                }
            }
        }
        if (MARG && (!a.node_mask || as_uniform(a.node_mask)[root])) {
            double *mv = a.MV + (size_t)root * a.k * n + slc;
#pragma unroll
            for (int i = 0; i < K; i++)
                if (i < a.k && valid) mv[(size_t)i * n] = macc[i] * inv;
        }
    }

    for (int u = 0; u < a.N; u++) {
        const int nd = as_uniform(a.preorder)[u];
        const int start = as_uniform(a.indptr)[nd], stop = as_uniform(a.indptr)[nd + 1];
        if (start == stop) continue;
        double bnd[K];
        const bool has = as_uniform(a.node_has_data)[nd];
        if (has) up_load_obs_reg<K>(a, nd, sg, bnd);
        const int slot = a.node_scale ? as_uniform(a.node_scale)[nd] : -1;
        for (int idx = start; idx < stop; idx++) {
            const int b = as_uniform(a.indices)[idx];
            const bool b_leaf = as_uniform(a.indptr)[b] == as_uniform(a.indptr)[b + 1];
            const bool want_d = DERIV && (!a.edge_mask || as_uniform(a.edge_mask)[idx]);
            const bool want_m = MARG && (!a.node_mask || as_uniform(a.node_mask)[b]);
            const bool want_f = !b_leaf || want_m;
            if (!want_d && !want_f) continue;
            double dsum = 0.0;
            double macc[K];
#pragma unroll
            for (int i = 0; i < K; i++) macc[i] = 0.0;
            for (int c = 0; c < a.C; c++) {
                double fe[K];
                const double *fa = a.FN + ((size_t)nd * a.C + c) * a.k * n + slc;
#pragma unroll
                for (int i = 0; i < K; i++) fe[i] = i < a.k ? fa[(size_t)i * n] : 0.0;
                if (has) {
#pragma unroll
                    for (int i = 0; i < K; i++) fe[i] *= bnd[i];
                }
                if (slot >= 0) {
                    const double sc = a.SC[((size_t)slot * a.C + c) * n + slc];
#pragma unroll
                    for (int i = 0; i < K; i++) fe[i] *= sc;
                }
                for (int idx2 = start; idx2 < stop; idx2++) {
                    if (idx2 == idx) continue;
                    const int b2 = as_uniform(a.indices)[idx2];
                    if (as_uniform(a.indptr)[b2] == as_uniform(a.indptr)[b2 + 1]) {
                        double m2[K];
                        up_stage_obs<K>(a, b2, sg, tid, xs);
                        if (idx2 == a.mod_edge) up_matvec<K, 2>(a.DT + ((size_t)c * a.E + idx2) * K * K, a.k, xs, tid, m2);
                        else up_matvec<K, 1>(a.PT + ((size_t)c * a.E + idx2) * K * K, a.k, xs, tid, m2);
#pragma unroll
                        for (int i = 0; i < K; i++) fe[i] *= m2[i];
                    } else {
                        const double *ev = a.EV + ((size_t)idx2 * a.C + c) * a.k * n + slc;
#pragma unroll
                        for (int i = 0; i < K; i++)
                            if (i < a.k) fe[i] *= ev[(size_t)i * n];
                    }
                }
                const double prior = as_uniform(a.cat_prior)[c] * (a.CW ? a.CW[(size_t)c * n + slc] : 1.0);
                if (want_d) {
                    /* y = dP_e * L_b ; d = fe . y */
                    if (b_leaf) up_stage_obs<K>(a, b, sg, tid, xs);
                    else {
                        const double *lb = a.LN + ((size_t)b * a.C + c) * a.k * n + slc;
                        for (int j = 0; j < a.k; j++) xs[j][tid] = lb[(size_t)j * n];
                    }
                    double y[K];
                    if (idx == a.mod_edge) up_matvec<K, 2>(a.D2T + ((size_t)c * a.E + idx) * K * K, a.k, xs, tid, y);
                    else if (a.dzero) up_matvec<K, 2>(a.DT + ((size_t)c * a.E + idx) * K * K, a.k, xs, tid, y);
                    else up_matvec<K, 0>(a.DT + ((size_t)c * a.E + idx) * K * K, a.k, xs, tid, y);
                    double d = 0.0;
#pragma unroll
                    for (int i = 0; i < K; i++) d = fma(fe[i], y[i], d);
                    dsum = fma(prior, d, dsum);
                }
                if (want_f) {
                    /* F_b[j] = sum_i P[i][j] fe[i]: stage fe, use the plain (row-major) P as the "transposed" operand */
#pragma unroll
                    for (int i = 0; i < K; i++) xs[i][tid] = fe[i];
                    double fb[K];
                    up_matvec<K, 0>((idx == a.mod_edge ? a.DN : a.PN) + ((size_t)c * a.E + idx) * K * K, a.k, xs, tid, fb);
                    if (!b_leaf) {
                        double *fo = a.FN + ((size_t)b * a.C + c) * a.k * n + slc;
#pragma unroll
                        for (int i = 0; i < K; i++)
                            if (i < a.k && valid) fo[(size_t)i * n] = fb[i];
                    }
                    if (want_m) {
                        if (b_leaf) {
                            double lb[K];
                            up_load_obs_reg<K>(a, b, sg, lb);
#pragma unroll
                            for (int i = 0; i < K; i++) macc[i] = fma(prior * fb[i], lb[i], macc[i]);
                        } else {
                            const double *lb = a.LN + ((size_t)b * a.C + c) * a.k * n + slc;
#pragma unroll
                            for (int i = 0; i < K; i++)
                                if (i < a.k) macc[i] = fma(prior * fb[i], lb[(size_t)i * n], macc[i]);
                        }
                    }
                }
            }
            if (want_d && valid) a.DV[(size_t)idx * n + sl] = dsum * inv;
            if (want_m) {
                double *mv = a.MV + (size_t)b * a.k * n + slc;
#pragma unroll
                for (int i = 0; i < K; i++)
                    if (i < a.k && valid) mv[(size_t)i * n] = macc[i] * inv;
            }
        }
    }
}

/* padded edge-indexed matrix streams for the up/down kernels:
 * mode 0: out[j*K+i] = M[i][j] (transposed), mode 1: out[i*K+j] = M[i][j] */
__global__ void k_build_edge_stream(int k, int K, int mode, const double *__restrict__ M, double *__restrict__ out)
{
    const size_t ce = blockIdx.x;
    const double *src = M + ce * k * k;
    double *dst = out + ce * K * K;
    for (int idx = threadIdx.x; idx < K * K; idx += blockDim.x) {
        int r = idx / K, q = idx - r * K;
        double v = 0.0;
        if (r < k && q < k) v = mode == 0 ? src[q * k + r] : src[r * k + q];
        dst[idx] = v;
    }
}

/* ====================================================================== */
/* host side                                                               */
/* ====================================================================== */

static int pad_K(int k)
{
    const int ks[] = {2, 4, 8, 16, 20, 32, 61, 64};
    for (int v : ks) if (k <= v) return v;
    return -1;
}

extern "C" const char *plk_create_error(void) { return g_create_error.c_str(); }

extern "C" int plk_create(plk_engine **out, int device)
{
    if (!out) return PLK_E_ARG;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_error = std::string("plk_create: no HIP device available (") +
                         (e != hipSuccess ? hipGetErrorString(e) : "device count 0") +
                         "); this engine has no CPU fallback";
        return PLK_E_DEVICE;
    }
    if (device < 0 || device >= ndev) {
        g_create_error = "plk_create: device index out of range";
        return PLK_E_ARG;
    }
    e = hipSetDevice(device);
    if (e != hipSuccess) {
        g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
        return PLK_E_DEVICE;
    }
    plk_engine *h = new plk_engine();
    h->device = device;
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) h->num_cus = cus; }
    /* ARBPLF_UP_NODES = 1: initial value of PLK_OPT_UP_NODES, so that the JSON drivers (which own their engines) can be
     * run with the node-visit up pass too: tests/test_gpu_differential.py compares the two */
    if (const char *v = getenv("ARBPLF_UP_NODES")) h->opt_up_nodes = atol(v);
    if (hipStreamCreate(&h->own_stream) != hipSuccess ||
        hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess ||
        hipEventCreate(&h->ev2) != hipSuccess || hipEventCreate(&h->ev3) != hipSuccess) {
        g_create_error = "plk_create: stream/event creation failed";
        delete h;
        return PLK_E_DEVICE;
    }
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        if (!g_atexit_set) {
            g_atexit_set = true;
            atexit([]() { std::lock_guard<std::mutex> lk2(g_live_mu); g_exiting = true; g_live.clear(); });
        }
        g_live.insert(h);
    }
    h->stream = h->own_stream;
    for (int i = 0; i < 64; i++)
        if (hipEventCreate(&h->evk[i][0]) != hipSuccess || hipEventCreate(&h->evk[i][1]) != hipSuccess) {
            g_create_error = "plk_create: event creation failed";
            plk_destroy(h);
            return PLK_E_DEVICE;
        }
    *out = h;
    return PLK_OK;
}

extern "C" void plk_destroy(plk_engine *h)
{
    if (!h) return;
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        if (g_exiting || !g_live.erase(h)) return;
    }
    (void)hipSetDevice(h->device);
    if (h->comm && g_rccl.CommDestroy) { (void)hipStreamSynchronize(h->stream); (void)g_rccl.CommDestroy(h->comm); h->comm = nullptr; }
    void *ptrs[] = {h->d_indptr, h->d_indices, h->d_preorder, h->d_Qn, h->d_edge_rates, h->d_cat_rates,
                    h->d_cat_prior, h->d_root_w, h->d_Pdd, h->d_P, h->d_dP, h->d_scratch, h->d_codes,
                    h->d_defs, h->d_B, h->d_w, h->d_ops, h->d_fops, h->d_words, h->d_mat_edge, h->d_edge_slot, h->d_op_edge, h->d_tip_edge, h->d_obs_nodes,
                    h->d_words_pt, h->d_row_nodes, h->d_tabs, h->d_PS, h->d_tip, h->d_frag, h->d_root_wd, h->d_mops, h->d_u4pack, h->d_u4tip, h->d_uvmat, h->d_stage, h->d_exL, h->d_exF, h->d_exmask, h->d_exscr, h->d_slots, h->d_site_ll, h->d_partial, h->d_work};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev2) (void)hipEventDestroy(h->ev2);
    if (h->ev3) (void)hipEventDestroy(h->ev3);
    for (int i = 0; i < 64; i++) for (int j = 0; j < 2; j++) if (h->evk[i][j]) (void)hipEventDestroy(h->evk[i][j]);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

extern "C" const char *plk_last_error(const plk_engine *h) { return plk_live(h) ? h->err.c_str() : "invalid engine handle"; }

extern "C" int plk_set_option(plk_engine *h, int option, long value)
{
    if (!plk_live(h)) return PLK_E_ARG;
    if (option == PLK_OPT_FORCE_GENERIC) { h->opt_force_generic = value; h->prog_dirty = true; return PLK_OK; }
    if (option == PLK_OPT_SITE_CHUNK) { h->opt_site_chunk = value; return PLK_OK; }
    if (option == PLK_OPT_FUSED_SITES_PER_LANE) { h->opt_fused_ns = value; h->fmt_dirty = true; return PLK_OK; }
    if (option == PLK_OPT_FUSED_ASM) { h->opt_fused_asm = value; h->fmt_dirty = true; return PLK_OK; }
    if (option == PLK_OPT_UP_NODES) { h->opt_up_nodes = value; return PLK_OK; }
    if (option == PLK_OPT_MFMA) { h->opt_mfma = value; return PLK_OK; }
    if (option == PLK_OPT_MFMA_NS2) { h->opt_mfma_ns2 = value; return PLK_OK; }
    if (option == PLK_OPT_VEC_REG_STACK) { h->opt_vec_reg_stack = value; return PLK_OK; }
    if (option == PLK_OPT_PAIR_TABLES) { h->opt_pair_tables = value; h->fmt_dirty = true; return PLK_OK; }
    h->err = "plk_set_option: unknown option";
    return PLK_E_ARG;
}

/* read the event pair of ring slot i (waits for it) into the running sum */
static int evk_resolve(plk_engine *h, int i)
{
    if (!h->evk_pending[i]) return PLK_OK;
    float ms = 0;
    HIPCHK(h, hipEventSynchronize(h->evk[i][1]));
    HIPCHK(h, hipEventElapsedTime(&ms, h->evk[i][0], h->evk[i][1]));
    h->evk_pending[i] = false;
    h->info_ll_kernel_ns = (long)(ms * 1e6);
    h->evk_sum_ns += h->info_ll_kernel_ns;
    h->evk_count++;
    return PLK_OK;
}

extern "C" int plk_get_info(plk_engine *h, int what, long *out)
{
    if (!plk_live(h) || !out) return PLK_E_ARG;
    if (what >= PLK_INFO_LAST_LL_KERNEL_NS && what <= PLK_INFO_LL_KERNEL_COUNT) {
        HIPCHK(h, hipSetDevice(h->device));
        int rc;
        for (int j = 0; j < 64; j++) if ((rc = evk_resolve(h, (h->evk_next + j) & 63))) return rc;     /* oldest first */
        if (h->info_pending) {
            float ms_t = 0;
            HIPCHK(h, hipEventSynchronize(h->ev3));
            HIPCHK(h, hipEventElapsedTime(&ms_t, h->ev0, h->ev3));
            h->info_ll_total_ns = (long)(ms_t * 1e6);
            h->info_pending = false;
        }
    }
    switch (what) {
    case PLK_INFO_LL_KERNEL_NS_SUM: *out = h->evk_sum_ns; h->evk_sum_ns = 0; return PLK_OK;
    case PLK_INFO_LL_KERNEL_COUNT: *out = h->evk_count; h->evk_count = 0; return PLK_OK;
    case PLK_INFO_LL_KERNEL: *out = h->info_ll_kernel; return PLK_OK;
    case PLK_INFO_LL_VARIANT: *out = h->info_ll_variant; return PLK_OK;
    case PLK_INFO_LL_EXEC_FLOPS: *out = h->info_ll_exec_flops; return PLK_OK;
    case PLK_INFO_PAIR_TABLES: *out = !h->fmt_dirty && ((h->fmt_pt && h->fmt_kind == 1) || (h->vec_pt && h->fmt_kind == 4)) ? h->fpt.npairs : 0; return PLK_OK;
    case PLK_INFO_STACK_SLOTS: *out = h->slots_needed; return PLK_OK;
    case PLK_INFO_PROGRAM_OPS: *out = (long)h->ops.size(); return PLK_OK;
    case PLK_INFO_LAST_LL_KERNEL_NS: *out = h->info_ll_kernel_ns; return PLK_OK;
    case PLK_INFO_LAST_LL_TOTAL_NS: *out = h->info_ll_total_ns; return PLK_OK;
    }
    h->err = "plk_get_info: unknown item";
    return PLK_E_ARG;
}

extern "C" int plk_set_tree(plk_engine *h, int N, const int *indptr, const int *indices, const int *preorder)
{
    if (!plk_live(h)) return PLK_E_ARG;
    if (N < 2 || !indptr || !indices || !preorder) { h->err = "plk_set_tree: bad arguments"; return PLK_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    const int E = N - 1;
    if (indptr[0] != 0 || indptr[N] != E) { h->err = "plk_set_tree: indptr does not describe N-1 edges"; return PLK_E_ARG; }
    std::vector<int> b2i(N, -1), i2a(E, -1);
    for (int a = 0; a < N; a++) {
        if (indptr[a + 1] < indptr[a]) { h->err = "plk_set_tree: indptr not monotone"; return PLK_E_ARG; }
        for (int idx = indptr[a]; idx < indptr[a + 1]; idx++) {
            int b = indices[idx];
            if (b < 0 || b >= N || b == a || b2i[b] != -1) { h->err = "plk_set_tree: not a tree"; return PLK_E_ARG; }
            b2i[b] = idx;
            i2a[idx] = a;
        }
    }
    std::vector<char> seen(N, 0);
    for (int u = 0; u < N; u++) {
        int a = preorder[u];
        if (a < 0 || a >= N || seen[a]) { h->err = "plk_set_tree: preorder is not a permutation"; return PLK_E_ARG; }
        if (u == 0 ? b2i[a] != -1 : (b2i[a] == -1 || !seen[i2a[b2i[a]]])) {
            h->err = "plk_set_tree: preorder does not start at the root / parents first";
            return PLK_E_ARG;
        }
        seen[a] = 1;
    }
    h->N = N; h->E = E;
    h->indptr.assign(indptr, indptr + N + 1);
    h->indices.assign(indices, indices + E);
    h->preorder.assign(preorder, preorder + N);
    h->b_to_idx = b2i; h->idx_to_a = i2a;
    int rc;
    if ((rc = dev_upload(h, &h->d_indptr, indptr, (size_t)N + 1))) return rc;
    if ((rc = dev_upload(h, &h->d_indices, indices, (size_t)E))) return rc;
    if ((rc = dev_upload(h, &h->d_preorder, preorder, (size_t)N))) return rc;
    h->prog_dirty = true;
    h->model_dirty = true;
    h->pat_mode = 0;
    h->k = 0;
    return PLK_OK;
}

static int run_expm(plk_engine *h, bool post = false, bool need_dP = true)
{
    const int k = h->k, C = h->C, E = h->E;
    const size_t kk = (size_t)k * k, n = (size_t)C * E * kk;
    int rc;
    if ((rc = dev_reserve(h, &h->d_Pdd, &h->pdd_cap, n))) return rc;
    if ((rc = dev_reserve(h, &h->d_P, &h->p_cap, n))) return rc;
    if ((rc = dev_reserve(h, &h->d_dP, &h->dp_cap, n))) return rc;
    /* one thread per few matrix entries: the dd matrix products are the whole cost for k = 61 */
    const int threads = kk >= 1024 ? 1024 : (kk >= 256 ? 256 : 64);
    int use_lds;
    const size_t lds_bytes = expm_lds_bytes((size_t)k, threads, &use_lds);
    if (!use_lds) { if ((rc = dev_reserve(h, &h->d_scratch, &h->scratch_cap, (size_t)C * E * EXPM_BUFFERS * kk))) return rc; }
    ExpmPost ep = {};
    post = post && k == 4 && h->fmt_kind == 1 && !h->fmt_dirty && h->d_edge_slot;
    if (post) {
        ep.edge_slot = h->d_edge_slot; ep.nmat1 = (int)h->mat_edge.size() + 1; ep.ntips1 = (int)h->tip_edge.size() + 1;
        ep.nchar = h->nchar; ep.defs = h->d_defs; ep.PS = h->d_PS; ep.tip = h->d_tip;
        if (h->fmt_pt) { ep.nmat1 = (int)h->fpt.mat_edge.size() + 1; ep.ntips1 = h->fpt.units; }    /* no tip slots in edge_slot: tables below */
    }
    ep.skip_dP = need_dP ? 0 : 1;
    hipLaunchKernelGGL(k_expm_dd<false>, dim3(C * E), dim3(threads), lds_bytes, h->stream,
                       k, E, h->d_Qn, h->d_edge_rates, h->d_cat_rates, h->d_Pdd, h->d_P, h->d_dP,
                       h->d_scratch, use_lds, (const double *)nullptr, 0, (const int *)nullptr, (double *)nullptr, ep);
    HIPCHK(h, hipGetLastError());
    if (post && h->fmt_pt) {
        const int ntab = (int)h->fpt.tab_unit.size();
        hipLaunchKernelGGL(k_build_tables_pt, dim3(ntab, C), dim3(64), 0, h->stream,
                           E, ntab, h->fpt.units, h->nchar, h->d_tabs, h->d_Pdd, h->d_defs, h->d_tip);
        HIPCHK(h, hipGetLastError());
    }
    h->model_dirty = false;
    h->dp_valid = need_dP;
    h->tables_dirty = !post;
    return PLK_OK;
}

/* the derivative paths read dP; an ll evaluation at new rates leaves it to the first of them */
static int ensure_dP(plk_engine *h)
{
    if (h->dp_valid || h->E == 0) return PLK_OK;
    const int k = h->k;
    const size_t kk = (size_t)k * k;
    const int threads = kk >= 1024 ? 1024 : (kk >= 256 ? 256 : 64);
    int use_lds;
    const size_t strip_bytes = expm_lds_bytes((size_t)k, threads, &use_lds);
    hipLaunchKernelGGL(k_dP_dd, dim3(h->C * h->E), dim3(threads), use_lds ? 0 : strip_bytes, h->stream,
                       k, h->E, h->d_Qn, h->d_cat_rates, h->d_Pdd, h->d_dP, use_lds ? 0 : 1);
    HIPCHK(h, hipGetLastError());
    h->dp_valid = true;
    return PLK_OK;
}

extern "C" int plk_set_model(plk_engine *h, int k, int C, const double *Qn, const double *Qn_lo,
                             const double *edge_rates_csr,
                             const double *cat_rates, const double *cat_prior, int root_mode,
                             const double *root_w)
{
    if (!plk_live(h)) return PLK_E_ARG;
    if (h->N == 0) { h->err = "plk_set_model: set the tree first"; return PLK_E_ARG; }
    if (k < 1 || C < 1 || !Qn || !edge_rates_csr || !cat_rates || !cat_prior) { h->err = "plk_set_model: bad arguments"; return PLK_E_ARG; }
    if (k > PLK_MAX_K) { h->err = "plk_set_model: more than 64 states is not supported yet"; return PLK_E_UNSUPPORTED; }
    if (C > PLK_MAX_C) { h->err = "plk_set_model: more than 64 rate categories is not supported"; return PLK_E_UNSUPPORTED; }
    if (root_mode < PLK_ROOT_NONE || root_mode > PLK_ROOT_EQUILIBRIUM) { h->err = "plk_set_model: bad root mode"; return PLK_E_ARG; }
    if ((root_mode == PLK_ROOT_CUSTOM || root_mode == PLK_ROOT_EQUILIBRIUM) && !root_w) { h->err = "plk_set_model: root_w required"; return PLK_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    if (h->k != k) { h->pat_mode = 0; }
    h->k = k; h->C = C; h->K = pad_K(k); h->root_mode = root_mode;
    h->Qn.assign(Qn, Qn + (size_t)k * k);
    h->Qn.resize(2 * (size_t)k * k, 0.0);             /* [hi | lo] */
    if (Qn_lo) std::copy(Qn_lo, Qn_lo + (size_t)k * k, h->Qn.begin() + (size_t)k * k);
    h->edge_rates.assign(edge_rates_csr, edge_rates_csr + h->E);
    h->cat_rates.assign(cat_rates, cat_rates + C);
    h->cat_prior.assign(cat_prior, cat_prior + C);
    std::vector<double> rw(h->K, 0.0);
    for (int i = 0; i < k; i++) {
        if (root_mode == PLK_ROOT_NONE) rw[i] = 1.0;
        else if (root_mode == PLK_ROOT_UNIFORM) rw[i] = 1.0 / (double)k;
        else rw[i] = root_w[i];
    }
    h->root_w = rw;
    int rc;
    if ((rc = dev_upload(h, &h->d_Qn, h->Qn.data(), h->Qn.size()))) return rc;
    if ((rc = dev_upload(h, &h->d_edge_rates, h->edge_rates.data(), h->edge_rates.size()))) return rc;
    if ((rc = dev_upload(h, &h->d_cat_rates, h->cat_rates.data(), h->cat_rates.size()))) return rc;
    if ((rc = dev_upload(h, &h->d_cat_prior, h->cat_prior.data(), h->cat_prior.size()))) return rc;
    if ((rc = dev_upload(h, &h->d_root_w, rw.data(), rw.size()))) return rc;
    h->prog_dirty = true;
    if ((rc = run_expm(h))) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return PLK_OK;
}

extern "C" int plk_update_edge_rates(plk_engine *h, const double *edge_rates_csr)
{
    if (!plk_live(h) || !edge_rates_csr) return PLK_E_ARG;
    if (h->k == 0) { h->err = "plk_update_edge_rates: set the model first"; return PLK_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    h->edge_rates.assign(edge_rates_csr, edge_rates_csr + h->E);
    HIPCHK(h, hipMemcpyAsync(h->d_edge_rates, h->edge_rates.data(), h->E * sizeof(double), hipMemcpyHostToDevice, h->stream));
    h->model_dirty = true;
    return PLK_OK;
}

extern "C" int plk_get_transition_matrices(plk_engine *h, double *P_out)
{
    if (!plk_live(h) || !P_out) return PLK_E_ARG;
    if (h->k == 0) { h->err = "plk_get_transition_matrices: set the model first"; return PLK_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    if (h->model_dirty) { int rc = run_expm(h); if (rc) return rc; }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(P_out, h->d_P, (size_t)h->C * h->E * h->k * h->k * sizeof(double), hipMemcpyDeviceToHost));
    return PLK_OK;
}

static int copy_in(plk_engine *h, void *dst, const void *src, size_t bytes, int where)
{
    HIPCHK(h, hipMemcpy(dst, src, bytes, where == PLK_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    return PLK_OK;
}

extern "C" int plk_set_patterns_codes(plk_engine *h, long S, const uint8_t *codes, int where, int nchar,
                                      const double *defs)
{
    if (!plk_live(h)) return PLK_E_ARG;
    if (h->k == 0) { h->err = "plk_set_patterns_codes: set the model first"; return PLK_E_ARG; }
    if (S < 1 || !codes || nchar < 1 || nchar > 256 || !defs) { h->err = "plk_set_patterns_codes: bad arguments"; return PLK_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    const int k = h->k, K = h->K, N = h->N;
    const long Spad = (S + 3071) / 3072 * 3072;      /* rows padded to a whole number of 1024- and 1536-site tiles */
    int rc;
    if ((rc = dev_alloc(h, &h->d_codes, (size_t)N * Spad))) return rc;
    HIPCHK(h, hipMemset(h->d_codes, 0, (size_t)N * Spad));
    HIPCHK(h, hipMemcpy2D(h->d_codes, Spad, codes, S, S, N,
                          where == PLK_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    h->defs.assign(defs, defs + (size_t)nchar * k);
    std::vector<double> dpad((size_t)nchar * K, 0.0);
    std::vector<int> trivial(nchar, 1);                 /* bit 0: all-ones definition; bit 1: a single 1.0 among zeros */
    for (int ch = 0; ch < nchar; ch++) {
        int nnz = 0, ones = 0;
        for (int j = 0; j < k; j++) {
            const double d = defs[(size_t)ch * k + j];
            dpad[(size_t)ch * K + j] = d;
            if (d != 1.0) trivial[ch] = 0;
            if (d != 0.0) nnz++;
            if (d == 1.0) ones++;
        }
        if (nnz == 1 && ones == 1) trivial[ch] |= 2;
    }
    if ((rc = dev_upload(h, &h->d_defs, dpad.data(), dpad.size()))) return rc;
    h->S = S; h->Spad = Spad; h->nchar = nchar; h->pat_mode = 1;
    if (h->d_B) { (void)hipFree(h->d_B); h->d_B = nullptr; }
    if (h->d_w) { (void)hipFree(h->d_w); h->d_w = nullptr; }
    /* which nodes carry observations that are not all-ones */
    int *d_triv = nullptr, *d_flags = nullptr;
    if ((rc = dev_upload(h, &d_triv, trivial.data(), trivial.size()))) return rc;
    if ((rc = dev_alloc(h, &d_flags, (size_t)N))) { (void)hipFree(d_triv); return rc; }
    HIPCHK(h, hipMemset(d_flags, 0, N * sizeof(int)));
    hipLaunchKernelGGL(k_node_flags, dim3((unsigned)std::min<long>(64, (S + 4095) / 4096), N), dim3(256), 0, h->stream,
                       S, Spad, h->d_codes, nchar, d_triv, d_flags);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    std::vector<int> flags(N);
    HIPCHK(h, hipMemcpy(flags.data(), d_flags, N * sizeof(int), hipMemcpyDeviceToHost));
    (void)hipFree(d_triv); (void)hipFree(d_flags);
    h->node_has_data.assign(N, 0);
    h->node_observed.assign(N, 0);
    for (int a = 0; a < N; a++) {
        if (flags[a] & 2) {
            h->pat_mode = 0;
            h->err = "plk_set_patterns_codes: a pattern code is not less than the number of character definitions";
            return PLK_E_ARG;
        }
        h->node_has_data[a] = (flags[a] & 3) ? 1 : 0;
        h->node_observed[a] = (flags[a] & 4) ? 0 : 1;
    }
    h->prog_dirty = true;
    return PLK_OK;
}

extern "C" int plk_set_patterns_dense(plk_engine *h, long S, const double *B, int where)
{
    if (!plk_live(h)) return PLK_E_ARG;
    if (h->k == 0) { h->err = "plk_set_patterns_dense: set the model first"; return PLK_E_ARG; }
    if (S < 1 || !B) { h->err = "plk_set_patterns_dense: bad arguments"; return PLK_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    const size_t n = (size_t)h->N * h->k * S;
    int rc;
    if ((rc = dev_alloc(h, &h->d_B, n))) return rc;
    if ((rc = copy_in(h, h->d_B, B, n * sizeof(double), where))) return rc;
    if (h->d_codes) { (void)hipFree(h->d_codes); h->d_codes = nullptr; }
    if (h->d_w) { (void)hipFree(h->d_w); h->d_w = nullptr; }
    h->S = S; h->Spad = S; h->pat_mode = 2; h->nchar = 0;
    h->node_has_data.assign(h->N, 1);
    h->node_observed.clear();
    h->prog_dirty = true;
    return PLK_OK;
}

extern "C" int plk_set_site_weights(plk_engine *h, const double *w, int where)
{
    if (!plk_live(h)) return PLK_E_ARG;
    if (h->pat_mode == 0) { h->err = "plk_set_site_weights: set the patterns first"; return PLK_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    if (!w) { if (h->d_w) { (void)hipFree(h->d_w); h->d_w = nullptr; } return PLK_OK; }
    int rc;
    if ((rc = dev_alloc(h, &h->d_w, (size_t)h->S))) return rc;
    return copy_in(h, h->d_w, w, (size_t)h->S * sizeof(double), where);
}

/* ---------------------------------------------------------------------- */
/* traversal program                                                       */
/* ---------------------------------------------------------------------- */

static int build_program(plk_engine *h)
{
    if (h->node_has_data.size() != (size_t)h->N) h->node_has_data.assign(h->N, 1);
    plk_program_build(h->N, h->indptr.data(), h->indices.data(), h->preorder.data(), h->node_has_data.data(), h->pg);
    const std::string bad = plk_program_check(h->N, h->indptr.data(), h->indices.data(), h->preorder.data(),
                                              h->node_has_data.data(), h->pg);
    if (!bad.empty()) { h->err = "internal: " + bad; return PLK_E_ARG; }
    h->prog_dirty = false;
    h->fmt_dirty = true;
    h->tables_dirty = true;
    return PLK_OK;
}

/* sites per lane of the fused kernel: 1 (the assembly interpreter, the fastest variant measured) unless the
 * option asks for the two-sites C++ variant and the stack (<= 8 slots) allows it */
static int fused_sites_per_lane(const plk_engine *h)
{
    return h->opt_fused_ns == 2 && h->slots_needed <= 8 ? 2 : 1;
}

/* register-resident vector kernel (plk_vec.h): 9 <= k <= 32 with compact codes (amino acids); PLK_OPT_MFMA = 2
 * forces the matrix-core kernel instead */
static bool use_vec(const plk_engine *h)
{
    return !h->opt_force_generic && h->opt_mfma == 1 && h->pat_mode == 1 && h->k >= 9 && h->k <= 32;
}

/* fp64 matrix-core kernel: larger state spaces with compact codes */
static bool use_mfma(const plk_engine *h)
{
    return !h->opt_force_generic && h->opt_mfma && h->pat_mode == 1 && h->k >= 9 && h->k <= 64;
}

/* dynamic LDS of k_ll_mfma: the A fragments of one matrix + one 64-byte code row per observed node */
static size_t mfma_ll_lds_bytes(const plk_engine *h)
{
    const int T = (h->k + 15) / 16, kk4 = (h->k + 3) / 4;
    return (size_t)T * kk4 * 64 * sizeof(double) + h->obs_nodes.size() * (size_t)MF_SITES;
}

/* k = 4 tile kernels: the assembly interpreter stages 4-bit codes when there are at most 16 character definitions,
 * which lets it take about twice the taxa of the C++ interpreter before the tile no longer fits the LDS */
static bool fused_asm_fits(const plk_engine *h)
{
    if (fused_sites_per_lane(h) != 1 || !plk_fused_asm_ok(h->pg) || !h->opt_fused_asm) return false;
    return plk_fused_lds_bytes(h->pg, h->nchar, h->nchar <= 16 ? PLK_TILE / 2 : PLK_TILE) <= PLK_LDS_LIMIT;
}

/* pair-table interpreter (k_ll_fused4_asm_pt): trees within the 4-slot VGPR stack and at most 16 character definitions
 * (the combined code of a cherry is one byte); as many cherries become tables as the two-workgroups-per-CU LDS budget
 * takes.  Fills fpt; false when even the table-less image does not fit (the 256-site kernels take over). */
/* PLK_OPT_PAIR_TABLES: which pair-table interpreter, how many sites per tile (candidates in order of preference) */
static bool build_fused_pt(plk_engine *h)
{
    if (!h->opt_pair_tables || !h->opt_fused_asm || h->opt_fused_ns == 2 || h->slots_needed > 4 || h->nchar > 16) return false;
    if (h->obs_nodes.empty()) return false;
    struct Cand { int kind, tile; };
    std::vector<Cand> cands;
    switch (h->opt_pair_tables & 7) {
    case 2: cands = {{1, 1024}}; break;
    case 3: cands = {{1, 512}}; break;
    case 5: cands = {{2, 1024}}; break;
    case 6: cands = {{2, 1536}}; break;
    default: {
        /* 1536-site tiles run 1.3 x as long as 1024-site tiles (109 against 84 us a round at BASELINE config 3) and carry 1.5 x
         * the sites; every CU runs ceil(tiles / CUs) rounds.  Few sites per GPU (a rank of an 8-way split) can need fewer
         * round-times with the smaller tile: 1.25M sites are 4 rounds of 1536 (436 us) or 5 rounds of 1024 (420 us). */
        auto rounds = [&](long tile) { const long nt = (h->S + tile - 1) / tile; return (double)((nt + h->num_cus - 1) / h->num_cus); };
        if (h->S > 0 && rounds(1024) * 1.0 < rounds(1536) * 1.3) cands = {{2, 1024}, {2, 1536}, {1, 1024}, {1, 512}};
        else cands = {{2, 1536}, {2, 1024}, {1, 1024}, {1, 512}};
        break;
    }
    }
    for (const Cand &cd : cands) {
        const long limit = (long)plk_pt_lds_limit(cd.tile);
        int max_pairs = INT_MAX;
        for (int attempt = 0; attempt < 3; attempt++) {
            plk_fused_pt_build(h->N, h->indptr.data(), h->indices.data(), h->pg, h->nchar, max_pairs, h->fpt, cd.kind == 2);
            const long bytes = (long)plk_fused_pt_lds_bytes(h->fpt, h->nchar, cd.tile);
            if (bytes <= limit && h->fpt.units < 2048 && h->fpt.row_node.size() < 65536) { h->pt_kind = cd.kind; h->pt_tile_sites = cd.tile; return true; }
            if (h->fpt.npairs == 0) break;
            /* a cherry as a table costs nchar - 2 more units and one row less than its two leaves */
            const long per_pair = (long)(h->nchar - 2) * h->nchar * 32 - cd.tile;
            if (per_pair <= 0) break;                       /* tables only shrink the image: nothing to give back */
            const long drop = (bytes - limit + per_pair - 1) / per_pair;
            max_pairs = drop >= h->fpt.npairs ? 0 : h->fpt.npairs - (int)drop;
        }
    }
    return false;
}

static bool use_fused(const plk_engine *h)
{
    if (h->opt_force_generic) return false;
    if (h->k != 4 || h->pat_mode != 1) return false;
    if (h->slots_needed > PLK_FUSED_SLOTS) return false;
    return fused_asm_fits(h) || plk_fused_lds_bytes(h->pg, h->nchar, PLK_TILE * fused_sites_per_lane(h)) <= PLK_LDS_LIMIT;
}

/* Device formats of the program for kernel kind (1 fused k = 4, 2 generic, 3 matrix cores, 4 vector): uploaded when
 * the program or the kind changes, not per evaluation. */
static int upload_formats(plk_engine *h, long kind)
{
    int rc;
    const int nops = (int)h->ops.size(), ntips = (int)h->tip_edge.size();
    std::vector<int> te = h->tip_edge;
    te.push_back(-1);                                /* pseudo slot: the raw definitions (internal nodes with data) */
    if ((rc = dev_upload(h, &h->d_tip_edge, te.data(), te.size()))) return rc;
    if ((rc = dev_upload(h, &h->d_obs_nodes, h->obs_nodes.data(), h->obs_nodes.size()))) return rc;
    h->fmt_pt = false;
    if (kind == 1 && build_fused_pt(h)) {
        /* pair-table interpreter: op words, staged-row nodes, table list; K1 writes the matrix stream itself, the tables
         * are built from the unrounded P after it (k_build_tables_pt) */
        const PlkFusedPT &f = h->fpt;
        const int nrows = (int)f.row_node.size(), ntab = (int)f.tab_unit.size();
        if (h->pt_kind == 2) {
            /* the table image starts where the kernel's static LDS ends */
            hipFuncAttributes fa;
            const void *fn = h->pt_tile_sites == 1536 ? (const void *)k_ll_fused4_v4<768> : (const void *)k_ll_fused4_v4<512>;
            HIPCHK(h, hipFuncGetAttributes(&fa, fn));
            h->v4_tip_base = (unsigned)fa.sharedSizeBytes;
            plk_fused_v4_words(f, h->nchar, h->pt_tile_sites, h->v4_tip_base, h->fv4);
            if ((rc = dev_upload(h, &h->d_words_pt, h->fv4.words.data(), h->fv4.words.size()))) return rc;
        } else if ((rc = dev_upload(h, &h->d_words_pt, f.words.data(), f.words.size()))) return rc;
        std::vector<int> rn(f.row_node);
        rn.insert(rn.end(), f.row_node2.begin(), f.row_node2.end());
        if ((rc = dev_upload(h, &h->d_row_nodes, rn.data(), rn.size()))) return rc;
        std::vector<int> tabs(f.tab_unit);
        tabs.insert(tabs.end(), f.tab_edge.begin(), f.tab_edge.end());
        tabs.insert(tabs.end(), f.tab_eb.begin(), f.tab_eb.end());
        tabs.insert(tabs.end(), f.tab_ec.begin(), f.tab_ec.end());
        if ((rc = dev_upload(h, &h->d_tabs, tabs.data(), tabs.size()))) return rc;
        std::vector<int> em(2 * (size_t)h->E, -1);
        for (size_t mi = 0; mi < f.mat_edge.size(); mi++) em[f.mat_edge[mi]] = (int)mi;
        if ((rc = dev_upload(h, &h->d_edge_slot, em.data(), em.size()))) return rc;
        if ((rc = dev_reserve(h, &h->d_PS, &h->ps_cap, (size_t)h->C * (f.mat_edge.size() + 1) * 16))) return rc;
        if ((rc = dev_reserve(h, &h->d_tip, &h->tip_cap, (size_t)h->C * f.units * h->nchar * 4))) return rc;
        HIPCHK(h, hipMemsetAsync(h->d_PS, 0, (size_t)h->C * (f.mat_edge.size() + 1) * 16 * sizeof(double), h->stream));
        (void)nrows; (void)ntab;
        h->fmt_pt = true;
    } else if (kind == 1) {
        /* int4 ops of the C++ interpreter, 32-bit op words of the assembly interpreter, compact matrix list */
        plk_fused_build(h->N, h->pg, h->fu);
        if ((rc = dev_upload(h, &h->d_fops, reinterpret_cast<const int4 *>(h->fu.fops.data()), h->fu.fops.size()))) return rc;
        if ((rc = dev_upload(h, &h->d_words, h->fu.words.data(), h->fu.words.size()))) return rc;
        std::vector<int> me = h->mat_edge;
        me.push_back(-1);                            /* the spare matrix is all zeros */
        if ((rc = dev_upload(h, &h->d_mat_edge, me.data(), me.size()))) return rc;
        /* where K1 itself puts P of edge e: its matrix of the stream (internal edges) or its tip slot (leaf edges) */
        std::vector<int> em(2 * (size_t)h->E, -1);
        for (size_t mi = 0; mi < h->mat_edge.size(); mi++) em[h->mat_edge[mi]] = (int)mi;
        for (int t = 0; t < ntips; t++) em[(size_t)h->E + h->tip_edge[t]] = t;
        if ((rc = dev_upload(h, &h->d_edge_slot, em.data(), em.size()))) return rc;
        if ((rc = dev_reserve(h, &h->d_PS, &h->ps_cap, (size_t)h->C * (h->mat_edge.size() + 1) * 16))) return rc;
        if ((rc = dev_reserve(h, &h->d_tip, &h->tip_cap, (size_t)h->C * (ntips + 1) * h->nchar * 4))) return rc;
        /* K1 fills the stream and the tip slots of the edges; the spare matrices stay zero and the pseudo tip slot
         * (raw definitions) is written here */
        HIPCHK(h, hipMemsetAsync(h->d_PS, 0, (size_t)h->C * (h->mat_edge.size() + 1) * 16 * sizeof(double), h->stream));
        hipLaunchKernelGGL(k_build_tip, dim3(ntips + 1, h->C), dim3(64), 0, h->stream,
                           h->E, ntips + 1, h->nchar, h->d_tip_edge, h->d_Pdd, h->d_defs, h->d_tip);
        HIPCHK(h, hipGetLastError());
    } else {
        /* OP_MATVEC ops name the next OP_MATVEC (y, wrapping to the first): the vector kernel touches the
         * cache lines of the next matrix while it multiplies with the current one */
        std::vector<plk_op2> gops(h->ops);
        int first = -1, prev = -1;
        for (int pc = 0; pc < nops; pc++)
            if ((gops[pc].x & 0xff) == OP_MATVEC) {
                if (first < 0) first = pc;
                if (prev >= 0) gops[prev].y = pc;
                prev = pc;
            }
        if (prev >= 0) gops[prev].y = first;
        if ((rc = dev_upload(h, &h->d_ops, reinterpret_cast<const int2 *>(gops.data()), gops.size()))) return rc;
        if ((rc = dev_upload(h, &h->d_op_edge, h->op_edge.data(), h->op_edge.size()))) return rc;
        if ((rc = dev_reserve(h, &h->d_PS, &h->ps_cap, (size_t)h->C * std::max(nops, 1) * h->K * h->K))) return rc;
        h->vec_pt = false;
        if (kind == 4 && h->opt_pair_tables && h->K <= 32 && !h->obs_nodes.empty()) {
            /* pair-table program for the vector kernel: two-leaf subtrees as rows of L2-resident tables (nchar^2 rows of K
             * doubles each), as many as 64 MB of tables per category take */
            const long per_pair = (long)h->nchar * h->nchar * h->K * (long)sizeof(double);
            const int max_pairs = (int)std::min<long>(INT_MAX, (64L << 20) / std::max<long>(per_pair, 1));
            plk_fused_pt_build(h->N, h->indptr.data(), h->indices.data(), h->pg, h->nchar, max_pairs, h->fpt, false);
            if (h->fpt.npairs > 0) {
                plk_vec_pt_build(h->fpt, h->nchar, h->vpt);
                const std::string bad = plk_vec_pt_check(h->N, h->indptr.data(), h->indices.data(), h->pg, h->fpt, h->vpt, h->nchar, h->E);
                if (!bad.empty()) { h->err = "internal: " + bad; return PLK_E_ARG; }
                const PlkFusedPT &f = h->fpt;
                std::vector<int> rn(f.row_node);
                rn.insert(rn.end(), f.row_node2.begin(), f.row_node2.end());
                std::vector<int> tabs(f.tab_unit);
                tabs.insert(tabs.end(), f.tab_edge.begin(), f.tab_edge.end());
                tabs.insert(tabs.end(), f.tab_eb.begin(), f.tab_eb.end());
                tabs.insert(tabs.end(), f.tab_ec.begin(), f.tab_ec.end());
                if ((rc = dev_upload(h, &h->d_row_nodes, rn.data(), rn.size())) || (rc = dev_upload(h, &h->d_tabs, tabs.data(), tabs.size())) ||
                    (rc = dev_upload(h, &h->d_mops, reinterpret_cast<const int4 *>(h->vpt.ops.data()), h->vpt.ops.size())) ||
                    (rc = dev_upload(h, &h->d_op_edge, h->vpt.op_edge.data(), h->vpt.op_edge.size())) ||
                    (rc = dev_reserve(h, &h->d_PS, &h->ps_cap, (size_t)h->C * h->vpt.ops.size() * h->K * h->K)) ||
                    (rc = dev_reserve(h, &h->d_tip, &h->tip_cap, (size_t)h->C * f.units * h->nchar * h->K))) return rc;
                h->vec_pt = true;
            }
        }
        if (kind == 4 && h->vec_pt) {
            /* formats uploaded above */
        } else if (kind == 4) {
            if ((rc = dev_reserve(h, &h->d_tip, &h->tip_cap, (size_t)h->C * (ntips + 1) * h->nchar * h->K))) return rc;
            /* vector program: observation ops chained two ahead (value of the next op, code of the one after) */
            PlkChain ch;
            plk_chain_build(h->N, h->pg, 3, h->indices.data(), nullptr, nullptr, nullptr, ch);
            const std::string bad = plk_chain_check(h->N, h->pg, ch, 3, INT_MAX, 0, 0, 0, 0, 0);
            if (!bad.empty()) { h->err = "internal: " + bad; return PLK_E_ARG; }
            h->mfma_first_slot = ch.first_slot; h->mfma_first_row = ch.first_row; h->vec_second_row = ch.second_row;
            if ((rc = dev_upload(h, &h->d_mops, reinterpret_cast<const int4 *>(ch.ops.data()), ch.ops.size()))) return rc;
        } else if (kind == 3) {
            const int T = (h->k + 15) / 16, R = 4 * T, kk4 = (h->k + 3) / 4;
            std::vector<double> rwd((size_t)4 * R, 0.0);
            for (int i = 0; i < h->k; i++) rwd[(size_t)(i & 3) * R + (i >> 2)] = h->root_w[i];
            if ((rc = dev_upload(h, &h->d_root_wd, rwd.data(), rwd.size()))) return rc;
            if ((rc = dev_reserve(h, &h->d_frag, &h->frag_cap, (size_t)h->C * nops * T * kk4 * 64))) return rc;
            if ((rc = dev_reserve(h, &h->d_tip, &h->tip_cap, (size_t)h->C * (ntips + 1) * h->nchar * 4 * R))) return rc;
            /* MFMA program: observation ops name their staged code row and the next observation op */
            PlkChain ch;
            plk_chain_build(h->N, h->pg, 0, nullptr, nullptr, nullptr, nullptr, ch);
            const std::string bad = plk_chain_check(h->N, h->pg, ch, 0, INT_MAX, 0, 0, 0, MF_SITES,
                                                    (size_t)h->obs_nodes.size() * MF_SITES);
            if (!bad.empty()) { h->err = "internal: " + bad; return PLK_E_ARG; }
            h->mfma_first_slot = ch.first_slot; h->mfma_first_row = ch.first_row;
            h->mfma_first_mv = -1;
            for (size_t pc = 0; pc < h->pg.ops.size() && h->mfma_first_mv < 0; pc++) if ((h->pg.ops[pc].x & 0xff) == OP_MATVEC) h->mfma_first_mv = (int)pc;
            if ((rc = dev_upload(h, &h->d_mops, reinterpret_cast<const int4 *>(ch.ops.data()), ch.ops.size()))) return rc;
        }
    }
    h->fmt_kind = kind;
    h->fmt_dirty = false;
    h->tables_dirty = true;
    return PLK_OK;
}

/* P-dependent tables of kernel kind, from the current P (used when K1 ran without writing them itself) */
static int build_tables(plk_engine *h, long kind)
{
    const int nops = (int)h->ops.size(), ntips = (int)h->tip_edge.size(), K = h->K, C = h->C;
    if (kind == 1) {
        /* the fused formats are written by K1 itself (ExpmPost) */
        return run_expm(h, true);
    } else {
        if (kind == 4 && h->vec_pt) {
            const int nv = (int)h->vpt.ops.size(), ntab = (int)h->fpt.tab_unit.size();
            hipLaunchKernelGGL(k_build_stream, dim3(nv, C), dim3(K * K >= 256 ? 256 : 64), 0, h->stream,
                               h->k, K, h->E, nv, h->d_op_edge, h->d_P, h->d_PS);
            hipLaunchKernelGGL(k_build_tables_pt_vec, dim3(ntab, C, 8), dim3(256), (size_t)4 * h->nchar * h->k * sizeof(double), h->stream,
                               h->k, K, h->E, ntab, h->fpt.units, h->nchar, h->d_tabs, h->d_Pdd, h->d_defs, h->d_tip);
            HIPCHK(h, hipGetLastError());
            h->tables_dirty = false;
            return PLK_OK;
        }
        if (nops > 0)
            hipLaunchKernelGGL(k_build_stream, dim3(nops, C), dim3(K * K >= 256 ? 256 : 64), 0, h->stream,
                               h->k, K, h->E, nops, h->d_op_edge, h->d_P, h->d_PS);
        if (kind == 4) {
            hipLaunchKernelGGL(k_build_tip_vec, dim3(ntips + 1, C), dim3(256), 0, h->stream,
                               h->k, K, h->E, ntips, h->nchar, h->d_tip_edge, h->d_Pdd, h->d_defs, h->d_tip);
        } else if (kind == 3) {
            const int T = (h->k + 15) / 16, R = 4 * T, kk4 = (h->k + 3) / 4;
            hipLaunchKernelGGL(k_build_frag, dim3(nops, C), dim3(256), 0, h->stream,
                               h->k, T, kk4, h->E, nops, h->d_op_edge, h->d_P, h->d_frag);
            hipLaunchKernelGGL(k_build_tip_dist, dim3(ntips + 1, C), dim3(256), 0, h->stream,
                               h->k, R, h->E, ntips, h->nchar, h->d_tip_edge, h->d_Pdd, h->d_defs, h->K, h->d_tip);
        }
    }
    HIPCHK(h, hipGetLastError());
    h->tables_dirty = false;
    return PLK_OK;
}

/* the window of three stack slots that takes most pushes (register stack of the vector kernels) */
static int vec_reg_window(const plk_engine *h)
{
    std::vector<long> per_slot(std::max(h->slots_needed, 1), 0);
    for (const plk_op2 &o : h->ops) if ((o.x & 0xff) == OP_PUSH && o.y < (int)per_slot.size()) per_slot[o.y]++;
    long best = -1;
    int lo_best = 0;
    for (int lo = 0; lo + 3 <= std::max(h->slots_needed, 3); lo++) {
        long cnt = 0;
        for (int q = lo; q < lo + 3 && q < (int)per_slot.size(); q++) cnt += per_slot[q];
        if (cnt > best) { best = cnt; lo_best = lo; }
    }
    return lo_best;
}

template <int D, int NS>
static void launch_fused(plk_engine *h, const FusedArgs &a, unsigned grid, size_t lds)
{
    hipLaunchKernelGGL((k_ll_fused4<D, NS>), dim3(grid), dim3(PLK_TILE), lds, h->stream, a);
}

template <int K>
static void launch_generic(plk_engine *h, const GenArgs &a, unsigned grid)
{
    hipLaunchKernelGGL(k_ll_generic<K>, dim3(grid), dim3(GEN_BLOCK), 0, h->stream, a);
}

#define PLK_PARTIAL_OFF 72      /* d_partial: [0] the sum, [4, 68) second-stage inputs, [72, ...) one partial per workgroup */

/* sum_dev != null: the {hi, lo} sum is left in device memory (2 doubles) and nothing is copied or waited for */
static int ll_impl(plk_engine *h, double *site_ll_out, int where, double *sum_out, double *sum_dev, bool async)
{
    if (h->k == 0 || h->pat_mode == 0) { h->err = "plk_ll: tree, model and patterns must be set"; return PLK_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    if (h->prog_dirty) { if ((rc = build_program(h))) return rc; }
    const bool fused = use_fused(h);
    /* trees with so many observed nodes that their staged code rows do not fit in LDS take the generic kernel */
    const bool vec = !fused && use_vec(h), mfma = !fused && !vec && use_mfma(h) && mfma_ll_lds_bytes(h) <= PLK_LDS_LIMIT;
    const long kind = fused ? 1 : (vec ? 4 : (mfma ? 3 : 2));
    if (h->fmt_dirty || kind != h->fmt_kind) { if ((rc = upload_formats(h, kind))) return rc; }
    if (h->model_dirty) { if ((rc = run_expm(h, true, false))) return rc; }      /* K1 without dP; for k = 4 it writes the stream and tip tables too */
    if (h->tables_dirty) { if ((rc = build_tables(h, kind))) return rc; }
    const bool want_sum = sum_out || sum_dev;
    const long S = h->S;
    double *d_out = nullptr;
    if (site_ll_out) {
        if (where == PLK_DEVICE) d_out = site_ll_out;
        else { if ((rc = dev_reserve(h, &h->d_site_ll, &h->site_ll_cap, (size_t)S))) return rc; d_out = h->d_site_ll; }
    }
    unsigned grid;
    const int evi = h->evk_next;
    h->evk_next = (evi + 1) & 63;
    if ((rc = evk_resolve(h, evi))) return rc;           /* only waits when 64 evaluations are queued */
    HIPCHK(h, hipEventRecord(h->evk[evi][0], h->stream));
    h->info_ll_variant = 0;
    if (fused) {
        const int NS = fused_sites_per_lane(h);
        grid = (unsigned)((S + PLK_TILE * NS - 1) / (PLK_TILE * NS));
        if (want_sum) { if ((rc = dev_reserve(h, &h->d_partial, &h->partial_cap, (size_t)grid + PLK_PARTIAL_OFF))) return rc; }
        FusedArgs a;
        a.S = S; a.Spad = h->Spad; a.C = h->C; a.nops = (int)h->ops.size(); a.nmat = (int)h->mat_edge.size();
        a.ntips = (int)h->tip_edge.size() + 1; a.nchar = h->nchar; a.nobs = (int)h->obs_nodes.size();
        a.ops = h->d_fops; a.PS = h->d_PS; a.tip = h->d_tip;
        a.codes = h->d_codes; a.obs_nodes = h->d_obs_nodes; a.defs = h->d_defs;
        a.cat_prior = h->d_cat_prior; a.root_w = h->d_root_w; a.w = h->d_w;
        a.site_ll = d_out; a.partial = want_sum ? h->d_partial + PLK_PARTIAL_OFF : nullptr;
        a.root_mode = h->root_mode; a.first_row = h->fu.first_row;
        const size_t lds = plk_fused_lds_bytes(h->pg, h->nchar, PLK_TILE * NS);
        const bool use_asm = fused_asm_fits(h);
        const int D = h->slots_needed <= 4 ? 4 : (h->slots_needed <= 8 ? 8 : 16);
        if (h->fmt_pt) {
            /* pair-table interpreter: one 1024-site workgroup per CU (or two of 512), every workgroup walks the tiles with a grid stride */
            const PlkFusedPT &f = h->fpt;
            const int tile = h->pt_tile_sites;
            const int ntiles = (int)((S + tile - 1) / tile);
            if (want_sum) { if ((rc = dev_reserve(h, &h->d_partial, &h->partial_cap, (size_t)ntiles + PLK_PARTIAL_OFF))) return rc; }
            const size_t lds_pt = plk_fused_pt_lds_bytes(f, h->nchar, tile);
            std::string bad = plk_fused_check_pt(h->N, h->indptr.data(), h->indices.data(), h->pg, f, h->nchar, tile, lds_pt, h->pt_kind == 2);
            if (bad.empty() && h->pt_kind == 2) bad = plk_fused_check_v4(f, h->fv4, h->nchar, tile, h->v4_tip_base, lds_pt);
            if (!bad.empty()) { h->err = "internal: " + bad; return PLK_E_ARG; }
            FusedPTArgs pa;
            pa.f = a;
            pa.f.nmat = (int)f.mat_edge.size(); pa.f.ntips = f.units; pa.f.nobs = (int)f.row_node.size();
            pa.f.partial = want_sum ? h->d_partial + PLK_PARTIAL_OFF : nullptr;
            pa.words = h->d_words_pt; pa.row_nodes = h->d_row_nodes; pa.nwords = (int)(h->pt_kind == 2 ? h->fv4.words.size() : f.words.size());
            pa.first_unit = f.first_unit; pa.first_row = f.first_row; pa.second_row = f.second_row; pa.ntiles = ntiles;
            pa.warm = (h->opt_pair_tables & 8) ? 0 : 1;      /* + 8: without the scalar-cache warm-up (measurements) */
            h->info_ll_variant = 5;
            if (h->pt_kind == 2) {
                /* two sites per lane: one workgroup of tile / 2 lanes per CU */
                FusedV4Args va;
                va.pt = pa; va.first_y = h->fv4.first_y; va.first_z = h->fv4.first_z; va.second_z = h->fv4.second_z;
                grid = (unsigned)std::min(ntiles, h->num_cus);
                if (tile == 1536) hipLaunchKernelGGL(k_ll_fused4_v4<768>, dim3(grid), dim3(768), lds_pt, h->stream, va);
                else hipLaunchKernelGGL(k_ll_fused4_v4<512>, dim3(grid), dim3(512), lds_pt, h->stream, va);
                h->info_ll_variant = 6;
            } else if (tile == 1024) {
                grid = (unsigned)std::min(ntiles, h->num_cus);
                hipLaunchKernelGGL(k_ll_fused4_asm_pt<1024>, dim3(grid), dim3(1024), lds_pt, h->stream, pa);
            } else {
                grid = (unsigned)std::min(ntiles, 2 * h->num_cus);
                hipLaunchKernelGGL(k_ll_fused4_asm_pt<512>, dim3(grid), dim3(512), lds_pt, h->stream, pa);
            }
            grid = (unsigned)ntiles;                      /* one partial sum per tile */
        } else if (use_asm) {
            FusedAsmArgs aa;
            aa.f = a; aa.words = h->d_words;
            aa.first_tip = h->fu.asm_first_tip; aa.first_row = h->fu.asm_first_row; aa.second_row = h->fu.asm_second_row;
            aa.pack4 = h->nchar <= 16 ? 1 : 0;
            const size_t lds_asm = plk_fused_lds_bytes(h->pg, h->nchar, aa.pack4 ? PLK_TILE / 2 : PLK_TILE);
            /* replay the interpreter's fetches and LDS addresses on the host before launching (plk_program.h) */
            const std::string bad = plk_fused_check_asm(h->N, h->pg, h->fu, h->nchar, D, aa.pack4, lds_asm);
            if (!bad.empty()) { h->err = "internal: " + bad; return PLK_E_ARG; }
            h->info_ll_variant = 1;
            if (D == 4) hipLaunchKernelGGL(k_ll_fused4_asm<4>, dim3(grid), dim3(PLK_TILE), lds_asm, h->stream, aa);
            else hipLaunchKernelGGL(k_ll_fused4_asm<8>, dim3(grid), dim3(PLK_TILE), lds_asm, h->stream, aa);
        } else {
            h->info_ll_variant = 3;
            const std::string bad = plk_fused_check_cpp(h->N, h->pg, h->fu, h->nchar, NS == 2 && D == 16 ? 8 : D, NS, lds);
            if (!bad.empty()) { h->err = "internal: " + bad; return PLK_E_ARG; }
            if (NS == 2) {
                if (D == 4) launch_fused<4, 2>(h, a, grid, lds);
                else launch_fused<8, 2>(h, a, grid, lds);
            } else {
                if (D == 4) launch_fused<4, 1>(h, a, grid, lds);
                else if (D == 8) launch_fused<8, 1>(h, a, grid, lds);
                else launch_fused<16, 1>(h, a, grid, lds);
            }
        }
        h->info_ll_kernel = 1;
    } else if (vec) {
        /* 9 <= k <= 32 with compact codes: register-resident vector kernel (plk_vec.h) */
        const int K = h->K, nops = (int)h->ops.size(), ntips = (int)h->tip_edge.size();
        grid = (unsigned)((S + VEC_BLOCK - 1) / VEC_BLOCK);
        if (want_sum) { if ((rc = dev_reserve(h, &h->d_partial, &h->partial_cap, (size_t)grid + PLK_PARTIAL_OFF))) return rc; }
        const int nslots = std::max(h->slots_needed, 1);
        if ((rc = dev_reserve(h, &h->d_slots, &h->slots_cap, (size_t)nslots * K * S))) return rc;
        VecArgs a;
        a.S = S; a.Spad = h->Spad; a.k = h->k; a.C = h->C; a.nops = nops; a.ntips = ntips; a.nchar = h->nchar;
        a.root_mode = h->root_mode; a.ops = h->d_mops; a.PS = h->d_PS; a.tip = h->d_tip; a.codes = h->d_codes;
        a.obs_nodes = h->d_obs_nodes; a.first_slot = h->mfma_first_slot; a.first_row = h->mfma_first_row; a.second_row = h->vec_second_row;
        a.obs_nodes2 = nullptr; a.units = 0;
        if (h->vec_pt) {
            const int nrows = (int)h->fpt.row_node.size();
            a.nops = (int)h->vpt.ops.size(); a.obs_nodes = h->d_row_nodes; a.obs_nodes2 = h->d_row_nodes + nrows; a.units = h->fpt.units;
            a.first_slot = 0; a.first_row = h->vpt.first_row;
        }
        a.cat_prior = h->d_cat_prior; a.root_w = h->d_root_w; a.w = h->d_w; a.slots = h->d_slots; a.site_ll = d_out;
        a.partial = want_sum ? h->d_partial + PLK_PARTIAL_OFF : nullptr;
        /* register stack (PLK_OPT_VEC_REG_STACK, default on): the window of three slots that takes most pushes */
        const bool rs = h->opt_vec_reg_stack && K <= 20 && h->slots_needed >= 1;
        a.reg_lo = rs ? vec_reg_window(h) : 0;
        if (rs && K == 16) hipLaunchKernelGGL((k_ll_vec_rs<16, 3>), dim3(grid), dim3(VEC_BLOCK), 0, h->stream, a);
        else if (rs) hipLaunchKernelGGL((k_ll_vec_rs<20, 3>), dim3(grid), dim3(VEC_BLOCK), 0, h->stream, a);
        else if (K == 16) hipLaunchKernelGGL(k_ll_vec<16>, dim3(grid), dim3(VEC_BLOCK), 0, h->stream, a);
        else if (K == 20) hipLaunchKernelGGL(k_ll_vec<20>, dim3(grid), dim3(VEC_BLOCK), 0, h->stream, a);
        else hipLaunchKernelGGL(k_ll_vec<32>, dim3(grid), dim3(VEC_BLOCK), 0, h->stream, a);
        h->info_ll_kernel = 4;
    } else if (mfma) {
        /* 9 <= k <= 64 with compact codes: fp64 matrix-core kernel (plk_mfma.h) */
        const int T = (h->k + 15) / 16, R = 4 * T, kk4 = (h->k + 3) / 4;
        const int nops = (int)h->ops.size(), ntips = (int)h->tip_edge.size();
        /* two 16-site groups per wave (PLK_OPT_MFMA_NS2, default on) when the staged code rows of 128 sites fit the LDS */
        const bool ns2 = h->opt_mfma_ns2 == 1 && T >= 3 && (size_t)T * kk4 * 64 * sizeof(double) + h->obs_nodes.size() * (size_t)(2 * MF_SITES) <= PLK_LDS_LIMIT;
        const int wg_sites = ns2 ? 2 * MF_SITES : MF_SITES;
        grid = (unsigned)((S + wg_sites - 1) / wg_sites);
        if (want_sum) { if ((rc = dev_reserve(h, &h->d_partial, &h->partial_cap, (size_t)grid + PLK_PARTIAL_OFF))) return rc; }
        const long slot_stride = (long)grid * wg_sites * 4;
        const int nslots = std::max(h->slots_needed, 1);
        if ((rc = dev_reserve(h, &h->d_slots, &h->slots_cap, (size_t)nslots * R * slot_stride))) return rc;
        MfmaArgs a;
        a.S = S; a.Spad = h->Spad; a.k = h->k; a.kk4 = kk4; a.C = h->C; a.nops = nops; a.ntips = ntips;
        a.nchar = h->nchar; a.root_mode = h->root_mode; a.ops = h->d_mops; a.frag = h->d_frag; a.tip = h->d_tip;
        a.obs_nodes = h->d_obs_nodes; a.nobs = (int)h->obs_nodes.size();
        a.first_slot = h->mfma_first_slot; a.first_row = h->mfma_first_row; a.first_mv = h->mfma_first_mv;
        a.codes = h->d_codes; a.cat_prior = h->d_cat_prior; a.root_wd = h->d_root_wd; a.w = h->d_w;
        a.slots = h->d_slots; a.slot_stride = slot_stride; a.site_ll = d_out;
        a.partial = want_sum ? h->d_partial + PLK_PARTIAL_OFF : nullptr;
        const size_t lds = ns2 ? (size_t)T * kk4 * 64 * sizeof(double) + h->obs_nodes.size() * (size_t)(2 * MF_SITES) : mfma_ll_lds_bytes(h);
        if (false) {}
        else if (ns2 && T == 3) hipLaunchKernelGGL(k_ll_mfma_ns2<3>, dim3(grid), dim3(MF_BLOCK), lds, h->stream, a);
        else if (ns2) hipLaunchKernelGGL(k_ll_mfma_ns2<4>, dim3(grid), dim3(MF_BLOCK), lds, h->stream, a);
        else if (T == 1) hipLaunchKernelGGL(k_ll_mfma<1>, dim3(grid), dim3(MF_BLOCK), lds, h->stream, a);
        else if (T == 2) hipLaunchKernelGGL(k_ll_mfma<2>, dim3(grid), dim3(MF_BLOCK), lds, h->stream, a);
        else if (T == 3) hipLaunchKernelGGL(k_ll_mfma_occ4<3>, dim3(grid), dim3(MF_BLOCK), lds, h->stream, a);
        else hipLaunchKernelGGL(k_ll_mfma_occ4<4>, dim3(grid), dim3(MF_BLOCK), lds, h->stream, a);
        h->info_ll_kernel = 3;
    } else {
        grid = (unsigned)((S + GEN_BLOCK - 1) / GEN_BLOCK);
        if (want_sum) { if ((rc = dev_reserve(h, &h->d_partial, &h->partial_cap, (size_t)grid + PLK_PARTIAL_OFF))) return rc; }
        const int nslots = std::max(h->slots_needed, 1);
        if ((rc = dev_reserve(h, &h->d_slots, &h->slots_cap, (size_t)nslots * h->k * S))) return rc;
        GenArgs a;
        a.S = S; a.Spad = h->Spad; a.k = h->k; a.C = h->C; a.nops = (int)h->ops.size(); a.nchar = h->nchar;
        a.pat_mode = h->pat_mode; a.root_mode = h->root_mode; a.ops = h->d_ops; a.PS = h->d_PS;
        a.codes = h->d_codes; a.defs = h->d_defs; a.B = h->d_B; a.cat_prior = h->d_cat_prior;
        a.root_w = h->d_root_w; a.w = h->d_w; a.slots = h->d_slots; a.site_ll = d_out;
        a.partial = want_sum ? h->d_partial + PLK_PARTIAL_OFF : nullptr;
        switch (h->K) {
        case 2: launch_generic<2>(h, a, grid); break;
        case 4: launch_generic<4>(h, a, grid); break;
        case 8: launch_generic<8>(h, a, grid); break;
        case 16: launch_generic<16>(h, a, grid); break;
        case 20: launch_generic<20>(h, a, grid); break;
        case 32: launch_generic<32>(h, a, grid); break;
        case 61: launch_generic<61>(h, a, grid); break;
        default: launch_generic<64>(h, a, grid); break;
        }
        h->info_ll_kernel = 2;
    }
    {
        /* executed fp64 work per site (PLK_INFO_LL_EXEC_FLOPS): products and elementwise multiplies of the program that ran */
        long nprod = 0, nmul = 0;
        if (vec && h->vec_pt) {
            for (const PlkFusedPT::VOp &o : h->fpt.vops) { if (o.code == OP_MATVEC) nprod++; if (o.code == OP_TIP_MUL || o.code == OP_POPMUL) nmul++; }
        } else if (fused && h->fmt_pt) {
            nprod = (long)h->fpt.mat_edge.size();
            for (unsigned w : h->fpt.words) {
                const unsigned hx = w & 31;
                if (hx == OP_TIP_MUL || hx == PLK_WORD_TIPMUL_NOWAIT || hx == PLK_WORD_MATVEC_TIPMUL) nmul++;
                if ((hx >= 16 && hx < 20) || hx >= 28 || (hx >= PLK_WORD_SET_POPMUL && hx < PLK_WORD_SET_POPMUL + 4)) nmul++;
            }
        } else {
            for (const plk_op2 &o : h->ops) {
                const int code = o.x & 0xff;
                if (code == OP_MATVEC) nprod++;
                /* leaf edges: a table row multiplied in (fused, vector and matrix-core kernels) or a full product (generic) */
                if (code == OP_TIP_MUL || code == OP_NODE_MUL || code == OP_POPMUL) nmul++;
                if ((code == OP_TIP_SET || code == OP_TIP_MUL) && !fused && !vec && !mfma) { nprod++; if (code == OP_TIP_MUL) nmul++; }
            }
        }
        const long kp = mfma ? (h->k + 15) / 16 * 16 : h->k;
        h->info_ll_exec_flops = (long)h->C * (nprod * (2 * kp * kp - kp) + nmul * h->k);
    }
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(h->evk[evi][1], h->stream));
    h->evk_pending[evi] = true;
    if (want_sum) {
        /* fixed-order double-double sum of the workgroups' partials: up to 64 slices, then one small block */
        dd *out = sum_dev ? reinterpret_cast<dd *>(sum_dev) : h->d_partial;
        const int g2 = grid > 1024 ? (int)std::min<unsigned>(64u, (grid + 511) / 512) : 1;
        if (g2 > 1) {
            hipLaunchKernelGGL(k_dd_slices, dim3(g2), dim3(256), 0, h->stream, (int)grid, h->d_partial + PLK_PARTIAL_OFF, h->d_partial + 4);
            hipLaunchKernelGGL(k_dd_final, dim3(1), dim3(64), 0, h->stream, g2, h->d_partial + 4, out);
        } else {
            hipLaunchKernelGGL(k_dd_final, dim3(1), dim3(256), 0, h->stream, (int)grid, h->d_partial + PLK_PARTIAL_OFF, out);
        }
        HIPCHK(h, hipGetLastError());
    }
    HIPCHK(h, hipEventRecord(h->ev3, h->stream));
    h->info_pending = true;
    if (async) return PLK_OK;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (sum_out) {
        dd r;
        HIPCHK(h, hipMemcpy(&r, h->d_partial, sizeof(dd), hipMemcpyDeviceToHost));
        sum_out[0] = r.hi; sum_out[1] = r.lo;
    }
    if (site_ll_out && where != PLK_DEVICE)
        HIPCHK(h, hipMemcpy(site_ll_out, h->d_site_ll, (size_t)S * sizeof(double), hipMemcpyDeviceToHost));
    return PLK_OK;
}

extern "C" int plk_ll(plk_engine *h, double *site_ll_out, int where, double *sum_out)
{
    if (!plk_live(h)) return PLK_E_ARG;
    return ll_impl(h, site_ll_out, where, sum_out, nullptr, false);
}

/* The same evaluation queued on the engine's stream without waiting: per-site values (optional) and the {hi, lo} sum
 * (optional) are left in DEVICE memory the caller owns.  plk_sync() or any synchronous call waits for it. */
extern "C" int plk_ll_async(plk_engine *h, double *site_ll_dev, double *sum_dev)
{
    if (!plk_live(h)) return PLK_E_ARG;
    return ll_impl(h, site_ll_dev, PLK_DEVICE, nullptr, sum_dev, true);
}

extern "C" int plk_sync(plk_engine *h)
{
    if (!plk_live(h)) return PLK_E_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return PLK_OK;
}

/* Run the engine's work on a stream of the caller (e.g. the framework's current stream, so that its collectives and
 * events are ordered with the engine's kernels); NULL returns to the engine's own stream. */
extern "C" int plk_set_stream(plk_engine *h, void *hip_stream)
{
    if (!plk_live(h)) return PLK_E_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    return PLK_OK;
}

template <int K>
static void launch_updown(plk_engine *h, const UpArgs &a, unsigned grid, bool deriv, bool marg)
{
    hipLaunchKernelGGL(k_down_store<K>, dim3(grid), dim3(GEN_BLOCK), 0, h->stream, a);
    if (deriv && marg) hipLaunchKernelGGL((k_up<K, true, true>), dim3(grid), dim3(GEN_BLOCK), 0, h->stream, a);
    else if (deriv) hipLaunchKernelGGL((k_up<K, true, false>), dim3(grid), dim3(GEN_BLOCK), 0, h->stream, a);
    else hipLaunchKernelGGL((k_up<K, false, true>), dim3(grid), dim3(GEN_BLOCK), 0, h->stream, a);
}

/* rows x n weighted sums of X (row stride n) accumulated into acc[rows] (long double pairs) */
static int wsum_rows(plk_engine *h, int rows, long n, const double *X, const double *w, long double *acc)
{
    int rc;
    const int nblocks = (int)std::min<long>(256, (n + 2047) / 2048);
    if ((rc = dev_reserve(h, &h->d_partial, &h->partial_cap, (size_t)rows * nblocks + rows + 8))) return rc;
    dd *outp = h->d_partial;
    dd *part = h->d_partial + rows + 8;
    hipLaunchKernelGGL(k_wsum_rows, dim3(nblocks, rows), dim3(256), 0, h->stream, n, n, X, w, nblocks, part);
    hipLaunchKernelGGL(k_dd_final, dim3(rows), dim3(64), 0, h->stream, nblocks, part, outp);
    HIPCHK(h, hipGetLastError());
    std::vector<dd> host(rows);
    HIPCHK(h, hipMemcpyAsync(host.data(), outp, rows * sizeof(dd), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int r = 0; r < rows; r++) acc[r] += (long double)host[r].hi + (long double)host[r].lo;
    return PLK_OK;
}

/* per-site outputs leave the kernels as [row][site] planes (site fastest: coalesced stores); callers want
 * [site][row].  Transposed on the device through LDS tiles, then one contiguous copy to the host. */
__global__ __launch_bounds__(256) void k_transpose_rows(long rows, long n, const double *__restrict__ src, double *__restrict__ dst)
{
    __shared__ double tile[32][33];
    const long r0 = (long)blockIdx.y * 32, s0 = (long)blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < rows && s0 + tx < n) tile[j][tx] = src[(size_t)(r0 + j) * n + s0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (s0 + j < n && r0 + tx < rows) dst[(size_t)(s0 + j) * rows + r0 + tx] = tile[tx][j];
}

static int copy_site_rows(plk_engine *h, size_t rows, long n, long s0, const double *d_src, double *site_out)
{
    if (rows == 0 || n == 0) return PLK_OK;
    int rc;
    if ((rc = dev_reserve(h, &h->d_stage, &h->stage_cap, rows * (size_t)n))) return rc;
    hipLaunchKernelGGL(k_transpose_rows, dim3((unsigned)((n + 31) / 32), (unsigned)((rows + 31) / 32)), dim3(256), 0, h->stream,
                       (long)rows, n, d_src, h->d_stage);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(site_out + (size_t)s0 * rows, h->d_stage, rows * (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return PLK_OK;
}

template <int T>
static void launch_updown_mfma(plk_engine *h, const MUpArgs &a, const MDownProg &pg, unsigned grid, size_t lds, bool deriv, bool marg, bool nodes)
{
    const size_t lds_down = lds + (size_t)pg.nobs * MF_SITES;
    if (nodes)          /* the caller has checked that the depth-first down pass fits */
        hipLaunchKernelGGL((k_down_fused_mfma<T, false>), dim3(grid), dim3(MF_BLOCK), lds_down, h->stream, a, pg);
    else if (lds_down <= 60 * 1024 && (a.s0 % MF_SITES) == 0)
        hipLaunchKernelGGL((k_down_fused_mfma<T, true>), dim3(grid), dim3(MF_BLOCK), lds_down, h->stream, a, pg);
    else
        hipLaunchKernelGGL(k_down_store_mfma<T>, dim3(grid), dim3(MF_BLOCK), lds, h->stream, a);
    if (nodes) hipLaunchKernelGGL(k_up_nodes_mfma<T>, dim3(grid), dim3(MF_BLOCK), lds, h->stream, a);
    else if (deriv && marg) hipLaunchKernelGGL((k_up_mfma<T, true, true>), dim3(grid), dim3(MF_BLOCK), lds, h->stream, a);
    else if (deriv) hipLaunchKernelGGL((k_up_mfma<T, true, false>), dim3(grid), dim3(MF_BLOCK), lds, h->stream, a);
    else hipLaunchKernelGGL((k_up_mfma<T, false, true>), dim3(grid), dim3(MF_BLOCK), lds, h->stream, a);
}

/* deriv / marginal for 9 <= k <= 64 with compact codes: matrix-core kernels (plk_mfma_updown.h) */
static int run_updown_mfma(plk_engine *h, bool deriv, bool marg, const int *edge_mask, const int *node_mask,
                           double *site_out, double *sums_out, const double *d_M, int dzero)
{
    int rc;
    const int N = h->N, E = h->E, k = h->k, C = h->C;
    const int T = (k + 15) / 16, R = 4 * T, kk4 = (k + 3) / 4;
    const long S = h->S;
    if (h->prog_dirty) { if ((rc = build_program(h))) return rc; }
    const int ntips = (int)h->tip_edge.size();
    std::vector<int> edge_tip(E, -1), edge_int(E, -1), node_int(N, -1);
    for (int t = 0; t < ntips; t++) edge_tip[h->tip_edge[t]] = t;
    int nie = 0, nin = 0;
    for (int e = 0; e < E; e++) if (edge_tip[e] < 0) edge_int[e] = nie++;
    for (int a = 0; a < N; a++) if (h->indptr[a + 1] > h->indptr[a]) node_int[a] = nin++;
    std::vector<int> te = h->tip_edge;
    te.push_back(-1);
    std::vector<double> rwd((size_t)4 * R, 0.0);
    for (int i = 0; i < k; i++) rwd[(size_t)(i & 3) * R + (i >> 2)] = h->root_w[i];
    std::vector<int> node_scale(N, -1);
    int nsc = 0;
    for (int a = 0; a < N; a++) if (node_int[a] >= 0 && h->scale_node[a]) node_scale[a] = nsc++;

    /* derivative queries without marginals on the edge-at-a-time up pass: nodes whose children are all leaves are finished
     * inside their parent's visit (their F never goes through HBM), and the up pass rebuilds their L from the tip tables,
     * so the down pass does not store it either */
    const bool nodes_wanted = deriv && !marg && (h->opt_up_nodes & 1);
    std::vector<int> node_inl(N, 0);
    std::vector<char> skip_l(N, 0);
    bool any_inl = false;
    if (deriv && !marg && !nodes_wanted)
        for (int u = 1; u < N; u++) {
            const int b = h->preorder[u];
            if (plk_up_inlinable(h->indptr.data(), edge_tip.data(), b, false)) { node_inl[b] = 1; skip_l[b] = 1; any_inl = true; }
        }
    /* program of the depth-first down pass (k_down_fused_mfma) */
    PlkChain dch;
    plk_chain_build(N, h->pg, 2, h->indices.data(), node_int.data(), edge_int.data(), node_scale.data(), dch, any_inl ? skip_l.data() : nullptr);
    {
        const std::string bad = plk_chain_check(N, h->pg, dch, 2, INT_MAX, nin, nie, nsc, MF_SITES, (size_t)h->obs_nodes.size() * MF_SITES);
        if (!bad.empty()) { h->err = "internal: " + bad; return PLK_E_ARG; }
    }
    const std::vector<plk_op4> &dops = dch.ops;
    const int nslots_m = std::max(h->slots_needed, 1);
    auto cleanup = [&]() {};        /* everything below lives in grow-only engine buffers: no per-call hipMalloc / hipFree */
    if (h->node_has_data.size() != (size_t)N) h->node_has_data.assign(N, 1);
    /* PLK_OPT_UP_NODES = 1: derivative queries without marginals take the node-visit up pass over edge vectors only
     * (k_up_nodes_mfma), when the depth-first down pass applies (its staged code rows fit the LDS).  Measured equal to
     * k_up_mfma at BASELINE config 5 (32.1 against 31.7 ms per step: the down pass gains what the up pass loses), so
     * not the default; DESIGN.md section 4 has the counters */
    const bool nodes = deriv && !marg && (h->opt_up_nodes & 1) &&
                       (size_t)T * kk4 * 64 * sizeof(double) + h->obs_nodes.size() * (size_t)MF_SITES <= 60 * 1024;
    PlkUpNodes un;
    if (nodes) {
        plk_up_nodes_build(N, h->indptr.data(), h->indices.data(), h->preorder.data(), h->node_has_data.data(), edge_tip.data(),
                           node_int.data(), node_scale.data(), edge_mask, un);
        const std::string bad = plk_up_nodes_check(N, E, h->indptr.data(), h->indices.data(), un, nin, ntips, nsc);
        if (!bad.empty()) { h->err = "internal: " + bad; return PLK_E_ARG; }
    }
    /* the call's integer tables, one upload: [ops (16-byte aligned first)][edge_tip][edge_int][node_int][tip edges][node_scale]
     * [obs nodes][has_data][edge mask][node mask] */
    std::vector<int> pack;
    auto put = [&](const int *src, size_t n) { const size_t o = pack.size(); pack.insert(pack.end(), src, src + n); while (pack.size() & 3) pack.push_back(0); return o; };
    const size_t o_ops = put(reinterpret_cast<const int *>(dops.data()), dops.size() * 4);
    const size_t o_et = put(edge_tip.data(), (size_t)E), o_ei = put(edge_int.data(), (size_t)E), o_ni = put(node_int.data(), (size_t)N);
    const size_t o_te = put(te.data(), te.size()), o_ns = put(node_scale.data(), (size_t)N);
    const size_t o_obs = put(h->obs_nodes.data(), h->obs_nodes.size());
    std::vector<int> hd(h->node_has_data.begin(), h->node_has_data.end());
    const size_t o_has = put(hd.data(), (size_t)N);
    const size_t o_em = edge_mask ? put(edge_mask, (size_t)E) : 0, o_nm = node_mask ? put(node_mask, (size_t)N) : 0;
    const size_t o_vis = nodes ? put(un.rec.data(), un.rec.size()) : 0;
    const size_t o_inl = any_inl ? put(node_inl.data(), (size_t)N) : 0;
    const size_t nfr = (size_t)C * E * T * kk4 * 64, ntab = (size_t)C * (ntips + 1) * h->nchar * 4 * R;
    if ((rc = dev_reserve(h, &h->d_u4pack, &h->u4pack_cap, pack.size() + 4)) ||
        (rc = dev_reserve(h, &h->d_u4tip, &h->u4tip_cap, 2 * ntab + rwd.size())) ||
        (rc = dev_reserve(h, &h->d_uvmat, &h->uvmat_cap, 3 * nfr))) return rc;
    HIPCHK(h, hipMemcpyAsync(h->d_u4pack, pack.data(), pack.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_u4tip + 2 * ntab, rwd.data(), rwd.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));      /* pack and rwd are locals */
    int *pk = h->d_u4pack;
    const int4 *d_dops = reinterpret_cast<const int4 *>(pk + o_ops);
    const int *d_et = pk + o_et, *d_ei = pk + o_ei, *d_ni = pk + o_ni, *d_te = pk + o_te, *d_ns = pk + o_ns, *d_obsm = pk + o_obs, *d_has = pk + o_has;
    const int *d_emask = edge_mask ? pk + o_em : nullptr, *d_nmask = node_mask ? pk + o_nm : nullptr;
    double *d_fP = h->d_uvmat, *d_fPT = d_fP + nfr, *d_fD = d_fPT + nfr;
    double *d_tipd = h->d_u4tip, *d_dtip = d_tipd + ntab, *d_rwd = d_dtip + ntab;
    hipLaunchKernelGGL(k_build_frag_edges, dim3(C * E), dim3(256), 0, h->stream, k, T, kk4, 0, h->d_P, d_fP);
    hipLaunchKernelGGL(k_build_frag_edges, dim3(C * E), dim3(256), 0, h->stream, k, T, kk4, 1, h->d_P, d_fPT);
    hipLaunchKernelGGL(k_build_frag_edges, dim3(C * E), dim3(256), 0, h->stream, k, T, kk4, nodes ? 1 : 0, d_M, d_fD);   /* node visits apply M^T */
    hipLaunchKernelGGL(k_build_tip_dist, dim3(ntips + 1, C), dim3(256), 0, h->stream,
                       k, R, E, ntips, h->nchar, d_te, h->d_Pdd, h->d_defs, h->K, d_tipd);
    hipLaunchKernelGGL(k_build_dtip_dist, dim3(ntips + 1, C), dim3(256), 0, h->stream,
                       k, R, E, ntips, h->nchar, d_te, d_M, h->d_defs, h->K, d_dtip, dzero);
    if (hipGetLastError() != hipSuccess) { cleanup(); h->err = "plk_deriv/plk_marginal: table build failed"; return PLK_E_DEVICE; }

    /* site-summed marginals without per-site output: the up pass leaves per-wave sums (MVS) instead of the N k planes */
    const bool msum_only = marg && !site_out && sums_out && !deriv;
    const size_t per_site = ((size_t)(nie + 2 * (size_t)nin) * C * R * 4 + (size_t)nslots_m * R * 4 + (size_t)(nsc + 2) * C + 1 + (deriv ? E : 0) + (marg && !msum_only ? (size_t)N * k : 0)) * sizeof(double);
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    size_t budget = free_b > (size_t)(6ull << 30) ? free_b - (size_t)(4ull << 30) : free_b / 2;
    budget += h->work_cap * sizeof(double);
    long chunk = (long)std::min<size_t>((size_t)((S + MF_SITES - 1) / MF_SITES * MF_SITES), budget / (per_site + (msum_only ? ((size_t)N * h->k / 64 + 1) * sizeof(double) : 0)));
    if (h->opt_site_chunk > 0) chunk = std::min<long>(chunk, h->opt_site_chunk);
    chunk = chunk / MF_SITES * MF_SITES;
    if (chunk < MF_SITES) { cleanup(); h->err = "plk_deriv/plk_marginal: not enough device memory"; return PLK_E_NOMEM; }
    /* site-summed marginals: one weighted sum per wave, node and state (sized for a whole chunk) */
    const size_t mvs_doubles = msum_only ? (size_t)N * k * (size_t)((chunk + MF_SITES - 1) / MF_SITES) * (MF_BLOCK / 64) : 0;
    if ((rc = dev_reserve(h, &h->d_work, &h->work_cap, per_site / sizeof(double) * (size_t)chunk + mvs_doubles))) { cleanup(); return rc; }

    std::vector<long double> dsum(deriv ? E : 0, 0.0L), msum(marg ? (size_t)N * k : 0, 0.0L);
    for (long s0 = 0; s0 < S; s0 += chunk) {
        const long n = std::min(chunk, S - s0);
        const unsigned grid = (unsigned)((n + MF_SITES - 1) / MF_SITES);
        MUpArgs a;
        a.S = S; a.Spad = h->Spad; a.s0 = s0; a.n = n;
        a.N = N; a.E = E; a.k = k; a.kk4 = kk4; a.C = C; a.nchar = h->nchar; a.ntips = ntips; a.root_mode = h->root_mode;
        a.dzero = dzero;
        a.indptr = h->d_indptr; a.indices = h->d_indices; a.preorder = h->d_preorder; a.node_has_data = d_has;
        a.edge_tip = d_et; a.edge_int = d_ei; a.node_int = d_ni;
        a.fragP = d_fP; a.fragPT = d_fPT; a.fragD = d_fD; a.tip = d_tipd; a.dtip = d_dtip;
        a.codes = h->d_codes; a.cat_prior = h->d_cat_prior; a.root_wd = d_rwd; a.edge_mask = d_emask; a.node_mask = d_nmask;
        a.stride = (long)grid * MF_SITES * 4;
        double *p = h->d_work;
        a.EV = p; p += (size_t)nie * C * R * a.stride;
        a.LN = p; p += (size_t)nin * C * R * a.stride;
        a.FN = p; p += (size_t)nin * C * R * a.stride;
        a.node_scale = d_ns;
        MDownProg pg;
        pg.ops = d_dops; pg.nops = (int)h->ops.size(); pg.nobs = (int)h->obs_nodes.size(); pg.obs_nodes = d_obsm;
        pg.slots = p; p += (size_t)nslots_m * R * a.stride;
        a.SC = p; p += (size_t)nsc * C * n;
        a.CW = p; p += (size_t)C * n;
        a.XC = p; p += (size_t)C * n;
        a.LH = p; p += n;
        a.DV = p; if (deriv) p += (size_t)E * n;
        const size_t nwv = (size_t)grid * (MF_BLOCK / 64);
        a.MV = p; a.MVS = nullptr; a.wsite = nullptr;
        if (marg && msum_only) { a.MVS = p; a.wsite = h->d_w ? h->d_w + s0 : nullptr; p += (size_t)N * k * nwv; }
        else if (marg) p += (size_t)N * k * n;
        if (deriv && edge_mask) HIPCHK(h, hipMemsetAsync(a.DV, 0, (size_t)E * n * sizeof(double), h->stream));   /* without a mask every edge's row is written by the up pass */
        if (marg) HIPCHK(h, hipMemsetAsync(a.MV, 0, (msum_only ? (size_t)N * k * nwv : (size_t)N * k * n) * sizeof(double), h->stream));
        const size_t lds = (size_t)T * kk4 * 64 * sizeof(double);
        a.visits = nodes ? pk + o_vis : nullptr; a.nvisits = un.nvisits;
        a.node_inline = any_inl ? pk + o_inl : nullptr;
        const bool nv = nodes && (s0 % MF_SITES) == 0;       /* chunks start at multiples of the site tile */
        if (T == 1) launch_updown_mfma<1>(h, a, pg, grid, lds, deriv, marg, nv);
        else if (T == 2) launch_updown_mfma<2>(h, a, pg, grid, lds, deriv, marg, nv);
        else if (T == 3) launch_updown_mfma<3>(h, a, pg, grid, lds, deriv, marg, nv);
        else launch_updown_mfma<4>(h, a, pg, grid, lds, deriv, marg, nv);
        if (hipGetLastError() != hipSuccess) { cleanup(); h->err = "plk_deriv/plk_marginal: kernel launch failed"; return PLK_E_DEVICE; }
        if (sums_out) {
            const double *w = h->d_w ? h->d_w + s0 : nullptr;
            if (deriv && (rc = wsum_rows(h, E, n, a.DV, w, dsum.data()))) { cleanup(); return rc; }
            if (marg && msum_only) { if ((rc = wsum_rows(h, N * k, (long)nwv, a.MVS, nullptr, msum.data()))) { cleanup(); return rc; } }
            else if (marg && (rc = wsum_rows(h, N * k, n, a.MV, w, msum.data()))) { cleanup(); return rc; }
        }
        if (site_out && (rc = copy_site_rows(h, deriv ? (size_t)E : (size_t)N * k, n, s0, deriv ? a.DV : a.MV, site_out))) { cleanup(); return rc; }
    }
    hipError_t e = hipStreamSynchronize(h->stream);
    cleanup();
    if (e != hipSuccess) { h->err = std::string("plk_deriv/plk_marginal: ") + hipGetErrorString(e); return PLK_E_DEVICE; }
    if (sums_out) {
        const std::vector<long double> &src = deriv ? dsum : msum;
        for (size_t r = 0; r < src.size(); r++) {
            const double hi = (double)src[r];
            sums_out[2 * r] = hi;
            sums_out[2 * r + 1] = (double)(src[r] - (long double)hi);
        }
    }
    return PLK_OK;
}

/* deriv / marginal for k = 4 with compact codes: interleaved-vector kernels (plk_updown4.h) */
static int run_updown4(plk_engine *h, bool deriv, bool marg, const int *edge_mask, const int *node_mask,
                       double *site_out, double *sums_out, const double *d_M, int dzero, int nM)
{
    int rc;
    const int N = h->N, E = h->E, C = h->C;
    const int ER = nM * E;                 /* rows of the edge-form output: [form][edge] */
    const long S = h->S;
    if (h->prog_dirty) { if ((rc = build_program(h))) return rc; }
    const int ntips = (int)h->tip_edge.size();
    std::vector<int> edge_tip(E, -1), edge_int(E, -1), node_int(N, -1);
    for (int t = 0; t < ntips; t++) edge_tip[h->tip_edge[t]] = t;
    int nie = 0, nin = 0;
    for (int e = 0; e < E; e++) if (edge_tip[e] < 0) edge_int[e] = nie++;
    for (int a = 0; a < N; a++) if (h->indptr[a + 1] > h->indptr[a]) node_int[a] = nin++;
    if (nin == 0) { node_int[h->preorder[0]] = nin++; }      /* a single-node tree still has a root vector */
    std::vector<int> te = h->tip_edge;
    te.push_back(-1);
    std::vector<int> node_scale(N, -1);
    int nsc = 0;
    for (int a = 0; a < N; a++) if (node_int[a] >= 0 && h->scale_node[a]) node_scale[a] = nsc++;

    int *d_et = nullptr, *d_ei = nullptr, *d_ni = nullptr, *d_te = nullptr, *d_emask = nullptr, *d_nmask = nullptr;
    int *d_has = nullptr, *d_ns = nullptr, *d_oe2 = nullptr, *d_obs2 = nullptr;
    int4 *d_ops2 = nullptr;
    /* nodes the up pass handles inside their parent's visit (see Up4Args.node_inline) */
    std::vector<int> node_inline(N, 0);
    int *d_inl = nullptr;
    for (int p = 0; p < N; p++) {
        const int pdeg = h->indptr[p + 1] - h->indptr[p];
        if (pdeg < 1 || pdeg > 2) continue;
        for (int idx = h->indptr[p]; idx < h->indptr[p + 1]; idx++) {
            const int b = h->indices[idx];
            const int bdeg = h->indptr[b + 1] - h->indptr[b];
            if (bdeg < 1 || bdeg > 2) continue;
            bool all_leaves = true;
            for (int j = h->indptr[b]; j < h->indptr[b + 1]; j++) all_leaves = all_leaves && edge_tip[j] >= 0;
            if (all_leaves) node_inline[b] = 1;
        }
    }
    /* derivative queries proper (edge form dP, no marginals, at most 4 rate categories; the expectation queries keep k_up4,
     * whose several-forms-per-pass variant they must agree with bit for bit): node-visit up pass k_up4_nodes, which
     * keeps the vector of a continued child in registers for all categories; PLK_OPT_UP_NODES bit 1 switches it off */
    const bool nodes4 = deriv && !marg && nM == 1 && dzero && C <= 4 && E > 0 && (h->opt_up_nodes & 2);
    /* Pair messages (round 3): a node whose two children are leaves, without data of its own, sends its parent one of
     * nchar^2 vectors P_a (P_b B_b o P_c B_c).  With the node-visit up pass those come from a table (k_build_tables_pt, the
     * ll kernels' pair tables), so the down pass does not store L_a and the up pass does not read it: a third of the
     * down pass's writes and a fifth of the up pass's reads at BASELINE config 3. */
    std::vector<int> pair_of(N, -1), pair_tabs;
    std::vector<char> skip_l(N, 0);
    int npairs4 = 0;
    if (nodes4 && h->nchar <= 16 && h->opt_pair_tables) {
        if (h->node_has_data.size() != (size_t)N) h->node_has_data.assign(N, 1);
        std::vector<int> t_unit, t_edge, t_eb, t_ec;
        for (int b = 0; b < N; b++) {
            const int e0 = h->indptr[b];
            if (h->indptr[b + 1] - e0 != 2 || h->b_to_idx[b] < 0 || h->node_has_data[b]) continue;
            if (edge_tip[e0] < 0 || edge_tip[e0 + 1] < 0) continue;
            pair_of[b] = npairs4; skip_l[b] = 1;
            t_unit.push_back(npairs4 * h->nchar); t_edge.push_back(h->b_to_idx[b]); t_eb.push_back(e0); t_ec.push_back(e0 + 1);
            npairs4++;
        }
        pair_tabs = t_unit;
        pair_tabs.insert(pair_tabs.end(), t_edge.begin(), t_edge.end());
        pair_tabs.insert(pair_tabs.end(), t_eb.begin(), t_eb.end());
        pair_tabs.insert(pair_tabs.end(), t_ec.begin(), t_ec.end());
    }
    /* One level up: a node whose two children are leaves or pair nodes has L = (row of one table) o (row of another).  The
     * down pass does not store it and the node-visit up pass multiplies the two rows (plk_up_rebuild_table; PLK_OPT_UP_NODES
     * bit 2 switches it off). */
    std::vector<char> rebuild_n;
    std::vector<int> rebuild_tab;
    bool any_rebuild = false;
    if (nodes4 && !(h->opt_up_nodes & 4)) {
        if (h->node_has_data.size() != (size_t)N) h->node_has_data.assign(N, 1);
        plk_up_rebuild_table(N, h->indptr.data(), h->indices.data(), h->node_has_data.data(), edge_tip.data(), node_int.data(),
                             node_scale.data(), pair_of.data(), nin, rebuild_n, rebuild_tab);
        for (int b = 0; b < N; b++) if (rebuild_n[b]) { skip_l[b] = 1; any_rebuild = true; }
    }
    /* down-pass program: observation ops carry their staged code row and the (slot, row) of the next one */
    PlkChain ch2;
    plk_chain_build(N, h->pg, 1, h->indices.data(), node_int.data(), nullptr, node_scale.data(), ch2, npairs4 || any_rebuild ? skip_l.data() : nullptr);
    const int nobs2 = (int)h->obs_nodes.size();
    {
        const int D2 = h->slots_needed <= 4 ? 4 : (h->slots_needed <= 8 ? 8 : (h->slots_needed <= 16 ? 16 : INT_MAX));
        const std::string bad = plk_chain_check(N, h->pg, ch2, 1, D2, nin, nie, nsc, UD4_BLOCK, (size_t)nobs2 * UD4_BLOCK);
        if (!bad.empty()) { h->err = "internal: " + bad; return PLK_E_ARG; }
    }
    const std::vector<plk_op4> &ops2 = ch2.ops;
    const int first_slot2 = ch2.first_slot, first_row2 = ch2.first_row;
    PlkUpNodes un4;
    if (nodes4) {
        if (h->node_has_data.size() != (size_t)N) h->node_has_data.assign(N, 1);
        plk_up_nodes_build(N, h->indptr.data(), h->indices.data(), h->preorder.data(), h->node_has_data.data(), edge_tip.data(),
                           node_int.data(), node_scale.data(), edge_mask, un4, npairs4 ? pair_of.data() : nullptr,
                           any_rebuild ? rebuild_n.data() : nullptr, npairs4 > 0 && C == 1 && !(h->opt_up_nodes & 8));
        const std::string bad = plk_up_nodes_check(N, E, h->indptr.data(), h->indices.data(), un4, nin, ntips, nsc, npairs4, edge_tip.data(),
                                                   any_rebuild ? rebuild_tab.data() : nullptr, pair_of.data());
        if (!bad.empty()) { h->err = "internal: " + bad; return PLK_E_ARG; }
    }
    const int *d_vis4 = nullptr, *d_ptabs = nullptr, *d_rebuild = nullptr;
    double *d_tip4 = nullptr, *d_dtip4 = nullptr, *d_ptab4 = nullptr;
    size_t nptab = 0;
    auto cleanup = [&]() {};        /* everything below lives in grow-only engine buffers: no per-call hipMalloc / hipFree */
    const size_t ntab = (size_t)C * (ntips + 1) * h->nchar * 4;
    {
        /* all small integer tables of this call in one host block, one upload */
        if (h->node_has_data.size() != (size_t)N) h->node_has_data.assign(N, 1);
        std::vector<int> hd(h->node_has_data.begin(), h->node_has_data.end());
        std::vector<int> pack;
        auto put = [&](const int *src, size_t n) { const size_t off = pack.size(); pack.insert(pack.end(), src, src + n); while (pack.size() % 4) pack.push_back(0); return off; };
        const size_t o_ops = put(reinterpret_cast<const int *>(ops2.data()), ops2.size() * 4);
        const size_t o_et = put(edge_tip.data(), (size_t)E), o_ei = put(edge_int.data(), (size_t)E), o_ni = put(node_int.data(), (size_t)N);
        const size_t o_te = put(te.data(), te.size()), o_has = put(hd.data(), (size_t)N), o_ns = put(node_scale.data(), (size_t)N);
        const size_t o_oe = put(h->op_edge.data(), h->op_edge.size()), o_obs = put(h->obs_nodes.data(), h->obs_nodes.size());
        const size_t o_inl = put(node_inline.data(), (size_t)N);
        const size_t o_em = edge_mask && E > 0 ? put(edge_mask, (size_t)E) : 0, o_nm = node_mask ? put(node_mask, (size_t)N) : 0;
        const size_t o_vis = nodes4 ? put(un4.rec.data(), un4.rec.size()) : 0;
        const size_t o_pt = npairs4 ? put(pair_tabs.data(), pair_tabs.size()) : 0;
        const size_t o_rb = any_rebuild ? put(rebuild_tab.data(), rebuild_tab.size()) : 0;
        nptab = (size_t)C * npairs4 * h->nchar * h->nchar * 4;
        if ((rc = dev_reserve(h, &h->d_u4pack, &h->u4pack_cap, pack.size() + 4))) return rc;
        if ((rc = dev_reserve(h, &h->d_u4tip, &h->u4tip_cap, ntab * (size_t)(1 + nM) + nptab))) return rc;
        HIPCHK(h, hipMemcpyAsync(h->d_u4pack, pack.data(), pack.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));      /* pack is a local */
        int *b = h->d_u4pack;
        d_ops2 = reinterpret_cast<int4 *>(b + o_ops);
        d_et = b + o_et; d_ei = b + o_ei; d_ni = b + o_ni; d_te = b + o_te; d_has = b + o_has; d_ns = b + o_ns;
        d_oe2 = b + o_oe; d_obs2 = b + o_obs; d_inl = b + o_inl;
        d_emask = edge_mask && E > 0 ? b + o_em : nullptr;
        d_nmask = node_mask ? b + o_nm : nullptr;
        d_vis4 = nodes4 ? b + o_vis : nullptr;
        d_ptabs = npairs4 ? b + o_pt : nullptr;
        d_rebuild = any_rebuild ? b + o_rb : nullptr;
        d_tip4 = h->d_u4tip; d_dtip4 = h->d_u4tip + ntab; d_ptab4 = h->d_u4tip + ntab * (size_t)(1 + nM);
    }
    hipLaunchKernelGGL(k_build_tip, dim3(ntips + 1, C), dim3(64), 0, h->stream,
                       E, ntips + 1, h->nchar, d_te, h->d_Pdd, h->d_defs, d_tip4);
    for (int m = 0; m < nM; m++)
        hipLaunchKernelGGL(k_build_dtip4, dim3(ntips + 1, C), dim3(64), 0, h->stream,
                           E, ntips, h->nchar, d_te, d_M + (size_t)m * C * E * 16, h->d_defs, d_dtip4 + (size_t)m * ntab, dzero);
    if (npairs4)       /* pair tables: [C][npairs * nchar units][nchar][4], the ll kernels' builder on a list of pairs only */
        hipLaunchKernelGGL(k_build_tables_pt, dim3(npairs4, C), dim3(64), 0, h->stream,
                           E, npairs4, npairs4 * h->nchar, h->nchar, d_ptabs, h->d_Pdd, h->d_defs, d_ptab4);
    if (hipGetLastError() != hipSuccess) { cleanup(); h->err = "plk_deriv/plk_marginal: table build failed"; return PLK_E_DEVICE; }

    const bool msum_only = marg && !site_out && sums_out && !deriv;      /* per-wave sums instead of the N x 4 planes */
    const size_t per_site = ((size_t)(2 * (size_t)nin) * C * 4 + (size_t)(nsc + 2) * C + 1 + (deriv ? ER : 0) + (marg && !msum_only ? (size_t)N * 4 : 0)) * sizeof(double);
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    size_t budget = free_b > (size_t)(6ull << 30) ? free_b - (size_t)(4ull << 30) : free_b / 2;
    budget += h->work_cap * sizeof(double);
    long chunk = (long)std::min<size_t>((size_t)S, budget / (per_site + (msum_only ? ((size_t)N * h->k / 64 + 1) * sizeof(double) : 0)));
    if (h->opt_site_chunk > 0) chunk = std::min<long>(chunk, h->opt_site_chunk);
    if (chunk < 1) { cleanup(); h->err = "plk_deriv/plk_marginal: not enough device memory for one site"; return PLK_E_NOMEM; }
    if (chunk < S) chunk = std::max<long>(UD4_BLOCK, chunk / UD4_BLOCK * UD4_BLOCK);
    const size_t mvs_doubles = msum_only ? (size_t)N * 4 * (size_t)((chunk + UD4_BLOCK - 1) / UD4_BLOCK) * (UD4_BLOCK / 64) : 0;   /* per-wave marginal sums */
    if ((rc = dev_reserve(h, &h->d_work, &h->work_cap, per_site / sizeof(double) * (size_t)chunk + mvs_doubles))) { cleanup(); return rc; }

    std::vector<long double> dsum(deriv ? ER : 0, 0.0L), msum(marg ? (size_t)N * 4 : 0, 0.0L);
    for (long s0 = 0; s0 < S; s0 += chunk) {
        const long n = std::min(chunk, S - s0);
        Up4Args a;
        a.S = S; a.Spad = h->Spad; a.s0 = s0; a.n = n;
        a.N = N; a.E = E; a.C = C; a.nchar = h->nchar; a.ntips = ntips; a.root_mode = h->root_mode;
        a.dzero = dzero;
        a.indptr = h->d_indptr; a.indices = h->d_indices; a.preorder = h->d_preorder; a.node_has_data = d_has;
        a.edge_tip = d_et; a.edge_int = d_ei; a.node_int = d_ni; a.node_scale = d_ns; a.node_inline = d_inl;
        a.P = h->d_P; a.dP = d_M; a.tip = d_tip4; a.dtip = d_dtip4; a.nM = nM;
        a.codes = h->d_codes; a.cat_prior = h->d_cat_prior; a.root_w = h->d_root_w; a.edge_mask = d_emask; a.node_mask = d_nmask;
        double *p = h->d_work;
        a.LN = p; p += (size_t)nin * C * 4 * n;
        a.FN = p; p += (size_t)nin * C * 4 * n;
        a.SC = p; p += (size_t)nsc * C * n;
        a.CW = p; p += (size_t)C * n;
        a.XC = p; p += (size_t)C * n;
        a.LH = p; p += n;
        a.DV = p; if (deriv) p += (size_t)ER * n;
        const unsigned grid = (unsigned)((n + UD4_BLOCK - 1) / UD4_BLOCK);
        const size_t nwv = (size_t)grid * (UD4_BLOCK / 64);
        a.MV = p; a.MVS = nullptr; a.wsite = nullptr;
        if (marg && msum_only) { a.MVS = p; a.wsite = h->d_w ? h->d_w + s0 : nullptr; p += (size_t)N * 4 * nwv; }
        else if (marg) p += (size_t)N * 4 * n;
        a.visits = d_vis4; a.nvisits = un4.nvisits; a.ptab = npairs4 ? d_ptab4 : nullptr; a.npairs = npairs4; a.rebuild = d_rebuild;
        /* (the node-visit pass writes the row of every wanted edge: rows need clearing only under a mask) */
        if (deriv && E > 0 && !(nodes4 && !edge_mask)) HIPCHK(h, hipMemsetAsync(a.DV, 0, (size_t)ER * n * sizeof(double), h->stream));
        if (marg) HIPCHK(h, hipMemsetAsync(a.MV, 0, (msum_only ? (size_t)N * 4 * nwv : (size_t)N * 4 * n) * sizeof(double), h->stream));
        const size_t lds_codes = (size_t)nobs2 * UD4_BLOCK;
        const bool fused_ok = d_ops2 && lds_codes <= 60 * 1024 && (s0 % UD4_BLOCK) == 0;
        if (fused_ok && h->slots_needed <= 4) hipLaunchKernelGGL(k_down_fused4<4>, dim3(grid), dim3(UD4_BLOCK), lds_codes, h->stream, a, d_ops2, d_oe2, (int)h->ops.size(), d_obs2, nobs2, first_slot2, first_row2);
        else if (fused_ok && h->slots_needed <= 8) hipLaunchKernelGGL(k_down_fused4<8>, dim3(grid), dim3(UD4_BLOCK), lds_codes, h->stream, a, d_ops2, d_oe2, (int)h->ops.size(), d_obs2, nobs2, first_slot2, first_row2);
        else if (fused_ok && h->slots_needed <= 16) hipLaunchKernelGGL(k_down_fused4<16>, dim3(grid), dim3(UD4_BLOCK), lds_codes, h->stream, a, d_ops2, d_oe2, (int)h->ops.size(), d_obs2, nobs2, first_slot2, first_row2);
        else hipLaunchKernelGGL(k_down_store4, dim3(grid), dim3(UD4_BLOCK), 0, h->stream, a);
        if (nodes4 && C == 1) hipLaunchKernelGGL(k_up4_nodes<1>, dim3(grid), dim3(UD4_BLOCK), 0, h->stream, a);
        else if (nodes4) hipLaunchKernelGGL(k_up4_nodes<4>, dim3(grid), dim3(UD4_BLOCK), 0, h->stream, a);
        else if (deriv && marg) hipLaunchKernelGGL((k_up4<true, true>), dim3(grid), dim3(UD4_BLOCK), 0, h->stream, a);
        else if (deriv && nM == 2) hipLaunchKernelGGL((k_up4<true, false, 2>), dim3(grid), dim3(UD4_BLOCK), 0, h->stream, a);
        else if (deriv && nM == 3) hipLaunchKernelGGL((k_up4<true, false, 3>), dim3(grid), dim3(UD4_BLOCK), 0, h->stream, a);
        else if (deriv && nM == 4) hipLaunchKernelGGL((k_up4<true, false, 4>), dim3(grid), dim3(UD4_BLOCK), 0, h->stream, a);
        else if (deriv) hipLaunchKernelGGL((k_up4<true, false>), dim3(grid), dim3(UD4_BLOCK), 0, h->stream, a);
        else hipLaunchKernelGGL((k_up4<false, true>), dim3(grid), dim3(UD4_BLOCK), 0, h->stream, a);
        if (hipGetLastError() != hipSuccess) { cleanup(); h->err = "plk_deriv/plk_marginal: kernel launch failed"; return PLK_E_DEVICE; }
        if (sums_out) {
            const double *w = h->d_w ? h->d_w + s0 : nullptr;
            if (deriv && E > 0 && (rc = wsum_rows(h, ER, n, a.DV, w, dsum.data()))) { cleanup(); return rc; }
            if (marg && msum_only) { if ((rc = wsum_rows(h, N * 4, (long)nwv, a.MVS, nullptr, msum.data()))) { cleanup(); return rc; } }
            else if (marg && (rc = wsum_rows(h, N * 4, n, a.MV, w, msum.data()))) { cleanup(); return rc; }
        }
        if (site_out && (rc = copy_site_rows(h, deriv ? (size_t)ER : (size_t)N * 4, n, s0, deriv ? a.DV : a.MV, site_out))) { cleanup(); return rc; }
    }
    hipError_t e = hipStreamSynchronize(h->stream);
    cleanup();
    if (e != hipSuccess) { h->err = std::string("plk_deriv/plk_marginal: ") + hipGetErrorString(e); return PLK_E_DEVICE; }
    if (sums_out) {
        const std::vector<long double> &src = deriv ? dsum : msum;
        for (size_t r = 0; r < src.size(); r++) {
            const double hi = (double)src[r];
            sums_out[2 * r] = hi;
            sums_out[2 * r + 1] = (double)(src[r] - (long double)hi);
        }
    }
    return PLK_OK;
}

/* deriv / marginal / edge expectations for 9 <= k <= 20 with compact codes: register-resident vector kernels
 * (plk_updown_vec.h) */
static bool use_updown_vec(const plk_engine *h)
{
    return !h->opt_force_generic && h->opt_mfma == 1 && h->pat_mode == 1 && h->k >= 9 && h->k <= 20 && h->E > 0;
}

template <int K>
static void launch_updown_vec(plk_engine *h, const UpVecArgs &a, const int *d_obs, unsigned grid, bool deriv, bool marg)
{
    hipLaunchKernelGGL(k_down_vec<K>, dim3(grid), dim3(UDV_BLOCK), 0, h->stream, a, d_obs);
    if (deriv && marg) hipLaunchKernelGGL((k_up_vec<K, true, true>), dim3(grid), dim3(UDV_BLOCK), 0, h->stream, a);
    else if (deriv) hipLaunchKernelGGL((k_up_vec<K, true, false>), dim3(grid), dim3(UDV_BLOCK), 0, h->stream, a);
    else hipLaunchKernelGGL((k_up_vec<K, false, true>), dim3(grid), dim3(UDV_BLOCK), 0, h->stream, a);
}

static int run_updown_vec(plk_engine *h, bool deriv, bool marg, const int *edge_mask, const int *node_mask,
                          double *site_out, double *sums_out, const double *d_M, int dzero)
{
    int rc;
    const int N = h->N, E = h->E, k = h->k, K = h->K, C = h->C;
    const long S = h->S;
    if (h->prog_dirty) { if ((rc = build_program(h))) return rc; }
    const int ntips = (int)h->tip_edge.size();
    std::vector<int> edge_tip(E, -1), edge_int(E, -1), node_int(N, -1), node_scale(N, -1);
    for (int t = 0; t < ntips; t++) edge_tip[h->tip_edge[t]] = t;
    int nie = 0, nin = 0, nsc = 0;
    for (int e = 0; e < E; e++) if (edge_tip[e] < 0) edge_int[e] = nie++;
    for (int a = 0; a < N; a++) if (h->indptr[a + 1] > h->indptr[a]) node_int[a] = nin++;
    for (int a = 0; a < N; a++) if (node_int[a] >= 0 && h->scale_node[a]) node_scale[a] = nsc++;
    /* down-pass program and up-pass visit records + matrix list, both checked before anything is launched */
    /* nodes finished inside their parent's visit (no marginals asked for): the up pass rebuilds their L vector from the
     * tip tables, so the down pass does not store it (a third of the stored vectors at BASELINE config 4) */
    std::vector<char> skip_l(N, 0);
    const char *leaf_obs = h->node_observed.size() == (size_t)N ? h->node_observed.data() : nullptr;
    for (int u = 1; u < N; u++) { const int b = h->preorder[u]; skip_l[b] = plk_up_inlinable(h->indptr.data(), edge_tip.data(), b, marg, h->indices.data(), leaf_obs); }
    PlkChain ch;
    plk_chain_build(N, h->pg, 3, h->indices.data(), node_int.data(), nullptr, node_scale.data(), ch, skip_l.data());
    PlkUpVisits uv;
    plk_up_visits_build(N, h->indptr.data(), h->indices.data(), h->preorder.data(), h->node_has_data.data(), edge_tip.data(),
                        node_int.data(), node_scale.data(), deriv, marg, edge_mask, node_mask, uv, leaf_obs);
    {
        std::string bad = plk_chain_check(N, h->pg, ch, 3, INT_MAX, nin, nie, nsc, 0, 0);
        if (bad.empty()) bad = plk_up_visits_check(N, E, uv, nin, ntips, nsc, deriv);
        if (!bad.empty()) { h->err = "internal: " + bad; return PLK_E_ARG; }
    }
    const int nstream = (int)uv.kind.size();
    const int nslots = std::max(h->slots_needed, 1);
    std::vector<int> te = h->tip_edge;
    te.push_back(-1);
    /* integer tables of this call: one upload into the grow-only block */
    std::vector<int> pack;
    auto put = [&](const int *src, size_t n) { const size_t off = pack.size(); pack.insert(pack.end(), src, src + n); while (pack.size() % 4) pack.push_back(0); return off; };
    const size_t o_ops = put(reinterpret_cast<const int *>(ch.ops.data()), ch.ops.size() * 4);
    const size_t o_obs = put(h->obs_nodes.data(), h->obs_nodes.size());
    const size_t o_vis = put(uv.rec.data(), uv.rec.size());
    const size_t o_kind = put(uv.kind.data(), uv.kind.size()), o_edge = put(uv.edge.data(), uv.edge.size());
    const size_t o_te = put(te.data(), te.size());
    /* the state a character code observes: its definition row is one 1.0 among zeros (leaf marginals of such codes are one
     * dot product in the up pass instead of a matrix-vector product) */
    std::vector<int> code_state((size_t)std::max(h->nchar, 1), -1);
    for (int cd = 0; cd < h->nchar && h->defs.size() == (size_t)h->nchar * k; cd++) {
        int nnz = 0, at = -1;
        for (int j = 0; j < k; j++) if (h->defs[(size_t)cd * k + j] != 0.0) { nnz++; at = j; }
        if (nnz == 1 && h->defs[(size_t)cd * k + at] == 1.0) code_state[cd] = at;
    }
    const size_t o_cs = put(code_state.data(), code_state.size());
    const size_t ntab = (size_t)C * (ntips + 1) * h->nchar * K;
    const size_t nmat = (size_t)C * E * K * K + (size_t)C * (nstream + 3) * K * K;
    if ((rc = dev_reserve(h, &h->d_u4pack, &h->u4pack_cap, pack.size() + 4)) ||
        (rc = dev_reserve(h, &h->d_u4tip, &h->u4tip_cap, 2 * ntab)) ||
        (rc = dev_reserve(h, &h->d_uvmat, &h->uvmat_cap, nmat))) return rc;
    HIPCHK(h, hipMemcpyAsync(h->d_u4pack, pack.data(), pack.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));      /* pack is a local */
    const int *b = h->d_u4pack;
    double *d_PT = h->d_uvmat, *d_MS = h->d_uvmat + (size_t)C * E * K * K;
    double *d_tipv = h->d_u4tip, *d_dtipv = h->d_u4tip + ntab;
    const int bt = K * K >= 256 ? 256 : 64;
    hipLaunchKernelGGL(k_build_edge_stream, dim3(C * E), dim3(bt), 0, h->stream, k, K, 0, h->d_P, d_PT);
    hipLaunchKernelGGL(k_build_up_stream, dim3(nstream + 3, C), dim3(bt), 0, h->stream, k, K, E, nstream, b + o_kind, b + o_edge,
                       h->d_P, d_M, d_MS);
    hipLaunchKernelGGL(k_build_tip_vec, dim3(ntips + 1, C), dim3(256), 0, h->stream,
                       k, K, E, ntips, h->nchar, b + o_te, h->d_Pdd, h->d_defs, d_tipv);
    hipLaunchKernelGGL(k_build_dtip_vec, dim3(ntips + 1, C), dim3(256), 0, h->stream,
                       k, K, E, ntips, h->nchar, b + o_te, d_M, h->d_defs, K, d_dtipv, dzero);
    if (hipGetLastError() != hipSuccess) { h->err = "plk_deriv/plk_marginal: table build failed"; return PLK_E_DEVICE; }

    const bool msum_only = marg && !site_out && sums_out && !deriv;      /* per-wave sums instead of the N x k planes */
    const size_t per_site = ((size_t)(2 * (size_t)nin) * C * K + (size_t)nslots * K + (size_t)(nsc + 2) * C + 1 + (deriv ? E : 0) + (marg && !msum_only ? (size_t)N * k : 0)) * sizeof(double);
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    size_t budget = free_b > (size_t)(6ull << 30) ? free_b - (size_t)(4ull << 30) : free_b / 2;
    budget += h->work_cap * sizeof(double);
    long chunk = (long)std::min<size_t>((size_t)S, budget / (per_site + (msum_only ? ((size_t)N * h->k / 64 + 1) * sizeof(double) : 0)));
    if (h->opt_site_chunk > 0) chunk = std::min<long>(chunk, h->opt_site_chunk);
    if (chunk < 1) { h->err = "plk_deriv/plk_marginal: not enough device memory for one site"; return PLK_E_NOMEM; }
    if (chunk < S) chunk = std::max<long>(UDV_BLOCK, chunk / UDV_BLOCK * UDV_BLOCK);
    const size_t mvs_doubles = msum_only ? (size_t)N * k * (size_t)((chunk + UDV_BLOCK - 1) / UDV_BLOCK) * (UDV_BLOCK / 64) : 0;   /* per-wave marginal sums */
    if ((rc = dev_reserve(h, &h->d_work, &h->work_cap, per_site / sizeof(double) * (size_t)chunk + mvs_doubles))) return rc;

    std::vector<long double> dsum(deriv ? E : 0, 0.0L), msum(marg ? (size_t)N * k : 0, 0.0L);
    for (long s0 = 0; s0 < S; s0 += chunk) {
        const long n = std::min(chunk, S - s0);
        UpVecArgs a;
        a.S = S; a.Spad = h->Spad; a.s0 = s0; a.n = n;
        a.N = N; a.E = E; a.k = k; a.C = C; a.nchar = h->nchar; a.ntips = ntips; a.root_mode = h->root_mode; a.dzero = dzero;
        a.ops = reinterpret_cast<const int4 *>(b + o_ops); a.nops = (int)h->ops.size(); a.root_int = node_int[h->preorder[0]];
        a.first_slot = ch.first_slot; a.first_row = ch.first_row; a.second_row = ch.second_row;
        a.PT = d_PT; a.tip = d_tipv; a.dtip = d_dtipv; a.codes = h->d_codes; a.cat_prior = h->d_cat_prior; a.root_w = h->d_root_w;
        a.visits = b + o_vis; a.nvisits = uv.nvisits; a.MS = d_MS; a.nstream = nstream;
        a.reg_lo = vec_reg_window(h);
        a.code_state = b + o_cs;
        double *p = h->d_work;
        a.LN = p; p += (size_t)nin * C * K * n;
        a.FN = p; p += (size_t)nin * C * K * n;
        a.slots = p; p += (size_t)nslots * K * n;
        a.SC = p; p += (size_t)nsc * C * n;
        a.CW = p; p += (size_t)C * n;
        a.XC = p; p += (size_t)C * n;
        a.LH = p; p += n;
        a.DV = p; if (deriv) p += (size_t)E * n;
        const unsigned grid = (unsigned)((n + UDV_BLOCK - 1) / UDV_BLOCK);
        const size_t nwv = (size_t)grid * (UDV_BLOCK / 64);
        a.MV = p; a.MVS = nullptr; a.wsite = nullptr;
        if (marg && msum_only) { a.MVS = p; a.wsite = h->d_w ? h->d_w + s0 : nullptr; p += (size_t)N * k * nwv; }
        else if (marg) p += (size_t)N * k * n;
        if (deriv && edge_mask) HIPCHK(h, hipMemsetAsync(a.DV, 0, (size_t)E * n * sizeof(double), h->stream));   /* without a mask every edge's row is written by the up pass */
        if (marg) HIPCHK(h, hipMemsetAsync(a.MV, 0, (msum_only ? (size_t)N * k * nwv : (size_t)N * k * n) * sizeof(double), h->stream));
        if (K == 16) launch_updown_vec<16>(h, a, b + o_obs, grid, deriv, marg);
        else launch_updown_vec<20>(h, a, b + o_obs, grid, deriv, marg);
        if (hipGetLastError() != hipSuccess) { h->err = "plk_deriv/plk_marginal: kernel launch failed"; return PLK_E_DEVICE; }
        if (sums_out) {
            const double *w = h->d_w ? h->d_w + s0 : nullptr;
            if (deriv && (rc = wsum_rows(h, E, n, a.DV, w, dsum.data()))) return rc;
            if (marg && msum_only) { if ((rc = wsum_rows(h, N * k, (long)nwv, a.MVS, nullptr, msum.data()))) return rc; }
            else if (marg && (rc = wsum_rows(h, N * k, n, a.MV, w, msum.data()))) return rc;
        }
        if (site_out && (rc = copy_site_rows(h, deriv ? (size_t)E : (size_t)N * k, n, s0, deriv ? a.DV : a.MV, site_out))) return rc;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (sums_out) {
        const std::vector<long double> &src = deriv ? dsum : msum;
        for (size_t r = 0; r < src.size(); r++) {
            const double hi = (double)src[r];
            sums_out[2 * r] = hi;
            sums_out[2 * r + 1] = (double)(src[r] - (long double)hi);
        }
    }
    return PLK_OK;
}

static bool use_updown4(const plk_engine *h)
{
    return !h->opt_force_generic && h->k == 4 && h->pat_mode == 1 && h->E > 0;
}

/* d_M: the per-(category, edge) matrices of the edge bilinear form fe^T M L_b: dP for the derivative
 * (dzero = 1: rows sum to zero), scaled Frechet matrices for dwell / trans / em-update (dzero = 0) */
static int run_updown(plk_engine *h, bool deriv, bool marg, const int *edge_mask, const int *node_mask,
                      double *site_out, double *sums_out, const double *d_M_in = nullptr, int dzero = 1, int nM = 1)
{
    if (h->k == 0 || h->pat_mode == 0) { h->err = "plk_deriv/plk_marginal: tree, model and patterns must be set"; return PLK_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    if (h->model_dirty) { if ((rc = run_expm(h))) return rc; }
    if (!d_M_in && (rc = ensure_dP(h))) return rc;
    const double *d_M = d_M_in ? d_M_in : h->d_dP;
    if (use_updown_vec(h) && nM == 1) return run_updown_vec(h, deriv, marg, edge_mask, node_mask, site_out, sums_out, d_M, dzero);
    if (use_mfma(h)) return run_updown_mfma(h, deriv, marg, edge_mask, node_mask, site_out, sums_out, d_M, dzero);
    if (use_updown4(h)) return run_updown4(h, deriv, marg, edge_mask, node_mask, site_out, sums_out, d_M, dzero, nM);
    if (nM != 1) { h->err = "internal: several edge forms per pass need the k = 4 kernels"; return PLK_E_ARG; }
    const int N = h->N, E = h->E, k = h->k, K = h->K, C = h->C;
    const long S = h->S;
    /* padded edge-indexed streams */
    double *d_PT = nullptr, *d_PN = nullptr, *d_DT = nullptr;
    int *d_emask = nullptr, *d_nmask = nullptr;
    int *d_has = nullptr, *d_ns = nullptr;
    const size_t strm = (size_t)C * E * K * K;
    auto cleanup = [&]() {
        if (d_PT) (void)hipFree(d_PT);
        if (d_PN) (void)hipFree(d_PN);
        if (d_DT) (void)hipFree(d_DT);
        if (d_emask) (void)hipFree(d_emask);
        if (d_nmask) (void)hipFree(d_nmask);
        if (d_has) (void)hipFree(d_has);
        if (d_ns) (void)hipFree(d_ns);
    };
    if ((rc = dev_alloc(h, &d_PT, strm)) || (rc = dev_alloc(h, &d_PN, strm)) || (rc = dev_alloc(h, &d_DT, strm))) { cleanup(); return rc; }
    const int bt = K * K >= 256 ? 256 : 64;
    hipLaunchKernelGGL(k_build_edge_stream, dim3(C * E), dim3(bt), 0, h->stream, k, K, 0, h->d_P, d_PT);
    hipLaunchKernelGGL(k_build_edge_stream, dim3(C * E), dim3(bt), 0, h->stream, k, K, 1, h->d_P, d_PN);
    hipLaunchKernelGGL(k_build_edge_stream, dim3(C * E), dim3(bt), 0, h->stream, k, K, 0, d_M, d_DT);
    if (edge_mask && (rc = dev_upload(h, &d_emask, edge_mask, (size_t)E))) { cleanup(); return rc; }
    if (node_mask && (rc = dev_upload(h, &d_nmask, node_mask, (size_t)N))) { cleanup(); return rc; }
    if (h->node_has_data.size() != (size_t)N) h->node_has_data.assign(N, 1);
    { std::vector<int> hd(h->node_has_data.begin(), h->node_has_data.end()); if ((rc = dev_upload(h, &d_has, hd.data(), (size_t)N))) { cleanup(); return rc; } }

    /* chunk the site axis so that the stored vectors fit */
    if (h->prog_dirty) { if ((rc = build_program(h))) { cleanup(); return rc; } }
    std::vector<int> node_scale(N, -1);
    int nsc = 0;
    for (int a = 0; a < N; a++) if (h->indptr[a + 1] > h->indptr[a] && h->scale_node[a]) node_scale[a] = nsc++;
    if ((rc = dev_upload(h, &d_ns, node_scale.data(), (size_t)N))) { cleanup(); return rc; }
    const size_t per_site = ((size_t)(E + 2 * (size_t)N) * C * k + (size_t)(nsc + 2) * C + 1 + (deriv ? E : 0) + (marg ? (size_t)N * k : 0)) * sizeof(double);
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    size_t budget = free_b > (size_t)(6ull << 30) ? free_b - (size_t)(4ull << 30) : free_b / 2;
    budget += h->work_cap * sizeof(double);
    long chunk = (long)std::min<size_t>((size_t)S, budget / per_site);
    if (h->opt_site_chunk > 0) chunk = std::min<long>(chunk, h->opt_site_chunk);
    if (chunk < 1) { cleanup(); h->err = "plk_deriv/plk_marginal: not enough device memory for one site"; return PLK_E_NOMEM; }
    if (chunk < S) chunk = std::max<long>(GEN_BLOCK, chunk / GEN_BLOCK * GEN_BLOCK);
    if ((rc = dev_reserve(h, &h->d_work, &h->work_cap, per_site / sizeof(double) * (size_t)chunk))) { cleanup(); return rc; }

    std::vector<long double> dsum(deriv ? E : 0, 0.0L), msum(marg ? (size_t)N * k : 0, 0.0L);
    for (long s0 = 0; s0 < S; s0 += chunk) {
        const long n = std::min(chunk, S - s0);
        UpArgs a;
        a.S = S; a.Spad = h->Spad; a.s0 = s0; a.n = n;
        a.N = N; a.E = E; a.k = k; a.C = C; a.nchar = h->nchar; a.pat_mode = h->pat_mode; a.root_mode = h->root_mode;
        a.dzero = dzero;
        a.mod_edge = -1; a.DN = nullptr; a.D2T = nullptr; a.LHdiv = nullptr; a.XM = nullptr; a.XMdiv = nullptr;
        a.node_scale = d_ns;
        a.indptr = h->d_indptr; a.indices = h->d_indices; a.preorder = h->d_preorder; a.node_has_data = d_has;
        a.PT = d_PT; a.PN = d_PN; a.DT = d_DT; a.codes = h->d_codes; a.defs = h->d_defs; a.B = h->d_B;
        a.cat_prior = h->d_cat_prior; a.root_w = h->d_root_w; a.edge_mask = d_emask; a.node_mask = d_nmask;
        double *p = h->d_work;
        a.EV = p; p += (size_t)E * C * k * n;
        a.LN = p; p += (size_t)N * C * k * n;
        a.FN = p; p += (size_t)N * C * k * n;
        a.SC = p; p += (size_t)nsc * C * n;
        a.CW = p; p += (size_t)C * n;
        a.XC = p; p += (size_t)C * n;
        a.LH = p; p += n;
        a.DV = p; if (deriv) p += (size_t)E * n;
        a.MV = p; if (marg) p += (size_t)N * k * n;
        if (deriv) HIPCHKC(h, hipMemsetAsync(a.DV, 0, (size_t)E * n * sizeof(double), h->stream));
        if (marg) HIPCHKC(h, hipMemsetAsync(a.MV, 0, (size_t)N * k * n * sizeof(double), h->stream));
        const unsigned grid = (unsigned)((n + GEN_BLOCK - 1) / GEN_BLOCK);
        switch (K) {
        case 2: launch_updown<2>(h, a, grid, deriv, marg); break;
        case 4: launch_updown<4>(h, a, grid, deriv, marg); break;
        case 8: launch_updown<8>(h, a, grid, deriv, marg); break;
        case 16: launch_updown<16>(h, a, grid, deriv, marg); break;
        case 20: launch_updown<20>(h, a, grid, deriv, marg); break;
        case 32: launch_updown<32>(h, a, grid, deriv, marg); break;
        case 61: launch_updown<61>(h, a, grid, deriv, marg); break;
        default: launch_updown<64>(h, a, grid, deriv, marg); break;
        }
        if (hipGetLastError() != hipSuccess) { cleanup(); h->err = "plk_deriv/plk_marginal: kernel launch failed"; return PLK_E_DEVICE; }
        if (sums_out) {
            const double *w = h->d_w ? h->d_w + s0 : nullptr;
            if (deriv && (rc = wsum_rows(h, E, n, a.DV, w, dsum.data()))) { cleanup(); return rc; }
            if (marg && (rc = wsum_rows(h, N * k, n, a.MV, w, msum.data()))) { cleanup(); return rc; }
        }
        if (site_out && (rc = copy_site_rows(h, deriv ? (size_t)E : (size_t)N * k, n, s0, deriv ? a.DV : a.MV, site_out))) { cleanup(); return rc; }
    }
    hipError_t e = hipStreamSynchronize(h->stream);
    cleanup();
    if (e != hipSuccess) { h->err = std::string("plk_deriv/plk_marginal: ") + hipGetErrorString(e); return PLK_E_DEVICE; }
    if (sums_out) {
        const std::vector<long double> &src = deriv ? dsum : msum;
        for (size_t r = 0; r < src.size(); r++) {
            const double hi = (double)src[r];
            sums_out[2 * r] = hi;
            sums_out[2 * r + 1] = (double)(src[r] - (long double)hi);
        }
    }
    return PLK_OK;
}

/* ====================================================================== */
/* X1: the reduction step over ranks on RCCL (one process per GPU)         */
/* ====================================================================== */

extern "C" int plk_comm_available(void) { return rccl_load() ? 1 : 0; }

extern "C" int plk_comm_unique_id(unsigned char id_out[128])
{
    if (!id_out) return PLK_E_ARG;
    if (!rccl_load()) { g_create_error = g_rccl.err; return PLK_E_UNSUPPORTED; }
    PlkNcclId id;
    const int rc = g_rccl.GetUniqueId(&id);
    if (rc) { g_create_error = "ncclGetUniqueId: " + rccl_text(rc); return PLK_E_DEVICE; }
    memcpy(id_out, id.internal, 128);
    return PLK_OK;
}

extern "C" int plk_comm_init(plk_engine *h, int nranks, int rank, const unsigned char id[128])
{
    if (!plk_live(h) || !id || nranks < 1 || rank < 0 || rank >= nranks) return PLK_E_ARG;
    if (!rccl_load()) { h->err = g_rccl.err; return PLK_E_UNSUPPORTED; }
    HIPCHK(h, hipSetDevice(h->device));
    if (h->comm) { (void)g_rccl.CommDestroy(h->comm); h->comm = nullptr; }
    PlkNcclId nid;
    memcpy(nid.internal, id, 128);
    const int rc = g_rccl.CommInitRank(&h->comm, nranks, nid, rank);
    if (rc) { h->comm = nullptr; h->err = "ncclCommInitRank: " + rccl_text(rc); return PLK_E_DEVICE; }
    h->comm_ranks = nranks;
    return PLK_OK;
}

extern "C" int plk_allreduce_sum_async(plk_engine *h, double *dev, long count)
{
    if (!plk_live(h) || !dev || count < 1) return PLK_E_ARG;
    if (!h->comm) { h->err = "plk_allreduce_sum_async: plk_comm_init first"; return PLK_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    const int rc = g_rccl.AllReduce(dev, dev, (size_t)count, /* ncclDouble */ 8, /* ncclSum */ 0, h->comm, h->stream);
    if (rc) { h->err = "ncclAllReduce: " + rccl_text(rc); return PLK_E_DEVICE; }
    return PLK_OK;
}

extern "C" int plk_comm_destroy(plk_engine *h)
{
    if (!plk_live(h)) return PLK_E_ARG;
    if (h->comm) {
        (void)hipSetDevice(h->device);
        (void)hipStreamSynchronize(h->stream);
        (void)g_rccl.CommDestroy(h->comm);
        h->comm = nullptr;
    }
    return PLK_OK;
}

extern "C" int plk_deriv(plk_engine *h, const int *edge_mask, double *site_edge_out, double *edge_sums_out)
{
    if (!plk_live(h)) return PLK_E_ARG;
    return run_updown(h, true, false, edge_mask, nullptr, site_edge_out, edge_sums_out);
}

/* Several direction matrices at once: the k = 4 kernels carry up to four edge forms per pass (one down pass, one
 * up pass, shared forward vectors); other state counts run one pass per direction.
 * site_out: NULL or [S][nL][E]; sums_out: NULL or [nL][E][2]. */
extern "C" int plk_edge_expect_multi(plk_engine *h, int nL, const double *L_hi, const double *L_lo, int coef_mode,
                                     const int *edge_mask, double *site_out, double *sums_out)
{
    if (!plk_live(h)) return PLK_E_ARG;
    if (h->k == 0 || h->pat_mode == 0) { h->err = "plk_edge_expect: tree, model and patterns must be set"; return PLK_E_ARG; }
    if (!L_hi || nL < 1 || coef_mode < PLK_COEF_PRIOR || coef_mode > PLK_COEF_PRIOR_RATE) { h->err = "plk_edge_expect: bad direction matrix or coefficient mode"; return PLK_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    if (h->model_dirty) { if ((rc = run_expm(h))) return rc; }
    const int k = h->k, C = h->C, E = h->E;
    const long S = h->S;
    if (E == 0) return PLK_OK;
    const size_t kk = (size_t)k * k, n2 = 4 * kk;
    /* how many directions one pass of the kernels can carry */
    const int per_pass = use_updown4(h) && !use_mfma(h) ? 4 : 1;
    /* grow-only engine buffers: no per-call hipMalloc / hipFree */
    auto cleanup = [&]() {};
    const int threads = n2 >= 1024 ? 1024 : (n2 >= 256 ? 256 : 64);
    int use_lds;
    const size_t lds_bytes = expm_lds_bytes((size_t)2 * k, threads, &use_lds);
    if ((rc = dev_reserve(h, &h->d_exL, &h->exL_cap, (size_t)per_pass * 2 * kk)) ||
        (rc = dev_reserve(h, &h->d_exF, &h->exF_cap, (size_t)per_pass * C * E * kk)) ||
        (rc = dev_reserve(h, &h->d_exmask, &h->exmask_cap, (size_t)E)) ||
        (!use_lds && (rc = dev_reserve(h, &h->d_exscr, &h->exscr_cap, (size_t)C * E * EXPM_BUFFERS * n2)))) return rc;
    double *d_L = h->d_exL, *d_F = h->d_exF;
    int *d_mask = nullptr;
    dd *d_scr = h->d_exscr;
    if (edge_mask) {
        HIPCHK(h, hipMemcpyAsync(h->d_exmask, edge_mask, (size_t)E * sizeof(int), hipMemcpyHostToDevice, h->stream));
        d_mask = h->d_exmask;
    }
    std::vector<double> L((size_t)per_pass * 2 * kk), tmp_site, tmp_sums;
    for (int m0 = 0; m0 < nL; m0 += per_pass) {
        const int nm = std::min(per_pass, nL - m0);
        for (int m = 0; m < nm; m++) {
            double *Lm = L.data() + (size_t)m * 2 * kk;
            std::copy(L_hi + (size_t)(m0 + m) * kk, L_hi + (size_t)(m0 + m + 1) * kk, Lm);
            if (L_lo) std::copy(L_lo + (size_t)(m0 + m) * kk, L_lo + (size_t)(m0 + m + 1) * kk, Lm + kk);
            else std::fill(Lm + kk, Lm + 2 * kk, 0.0);
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));          /* the previous pass may still read d_L */
        HIPCHK(h, hipMemcpyAsync(d_L, L.data(), (size_t)nm * 2 * kk * sizeof(double), hipMemcpyHostToDevice, h->stream));
        for (int m = 0; m < nm; m++) {
            hipLaunchKernelGGL(k_expm_dd<true>, dim3(C * E), dim3(threads), lds_bytes, h->stream,
                               k, E, h->d_Qn, h->d_edge_rates, h->d_cat_rates, (dd *)nullptr, (double *)nullptr, (double *)nullptr,
                               d_scr, use_lds, d_L + (size_t)m * 2 * kk, coef_mode, d_mask, d_F + (size_t)m * C * E * kk, ExpmPost{});
        }
        if (hipGetLastError() != hipSuccess) { h->err = "plk_edge_expect: Frechet kernel launch failed"; return PLK_E_DEVICE; }
        HIPCHK(h, hipStreamSynchronize(h->stream));          /* L is a host local */
        /* outputs of this pass: [S][nm*E] and [nm*E][2]; scattered into [S][nL*E] / [nL*E][2] */
        double *so = nullptr, *su = nullptr;
        if (site_out) { if (nm == nL) so = site_out; else { tmp_site.assign((size_t)S * nm * E, 0.0); so = tmp_site.data(); } }
        if (sums_out) su = sums_out + (size_t)m0 * E * 2;
        rc = run_updown(h, true, false, edge_mask, nullptr, so, su, d_F, 0, nm);
        if (rc) { cleanup(); return rc; }
        if (site_out && nm != nL)
            for (long s = 0; s < S; s++)
                for (int r = 0; r < nm * E; r++)
                    site_out[((size_t)s * nL + m0) * E + r] = tmp_site[(size_t)s * nm * E + r];
    }
    cleanup();
    return PLK_OK;
}

/* Conditional edge expectations (dwell / trans / em-update numerators): replaces
 * src/evaluate_site_frechet.c:5-42 + the Frechet matrix set-up of src/arbplfdwell.c:117-204,
 * src/arbplftrans.c:116-224, src/arbplfem.c:100-158. */
extern "C" int plk_edge_expect(plk_engine *h, const double *L_hi, const double *L_lo, int coef_mode,
                               const int *edge_mask, double *site_edge_out, double *edge_sums_out)
{
    return plk_edge_expect_multi(h, 1, L_hi, L_lo, coef_mode, edge_mask, site_edge_out, edge_sums_out);
}

extern "C" int plk_get_frechet_matrices(plk_engine *h, const double *L_hi, const double *L_lo, int coef_mode, double *F_out)
{
    if (!plk_live(h) || !L_hi || !F_out) return PLK_E_ARG;
    if (h->k == 0) { h->err = "plk_get_frechet_matrices: model must be set"; return PLK_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    if (h->model_dirty) { if ((rc = run_expm(h))) return rc; }
    const int k = h->k, C = h->C, E = h->E;
    if (E == 0) return PLK_OK;
    const size_t kk = (size_t)k * k, n2 = 4 * kk;
    std::vector<double> L(2 * kk, 0.0);
    std::copy(L_hi, L_hi + kk, L.begin());
    if (L_lo) std::copy(L_lo, L_lo + kk, L.begin() + kk);
    double *d_L = nullptr, *d_F = nullptr;
    dd *d_scr = nullptr;
    auto cleanup = [&]() {
        if (d_L) (void)hipFree(d_L);
        if (d_F) (void)hipFree(d_F);
        if (d_scr) (void)hipFree(d_scr);
    };
    if ((rc = dev_upload(h, &d_L, L.data(), L.size())) || (rc = dev_alloc(h, &d_F, (size_t)C * E * kk))) { cleanup(); return rc; }
    const int threads = n2 >= 1024 ? 1024 : (n2 >= 256 ? 256 : 64);
    int use_lds;
    const size_t lds_bytes = expm_lds_bytes((size_t)2 * k, threads, &use_lds);
    if (!use_lds && (rc = dev_alloc(h, &d_scr, (size_t)C * E * EXPM_BUFFERS * n2))) { cleanup(); return rc; }
    hipLaunchKernelGGL(k_expm_dd<true>, dim3(C * E), dim3(threads), lds_bytes, h->stream,
                       k, E, h->d_Qn, h->d_edge_rates, h->d_cat_rates, (dd *)nullptr, (double *)nullptr, (double *)nullptr,
                       d_scr, use_lds, d_L, coef_mode, (const int *)nullptr, d_F, ExpmPost{});
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(F_out, d_F, (size_t)C * E * kk * sizeof(double), hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) { h->err = std::string("plk_get_frechet_matrices: ") + hipGetErrorString(e); return PLK_E_DEVICE; }
    return PLK_OK;
}

extern "C" int plk_marginal(plk_engine *h, const int *node_mask, double *site_out, double *sums_out)
{
    if (!plk_live(h)) return PLK_E_ARG;
    return run_updown(h, false, true, nullptr, node_mask, site_out, sums_out);
}

/* ====================================================================== */
/* Edge-rate optimisation with the patterns resident      */
/* ====================================================================== */

static int fit_objective(plk_engine *h, const std::vector<double> &rates, long double *ll_out)
{
    int rc;
    if ((rc = plk_update_edge_rates(h, rates.data()))) return rc;
    double s[2] = {0.0, 0.0};
    if ((rc = plk_ll(h, nullptr, PLK_HOST, s))) return rc;
    *ll_out = (long double)s[0] + (long double)s[1];
    return PLK_OK;
}

/*
 * Maximise sum_s w_s ll_s over the edge rate coefficients.  The reference has no driver for this; its
 * old-examples/opt.py and old-examples/gell_dna_opt.py run scipy's L-BFGS-B over arbplf_ll / arbplf_deriv
 * through the Python API, and test_scripts/test_em_monotonicity.py iterates arbplf_em_update.  Here the loop
 * runs next to the device: patterns, weights and the tree stay resident, an iteration re-runs K1 and the
 * traversal kernels only.
 *   PLK_FIT_EM     r_e <- r_e * E[transitions on e] / E[rate-weighted dwell on e]   (src/arbplfem.c:397-503)
 *   PLK_FIT_LBFGS  L-BFGS (8 pairs, Armijo backtracking) on log r_e with the gradient r_e * d ll / d r_e
 */
extern "C" int plk_fit_edge_rates(plk_engine *h, int method, int max_iter, double ftol, const int *edge_mask,
                                  double *rates_inout, double *ll_trace, int *iters_out, long *evals_out)
{
    if (!plk_live(h) || !rates_inout) return PLK_E_ARG;
    if (h->k == 0 || h->pat_mode == 0) { h->err = "plk_fit_edge_rates: tree, model and patterns must be set"; return PLK_E_ARG; }
    if (method != PLK_FIT_EM && method != PLK_FIT_LBFGS) { h->err = "plk_fit_edge_rates: unknown method"; return PLK_E_ARG; }
    const int E = h->E, k = h->k;
    const size_t kk = (size_t)k * k;
    int rc;
    long evals = 0;
    std::vector<double> rates(rates_inout, rates_inout + E);
    for (int e = 0; e < E; e++)
        if (!(rates[e] >= 0.0) || !std::isfinite(rates[e])) { h->err = "plk_fit_edge_rates: rates must be finite and non-negative"; return PLK_E_ARG; }
    std::vector<int> free_e;
    for (int e = 0; e < E; e++) if ((!edge_mask || edge_mask[e]) && rates[e] > 0.0) free_e.push_back(e);
    std::vector<int> mask(std::max(E, 1), 0);
    for (int e : free_e) mask[e] = 1;
    long double ll = 0;
    if ((rc = fit_objective(h, rates, &ll))) return rc;
    evals++;
    if (ll_trace) ll_trace[0] = (double)ll;
    int it = 0;
    if (!std::isfinite((double)ll)) { h->err = "plk_fit_edge_rates: the initial log likelihood is not finite"; return PLK_E_ARG; }

    if (method == PLK_FIT_EM) {
        std::vector<double> Ld(2 * kk, 0.0), Lt(2 * kk, 0.0), dw((size_t)2 * E + 2), tr((size_t)2 * E + 2);
        for (int i = 0; i < k; i++)
            for (int j = 0; j < k; j++) {
                const size_t q = (size_t)i * k + j;
                if (i == j) { Ld[q] = -h->Qn[q]; Ld[kk + q] = -h->Qn[kk + q]; }
                else { Lt[q] = h->Qn[q]; Lt[kk + q] = h->Qn[kk + q]; }
            }
        for (; it < max_iter && !free_e.empty(); ) {
            /* both expectations in one pass where the kernels allow it */
            std::vector<double> Lh(2 * kk), Ll(2 * kk), both((size_t)4 * E + 4);
            std::copy(Ld.begin(), Ld.begin() + kk, Lh.begin()); std::copy(Lt.begin(), Lt.begin() + kk, Lh.begin() + kk);
            std::copy(Ld.begin() + kk, Ld.end(), Ll.begin()); std::copy(Lt.begin() + kk, Lt.end(), Ll.begin() + kk);
            if ((rc = plk_edge_expect_multi(h, 2, Lh.data(), Ll.data(), PLK_COEF_PRIOR_RATE, mask.data(), nullptr, both.data()))) return rc;
            std::copy(both.begin(), both.begin() + 2 * E, dw.begin());
            std::copy(both.begin() + 2 * E, both.begin() + 4 * E, tr.begin());
            for (int e : free_e) {
                const long double t = (long double)tr[2 * e] + (long double)tr[2 * e + 1];
                const long double d = (long double)dw[2 * e] + (long double)dw[2 * e + 1];
                const double v = t == 0 ? 0.0 : (double)(t / d * (long double)rates[e]);
                if (!std::isfinite(v) || v < 0) { h->err = "plk_fit_edge_rates: EM update is not finite"; return PLK_E_DEVICE; }
                rates[e] = v;
            }
            long double ll_new = 0;
            if ((rc = fit_objective(h, rates, &ll_new))) return rc;
            evals++;
            it++;
            if (ll_trace) ll_trace[it] = (double)ll_new;
            const long double gain = ll_new - ll;
            ll = ll_new;
            if (fabsl(gain) <= (long double)ftol * std::max<long double>(1.0L, fabsl(ll))) break;
        }
    } else {
        const int n = (int)free_e.size(), M = 8;
        std::vector<double> x(n), g(n), xn(n), gn(n), d(n), sums((size_t)2 * E + 2);
        std::vector<std::vector<double>> Sh, Yh;
        std::vector<double> rho;
        auto gradient = [&](const std::vector<double> &r, std::vector<double> &out) -> int {
            /* rates are already current on the device (set by the accepted objective evaluation) */
            int rc2 = plk_deriv(h, mask.data(), nullptr, sums.data());
            if (rc2) return rc2;
            for (int i = 0; i < n; i++) {
                const int e = free_e[i];
                out[i] = -(double)(((long double)sums[2 * e] + (long double)sums[2 * e + 1]) * (long double)r[e]);
            }
            return PLK_OK;
        };
        for (int i = 0; i < n; i++) x[i] = std::log(rates[free_e[i]]);
        if (n > 0 && (rc = gradient(rates, g))) return rc;
        double f = -(double)ll;
        for (; it < max_iter && n > 0; ) {
            /* two-loop recursion */
            d = g;
            const int hs = (int)Sh.size();
            std::vector<double> alpha(hs);
            for (int j = hs - 1; j >= 0; j--) {
                double a = 0; for (int i = 0; i < n; i++) a += Sh[j][i] * d[i];
                a *= rho[j]; alpha[j] = a;
                for (int i = 0; i < n; i++) d[i] -= a * Yh[j][i];
            }
            if (hs > 0) {
                double sy = 0, yy = 0;
                for (int i = 0; i < n; i++) { sy += Sh[hs - 1][i] * Yh[hs - 1][i]; yy += Yh[hs - 1][i] * Yh[hs - 1][i]; }
                const double gam = sy / yy;
                for (int i = 0; i < n; i++) d[i] *= gam;
            }
            for (int j = 0; j < hs; j++) {
                double b = 0; for (int i = 0; i < n; i++) b += Yh[j][i] * d[i];
                b *= rho[j];
                for (int i = 0; i < n; i++) d[i] += Sh[j][i] * (alpha[j] - b);
            }
            double gd = 0, gnorm = 0, dmax = 0;
            for (int i = 0; i < n; i++) { d[i] = -d[i]; gd += g[i] * d[i]; gnorm = std::max(gnorm, std::fabs(g[i])); }
            if (!(gd < 0)) { Sh.clear(); Yh.clear(); rho.clear(); gd = 0; for (int i = 0; i < n; i++) { d[i] = -g[i]; gd += g[i] * d[i]; } }
            if (gnorm == 0) break;
            for (int i = 0; i < n; i++) dmax = std::max(dmax, std::fabs(d[i]));
            double t = Sh.empty() ? std::min(1.0, 1.0 / dmax) : 1.0;
            if (t * dmax > 5.0) t = 5.0 / dmax;                 /* at most a factor e^5 per step on any rate */
            bool ok = false;
            long double ll_new = 0;
            std::vector<double> rn = rates;
            for (int ls = 0; ls < 40; ls++, t *= 0.5) {
                for (int i = 0; i < n; i++) { xn[i] = x[i] + t * d[i]; rn[free_e[i]] = std::exp(xn[i]); }
                if ((rc = fit_objective(h, rn, &ll_new))) return rc;
                evals++;
                const double fn = -(double)ll_new;
                if (std::isfinite(fn) && fn <= f + 1e-4 * t * gd) { ok = true; break; }
            }
            if (!ok) { if ((rc = plk_update_edge_rates(h, rates.data()))) return rc; break; }
            if ((rc = gradient(rn, gn))) return rc;
            std::vector<double> s(n), y(n);
            double sy = 0, ss = 0, yy = 0;
            for (int i = 0; i < n; i++) { s[i] = xn[i] - x[i]; y[i] = gn[i] - g[i]; sy += s[i] * y[i]; ss += s[i] * s[i]; yy += y[i] * y[i]; }
            if (sy > 1e-10 * std::sqrt(ss * yy)) {
                if ((int)Sh.size() == M) { Sh.erase(Sh.begin()); Yh.erase(Yh.begin()); rho.erase(rho.begin()); }
                Sh.push_back(s); Yh.push_back(y); rho.push_back(1.0 / sy);
            }
            const double fn = -(double)ll_new, gain = f - fn;
            x = xn; g = gn; rates = rn; f = fn; ll = ll_new;
            it++;
            if (ll_trace) ll_trace[it] = (double)ll;
            if (gain <= ftol * std::max(1.0, std::fabs(f))) break;
        }
    }
    if ((rc = plk_update_edge_rates(h, rates.data()))) return rc;
    std::copy(rates.begin(), rates.end(), rates_inout);
    if (iters_out) *iters_out = it;
    if (evals_out) *evals_out = evals;
    return PLK_OK;
}

/* ====================================================================== */
/* Second derivatives of the log likelihood               */
/* ====================================================================== */

/* d2P[c][e] = r_c^2 * (Qn Qn) * P[c][e], accumulated in double-double from the unrounded P */
__global__ void k_d2p(int k, int E, const double *__restrict__ Q2 /* [2][k*k] hi, lo */, const dd *__restrict__ Pdd,
                      const double *__restrict__ cat_rates, double *__restrict__ out)
{
    const int ce = blockIdx.x, c = ce / E;
    const int kk = k * k;
    const dd *P = Pdd + (size_t)ce * kk;
    const double r = cat_rates[c];
    for (int idx = threadIdx.x; idx < kk; idx += blockDim.x) {
        const int i = idx / k, j = idx - i * k;
        dd acc = dd_make(0.0, 0.0);
        for (int l = 0; l < k; l++) acc = dd_add(acc, dd_mul(dd_make(Q2[i * k + l], Q2[kk + i * k + l]), P[l * k + j]));
        acc = dd_mul_d(dd_mul_d(acc, r), r);
        out[(size_t)ce * kk + idx] = acc.hi;
    }
}

/* G[i][j] = sum_s w_s D[i][s] D[j][s] for i >= j, one workgroup per pair, double-double accumulation */
__global__ __launch_bounds__(256) void k_gram(int E, long n, const double *__restrict__ D, const double *__restrict__ w,
                                              dd *__restrict__ out)
{
    const int i = blockIdx.x, j = blockIdx.y;
    if (j > i) return;
    const double *di = D + (size_t)i * n, *dj = D + (size_t)j * n;
    dd acc = dd_make(0.0, 0.0);
    for (long s = threadIdx.x; s < n; s += 256) {
        dd p = dd_two_prod(di[s], dj[s]);
        if (w) p = dd_mul_d(p, w[s]);
        acc = dd_add(acc, p);
    }
    acc = dd_block_sum(acc);
    if (threadIdx.x == 0) out[(size_t)i * E + j] = acc;
}

template <int K>
static void launch_hess_pass(plk_engine *h, const UpArgs &a, unsigned grid)
{
    hipLaunchKernelGGL(k_down_store<K>, dim3(grid), dim3(GEN_BLOCK), 0, h->stream, a);
    hipLaunchKernelGGL((k_up<K, true, false>), dim3(grid), dim3(GEN_BLOCK), 0, h->stream, a);
}

static void launch_hess_pass_k(plk_engine *h, const UpArgs &a, unsigned grid)
{
    switch (h->K) {
    case 2: launch_hess_pass<2>(h, a, grid); break;
    case 4: launch_hess_pass<4>(h, a, grid); break;
    case 8: launch_hess_pass<8>(h, a, grid); break;
    case 16: launch_hess_pass<16>(h, a, grid); break;
    case 20: launch_hess_pass<20>(h, a, grid); break;
    case 32: launch_hess_pass<32>(h, a, grid); break;
    case 61: launch_hess_pass<61>(h, a, grid); break;
    default: launch_hess_pass<64>(h, a, grid); break;
    }
}

/*
 * Hessian of sum_s w_s ll_s with respect to the edge rate coefficients (CSR edge order), replacing
 * _recompute_second_order of src/arbplfhess.c:503-760 for the fp64, uncertified case.
 *
 *   d^2 ll / dr_i dr_j = H_ij / f - (g_i / f)(g_j / f)            (src/arbplfhess.c:455-493)
 * with f the site likelihood, g its gradient and H its Hessian.  g_i / f is the ordinary derivative pass.
 * Row j of H is obtained from the same two kernels run on a modified model in which edge j carries
 * dP_j = r Q P_j in the role of P_j and r^2 Q Q P_j in the role of dP_j: the derivative of that model with
 * respect to edge i is the second derivative (i, j) of the original one -- the reference's substitution of Q
 * along both root paths (src/arbplfhess.c:343-437) done for all i at once in O(E k^2) per site and row.
 * E + 1 passes per site chunk on the generic vector kernels, with the same exact power-of-two rescaling as the
 * first-order passes (each pass stores its factors; the common exponents of the modified and the unmodified model
 * are reconciled in the division).
 */
extern "C" int plk_hess(plk_engine *h, double *hess_sums_out /* [E][E][2] */)
{
    if (!plk_live(h) || !hess_sums_out) return PLK_E_ARG;
    if (h->k == 0 || h->pat_mode == 0) { h->err = "plk_hess: tree, model and patterns must be set"; return PLK_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    if (h->model_dirty) { if ((rc = run_expm(h))) return rc; }
    if ((rc = ensure_dP(h))) return rc;
    const int N = h->N, E = h->E, k = h->k, K = h->K, C = h->C;
    const long S = h->S;
    const size_t kk = (size_t)k * k, strm = (size_t)C * E * K * K;
    if (E == 0) return PLK_OK;

    /* Qn^2 in double-double on the host */
    std::vector<double> Q2(2 * kk);
    for (int i = 0; i < k; i++)
        for (int j = 0; j < k; j++) {
            dd acc = dd_make(0.0, 0.0);
            for (int l = 0; l < k; l++)
                acc = dd_add(acc, dd_mul(dd_make(h->Qn[(size_t)i * k + l], h->Qn[kk + (size_t)i * k + l]),
                                         dd_make(h->Qn[(size_t)l * k + j], h->Qn[kk + (size_t)l * k + j])));
            Q2[(size_t)i * k + j] = acc.hi; Q2[kk + (size_t)i * k + j] = acc.lo;
        }
    double *d_Q2 = nullptr, *d_d2P = nullptr, *d_PT = nullptr, *d_PN = nullptr, *d_DT = nullptr, *d_DN = nullptr, *d_D2T = nullptr;
    double *d_LH0 = nullptr, *d_D0 = nullptr, *d_XM0 = nullptr;
    dd *d_G = nullptr;
    int *d_has = nullptr, *d_ns = nullptr;
    auto cleanup = [&]() {
        void *ps[] = {d_Q2, d_d2P, d_PT, d_PN, d_DT, d_DN, d_D2T, d_LH0, d_D0, d_XM0, d_G, d_has, d_ns};
        for (void *p : ps) if (p) (void)hipFree(p);
    };
    if ((rc = dev_upload(h, &d_Q2, Q2.data(), Q2.size())) || (rc = dev_alloc(h, &d_d2P, (size_t)C * E * kk)) ||
        (rc = dev_alloc(h, &d_PT, strm)) || (rc = dev_alloc(h, &d_PN, strm)) || (rc = dev_alloc(h, &d_DT, strm)) ||
        (rc = dev_alloc(h, &d_DN, strm)) || (rc = dev_alloc(h, &d_D2T, strm)) || (rc = dev_alloc(h, &d_G, (size_t)E * E))) { cleanup(); return rc; }
    hipLaunchKernelGGL(k_d2p, dim3(C * E), dim3(kk >= 256 ? 256 : 64), 0, h->stream, k, E, d_Q2, h->d_Pdd, h->d_cat_rates, d_d2P);
    const int bt = K * K >= 256 ? 256 : 64;
    hipLaunchKernelGGL(k_build_edge_stream, dim3(C * E), dim3(bt), 0, h->stream, k, K, 0, h->d_P, d_PT);
    hipLaunchKernelGGL(k_build_edge_stream, dim3(C * E), dim3(bt), 0, h->stream, k, K, 1, h->d_P, d_PN);
    hipLaunchKernelGGL(k_build_edge_stream, dim3(C * E), dim3(bt), 0, h->stream, k, K, 0, h->d_dP, d_DT);
    hipLaunchKernelGGL(k_build_edge_stream, dim3(C * E), dim3(bt), 0, h->stream, k, K, 1, h->d_dP, d_DN);
    hipLaunchKernelGGL(k_build_edge_stream, dim3(C * E), dim3(bt), 0, h->stream, k, K, 0, d_d2P, d_D2T);
    if (hipGetLastError() != hipSuccess) { cleanup(); h->err = "plk_hess: matrix set-up failed"; return PLK_E_DEVICE; }
    if (h->node_has_data.size() != (size_t)N) h->node_has_data.assign(N, 1);
    { std::vector<int> hd(h->node_has_data.begin(), h->node_has_data.end()); if ((rc = dev_upload(h, &d_has, hd.data(), (size_t)N))) { cleanup(); return rc; } }

    /* exact power-of-two rescaling as in the first-order passes (trees of any size): the traversal program marks the
     * nodes, the factors are stored per pass, every pass reports the common exponent of its site likelihoods */
    if (h->prog_dirty) { if ((rc = build_program(h))) { cleanup(); return rc; } }
    std::vector<int> node_scale(N, -1);
    int nsc = 0;
    for (int a = 0; a < N; a++) if (h->indptr[a + 1] > h->indptr[a] && h->scale_node[a]) node_scale[a] = nsc++;
    if ((rc = dev_upload(h, &d_ns, node_scale.data(), (size_t)N))) { cleanup(); return rc; }
    const size_t per_site_d = (size_t)(E + 2 * (size_t)N) * C * k + (size_t)(nsc + 2) * C + 2 + (size_t)E;
    const size_t per_site = (per_site_d + 2 + (size_t)E) * sizeof(double);
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    size_t budget = free_b > (size_t)(6ull << 30) ? free_b - (size_t)(4ull << 30) : free_b / 2;
    budget += h->work_cap * sizeof(double);
    long chunk = (long)std::min<size_t>((size_t)S, budget / per_site);
    if (h->opt_site_chunk > 0) chunk = std::min<long>(chunk, h->opt_site_chunk);
    if (chunk < 1) { cleanup(); h->err = "plk_hess: not enough device memory for one site"; return PLK_E_NOMEM; }
    if (chunk < S) chunk = std::max<long>(GEN_BLOCK, chunk / GEN_BLOCK * GEN_BLOCK);
    if ((rc = dev_reserve(h, &h->d_work, &h->work_cap, per_site_d * (size_t)chunk)) ||
        (rc = dev_alloc(h, &d_LH0, (size_t)chunk)) || (rc = dev_alloc(h, &d_XM0, (size_t)chunk)) ||
        (rc = dev_alloc(h, &d_D0, (size_t)E * chunk))) { cleanup(); return rc; }

    std::vector<long double> Hrow((size_t)E * E, 0.0L), G((size_t)E * E, 0.0L);
    std::vector<dd> gh((size_t)E * E);
    for (long s0 = 0; s0 < S; s0 += chunk) {
        const long n = std::min(chunk, S - s0);
        UpArgs a;
        a.S = S; a.Spad = h->Spad; a.s0 = s0; a.n = n;
        a.N = N; a.E = E; a.k = k; a.C = C; a.nchar = h->nchar; a.pat_mode = h->pat_mode; a.root_mode = h->root_mode;
        a.dzero = 1;
        a.indptr = h->d_indptr; a.indices = h->d_indices; a.preorder = h->d_preorder; a.node_has_data = d_has;
        a.PT = d_PT; a.PN = d_PN; a.DT = d_DT; a.DN = d_DN; a.D2T = d_D2T;
        a.node_scale = d_ns;
        a.codes = h->d_codes; a.defs = h->d_defs; a.B = h->d_B;
        a.cat_prior = h->d_cat_prior; a.root_w = h->d_root_w; a.edge_mask = nullptr; a.node_mask = nullptr;
        double *p = h->d_work;
        a.EV = p; p += (size_t)E * C * k * n;
        a.LN = p; p += (size_t)N * C * k * n;
        a.FN = p; p += (size_t)N * C * k * n;
        a.SC = p; p += (size_t)nsc * C * n;
        a.CW = p; p += (size_t)C * n;
        a.XC = p; p += (size_t)C * n;
        a.XM = p; p += n;
        a.LH = p; p += n;
        a.DV = p; p += (size_t)E * n;
        a.MV = nullptr;
        const unsigned grid = (unsigned)((n + GEN_BLOCK - 1) / GEN_BLOCK);
        const double *w = h->d_w ? h->d_w + s0 : nullptr;
        /* pass 0: the model itself -> f and g / f */
        a.mod_edge = -1; a.LHdiv = nullptr; a.XMdiv = nullptr;
        launch_hess_pass_k(h, a, grid);
        hipError_t e = hipMemcpyAsync(d_LH0, a.LH, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_XM0, a.XM, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_D0, a.DV, (size_t)E * n * sizeof(double), hipMemcpyDeviceToDevice, h->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_gram, dim3(E, E), dim3(256), 0, h->stream, E, n, d_D0, w, d_G);
            e = hipMemcpyAsync(gh.data(), d_G, (size_t)E * E * sizeof(dd), hipMemcpyDeviceToHost, h->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { cleanup(); h->err = std::string("plk_hess: ") + hipGetErrorString(e); return PLK_E_DEVICE; }
        for (int i = 0; i < E; i++)
            for (int j = 0; j <= i; j++) G[(size_t)i * E + j] += (long double)gh[(size_t)i * E + j].hi + (long double)gh[(size_t)i * E + j].lo;
        /* passes 1..E: row j of the likelihood Hessian, normalised by the unmodified likelihood */
        a.LHdiv = d_LH0; a.XMdiv = d_XM0;
        for (int j = 0; j < E; j++) {
            a.mod_edge = j;
            launch_hess_pass_k(h, a, grid);
            if (hipGetLastError() != hipSuccess) { cleanup(); h->err = "plk_hess: kernel launch failed"; return PLK_E_DEVICE; }
            if ((rc = wsum_rows(h, E, n, a.DV, w, Hrow.data() + (size_t)j * E))) { cleanup(); return rc; }
        }
    }
    cleanup();
    for (int i = 0; i < E; i++)
        for (int j = 0; j <= i; j++) {
            /* both triangles of the likelihood Hessian were computed; average them */
            const long double v = (Hrow[(size_t)i * E + j] + Hrow[(size_t)j * E + i]) * 0.5L - G[(size_t)i * E + j];
            const double hi = (double)v, lo = (double)(v - (long double)hi);
            hess_sums_out[2 * ((size_t)i * E + j)] = hess_sums_out[2 * ((size_t)j * E + i)] = hi;
            hess_sums_out[2 * ((size_t)i * E + j) + 1] = hess_sums_out[2 * ((size_t)j * E + i) + 1] = lo;
        }
    return PLK_OK;
}
