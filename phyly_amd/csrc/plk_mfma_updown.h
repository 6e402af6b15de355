/*
 * plk_mfma_updown.h -- down pass with stored vectors and BFS-order up pass (edge
 * derivatives, marginals) for 9 <= k <= 64 on the fp64 matrix cores.  Included by
 * plk_engine.hip after plk_mfma.h; same distributed layout (a site's vector spread over
 * the 4 lanes {s, s+16, s+32, s+48}, lane group g holding states g, g+4, ...).
 *
 * Replaces for large state spaces (same formulas as k_down_store / k_up):
 *   src/evaluate_site_lhood.c:7-63 (with edge vectors), src/evaluate_site_forward.c:32-105,
 *   src/arbplfderiv.c:112-207,:312-342, src/evaluate_site_marginal.c:7-21, src/arbplfmarginal.c:206-234.
 *
 * Leaf edges never touch the matrix cores: their edge vectors P_e B_b and derivative
 * vectors dP_e B_b are gathered from per-edge tables indexed by the pattern code
 * (built in double-double).  Internal edges stage the A fragments of P_e, P_e^T or dP_e
 * through LDS once per workgroup.
 */
#ifndef PLK_MFMA_UPDOWN_H
#define PLK_MFMA_UPDOWN_H

struct MUpArgs {
    long S, Spad, s0, n;          /* chunk [s0, s0+n) of the pattern extent */
    int N, E, k, kk4, C, nchar, ntips, root_mode;
    int dzero;                    /* 1: edge-form matrices have zero row sums (dP) */
    const int *indptr, *indices, *preorder;
    const int *node_has_data;
    const int *edge_tip;          /* E: tip slot of a leaf edge, -1 for internal edges */
    const int *edge_int;          /* E: index among internal edges, -1 for leaf edges */
    const int *node_int;          /* N: index among internal nodes, -1 for leaves */
    const double *fragP, *fragPT, *fragD;   /* [C][E][T][kk4][64] */
    const double *tip;            /* [C][ntips+1][nchar][4][R]: P_e defs, last slot raw defs */
    const double *dtip;           /* [C][ntips+1][nchar][4][R]: dP_e defs */
    const uint8_t *codes;
    const double *cat_prior, *root_wd;
    const int *edge_mask, *node_mask;
    double *EV, *LN, *FN;         /* [(ent*C + c)*R + r][stride] planes, lane-linear */
    long stride;
    const int *node_scale;        /* N: rescaling slot of the node, -1 = not rescaled */
    double *SC;                   /* [(slot*C + c)][n]: 2^-e applied to L_a at rescaled nodes */
    double *CW, *XC;              /* [C][n]: 2^(X_c - Xmax); scratch */
    double *LH;                   /* [n], at the common exponent Xmax */
    double *DV;                   /* [E][n] */
    double *MV;                   /* [N][k][n] */
    double *MVS;                  /* site-summed marginals only: [(node * k + state)][nwaves] weighted sums of a wave's 16 sites */
    const double *wsite;          /* [n] site weights of the chunk or null */
    /* node-visit up pass (k_up_nodes_mfma): records of plk_up_nodes_build(), and fragD holds the fragments of M^T */
    const int *visits;
    int nvisits;
    /* k_up_mfma, derivative queries without marginals: node_inline[b] != 0 for a non-root node whose one or two children are
     * all leaves -- finished inside its parent's visit while F_b is in registers (F_b and L_b never go through HBM) */
    const int *node_inline;
};

template <int R>
__device__ static inline void mf_gather(const double *tab, int nchar, int slot, int ch, int g, double (&out)[R])
{
    const double2 *tp = reinterpret_cast<const double2 *>(tab + (((size_t)slot * nchar + ch) * 4 + g) * R);
#pragma unroll
    for (int r = 0; r < R; r += 2) { const double2 v = tp[r >> 1]; out[r] = v.x; out[r + 1] = v.y; }
}

template <int R>
__device__ static inline void mf_load(const double *base, long stride, long lin, double (&out)[R])
{
#pragma unroll
    for (int r = 0; r < R; r++) out[r] = base[(size_t)r * stride + lin];
}

template <int R>
__device__ static inline void mf_store(double *base, long stride, long lin, const double (&v)[R])
{
#pragma unroll
    for (int r = 0; r < R; r++) base[(size_t)r * stride + lin] = v[r];
}

/* the same with the plane base pinned in an SGPR pair and the lane's index as a 32-bit offset: one VGPR addresses all
 * R planes, where the forms above let the compiler keep R 64-bit lane addresses per array live across the loops
 * (k_up_nodes_mfma<1>, vectors of 8 registers, needed 104 VGPRs with them) */
template <int R>
__device__ __forceinline__ void mfn_load(const double *base, long stride, long lin, double (&out)[R])
{
    const unsigned off = (unsigned)lin;
    asm volatile("" : "+s"(base));
#pragma unroll
    for (int r = 0; r < R; r++) out[r] = (base + (size_t)r * stride)[off];
}

template <int R>
__device__ __forceinline__ void mfn_store(double *base, long stride, long lin, const double (&v)[R])
{
    const unsigned off = (unsigned)lin;
    asm volatile("" : "+s"(base));
#pragma unroll
    for (int r = 0; r < R; r++) (base + (size_t)r * stride)[off] = v[r];
}

/* table row of one pattern code: uniform base of the slot's table in SGPRs, the lane's (code, lane group) as a 32-bit offset */
template <int R>
__device__ __forceinline__ void mfn_gather(const double *tab, int nchar, int slot, unsigned ch, unsigned g, double (&out)[R])
{
    const double *base = tab + (size_t)slot * nchar * 4 * R;
    asm volatile("" : "+s"(base));
    const double2 *tp = reinterpret_cast<const double2 *>(base);
    const unsigned off = ((ch * 4u + g) * (unsigned)R) >> 1;
#pragma unroll
    for (int r = 0; r < R; r += 2) { const double2 v = tp[off + (r >> 1)]; out[r] = v.x; out[r + 1] = v.y; }
}
/* pattern code of the lane's site at one node: row base in SGPRs, site as a 32-bit offset */
__device__ __forceinline__ unsigned mfn_code(const uint8_t *codes, long Spad, long s0, int node, unsigned site)
{
    const uint8_t *row = codes + (size_t)node * Spad + s0;
    asm volatile("" : "+s"(row));
    return row[site];
}
template <class V>
__device__ __forceinline__ V mfn_at(const V *base, unsigned off)
{
    asm volatile("" : "+s"(base));
    return base[off];
}

/* all k states of the site equal? (the reference's exact constant-column test, src/arb_mat_extras.c:36-51) */
template <int R>
__device__ static inline bool mf_is_const(const double (&x)[R], int g, int k, double &x0)
{
    double lo = INFINITY, hi = -INFINITY;
#pragma unroll
    for (int r = 0; r < R; r++)
        if (g + 4 * r < k) { lo = fmin(lo, x[r]); hi = fmax(hi, x[r]); }
    lo = fmin(lo, __shfl_xor(lo, 16, 64)); lo = fmin(lo, __shfl_xor(lo, 32, 64));
    hi = fmax(hi, __shfl_xor(hi, 16, 64)); hi = fmax(hi, __shfl_xor(hi, 32, 64));
    x0 = lo;
    return lo == hi;
}

/* mf_is_const with the lane-group index laundered through an empty asm: the 4T validity masks (g + 4r < k) are then
 * recomputed at each call instead of being hoisted out of the visit loop, where they cost 2 SGPRs each and, once those
 * ran out, spilled VGPR lanes and scratch (k_up_nodes_mfma<4>: 86 loop-invariant registers before this) */
template <int R>
__device__ __forceinline__ bool mfn_is_const(const double (&x)[R], int g, int k, double &x0)
{
    asm volatile("" : "+v"(g));
    return mf_is_const<R>(x, g, k, x0);
}
/* y = M x with M given as A fragments already staged in LDS */
template <int T>
__device__ static inline void mf_matvec(const double *lds_frag, int kk4, int lane, const double (&x)[4 * T], double (&y)[4 * T])
{
    plk_d4 acc[T];
#pragma unroll
    for (int t = 0; t < T; t++) acc[t] = (plk_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 4 * T; q++) {
        if (q < kk4) {
#pragma unroll
            for (int t = 0; t < T; t++)
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(lds_frag[(t * kk4 + q) * 64 + lane], x[q], acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < T; t++) {
        y[4 * t + 0] = acc[t][0]; y[4 * t + 1] = acc[t][1]; y[4 * t + 2] = acc[t][2]; y[4 * t + 3] = acc[t][3];
    }
}

template <int T>
__global__ __launch_bounds__(MF_BLOCK) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_down_store_mfma(MUpArgs a)
{
    extern __shared__ double lds_frag[];
    constexpr int R = 4 * T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
    const long sl = (long)blockIdx.x * MF_SITES + wave * 16 + (lane & 15);
    const bool valid = sl < a.n;
    const long sg = a.s0 + (valid ? sl : a.n - 1);
    const long lin = ((long)blockIdx.x * MF_SITES + wave * 16) * 4 + lane;
    const int nfrag = T * a.kk4 * 64;
    const int NT = a.ntips + 1;
    const size_t n = (size_t)a.n;
    const long slc = valid ? sl : a.n - 1;
    int xmax = INT_MIN;
    for (int c = 0; c < a.C; c++) {
        const double *tipc = a.tip + (size_t)c * NT * a.nchar * 4 * R;
        double lh_c = 0.0;
        int X = 0;
        for (int u = a.N - 1; u >= 0; u--) {
            const int nd = as_uniform(a.preorder)[u];
            const int start = as_uniform(a.indptr)[nd], stop = as_uniform(a.indptr)[nd + 1];
            if (start == stop) continue;
            double acc[R];
            if (as_uniform(a.node_has_data)[nd]) mf_gather<R>(tipc, a.nchar, a.ntips, a.codes[(size_t)nd * a.Spad + sg], g, acc);
            else {
#pragma unroll
                for (int r = 0; r < R; r++) acc[r] = (g + 4 * r < a.k) ? 1.0 : 0.0;
            }
            for (int idx = start; idx < stop; idx++) {
                const int b = as_uniform(a.indices)[idx];
                double m[R];
                if (as_uniform(a.edge_tip)[idx] >= 0) {
                    mf_gather<R>(tipc, a.nchar, as_uniform(a.edge_tip)[idx], a.codes[(size_t)b * a.Spad + sg], g, m);
                } else {
                    double x[R];
                    mf_load<R>(a.LN + ((size_t)as_uniform(a.node_int)[b] * a.C + c) * R * a.stride, a.stride, lin, x);
                    mf_stage(lds_frag, a.fragP + ((size_t)c * a.E + idx) * nfrag, nfrag, tid);
                    mf_matvec<T>(lds_frag, a.kk4, lane, x, m);
                    double x0;
                    if (mf_is_const<R>(x, g, a.k, x0)) {
#pragma unroll
                        for (int r = 0; r < R; r++) m[r] = (g + 4 * r < a.k) ? x0 : 0.0;
                    }
                    mf_store<R>(a.EV + ((size_t)as_uniform(a.edge_int)[idx] * a.C + c) * R * a.stride, a.stride, lin, m);
                }
#pragma unroll
                for (int r = 0; r < R; r++) acc[r] *= m[r];
            }
            const int slot = as_uniform(a.node_scale)[nd];
            if (slot >= 0) {
                /* exact power-of-two rescaling of the site's vector (its k states live in 4 lanes) */
                double mx = 0.0;
#pragma unroll
                for (int r = 0; r < R; r++) mx = fmax(mx, acc[r]);
                mx = fmax(mx, __shfl_xor(mx, 16, 64));
                mx = fmax(mx, __shfl_xor(mx, 32, 64));
                double sc = 1.0;
                if (mx > 0x1p-1000 && mx < 0x1p+1000) {
                    const int e = ilogb(mx);
                    sc = ldexp(1.0, -e);
#pragma unroll
                    for (int r = 0; r < R; r++) acc[r] *= sc;
                    X += e;
                }
                if (valid && g == 0) a.SC[((size_t)slot * a.C + c) * n + sl] = sc;
            }
            mf_store<R>(a.LN + ((size_t)as_uniform(a.node_int)[nd] * a.C + c) * R * a.stride, a.stride, lin, acc);
            if (u == 0) {
                const double *rw = a.root_wd + g * R;
#pragma unroll
                for (int r = 0; r < R; r++) lh_c = fma(rw[r], acc[r], lh_c);
                lh_c += __shfl_xor(lh_c, 16, 64);
                lh_c += __shfl_xor(lh_c, 32, 64);
            }
        }
        lh_c *= as_uniform(a.cat_prior)[c];
        if (lh_c > 0.0 && X > xmax) xmax = X;
        if (valid && g == 0) { a.XC[(size_t)c * n + sl] = (double)X; a.CW[(size_t)c * n + sl] = lh_c; }
    }
    /* combine the categories at the largest exponent: LH = sum_c prior_c lh_c 2^(X_c - Xmax) */
    if (xmax == INT_MIN) xmax = 0;
    if (valid && g == 0) {
        double lh_total = 0.0;
        for (int c = 0; c < a.C; c++) {
            const double w = ldexp(1.0, (int)a.XC[(size_t)c * n + slc] - xmax);
            lh_total = fma(a.CW[(size_t)c * n + slc], w, lh_total);
            a.CW[(size_t)c * n + slc] = w;
        }
        a.LH[sl] = lh_total;
    }
}

/*
 * The down pass as a depth-first traversal of the post-order program (the structure of k_ll_mfma): the vector under
 * construction stays in registers, waiting vectors go to HBM stack slots, and every internal node vector L_a and
 * internal edge vector Ev_e is written exactly once, at the OP_MATVEC that consumes / produces it.  Program (int4):
 * observation ops y = staged code row; OP_MATVEC y = CSR edge, z = storage index of the child node, w = storage
 * index of the edge; OP_PUSH / OP_POPMUL y = stack slot; OP_SCALE y = rescaling slot or -1.
 */
struct MDownProg {
    const int4 *ops;
    int nops, nobs;
    const int *obs_nodes;
    double *slots;                /* [nslots][R][stride] */
};

template <int T, bool STORE_L>
__global__ __launch_bounds__(MF_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_down_fused_mfma(MUpArgs a, MDownProg pg)
{
    extern __shared__ double lds_frag[];          /* T * kk4 * 64 doubles, then nobs x 64 staged pattern codes */
    constexpr int R = 4 * T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
    const long sl = (long)blockIdx.x * MF_SITES + wave * 16 + (lane & 15);
    const bool valid = sl < a.n;
    const long slc = valid ? sl : a.n - 1;
    const long lin = ((long)blockIdx.x * MF_SITES + wave * 16) * 4 + lane;
    const int nfrag = T * a.kk4 * 64;
    const int NT = a.ntips + 1;
    const size_t n = (size_t)a.n;
    uint8_t *code_lds = reinterpret_cast<uint8_t *>(lds_frag + nfrag);
    mf_stage_codes(code_lds, a.codes, pg.obs_nodes, pg.nobs, a.Spad, (size_t)a.s0 + (size_t)blockIdx.x * MF_SITES, tid);
    __syncthreads();
    const int scol = wave * 16 + (lane & 15);
    const PLK_AS4 int *ops = as_uniform(reinterpret_cast<const int *>(pg.ops));
    const int root_int = as_uniform(a.node_int)[as_uniform(a.preorder)[0]];
    int xmax = INT_MIN;
    for (int c = 0; c < a.C; c++) {
        const double *tipc = a.tip + (size_t)c * NT * a.nchar * 4 * R;
        double x[R];
#pragma unroll
        for (int r = 0; r < R; r++) x[r] = (g + 4 * r < a.k) ? 1.0 : 0.0;
        int X = 0;
        for (int pc = 0; pc < pg.nops; pc++) {
            const int ox = ops[4 * pc], oy = ops[4 * pc + 1];
            const int code = ox & 0xff;
            if (code == OP_MATVEC) {
                const int oz = ops[4 * pc + 2], ow = ops[4 * pc + 3];
                if (STORE_L && oz >= 0) mf_store<R>(a.LN + ((size_t)oz * a.C + c) * R * a.stride, a.stride, lin, x);   /* the node-visit up pass reads edge vectors only; oz < 0: a node the up pass finishes inline (it rebuilds L from the tip tables) */
                mf_stage(lds_frag, a.fragP + ((size_t)c * a.E + oy) * nfrag, nfrag, tid);
                double m[R], x0;
                mf_matvec<T>(lds_frag, a.kk4, lane, x, m);
                if (mf_is_const<R>(x, g, a.k, x0)) {
#pragma unroll
                    for (int r = 0; r < R; r++) m[r] = (g + 4 * r < a.k) ? x0 : 0.0;
                }
                mf_store<R>(a.EV + ((size_t)ow * a.C + c) * R * a.stride, a.stride, lin, m);
#pragma unroll
                for (int r = 0; r < R; r++) x[r] = m[r];
            } else if (code == OP_TIP_SET || code == OP_TIP_MUL || code == OP_NODE_MUL) {
                const int t = code == OP_NODE_MUL ? a.ntips : (ox >> 8);
                double v[R];
                mf_gather<R>(tipc, a.nchar, t, code_lds[oy * MF_SITES + scol], g, v);
                if (code == OP_TIP_SET) {
#pragma unroll
                    for (int r = 0; r < R; r++) x[r] = v[r];
                } else {
#pragma unroll
                    for (int r = 0; r < R; r++) x[r] *= v[r];
                }
            } else if (code == OP_PUSH) {
                mf_store<R>(pg.slots + (size_t)oy * R * a.stride, a.stride, lin, x);
            } else if (code == OP_POPMUL) {
                double v[R];
                mf_load<R>(pg.slots + (size_t)oy * R * a.stride, a.stride, lin, v);
#pragma unroll
                for (int r = 0; r < R; r++) x[r] *= v[r];
            } else if (code == OP_SCALE) {
                if (oy >= 0) {
                    double mx = 0.0;
#pragma unroll
                    for (int r = 0; r < R; r++) mx = fmax(mx, x[r]);
                    mx = fmax(mx, __shfl_xor(mx, 16, 64));
                    mx = fmax(mx, __shfl_xor(mx, 32, 64));
                    double sc = 1.0;
                    if (mx > 0x1p-1000 && mx < 0x1p+1000) {
                        const int e = ilogb(mx);
                        sc = ldexp(1.0, -e);
#pragma unroll
                        for (int r = 0; r < R; r++) x[r] *= sc;
                        X += e;
                    }
                    if (valid && g == 0) a.SC[((size_t)oy * a.C + c) * n + sl] = sc;
                }
            }
        }
        if (STORE_L) mf_store<R>(a.LN + ((size_t)root_int * a.C + c) * R * a.stride, a.stride, lin, x);
        double lh_c = 0.0;
        const double *rw = a.root_wd + g * R;
#pragma unroll
        for (int r = 0; r < R; r++) lh_c = fma(rw[r], x[r], lh_c);
        lh_c += __shfl_xor(lh_c, 16, 64);
        lh_c += __shfl_xor(lh_c, 32, 64);
        lh_c *= as_uniform(a.cat_prior)[c];
        if (lh_c > 0.0 && X > xmax) xmax = X;
        if (valid && g == 0) { a.XC[(size_t)c * n + sl] = (double)X; a.CW[(size_t)c * n + sl] = lh_c; }
    }
    if (xmax == INT_MIN) xmax = 0;
    if (valid && g == 0) {
        double lh_total = 0.0;
        for (int c = 0; c < a.C; c++) {
            const double w = ldexp(1.0, (int)a.XC[(size_t)c * n + slc] - xmax);
            lh_total = fma(a.CW[(size_t)c * n + slc], w, lh_total);
            a.CW[(size_t)c * n + slc] = w;
        }
        a.LH[sl] = lh_total;
    }
}

/* marginal of NODE: states g, g + 4, ... of this lane's site (a site is spread over the four lanes {s, s + 16, s + 32,
 * s + 48}).  Per-site planes, or -- site sums only -- the weighted sum over the 16 sites of the wave, taken within each
 * row of 16 lanes; lane 15 of row g stores the states g + 4r.  All lanes take part. */
#define MF_OUT_M(NODE, MACC)                                                                              \
    do { if (a.MVS) {                                                                                      \
             const double ws_ = valid ? (a.wsite ? a.wsite[sl] : 1.0) * inv : 0.0;                        \
             const size_t nwv_ = (size_t)gridDim.x * (MF_BLOCK / 64), wv_ = (size_t)blockIdx.x * (MF_BLOCK / 64) + wave; \
             _Pragma("unroll") for (int r = 0; r < R; r++) {                                               \
                 const double t_ = row16_sum(MACC[r] * ws_);                                               \
                 if (g + 4 * r < a.k && (lane & 15) == 15) a.MVS[((size_t)(NODE) * a.k + g + 4 * r) * nwv_ + wv_] = t_; } \
         } else {                                                                                          \
             _Pragma("unroll") for (int r = 0; r < R; r++)                                                 \
                 if (g + 4 * r < a.k && valid) a.MV[((size_t)(NODE) * a.k + g + 4 * r) * n + sl] = MACC[r] * inv; } } while (0)

template <int T, bool DERIV, bool MARG>
__global__ __launch_bounds__(MF_BLOCK) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_up_mfma(MUpArgs a)
{
    extern __shared__ double lds_frag[];
    constexpr int R = 4 * T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
    const long sl = (long)blockIdx.x * MF_SITES + wave * 16 + (lane & 15);
    const bool valid = sl < a.n;
    const long slc = valid ? sl : a.n - 1;
    const long sg = a.s0 + slc;
    const long lin = ((long)blockIdx.x * MF_SITES + wave * 16) * 4 + lane;
    const int nfrag = T * a.kk4 * 64;
    const int NT = a.ntips + 1;
    const size_t tabc = (size_t)NT * a.nchar * 4 * R;
    const size_t n = (size_t)a.n;
    const double inv = 1.0 / a.LH[slc];
    const int root = as_uniform(a.preorder)[0];
    const double *rw = a.root_wd + g * R;

    {   /* root: forward vector = root weights; its marginal */
        double macc[R];
#pragma unroll
        for (int r = 0; r < R; r++) macc[r] = 0.0;
        for (int c = 0; c < a.C; c++) {
            double f[R];
#pragma unroll
            for (int r = 0; r < R; r++) f[r] = rw[r];
            mf_store<R>(a.FN + ((size_t)as_uniform(a.node_int)[root] * a.C + c) * R * a.stride, a.stride, lin, f);
            if (MARG) {
                double l[R];
                mf_load<R>(a.LN + ((size_t)as_uniform(a.node_int)[root] * a.C + c) * R * a.stride, a.stride, lin, l);
#pragma unroll
                for (int r = 0; r < R; r++) macc[r] = fma(as_uniform(a.cat_prior)[c] * a.CW[(size_t)c * n + slc] * f[r], l[r], macc[r]);
            }
        }
        if (MARG && (!a.node_mask || as_uniform(a.node_mask)[root])) MF_OUT_M(root, macc);
    }

    for (int u = 0; u < a.N; u++) {
        const int nd = as_uniform(a.preorder)[u];
        const int start = as_uniform(a.indptr)[nd], stop = as_uniform(a.indptr)[nd + 1];
        if (start == stop) continue;
        if (DERIV && !MARG && a.node_inline && u > 0 && as_uniform(a.node_inline)[nd]) continue;     /* finished inside its parent's visit */
        const bool has = as_uniform(a.node_has_data)[nd];
        const int chn = has ? a.codes[(size_t)nd * a.Spad + sg] : 0;
        const int slot = as_uniform(a.node_scale)[nd];
        for (int idx = start; idx < stop; idx++) {
            const int b = as_uniform(a.indices)[idx];
            const bool b_leaf = as_uniform(a.edge_tip)[idx] >= 0;
            const bool want_d = DERIV && (!a.edge_mask || as_uniform(a.edge_mask)[idx]);
            const bool want_m = MARG && (!a.node_mask || as_uniform(a.node_mask)[b]);
            const bool want_f = !b_leaf || want_m;
            if (!want_d && !want_f) continue;
            const int chb = b_leaf ? a.codes[(size_t)b * a.Spad + sg] : 0;
            /* inline child: its one or two leaves (CSR edges bs, bs + 1), their codes, the child's own observation and rescaling slot */
            const bool inl = DERIV && !MARG && !b_leaf && a.node_inline && as_uniform(a.node_inline)[b];
            const int bs = inl ? as_uniform(a.indptr)[b] : 0, bn2 = inl && as_uniform(a.indptr)[b + 1] - bs == 2;
            const int l0 = inl ? as_uniform(a.indices)[bs] : 0, l1 = bn2 ? as_uniform(a.indices)[bs + 1] : 0;
            const int t0 = inl ? as_uniform(a.edge_tip)[bs] : 0, t1 = bn2 ? as_uniform(a.edge_tip)[bs + 1] : 0;
            const int cl0 = inl ? a.codes[(size_t)l0 * a.Spad + sg] : 0, cl1 = bn2 ? a.codes[(size_t)l1 * a.Spad + sg] : 0;
            const bool bhas = inl && as_uniform(a.node_has_data)[b];
            const int cbn = bhas ? a.codes[(size_t)b * a.Spad + sg] : 0;
            const int bslot = inl ? as_uniform(a.node_scale)[b] : -1;
            const bool wl0 = inl && (!a.edge_mask || as_uniform(a.edge_mask)[bs]), wl1 = bn2 && (!a.edge_mask || as_uniform(a.edge_mask)[bs + 1]);
            double dl0 = 0.0, dl1 = 0.0;
            double dsum = 0.0;
            double macc[R];
#pragma unroll
            for (int r = 0; r < R; r++) macc[r] = 0.0;
            for (int c = 0; c < a.C; c++) {
                const double *tipc = a.tip + (size_t)c * tabc;
                double fe[R];
                mf_load<R>(a.FN + ((size_t)as_uniform(a.node_int)[nd] * a.C + c) * R * a.stride, a.stride, lin, fe);
                if (has) {
                    double bn[R];
                    mf_gather<R>(tipc, a.nchar, a.ntips, chn, g, bn);
#pragma unroll
                    for (int r = 0; r < R; r++) fe[r] *= bn[r];
                }
                if (slot >= 0) {
                    /* the forward vector below a rescaled node carries that node's factor (L_a was scaled after
                     * the child messages were multiplied in) */
                    const double sc = a.SC[((size_t)slot * a.C + c) * n + slc];
#pragma unroll
                    for (int r = 0; r < R; r++) fe[r] *= sc;
                }
                for (int idx2 = start; idx2 < stop; idx2++) {
                    if (idx2 == idx) continue;
                    double ev[R];
                    if (as_uniform(a.edge_tip)[idx2] >= 0)
                        mf_gather<R>(tipc, a.nchar, as_uniform(a.edge_tip)[idx2], a.codes[(size_t)as_uniform(a.indices)[idx2] * a.Spad + sg], g, ev);
                    else
                        mf_load<R>(a.EV + ((size_t)as_uniform(a.edge_int)[idx2] * a.C + c) * R * a.stride, a.stride, lin, ev);
#pragma unroll
                    for (int r = 0; r < R; r++) fe[r] *= ev[r];
                }
                const double prior = as_uniform(a.cat_prior)[c] * a.CW[(size_t)c * n + slc];
                if (want_d) {
                    double y[R];
                    if (b_leaf) {
                        mf_gather<R>(a.dtip + (size_t)c * tabc, a.nchar, as_uniform(a.edge_tip)[idx], chb, g, y);
                    } else {
                        double x[R], x0;
                        if (!inl) mf_load<R>(a.LN + ((size_t)as_uniform(a.node_int)[b] * a.C + c) * R * a.stride, a.stride, lin, x);
                        else {
                            /* L_b rebuilt in the order the down pass multiplied it: leaf rows, own observation, rescaling factor */
                            mf_gather<R>(tipc, a.nchar, t0, cl0, g, x);
                            if (bn2) {
                                double m2[R];
                                mf_gather<R>(tipc, a.nchar, t1, cl1, g, m2);
#pragma unroll
                                for (int r = 0; r < R; r++) x[r] *= m2[r];
                            }
                            if (bhas) {
                                double m2[R];
                                mf_gather<R>(tipc, a.nchar, a.ntips, cbn, g, m2);
#pragma unroll
                                for (int r = 0; r < R; r++) x[r] *= m2[r];
                            }
                            if (bslot >= 0) {
                                const double sb = a.SC[((size_t)bslot * a.C + c) * n + slc];
#pragma unroll
                                for (int r = 0; r < R; r++) x[r] *= sb;
                            }
                        }
                        mf_stage(lds_frag, a.fragD + ((size_t)c * a.E + idx) * nfrag, nfrag, tid);
                        mf_matvec<T>(lds_frag, a.kk4, lane, x, y);
                        if (a.dzero && mf_is_const<R>(x, g, a.k, x0)) {
#pragma unroll
                            for (int r = 0; r < R; r++) y[r] = 0.0;
                        }
                    }
                    double d = 0.0;
#pragma unroll
                    for (int r = 0; r < R; r++) d = fma(fe[r], y[r], d);
                    d += __shfl_xor(d, 16, 64);
                    d += __shfl_xor(d, 32, 64);
                    dsum = fma(prior, d, dsum);
                }
                if (want_f) {
                    double fb[R];
                    mf_stage(lds_frag, a.fragPT + ((size_t)c * a.E + idx) * nfrag, nfrag, tid);
                    mf_matvec<T>(lds_frag, a.kk4, lane, fe, fb);
                    if (!b_leaf && !inl) mf_store<R>(a.FN + ((size_t)as_uniform(a.node_int)[b] * a.C + c) * R * a.stride, a.stride, lin, fb);
                    if (inl && (wl0 || wl1)) {
                        /* the child's leaf edges, while its forward vector is in registers: fb o B_b x s_b, then per leaf the
                         * edge-form tip row times the other leaf's message (what b's own visit would have computed) */
                        if (bhas) {
                            double m2[R];
                            mf_gather<R>(tipc, a.nchar, a.ntips, cbn, g, m2);
#pragma unroll
                            for (int r = 0; r < R; r++) fb[r] *= m2[r];
                        }
                        if (bslot >= 0) {
                            const double sb = a.SC[((size_t)bslot * a.C + c) * n + slc];
#pragma unroll
                            for (int r = 0; r < R; r++) fb[r] *= sb;
                        }
                        if (wl0) {
                            double y2[R];
                            mf_gather<R>(a.dtip + (size_t)c * tabc, a.nchar, t0, cl0, g, y2);
                            if (bn2) {
                                double m2[R];
                                mf_gather<R>(tipc, a.nchar, t1, cl1, g, m2);
#pragma unroll
                                for (int r = 0; r < R; r++) y2[r] *= m2[r];
                            }
                            double d2 = 0.0;
#pragma unroll
                            for (int r = 0; r < R; r++) d2 = fma(fb[r], y2[r], d2);
                            d2 += __shfl_xor(d2, 16, 64);
                            d2 += __shfl_xor(d2, 32, 64);
                            dl0 = fma(prior, d2, dl0);
                        }
                        if (wl1) {
                            double y2[R], m2[R];
                            mf_gather<R>(a.dtip + (size_t)c * tabc, a.nchar, t1, cl1, g, y2);
                            mf_gather<R>(tipc, a.nchar, t0, cl0, g, m2);
#pragma unroll
                            for (int r = 0; r < R; r++) y2[r] *= m2[r];
                            double d2 = 0.0;
#pragma unroll
                            for (int r = 0; r < R; r++) d2 = fma(fb[r], y2[r], d2);
                            d2 += __shfl_xor(d2, 16, 64);
                            d2 += __shfl_xor(d2, 32, 64);
                            dl1 = fma(prior, d2, dl1);
                        }
                    }
                    if (want_m) {
                        double lb[R];
                        if (b_leaf) mf_gather<R>(tipc, a.nchar, a.ntips, chb, g, lb);
                        else mf_load<R>(a.LN + ((size_t)as_uniform(a.node_int)[b] * a.C + c) * R * a.stride, a.stride, lin, lb);
#pragma unroll
                        for (int r = 0; r < R; r++) macc[r] = fma(prior * fb[r], lb[r], macc[r]);
                    }
                }
            }
            if (want_d && valid && g == 0) a.DV[(size_t)idx * n + sl] = dsum * inv;
            if (wl0 && valid && g == 0) a.DV[(size_t)bs * n + sl] = dl0 * inv;
            if (wl1 && valid && g == 0) a.DV[(size_t)(bs + 1) * n + sl] = dl1 * inv;
            if (want_m) MF_OUT_M(b, macc);
        }
    }
}

/*
 * Up pass of the derivative query by node visits (records: plk_up_nodes_build() of plk_program.h).
 * k_up_mfma above reads, per internal edge, F_a, the sibling's edge vector and L_b: 81 GB per 500 k codon sites
 * (rocprofv3 FETCH_SIZE, profiles/r02_cfg5_*), and the down pass writes both L_b and the edge vector P_b L_b for it.
 * Here the down pass stores edge vectors only (k_down_fused_mfma<T, false>), and a visit of node a
 *   - takes G_a (top of a's own edge) from registers or from FN,
 *   - derivative of a's own edge = sum of (M_a^T G_a) o B_a o (all child messages) x s_a, the factors multiplied in one
 *     at a time as they are read (stored edge vectors of internal children, tip-table rows of leaves); L_a is not read,
 *   - F_a = P_a^T G_a, and for every child with work G_b = F_a o B_a o s_a o (the other children's messages): leaf
 *     edges are finished with the edge-form tip tables, G_b of internal children is stored, except the last one,
 *     which the next visit takes over in registers.
 * Two products per internal node as before, a quarter of the reads.  a.fragD holds the fragments of M^T here.
 *
 * Register discipline: three vectors live at a time (gv / one product / one factor being read).  There is one code
 * path for every node degree, every vector has one unconditional definition that dominates its uses, and a factor is
 * read by one load sequence with selected (wave-uniform) base and step rather than by two arms of a branch: a vector
 * written in both arms of a wave-uniform if / else is live from the top of the kernel for the register allocator (it
 * sees a path around both arms).  All addressing is "SGPR base + 32-bit lane offset".  tools/vgpr_liveness.py
 * prints the live sets this was tuned with.
 */
template <int T>
__global__ __launch_bounds__(MF_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_up_nodes_mfma(MUpArgs a)
{
    extern __shared__ double lds_frag[];
    constexpr int R = 4 * T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
    const long sl = (long)blockIdx.x * MF_SITES + wave * 16 + (lane & 15);
    const bool valid = sl < a.n;
    const unsigned slc = (unsigned)(valid ? sl : a.n - 1), ug = (unsigned)g;     /* 32-bit lane offsets: every base below is wave-uniform */
    const long lin = ((long)blockIdx.x * MF_SITES + wave * 16) * 4 + lane;
    const int nfrag = T * a.kk4 * 64;
    const int NT = a.ntips + 1;
    const size_t tabc = (size_t)NT * a.nchar * 4 * R;
    const size_t n = (size_t)a.n;
    const double inv = 1.0 / mfn_at(a.LH, slc);
    const PLK_AS4 int *vis = as_uniform(a.visits);
    const PLK_AS4 int *eint = as_uniform(a.edge_int);

    for (int c = 0; c < a.C; c++) {
        const double *tipc = a.tip + (size_t)c * tabc;
        const double *dtipc = a.dtip + (size_t)c * tabc;
        const double pc = as_uniform(a.cat_prior)[c] * mfn_at(a.CW + (size_t)c * n, slc);
        const bool first_cat = c == 0, last_cat = c == a.C - 1;
#define MUN_OUT_D(EDGE, VAL)                                                                              \
        do { double t_ = (VAL); t_ += __shfl_xor(t_, 16, 64); t_ += __shfl_xor(t_, 32, 64);               \
             if (valid && g == 0) { double *dp_ = a.DV + (size_t)(EDGE) * n; asm volatile("" : "+s"(dp_));  \
                 dp_[slc] = (first_cat ? t_ : dp_[slc] + t_) * (last_cat ? inv : 1.0); } } while (0)
        /* message of the child in record J into M: the tip-table row of its pattern code, or its stored edge vector */
#define MUN_MESSAGE(J, M)                                                                                 \
        do { const int b_ = ch[4 * (J)], t_ = ch[4 * (J) + 1], pos_ = ch[4 * (J) + 2] >> PLK_UN_POS_SHIFT; \
             const unsigned cd_ = mfn_code(a.codes, a.Spad, a.s0, b_, slc);                                \
             const bool tip_ = t_ >= 0;                                                                   \
             const double *base_ = tip_ ? tipc + (size_t)t_ * a.nchar * 4 * R                            \
                                        : a.EV + ((size_t)(tip_ ? 0 : eint[e0 + pos_]) * a.C + c) * R * a.stride; \
             const size_t step_ = tip_ ? 1 : (size_t)a.stride;                                            \
             const unsigned off_ = tip_ ? (cd_ * 4u + ug) * (unsigned)R : (unsigned)lin;                  \
             _Pragma("unroll") for (int r = 0; r < R; r++) M[r] = mfn_at(base_ + (size_t)r * step_, off_); } while (0)
        /* the node's own observation (tip-table slot ntips) */
#define MUN_OWN(M)                                                                                        \
        do { const unsigned cd_ = mfn_code(a.codes, a.Spad, a.s0, nd, slc);                                \
             const double *base_ = tipc + (size_t)a.ntips * a.nchar * 4 * R;                              \
             const unsigned off_ = (cd_ * 4u + ug) * (unsigned)R;                                          \
             _Pragma("unroll") for (int r = 0; r < R; r++) M[r] = mfn_at(base_ + r, off_); } while (0)
        double gv[R];
#pragma unroll
        for (int r = 0; r < R; r++) gv[r] = mfn_at(a.root_wd, ug * (unsigned)R + r);   /* the root's visit comes first and takes it from registers */
        int vp = 0;
        for (int v = 0; v < a.nvisits; v++) {
            const int nd = vis[vp], deg = vis[vp + 1], nd_int = vis[vp + 2], slot = vis[vp + 3], hd = vis[vp + 4], e0 = vis[vp + 5];
            const int ea = vis[vp + 6], hfl = vis[vp + 7];
            const PLK_AS4 int *ch = vis + vp + 8;
            vp += 8 + 4 * deg;
            const double sc = slot >= 0 ? mfn_at(a.SC + ((size_t)slot * a.C + c) * n, slc) : 1.0;
            if (!(hfl & PLK_UN_FROM_REGS)) mfn_load<R>(a.FN + ((size_t)nd_int * a.C + c) * R * a.stride, a.stride, lin, gv);
            if (hfl & PLK_UN_OWN_D) {
                double z[R], x0;
                mf_stage(lds_frag, a.fragD + ((size_t)c * a.E + ea) * nfrag, nfrag, tid);
                mf_matvec<T>(lds_frag, a.kk4, lane, gv, z);
                /* (M_a^T G_a) . L_a with L_a = s_a B_a o (all messages), one factor at a time.  Zero row sums and every
                 * factor constant (then L_a is): exactly zero, src/util.c:338-345 -- non-constant factors with a
                 * constant product, which the reference would also zero, give a rounding-level value here */
                bool cst = true;
                if (hd) {
                    double m[R];
                    MUN_OWN(m);
                    cst = mfn_is_const<R>(m, g, a.k, x0);
#pragma unroll
                    for (int r = 0; r < R; r++) z[r] *= m[r];
                }
                for (int j = 0; j < deg; j++) {
                    double m[R];
                    MUN_MESSAGE(j, m);
                    cst = mfn_is_const<R>(m, g, a.k, x0) && cst;
#pragma unroll
                    for (int r = 0; r < R; r++) z[r] *= m[r];
                }
                double d = 0.0;
#pragma unroll
                for (int r = 0; r < R; r++) d += z[r];
                if (a.dzero && cst) d = 0.0;
                MUN_OUT_D(ea, pc * sc * d);
            }
            /* F_a = P_a^T G_a; the root runs the same product (any edge's matrix) and then keeps its prior: one
             * unconditional definition of fe */
            double fe[R];
            mf_stage(lds_frag, a.fragPT + ((size_t)c * a.E + (ea >= 0 ? ea : 0)) * nfrag, nfrag, tid);
            mf_matvec<T>(lds_frag, a.kk4, lane, gv, fe);
#pragma unroll
            for (int r = 0; r < R; r++) fe[r] = (ea >= 0 ? fe[r] : gv[r]) * sc;
            if (hd) {
                double m[R];
                MUN_OWN(m);
#pragma unroll
                for (int r = 0; r < R; r++) fe[r] *= m[r];
            }
            /* children in record order (the one that continues in registers is last); gv is re-formed whether or not
             * anything continues */
#pragma unroll
            for (int r = 0; r < R; r++) gv[r] = fe[r];
            for (int j = 0; j < deg; j++) {
                const int b = ch[4 * j], t = ch[4 * j + 1], fl = ch[4 * j + 2], bi = ch[4 * j + 3];
                if (!(fl & PLK_UN_WORK)) continue;
#pragma unroll
                for (int r = 0; r < R; r++) gv[r] = fe[r];
                for (int j2 = 0; j2 < deg; j2++) {
                    if (j2 == j) continue;
                    double m[R];
                    MUN_MESSAGE(j2, m);
#pragma unroll
                    for (int r = 0; r < R; r++) gv[r] *= m[r];
                }
                if (fl & PLK_UN_LEAF_D) {
                    double y[R], d = 0.0;
                    const unsigned cd = mfn_code(a.codes, a.Spad, a.s0, b, slc);
                    mfn_gather<R>(dtipc, a.nchar, t, cd, ug, y);
#pragma unroll
                    for (int r = 0; r < R; r++) d = fma(gv[r], y[r], d);
                    MUN_OUT_D(e0 + (fl >> PLK_UN_POS_SHIFT), pc * d);
                } else if (fl & PLK_UN_STORE_G) mfn_store<R>(a.FN + ((size_t)bi * a.C + c) * R * a.stride, a.stride, lin, gv);
            }
        }
#undef MUN_OUT_D
#undef MUN_MESSAGE
#undef MUN_OWN
    }
}

/* A fragments of an edge-indexed matrix set M[C*E][k][k]: transpose = 0: of M, 1: of M^T */
__global__ void k_build_frag_edges(int k, int T, int kk4, int transpose, const double *__restrict__ M, double *__restrict__ frag)
{
    const size_t ce = blockIdx.x;
    const double *src = M + ce * k * k;
    double *dst = frag + ce * T * kk4 * 64;
    const int nn = T * kk4 * 64;
    for (int idx = threadIdx.x; idx < nn; idx += blockDim.x) {
        const int l = idx & 63, tq = idx >> 6;
        const int t = tq / kk4, q = tq - t * kk4;
        const int i = 16 * t + (l & 15), j = 4 * q + (l >> 4);
        double v = 0.0;
        if (i < k && j < k) v = transpose ? src[(size_t)j * k + i] : src[(size_t)i * k + j];
        dst[idx] = v;
    }
}

/* derivative tip table: dtip[...][g][r] = (dP_e defs[code])[g + 4r], zero for constant rows
 * (rows of dP sum to zero; the reference's exact shortcut, src/util.c:338-345) */
__global__ void k_build_dtip_dist(int k, int R, int E, int ntips, int nchar, const int *__restrict__ tip_edge,
                                  const double *__restrict__ dP, const double *__restrict__ defs, int Kpad,
                                  double *__restrict__ dtip, int dzero)
{
    __shared__ int s_kind[PLK_DEF_KIND_CACHE];
    const int t = blockIdx.x, c = blockIdx.y;
    const int e = tip_edge[t];
    const int nn = nchar * 4 * R;
    for (int code = threadIdx.x; code < nchar && code < PLK_DEF_KIND_CACHE; code += blockDim.x) s_kind[code] = def_row_kind(defs + (size_t)code * Kpad, k);
    __syncthreads();
    for (int idx = threadIdx.x; idx < nn; idx += blockDim.x) {
        const int r = idx % R, gq = idx / R;
        const int g = gq & 3, code = gq >> 2;
        const int i = g + 4 * r;
        const double *d = defs + (size_t)code * Kpad;
        double out = 0.0;
        if (i < k && e >= 0) {
            const int kind = code < PLK_DEF_KIND_CACHE ? s_kind[code] : def_row_kind(d, k);
            if (!(kind == -2 && dzero)) {
                const double *row = dP + ((size_t)c * E + e) * k * k + (size_t)i * k;
                if (kind >= 0) out = row[kind];             /* an observed state: column `kind` of dP_e (one non-zero term, exact) */
                else {
                    dd acc = dd_make(0.0, 0.0);
                    for (int j = 0; j < k; j++) acc = dd_add(acc, dd_two_prod(row[j], d[j]));
                    out = acc.hi;
                }
            }
        }
        dtip[(((size_t)c * (ntips + 1) + t) * nchar + code) * 4 * R + (size_t)g * R + r] = out;
    }
}

#endif
