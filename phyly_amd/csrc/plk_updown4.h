/*
 * plk_updown4.h -- down pass with stored vectors and BFS-order up pass for k = 4 with
 * compact character data: the nucleotide fast path of arbplf-deriv / arbplf-marginal.
 * Included by plk_engine.hip.
 *
 * Same formulas as k_down_store / k_up (src/evaluate_site_lhood.c:7-63 with edge vectors,
 * src/evaluate_site_forward.c:32-105, src/arbplfderiv.c:112-207,:312-342,
 * src/evaluate_site_marginal.c:7-21, src/arbplfmarginal.c:206-234), specialised for the
 * HBM roofline that bounds them:
 *   - stored vectors are interleaved [entity][category][site][4]: one lane moves its whole
 *     4-vector with two 16-byte accesses, a wavefront moves 2 KB contiguously;
 *   - only the node vectors L_a and F_a of internal nodes are stored; edge vectors
 *     Ev_e = P_e L_b are recomputed in the up pass from the L_b that the edge form needs
 *     anyway (all children of a node are handled together, so L_b and F_a are read once);
 *     leaf-edge vectors P_e B_b and M_e B_b are gathered by pattern code from small tables
 *     (double-double built) that live in L1/L2;
 *   - P_e, P_e^T and M_e (dP or a Frechet matrix) are wave-uniform: scalar loads, SGPR
 *     operands, no LDS;
 *   - node vectors are rescaled by exact powers of two at the nodes the traversal program
 *     marks (every >= 16 accumulated edges); the factors are stored (SC) and re-applied to
 *     the forward vectors, categories are combined at a common exponent (CW), so trees of
 *     any size stay inside the double range.
 */
#ifndef PLK_UPDOWN4_H
#define PLK_UPDOWN4_H

struct Up4Args {
    long S, Spad, s0, n;
    int N, E, C, nchar, ntips, root_mode;
    int dzero;                     /* 1: edge-form matrices have zero row sums (dP) */
    const int *indptr, *indices, *preorder;
    const int *node_has_data, *edge_tip, *edge_int, *node_int;
    const double *P, *dP;          /* [C][E][4][4] row-major; dP: [nM] such sets (the edge-form matrices) */
    const double *tip, *dtip;      /* [C][ntips+1][nchar][4]; slot ntips of tip = raw definitions; dtip: [nM] sets */
    int nM;                        /* edge forms evaluated in this pass (1 for deriv; DV is [nM][E][n]) */
    const uint8_t *codes;
    const double *cat_prior, *root_w;
    const int *edge_mask, *node_mask;
    const int *node_scale;         /* N: index of the node's rescaling slot, -1 = not rescaled */
    const int *node_inline;        /* N: 1 = all children are leaves (at most two) and the parent has at most two
                                      children: the up pass handles the node inside its parent's visit, so its
                                      forward vector is never stored */
    double *LN, *FN;               /* [(ent*C + c)][n][4] */
    double *SC;                    /* [(slot*C + c)][n]: 2^-e applied to L_a at rescaled nodes */
    double *CW, *XC;               /* [C][n]: 2^(X_c - Xmax); scratch: X_c and category likelihood mantissas */
    double *LH, *DV, *MV;          /* [n] (at exponent Xmax), [E][n], [N][4][n] */
    /* site-summed marginals (no per-site output asked for): MV is not written; every wave leaves the weighted sum of its
     * 64 sites in MVS[(node * 4 + state) * nwaves + wave] (src/arbplfmarginal.c:237-256 accumulates as it goes, too) */
    double *MVS;                   /* null: per-site planes in MV */
    const double *wsite;           /* [n] site weights of the chunk or null */
    const int *visits;             /* k_up4_nodes: records of plk_up_nodes_build() (plk_program.h) */
    const double *ptab;            /* k_up4_nodes: pair tables [C][npairs][nchar * nchar][4]: the message P_a (P_b B_b o P_c B_c) of a
                                      two-leaf node, by the combined code of its leaves (k_build_tables_pt); null = none */
    int npairs;
    int nvisits;
    const int *rebuild;            /* k_up4_nodes: table of plk_up_rebuild_table() (children whose L is a product of two table rows) */
};

struct v4 { double a, b, c, d; };

__device__ static inline v4 ld4(const double *p)
{
    const double2 lo = reinterpret_cast<const double2 *>(p)[0], hi = reinterpret_cast<const double2 *>(p)[1];
    return v4{lo.x, lo.y, hi.x, hi.y};
}
__device__ static inline void st4(double *p, const v4 &v)
{
    reinterpret_cast<double2 *>(p)[0] = double2{v.a, v.b};
    reinterpret_cast<double2 *>(p)[1] = double2{v.c, v.d};
}
__device__ static inline v4 mul4(const v4 &x, const v4 &y) { return v4{x.a * y.a, x.b * y.b, x.c * y.c, x.d * y.d}; }
__device__ static inline bool const4(const v4 &x) { return x.a == x.b && x.a == x.c && x.a == x.d; }

/* y = M x, M row-major 4x4 behind a uniform pointer */
__device__ static inline v4 mv4(const PLK_AS4 double *M, const v4 &x)
{
    v4 y;
    y.a = fma(M[3], x.d, fma(M[2], x.c, fma(M[1], x.b, M[0] * x.a)));
    y.b = fma(M[7], x.d, fma(M[6], x.c, fma(M[5], x.b, M[4] * x.a)));
    y.c = fma(M[11], x.d, fma(M[10], x.c, fma(M[9], x.b, M[8] * x.a)));
    y.d = fma(M[15], x.d, fma(M[14], x.c, fma(M[13], x.b, M[12] * x.a)));
    return y;
}
/* y = M^T x */
__device__ static inline v4 mtv4(const PLK_AS4 double *M, const v4 &x)
{
    v4 y;
    y.a = fma(M[12], x.d, fma(M[8], x.c, fma(M[4], x.b, M[0] * x.a)));
    y.b = fma(M[13], x.d, fma(M[9], x.c, fma(M[5], x.b, M[1] * x.a)));
    y.c = fma(M[14], x.d, fma(M[10], x.c, fma(M[6], x.b, M[2] * x.a)));
    y.d = fma(M[15], x.d, fma(M[11], x.c, fma(M[7], x.b, M[3] * x.a)));
    return y;
}

#define UD4_BLOCK 256

/* child message: table gather for a leaf edge, P_e L_b (exact for constant L_b, src/util.c:276-283) otherwise */
__device__ static inline v4 ud4_child_msg(const Up4Args &a, int c, int idx, int t, int code, const double *tipc,
                                          const PLK_AS4 double *Pm, const v4 &x)
{
    if (t >= 0) return ld4(tipc + ((size_t)t * a.nchar + code) * 4);
    v4 m = mv4(Pm + ((size_t)c * a.E + idx) * 16, x);
    if (const4(x)) m = x;
    return m;
}

__global__ __launch_bounds__(UD4_BLOCK) void k_down_store4(Up4Args a)
{
    const long sl = (long)blockIdx.x * UD4_BLOCK + threadIdx.x;
    const bool valid = sl < a.n;
    const long slc = valid ? sl : a.n - 1;
    const long sg = a.s0 + slc;
    const size_t n = (size_t)a.n;
    const PLK_AS4 int *pre = as_uniform(a.preorder), *ip = as_uniform(a.indptr), *ix = as_uniform(a.indices);
    const PLK_AS4 int *has = as_uniform(a.node_has_data), *etip = as_uniform(a.edge_tip);
    const PLK_AS4 int *nint = as_uniform(a.node_int), *nsc = as_uniform(a.node_scale);
    const PLK_AS4 double *Pm = as_uniform(a.P), *prior = as_uniform(a.cat_prior), *rw = as_uniform(a.root_w);
    const size_t tabc = (size_t)(a.ntips + 1) * a.nchar * 4;
    int xmax = INT_MIN;
    for (int c = 0; c < a.C; c++) {
        const double *tipc = a.tip + (size_t)c * tabc;
        double lh_c = 0.0;
        int X = 0;
        for (int u = a.N - 1; u >= 0; u--) {
            const int nd = pre[u];
            const int start = ip[nd], stop = ip[nd + 1];
            if (start == stop) continue;
            v4 acc = v4{1.0, 1.0, 1.0, 1.0};
            if (has[nd]) acc = ld4(tipc + ((size_t)a.ntips * a.nchar + a.codes[(size_t)nd * a.Spad + sg]) * 4);
            for (int idx = start; idx < stop; idx++) {
                const int b = ix[idx];
                const int t = etip[idx];
                v4 x = v4{0.0, 0.0, 0.0, 0.0};
                int code = 0;
                if (t >= 0) code = a.codes[(size_t)b * a.Spad + sg];
                else x = ld4(a.LN + (((size_t)nint[b] * a.C + c) * n + slc) * 4);
                acc = mul4(acc, ud4_child_msg(a, c, idx, t, code, tipc, Pm, x));
            }
            const int slot = nsc[nd];
            if (slot >= 0) {
                const double mx = fmax(fmax(acc.a, acc.b), fmax(acc.c, acc.d));
                double sc = 1.0;
                if (mx > 0x1p-1000 && mx < 0x1p+1000) {
                    const int e = ilogb(mx);
                    sc = ldexp(1.0, -e);
                    acc.a *= sc; acc.b *= sc; acc.c *= sc; acc.d *= sc;
                    X += e;
                }
                if (valid) a.SC[((size_t)slot * a.C + c) * n + slc] = sc;
            }
            if (valid) st4(a.LN + (((size_t)nint[nd] * a.C + c) * n + slc) * 4, acc);
            if (u == 0) {
                if (a.root_mode == PLK_ROOT_NONE) lh_c = ((acc.a + acc.b) + acc.c) + acc.d;
                else if (a.root_mode == PLK_ROOT_UNIFORM) lh_c = (((acc.a + acc.b) + acc.c) + acc.d) * 0.25;
                else lh_c = fma(rw[3], acc.d, fma(rw[2], acc.c, fma(rw[1], acc.b, rw[0] * acc.a)));
            }
        }
        lh_c *= prior[c];
        if (lh_c > 0.0 && X > xmax) xmax = X;
        if (valid) { a.XC[(size_t)c * n + slc] = (double)X; a.CW[(size_t)c * n + slc] = lh_c; }
    }
    /* combine the categories at the largest exponent: LH = sum_c prior_c lh_c 2^(X_c - Xmax) */
    if (xmax == INT_MIN) xmax = 0;
    double lh_total = 0.0;
    if (valid) {
        for (int c = 0; c < a.C; c++) {
            const double w = ldexp(1.0, (int)a.XC[(size_t)c * n + slc] - xmax);
            lh_total = fma(a.CW[(size_t)c * n + slc], w, lh_total);
            a.CW[(size_t)c * n + slc] = w;
        }
        a.LH[sl] = lh_total;
    }
}

/*
 * The same down pass as a depth-first traversal of the post-order program (OP_* of plk_engine.hip): the vector
 * under construction stays in registers, vectors waiting for a sibling subtree are parked in AGPRs (as in the
 * fused ll kernels), so every internal node vector is written once and never read back here -- half the HBM
 * traffic of k_down_store4.  A node's vector is final when its parent multiplies it by P (OP_MATVEC) or, for the
 * root, when the program ends.
 */
template <int D>
__global__ __launch_bounds__(UD4_BLOCK) void k_down_fused4(Up4Args a, const int4 *ops_, const int *op_edge_, int nops,
                                                           const int *obs_nodes, int nobs, int first_slot, int first_row)
{
    if constexpr (D <= 4) asm volatile("" ::: PLK_CLOBBER_A0_31);
    else if constexpr (D <= 8) asm volatile("" ::: PLK_CLOBBER_A0_31, PLK_CLOBBER_A32_63);
    else asm volatile("" ::: PLK_CLOBBER_A0_31, PLK_CLOBBER_A32_63, PLK_CLOBBER_A64_127);
    extern __shared__ uint8_t ud4_codes[];            /* nobs x 256 pattern codes of this block's sites */
    const int tid = threadIdx.x;
    const long sl = (long)blockIdx.x * UD4_BLOCK + tid;
    const bool valid = sl < a.n;
    const long slc = valid ? sl : a.n - 1;
    const size_t n = (size_t)a.n;
    {
        /* rows are padded to Spad (a multiple of 1024) and chunks start at multiples of 256; wave w stages rows w, w + 4,
         * ... with the node looked up through a scalar load and eight rows in flight (see k_ll_fused4_asm) */
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const PLK_AS4 int *obs = as_uniform(obs_nodes);
        uint32_t *dst = reinterpret_cast<uint32_t *>(ud4_codes);
        for (int r0 = wave; r0 < nobs; r0 += 32) {
            uint32_t q[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int row = r0 + 4 * u;
                q[u] = row < nobs ? reinterpret_cast<const uint32_t *>(a.codes + (size_t)obs[row] * a.Spad + a.s0 +
                                                                         (size_t)blockIdx.x * UD4_BLOCK)[lane] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int row = r0 + 4 * u;
                if (row < nobs) dst[row * (UD4_BLOCK / 4) + lane] = q[u];
            }
        }
    }
    __syncthreads();
    const PLK_AS4 int *ops = as_uniform(reinterpret_cast<const int *>(ops_));
    const PLK_AS4 int *nint = as_uniform(a.node_int);
    const PLK_AS4 double *Pm = as_uniform(a.P), *prior = as_uniform(a.cat_prior), *rw = as_uniform(a.root_w);
    const size_t tabc = (size_t)(a.ntips + 1) * a.nchar * 4;
    const int root = as_uniform(a.preorder)[0];
    (void)op_edge_;
    int xmax = INT_MIN;
    /* observation values are gathered one observation op ahead (the code comes from LDS, so the address of the
     * next gather is known as soon as the current op starts) */
    v4 nxt = v4{1.0, 1.0, 1.0, 1.0};
    if (first_slot >= 0) nxt = ld4(a.tip + ((size_t)first_slot * a.nchar + ud4_codes[first_row * UD4_BLOCK + tid]) * 4);
    for (int c = 0; c < a.C; c++) {
        v4 cur = v4{1.0, 1.0, 1.0, 1.0};
        int X = 0;
        /* op words are fetched one op ahead (the program is padded with one OP_END) */
        int nx = ops[0], ny = ops[1], nz = ops[2], nw = ops[3];
        for (int pc = 0; pc < nops; pc++) {
            const int ox = nx, oy = ny, oz = nz, ow = nw;
            nx = ops[4 * pc + 4]; ny = ops[4 * pc + 5]; nz = ops[4 * pc + 6]; nw = ops[4 * pc + 7];
            const int code = ox & 0xff;
            if (code == OP_MATVEC) {
                /* y = CSR edge, z = storage index of the child node whose vector is now final */
                if (valid && oz >= 0) st4(a.LN + (((size_t)oz * a.C + c) * n + slc) * 4, cur);   /* oz < 0: a two-leaf node whose message the up pass takes from a pair table */
                v4 m = mv4(Pm + ((size_t)c * a.E + oy) * 16, cur);
                if (const4(cur)) m = cur;
                cur = m;
            } else if (code == OP_TIP_SET || code == OP_TIP_MUL || code == OP_NODE_MUL) {
                const v4 v = nxt;
                const int wrap = (oz >> 30) & 1;
                if (!wrap || c + 1 < a.C)
                    nxt = ld4(a.tip + (size_t)(c + wrap) * tabc +
                              ((size_t)(oz & 0x3fffffff) * a.nchar + ud4_codes[ow * UD4_BLOCK + tid]) * 4);
                cur = code == OP_TIP_SET ? v : mul4(cur, v);
            } else if (code == OP_PUSH) {
                stack_push<D, 1>(oy, 0, cur.a, cur.b, cur.c, cur.d);
            } else if (code == OP_POPMUL) {
                stack_popmul<D, 1>(oy, 0, cur.a, cur.b, cur.c, cur.d);
            } else if (code == OP_SCALE) {
                /* y = rescaling slot of the node (-1: none) */
                if (oy >= 0) {
                    const double mx = fmax(fmax(cur.a, cur.b), fmax(cur.c, cur.d));
                    double sc = 1.0;
                    if (mx > 0x1p-1000 && mx < 0x1p+1000) {
                        const int ex = ilogb(mx);
                        sc = ldexp(1.0, -ex);
                        cur.a *= sc; cur.b *= sc; cur.c *= sc; cur.d *= sc;
                        X += ex;
                    }
                    if (valid) a.SC[((size_t)oy * a.C + c) * n + slc] = sc;
                }
            }
        }
        if (valid) st4(a.LN + (((size_t)nint[root] * a.C + c) * n + slc) * 4, cur);
        double lh_c;
        if (a.root_mode == PLK_ROOT_NONE) lh_c = ((cur.a + cur.b) + cur.c) + cur.d;
        else if (a.root_mode == PLK_ROOT_UNIFORM) lh_c = (((cur.a + cur.b) + cur.c) + cur.d) * 0.25;
        else lh_c = fma(rw[3], cur.d, fma(rw[2], cur.c, fma(rw[1], cur.b, rw[0] * cur.a)));
        lh_c *= prior[c];
        if (lh_c > 0.0 && X > xmax) xmax = X;
        if (valid) { a.XC[(size_t)c * n + slc] = (double)X; a.CW[(size_t)c * n + slc] = lh_c; }
    }
    if (xmax == INT_MIN) xmax = 0;
    if (valid) {
        double lh_total = 0.0;
        for (int c = 0; c < a.C; c++) {
            const double w = ldexp(1.0, (int)a.XC[(size_t)c * n + slc] - xmax);
            lh_total = fma(a.CW[(size_t)c * n + slc], w, lh_total);
            a.CW[(size_t)c * n + slc] = w;
        }
        a.LH[sl] = lh_total;
    }
}

/* per-child state of the up pass */
struct Ud4Child {
    int idx, b, t, code;
    bool want_d, want_m, want_f;
};

__device__ static inline Ud4Child ud4_child(const Up4Args &a, int idx, long sg)
{
    Ud4Child ch;
    ch.idx = idx;
    ch.b = as_uniform(a.indices)[idx];
    ch.t = as_uniform(a.edge_tip)[idx];
    ch.code = ch.t >= 0 ? a.codes[(size_t)ch.b * a.Spad + sg] : 0;
    return ch;
}


/* an internal node whose (one or two) children are all leaves, handled inside its parent's visit */
struct Ud4Pre {
    int deg, idx0, idx1, t0, t1, cd0, cd1, b0, b1, chn, slot;
    bool hd, wd0, wd1, wm0, wm1;
};

template <bool DERIV, bool MARG>
__device__ static inline Ud4Pre ud4_pre(const Up4Args &a, int b, long sg)
{
    Ud4Pre q;
    const int start = as_uniform(a.indptr)[b];
    q.deg = as_uniform(a.indptr)[b + 1] - start;
    q.idx0 = start; q.idx1 = q.deg == 2 ? start + 1 : start;
    q.b0 = as_uniform(a.indices)[q.idx0]; q.b1 = as_uniform(a.indices)[q.idx1];
    q.t0 = as_uniform(a.edge_tip)[q.idx0]; q.t1 = as_uniform(a.edge_tip)[q.idx1];
    q.cd0 = a.codes[(size_t)q.b0 * a.Spad + sg];
    q.cd1 = a.codes[(size_t)q.b1 * a.Spad + sg];
    q.hd = as_uniform(a.node_has_data)[b] != 0;
    q.chn = q.hd ? a.codes[(size_t)b * a.Spad + sg] : 0;
    q.slot = as_uniform(a.node_scale)[b];
    q.wd0 = DERIV && (!a.edge_mask || as_uniform(a.edge_mask)[q.idx0]);
    q.wd1 = DERIV && q.deg == 2 && (!a.edge_mask || as_uniform(a.edge_mask)[q.idx1]);
    q.wm0 = MARG && (!a.node_mask || as_uniform(a.node_mask)[q.b0]);
    q.wm1 = MARG && q.deg == 2 && (!a.node_mask || as_uniform(a.node_mask)[q.b1]);
    return q;
}

/* one category of such a node: fb is its forward vector */
template <int NM>
__device__ static inline void ud4_pre_step(const Up4Args &a, int c, const double *tipc, const double *dtipc, size_t dstride,
                                           const PLK_AS4 double *Pm, double pc, size_t n, long slc, const v4 &fb,
                                           const Ud4Pre &q, double (&pd0)[NM], double (&pd1)[NM], v4 &pm0, v4 &pm1)
{
    v4 g = fb;
    if (q.hd) g = mul4(g, ld4(tipc + ((size_t)a.ntips * a.nchar + q.chn) * 4));
    if (q.slot >= 0) {
        const double sc = a.SC[((size_t)q.slot * a.C + c) * n + slc];
        g.a *= sc; g.b *= sc; g.c *= sc; g.d *= sc;
    }
    v4 fe0 = g, fe1 = g;
    if (q.deg == 2) {
        fe0 = mul4(g, ld4(tipc + ((size_t)q.t1 * a.nchar + q.cd1) * 4));
        fe1 = mul4(g, ld4(tipc + ((size_t)q.t0 * a.nchar + q.cd0) * 4));
    }
    if (q.wd0) {
#pragma unroll
        for (int m = 0; m < NM; m++) {
            const v4 y = ld4(dtipc + m * dstride + ((size_t)q.t0 * a.nchar + q.cd0) * 4);
            pd0[m] = fma(pc, fma(fe0.d, y.d, fma(fe0.c, y.c, fma(fe0.b, y.b, fe0.a * y.a))), pd0[m]);
        }
    }
    if (q.wd1) {
#pragma unroll
        for (int m = 0; m < NM; m++) {
            const v4 y = ld4(dtipc + m * dstride + ((size_t)q.t1 * a.nchar + q.cd1) * 4);
            pd1[m] = fma(pc, fma(fe1.d, y.d, fma(fe1.c, y.c, fma(fe1.b, y.b, fe1.a * y.a))), pd1[m]);
        }
    }
    if (q.wm0) {
        const v4 f = mtv4(Pm + ((size_t)c * a.E + q.idx0) * 16, fe0);
        const v4 lb = ld4(tipc + ((size_t)a.ntips * a.nchar + q.cd0) * 4);
        pm0.a = fma(pc * f.a, lb.a, pm0.a); pm0.b = fma(pc * f.b, lb.b, pm0.b);
        pm0.c = fma(pc * f.c, lb.c, pm0.c); pm0.d = fma(pc * f.d, lb.d, pm0.d);
    }
    if (q.wm1) {
        const v4 f = mtv4(Pm + ((size_t)c * a.E + q.idx1) * 16, fe1);
        const v4 lb = ld4(tipc + ((size_t)a.ntips * a.nchar + q.cd1) * 4);
        pm1.a = fma(pc * f.a, lb.a, pm1.a); pm1.b = fma(pc * f.b, lb.b, pm1.b);
        pm1.c = fma(pc * f.c, lb.c, pm1.c); pm1.d = fma(pc * f.d, lb.d, pm1.d);
    }
}


/* marginal of one node at this lane's site: a plane entry per state, or -- site sums only -- the wave's weighted sum.
 * Called by all lanes of the wave (uniform control flow); lanes without a site pass valid = false. */
__device__ static inline void ud4_out_m(const Up4Args &a, size_t n, long sl, bool valid, int node, const v4 &m, double inv)
{
    if (a.MVS) {
        const double ws = valid ? (a.wsite ? a.wsite[sl] : 1.0) * inv : 0.0;
        const double s0 = wave64_sum_lane63(m.a * ws), s1 = wave64_sum_lane63(m.b * ws);
        const double s2 = wave64_sum_lane63(m.c * ws), s3 = wave64_sum_lane63(m.d * ws);
        if ((threadIdx.x & 63) == 63) {
            const size_t nw = (size_t)gridDim.x * (UD4_BLOCK / 64), wv = (size_t)blockIdx.x * (UD4_BLOCK / 64) + (threadIdx.x >> 6);
            double *p = a.MVS + (size_t)node * 4 * nw + wv;
            p[0] = s0; p[nw] = s1; p[2 * nw] = s2; p[3 * nw] = s3;
        }
    } else if (valid) {
        double *mv = a.MV + (size_t)node * 4 * n + sl;
        mv[0] = m.a * inv; mv[n] = m.b * inv; mv[2 * n] = m.c * inv; mv[3 * n] = m.d * inv;
    }
}

template <int NM>
__device__ static inline void ud4_pre_write(const Up4Args &a, size_t n, long sl, double inv, const Ud4Pre &q,
                                            const double (&pd0)[NM], const double (&pd1)[NM], const v4 &pm0, const v4 &pm1)
{
#pragma unroll
    for (int m = 0; m < NM; m++) {
        if (q.wd0) a.DV[((size_t)m * a.E + q.idx0) * n + sl] = pd0[m] * inv;
        if (q.wd1) a.DV[((size_t)m * a.E + q.idx1) * n + sl] = pd1[m] * inv;
    }
    (void)pm0; (void)pm1;          /* the marginals of the two leaves go through ud4_out_m (all lanes of the wave) */
}

template <bool DERIV, bool MARG, int NM = 1>
__global__ __launch_bounds__(UD4_BLOCK) void k_up4(Up4Args a)
{
    const long sl = (long)blockIdx.x * UD4_BLOCK + threadIdx.x;
    const bool valid = sl < a.n;
    const long slc = valid ? sl : a.n - 1;
    const long sg = a.s0 + slc;
    const size_t n = (size_t)a.n;
    const PLK_AS4 int *pre = as_uniform(a.preorder), *ip = as_uniform(a.indptr), *ix = as_uniform(a.indices);
    const PLK_AS4 int *has = as_uniform(a.node_has_data), *etip = as_uniform(a.edge_tip);
    const PLK_AS4 int *nint = as_uniform(a.node_int), *nsc = as_uniform(a.node_scale);
    const PLK_AS4 double *Pm = as_uniform(a.P), *Mm = as_uniform(a.dP);
    const PLK_AS4 double *prior = as_uniform(a.cat_prior), *rw = as_uniform(a.root_w);
    const size_t tabc = (size_t)(a.ntips + 1) * a.nchar * 4;
    const size_t mstride = (size_t)a.C * a.E * 16, dstride = (size_t)a.C * tabc;   /* between edge-form sets */
    const double inv = 1.0 / a.LH[slc];
    const int root = pre[0];
    const v4 w = v4{rw[0], rw[1], rw[2], rw[3]};
    const v4 zero = v4{0.0, 0.0, 0.0, 0.0};

    {   /* root: forward vector = root prior weights; its marginal */
        v4 macc = zero;
        for (int c = 0; c < a.C; c++) {
            if (valid) st4(a.FN + (((size_t)nint[root] * a.C + c) * n + slc) * 4, w);
            if (MARG) {
                /* L_root carries every rescaling; F_root none */
                const v4 l = ld4(a.LN + (((size_t)nint[root] * a.C + c) * n + slc) * 4);
                const double pc = prior[c] * a.CW[(size_t)c * n + slc];
                macc.a = fma(pc * w.a, l.a, macc.a); macc.b = fma(pc * w.b, l.b, macc.b);
                macc.c = fma(pc * w.c, l.c, macc.c); macc.d = fma(pc * w.d, l.d, macc.d);
            }
        }
        if (MARG && (!a.node_mask || as_uniform(a.node_mask)[root])) ud4_out_m(a, n, sl, valid, root, macc, inv);
    }

    for (int u = 0; u < a.N; u++) {
        const int nd = pre[u];
        const int start = ip[nd], stop = ip[nd + 1];
        const int deg = stop - start;
        if (deg == 0 || as_uniform(a.node_inline)[nd]) continue;
        const bool hd = has[nd] != 0;
        const int chn = hd ? a.codes[(size_t)nd * a.Spad + sg] : 0;
        const int slot = nsc[nd];
        const double *fn_nd = a.FN + ((size_t)nint[nd] * a.C) * n * 4;

        if (deg <= 2) {
            /* both children together: F_a, L_b0, L_b1 are read once, nothing is recomputed */
            Ud4Child c0 = ud4_child(a, start, sg), c1 = ud4_child(a, deg == 2 ? start + 1 : start, sg);
            c0.want_d = DERIV && (!a.edge_mask || as_uniform(a.edge_mask)[c0.idx]);
            c1.want_d = DERIV && deg == 2 && (!a.edge_mask || as_uniform(a.edge_mask)[c1.idx]);
            c0.want_m = MARG && (!a.node_mask || as_uniform(a.node_mask)[c0.b]);
            c1.want_m = MARG && deg == 2 && (!a.node_mask || as_uniform(a.node_mask)[c1.b]);
            c0.want_f = c0.t < 0 || c0.want_m;
            c1.want_f = deg == 2 && (c1.t < 0 || c1.want_m);
            double d0[NM], d1[NM];
#pragma unroll
            for (int m = 0; m < NM; m++) d0[m] = d1[m] = 0.0;
            v4 m0 = zero, m1 = zero;
            const bool in0 = c0.t < 0 && as_uniform(a.node_inline)[c0.b];
            const bool in1 = deg == 2 && c1.t < 0 && as_uniform(a.node_inline)[c1.b];
            Ud4Pre q0 = {}, q1 = {};
            if (in0) q0 = ud4_pre<DERIV, MARG>(a, c0.b, sg);
            if (in1) q1 = ud4_pre<DERIV, MARG>(a, c1.b, sg);
            double p00[NM], p01[NM], p10[NM], p11[NM];
#pragma unroll
            for (int m = 0; m < NM; m++) p00[m] = p01[m] = p10[m] = p11[m] = 0.0;
            v4 r00 = zero, r01 = zero, r10 = zero, r11 = zero;
            for (int c = 0; c < a.C; c++) {
                const double *tipc = a.tip + (size_t)c * tabc;
                const double *dtipc = a.dtip + (size_t)c * tabc;
                v4 g = ld4(fn_nd + ((size_t)c * n + slc) * 4);
                if (hd) g = mul4(g, ld4(tipc + ((size_t)a.ntips * a.nchar + chn) * 4));
                if (slot >= 0) {
                    const double sc = a.SC[((size_t)slot * a.C + c) * n + slc];
                    g.a *= sc; g.b *= sc; g.c *= sc; g.d *= sc;
                }
                const double pc = prior[c] * a.CW[(size_t)c * n + slc];
                v4 x0 = zero, x1 = zero;
                if (c0.t < 0) x0 = ld4(a.LN + (((size_t)nint[c0.b] * a.C + c) * n + slc) * 4);
                if (deg == 2 && c1.t < 0) x1 = ld4(a.LN + (((size_t)nint[c1.b] * a.C + c) * n + slc) * 4);
                v4 fe0 = g, fe1 = g;
                if (deg == 2) {
                    fe0 = mul4(g, ud4_child_msg(a, c, c1.idx, c1.t, c1.code, tipc, Pm, x1));
                    fe1 = mul4(g, ud4_child_msg(a, c, c0.idx, c0.t, c0.code, tipc, Pm, x0));
                }
#define UD4_EDGE(CH, FE, X, DS, MS, INL, Q, PA, PB, RA, RB)                                                 \
                if (CH.want_d) {                                                                            \
                    _Pragma("unroll") for (int m = 0; m < NM; m++) {                                        \
                        v4 y;                                                                               \
                        if (CH.t >= 0) y = ld4(dtipc + m * dstride + ((size_t)CH.t * a.nchar + CH.code) * 4); \
                        else {                                                                              \
                            y = mv4(Mm + m * mstride + ((size_t)c * a.E + CH.idx) * 16, X);                 \
                            if (a.dzero && const4(X)) y = zero;  /* rows of dP sum to zero (src/util.c:338-345) */ \
                        }                                                                                   \
                        DS[m] = fma(pc, fma(FE.d, y.d, fma(FE.c, y.c, fma(FE.b, y.b, FE.a * y.a))), DS[m]); \
                    }                                                                                       \
                }                                                                                           \
                if (CH.want_f) {                                                                            \
                    const v4 fb = mtv4(Pm + ((size_t)c * a.E + CH.idx) * 16, FE);                           \
                    if (INL) ud4_pre_step<NM>(a, c, tipc, dtipc, dstride, Pm, pc, n, slc, fb, Q, PA, PB, RA, RB); \
                    else if (CH.t < 0 && valid) st4(a.FN + (((size_t)nint[CH.b] * a.C + c) * n + slc) * 4, fb);  \
                    if (CH.want_m) {                                                                        \
                        const v4 lb = CH.t >= 0 ? ld4(tipc + ((size_t)a.ntips * a.nchar + CH.code) * 4) : X; \
                        MS.a = fma(pc * fb.a, lb.a, MS.a); MS.b = fma(pc * fb.b, lb.b, MS.b);               \
                        MS.c = fma(pc * fb.c, lb.c, MS.c); MS.d = fma(pc * fb.d, lb.d, MS.d);               \
                    }                                                                                       \
                }
                UD4_EDGE(c0, fe0, x0, d0, m0, in0, q0, p00, p01, r00, r01)
                if (deg == 2) { UD4_EDGE(c1, fe1, x1, d1, m1, in1, q1, p10, p11, r10, r11) }
#undef UD4_EDGE
            }
            if (valid) {
                if (in0) ud4_pre_write<NM>(a, n, sl, inv, q0, p00, p01, r00, r01);
                if (in1) ud4_pre_write<NM>(a, n, sl, inv, q1, p10, p11, r10, r11);
#pragma unroll
                for (int m = 0; m < NM; m++) {
                    if (c0.want_d) a.DV[((size_t)m * a.E + c0.idx) * n + sl] = d0[m] * inv;
                    if (c1.want_d) a.DV[((size_t)m * a.E + c1.idx) * n + sl] = d1[m] * inv;
                }
            }
            if (MARG) {
                if (in0 && q0.wm0) ud4_out_m(a, n, sl, valid, q0.b0, r00, inv);
                if (in0 && q0.wm1) ud4_out_m(a, n, sl, valid, q0.b1, r01, inv);
                if (in1 && q1.wm0) ud4_out_m(a, n, sl, valid, q1.b0, r10, inv);
                if (in1 && q1.wm1) ud4_out_m(a, n, sl, valid, q1.b1, r11, inv);
                if (c0.want_m) ud4_out_m(a, n, sl, valid, c0.b, m0, inv);
                if (c1.want_m) ud4_out_m(a, n, sl, valid, c1.b, m1, inv);
            }
            continue;
        }

        /* three or more children (e.g. an unrooted tree's root): one edge at a time, sibling messages recomputed */
        for (int idx = start; idx < stop; idx++) {
            Ud4Child ch = ud4_child(a, idx, sg);
            ch.want_d = DERIV && (!a.edge_mask || as_uniform(a.edge_mask)[idx]);
            ch.want_m = MARG && (!a.node_mask || as_uniform(a.node_mask)[ch.b]);
            ch.want_f = ch.t < 0 || ch.want_m;
            if (!ch.want_d && !ch.want_f) continue;
            double dsum[NM];
#pragma unroll
            for (int m = 0; m < NM; m++) dsum[m] = 0.0;
            v4 macc = zero;
            for (int c = 0; c < a.C; c++) {
                const double *tipc = a.tip + (size_t)c * tabc;
                const double *dtipc = a.dtip + (size_t)c * tabc;
                v4 fe = ld4(fn_nd + ((size_t)c * n + slc) * 4);
                if (hd) fe = mul4(fe, ld4(tipc + ((size_t)a.ntips * a.nchar + chn) * 4));
                if (slot >= 0) {
                    const double sc = a.SC[((size_t)slot * a.C + c) * n + slc];
                    fe.a *= sc; fe.b *= sc; fe.c *= sc; fe.d *= sc;
                }
                for (int idx2 = start; idx2 < stop; idx2++) {
                    if (idx2 == idx) continue;
                    const int t2 = etip[idx2], b2 = ix[idx2];
                    v4 x2 = zero;
                    int code2 = 0;
                    if (t2 >= 0) code2 = a.codes[(size_t)b2 * a.Spad + sg];
                    else x2 = ld4(a.LN + (((size_t)nint[b2] * a.C + c) * n + slc) * 4);
                    fe = mul4(fe, ud4_child_msg(a, c, idx2, t2, code2, tipc, Pm, x2));
                }
                const double pc = prior[c] * a.CW[(size_t)c * n + slc];
                v4 x = zero;
                if (ch.t < 0) x = ld4(a.LN + (((size_t)nint[ch.b] * a.C + c) * n + slc) * 4);
                if (ch.want_d) {
#pragma unroll
                    for (int m = 0; m < NM; m++) {
                        v4 y;
                        if (ch.t >= 0) y = ld4(dtipc + m * dstride + ((size_t)ch.t * a.nchar + ch.code) * 4);
                        else {
                            y = mv4(Mm + m * mstride + ((size_t)c * a.E + idx) * 16, x);
                            if (a.dzero && const4(x)) y = zero;
                        }
                        dsum[m] = fma(pc, fma(fe.d, y.d, fma(fe.c, y.c, fma(fe.b, y.b, fe.a * y.a))), dsum[m]);
                    }
                }
                if (ch.want_f) {
                    const v4 fb = mtv4(Pm + ((size_t)c * a.E + idx) * 16, fe);
                    if (ch.t < 0 && valid) st4(a.FN + (((size_t)nint[ch.b] * a.C + c) * n + slc) * 4, fb);
                    if (ch.want_m) {
                        const v4 lb = ch.t >= 0 ? ld4(tipc + ((size_t)a.ntips * a.nchar + ch.code) * 4) : x;
                        macc.a = fma(pc * fb.a, lb.a, macc.a); macc.b = fma(pc * fb.b, lb.b, macc.b);
                        macc.c = fma(pc * fb.c, lb.c, macc.c); macc.d = fma(pc * fb.d, lb.d, macc.d);
                    }
                }
            }
            if (ch.want_d && valid) {
#pragma unroll
                for (int m = 0; m < NM; m++) a.DV[((size_t)m * a.E + idx) * n + sl] = dsum[m] * inv;
            }
            if (ch.want_m) ud4_out_m(a, n, sl, valid, ch.b, macc, inv);
        }
    }
}

/*
 * Up pass of the derivative query by node visits, k = 4 (records: plk_up_nodes_build() of plk_program.h; the scheme of
 * k_up_nodes_mfma).  k_up4 stores the forward vector of every internal node and reads it back in the node's own visit:
 * 47.7 GB read + 20.1 GB written per 2M sites at BASELINE config 3, at 6 TB/s -- the HBM roof.  Here what is stored per
 * internal node is G_a, the vector at the top of the edge into a, visits run depth first, and the G of the last internal
 * child with work stays in registers for the next visit (all C categories of it: 8 C registers, C <= CM), so a G is
 * written and read only where the tree forks into two internal subtrees.  A visit reads L_b once per internal child,
 * recomputes the child messages (16 FMAs), finishes the derivative of a's own edge as (M_a^T G_a) . (s_a B_a o messages)
 * -- L_a itself is never read, and the constant-vector rule of src/util.c:338-345 is applied to the recomputed L_a
 * exactly --, forms F_a = P_a^T G_a and the G of every child; leaf edges are finished with the edge-form tip tables.
 * Categories run inside the visit, so every edge's derivative is written once (no read-modify-write of DV).
 */
/* "SGPR base + 32-bit lane offset" forms of ld4 / st4 and of scalar element access: the (wave-uniform) base is pinned in an
 * SGPR pair, so that no 64-bit lane address per array is kept across the visit loop (k_up4_nodes: 40 of them otherwise) */
__device__ __forceinline__ v4 ld4u(const double *base, unsigned off)
{
    asm volatile("" : "+s"(base));
    const double2 *p = reinterpret_cast<const double2 *>(base);
    const double2 lo = p[off >> 1], hi = p[(off >> 1) + 1];
    return v4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ void st4u(double *base, unsigned off, const v4 &v)
{
    asm volatile("" : "+s"(base));
    double2 *p = reinterpret_cast<double2 *>(base);
    p[off >> 1] = double2{v.a, v.b};
    p[(off >> 1) + 1] = double2{v.c, v.d};
}
template <class V>
__device__ __forceinline__ V at_u(const V *base, unsigned off)
{
    asm volatile("" : "+s"(base));
    return base[off];
}

/* message of a two-leaf node b towards its parent, P_b (P_b0 B_b0 o P_b1 B_b1), as a row of its pair table: the
 * reference forms it by two _prune_update_prob calls and one product per site (src/evaluate_site_lhood.c:36-56); the
 * table is built once per (category, node) in double-double.  Neither pass moves L_b through HBM for such a node. */
__device__ static inline v4 u4n_pair_message(const Up4Args &a, int c, int pair, int b, unsigned us)
{
    const int e0 = as_uniform(a.indptr)[b];
    const int b0 = as_uniform(a.indices)[e0], b1 = as_uniform(a.indices)[e0 + 1];
    const unsigned comb = (unsigned)at_u(a.codes + (size_t)b0 * a.Spad + a.s0, us) * (unsigned)a.nchar + (unsigned)at_u(a.codes + (size_t)b1 * a.Spad + a.s0, us);
    return ld4u(a.ptab + ((size_t)c * a.npairs + pair) * a.nchar * a.nchar * 4, 4u * comb);
}

/* L_b of internal child b (record flags fl, storage index bi): read from LN, or -- PLK_UN_REBUILD: both of b's children are
 * leaves or pair nodes -- the product of their two messages, each one row of a tip table or of a pair table (64 bytes of
 * L2-resident tables instead of 32 bytes written by the down pass and read here, per category) */
__device__ static inline v4 u4n_child_L(const Up4Args &a, int c, int fl, int bi, size_t tabc, unsigned us, unsigned us4, size_t n)
{
    if (!(fl & PLK_UN_REBUILD)) return ld4u(a.LN + ((size_t)bi * a.C + c) * n * 4, us4);
    const PLK_AS4 int *rt = as_uniform(a.rebuild) + 4 * bi;
    v4 m[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int t = rt[2 * q], nd = rt[2 * q + 1];
        if (t >= 0) m[q] = ld4u(a.tip + (size_t)c * tabc + (size_t)t * a.nchar * 4, 4u * at_u(a.codes + (size_t)nd * a.Spad + a.s0, us));
        else m[q] = u4n_pair_message(a, c, -2 - t, nd, us);
    }
    return v4{m[0].a * m[1].a, m[0].b * m[1].b, m[0].c * m[1].c, m[0].d * m[1].d};
}

/* The table rows a child's message needs depend on the site's pattern codes only, not on the category: they are looked up
 * once per visit (row offsets, in doubles) and used by every category.
 *   leaf: r0 = 4 x code;  pair child: r0 = 4 x combined code;  rebuilt child: r0, r1 = the rows of its two items;  stored child: unused */
struct U4nRows { unsigned r0, r1; };
__device__ static inline unsigned u4n_pair_row(const Up4Args &a, int b, unsigned us)
{
    const int e0 = as_uniform(a.indptr)[b];
    const int b0 = as_uniform(a.indices)[e0], b1 = as_uniform(a.indices)[e0 + 1];
    return 4u * ((unsigned)at_u(a.codes + (size_t)b0 * a.Spad + a.s0, us) * (unsigned)a.nchar + (unsigned)at_u(a.codes + (size_t)b1 * a.Spad + a.s0, us));
}
__device__ static inline U4nRows u4n_child_rows(const Up4Args &a, int b, int t, int fl, int bi, unsigned us)
{
    U4nRows r = {0u, 0u};
    if (t >= 0) r.r0 = 4u * at_u(a.codes + (size_t)b * a.Spad + a.s0, us);
    else if (t < -1) r.r0 = u4n_pair_row(a, b, us);
    else if (fl & PLK_UN_REBUILD) {
        const PLK_AS4 int *rt = as_uniform(a.rebuild) + 4 * bi;
        r.r0 = rt[0] >= 0 ? 4u * at_u(a.codes + (size_t)rt[1] * a.Spad + a.s0, us) : u4n_pair_row(a, rt[1], us);
        r.r1 = rt[2] >= 0 ? 4u * at_u(a.codes + (size_t)rt[3] * a.Spad + a.s0, us) : u4n_pair_row(a, rt[3], us);
    }
    return r;
}
/* one item of a rebuilt vector / a pair child's message for category c: tip-table row (t >= 0) or pair-table row (t <= -2) */
__device__ static inline v4 u4n_row(const Up4Args &a, int c, int t, unsigned row, size_t tabc)
{
    if (t >= 0) return ld4u(a.tip + (size_t)c * tabc + (size_t)t * a.nchar * 4, row);
    return ld4u(a.ptab + ((size_t)c * a.npairs + (-2 - t)) * a.nchar * a.nchar * 4, row);
}
/* L_b of an internal child that is not a pair node: stored, or the product of its two items' rows */
__device__ static inline v4 u4n_child_L_rows(const Up4Args &a, int c, int fl, int bi, const U4nRows &rows, size_t tabc, unsigned us4, size_t n)
{
    if (!(fl & PLK_UN_REBUILD)) return ld4u(a.LN + ((size_t)bi * a.C + c) * n * 4, us4);
    const PLK_AS4 int *rt = as_uniform(a.rebuild) + 4 * bi;
    const v4 m0 = u4n_row(a, c, rt[0], rows.r0, tabc), m1 = u4n_row(a, c, rt[2], rows.r1, tabc);
    return v4{m0.a * m1.a, m0.b * m1.b, m0.c * m1.c, m0.d * m1.d};
}

template <int CM>
__global__ __launch_bounds__(UD4_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_up4_nodes(Up4Args a)
{
    const long sl = (long)blockIdx.x * UD4_BLOCK + threadIdx.x;
    const bool valid = sl < a.n;
    const long slc = valid ? sl : a.n - 1;
    const unsigned us = (unsigned)slc, us4 = us * 4u;          /* lane offsets: site, and site x 4 doubles */
    const size_t n = (size_t)a.n;
    const PLK_AS4 int *vis = as_uniform(a.visits);
    const PLK_AS4 double *Pm = as_uniform(a.P), *Mm = as_uniform(a.dP);
    const PLK_AS4 double *prior = as_uniform(a.cat_prior), *rw = as_uniform(a.root_w);
    const size_t tabc = (size_t)(a.ntips + 1) * a.nchar * 4;
    const double inv = 1.0 / at_u(a.LH, us);
    const v4 one = v4{1.0, 1.0, 1.0, 1.0};
    double pcw[CM];                           /* prior_c x 2^(X_c - Xmax) */
#pragma unroll
    for (int c = 0; c < CM; c++) pcw[c] = c < a.C ? prior[c] * at_u(a.CW + (size_t)c * n, us) : 0.0;
    v4 gc[CM];                                /* G handed from visit to visit in registers, per category (consumed, then re-formed in place) */
#pragma unroll
    for (int c = 0; c < CM; c++) gc[c] = v4{rw[0], rw[1], rw[2], rw[3]};     /* the root's visit comes first */

    int vp = 0;
    for (int v = 0; v < a.nvisits; v++) {
        const int nd = vis[vp], deg = vis[vp + 1], nd_int = vis[vp + 2], slot = vis[vp + 3], hd = vis[vp + 4], e0 = vis[vp + 5];
        const int ea = vis[vp + 6], hfl = vis[vp + 7];
        const PLK_AS4 int *ch = vis + vp + 8;
        vp += 8 + 4 * deg;
        const int code_nd = hd ? at_u(a.codes + (size_t)nd * a.Spad + a.s0, us) : 0;
        /* message of the child in record J for category C_: tip-table row, or P_b L_b (L_b constant: the constant) */
#define U4N_MESSAGE(J, C_, M)                                                                             \
        do { const int b_ = ch[4 * (J)], t_ = ch[4 * (J) + 1], pos_ = ch[4 * (J) + 2] >> PLK_UN_POS_SHIFT; \
             if (t_ >= 0) M = ld4u(a.tip + (size_t)(C_) * tabc + (size_t)t_ * a.nchar * 4, 4u * at_u(a.codes + (size_t)b_ * a.Spad + a.s0, us)); \
             else if (t_ < -1) M = u4n_pair_message(a, C_, -2 - t_, b_, us);                              \
             else { const v4 x_ = u4n_child_L(a, C_, ch[4 * (J) + 2], ch[4 * (J) + 3], tabc, us, us4, n);   \
                    M = mv4(Pm + ((size_t)(C_) * a.E + e0 + pos_) * 16, x_);                              \
                    if (const4(x_)) M = x_; } } while (0)
        if (deg <= 2) {
            double d_own = 0.0, d0 = 0.0, d1 = 0.0;
            const int fl0 = ch[2], fl1 = deg == 2 ? ch[6] : 0;
            /* table rows of the children (pattern codes of leaves, combined codes of pair nodes, the items of rebuilt
             * vectors), once per visit */
            const U4nRows rw0 = u4n_child_rows(a, ch[0], ch[1], fl0, ch[3], us);
            const U4nRows rw1 = deg == 2 ? u4n_child_rows(a, ch[4], ch[5], fl1, ch[7], us) : U4nRows{0u, 0u};
            const unsigned cd0 = rw0.r0, cd1 = rw1.r0;
            /* pair children finished inside this visit (PLK_UN_INL_*): CSR edge of their first leaf, tip slots and code rows of
             * their two leaves, derivative accumulators (edge into the child, first leaf edge, second leaf edge) */
            /* (one-category instantiation only: with four categories in flight the visit has no registers left for it -- measured
             * slower --, and the host asks for inline pair children only when C = 1) */
            const bool in0 = CM == 1 && (fl0 & PLK_UN_INL) != 0, in1 = CM == 1 && (fl1 & PLK_UN_INL) != 0;
            const int le0 = in0 ? as_uniform(a.indptr)[ch[0]] : 0, le1 = in1 ? as_uniform(a.indptr)[ch[4]] : 0;
            const int ts00 = in0 ? as_uniform(a.edge_tip)[le0] : 0, ts01 = in0 ? as_uniform(a.edge_tip)[le0 + 1] : 0;
            const int ts10 = in1 ? as_uniform(a.edge_tip)[le1] : 0, ts11 = in1 ? as_uniform(a.edge_tip)[le1 + 1] : 0;
            const unsigned cr00 = in0 ? 4u * at_u(a.codes + (size_t)as_uniform(a.indices)[le0] * a.Spad + a.s0, us) : 0u;
            const unsigned cr01 = in0 ? 4u * at_u(a.codes + (size_t)as_uniform(a.indices)[le0 + 1] * a.Spad + a.s0, us) : 0u;
            const unsigned cr10 = in1 ? 4u * at_u(a.codes + (size_t)as_uniform(a.indices)[le1] * a.Spad + a.s0, us) : 0u;
            const unsigned cr11 = in1 ? 4u * at_u(a.codes + (size_t)as_uniform(a.indices)[le1 + 1] * a.Spad + a.s0, us) : 0u;
            double di0[3] = {0.0, 0.0, 0.0}, di1[3] = {0.0, 0.0, 0.0};
            /* what the pair child's own visit would compute from its G (here G_) for category c: both its children are leaves,
             * it has no data and no rescaling (the same operations in the same order: the same bits) */
#define U4N_INLINE_PAIR(FL_, G_, EDGE_, TS0_, TS1_, CR0_, CR1_, LE_, D_)                                            \
            do { const v4 t0_ = ld4u(tipc + (size_t)(TS0_) * a.nchar * 4, CR0_), t1_ = ld4u(tipc + (size_t)(TS1_) * a.nchar * 4, CR1_); \
                 if ((FL_) & PLK_UN_INL_OWN) {                                                                      \
                     const v4 z_ = mtv4(Mm + ((size_t)c * a.E + (EDGE_)) * 16, G_);                                 \
                     const v4 l_ = mul4(mul4(t0_, t1_), one);                                                        \
                     const double d_ = (a.dzero && const4(l_)) ? 0.0 : fma(z_.d, l_.d, fma(z_.c, l_.c, fma(z_.b, l_.b, z_.a * l_.a))); \
                     D_[0] = fma(pcw[c] * 1.0, d_, D_[0]); }                                                         \
                 v4 f_ = mtv4(Pm + ((size_t)c * a.E + (EDGE_)) * 16, G_);                                           \
                 f_ = mul4(f_, one);                                                                                 \
                 f_.a *= 1.0; f_.b *= 1.0; f_.c *= 1.0; f_.d *= 1.0;                                                 \
                 if ((FL_) & PLK_UN_INL_L0) { const v4 y_ = ld4u(dtipc + (size_t)(TS0_) * a.nchar * 4, CR0_); const v4 q_ = mul4(f_, t1_); \
                     D_[1] = fma(pcw[c], fma(q_.d, y_.d, fma(q_.c, y_.c, fma(q_.b, y_.b, q_.a * y_.a))), D_[1]); }   \
                 if ((FL_) & PLK_UN_INL_L1) { const v4 y_ = ld4u(dtipc + (size_t)(TS1_) * a.nchar * 4, CR1_); const v4 q_ = mul4(f_, t0_); \
                     D_[2] = fma(pcw[c], fma(q_.d, y_.d, fma(q_.c, y_.c, fma(q_.b, y_.b, q_.a * y_.a))), D_[2]); } } while (0)
#pragma unroll
            for (int c = 0; c < CM; c++) {
                if (c >= a.C) break;
                const double *tipc = a.tip + (size_t)c * tabc, *dtipc = a.dtip + (size_t)c * tabc;
                v4 g = gc[c];
                if (!(hfl & PLK_UN_FROM_REGS)) g = ld4u(a.FN + ((size_t)nd_int * a.C + c) * n * 4, us4);
                const double sc = slot >= 0 ? at_u(a.SC + ((size_t)slot * a.C + c) * n, us) : 1.0;
                v4 m0, m1 = one, ob = one;
                if (ch[1] >= 0) m0 = ld4u(tipc + (size_t)ch[1] * a.nchar * 4, cd0);
                else if (ch[1] < -1) m0 = u4n_row(a, c, ch[1], rw0.r0, tabc);
                else {
                    const v4 x = u4n_child_L_rows(a, c, fl0, ch[3], rw0, tabc, us4, n);
                    m0 = mv4(Pm + ((size_t)c * a.E + e0 + (fl0 >> PLK_UN_POS_SHIFT)) * 16, x);
                    if (const4(x)) m0 = x;
                }
                if (deg == 2) {
                    if (ch[5] >= 0) m1 = ld4u(tipc + (size_t)ch[5] * a.nchar * 4, cd1);
                    else if (ch[5] < -1) m1 = u4n_row(a, c, ch[5], rw1.r0, tabc);
                    else {
                        const v4 x = u4n_child_L_rows(a, c, fl1, ch[7], rw1, tabc, us4, n);
                        m1 = mv4(Pm + ((size_t)c * a.E + e0 + (fl1 >> PLK_UN_POS_SHIFT)) * 16, x);
                        if (const4(x)) m1 = x;
                    }
                }
                if (hd) ob = ld4u(tipc + (size_t)a.ntips * a.nchar * 4, 4u * (unsigned)code_nd);
                if (hfl & PLK_UN_OWN_D) {
                    const v4 z = mtv4(Mm + ((size_t)c * a.E + ea) * 16, g);
                    const v4 l = mul4(mul4(m0, m1), ob);                 /* L_a / s_a */
                    const double d = (a.dzero && const4(l)) ? 0.0 : fma(z.d, l.d, fma(z.c, l.c, fma(z.b, l.b, z.a * l.a)));
                    d_own = fma(pcw[c] * sc, d, d_own);
                }
                v4 fe = ea >= 0 ? mtv4(Pm + ((size_t)c * a.E + ea) * 16, g) : g;
                fe = mul4(fe, ob);
                fe.a *= sc; fe.b *= sc; fe.c *= sc; fe.d *= sc;
                const v4 g0 = mul4(fe, m1), g1 = mul4(fe, m0);        /* child 0 sees child 1's message and vice versa */
                if (fl0 & PLK_UN_LEAF_D) {
                    const v4 y = ld4u(dtipc + (size_t)ch[1] * a.nchar * 4, cd0);
                    d0 = fma(pcw[c], fma(g0.d, y.d, fma(g0.c, y.c, fma(g0.b, y.b, g0.a * y.a))), d0);
                } else if ((fl0 & PLK_UN_STORE_G) && valid) st4u(a.FN + ((size_t)ch[3] * a.C + c) * n * 4, us4, g0);
                if (fl1 & PLK_UN_LEAF_D) {
                    const v4 y = ld4u(dtipc + (size_t)ch[5] * a.nchar * 4, cd1);
                    d1 = fma(pcw[c], fma(g1.d, y.d, fma(g1.c, y.c, fma(g1.b, y.b, g1.a * y.a))), d1);
                } else if ((fl1 & PLK_UN_STORE_G) && valid) st4u(a.FN + ((size_t)ch[7] * a.C + c) * n * 4, us4, g1);
                if (CM == 1 && in0) U4N_INLINE_PAIR(fl0, g0, e0 + (fl0 >> PLK_UN_POS_SHIFT), ts00, ts01, cr00, cr01, le0, di0);
                if (CM == 1 && in1) U4N_INLINE_PAIR(fl1, g1, e0 + (fl1 >> PLK_UN_POS_SHIFT), ts10, ts11, cr10, cr11, le1, di1);
                /* the continued child is the last record: record 1 of two, record 0 of one */
                gc[c] = deg == 2 ? g1 : g0;
            }
            if (valid) {
                if (hfl & PLK_UN_OWN_D) { double *dp = a.DV + (size_t)ea * n; asm volatile("" : "+s"(dp)); dp[us] = d_own * inv; }
                if (fl0 & PLK_UN_LEAF_D) { double *dp = a.DV + (size_t)(e0 + (fl0 >> PLK_UN_POS_SHIFT)) * n; asm volatile("" : "+s"(dp)); dp[us] = d0 * inv; }
                if (fl1 & PLK_UN_LEAF_D) { double *dp = a.DV + (size_t)(e0 + (fl1 >> PLK_UN_POS_SHIFT)) * n; asm volatile("" : "+s"(dp)); dp[us] = d1 * inv; }
                if (CM == 1 && (fl0 & PLK_UN_INL_OWN)) { double *dp = a.DV + (size_t)(e0 + (fl0 >> PLK_UN_POS_SHIFT)) * n; asm volatile("" : "+s"(dp)); dp[us] = di0[0] * inv; }
                if (CM == 1 && (fl0 & PLK_UN_INL_L0)) { double *dp = a.DV + (size_t)le0 * n; asm volatile("" : "+s"(dp)); dp[us] = di0[1] * inv; }
                if (CM == 1 && (fl0 & PLK_UN_INL_L1)) { double *dp = a.DV + (size_t)(le0 + 1) * n; asm volatile("" : "+s"(dp)); dp[us] = di0[2] * inv; }
                if (CM == 1 && (fl1 & PLK_UN_INL_OWN)) { double *dp = a.DV + (size_t)(e0 + (fl1 >> PLK_UN_POS_SHIFT)) * n; asm volatile("" : "+s"(dp)); dp[us] = di1[0] * inv; }
                if (CM == 1 && (fl1 & PLK_UN_INL_L0)) { double *dp = a.DV + (size_t)le1 * n; asm volatile("" : "+s"(dp)); dp[us] = di1[1] * inv; }
                if (CM == 1 && (fl1 & PLK_UN_INL_L1)) { double *dp = a.DV + (size_t)(le1 + 1) * n; asm volatile("" : "+s"(dp)); dp[us] = di1[2] * inv; }
            }
#undef U4N_INLINE_PAIR
        } else {
            /* more than two children: messages recomputed per child (as the one-edge-at-a-time pass does); one child at
             * a time over all categories, so that its derivative is still written once */
            double d_own = 0.0;
            v4 fe[CM];
#pragma unroll
            for (int c = 0; c < CM; c++) fe[c] = one;        /* defined for every c, also past a.C (one definition dominating the uses) */
#pragma unroll
            for (int c = 0; c < CM; c++) {
                if (c >= a.C) break;
                const double *tipc = a.tip + (size_t)c * tabc;
                v4 g = gc[c];
                if (!(hfl & PLK_UN_FROM_REGS)) g = ld4u(a.FN + ((size_t)nd_int * a.C + c) * n * 4, us4);
                const double sc = slot >= 0 ? at_u(a.SC + ((size_t)slot * a.C + c) * n, us) : 1.0;
                const v4 ob = hd ? ld4u(tipc + (size_t)a.ntips * a.nchar * 4, 4u * (unsigned)code_nd) : one;
                if (hfl & PLK_UN_OWN_D) {
                    const v4 z = mtv4(Mm + ((size_t)c * a.E + ea) * 16, g);
                    v4 l = ob;
                    for (int j = 0; j < deg; j++) { v4 m; U4N_MESSAGE(j, c, m); l = mul4(l, m); }
                    const double d = (a.dzero && const4(l)) ? 0.0 : fma(z.d, l.d, fma(z.c, l.c, fma(z.b, l.b, z.a * l.a)));
                    d_own = fma(pcw[c] * sc, d, d_own);
                }
                v4 f = ea >= 0 ? mtv4(Pm + ((size_t)c * a.E + ea) * 16, g) : g;
                f = mul4(f, ob);
                f.a *= sc; f.b *= sc; f.c *= sc; f.d *= sc;
                fe[c] = f;
            }
            if (valid && (hfl & PLK_UN_OWN_D)) { double *dp = a.DV + (size_t)ea * n; asm volatile("" : "+s"(dp)); dp[us] = d_own * inv; }
            for (int j = 0; j < deg; j++) {
                const int fl = ch[4 * j + 2];
                if (!(fl & PLK_UN_WORK)) continue;
                double dj = 0.0;
                const unsigned cdj = ch[4 * j + 1] >= 0 ? 4u * at_u(a.codes + (size_t)ch[4 * j] * a.Spad + a.s0, us) : 0u;
#pragma unroll
                for (int c = 0; c < CM; c++) {
                    if (c >= a.C) break;
                    v4 gb = fe[c];
                    for (int j2 = 0; j2 < deg; j2++) {
                        if (j2 == j) continue;
                        v4 m;
                        U4N_MESSAGE(j2, c, m);
                        gb = mul4(gb, m);
                    }
                    if (fl & PLK_UN_LEAF_D) {
                        const v4 y = ld4u(a.dtip + (size_t)c * tabc + (size_t)ch[4 * j + 1] * a.nchar * 4, cdj);
                        dj = fma(pcw[c], fma(gb.d, y.d, fma(gb.c, y.c, fma(gb.b, y.b, gb.a * y.a))), dj);
                    } else if ((fl & PLK_UN_STORE_G) && valid) st4u(a.FN + ((size_t)ch[4 * j + 3] * a.C + c) * n * 4, us4, gb);
                    gc[c] = gb;                       /* the continued child is the last record */
                }
                if (valid && (fl & PLK_UN_LEAF_D)) { double *dp = a.DV + (size_t)(e0 + (fl >> PLK_UN_POS_SHIFT)) * n; asm volatile("" : "+s"(dp)); dp[us] = dj * inv; }
            }
        }
#undef U4N_MESSAGE
    }
}

/* dtip[c][t][code][i] = (dP_e defs[code])[i] (zero for constant definition rows) */
__global__ void k_build_dtip4(int E, int ntips, int nchar, const int *__restrict__ tip_edge,
                              const double *__restrict__ dP, const double *__restrict__ defs, double *__restrict__ dtip,
                              int dzero)
{
    const int t = blockIdx.x, c = blockIdx.y;
    const int e = tip_edge[t];
    for (int idx = threadIdx.x; idx < nchar * 4; idx += blockDim.x) {
        const int code = idx >> 2, i = idx & 3;
        const double *d = defs + code * 4;
        double out = 0.0;
        if (e >= 0 && !(dzero && d[0] == d[1] && d[0] == d[2] && d[0] == d[3])) {
            const double *row = dP + ((size_t)c * E + e) * 16 + i * 4;
            dd acc = dd_make(0.0, 0.0);
            for (int j = 0; j < 4; j++) acc = dd_add(acc, dd_two_prod(row[j], d[j]));
            out = acc.hi;
        }
        dtip[(((size_t)c * (ntips + 1) + t) * nchar + code) * 4 + i] = out;
    }
}

#endif
