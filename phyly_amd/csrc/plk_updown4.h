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
 *   - only internal edges and internal nodes are stored; leaf-edge vectors P_e B_b and
 *     derivative vectors dP_e B_b are gathered by pattern code from small tables
 *     (double-double built) that live in L1/L2;
 *   - P_e, P_e^T and dP_e are wave-uniform: scalar loads, SGPR operands, no LDS.
 */
#ifndef PLK_UPDOWN4_H
#define PLK_UPDOWN4_H

struct Up4Args {
    long S, Spad, s0, n;
    int N, E, C, nchar, ntips, root_mode;
    int dzero;                     /* 1: edge-form matrices have zero row sums (dP) */
    const int *indptr, *indices, *preorder;
    const int *node_has_data, *edge_tip, *edge_int, *node_int;
    const double *P, *dP;          /* [C][E][4][4] row-major */
    const double *tip, *dtip;      /* [C][ntips+1][nchar][4]; slot ntips of tip = raw definitions */
    const uint8_t *codes;
    const double *cat_prior, *root_w;
    const int *edge_mask, *node_mask;
    double *EV, *LN, *FN;          /* [(ent*C + c)][n][4] */
    double *LH, *DV, *MV;          /* [n], [E][n], [N][4][n] */
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

__global__ __launch_bounds__(UD4_BLOCK) void k_down_store4(Up4Args a)
{
    const long sl = (long)blockIdx.x * UD4_BLOCK + threadIdx.x;
    const bool valid = sl < a.n;
    const long slc = valid ? sl : a.n - 1;
    const long sg = a.s0 + slc;
    const size_t n = (size_t)a.n;
    const PLK_AS4 int *pre = as_uniform(a.preorder), *ip = as_uniform(a.indptr), *ix = as_uniform(a.indices);
    const PLK_AS4 int *has = as_uniform(a.node_has_data), *etip = as_uniform(a.edge_tip);
    const PLK_AS4 int *eint = as_uniform(a.edge_int), *nint = as_uniform(a.node_int);
    const PLK_AS4 double *Pm = as_uniform(a.P), *prior = as_uniform(a.cat_prior), *rw = as_uniform(a.root_w);
    const size_t tabc = (size_t)(a.ntips + 1) * a.nchar * 4;
    double lh_total = 0.0;
    for (int c = 0; c < a.C; c++) {
        const double *tipc = a.tip + (size_t)c * tabc;
        double lh_c = 0.0;
        for (int u = a.N - 1; u >= 0; u--) {
            const int nd = pre[u];
            const int start = ip[nd], stop = ip[nd + 1];
            if (start == stop) continue;
            v4 acc = v4{1.0, 1.0, 1.0, 1.0};
            if (has[nd]) acc = ld4(tipc + ((size_t)a.ntips * a.nchar + a.codes[(size_t)nd * a.Spad + sg]) * 4);
            for (int idx = start; idx < stop; idx++) {
                const int b = ix[idx];
                v4 m;
                const int t = etip[idx];
                if (t >= 0) {
                    m = ld4(tipc + ((size_t)t * a.nchar + a.codes[(size_t)b * a.Spad + sg]) * 4);
                } else {
                    const v4 x = ld4(a.LN + (((size_t)nint[b] * a.C + c) * n + slc) * 4);
                    m = mv4(Pm + ((size_t)c * a.E + idx) * 16, x);
                    if (const4(x)) m = x;          /* exact constant column (src/util.c:276-283) */
                    if (valid) st4(a.EV + (((size_t)eint[idx] * a.C + c) * n + slc) * 4, m);
                }
                acc = mul4(acc, m);
            }
            if (valid) st4(a.LN + (((size_t)nint[nd] * a.C + c) * n + slc) * 4, acc);
            if (u == 0) {
                if (a.root_mode == PLK_ROOT_NONE) lh_c = ((acc.a + acc.b) + acc.c) + acc.d;
                else if (a.root_mode == PLK_ROOT_UNIFORM) lh_c = (((acc.a + acc.b) + acc.c) + acc.d) * 0.25;
                else lh_c = fma(rw[3], acc.d, fma(rw[2], acc.c, fma(rw[1], acc.b, rw[0] * acc.a)));
            }
        }
        lh_total = fma(prior[c], lh_c, lh_total);
    }
    if (valid) a.LH[sl] = lh_total;
}

template <bool DERIV, bool MARG>
__global__ __launch_bounds__(UD4_BLOCK) void k_up4(Up4Args a)
{
    const long sl = (long)blockIdx.x * UD4_BLOCK + threadIdx.x;
    const bool valid = sl < a.n;
    const long slc = valid ? sl : a.n - 1;
    const long sg = a.s0 + slc;
    const size_t n = (size_t)a.n;
    const PLK_AS4 int *pre = as_uniform(a.preorder), *ip = as_uniform(a.indptr), *ix = as_uniform(a.indices);
    const PLK_AS4 int *has = as_uniform(a.node_has_data), *etip = as_uniform(a.edge_tip);
    const PLK_AS4 int *eint = as_uniform(a.edge_int), *nint = as_uniform(a.node_int);
    const PLK_AS4 double *Pm = as_uniform(a.P), *dPm = as_uniform(a.dP);
    const PLK_AS4 double *prior = as_uniform(a.cat_prior), *rw = as_uniform(a.root_w);
    const size_t tabc = (size_t)(a.ntips + 1) * a.nchar * 4;
    const double inv = 1.0 / a.LH[slc];
    const int root = pre[0];
    const v4 w = v4{rw[0], rw[1], rw[2], rw[3]};

    {   /* root: forward vector = root prior weights; its marginal */
        v4 macc = v4{0.0, 0.0, 0.0, 0.0};
        for (int c = 0; c < a.C; c++) {
            if (valid) st4(a.FN + (((size_t)nint[root] * a.C + c) * n + slc) * 4, w);
            if (MARG) {
                const v4 l = ld4(a.LN + (((size_t)nint[root] * a.C + c) * n + slc) * 4);
                const double pc = prior[c];
                macc.a = fma(pc * w.a, l.a, macc.a); macc.b = fma(pc * w.b, l.b, macc.b);
                macc.c = fma(pc * w.c, l.c, macc.c); macc.d = fma(pc * w.d, l.d, macc.d);
            }
        }
        if (MARG && valid && (!a.node_mask || as_uniform(a.node_mask)[root])) {
            double *mv = a.MV + (size_t)root * 4 * n + sl;
            mv[0] = macc.a * inv; mv[n] = macc.b * inv; mv[2 * n] = macc.c * inv; mv[3 * n] = macc.d * inv;
        }
    }

    for (int u = 0; u < a.N; u++) {
        const int nd = pre[u];
        const int start = ip[nd], stop = ip[nd + 1];
        if (start == stop) continue;
        const bool hd = has[nd] != 0;
        const int chn = hd ? a.codes[(size_t)nd * a.Spad + sg] : 0;
        for (int idx = start; idx < stop; idx++) {
            const int b = ix[idx];
            const int tb = etip[idx];
            const bool b_leaf = tb >= 0;
            const bool want_d = DERIV && (!a.edge_mask || as_uniform(a.edge_mask)[idx]);
            const bool want_m = MARG && (!a.node_mask || as_uniform(a.node_mask)[b]);
            const bool want_f = !b_leaf || want_m;
            if (!want_d && !want_f) continue;
            const int chb = b_leaf ? a.codes[(size_t)b * a.Spad + sg] : 0;
            double dsum = 0.0;
            v4 macc = v4{0.0, 0.0, 0.0, 0.0};
            for (int c = 0; c < a.C; c++) {
                const double *tipc = a.tip + (size_t)c * tabc;
                v4 fe = ld4(a.FN + (((size_t)nint[nd] * a.C + c) * n + slc) * 4);
                if (hd) fe = mul4(fe, ld4(tipc + ((size_t)a.ntips * a.nchar + chn) * 4));
                for (int idx2 = start; idx2 < stop; idx2++) {
                    if (idx2 == idx) continue;
                    const int t2 = etip[idx2];
                    if (t2 >= 0) fe = mul4(fe, ld4(tipc + ((size_t)t2 * a.nchar + a.codes[(size_t)ix[idx2] * a.Spad + sg]) * 4));
                    else fe = mul4(fe, ld4(a.EV + (((size_t)eint[idx2] * a.C + c) * n + slc) * 4));
                }
                const double pc = prior[c];
                if (want_d) {
                    v4 y;
                    if (b_leaf) y = ld4(a.dtip + (size_t)c * tabc + ((size_t)tb * a.nchar + chb) * 4);
                    else {
                        const v4 x = ld4(a.LN + (((size_t)nint[b] * a.C + c) * n + slc) * 4);
                        y = mv4(dPm + ((size_t)c * a.E + idx) * 16, x);
                        if (a.dzero && const4(x)) y = v4{0.0, 0.0, 0.0, 0.0};     /* rows of dP sum to zero (src/util.c:338-345) */
                    }
                    const double d = fma(fe.d, y.d, fma(fe.c, y.c, fma(fe.b, y.b, fe.a * y.a)));
                    dsum = fma(pc, d, dsum);
                }
                if (want_f) {
                    const v4 fb = mtv4(Pm + ((size_t)c * a.E + idx) * 16, fe);
                    if (!b_leaf && valid) st4(a.FN + (((size_t)nint[b] * a.C + c) * n + slc) * 4, fb);
                    if (want_m) {
                        const v4 lb = b_leaf ? ld4(tipc + ((size_t)a.ntips * a.nchar + chb) * 4)
                                             : ld4(a.LN + (((size_t)nint[b] * a.C + c) * n + slc) * 4);
                        macc.a = fma(pc * fb.a, lb.a, macc.a); macc.b = fma(pc * fb.b, lb.b, macc.b);
                        macc.c = fma(pc * fb.c, lb.c, macc.c); macc.d = fma(pc * fb.d, lb.d, macc.d);
                    }
                }
            }
            if (want_d && valid) a.DV[(size_t)idx * n + sl] = dsum * inv;
            if (want_m && valid) {
                double *mv = a.MV + (size_t)b * 4 * n + sl;
                mv[0] = macc.a * inv; mv[n] = macc.b * inv; mv[2 * n] = macc.c * inv; mv[3 * n] = macc.d * inv;
            }
        }
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
