/*
 * plk_vec.h -- K2+K3 for medium state spaces (9 <= k <= 32: amino acids) on the vector fp64 pipe.
 * Included by plk_engine.hip.
 *
 * For k = 20 the 16-row granularity of v_mfma_f64_16x16x4_f64 wastes 37 % of the matrix pipe (20 states padded
 * to 32 rows) and every product costs two workgroup barriers for the fragment staging; the fp64 vector peak on
 * MI355X equals the fp64 matrix peak, so a register-resident vector kernel does strictly less work:
 * one site per lane, the vector under construction in K VGPR pairs, P_e (transposed, in program order, the
 * stream of the generic kernel) through scalar loads as SGPR operands of v_fma_f64 -- k*k FMAs per product and
 * no LDS, no barriers.  Leaves use tip tables P_e * defs[code] (double-double built) gathered from L2; vectors
 * waiting for a sibling subtree live in HBM slots (site fastest).
 *
 * Replaces: src/arbplfll.c:139-170 x src/evaluate_site_lhood.c:21-57 x src/util.c:242-301.
 */
#ifndef PLK_VEC_H
#define PLK_VEC_H

#define VEC_BLOCK 256

struct VecArgs {
    long S, Spad;
    int k, C, nops, ntips, nchar, root_mode;
    const int4 *ops;          /* plk_chain_build mode 3 without node storage: observation ops y = staged row of the observation
                                 after next, z = tip slot of the next one (bit 30: it belongs to the next category), w = its
                                 row; MATVEC z = op index of the next MATVEC (wrapping); PUSH / POPMUL y = slot */
    const int *obs_nodes;     /* node of every staged row */
    int first_slot, first_row, second_row;
    const double *PS;         /* [C][nops][K*K]: PS[j*K + i] = P[i][j], zero padded */
    const double *tip;        /* [C][ntips+1][nchar][K]: P_e defs[code]; last slot = the definitions themselves */
    const uint8_t *codes;     /* [N][Spad] */
    const double *cat_prior, *root_w, *w;
    double *slots;            /* [nslots][K][S] */
    int reg_lo;               /* NREG > 0: stack slots reg_lo .. reg_lo + NREG - 1 live in registers, the others in `slots` */
    /* pair-table program (round 3; plk_vec_pt_build of plk_program.h): obs_nodes2 != null.  An observation op's y is then the
     * first row of its table (tip = [C][units * nchar][K]), a staged row may name two nodes (a two-leaf subtree: its code is
     * code(b) * nchar + code(c), its table holds P_a (P_b def_i o P_c def_j)), and the matrix stream has no entry for the
     * edges folded into tables */
    const int *obs_nodes2;
    int units;
    double *site_ll;
    dd *partial;
};

/* Touch every 64-byte line of the K x K matrix at p with scalar loads whose results are never used.  All waves
 * of a CU walk the same matrix stream (C * nmat * K * K * 8 bytes, far larger than the 16 KB scalar cache), so
 * without this every s_load of a product waits for L2 in turn (SQC_DCACHE_MISSES + _DUPLICATE = 60 % of the
 * requests); touched one product ahead, all lines of a matrix are requested at once and the operand loads of the
 * next product hit.
 *
 * The loads are WAITED FOR INSIDE THE SAME asm STATEMENT.  Round 1 left them in flight across compiler-generated
 * code (waiting at the next product): the compiler cannot know that a register named as an asm output is still
 * going to be written after the statement, and in the generated code it reused those SGPRs at once -- for the
 * op-word index and address at the loop head in k_ll_vec<16>, for reloaded spilled pointers at the loop exit in
 * k_ll_vec<20> / <32> -- so a line arriving late replaced an address with matrix bytes: a wild scalar load, i.e.
 * the intermittent GPU memory fault (process abort) recorded in DESIGN.md section 6.  tools/isa_lint.py now checks
 * the generated ISA for any scalar load left pending at the end of an asm block. */
#define VEC_T4(B) "s_load_dword %0, %1, " #B "+0x0\n\ts_load_dword %0, %1, " #B "+0x40\n\t" \
                  "s_load_dword %0, %1, " #B "+0x80\n\ts_load_dword %0, %1, " #B "+0xc0\n\t"
#define VEC_T16(B) VEC_T4(B) VEC_T4(B+0x100) VEC_T4(B+0x200) VEC_T4(B+0x300)
#define VEC_T2(B) "s_load_dword %0, %1, " #B "+0x0\n\ts_load_dword %0, %1, " #B "+0x40\n\t"

template <int K>
__device__ __forceinline__ void vec_touch(const PLK_AS4 double *p)
{
    unsigned d;
    static_assert(K == 16 || K == 20 || K == 32, "line count");
    if constexpr (K == 16)          /* 32 lines */
        asm volatile(VEC_T16(0x0) VEC_T16(0x400) "s_waitcnt lgkmcnt(0)" : "=&s"(d) : "s"(p) : "memory");
    else if constexpr (K == 20)     /* 50 lines */
        asm volatile(VEC_T16(0x0) VEC_T16(0x400) VEC_T16(0x800) VEC_T2(0xc00) "s_waitcnt lgkmcnt(0)" : "=&s"(d) : "s"(p) : "memory");
    else                            /* 128 lines */
        asm volatile(VEC_T16(0x0) VEC_T16(0x400) VEC_T16(0x800) VEC_T16(0xc00) VEC_T16(0x1000) VEC_T16(0x1400) VEC_T16(0x1800)
                     VEC_T16(0x1c00) "s_waitcnt lgkmcnt(0)" : "=&s"(d) : "s"(p) : "memory");
    (void)d;
}

#include "plk_vec_matvec_asm.h"   /* vec_matvec<K>: the product itself, explicit SGPR banks for K = 16, 20 */

/* NREG: stack slots kept in registers (round 3).  With every waiting vector in HBM the kernel moved 22 GB per million
 * sites at BASELINE config 4 (70 pushes and pops of 160 bytes per site and category, 3.5 TB/s: it was bound by that
 * stream, not by arithmetic).  NREG = 3 keeps the three busiest of the tree's four slots on chip (K register pairs each,
 * 2 waves per SIMD instead of 4); the remaining slot still goes through `slots`. */
/* slot Q of the register stack = accumulation registers a[2 Q K .. 2 (Q + 1) K - 1], moved with explicit
 * v_accvgpr_write / read (acc_write / acc_read of plk_fused4.h); tools/isa_lint.py checks that the compiler itself
 * leaves the accumulation registers of these kernels alone */
template <int Q, int K, int... I>
__device__ __forceinline__ void vec_acc_push(const double (&cur)[K], std::integer_sequence<int, I...>)
{
    (acc_write<Q * K + I>(cur[I]), ...);
}
template <int Q, int K, int... I>
__device__ __forceinline__ void vec_acc_popmul(double (&cur)[K], std::integer_sequence<int, I...>)
{
    ((cur[I] *= acc_read<Q * K + I>()), ...);
}

template <int K, int NREG>
__device__ __forceinline__ void k_ll_vec_body(const VecArgs &a)
{
    if constexpr (NREG > 0) {
        static_assert(2 * NREG * K <= 128, "register stack: accumulation registers a0..a127");
        asm volatile("" ::: PLK_CLOBBER_A0_31, PLK_CLOBBER_A32_63, PLK_CLOBBER_A64_127);     /* the kernel owns a0..a127 */
    }
    const int tid = threadIdx.x;
    const long s = (long)blockIdx.x * VEC_BLOCK + tid;
    const bool valid = s < a.S;
    const long sc = valid ? s : a.S - 1;
    const PLK_AS4 int *ops = as_uniform(reinterpret_cast<const int *>(a.ops));
    const PLK_AS4 double *prior = as_uniform(a.cat_prior), *rw = as_uniform(a.root_w);
    const size_t tabc = (a.obs_nodes2 ? (size_t)a.units : (size_t)(a.ntips + 1)) * a.nchar * K;

    double sum = 0.0;
    int Eexp = 0;
    bool have = false;
    /* Observation ops wait for two dependent global loads (pattern code, then its tip-table row).  The code of the NEXT
     * observation op is requested one op ahead (one register); the two-ahead chain that also prefetches the row (K
     * register pairs) is what k_down_vec uses, but here it cost a wave of occupancy (117 against 99 VGPRs) and 9 % of
     * the speed (6.77 against 6.2 ms at BASELINE config 4). */
    const PLK_AS4 int *obs = as_uniform(a.obs_nodes);
    const PLK_AS4 int *obs2 = as_uniform(a.obs_nodes2);
    const bool pt = a.obs_nodes2 != nullptr;
    /* pattern code of a staged row: one node's code, or the combined code of a two-leaf subtree */
    auto row_code = [&](int row) -> int {
        int cd = a.codes[(size_t)obs[row] * a.Spad + sc];
        if (pt) { const int n2 = obs2[row]; if (n2 >= 0) cd = cd * a.nchar + a.codes[(size_t)n2 * a.Spad + sc]; }
        return cd;
    };
    int code_next = a.first_slot >= 0 ? row_code(a.first_row) : 0;
    for (int c = 0; c < a.C; c++) {
        double cur[K];
#pragma unroll
        for (int i = 0; i < K; i++) cur[i] = 1.0;
        int esc = 0;
        const PLK_AS4 double *PSc = as_uniform(a.PS) + (size_t)c * a.nops * K * K;
        const double *tipc = a.tip + (size_t)c * tabc;
        for (int pc = 0; pc < a.nops; pc++) {
            const int ox = ops[4 * pc], oy = ops[4 * pc + 1];
            const int code = ox & 0xff;
            if (code == OP_MATVEC) {
                const int oz = ops[4 * pc + 2];
                const PLK_AS4 double *M = PSc + (size_t)pc * K * K;
                /* request (and wait for) the lines of the NEXT product's matrix -- the next category's first one at
                 * the end -- before multiplying with this one, whose lines the previous product requested */
                const int nc = oz <= pc ? c + 1 : c;
                vec_touch<K>(as_uniform(a.PS) + ((size_t)(nc < a.C ? nc : c) * a.nops + oz) * K * K);
                double acc[K];
                vec_matvec<K>(M, cur, acc);
#pragma unroll
                for (int i = 0; i < K; i++) cur[i] = acc[i];
            } else if (code == OP_TIP_SET || code == OP_TIP_MUL || code == OP_NODE_MUL) {
                const int ow = ops[4 * pc + 3];                     /* staged row of the next observation op (cyclic) */
                const int t = code == OP_NODE_MUL ? a.ntips : (ox >> 8);
                const size_t trow = pt ? (size_t)oy + code_next : (size_t)t * a.nchar + code_next;
                const double2 *tp = reinterpret_cast<const double2 *>(tipc + trow * K);
                if (code == OP_TIP_SET) {
#pragma unroll
                    for (int i = 0; i < K; i += 2) { const double2 v = tp[i >> 1]; cur[i] = v.x; cur[i + 1] = v.y; }
                } else {
#pragma unroll
                    for (int i = 0; i < K; i += 2) { const double2 v = tp[i >> 1]; cur[i] *= v.x; cur[i + 1] *= v.y; }
                }
                code_next = row_code(ow);
            } else if (NREG > 0 && code == OP_PUSH && oy - a.reg_lo >= 0 && oy - a.reg_lo < NREG) {
                const int r = oy - a.reg_lo;
                if (r == 0) vec_acc_push<0, K>(cur, std::make_integer_sequence<int, K>());
                else if (NREG > 1 && r == 1) vec_acc_push<(NREG > 1 ? 1 : 0), K>(cur, std::make_integer_sequence<int, K>());
                else vec_acc_push<(NREG > 2 ? 2 : 0), K>(cur, std::make_integer_sequence<int, K>());
            } else if (NREG > 0 && code == OP_POPMUL && oy - a.reg_lo >= 0 && oy - a.reg_lo < NREG) {
                const int r = oy - a.reg_lo;
                if (r == 0) vec_acc_popmul<0, K>(cur, std::make_integer_sequence<int, K>());
                else if (NREG > 1 && r == 1) vec_acc_popmul<(NREG > 1 ? 1 : 0), K>(cur, std::make_integer_sequence<int, K>());
                else vec_acc_popmul<(NREG > 2 ? 2 : 0), K>(cur, std::make_integer_sequence<int, K>());
            } else if (code == OP_PUSH) {
                /* plane base pinned in SGPRs, the lane's site as a 32-bit offset: otherwise the compiler keeps K 64-bit
                 * lane addresses live across the loop (40 VGPRs at K = 20, a wave of occupancy) */
                double *sp = a.slots + (size_t)oy * K * a.S;
                asm volatile("" : "+s"(sp));
                if (valid) {
#pragma unroll
                    for (int i = 0; i < K; i++) (sp + (size_t)i * a.S)[(unsigned)sc] = cur[i];
                }
            } else if (code == OP_POPMUL) {
                const double *sp = a.slots + (size_t)oy * K * a.S;
                asm volatile("" : "+s"(sp));
#pragma unroll
                for (int i = 0; i < K; i++) cur[i] *= (sp + (size_t)i * a.S)[(unsigned)sc];
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
            for (int i = 0; i < K; i++) lh = fma(rw[i], cur[i], lh);
        }
        const double term = prior[c] * lh;
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

template <int K>
__global__ __launch_bounds__(VEC_BLOCK) void k_ll_vec(VecArgs a) { k_ll_vec_body<K, 0>(a); }

/* the register-stack variant: 2 waves per SIMD (256 registers per lane) */
template <int K, int NREG>
__global__ __launch_bounds__(VEC_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_ll_vec_rs(VecArgs a) { k_ll_vec_body<K, NREG>(a); }

/* Tables of the pair-table program for K-state vectors (the k = 4 version is k_build_tables_pt in plk_engine.hip):
 * tab[4][ntab] = first unit | edge | pair: the two leaf edges, else -1.  One block per (table, category).
 *   single  out[code][i]         = (P_e defs[code])_i          pair  out[cb*nchar+cc][i] = (P_a (P_b defs[cb] o P_c defs[cc]))_i
 *   pseudo  out[code][i]         = defs[code][i]
 * in double-double from the unrounded P; a constant vector maps to itself exactly (src/util.c:276-283). */
__global__ __launch_bounds__(256) void k_build_tables_pt_vec(int k, int K, int E, int ntab, int units, int nchar, const int *__restrict__ tab,
                                                              const dd *__restrict__ Pdd, const double *__restrict__ defs /* [nchar][K] */,
                                                              double *__restrict__ tip)
{
    const int t = blockIdx.x, c = blockIdx.y;
    const int unit = tab[t], e = tab[ntab + t], eb = tab[2 * ntab + t], ec = tab[3 * ntab + t];
    double *out = tip + ((size_t)c * units + unit) * nchar * K;
    const size_t kk = (size_t)k * k;
    auto is_const = [&](const double *d) { bool cst = true; for (int j = 1; j < k; j++) cst = cst && d[j] == d[0]; return cst; };
    /* (P_edge d)_i in double-double; d constant: d_0 */
    auto leaf = [&](int edge, const double *d, int i) -> dd {
        if (is_const(d)) return dd_make(d[0], 0.0);
        const dd *Pm = Pdd + ((size_t)c * E + edge) * kk + (size_t)i * k;
        dd acc = dd_make(0.0, 0.0);
        for (int j = 0; j < k; j++) acc = dd_add(acc, dd_mul_d(Pm[j], d[j]));
        return acc;
    };
    if (eb < 0) {
        if (blockIdx.z != 0) return;                 /* the slices of gridDim.z are for the pair tables */
        for (int idx = threadIdx.x; idx < nchar * K; idx += blockDim.x) {
            const int code = idx / K, i = idx - code * K;
            const double *d = defs + (size_t)code * K;
            out[idx] = i >= k ? 0.0 : (e < 0 ? d[i] : leaf(e, d, i).hi);
        }
        return;
    }
    /* a pair.  Phase 1: the 2 x nchar leaf vectors P_b def[cb], P_c def[cc] (double-double) into LDS; phase 2: for this
     * block's slice of the nchar^2 code pairs (gridDim.z slices) v = vb o vc and P_a v, one output entry per lane and step */
    extern __shared__ double pt_lds[];                       /* [2][nchar][k] hi, then the same lo */
    double *lh = pt_lds, *ll_ = pt_lds + (size_t)2 * nchar * k;
    for (int idx = threadIdx.x; idx < 2 * nchar * k; idx += blockDim.x) {
        const int which = idx / (nchar * k), rem = idx - which * nchar * k, code = rem / k, i = rem - code * k;
        const dd v = leaf(which ? ec : eb, defs + (size_t)code * K, i);
        lh[idx] = v.hi; ll_[idx] = v.lo;
    }
    __syncthreads();
    const int ncomb = nchar * nchar, per = (ncomb + gridDim.z - 1) / gridDim.z;
    const int c0 = blockIdx.z * per, c1 = c0 + per < ncomb ? c0 + per : ncomb;
    const dd *Pa = Pdd + ((size_t)c * E + e) * kk;
    for (int idx = c0 * K + threadIdx.x; idx < c1 * K; idx += blockDim.x) {
        const int comb = idx / K, i = idx - comb * K;
        const int cb = comb / nchar, cc = comb - cb * nchar;
        const double *bh = lh + (size_t)cb * k, *bl = ll_ + (size_t)cb * k;
        const double *ch = lh + (size_t)(nchar + cc) * k, *cl = ll_ + (size_t)(nchar + cc) * k;
        double o = 0.0;
        if (i < k) {
            /* is the product vector constant?  (both leaves missing: it maps to itself) */
            const dd p0 = dd_mul(dd_make(bh[0], bl[0]), dd_make(ch[0], cl[0]));
            bool cst = true;
            dd acc = dd_make(0.0, 0.0);
            for (int j = 0; j < k; j++) {
                const dd pj = dd_mul(dd_make(bh[j], bl[j]), dd_make(ch[j], cl[j]));
                cst = cst && pj.hi == p0.hi && pj.lo == p0.lo;
                acc = dd_add(acc, dd_mul(Pa[(size_t)i * k + j], pj));
            }
            o = cst ? p0.hi : acc.hi;
        }
        out[idx] = o;
    }
}

/* tip[((c*(ntips+1) + t)*nchar + code)*K + i] = (P_e defs[code])[i] in double-double (exact for constant
 * definition rows, src/util.c:276-283); slot ntips (edge -1) holds defs[code] itself; entries i >= k are 0
 * for a product and 1... 0 for the padding of the definitions (the padded states never enter a product) */
__global__ void k_build_tip_vec(int k, int K, int E, int ntips, int nchar, const int *__restrict__ tip_edge,
                                const dd *__restrict__ Pdd, const double *__restrict__ defs /* [nchar][K] */,
                                double *__restrict__ tip)
{
    __shared__ int s_kind[PLK_DEF_KIND_CACHE];
    const int t = blockIdx.x, c = blockIdx.y;
    const int e = tip_edge[t];
    for (int code = threadIdx.x; code < nchar && code < PLK_DEF_KIND_CACHE; code += blockDim.x) s_kind[code] = def_row_kind(defs + (size_t)code * K, k);
    __syncthreads();
    for (int idx = threadIdx.x; idx < nchar * K; idx += blockDim.x) {
        const int code = idx / K, i = idx - code * K;
        const double *d = defs + (size_t)code * K;
        double out = 0.0;
        if (i < k) {
            if (e < 0) out = d[i];
            else {
                /* constant rows map to themselves, observed-state rows copy a column of P_e (def_row_kind, plk_mfma.h) */
                const int kind = code < PLK_DEF_KIND_CACHE ? s_kind[code] : def_row_kind(d, k);
                const dd *Pm = Pdd + ((size_t)c * E + e) * k * k + (size_t)i * k;
                if (kind == -2) out = d[0];
                else if (kind >= 0) out = Pm[kind].hi;
                else {
                    dd acc = dd_make(0.0, 0.0);
                    for (int j = 0; j < k; j++) acc = dd_add(acc, dd_mul_d(Pm[j], d[j]));
                    out = acc.hi;
                }
            }
        }
        tip[(((size_t)c * (ntips + 1) + t) * nchar + code) * K + i] = out;
    }
}

#endif
