/* plk_fused4.h -- fused k = 4 traversal kernel (v2: C++ with ping-pong register sets); included by plk_engine.hip */
#ifndef PLK_FUSED4_H
#define PLK_FUSED4_H

/* ====================================================================== */
/* K2+K3 fused: k = 4, accumulation-register stack                         */
/* ====================================================================== */

/*
 * One alignment site per lane.  The partial-likelihood vector being built lives in
 * 4 VGPR pairs; vectors that must wait for a sibling subtree are parked in the
 * AGPR half of the unified register file (gfx950: 512 registers per lane), so the
 * traversal makes no HBM or LDS traffic for partials at all.  The program (ops)
 * and the P matrices are wave-uniform and are fetched with scalar loads through
 * the constant address space, one op ahead; FMAs take P entries as SGPR operands.
 * Tip tables (P_e * definitions) of the current category and the tile's pattern
 * codes sit in LDS.
 */
#define PLK_AS4 __attribute__((address_space(4)))
template <typename T>
__device__ static inline const PLK_AS4 T *as_uniform(const T *p)
{
    return (const PLK_AS4 T *)(p);
}

struct FusedArgs {
    long S, Spad;
    int C, nops, nmat, ntips, nchar, nobs;
    int root_mode, first_row;
    const int4 *ops;         /* x = opcode | tip<<8, y = row / slot, z = next observation row, w = matrix */
    const double *PS;        /* [C][nmat][16] transposed: PS[j*4+i] = P[i][j] */
    const double *tip;       /* [C][ntips][nchar][4] */
    const uint8_t *codes;    /* [N][Spad] */
    const int *obs_nodes;    /* [nobs] node per staged row */
    const double *defs;      /* [nchar][4] */
    const double *cat_prior; /* [C] */
    const double *root_w;    /* [4] */
    const double *w;         /* site weights or null */
    double *site_ll;         /* [S] or null */
    dd *partial;             /* [gridDim.x] or null */
};

__device__ static inline int frexp_exp(double m)
{
    /* exponent e with m = f * 2^e, 0.5 <= f < 1; 0 for m == 0 */
    return m > 0.0 ? __builtin_amdgcn_frexp_exp(m) : 0;
}

template <int IDX>
__device__ __forceinline__ void acc_write(double x)
{
    const int lo = __double2loint(x), hi = __double2hiint(x);
    asm volatile("v_accvgpr_write_b32 a[%2], %0\n\tv_accvgpr_write_b32 a[%3], %1"
                 :: "v"(lo), "v"(hi), "n"(2 * IDX), "n"(2 * IDX + 1));
}
template <int IDX>
__device__ __forceinline__ double acc_read()
{
    int lo, hi;
    asm volatile("v_accvgpr_read_b32 %0, a[%2]\n\tv_accvgpr_read_b32 %1, a[%3]"
                 : "=v"(lo), "=v"(hi) : "n"(2 * IDX), "n"(2 * IDX + 1));
    return __hiloint2double(hi, lo);
}
/* slot d of site j (compile-time j) lives in AGPR doubles ((d*NS + j)*4 .. +3) */
template <int D, int NS>
__device__ __forceinline__ void stack_push(int d, int j, double c0, double c1, double c2, double c3)
{
    if constexpr (D > 0) {
        if (d == D - 1) {
            if (j == 0) {
                acc_write<((D - 1) * NS) * 4 + 0>(c0); acc_write<((D - 1) * NS) * 4 + 1>(c1);
                acc_write<((D - 1) * NS) * 4 + 2>(c2); acc_write<((D - 1) * NS) * 4 + 3>(c3);
            } else {
                acc_write<((D - 1) * NS + (NS - 1)) * 4 + 0>(c0); acc_write<((D - 1) * NS + (NS - 1)) * 4 + 1>(c1);
                acc_write<((D - 1) * NS + (NS - 1)) * 4 + 2>(c2); acc_write<((D - 1) * NS + (NS - 1)) * 4 + 3>(c3);
            }
        } else stack_push<D - 1, NS>(d, j, c0, c1, c2, c3);
    }
}
template <int D, int NS>
__device__ __forceinline__ void stack_popmul(int d, int j, double &c0, double &c1, double &c2, double &c3)
{
    if constexpr (D > 0) {
        if (d == D - 1) {
            if (j == 0) {
                c0 *= acc_read<((D - 1) * NS) * 4 + 0>(); c1 *= acc_read<((D - 1) * NS) * 4 + 1>();
                c2 *= acc_read<((D - 1) * NS) * 4 + 2>(); c3 *= acc_read<((D - 1) * NS) * 4 + 3>();
            } else {
                c0 *= acc_read<((D - 1) * NS + (NS - 1)) * 4 + 0>(); c1 *= acc_read<((D - 1) * NS + (NS - 1)) * 4 + 1>();
                c2 *= acc_read<((D - 1) * NS + (NS - 1)) * 4 + 2>(); c3 *= acc_read<((D - 1) * NS + (NS - 1)) * 4 + 3>();
            }
        } else stack_popmul<D - 1, NS>(d, j, c0, c1, c2, c3);
    }
}

#define PLK_CLOBBER_A0_31 "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", \
    "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31"
#define PLK_CLOBBER_A32_63 "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", \
    "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63"
#define PLK_CLOBBER_A64_127 "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", \
    "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", \
    "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", \
    "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"

/* one traversal op: OUT = f(IN) for the NS sites of this lane (IN and OUT are distinct
 * register sets; the caller alternates them so that no result has to be copied back) */
#define PLK_FUSED_EXEC(OX, OY, OZ, IN, OUT)                                                              \
    do {                                                                                                  \
        const int code_ = (OX) & 0xff;                                                                    \
        if (code_ == OP_MATVEC) {                                                                         \
            _Pragma("unroll") for (int j = 0; j < NS; j++) {                                              \
                double n0 = m0 * IN[j][0], n1 = m1 * IN[j][0], n2 = m2 * IN[j][0], n3 = m3 * IN[j][0];    \
                n0 = fma(m4, IN[j][1], n0); n1 = fma(m5, IN[j][1], n1); n2 = fma(m6, IN[j][1], n2); n3 = fma(m7, IN[j][1], n3);     \
                n0 = fma(m8, IN[j][2], n0); n1 = fma(m9, IN[j][2], n1); n2 = fma(m10, IN[j][2], n2); n3 = fma(m11, IN[j][2], n3);   \
                n0 = fma(m12, IN[j][3], n0); n1 = fma(m13, IN[j][3], n1); n2 = fma(m14, IN[j][3], n2); n3 = fma(m15, IN[j][3], n3); \
                OUT[j][0] = n0; OUT[j][1] = n1; OUT[j][2] = n2; OUT[j][3] = n3;                           \
            }                                                                                             \
            mi++;                                                                                         \
            PLK_LOAD_M(PSc + mi * 16);   /* matrices are consumed in stream order */                      \
        } else if (code_ == OP_TIP_MUL || code_ == OP_TIP_SET) {                                          \
            const int t_ = (OX) >> 8;                                                                     \
            _Pragma("unroll") for (int j = 0; j < NS; j++) {                                              \
                const double2 *tp = reinterpret_cast<const double2 *>(tip_lds + t_ * nchar4 + ch_next[j] * 4); \
                const double2 v01 = tp[0], v23 = tp[1];                                                   \
                ch_next[j] = code_lds[(OZ) * (PLK_TILE * NS) + j * PLK_TILE + tid];                       \
                if (code_ == OP_TIP_SET) { OUT[j][0] = v01.x; OUT[j][1] = v01.y; OUT[j][2] = v23.x; OUT[j][3] = v23.y; } \
                else { OUT[j][0] = IN[j][0] * v01.x; OUT[j][1] = IN[j][1] * v01.y; OUT[j][2] = IN[j][2] * v23.x; OUT[j][3] = IN[j][3] * v23.y; } \
            }                                                                                             \
        } else if (code_ == OP_POPMUL) {                                                                  \
            _Pragma("unroll") for (int j = 0; j < NS; j++) {                                              \
                OUT[j][0] = IN[j][0]; OUT[j][1] = IN[j][1]; OUT[j][2] = IN[j][2]; OUT[j][3] = IN[j][3];   \
                stack_popmul<D, NS>((OY), j, OUT[j][0], OUT[j][1], OUT[j][2], OUT[j][3]);                 \
            }                                                                                             \
        } else if (code_ == OP_PUSH) {                                                                    \
            _Pragma("unroll") for (int j = 0; j < NS; j++) {                                              \
                stack_push<D, NS>((OY), j, IN[j][0], IN[j][1], IN[j][2], IN[j][3]);                       \
                OUT[j][0] = IN[j][0]; OUT[j][1] = IN[j][1]; OUT[j][2] = IN[j][2]; OUT[j][3] = IN[j][3];   \
            }                                                                                             \
        } else if (code_ == OP_SCALE) {                                                                   \
            _Pragma("unroll") for (int j = 0; j < NS; j++) {                                              \
                const double mx = fmax(fmax(IN[j][0], IN[j][1]), fmax(IN[j][2], IN[j][3]));               \
                const int e = frexp_exp(mx);                                                              \
                OUT[j][0] = ldexp(IN[j][0], -e); OUT[j][1] = ldexp(IN[j][1], -e);                         \
                OUT[j][2] = ldexp(IN[j][2], -e); OUT[j][3] = ldexp(IN[j][3], -e);                         \
                esc[j] += e;                                                                              \
            }                                                                                             \
        } else if (code_ == OP_NODE_MUL) {                                                                \
            _Pragma("unroll") for (int j = 0; j < NS; j++) {                                              \
                const double *dv = a.defs + ch_next[j] * 4;                                               \
                ch_next[j] = code_lds[(OZ) * (PLK_TILE * NS) + j * PLK_TILE + tid];                       \
                OUT[j][0] = IN[j][0] * dv[0]; OUT[j][1] = IN[j][1] * dv[1];                               \
                OUT[j][2] = IN[j][2] * dv[2]; OUT[j][3] = IN[j][3] * dv[3];                               \
            }                                                                                             \
        } else { /* OP_END / padding: pass through */                                                     \
            _Pragma("unroll") for (int j = 0; j < NS; j++) {                                              \
                OUT[j][0] = IN[j][0]; OUT[j][1] = IN[j][1]; OUT[j][2] = IN[j][2]; OUT[j][3] = IN[j][3];   \
            }                                                                                             \
        }                                                                                                 \
    } while (0)

#define PLK_LOAD_M(P_)                                                                          \
    do {                                                                                        \
        m0 = (P_)[0]; m1 = (P_)[1]; m2 = (P_)[2]; m3 = (P_)[3]; m4 = (P_)[4]; m5 = (P_)[5]; m6 = (P_)[6]; m7 = (P_)[7]; \
        m8 = (P_)[8]; m9 = (P_)[9]; m10 = (P_)[10]; m11 = (P_)[11]; m12 = (P_)[12]; m13 = (P_)[13]; m14 = (P_)[14]; m15 = (P_)[15]; \
    } while (0)

template <int D, int NS>
__global__ __launch_bounds__(PLK_TILE) void k_ll_fused4(FusedArgs a)
{
    /* reserve the AGPRs the stack uses (8 per slot and site) */
    if constexpr (D * NS <= 4) asm volatile("" ::: PLK_CLOBBER_A0_31);
    else if constexpr (D * NS <= 8) asm volatile("" ::: PLK_CLOBBER_A0_31, PLK_CLOBBER_A32_63);
    else asm volatile("" ::: PLK_CLOBBER_A0_31, PLK_CLOBBER_A32_63, PLK_CLOBBER_A64_127);

    extern __shared__ double lds_dyn[];
    /* LDS: tip table of the current category, then the staged codes of this tile */
    double *tip_lds = lds_dyn;
    const int tip_doubles = a.ntips * a.nchar * 4;
    uint8_t *code_lds = reinterpret_cast<uint8_t *>(lds_dyn + tip_doubles);
    constexpr int TILE = PLK_TILE * NS;

    const long tile0 = (long)blockIdx.x * TILE;
    const int tid = threadIdx.x;

    /* stage codes[obs][TILE] for this tile: rows are padded to Spad (multiple of 1024) */
    {
        const int ndw = a.nobs * (TILE / 4);
        uint32_t *dst = reinterpret_cast<uint32_t *>(code_lds);
        for (int idx = tid; idx < ndw; idx += PLK_TILE) {
            int row = idx / (TILE / 4), col = idx - row * (TILE / 4);
            const uint32_t *src = reinterpret_cast<const uint32_t *>(a.codes + (size_t)a.obs_nodes[row] * a.Spad + tile0);
            dst[idx] = src[col];
        }
    }

    const PLK_AS4 int *ops = as_uniform(reinterpret_cast<const int *>(a.ops));
    const PLK_AS4 double *prior = as_uniform(a.cat_prior);
    const PLK_AS4 double *rootw = as_uniform(a.root_w);
    const int nchar4 = a.nchar * 4;

    double sum[NS];
    int Eexp[NS];
    bool have[NS];
#pragma unroll
    for (int j = 0; j < NS; j++) { sum[j] = 0.0; Eexp[j] = 0; have[j] = false; }

    for (int c = 0; c < a.C; c++) {
        __syncthreads();
        {
            const double2 *src = reinterpret_cast<const double2 *>(a.tip + (size_t)c * tip_doubles);
            double2 *dst = reinterpret_cast<double2 *>(tip_lds);
            for (int idx = tid; idx < tip_doubles / 2; idx += PLK_TILE) dst[idx] = src[idx];
        }
        __syncthreads();

        double A[NS][4], B[NS][4];
        int esc[NS], ch_next[NS];
#pragma unroll
        for (int j = 0; j < NS; j++) {
            A[j][0] = A[j][1] = A[j][2] = A[j][3] = 1.0;
            esc[j] = 0;
            ch_next[j] = code_lds[a.first_row * TILE + j * PLK_TILE + tid];   /* code for the first observation op */
        }
        const PLK_AS4 double *PSc = as_uniform(a.PS) + (size_t)c * (a.nmat + 1) * 16;
        int mi = 0;
        double m0, m1, m2, m3, m4, m5, m6, m7, m8, m9, m10, m11, m12, m13, m14, m15;
        PLK_LOAD_M(PSc);
        int ax = ops[0], ay = ops[1], az = ops[2];
        int bx = ops[4], by = ops[5], bz = ops[6];

        /* ops are executed in pairs (program padded to an even count + one spare pair) */
        for (int pc = 0; pc < a.nops; pc += 2) {
            const int nax = ops[4 * pc + 8], nay = ops[4 * pc + 9], naz = ops[4 * pc + 10];
            const int nbx = ops[4 * pc + 12], nby = ops[4 * pc + 13], nbz = ops[4 * pc + 14];
            PLK_FUSED_EXEC(ax, ay, az, A, B);
            PLK_FUSED_EXEC(bx, by, bz, B, A);
            ax = nax; ay = nay; az = naz; bx = nbx; by = nby; bz = nbz;
        }
        /* root expectation (src/model.c:283-350) and category mixing (src/arbplfll.c:165) */
#pragma unroll
        for (int j = 0; j < NS; j++) {
            double lh;
            if (a.root_mode == PLK_ROOT_NONE) lh = ((A[j][0] + A[j][1]) + A[j][2]) + A[j][3];
            else if (a.root_mode == PLK_ROOT_UNIFORM) lh = (((A[j][0] + A[j][1]) + A[j][2]) + A[j][3]) * 0.25;
            else lh = fma(rootw[3], A[j][3], fma(rootw[2], A[j][2], fma(rootw[1], A[j][1], rootw[0] * A[j][0])));
            const double term = prior[c] * lh;
            if (term != 0.0) {
                if (!have[j]) { sum[j] = term; Eexp[j] = esc[j]; have[j] = true; }
                else if (esc[j] > Eexp[j]) { sum[j] = ldexp(sum[j], Eexp[j] - esc[j]) + term; Eexp[j] = esc[j]; }
                else sum[j] += ldexp(term, esc[j] - Eexp[j]);
            }
        }
    }
    dd v = dd_make(0.0, 0.0);
#pragma unroll
    for (int j = 0; j < NS; j++) {
        const long s = tile0 + j * PLK_TILE + tid;
        const double ll = have[j] ? log(sum[j]) + (double)Eexp[j] * 0.6931471805599453094 : -INFINITY;
        if (s < a.S) {
            if (a.site_ll) a.site_ll[s] = ll;
            v = dd_add(v, a.w ? dd_two_prod(a.w[s], ll) : dd_make(ll, 0.0));
        }
    }
    if (a.partial) {
        dd r = dd_block_sum(v);
        if (tid == 0) a.partial[blockIdx.x] = r;
    }
}
#undef PLK_LOAD_M
#undef PLK_FUSED_EXEC


#endif
