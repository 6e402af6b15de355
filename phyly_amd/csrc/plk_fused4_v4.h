/*
 * plk_fused4_v4.h -- k_ll_fused4_v4: the k = 4 pair-table interpreter with TWO sites per lane (round 3).
 * Included by plk_engine.hip after plk_fused4_asm.h (FusedPTArgs, fused_touch_lines).
 *
 * Same traversal program, tables and staged rows as k_ll_fused4_asm_pt (plk_fused4_asm.h); what changes is how the
 * interpreter spends its instruction slots.  Counters and timing experiments on the one-site interpreters
 * (profiles/r03_exp_headline_kernel_variants.json) showed a wave advancing by one instruction every ~12.5 cycles whatever
 * the instruction was, with the scalar bookkeeping of an op (15 scalar instructions: dispatch, field extraction, address
 * arithmetic) as long as its vector work (18), and neither the matrix loads nor the LDS reads on the critical path.
 * Hence:
 *   - a lane carries two sites (A = its own, B = THREADS sites further in the tile): every op is dispatched once for
 *     128 sites, and the two sites' vector instructions alternate in the issue stream (two independent dependency chains);
 *   - 64-bit op words whose fields are final: the low dword is the handler's byte offset, the high dword holds the
 *     LDS offsets of the next table and the code row after next already divided by their granule (2 scalar
 *     instructions instead of 5), built on the host by plk_fused_v4_words (plk_program.h);
 *   - threaded dispatch: no interpreter loop, every handler ends by fetching the next op from a ring of two op
 *     blocks in SGPRs (s_movrels_b64 indexed by M0) and jumping to its handler -- one taken jump per op instead of
 *     a call and a return; the last op of a block is a REFILL op that requests the block after next.  Handlers are
 *     512 bytes apart in a 32 KB aligned table.
 * The assembly text is generated (tools/gen_fused4_v4.py -> plk_fused4_v4_asm.h).
 * Reference semantics: src/evaluate_site_lhood.c:21-57, src/util.c:242-301, src/arbplfll.c:139-170.
 */
#ifndef PLK_FUSED4_V4_H
#define PLK_FUSED4_V4_H

#include "plk_fused4_v4_asm.h"

struct FusedV4Args {
    FusedPTArgs pt;             /* as for k_ll_fused4_asm_pt; words = 64-bit ops (pairs of dwords), nwords = dwords */
    unsigned first_y, first_z, second_z;     /* pre-scaled fields of the prefetch chain's start */
};

/* runs the program of one category for the lane's two sites: root dot products lh (w . root vector) and scale exponents */
#define PLK_V4_RUN(HALF)                                                                                             \
    asm volatile(PLK_V4_PROGRAM(HALF)                                                                                \
                 : [al] "=v"(al), [ah] "=v"(ah), [bl] "=v"(bl), [bh] "=v"(bh), [ea] "=v"(ea), [eb] "=v"(eb)          \
                 : [clane] "v"(clane), [ops] "s"(ops), [mstream] "s"(mstream), [y0] "s"(y0), [z0] "s"(z0), [z1] "s"(z1), \
                   [w0] "s"(w0), [w1] "s"(w1), [w2] "s"(w2), [w3] "s"(w3)                                            \
                 : PLK_V4_CLOBBERS)

template <int THREADS>
__device__ __forceinline__ void fused_run_program_v4(double &lha, double &lhb, int &ea, int &eb, const void *ops, const void *mstream,
                                                      unsigned clane, unsigned y0, unsigned z0, unsigned z1,
                                                      double w0, double w1, double w2, double w3)
{
    int al, ah, bl, bh;
#ifdef PLK_EXP_EMPTY
    al = bl = 0; ah = bh = 0x3ff00000; ea = eb = 0;
#else
    if constexpr (THREADS == 512) { PLK_V4_RUN(512); } else { static_assert(THREADS == 768, "tile"); PLK_V4_RUN(768); }
#endif
    lha = __hiloint2double(ah, al);
    lhb = __hiloint2double(bh, bl);
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_ll_fused4_v4(FusedV4Args va)
{
    constexpr int TILE = 2 * THREADS, NW = THREADS / 64, ND = TILE / 256;    /* sites per tile, waves, dwords of a row per lane */
    const FusedPTArgs &aa = va.pt;
    const FusedArgs &a = aa.f;
    extern __shared__ double lds_dyn[];
    double *tip_lds = lds_dyn;
    const int tip_doubles = a.ntips * a.nchar * 4;
    uint8_t *code_lds = reinterpret_cast<uint8_t *>(lds_dyn + tip_doubles);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const PLK_AS4 int *rown = as_uniform(aa.row_nodes);
    const PLK_AS4 double *prior = as_uniform(a.cat_prior);
    const PLK_AS4 double *rootw = as_uniform(a.root_w);
    const unsigned clane = (unsigned)(size_t)code_lds + (unsigned)tid;
    /* root weights of the dot product at the end of the program: 1 (no prior), 1/4 (uniform) or the given distribution */
    const bool plain = a.root_mode == PLK_ROOT_NONE || a.root_mode == PLK_ROOT_UNIFORM;
    const double wu = a.root_mode == PLK_ROOT_NONE ? 1.0 : 0.25;
    const double rw0 = plain ? wu : rootw[0], rw1 = plain ? wu : rootw[1], rw2 = plain ? wu : rootw[2], rw3 = plain ? wu : rootw[3];

    for (int tile = blockIdx.x; tile < aa.ntiles; tile += gridDim.x) {
        const long tile0 = (long)tile * TILE;
        __syncthreads();                 /* the previous tile's last category has been read */
        /* staged rows (one byte per site; a pair row holds code(b) * nchar + code(c)): wave w takes rows w, w + NW, ...,
         * a lane moves TILE / 64 codes of a row, four rows are requested before the first is stored */
        for (int r0 = wave; r0 < a.nobs; r0 += 4 * NW) {
            struct alignas(8) Chunk { unsigned d[ND]; };
            Chunk q[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int row = r0 + NW * u;
#pragma unroll
                for (int j = 0; j < ND; j++) q[u].d[j] = 0u;
                if (row < a.nobs) {
                    const int nb = rown[row], nc = rown[a.nobs + row];
                    q[u] = reinterpret_cast<const Chunk *>(a.codes + (size_t)nb * a.Spad + tile0)[lane];
                    if (nc >= 0) {
                        const Chunk q2 = reinterpret_cast<const Chunk *>(a.codes + (size_t)nc * a.Spad + tile0)[lane];
#pragma unroll
                        for (int j = 0; j < ND; j++) q[u].d[j] = q[u].d[j] * (unsigned)a.nchar + q2.d[j];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int row = r0 + NW * u;
                if (row < a.nobs) reinterpret_cast<Chunk *>(code_lds)[row * 64 + lane] = q[u];
            }
        }
        double sum[2] = {0.0, 0.0};
        int Eexp[2] = {0, 0};
        bool have[2] = {false, false};
        for (int c = 0; c < a.C; c++) {
            __syncthreads();
            if (aa.warm) {
                /* scalar-cache warm-up of the category's matrix stream and of the op words (see k_ll_fused4_asm_pt) */
                fused_touch_lines(a.PS + (size_t)c * (a.nmat + 1) * 16, (unsigned)(a.nmat + 1) * 128u, (unsigned)wave * 64u, NW * 64u);
                fused_touch_lines(aa.words, (unsigned)aa.nwords * 4u, (unsigned)wave * 64u, NW * 64u);
            }
            {
                const double2 *src = reinterpret_cast<const double2 *>(a.tip + (size_t)c * tip_doubles);
                double2 *dst = reinterpret_cast<double2 *>(tip_lds);
                const int n2 = tip_doubles / 2;
                for (int i0 = tid; i0 < n2; i0 += 4 * THREADS) {
                    double2 v[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) v[u] = i0 + u * THREADS < n2 ? src[i0 + u * THREADS] : double2{0.0, 0.0};
#pragma unroll
                    for (int u = 0; u < 4; u++) if (i0 + u * THREADS < n2) dst[i0 + u * THREADS] = v[u];
                }
            }
            __syncthreads();
            double lhs[2];
            int esc[2];
            fused_run_program_v4<THREADS>(lhs[0], lhs[1], esc[0], esc[1], aa.words, a.PS + (size_t)c * (a.nmat + 1) * 16, clane,
                                          va.first_y, va.first_z, va.second_z, rw0, rw1, rw2, rw3);
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const double lh = lhs[j];
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
        for (int j = 0; j < 2; j++) {
            const long s = tile0 + tid + j * THREADS;
            const double ll = have[j] ? log(sum[j]) + (double)Eexp[j] * 0.6931471805599453094 : -INFINITY;
            if (s < a.S) {
                if (a.site_ll) a.site_ll[s] = ll;
                v = dd_add(v, a.w ? dd_two_prod(a.w[s], ll) : dd_make(ll, 0.0));
            }
        }
        if (a.partial) {
            dd r = dd_block_sum(v);
            if (tid == 0) a.partial[tile] = r;
        }
    }
}

#endif
