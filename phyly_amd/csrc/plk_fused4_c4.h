/*
 * plk_fused4_c4.h -- k_ll_fused4_cn<NC, D>: the k = 4 fused ll traversal with NC = 2 or 4 rate categories of a site
 * carried through one pass of the traversal program (interpreter in CDNA4 assembly: plk_fused4_c4_asm.h, generated
 * by tools/gen_fused4_c4.py, which also documents the register map).  Included by plk_engine.hip after
 * plk_fused4_asm.h.  Used when the number of categories is a multiple of NC (GTR+G4: BASELINE config 3) and the
 * stack fits (NC = 4: four slots; NC = 2: eight); everything else runs k_ll_fused4_asm, one category per pass.
 *
 * Replaces the same reference code as k_ll_fused4_asm: src/arbplfll.c:139-170 (site and category loops) x
 * src/evaluate_site_lhood.c:21-57 x src/util.c:242-301 x src/model.c:283-350.
 *
 * LDS: the tip tables of the NC categories of the group (NC x ntips x nchar x 32 bytes), then the staged code rows of
 * the tile.  Matrix stream: [group][matrix][category in group][16] (K1 writes it in this order).
 */
#ifndef PLK_FUSED4_C4_H
#define PLK_FUSED4_C4_H

#include "plk_fused4_c4_asm.h"

template <int NC, int D>
__global__ __launch_bounds__(PLK_TILE) void k_ll_fused4_cn(FusedAsmArgs aa)
{
    const FusedArgs &a = aa.f;
    extern __shared__ double lds_dyn[];
    double *tip_lds = lds_dyn;
    const int tip_doubles = a.ntips * a.nchar * 4;                 /* one category */
    uint8_t *code_lds = reinterpret_cast<uint8_t *>(lds_dyn + NC * (size_t)tip_doubles);
    const long tile0 = (long)blockIdx.x * PLK_TILE;
    const int tid = threadIdx.x;
    const long s = tile0 + tid;
    const int row_bytes = aa.pack4 ? PLK_TILE / 2 : PLK_TILE;
    {
        const int ndw = a.nobs * (PLK_TILE / 4);
        for (int idx = tid; idx < ndw; idx += PLK_TILE) {
            const int row = idx >> 6, col = idx & 63;
            const uint32_t *src = reinterpret_cast<const uint32_t *>(a.codes + (size_t)a.obs_nodes[row] * a.Spad + tile0);
            const uint32_t q = src[col];
            if (aa.pack4) {
                const uint32_t packed = (q & 0xf) | ((q >> 4) & 0xf0) | ((q >> 8) & 0xf00) | ((q >> 12) & 0xf000);
                reinterpret_cast<uint16_t *>(code_lds)[row * (PLK_TILE / 4) + col] = (uint16_t)packed;
            } else {
                reinterpret_cast<uint32_t *>(code_lds)[idx] = q;
            }
        }
    }
    const PLK_AS4 double *prior = as_uniform(a.cat_prior);
    const PLK_AS4 double *rootw = as_uniform(a.root_w);

    FusedC4Params p;
    const unsigned code_base = (unsigned)(size_t)code_lds;
    const unsigned tipbase = (unsigned)(size_t)tip_lds, nchar32 = (unsigned)a.nchar * 32u;
    p.clane = code_base + (aa.pack4 ? (unsigned)(tid >> 1) : (unsigned)tid);
    p.nshift = aa.pack4 ? (unsigned)(tid & 1) * 4u : 0u;
    p.secaddr = p.clane + (unsigned)aa.second_row * (unsigned)row_bytes;
    /* uniform parameters, one per lane of a single register (the interpreter reads them with v_readlane) */
    double rw[4];
#pragma unroll
    for (int i = 0; i < 4; i++) rw[i] = a.root_mode == PLK_ROOT_NONE ? 1.0 : (a.root_mode == PLK_ROOT_UNIFORM ? 0.25 : rootw[i]);
    const int lane = tid & 63;
    unsigned pv_fixed = 0;
    switch (lane) {
    case 0: pv_fixed = (unsigned)(size_t)aa.words; break;
    case 1: pv_fixed = (unsigned)((size_t)aa.words >> 32); break;
    case 4: pv_fixed = tipbase; break;
    case 5: pv_fixed = nchar32; break;
    case 6: pv_fixed = (unsigned)row_bytes; break;
    case 7: pv_fixed = aa.pack4 ? 4u : 8u; break;
    case 8: pv_fixed = (unsigned)tip_doubles * 8u; break;
    case 9: pv_fixed = tipbase + (unsigned)aa.first_tip * nchar32; break;
    case 10: pv_fixed = (unsigned)__double2loint(rw[0]); break;
    case 11: pv_fixed = (unsigned)__double2hiint(rw[0]); break;
    case 12: pv_fixed = (unsigned)__double2loint(rw[1]); break;
    case 13: pv_fixed = (unsigned)__double2hiint(rw[1]); break;
    case 14: pv_fixed = (unsigned)__double2loint(rw[2]); break;
    case 15: pv_fixed = (unsigned)__double2hiint(rw[2]); break;
    case 16: pv_fixed = (unsigned)__double2loint(rw[3]); break;
    case 17: pv_fixed = (unsigned)__double2hiint(rw[3]); break;
    default: break;
    }

    double sum = 0.0;
    int Eexp = 0;
    bool have = false;
    for (int g = 0; g < a.C; g += NC) {
        __syncthreads();
        {
            const double2 *src = reinterpret_cast<const double2 *>(a.tip + (size_t)g * tip_doubles);
            double2 *dst = reinterpret_cast<double2 *>(tip_lds);
            for (int idx = tid; idx < NC * tip_doubles / 2; idx += PLK_TILE) dst[idx] = src[idx];      /* NC categories */
        }
        __syncthreads();
        const double *ms = a.PS + (size_t)g * (a.nmat + 1) * 16;      /* group g / NC: (nmat + 1) x NC matrices */
        p.pv = lane == 2 ? (unsigned)(size_t)ms : (lane == 3 ? (unsigned)((size_t)ms >> 32) : pv_fixed);
        p.ch = code_lds[(p.clane - code_base) + aa.first_row * row_bytes];
        double lh[NC];
        int esc[NC];
        fused_run_program_cn<NC, D>(lh, esc, p);
#pragma unroll
        for (int cc = 0; cc < NC; cc++) {
            const double term = prior[g + cc] * lh[cc];
            if (term != 0.0) {
                if (!have) { sum = term; Eexp = esc[cc]; have = true; }
                else if (esc[cc] > Eexp) { sum = ldexp(sum, Eexp - esc[cc]) + term; Eexp = esc[cc]; }
                else sum += ldexp(term, esc[cc] - Eexp);
            }
        }
    }
    const double ll = have ? log(sum) + (double)Eexp * 0.6931471805599453094 : -INFINITY;
    dd v = dd_make(0.0, 0.0);
    if (s < a.S) {
        if (a.site_ll) a.site_ll[s] = ll;
        v = a.w ? dd_two_prod(a.w[s], ll) : dd_make(ll, 0.0);
    }
    if (a.partial) {
        dd r = dd_block_sum(v);
        if (tid == 0) a.partial[blockIdx.x] = r;
    }
}

#endif
