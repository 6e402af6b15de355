/*
 * plk_fused4_asm.h -- the k = 4 fused traversal with its interpreter loop written in
 * CDNA4 assembly (one inline-asm statement per rate category).  Included by
 * plk_engine.hip after plk_fused4.h (same FusedArgs, same program format).
 *
 * Why assembly: the loop is a wave-uniform interpreter (scalar dispatch over ~7
 * opcodes) around short vector handlers.  hipcc keeps inserting loop-carried
 * v_mov copies of the 4-vector and rebuilding branch conditions (about 2x the
 * necessary VALU and SALU instructions, measured with SQ_INSTS_VALU/SALU); here
 * every handler works in place on fixed registers:
 *
 *   v[24:31]  x0..x3   the partial-likelihood vector under construction (one site per lane)
 *   v[32:39]  t0..t3   temporaries (matvec partial sums, LDS / AGPR reads)
 *   v40 address temp, v41 next pattern code, v42 scale exponent, v43 temp, v44 LDS address of this lane's code column
 *   a[0:63]   stack of up to 8 waiting vectors (8 AGPRs per slot)
 *   s[36:67]  current P matrix, transposed (s[36+2(4j+i)] = P[i][j]), used as FMA scalar operands
 *   s[68:71]  next op (fetched one ahead), s72..s75 current op x,y,z + opcode
 *   s[76:77]  program pointer, s[78:79] matrix stream pointer, s[82:83] definitions table
 *   s84 LDS address of the tip table, s85 nchar*32, s86 bytes per staged code row, s87 = -1022
 *
 * Handlers (opcodes of plk_engine.hip):
 *   MATVEC   x = P x            16 fp64 VALU, then the next matrix of the stream is requested
 *   TIP_SET / TIP_MUL           x (*)= tip[t][code]   4 ds_read_b64 + 4 VALU; prefetches the next code
 *   PUSH d / POPMUL d           x <-> a[8d..8d+7]     8 accvgpr moves (+ 4 VALU)
 *   SCALE    x *= 2^-e, esc += e with e from the largest high word (exact)
 *   NODE_MUL x *= defs[code]    (internal node carrying data; rare)
 */
#ifndef PLK_FUSED4_ASM_H
#define PLK_FUSED4_ASM_H

#define PLK_ASM_POP(D_, R0, R1, R2, R3, R4, R5, R6, R7)                              \
    ".Lpop" #D_ "_%=:\n\t"                                                           \
    "v_accvgpr_read_b32 v32, a" #R0 "\n\tv_accvgpr_read_b32 v33, a" #R1 "\n\t"       \
    "v_accvgpr_read_b32 v34, a" #R2 "\n\tv_accvgpr_read_b32 v35, a" #R3 "\n\t"       \
    "v_accvgpr_read_b32 v36, a" #R4 "\n\tv_accvgpr_read_b32 v37, a" #R5 "\n\t"       \
    "v_accvgpr_read_b32 v38, a" #R6 "\n\tv_accvgpr_read_b32 v39, a" #R7 "\n\t"       \
    "s_branch .Lpopmul_%=\n"

#define PLK_ASM_PUSH(D_, R0, R1, R2, R3, R4, R5, R6, R7)                             \
    ".Lpush" #D_ "_%=:\n\t"                                                          \
    "v_accvgpr_write_b32 a" #R0 ", v24\n\tv_accvgpr_write_b32 a" #R1 ", v25\n\t"     \
    "v_accvgpr_write_b32 a" #R2 ", v26\n\tv_accvgpr_write_b32 a" #R3 ", v27\n\t"     \
    "v_accvgpr_write_b32 a" #R4 ", v28\n\tv_accvgpr_write_b32 a" #R5 ", v29\n\t"     \
    "v_accvgpr_write_b32 a" #R6 ", v30\n\tv_accvgpr_write_b32 a" #R7 ", v31\n\t"     \
    "s_branch .Lnext_%=\n"

/* Runs the whole program of one category for this lane's site.
 * x[4]: in = ones, out = root vector.  ch: code of the first observation op.  esc: out, scale exponent. */
__device__ __forceinline__ void fused_run_program_asm(
    double &x0, double &x1, double &x2, double &x3, int &esc, int ch_first,
    const void *ops, const void *mstream, const void *defs,
    unsigned tip_lds_addr, unsigned nchar32, unsigned tile_bytes, unsigned code_lane_addr)
{
    int x0lo = __double2loint(x0), x0hi = __double2hiint(x0), x1lo = __double2loint(x1), x1hi = __double2hiint(x1);
    int x2lo = __double2loint(x2), x2hi = __double2hiint(x2), x3lo = __double2loint(x3), x3hi = __double2hiint(x3);
    asm volatile(
        /* ---- prologue: operands into the fixed registers ---- */
        "v_mov_b32 v24, %[x0lo]\n\tv_mov_b32 v25, %[x0hi]\n\t"
        "v_mov_b32 v26, %[x1lo]\n\tv_mov_b32 v27, %[x1hi]\n\t"
        "v_mov_b32 v28, %[x2lo]\n\tv_mov_b32 v29, %[x2hi]\n\t"
        "v_mov_b32 v30, %[x3lo]\n\tv_mov_b32 v31, %[x3hi]\n\t"
        "v_mov_b32 v41, %[ch]\n\t"
        "v_mov_b32 v42, 0\n\t"
        "v_mov_b32 v44, %[clane]\n\t"
        "s_mov_b64 s[76:77], %[ops]\n\t"
        "s_mov_b64 s[78:79], %[mstream]\n\t"
        "s_mov_b64 s[82:83], %[defs]\n\t"
        "s_mov_b32 s84, %[tipbase]\n\t"
        "s_mov_b32 s85, %[nchar32]\n\t"
        "s_mov_b32 s86, %[tile]\n\t"
        "s_movk_i32 s87, 0xfc02\n\t"
        "s_load_dwordx4 s[72:75], s[76:77], 0x0\n\t"
        "s_load_dwordx16 s[36:51], s[78:79], 0x0\n\t"
        "s_load_dwordx16 s[52:67], s[78:79], 0x40\n\t"
        "s_waitcnt lgkmcnt(0)\n"
        /* ---- dispatch ---- */
        ".Lloop_%=:\n\t"
        "s_load_dwordx4 s[68:71], s[76:77], 0x10\n\t"
        "s_add_u32 s76, s76, 16\n\t"
        "s_addc_u32 s77, s77, 0\n\t"
        "s_and_b32 s75, s72, 0xff\n\t"
        "s_cmp_eq_u32 s75, 2\n\t"
        "s_cbranch_scc1 .Lmatvec_%=\n\t"
        "s_cmp_lt_u32 s75, 2\n\t"
        "s_cbranch_scc1 .Ltip_%=\n\t"
        "s_cmp_eq_u32 s75, 4\n\t"
        "s_cbranch_scc1 .Lpop_%=\n\t"
        "s_cmp_eq_u32 s75, 3\n\t"
        "s_cbranch_scc1 .Lpush_%=\n\t"
        "s_cmp_eq_u32 s75, 6\n\t"
        "s_cbranch_scc1 .Lscale_%=\n\t"
        "s_cmp_eq_u32 s75, 5\n\t"
        "s_cbranch_scc1 .Lnode_%=\n\t"
        "s_branch .Ldone_%=\n"
        /* ---- MATVEC: x = M x in place ---- */
        ".Lmatvec_%=:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_mul_f64 v[32:33], s[36:37], v[24:25]\n\t"
        "v_mul_f64 v[34:35], s[38:39], v[24:25]\n\t"
        "v_mul_f64 v[36:37], s[40:41], v[24:25]\n\t"
        "v_mul_f64 v[38:39], s[42:43], v[24:25]\n\t"
        "v_fma_f64 v[32:33], s[44:45], v[26:27], v[32:33]\n\t"
        "v_fma_f64 v[34:35], s[46:47], v[26:27], v[34:35]\n\t"
        "v_fma_f64 v[36:37], s[48:49], v[26:27], v[36:37]\n\t"
        "v_fma_f64 v[38:39], s[50:51], v[26:27], v[38:39]\n\t"
        "v_fma_f64 v[32:33], s[52:53], v[28:29], v[32:33]\n\t"
        "v_fma_f64 v[34:35], s[54:55], v[28:29], v[34:35]\n\t"
        "v_fma_f64 v[36:37], s[56:57], v[28:29], v[36:37]\n\t"
        "v_fma_f64 v[38:39], s[58:59], v[28:29], v[38:39]\n\t"
        "v_fma_f64 v[24:25], s[60:61], v[30:31], v[32:33]\n\t"
        "v_fma_f64 v[26:27], s[62:63], v[30:31], v[34:35]\n\t"
        "v_fma_f64 v[28:29], s[64:65], v[30:31], v[36:37]\n\t"
        "v_fma_f64 v[30:31], s[66:67], v[30:31], v[38:39]\n\t"
        "s_add_u32 s78, s78, 0x80\n\t"
        "s_addc_u32 s79, s79, 0\n\t"
        "s_load_dwordx16 s[36:51], s[78:79], 0x0\n\t"
        "s_load_dwordx16 s[52:67], s[78:79], 0x40\n\t"
        "s_branch .Lnext_%=\n"
        /* ---- TIP_SET / TIP_MUL ---- */
        ".Ltip_%=:\n\t"
        "s_lshr_b32 s80, s72, 8\n\t"
        "s_mul_i32 s80, s80, s85\n\t"
        "s_add_u32 s80, s80, s84\n\t"
        "s_mul_i32 s81, s74, s86\n\t"
        "v_lshl_add_u32 v40, v41, 5, s80\n\t"
        "ds_read_b64 v[32:33], v40\n\t"
        "ds_read_b64 v[34:35], v40 offset:8\n\t"
        "ds_read_b64 v[36:37], v40 offset:16\n\t"
        "ds_read_b64 v[38:39], v40 offset:24\n\t"
        "v_add_u32 v43, s81, v44\n\t"
        "ds_read_u8 v41, v43\n\t"
        "s_cmp_eq_u32 s75, 0\n\t"
        "s_cbranch_scc1 .Ltipset_%=\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_mul_f64 v[24:25], v[24:25], v[32:33]\n\t"
        "v_mul_f64 v[26:27], v[26:27], v[34:35]\n\t"
        "v_mul_f64 v[28:29], v[28:29], v[36:37]\n\t"
        "v_mul_f64 v[30:31], v[30:31], v[38:39]\n\t"
        "s_branch .Lnext_%=\n"
        ".Ltipset_%=:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_mov_b64 v[24:25], v[32:33]\n\t"
        "v_mov_b64 v[26:27], v[34:35]\n\t"
        "v_mov_b64 v[28:29], v[36:37]\n\t"
        "v_mov_b64 v[30:31], v[38:39]\n\t"
        "s_branch .Lnext_%=\n"
        /* ---- POPMUL d ---- */
        ".Lpop_%=:\n\t"
        "s_cmp_eq_u32 s73, 0\n\ts_cbranch_scc1 .Lpop0_%=\n\t"
        "s_cmp_eq_u32 s73, 1\n\ts_cbranch_scc1 .Lpop1_%=\n\t"
        "s_cmp_eq_u32 s73, 2\n\ts_cbranch_scc1 .Lpop2_%=\n\t"
        "s_cmp_eq_u32 s73, 3\n\ts_cbranch_scc1 .Lpop3_%=\n\t"
        "s_cmp_eq_u32 s73, 4\n\ts_cbranch_scc1 .Lpop4_%=\n\t"
        "s_cmp_eq_u32 s73, 5\n\ts_cbranch_scc1 .Lpop5_%=\n\t"
        "s_cmp_eq_u32 s73, 6\n\ts_cbranch_scc1 .Lpop6_%=\n\t"
        "s_branch .Lpop7_%=\n"
        PLK_ASM_POP(0, 0, 1, 2, 3, 4, 5, 6, 7)
        PLK_ASM_POP(1, 8, 9, 10, 11, 12, 13, 14, 15)
        PLK_ASM_POP(2, 16, 17, 18, 19, 20, 21, 22, 23)
        PLK_ASM_POP(3, 24, 25, 26, 27, 28, 29, 30, 31)
        PLK_ASM_POP(4, 32, 33, 34, 35, 36, 37, 38, 39)
        PLK_ASM_POP(5, 40, 41, 42, 43, 44, 45, 46, 47)
        PLK_ASM_POP(6, 48, 49, 50, 51, 52, 53, 54, 55)
        PLK_ASM_POP(7, 56, 57, 58, 59, 60, 61, 62, 63)
        ".Lpopmul_%=:\n\t"
        "v_mul_f64 v[24:25], v[24:25], v[32:33]\n\t"
        "v_mul_f64 v[26:27], v[26:27], v[34:35]\n\t"
        "v_mul_f64 v[28:29], v[28:29], v[36:37]\n\t"
        "v_mul_f64 v[30:31], v[30:31], v[38:39]\n\t"
        "s_branch .Lnext_%=\n"
        /* ---- PUSH d ---- */
        ".Lpush_%=:\n\t"
        "s_cmp_eq_u32 s73, 0\n\ts_cbranch_scc1 .Lpush0_%=\n\t"
        "s_cmp_eq_u32 s73, 1\n\ts_cbranch_scc1 .Lpush1_%=\n\t"
        "s_cmp_eq_u32 s73, 2\n\ts_cbranch_scc1 .Lpush2_%=\n\t"
        "s_cmp_eq_u32 s73, 3\n\ts_cbranch_scc1 .Lpush3_%=\n\t"
        "s_cmp_eq_u32 s73, 4\n\ts_cbranch_scc1 .Lpush4_%=\n\t"
        "s_cmp_eq_u32 s73, 5\n\ts_cbranch_scc1 .Lpush5_%=\n\t"
        "s_cmp_eq_u32 s73, 6\n\ts_cbranch_scc1 .Lpush6_%=\n\t"
        "s_branch .Lpush7_%=\n"
        PLK_ASM_PUSH(0, 0, 1, 2, 3, 4, 5, 6, 7)
        PLK_ASM_PUSH(1, 8, 9, 10, 11, 12, 13, 14, 15)
        PLK_ASM_PUSH(2, 16, 17, 18, 19, 20, 21, 22, 23)
        PLK_ASM_PUSH(3, 24, 25, 26, 27, 28, 29, 30, 31)
        PLK_ASM_PUSH(4, 32, 33, 34, 35, 36, 37, 38, 39)
        PLK_ASM_PUSH(5, 40, 41, 42, 43, 44, 45, 46, 47)
        PLK_ASM_PUSH(6, 48, 49, 50, 51, 52, 53, 54, 55)
        PLK_ASM_PUSH(7, 56, 57, 58, 59, 60, 61, 62, 63)
        /* ---- SCALE: exact 2^-e, e = biased exponent of the largest entry - 1022 ---- */
        ".Lscale_%=:\n\t"
        "v_max_u32 v43, v25, v27\n\t"
        "v_max3_u32 v43, v29, v31, v43\n\t"
        "v_lshrrev_b32 v43, 20, v43\n\t"
        "v_sub_u32 v40, 0x3fe, v43\n\t"
        "v_ldexp_f64 v[24:25], v[24:25], v40\n\t"
        "v_ldexp_f64 v[26:27], v[26:27], v40\n\t"
        "v_ldexp_f64 v[28:29], v[28:29], v40\n\t"
        "v_ldexp_f64 v[30:31], v[30:31], v40\n\t"
        "v_add3_u32 v42, v42, v43, s87\n\t"
        "s_branch .Lnext_%=\n"
        /* ---- NODE_MUL: x *= defs[code] (definitions in global memory) ---- */
        ".Lnode_%=:\n\t"
        "v_lshlrev_b32 v40, 5, v41\n\t"
        "global_load_dwordx2 v[32:33], v40, s[82:83]\n\t"
        "global_load_dwordx2 v[34:35], v40, s[82:83] offset:8\n\t"
        "global_load_dwordx2 v[36:37], v40, s[82:83] offset:16\n\t"
        "global_load_dwordx2 v[38:39], v40, s[82:83] offset:24\n\t"
        "s_mul_i32 s81, s74, s86\n\t"
        "v_add_u32 v43, s81, v44\n\t"
        "ds_read_u8 v41, v43\n\t"
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
        "s_branch .Lpopmul_%=\n"
        /* ---- next op ---- */
        ".Lnext_%=:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_mov_b64 s[72:73], s[68:69]\n\t"
        "s_mov_b32 s74, s70\n\t"
        "s_branch .Lloop_%=\n"
        /* ---- epilogue ---- */
        ".Ldone_%=:\n\t"
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
        "v_mov_b32 %[x0lo], v24\n\tv_mov_b32 %[x0hi], v25\n\t"
        "v_mov_b32 %[x1lo], v26\n\tv_mov_b32 %[x1hi], v27\n\t"
        "v_mov_b32 %[x2lo], v28\n\tv_mov_b32 %[x2hi], v29\n\t"
        "v_mov_b32 %[x3lo], v30\n\tv_mov_b32 %[x3hi], v31\n\t"
        "v_mov_b32 %[esc], v42\n\t"
        "s_nop 1"
        : [x0lo] "+v"(x0lo), [x0hi] "+v"(x0hi), [x1lo] "+v"(x1lo), [x1hi] "+v"(x1hi),
          [x2lo] "+v"(x2lo), [x2hi] "+v"(x2hi), [x3lo] "+v"(x3lo), [x3hi] "+v"(x3hi), [esc] "=v"(esc)
        : [ch] "v"(ch_first), [clane] "v"(code_lane_addr), [ops] "s"(ops), [mstream] "s"(mstream), [defs] "s"(defs),
          [tipbase] "s"(tip_lds_addr), [nchar32] "s"(nchar32), [tile] "s"(tile_bytes)
        : "memory", "scc", "vcc",
          "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39",
          "v40", "v41", "v42", "v43", "v44",
          "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",
          "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67",
          "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83",
          "s84", "s85", "s86", "s87",
          PLK_CLOBBER_A0_31, PLK_CLOBBER_A32_63);
    x0 = __hiloint2double(x0hi, x0lo); x1 = __hiloint2double(x1hi, x1lo);
    x2 = __hiloint2double(x2hi, x2lo); x3 = __hiloint2double(x3hi, x3lo);
}

/* same staging, category loop and epilogue as k_ll_fused4<8,1>, with the program run in assembly */
__global__ __launch_bounds__(PLK_TILE) void k_ll_fused4_asm(FusedArgs a)
{
    extern __shared__ double lds_dyn[];
    double *tip_lds = lds_dyn;
    const int tip_doubles = a.ntips * a.nchar * 4;
    uint8_t *code_lds = reinterpret_cast<uint8_t *>(lds_dyn + tip_doubles);
    const long tile0 = (long)blockIdx.x * PLK_TILE;
    const int tid = threadIdx.x;
    const long s = tile0 + tid;
    {
        const int ndw = a.nobs * (PLK_TILE / 4);
        uint32_t *dst = reinterpret_cast<uint32_t *>(code_lds);
        for (int idx = tid; idx < ndw; idx += PLK_TILE) {
            int row = idx >> 6, col = idx & 63;
            const uint32_t *src = reinterpret_cast<const uint32_t *>(a.codes + (size_t)a.obs_nodes[row] * a.Spad + tile0);
            dst[idx] = src[col];
        }
    }
    const PLK_AS4 double *prior = as_uniform(a.cat_prior);
    const PLK_AS4 double *rootw = as_uniform(a.root_w);
    const unsigned tip_addr = (unsigned)(size_t)tip_lds;
    const unsigned lane_addr = (unsigned)(size_t)(code_lds + tid);

    double sum = 0.0;
    int Eexp = 0;
    bool have = false;
    for (int c = 0; c < a.C; c++) {
        __syncthreads();
        {
            const double2 *src = reinterpret_cast<const double2 *>(a.tip + (size_t)c * tip_doubles);
            double2 *dst = reinterpret_cast<double2 *>(tip_lds);
            for (int idx = tid; idx < tip_doubles / 2; idx += PLK_TILE) dst[idx] = src[idx];
        }
        __syncthreads();
        double x0 = 1.0, x1 = 1.0, x2 = 1.0, x3 = 1.0;
        int esc = 0;
        const int ch_first = code_lds[a.first_row * PLK_TILE + tid];
        fused_run_program_asm(x0, x1, x2, x3, esc, ch_first, a.ops, a.PS + (size_t)c * (a.nmat + 1) * 16, a.defs,
                              tip_addr, (unsigned)a.nchar * 32u, (unsigned)PLK_TILE, lane_addr);
        double lh;
        if (a.root_mode == PLK_ROOT_NONE) lh = ((x0 + x1) + x2) + x3;
        else if (a.root_mode == PLK_ROOT_UNIFORM) lh = (((x0 + x1) + x2) + x3) * 0.25;
        else lh = fma(rootw[3], x3, fma(rootw[2], x2, fma(rootw[1], x1, rootw[0] * x0)));
        const double term = prior[c] * lh;
        if (term != 0.0) {
            if (!have) { sum = term; Eexp = esc; have = true; }
            else if (esc > Eexp) { sum = ldexp(sum, Eexp - esc) + term; Eexp = esc; }
            else sum += ldexp(term, esc - Eexp);
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
