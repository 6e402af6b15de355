/*
 * plk_fused4_asm.h -- the k = 4 fused traversal with its interpreter written in CDNA4
 * assembly (one inline-asm statement per rate category).  Included by plk_engine.hip
 * after plk_fused4.h.
 *
 * Why assembly: the loop is a wave-uniform interpreter (scalar dispatch over 6 opcodes)
 * around short vector handlers.  hipcc inserts loop-carried v_mov copies of the 4-vector
 * and rebuilds branch conditions (about 2x the necessary VALU and SALU instructions,
 * measured with SQ_INSTS_VALU / SQ_INSTS_SALU); here every handler works in place on
 * fixed registers and nothing on the critical path waits for a load issued in the same op.
 *
 * Program format (built by prepare_stream_asm in plk_engine.hip): 32-bit op words in
 * blocks of 8, fetched a whole block ahead with one s_load_dwordx8;
 *   bits 4:0 handler index (0 TIP_SET, 1 TIP_MUL, 2 MATVEC, 3 TIP_MUL without wait, 4 MATVEC + TIP_MUL in one word (slots 4 and 5), 6 SCALE, 7 END, 8 + d PUSH to
 *   stack slot d, 16 + d POPMUL of slot d, 24 + d / 28 + d MATVEC followed by PUSH / POPMUL of slot d < 4 when the
 *   tree needs at most 4 slots), bits 15:5 field y, bits 31:16 field z
 *   MATVEC            x = P x; matrices are consumed in stream order, the next one is
 *                     requested as soon as the current one has been used
 *   TIP_SET / TIP_MUL x (*)= tip value of this observation, which was fetched from LDS
 *                     during the previous observation op; y = tip slot of the NEXT
 *                     observation op, z = code row of the one after next (prefetch chain)
 *   PUSH / POPMUL     the stack slot is part of the handler index
 *   SCALE             exact power-of-two rescale, exponent accumulated
 *   END
 * (an internal node with data is a TIP_MUL on the pseudo tip slot that holds the raw
 * character definitions)
 *
 * Registers:
 *   v[24:31] x0..x3  vector under construction      v[32:39] t0..t3 temporaries
 *   v40 address temp  v41 code of the next observation (raw)  v42 scale exponent  v43 temp
 *   v44 LDS address of this lane's code column   v45 nibble shift  v[46:53] prefetched tip value
 *   D = 4: v[54:85] stack of 4 waiting vectors (operands of the multiply that pops them);
 *   D = 8: a[0:63] stack of 8 waiting vectors (accumulation registers, moved through v[32:39])
 *   s[36:67] current P (transposed), FMA scalar operands
 *   s[68:75] current op block, s[76:83] next op block
 *   s[84:85] program pointer, s[86:87] matrix stream pointer, s[88:89] return address,
 *   s[90:91] address of handler 0 (handler i at + 256 i), s92 LDS address of the tip table, s93 nchar*32,
 *   s94 bytes per staged code row, s95 = -1022, s96 current op word, s97..s99 temps,
 *   s35 code field width (8, or 4 when codes are packed two per byte)
 */
#ifndef PLK_FUSED4_ASM_H
#define PLK_FUSED4_ASM_H

/* Dispatch: every handler starts at a 256-byte boundary, in the order of its 5-bit handler index (the low bits of the
 * op word), so the address of the handler is base + (index << 8): an op costs s_swappc into the handler and s_setpc
 * back -- no dispatcher with its compare-and-branch chain (one more taken branch per op, up to five compares for a
 * stack op), and PUSH / POPMUL have one handler per stack slot instead of a second chain. */
#define PLK_H_ALIGN ".p2align 8\n"
#define PLK_ASM_CALL(SREG)                                                            \
    "s_and_b32 s97, " #SREG ", 31\n\t"                                                \
    "s_lshl_b32 s97, s97, 8\n\t"                                                      \
    "s_or_b32 s98, s90, s97\n\t"          /* handler 0 sits on an 8 KB boundary; s99 = high half, set once */ \
    "s_mov_b32 s96, " #SREG "\n\t"                                                    \
    "s_swappc_b64 s[88:89], s[98:99]\n\t"
#define PLK_H_UNUSED PLK_H_ALIGN "s_setpc_b64 s[88:89]\n"

/* D <= 4: the stack lives in ordinary VGPRs v[54:85] (slot d = v[54 + 8d : 61 + 8d]).  A waiting vector is then an
 * operand of the multiply that pops it: PUSH = 4 moves, POPMUL = 4 multiplies, against 8 + 8 accumulation-register
 * moves + 4 multiplies with the AGPR stack -- 12 of 20 vector instructions per push / pop pair, ~12 % of all vector
 * instructions of the kernel at BASELINE config 3 (the kernel is bound by vector issue, not by registers: 88 VGPRs
 * still give 5 waves per SIMD). */
#define PLK_ASM_VPOP(A0, A1, B0, B1, C0, C1, E0, E1)                                      \
    PLK_H_ALIGN                                                                           \
    "v_mul_f64 v[24:25], v[24:25], v[" #A0 ":" #A1 "]\n\t"                                \
    "v_mul_f64 v[26:27], v[26:27], v[" #B0 ":" #B1 "]\n\t"                                \
    "v_mul_f64 v[28:29], v[28:29], v[" #C0 ":" #C1 "]\n\t"                                \
    "v_mul_f64 v[30:31], v[30:31], v[" #E0 ":" #E1 "]\n\t"                                \
    "s_setpc_b64 s[88:89]\n"
#define PLK_ASM_VPUSH(A0, A1, B0, B1, C0, C1, E0, E1)                                     \
    PLK_H_ALIGN                                                                           \
    "v_mov_b64 v[" #A0 ":" #A1 "], v[24:25]\n\t"                                          \
    "v_mov_b64 v[" #B0 ":" #B1 "], v[26:27]\n\t"                                          \
    "v_mov_b64 v[" #C0 ":" #C1 "], v[28:29]\n\t"                                          \
    "v_mov_b64 v[" #E0 ":" #E1 "], v[30:31]\n\t"                                          \
    "s_setpc_b64 s[88:89]\n"
/* handler indices 8..15 (PUSH slot 0..7) and 16..23 (POPMUL slot 0..7) */
#define PLK_ASM_SLOTS_D4_PUSH                                                                             \
    PLK_ASM_VPUSH(54, 55, 56, 57, 58, 59, 60, 61) PLK_ASM_VPUSH(62, 63, 64, 65, 66, 67, 68, 69)             \
    PLK_ASM_VPUSH(70, 71, 72, 73, 74, 75, 76, 77) PLK_ASM_VPUSH(78, 79, 80, 81, 82, 83, 84, 85)             \
    PLK_H_UNUSED PLK_H_UNUSED PLK_H_UNUSED PLK_H_UNUSED
#define PLK_ASM_SLOTS_D4_POP                                                                              \
    PLK_ASM_VPOP(54, 55, 56, 57, 58, 59, 60, 61) PLK_ASM_VPOP(62, 63, 64, 65, 66, 67, 68, 69)               \
    PLK_ASM_VPOP(70, 71, 72, 73, 74, 75, 76, 77) PLK_ASM_VPOP(78, 79, 80, 81, 82, 83, 84, 85)               \
    PLK_H_UNUSED PLK_H_UNUSED PLK_H_UNUSED PLK_H_UNUSED
/* handler indices 24..27: MATVEC + PUSH slot d, 28..31: MATVEC + POPMUL slot d (d < 4), one op word each */
#define PLK_ASM_MV_VPUSH(A0, A1, B0, B1, C0, C1, E0, E1)                                  \
    PLK_H_ALIGN PLK_ASM_MATVEC_CORE                                                       \
    "v_mov_b64 v[" #A0 ":" #A1 "], v[24:25]\n\t"                                          \
    "v_mov_b64 v[" #B0 ":" #B1 "], v[26:27]\n\t"                                          \
    "v_mov_b64 v[" #C0 ":" #C1 "], v[28:29]\n\t"                                          \
    "v_mov_b64 v[" #E0 ":" #E1 "], v[30:31]\n\t"                                          \
    "s_setpc_b64 s[88:89]\n"
/* MATVEC + PUSH with the product written straight into the stack slot: the last column's FMAs take the slot's
 * registers as destination, no moves (the vector under construction is dead after a PUSH: the next subtree starts
 * with a TIP_SET) */
#define PLK_ASM_MV_VPUSH_D(A0, A1, B0, B1, C0, C1, E0, E1)                                \
    PLK_H_ALIGN                                                                           \
        "s_waitcnt lgkmcnt(0)\n\t"                                                    \
        "v_mul_f64 v[32:33], s[36:37], v[24:25]\n\t"                                  \
        "v_mul_f64 v[34:35], s[38:39], v[24:25]\n\t"                                  \
        "v_mul_f64 v[36:37], s[40:41], v[24:25]\n\t"                                  \
        "v_mul_f64 v[38:39], s[42:43], v[24:25]\n\t"                                  \
        "v_fma_f64 v[32:33], s[44:45], v[26:27], v[32:33]\n\t"                        \
        "v_fma_f64 v[34:35], s[46:47], v[26:27], v[34:35]\n\t"                        \
        "v_fma_f64 v[36:37], s[48:49], v[26:27], v[36:37]\n\t"                        \
        "v_fma_f64 v[38:39], s[50:51], v[26:27], v[38:39]\n\t"                        \
        "v_fma_f64 v[32:33], s[52:53], v[28:29], v[32:33]\n\t"                        \
        "v_fma_f64 v[34:35], s[54:55], v[28:29], v[34:35]\n\t"                        \
        "v_fma_f64 v[36:37], s[56:57], v[28:29], v[36:37]\n\t"                        \
        "v_fma_f64 v[38:39], s[58:59], v[28:29], v[38:39]\n\t"                        \
        "v_fma_f64 v[" #A0 ":" #A1 "], s[60:61], v[30:31], v[32:33]\n\t"              \
        "v_fma_f64 v[" #B0 ":" #B1 "], s[62:63], v[30:31], v[34:35]\n\t"              \
        "v_fma_f64 v[" #C0 ":" #C1 "], s[64:65], v[30:31], v[36:37]\n\t"              \
        "v_fma_f64 v[" #E0 ":" #E1 "], s[66:67], v[30:31], v[38:39]\n\t"              \
        PLK_ASM_NEXT_MATRIX                                \
    "s_setpc_b64 s[88:89]\n"
#define PLK_ASM_MV_VPOP(A0, A1, B0, B1, C0, C1, E0, E1)                                   \
    PLK_H_ALIGN PLK_ASM_MATVEC_CORE                                                       \
    "v_mul_f64 v[24:25], v[24:25], v[" #A0 ":" #A1 "]\n\t"                                \
    "v_mul_f64 v[26:27], v[26:27], v[" #B0 ":" #B1 "]\n\t"                                \
    "v_mul_f64 v[28:29], v[28:29], v[" #C0 ":" #C1 "]\n\t"                                \
    "v_mul_f64 v[30:31], v[30:31], v[" #E0 ":" #E1 "]\n\t"                                \
    "s_setpc_b64 s[88:89]\n"
#define PLK_ASM_SLOTS_D4_PAIRS_D                                                                          \
    PLK_ASM_MV_VPUSH_D(54, 55, 56, 57, 58, 59, 60, 61) PLK_ASM_MV_VPUSH_D(62, 63, 64, 65, 66, 67, 68, 69)   \
    PLK_ASM_MV_VPUSH_D(70, 71, 72, 73, 74, 75, 76, 77) PLK_ASM_MV_VPUSH_D(78, 79, 80, 81, 82, 83, 84, 85)   \
    PLK_ASM_MV_VPOP(54, 55, 56, 57, 58, 59, 60, 61) PLK_ASM_MV_VPOP(62, 63, 64, 65, 66, 67, 68, 69)         \
    PLK_ASM_MV_VPOP(70, 71, 72, 73, 74, 75, 76, 77) PLK_ASM_MV_VPOP(78, 79, 80, 81, 82, 83, 84, 85)
#define PLK_ASM_SLOTS_D4_PAIRS                                                                            \
    PLK_ASM_MV_VPUSH(54, 55, 56, 57, 58, 59, 60, 61) PLK_ASM_MV_VPUSH(62, 63, 64, 65, 66, 67, 68, 69)       \
    PLK_ASM_MV_VPUSH(70, 71, 72, 73, 74, 75, 76, 77) PLK_ASM_MV_VPUSH(78, 79, 80, 81, 82, 83, 84, 85)       \
    PLK_ASM_MV_VPOP(54, 55, 56, 57, 58, 59, 60, 61) PLK_ASM_MV_VPOP(62, 63, 64, 65, 66, 67, 68, 69)         \
    PLK_ASM_MV_VPOP(70, 71, 72, 73, 74, 75, 76, 77) PLK_ASM_MV_VPOP(78, 79, 80, 81, 82, 83, 84, 85)
#define PLK_ASM_SLOTS_D8_PAIRS  /* the deeper stack's programs do not use the pair words */ \
    PLK_H_UNUSED PLK_H_UNUSED PLK_H_UNUSED PLK_H_UNUSED PLK_H_UNUSED PLK_H_UNUSED PLK_H_UNUSED PLK_H_UNUSED
#define PLK_CLOBBER_V54_85 "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", \
    "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85"

/* D = 8: the stack lives in the accumulation registers a[0:63], moved through v[32:39] */
#define PLK_ASM_POP(R0, R1, R2, R3, R4, R5, R6, R7)                                  \
    PLK_H_ALIGN                                                                      \
    "v_accvgpr_read_b32 v32, a" #R0 "\n\tv_accvgpr_read_b32 v33, a" #R1 "\n\t"       \
    "v_accvgpr_read_b32 v34, a" #R2 "\n\tv_accvgpr_read_b32 v35, a" #R3 "\n\t"       \
    "v_accvgpr_read_b32 v36, a" #R4 "\n\tv_accvgpr_read_b32 v37, a" #R5 "\n\t"       \
    "v_accvgpr_read_b32 v38, a" #R6 "\n\tv_accvgpr_read_b32 v39, a" #R7 "\n\t"       \
    "v_mul_f64 v[24:25], v[24:25], v[32:33]\n\t"                                     \
    "v_mul_f64 v[26:27], v[26:27], v[34:35]\n\t"                                     \
    "v_mul_f64 v[28:29], v[28:29], v[36:37]\n\t"                                     \
    "v_mul_f64 v[30:31], v[30:31], v[38:39]\n\t"                                     \
    "s_setpc_b64 s[88:89]\n"
#define PLK_ASM_PUSH(R0, R1, R2, R3, R4, R5, R6, R7)                                 \
    PLK_H_ALIGN                                                                      \
    "v_accvgpr_write_b32 a" #R0 ", v24\n\tv_accvgpr_write_b32 a" #R1 ", v25\n\t"     \
    "v_accvgpr_write_b32 a" #R2 ", v26\n\tv_accvgpr_write_b32 a" #R3 ", v27\n\t"     \
    "v_accvgpr_write_b32 a" #R4 ", v28\n\tv_accvgpr_write_b32 a" #R5 ", v29\n\t"     \
    "v_accvgpr_write_b32 a" #R6 ", v30\n\tv_accvgpr_write_b32 a" #R7 ", v31\n\t"     \
    "s_setpc_b64 s[88:89]\n"
#define PLK_ASM_SLOTS_D8_PUSH                                                         \
    PLK_ASM_PUSH(0, 1, 2, 3, 4, 5, 6, 7) PLK_ASM_PUSH(8, 9, 10, 11, 12, 13, 14, 15)           \
    PLK_ASM_PUSH(16, 17, 18, 19, 20, 21, 22, 23) PLK_ASM_PUSH(24, 25, 26, 27, 28, 29, 30, 31) \
    PLK_ASM_PUSH(32, 33, 34, 35, 36, 37, 38, 39) PLK_ASM_PUSH(40, 41, 42, 43, 44, 45, 46, 47) \
    PLK_ASM_PUSH(48, 49, 50, 51, 52, 53, 54, 55) PLK_ASM_PUSH(56, 57, 58, 59, 60, 61, 62, 63)
#define PLK_ASM_SLOTS_D8_POP                                                          \
    PLK_ASM_POP(0, 1, 2, 3, 4, 5, 6, 7) PLK_ASM_POP(8, 9, 10, 11, 12, 13, 14, 15)             \
    PLK_ASM_POP(16, 17, 18, 19, 20, 21, 22, 23) PLK_ASM_POP(24, 25, 26, 27, 28, 29, 30, 31)   \
    PLK_ASM_POP(32, 33, 34, 35, 36, 37, 38, 39) PLK_ASM_POP(40, 41, 42, 43, 44, 45, 46, 47)   \
    PLK_ASM_POP(48, 49, 50, 51, 52, 53, 54, 55) PLK_ASM_POP(56, 57, 58, 59, 60, 61, 62, 63)

/* (timing experiments only, never in the product build: -DPLK_EXP_NOMAT keeps the first matrix for every product,
 * -DPLK_EXP_NOLDS drops the tip-value and code reads; results are wrong, the instruction stream is otherwise the same) */
#ifdef PLK_EXP_NOMAT
#define PLK_ASM_NEXT_MATRIX "s_add_u32 s86, s86, 0x80\n\t" "s_addc_u32 s87, s87, 0\n\t"
#else
#define PLK_ASM_NEXT_MATRIX                                                           \
        "s_add_u32 s86, s86, 0x80\n\t"                                                \
        "s_addc_u32 s87, s87, 0\n\t"                                                  \
        "s_load_dwordx16 s[36:51], s[86:87], 0x0\n\t"                                 \
        "s_load_dwordx16 s[52:67], s[86:87], 0x40\n\t"
#endif
#ifdef PLK_EXP_NOLDS
#define PLK_ASM_TIPREADS(A)
#else
#define PLK_ASM_TIPREADS(A) A
#endif
/* x = M x in place with the current matrix of the stream, then request the next one */
#define PLK_ASM_MATVEC_CORE                                                           \
        "s_waitcnt lgkmcnt(0)\n\t"                                                    \
        "v_mul_f64 v[32:33], s[36:37], v[24:25]\n\t"                                  \
        "v_mul_f64 v[34:35], s[38:39], v[24:25]\n\t"                                  \
        "v_mul_f64 v[36:37], s[40:41], v[24:25]\n\t"                                  \
        "v_mul_f64 v[38:39], s[42:43], v[24:25]\n\t"                                  \
        "v_fma_f64 v[32:33], s[44:45], v[26:27], v[32:33]\n\t"                        \
        "v_fma_f64 v[34:35], s[46:47], v[26:27], v[34:35]\n\t"                        \
        "v_fma_f64 v[36:37], s[48:49], v[26:27], v[36:37]\n\t"                        \
        "v_fma_f64 v[38:39], s[50:51], v[26:27], v[38:39]\n\t"                        \
        "v_fma_f64 v[32:33], s[52:53], v[28:29], v[32:33]\n\t"                        \
        "v_fma_f64 v[34:35], s[54:55], v[28:29], v[34:35]\n\t"                        \
        "v_fma_f64 v[36:37], s[56:57], v[28:29], v[36:37]\n\t"                        \
        "v_fma_f64 v[38:39], s[58:59], v[28:29], v[38:39]\n\t"                        \
        "v_fma_f64 v[24:25], s[60:61], v[30:31], v[32:33]\n\t"                        \
        "v_fma_f64 v[26:27], s[62:63], v[30:31], v[34:35]\n\t"                        \
        "v_fma_f64 v[28:29], s[64:65], v[30:31], v[36:37]\n\t"                        \
        "v_fma_f64 v[30:31], s[66:67], v[30:31], v[38:39]\n\t"                        \
        PLK_ASM_NEXT_MATRIX

/* the tail of an observation handler: request the value of the next observation op and the code of the one after */
#define PLK_ASM_TIPNEXT                                                               \
        "s_bfe_u32 s98, s96, 0xb0005\n\t"                                             \
        "s_mul_i32 s98, s98, s93\n\t"                                                 \
        "s_add_u32 s98, s98, s92\n\t"                                                 \
        "s_lshr_b32 s97, s96, 16\n\t"                                                 \
        "s_mul_i32 s97, s97, s94\n\t"                                                 \
        "v_bfe_u32 v43, v41, v45, s35\n\t"                                           \
        "v_lshl_add_u32 v40, v43, 5, s98\n\t"                                         \
        "ds_read_b128 v[46:49], v40\n\t"                                              \
        "ds_read_b128 v[50:53], v40 offset:16\n\t"                                    \
        "v_add_u32 v43, s97, v44\n\t"                                                 \
        "ds_read_u8 v41, v43\n\t"                                                     \
        "s_setpc_b64 s[88:89]\n"
/* the same for the pair-table kernel (V = 1): every staged row holds one byte per site (a pattern code, or the combined
 * code of a two-leaf subtree), so the code is the table index as it comes from LDS: one integer instruction less */
#ifdef PLK_EXP_NOTIPSALU
#define PLK_ASM_TIPSALU(A)
#else
#define PLK_ASM_TIPSALU(A) A
#endif
#define PLK_ASM_TIPNEXT_B                                                             \
        PLK_ASM_TIPSALU("s_bfe_u32 s98, s96, 0xb0005\n\t")                            \
        PLK_ASM_TIPSALU("s_mul_i32 s98, s98, s93\n\t")                                \
        PLK_ASM_TIPSALU("s_add_u32 s98, s98, s92\n\t")                                \
        PLK_ASM_TIPSALU("s_lshr_b32 s97, s96, 16\n\t")                                \
        PLK_ASM_TIPSALU("s_mul_i32 s97, s97, s94\n\t")                                \
        "v_lshl_add_u32 v40, v41, 5, s98\n\t"                                         \
        PLK_ASM_TIPREADS("ds_read_b128 v[46:49], v40\n\t")                            \
        PLK_ASM_TIPREADS("ds_read_b128 v[50:53], v40 offset:16\n\t")                  \
        "v_add_u32 v43, s97, v44\n\t"                                                 \
        PLK_ASM_TIPREADS("ds_read_u8 v41, v43\n\t")                                   \
        "s_setpc_b64 s[88:89]\n"
#define PLK_ASM_TIPMUL                                                                \
        "v_mul_f64 v[24:25], v[24:25], v[46:47]\n\t"                                  \
        "v_mul_f64 v[26:27], v[26:27], v[48:49]\n\t"                                  \
        "v_mul_f64 v[28:29], v[28:29], v[50:51]\n\t"                                  \
        "v_mul_f64 v[30:31], v[30:31], v[52:53]\n\t"

#define PLK_ASM_PROGRAM(POP_SLOTS, PUSH_SLOTS, PAIR_SLOTS, TIPNEXT)                                        \
        /* ---- prologue: operands into the fixed registers ---- */                   \
        "v_mov_b32 v24, %[x0lo]\n\tv_mov_b32 v25, %[x0hi]\n\t"                        \
        "v_mov_b32 v26, %[x1lo]\n\tv_mov_b32 v27, %[x1hi]\n\t"                        \
        "v_mov_b32 v28, %[x2lo]\n\tv_mov_b32 v29, %[x2hi]\n\t"                        \
        "v_mov_b32 v30, %[x3lo]\n\tv_mov_b32 v31, %[x3hi]\n\t"                        \
        "v_mov_b32 v42, 0\n\t"                                                        \
        "v_mov_b32 v44, %[clane]\n\t"                                                 \
        "v_mov_b32 v45, %[nshift]\n\t"                                                \
        "s_mov_b64 s[84:85], %[ops]\n\t"                                              \
        "s_mov_b64 s[86:87], %[mstream]\n\t"                                          \
        "s_mov_b32 s92, %[tipbase]\n\t"                                               \
        "s_mov_b32 s93, %[nchar32]\n\t"                                               \
        "s_mov_b32 s94, %[tile]\n\t"                                                  \
        "s_mov_b32 s35, %[cwidth]\n\t"                                               \
        "s_movk_i32 s95, 0xfc02\n\t"                                                  \
        "s_load_dwordx8 s[68:75], s[84:85], 0x0\n\t"                                  \
        "s_load_dwordx16 s[36:51], s[86:87], 0x0\n\t"                                 \
        "s_load_dwordx16 s[52:67], s[86:87], 0x40\n\t"                                \
        /* prefetch chain start: tip value of the first observation, code of the second */ \
        "v_bfe_u32 v43, %[ch], v45, s35\n\t"                                         \
        "v_lshl_add_u32 v40, v43, 5, %[firsttip]\n\t"                                 \
        "ds_read_b128 v[46:49], v40\n\t"                                              \
        "ds_read_b128 v[50:53], v40 offset:16\n\t"                                    \
        "ds_read_u8 v41, %[secaddr]\n\t"                                              \
        "s_getpc_b64 s[90:91]\n"                                                      \
        ".Lpcref_%=:\n\t"                                                             \
        "s_add_u32 s90, s90, .Lh0_%=-.Lpcref_%=\n\t"                                  \
        "s_addc_u32 s91, s91, 0\n\t"                                                  \
        "s_mov_b32 s99, s91\n\t"                                                      \
        "s_waitcnt lgkmcnt(0)\n"                                                      \
        /* ---- one block of 8 ops per iteration; the next block is already in flight ---- */ \
        ".Lblock_%=:\n\t"                                                             \
        "s_load_dwordx8 s[76:83], s[84:85], 0x20\n\t"                                 \
        "s_add_u32 s84, s84, 32\n\t"                                                  \
        "s_addc_u32 s85, s85, 0\n\t"                                                  \
        PLK_ASM_CALL(s68) PLK_ASM_CALL(s69) PLK_ASM_CALL(s70) PLK_ASM_CALL(s71)       \
        PLK_ASM_CALL(s72) PLK_ASM_CALL(s73) PLK_ASM_CALL(s74) PLK_ASM_CALL(s75)       \
        "s_waitcnt lgkmcnt(0)\n\t"                                                    \
        "s_mov_b64 s[68:69], s[76:77]\n\t"                                            \
        "s_mov_b64 s[70:71], s[78:79]\n\t"                                            \
        "s_mov_b64 s[72:73], s[80:81]\n\t"                                            \
        "s_mov_b64 s[74:75], s[82:83]\n\t"                                            \
        "s_branch .Lblock_%=\n"                                                       \
        /* ==== handlers, 256 bytes apart, in handler-index order; s96 = op word ==== */ \
        /* ---- 0 TIP_SET, 1 TIP_MUL: consume the prefetched value, start the next fetches ---- */ \
        ".p2align 13\n"                                                               \
        ".Lh0_%=:\n\t"                                                                \
        "s_waitcnt lgkmcnt(0)\n\t"                                                    \
        "v_mov_b64 v[24:25], v[46:47]\n\t"                                            \
        "v_mov_b64 v[26:27], v[48:49]\n\t"                                            \
        "v_mov_b64 v[28:29], v[50:51]\n\t"                                            \
        "v_mov_b64 v[30:31], v[52:53]\n\t"                                            \
        TIPNEXT                                                                       \
        PLK_H_ALIGN                                                                   \
        "s_waitcnt lgkmcnt(0)\n\t"                                                    \
        PLK_ASM_TIPMUL                                                                \
        TIPNEXT                                                                       \
        /* ---- 2 MATVEC: x = M x in place ---- */                                    \
        PLK_H_ALIGN                                                                   \
        PLK_ASM_MATVEC_CORE                                                           \
        "s_setpc_b64 s[88:89]\n"                                                      \
        /* ---- 3 = TIP_MUL whose prefetched value is known to have landed: a MATVEC (which starts with a full \
         * wait) ran since the value was requested, so no wait is needed here and the matrix load that MATVEC   \
         * left in flight stays in flight ---- */                                     \
        PLK_H_ALIGN                                                                   \
        PLK_ASM_TIPMUL                                                                \
        TIPNEXT                                                                       \
        /* ---- 4 (+ the space of 5) MATVEC followed by TIP_MUL: one op word, one dispatch.  The product's wait covers
         * the prefetched tip value, and the word carries the observation's fields.  The handler starts on a 512-byte
         * boundary (slot 4) and may be up to 512 bytes long: the next handler is aligned to 512 bytes, so it is slot 6's
         * whatever the length of this body ---- */ \
        PLK_H_ALIGN                                                                   \
        PLK_ASM_MATVEC_CORE                                                           \
        PLK_ASM_TIPMUL                                                                \
        TIPNEXT                                                                       \
        ".p2align 9\n"                                                                \
        /* ---- 6 SCALE: exact 2^-e, e = biased exponent of the largest entry - 1022 ---- */ \
        "v_max_u32 v43, v25, v27\n\t"                                                 \
        "v_max3_u32 v43, v29, v31, v43\n\t"                                           \
        "v_lshrrev_b32 v43, 20, v43\n\t"                                              \
        "v_sub_u32 v40, 0x3fe, v43\n\t"                                               \
        "v_ldexp_f64 v[24:25], v[24:25], v40\n\t"                                     \
        "v_ldexp_f64 v[26:27], v[26:27], v40\n\t"                                     \
        "v_ldexp_f64 v[28:29], v[28:29], v40\n\t"                                     \
        "v_ldexp_f64 v[30:31], v[30:31], v40\n\t"                                     \
        "v_add3_u32 v42, v42, v43, s95\n\t"                                           \
        "s_setpc_b64 s[88:89]\n"                                                      \
        /* ---- 7 END ---- */                                                         \
        PLK_H_ALIGN                                                                   \
        "s_branch .Ldone_%=\n"                                                        \
        /* ---- 8..15 PUSH slot d, 16..23 POPMUL slot d ---- */                       \
        PUSH_SLOTS                                                                    \
        POP_SLOTS                                                                     \
        /* ---- 24..31 MATVEC + PUSH / POPMUL of slots 0..3 ---- */                   \
        PAIR_SLOTS                                                                    \
        /* ---- epilogue ---- */                                                      \
        PLK_H_ALIGN                                                                   \
        ".Ldone_%=:\n\t"                                                              \
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"                                           \
        "v_mov_b32 %[x0lo], v24\n\tv_mov_b32 %[x0hi], v25\n\t"                        \
        "v_mov_b32 %[x1lo], v26\n\tv_mov_b32 %[x1hi], v27\n\t"                        \
        "v_mov_b32 %[x2lo], v28\n\tv_mov_b32 %[x2hi], v29\n\t"                        \
        "v_mov_b32 %[x3lo], v30\n\tv_mov_b32 %[x3hi], v31\n\t"                        \
        "v_mov_b32 %[esc], v42\n\t"                                                   \
        "s_nop 1"

#define PLK_ASM_OPERANDS                                                              \
        : [x0lo] "+v"(x0lo), [x0hi] "+v"(x0hi), [x1lo] "+v"(x1lo), [x1hi] "+v"(x1hi), \
          [x2lo] "+v"(x2lo), [x2hi] "+v"(x2hi), [x3lo] "+v"(x3lo), [x3hi] "+v"(x3hi), [esc] "=v"(esc) \
        : [ch] "v"(p.ch_first), [clane] "v"(p.code_lane_addr), [nshift] "v"(p.nibble_shift),          \
          [secaddr] "v"(p.second_code_addr), [ops] "s"(p.ops), [mstream] "s"(p.mstream),              \
          [tipbase] "s"(p.tip_lds_addr), [nchar32] "s"(p.nchar32), [tile] "s"(p.row_bytes),           \
          [cwidth] "s"(p.code_width), [firsttip] "s"(p.first_tip_addr)

#define PLK_ASM_CLOBBERS_COMMON                                                       \
          "memory", "scc", "vcc",                                                     \
          "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", \
          "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53",               \
          "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", \
          "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", \
          "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", \
          "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "s35"

struct FusedAsmParams {
    const void *ops;            /* this program: op words in blocks of 8 */
    const void *mstream;        /* matrix stream of the category */
    unsigned tip_lds_addr;      /* LDS byte address of the tip table */
    unsigned nchar32;           /* bytes per tip slot = nchar * 32 */
    unsigned row_bytes;         /* bytes per staged code row */
    unsigned code_width;        /* 8, or 4 when two codes share a byte */
    unsigned first_tip_addr;    /* tip_lds_addr + first tip slot * nchar32 */
    unsigned code_lane_addr;    /* LDS byte address of this lane's byte in row 0 */
    unsigned nibble_shift;      /* 0, or 4 for odd lanes when packed */
    unsigned second_code_addr;  /* LDS byte address of the code of the second observation op */
    int ch_first;               /* raw code byte of the first observation op */
};

/* Runs the whole program of one category for this lane's site: x = ones in, root vector out. */
/* V = 0: 4-bit / 8-bit codes extracted with v_bfe (k_ll_fused4_asm); V = 1: byte codes used as they are and MATVEC + PUSH
 * written straight into the stack slot (k_ll_fused4_asm_pt, D = 4 only) */
template <int D, int V = 0>
__device__ __forceinline__ void fused_run_program_asm(double &x0, double &x1, double &x2, double &x3, int &esc,
                                                       const FusedAsmParams &p)
{
    int x0lo = __double2loint(x0), x0hi = __double2hiint(x0), x1lo = __double2loint(x1), x1hi = __double2hiint(x1);
    int x2lo = __double2loint(x2), x2hi = __double2hiint(x2), x3lo = __double2loint(x3), x3hi = __double2hiint(x3);
    if constexpr (V == 1) {
        static_assert(D <= 4, "the pair-table interpreter keeps its stack in VGPRs");
        asm volatile(PLK_ASM_PROGRAM(PLK_ASM_SLOTS_D4_POP, PLK_ASM_SLOTS_D4_PUSH, PLK_ASM_SLOTS_D4_PAIRS_D, PLK_ASM_TIPNEXT_B)
                     PLK_ASM_OPERANDS : PLK_ASM_CLOBBERS_COMMON, PLK_CLOBBER_V54_85);
    } else if constexpr (D <= 4) {
        asm volatile(PLK_ASM_PROGRAM(PLK_ASM_SLOTS_D4_POP, PLK_ASM_SLOTS_D4_PUSH, PLK_ASM_SLOTS_D4_PAIRS, PLK_ASM_TIPNEXT)
                     PLK_ASM_OPERANDS : PLK_ASM_CLOBBERS_COMMON, PLK_CLOBBER_V54_85);
    } else {
        asm volatile(PLK_ASM_PROGRAM(PLK_ASM_SLOTS_D8_POP, PLK_ASM_SLOTS_D8_PUSH, PLK_ASM_SLOTS_D8_PAIRS, PLK_ASM_TIPNEXT)
                     PLK_ASM_OPERANDS : PLK_ASM_CLOBBERS_COMMON, PLK_CLOBBER_A0_31, PLK_CLOBBER_A32_63);
    }
    x0 = __hiloint2double(x0hi, x0lo); x1 = __hiloint2double(x1hi, x1lo);
    x2 = __hiloint2double(x2hi, x2lo); x3 = __hiloint2double(x3hi, x3lo);
}

struct FusedAsmArgs {
    FusedArgs f;                /* S, Spad, C, nmat, ntips (incl. the pseudo slot), nchar, nobs, root_mode, codes, ... */
    const unsigned *words;      /* op words, blocks of 8, END padded, one spare block */
    int first_tip, first_row, second_row;
    int pack4;                  /* two codes per staged byte (nchar <= 16) */
};

/* staging, category loop and epilogue in C++, the program run in assembly */
template <int D>
__global__ __launch_bounds__(PLK_TILE) void k_ll_fused4_asm(FusedAsmArgs aa)
{
    const FusedArgs &a = aa.f;
    extern __shared__ double lds_dyn[];
    double *tip_lds = lds_dyn;
    const int tip_doubles = a.ntips * a.nchar * 4;
    uint8_t *code_lds = reinterpret_cast<uint8_t *>(lds_dyn + tip_doubles);
    const long tile0 = (long)blockIdx.x * PLK_TILE;
    const int tid = threadIdx.x;
    const long s = tile0 + tid;
    const int row_bytes = aa.pack4 ? PLK_TILE / 2 : PLK_TILE;
    {
        /* Staged code rows: wave w takes rows w, w + 4, ...; a lane moves 4 codes of a row (one dword in, one dword or
         * one 16-bit pair out).  The row's node comes through a scalar load and eight rows are requested before the first
         * one is stored: written as one load per loop iteration with the node looked up by the lane, the compiler
         * serialised two dependent global loads per row -- 25 round trips per tile, ~20 us of the 57 us a tile takes
         * for one rate category (profiles/r02_exp_headline_kernel_variants.json: 1.73 ms for one category against
         * 1.27 ms per category with four). */
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const PLK_AS4 int *obs = as_uniform(a.obs_nodes);
        for (int r0 = wave; r0 < a.nobs; r0 += 32) {
            uint32_t q[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int row = r0 + 4 * u;
                q[u] = row < a.nobs ? reinterpret_cast<const uint32_t *>(a.codes + (size_t)obs[row] * a.Spad + tile0)[lane] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int row = r0 + 4 * u;
                if (row < a.nobs) {
                    if (aa.pack4) {
                        const uint32_t packed = (q[u] & 0xf) | ((q[u] >> 4) & 0xf0) | ((q[u] >> 8) & 0xf00) | ((q[u] >> 12) & 0xf000);
                        reinterpret_cast<uint16_t *>(code_lds)[row * (PLK_TILE / 4) + lane] = (uint16_t)packed;
                    } else {
                        reinterpret_cast<uint32_t *>(code_lds)[row * (PLK_TILE / 4) + lane] = q[u];
                    }
                }
            }
        }
    }
    const PLK_AS4 double *prior = as_uniform(a.cat_prior);
    const PLK_AS4 double *rootw = as_uniform(a.root_w);

    FusedAsmParams p;
    p.ops = aa.words;
    p.tip_lds_addr = (unsigned)(size_t)tip_lds;
    p.nchar32 = (unsigned)a.nchar * 32u;
    p.row_bytes = (unsigned)row_bytes;
    p.code_width = aa.pack4 ? 4u : 8u;
    p.first_tip_addr = p.tip_lds_addr + (unsigned)aa.first_tip * p.nchar32;
    p.code_lane_addr = (unsigned)(size_t)code_lds + (aa.pack4 ? (unsigned)(tid >> 1) : (unsigned)tid);
    p.nibble_shift = aa.pack4 ? (unsigned)(tid & 1) * 4u : 0u;
    p.second_code_addr = p.code_lane_addr + (unsigned)aa.second_row * (unsigned)row_bytes;

    double sum = 0.0;
    int Eexp = 0;
    bool have = false;
    for (int c = 0; c < a.C; c++) {
        __syncthreads();
        {
            /* tip table of the category: four 16-byte loads in flight per lane */
            const double2 *src = reinterpret_cast<const double2 *>(a.tip + (size_t)c * tip_doubles);
            double2 *dst = reinterpret_cast<double2 *>(tip_lds);
            const int n2 = tip_doubles / 2;
            for (int i0 = tid; i0 < n2; i0 += 4 * PLK_TILE) {
                double2 v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = i0 + u * PLK_TILE < n2 ? src[i0 + u * PLK_TILE] : double2{0.0, 0.0};
#pragma unroll
                for (int u = 0; u < 4; u++) if (i0 + u * PLK_TILE < n2) dst[i0 + u * PLK_TILE] = v[u];
            }
        }
        __syncthreads();
        double x0 = 1.0, x1 = 1.0, x2 = 1.0, x3 = 1.0;
        int esc = 0;
        p.mstream = a.PS + (size_t)c * (a.nmat + 1) * 16;
        p.ch_first = code_lds[(p.code_lane_addr - (unsigned)(size_t)code_lds) + aa.first_row * row_bytes];
        fused_run_program_asm<D>(x0, x1, x2, x3, esc, p);
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

/* ------------------------------------------------------------------------------------------------------------
 * k_ll_fused4_asm_pt: the same interpreter over 512-site tiles with PAIR TABLES (round 3).
 *
 * A node whose two children are both leaves (a "cherry") and that hangs on an edge contributes, per site, one of
 * nchar^2 possible vectors P_a (P_b B_b o P_c B_c) -- the reference evaluates it by two _prune_update_prob calls and
 * one product per site (src/evaluate_site_lhood.c:36-56).  Here K1 tabulates those vectors once per (category,
 * cherry) in double-double (k_build_pair_tables), the staging step combines the two leaves' pattern codes into one
 * byte, and the cherry becomes a single table look-up in the program: TIP_SET, TIP_MUL and the MATVEC of the edge
 * above are one observation op.  At BASELINE config 3 that removes 33 of the 98 products and 66 of the 100 leaf
 * multiplies per category.  A pair table takes nchar ordinary tip slots ("units" of nchar * 32 bytes); every staged
 * row is one byte per site.  One workgroup of 1024 lanes per CU (4 waves per SIMD) shares one table image -- all waves
 * of a CU are then in the same rate category, whose matrix stream fits the scalar cache -- and the workgroups walk the
 * tiles with a grid stride: every CU gets the same number of tiles to within one.  (TILE = 512: two workgroups per CU.)
 * ------------------------------------------------------------------------------------------------------------ */

/* one dword of every 64-byte line of [base + first, base + bytes) at the given stride through the scalar cache, up to
 * 12 requests in flight (the counter holds 15), all waited for before the statement ends */
__device__ __forceinline__ void fused_touch_lines(const void *base, unsigned bytes, unsigned first, unsigned stride)
{
    unsigned off = first, batch;
    asm volatile("s_cmp_ge_u32 %[off], %[end]\n\t"
                 "s_cbranch_scc1 .Ltouch_done_%=\n"
                 ".Ltouch_batch_%=:\n\t"
                 "s_mov_b32 %[batch], 12\n"
                 ".Ltouch_next_%=:\n\t"
                 "s_load_dword s96, %[p], %[off]\n\t"
                 "s_add_u32 %[off], %[off], %[stride]\n\t"
                 "s_cmp_ge_u32 %[off], %[end]\n\t"
                 "s_cbranch_scc1 .Ltouch_done_%=\n\t"
                 "s_sub_u32 %[batch], %[batch], 1\n\t"
                 "s_cmp_lg_u32 %[batch], 0\n\t"
                 "s_cbranch_scc1 .Ltouch_next_%=\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "s_branch .Ltouch_batch_%=\n"
                 ".Ltouch_done_%=:\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : [off] "+s"(off), [batch] "=&s"(batch)
                 : [p] "s"(base), [end] "s"(bytes), [stride] "s"(stride)
                 : "s96", "scc", "memory");
}

struct FusedPTArgs {
    FusedArgs f;                /* ntips = table units per category, nobs = staged rows; ops / obs_nodes / first_row unused */
    const unsigned *words;      /* op words, blocks of 8, END padded, one spare block */
    const int *row_nodes;       /* [2][nobs]: node of the row | second node of a pair row or -1 */
    int first_unit, first_row, second_row;
    int ntiles, nwords;
    int warm;                   /* touch the category's matrix stream through the scalar cache before running the program */
};

template <int TILE>
__global__ __launch_bounds__(TILE) void k_ll_fused4_asm_pt(FusedPTArgs aa)
{
    const FusedArgs &a = aa.f;
    extern __shared__ double lds_dyn[];
    double *tip_lds = lds_dyn;
    const int tip_doubles = a.ntips * a.nchar * 4;
    uint8_t *code_lds = reinterpret_cast<uint8_t *>(lds_dyn + tip_doubles);
    const int tid = threadIdx.x;
    constexpr int NW = TILE / 64;         /* waves of the workgroup */
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const PLK_AS4 int *rown = as_uniform(aa.row_nodes);
    const PLK_AS4 double *prior = as_uniform(a.cat_prior);
    const PLK_AS4 double *rootw = as_uniform(a.root_w);

    FusedAsmParams p;
    p.ops = aa.words;
    p.tip_lds_addr = (unsigned)(size_t)tip_lds;
    p.nchar32 = (unsigned)a.nchar * 32u;
    p.row_bytes = TILE;
    p.code_width = 8u;
    p.first_tip_addr = p.tip_lds_addr + (unsigned)aa.first_unit * p.nchar32;
    p.code_lane_addr = (unsigned)(size_t)code_lds + (unsigned)tid;
    p.nibble_shift = 0u;
    p.second_code_addr = p.code_lane_addr + (unsigned)aa.second_row * TILE;

    for (int tile = blockIdx.x; tile < aa.ntiles; tile += gridDim.x) {
        const long tile0 = (long)tile * TILE;
        const long s = tile0 + tid;
        __syncthreads();                 /* the previous tile's last category has been read */
        /* staged rows: wave w takes rows w, w + 8, ...; a lane moves 8 codes of a row; four rows are requested before the
         * first is stored.  A pair row holds code(b) * nchar + code(c): bytes do not carry into each other (nchar <= 16). */
        for (int r0 = wave; r0 < a.nobs; r0 += 4 * NW) {
            /* a lane moves TILE / 64 codes of a row: two (TILE = 512) or four (1024) dwords */
            constexpr int ND = TILE / 256;
            struct alignas(4 * ND) Chunk { unsigned d[ND]; };
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
        double sum = 0.0;
        int Eexp = 0;
        bool have = false;
        for (int c = 0; c < a.C; c++) {
            __syncthreads();
            {
                /* Scalar-cache warm-up: the matrices of this category (nmat x 128 bytes, 8 KB at BASELINE config 3) and the op
                 * words are what every wave of the workgroup fetches through the scalar cache, one matrix at a time and in step
                 * with the other waves, so that a line missing in the 16 KB cache is waited for by all of them (44 % of the
                 * requests of the round-2 kernel: profiles/r03_exp_headline_kernel_variants.json).  Here every wave touches a
                 * slice of the lines, with all of its requests in flight at once, before the barrier below; the program then
                 * runs on cache hits. */
                if (aa.warm) {
                    fused_touch_lines(a.PS + (size_t)c * (a.nmat + 1) * 16, (unsigned)(a.nmat + 1) * 128u, (unsigned)wave * 64u, NW * 64u);
                    fused_touch_lines(aa.words, (unsigned)aa.nwords * 4u, (unsigned)wave * 64u, NW * 64u);
                }
            }
            {
                /* table image of the category: four 16-byte loads in flight per lane */
                const double2 *src = reinterpret_cast<const double2 *>(a.tip + (size_t)c * tip_doubles);
                double2 *dst = reinterpret_cast<double2 *>(tip_lds);
                const int n2 = tip_doubles / 2;
                for (int i0 = tid; i0 < n2; i0 += 4 * TILE) {
                    double2 v[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) v[u] = i0 + u * TILE < n2 ? src[i0 + u * TILE] : double2{0.0, 0.0};
#pragma unroll
                    for (int u = 0; u < 4; u++) if (i0 + u * TILE < n2) dst[i0 + u * TILE] = v[u];
                }
            }
            __syncthreads();
            double x0 = 1.0, x1 = 1.0, x2 = 1.0, x3 = 1.0;
            int esc = 0;
            p.mstream = a.PS + (size_t)c * (a.nmat + 1) * 16;
            p.ch_first = code_lds[tid + aa.first_row * TILE];
            fused_run_program_asm<4, 1>(x0, x1, x2, x3, esc, p);
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
            if (tid == 0) a.partial[tile] = r;
        }
    }
}

#endif
