/*
 * plk_fused4_asm2.h -- the assembly interpreter of plk_fused4_asm.h with TWO sites per lane.
 * Included by plk_engine.hip after plk_fused4_asm.h.
 *
 * The interpreter's scalar work (op fetch, dispatch, matrix stream, slot selection) is per
 * wavefront, not per site; with one site per lane the vector pipe is busy ~60 % of the time
 * (SQ_ACTIVE_INST_VALU, profiles/r01_v6_pmc7_counters.csv).  Here every handler updates two
 * independent 4-vectors (sites tid and tid + 256 of a 512-site tile), so the same scalar
 * stream feeds twice the vector work.  Program format, prefetch chain and arithmetic are
 * exactly those of plk_fused4_asm.h (see there), per site.
 *
 * Registers (site A / site B):
 *   x v[24:31] / v[54:61]   temporaries v[32:39] / v[62:69]   prefetched tip value v[46:53] / v[76:83]
 *   address temp v40 / v70   next code (raw) v41 / v71   scale exponent v42 / v72   temp v43 / v73
 *   LDS address of the lane's code column v44 / v74   nibble shift v45 (shared)
 *   stack slot d: a[16d : 16d+7] / a[16d+8 : 16d+15]
 *   scalar registers as in plk_fused4_asm.h
 */
#ifndef PLK_FUSED4_ASM2_H
#define PLK_FUSED4_ASM2_H


#define PLK_ASM2_POP0 \
    ".Lpop0_%=:\n\t" \
    "v_accvgpr_read_b32 v32, a0\n\t" \
    "v_accvgpr_read_b32 v33, a1\n\t" \
    "v_accvgpr_read_b32 v34, a2\n\t" \
    "v_accvgpr_read_b32 v35, a3\n\t" \
    "v_accvgpr_read_b32 v36, a4\n\t" \
    "v_accvgpr_read_b32 v37, a5\n\t" \
    "v_accvgpr_read_b32 v38, a6\n\t" \
    "v_accvgpr_read_b32 v39, a7\n\t" \
    "v_accvgpr_read_b32 v62, a8\n\t" \
    "v_accvgpr_read_b32 v63, a9\n\t" \
    "v_accvgpr_read_b32 v64, a10\n\t" \
    "v_accvgpr_read_b32 v65, a11\n\t" \
    "v_accvgpr_read_b32 v66, a12\n\t" \
    "v_accvgpr_read_b32 v67, a13\n\t" \
    "v_accvgpr_read_b32 v68, a14\n\t" \
    "v_accvgpr_read_b32 v69, a15\n\t" \
    "s_branch .Lpopmul_%=\n"

#define PLK_ASM2_PUSH0 \
    ".Lpush0_%=:\n\t" \
    "v_accvgpr_write_b32 a0, v24\n\t" \
    "v_accvgpr_write_b32 a1, v25\n\t" \
    "v_accvgpr_write_b32 a2, v26\n\t" \
    "v_accvgpr_write_b32 a3, v27\n\t" \
    "v_accvgpr_write_b32 a4, v28\n\t" \
    "v_accvgpr_write_b32 a5, v29\n\t" \
    "v_accvgpr_write_b32 a6, v30\n\t" \
    "v_accvgpr_write_b32 a7, v31\n\t" \
    "v_accvgpr_write_b32 a8, v54\n\t" \
    "v_accvgpr_write_b32 a9, v55\n\t" \
    "v_accvgpr_write_b32 a10, v56\n\t" \
    "v_accvgpr_write_b32 a11, v57\n\t" \
    "v_accvgpr_write_b32 a12, v58\n\t" \
    "v_accvgpr_write_b32 a13, v59\n\t" \
    "v_accvgpr_write_b32 a14, v60\n\t" \
    "v_accvgpr_write_b32 a15, v61\n\t" \
    "s_setpc_b64 s[88:89]\n"

#define PLK_ASM2_POP1 \
    ".Lpop1_%=:\n\t" \
    "v_accvgpr_read_b32 v32, a16\n\t" \
    "v_accvgpr_read_b32 v33, a17\n\t" \
    "v_accvgpr_read_b32 v34, a18\n\t" \
    "v_accvgpr_read_b32 v35, a19\n\t" \
    "v_accvgpr_read_b32 v36, a20\n\t" \
    "v_accvgpr_read_b32 v37, a21\n\t" \
    "v_accvgpr_read_b32 v38, a22\n\t" \
    "v_accvgpr_read_b32 v39, a23\n\t" \
    "v_accvgpr_read_b32 v62, a24\n\t" \
    "v_accvgpr_read_b32 v63, a25\n\t" \
    "v_accvgpr_read_b32 v64, a26\n\t" \
    "v_accvgpr_read_b32 v65, a27\n\t" \
    "v_accvgpr_read_b32 v66, a28\n\t" \
    "v_accvgpr_read_b32 v67, a29\n\t" \
    "v_accvgpr_read_b32 v68, a30\n\t" \
    "v_accvgpr_read_b32 v69, a31\n\t" \
    "s_branch .Lpopmul_%=\n"

#define PLK_ASM2_PUSH1 \
    ".Lpush1_%=:\n\t" \
    "v_accvgpr_write_b32 a16, v24\n\t" \
    "v_accvgpr_write_b32 a17, v25\n\t" \
    "v_accvgpr_write_b32 a18, v26\n\t" \
    "v_accvgpr_write_b32 a19, v27\n\t" \
    "v_accvgpr_write_b32 a20, v28\n\t" \
    "v_accvgpr_write_b32 a21, v29\n\t" \
    "v_accvgpr_write_b32 a22, v30\n\t" \
    "v_accvgpr_write_b32 a23, v31\n\t" \
    "v_accvgpr_write_b32 a24, v54\n\t" \
    "v_accvgpr_write_b32 a25, v55\n\t" \
    "v_accvgpr_write_b32 a26, v56\n\t" \
    "v_accvgpr_write_b32 a27, v57\n\t" \
    "v_accvgpr_write_b32 a28, v58\n\t" \
    "v_accvgpr_write_b32 a29, v59\n\t" \
    "v_accvgpr_write_b32 a30, v60\n\t" \
    "v_accvgpr_write_b32 a31, v61\n\t" \
    "s_setpc_b64 s[88:89]\n"

#define PLK_ASM2_POP2 \
    ".Lpop2_%=:\n\t" \
    "v_accvgpr_read_b32 v32, a32\n\t" \
    "v_accvgpr_read_b32 v33, a33\n\t" \
    "v_accvgpr_read_b32 v34, a34\n\t" \
    "v_accvgpr_read_b32 v35, a35\n\t" \
    "v_accvgpr_read_b32 v36, a36\n\t" \
    "v_accvgpr_read_b32 v37, a37\n\t" \
    "v_accvgpr_read_b32 v38, a38\n\t" \
    "v_accvgpr_read_b32 v39, a39\n\t" \
    "v_accvgpr_read_b32 v62, a40\n\t" \
    "v_accvgpr_read_b32 v63, a41\n\t" \
    "v_accvgpr_read_b32 v64, a42\n\t" \
    "v_accvgpr_read_b32 v65, a43\n\t" \
    "v_accvgpr_read_b32 v66, a44\n\t" \
    "v_accvgpr_read_b32 v67, a45\n\t" \
    "v_accvgpr_read_b32 v68, a46\n\t" \
    "v_accvgpr_read_b32 v69, a47\n\t" \
    "s_branch .Lpopmul_%=\n"

#define PLK_ASM2_PUSH2 \
    ".Lpush2_%=:\n\t" \
    "v_accvgpr_write_b32 a32, v24\n\t" \
    "v_accvgpr_write_b32 a33, v25\n\t" \
    "v_accvgpr_write_b32 a34, v26\n\t" \
    "v_accvgpr_write_b32 a35, v27\n\t" \
    "v_accvgpr_write_b32 a36, v28\n\t" \
    "v_accvgpr_write_b32 a37, v29\n\t" \
    "v_accvgpr_write_b32 a38, v30\n\t" \
    "v_accvgpr_write_b32 a39, v31\n\t" \
    "v_accvgpr_write_b32 a40, v54\n\t" \
    "v_accvgpr_write_b32 a41, v55\n\t" \
    "v_accvgpr_write_b32 a42, v56\n\t" \
    "v_accvgpr_write_b32 a43, v57\n\t" \
    "v_accvgpr_write_b32 a44, v58\n\t" \
    "v_accvgpr_write_b32 a45, v59\n\t" \
    "v_accvgpr_write_b32 a46, v60\n\t" \
    "v_accvgpr_write_b32 a47, v61\n\t" \
    "s_setpc_b64 s[88:89]\n"

#define PLK_ASM2_POP3 \
    ".Lpop3_%=:\n\t" \
    "v_accvgpr_read_b32 v32, a48\n\t" \
    "v_accvgpr_read_b32 v33, a49\n\t" \
    "v_accvgpr_read_b32 v34, a50\n\t" \
    "v_accvgpr_read_b32 v35, a51\n\t" \
    "v_accvgpr_read_b32 v36, a52\n\t" \
    "v_accvgpr_read_b32 v37, a53\n\t" \
    "v_accvgpr_read_b32 v38, a54\n\t" \
    "v_accvgpr_read_b32 v39, a55\n\t" \
    "v_accvgpr_read_b32 v62, a56\n\t" \
    "v_accvgpr_read_b32 v63, a57\n\t" \
    "v_accvgpr_read_b32 v64, a58\n\t" \
    "v_accvgpr_read_b32 v65, a59\n\t" \
    "v_accvgpr_read_b32 v66, a60\n\t" \
    "v_accvgpr_read_b32 v67, a61\n\t" \
    "v_accvgpr_read_b32 v68, a62\n\t" \
    "v_accvgpr_read_b32 v69, a63\n\t" \
    "s_branch .Lpopmul_%=\n"

#define PLK_ASM2_PUSH3 \
    ".Lpush3_%=:\n\t" \
    "v_accvgpr_write_b32 a48, v24\n\t" \
    "v_accvgpr_write_b32 a49, v25\n\t" \
    "v_accvgpr_write_b32 a50, v26\n\t" \
    "v_accvgpr_write_b32 a51, v27\n\t" \
    "v_accvgpr_write_b32 a52, v28\n\t" \
    "v_accvgpr_write_b32 a53, v29\n\t" \
    "v_accvgpr_write_b32 a54, v30\n\t" \
    "v_accvgpr_write_b32 a55, v31\n\t" \
    "v_accvgpr_write_b32 a56, v54\n\t" \
    "v_accvgpr_write_b32 a57, v55\n\t" \
    "v_accvgpr_write_b32 a58, v56\n\t" \
    "v_accvgpr_write_b32 a59, v57\n\t" \
    "v_accvgpr_write_b32 a60, v58\n\t" \
    "v_accvgpr_write_b32 a61, v59\n\t" \
    "v_accvgpr_write_b32 a62, v60\n\t" \
    "v_accvgpr_write_b32 a63, v61\n\t" \
    "s_setpc_b64 s[88:89]\n"

#define PLK_ASM2_POP4 \
    ".Lpop4_%=:\n\t" \
    "v_accvgpr_read_b32 v32, a64\n\t" \
    "v_accvgpr_read_b32 v33, a65\n\t" \
    "v_accvgpr_read_b32 v34, a66\n\t" \
    "v_accvgpr_read_b32 v35, a67\n\t" \
    "v_accvgpr_read_b32 v36, a68\n\t" \
    "v_accvgpr_read_b32 v37, a69\n\t" \
    "v_accvgpr_read_b32 v38, a70\n\t" \
    "v_accvgpr_read_b32 v39, a71\n\t" \
    "v_accvgpr_read_b32 v62, a72\n\t" \
    "v_accvgpr_read_b32 v63, a73\n\t" \
    "v_accvgpr_read_b32 v64, a74\n\t" \
    "v_accvgpr_read_b32 v65, a75\n\t" \
    "v_accvgpr_read_b32 v66, a76\n\t" \
    "v_accvgpr_read_b32 v67, a77\n\t" \
    "v_accvgpr_read_b32 v68, a78\n\t" \
    "v_accvgpr_read_b32 v69, a79\n\t" \
    "s_branch .Lpopmul_%=\n"

#define PLK_ASM2_PUSH4 \
    ".Lpush4_%=:\n\t" \
    "v_accvgpr_write_b32 a64, v24\n\t" \
    "v_accvgpr_write_b32 a65, v25\n\t" \
    "v_accvgpr_write_b32 a66, v26\n\t" \
    "v_accvgpr_write_b32 a67, v27\n\t" \
    "v_accvgpr_write_b32 a68, v28\n\t" \
    "v_accvgpr_write_b32 a69, v29\n\t" \
    "v_accvgpr_write_b32 a70, v30\n\t" \
    "v_accvgpr_write_b32 a71, v31\n\t" \
    "v_accvgpr_write_b32 a72, v54\n\t" \
    "v_accvgpr_write_b32 a73, v55\n\t" \
    "v_accvgpr_write_b32 a74, v56\n\t" \
    "v_accvgpr_write_b32 a75, v57\n\t" \
    "v_accvgpr_write_b32 a76, v58\n\t" \
    "v_accvgpr_write_b32 a77, v59\n\t" \
    "v_accvgpr_write_b32 a78, v60\n\t" \
    "v_accvgpr_write_b32 a79, v61\n\t" \
    "s_setpc_b64 s[88:89]\n"

#define PLK_ASM2_POP5 \
    ".Lpop5_%=:\n\t" \
    "v_accvgpr_read_b32 v32, a80\n\t" \
    "v_accvgpr_read_b32 v33, a81\n\t" \
    "v_accvgpr_read_b32 v34, a82\n\t" \
    "v_accvgpr_read_b32 v35, a83\n\t" \
    "v_accvgpr_read_b32 v36, a84\n\t" \
    "v_accvgpr_read_b32 v37, a85\n\t" \
    "v_accvgpr_read_b32 v38, a86\n\t" \
    "v_accvgpr_read_b32 v39, a87\n\t" \
    "v_accvgpr_read_b32 v62, a88\n\t" \
    "v_accvgpr_read_b32 v63, a89\n\t" \
    "v_accvgpr_read_b32 v64, a90\n\t" \
    "v_accvgpr_read_b32 v65, a91\n\t" \
    "v_accvgpr_read_b32 v66, a92\n\t" \
    "v_accvgpr_read_b32 v67, a93\n\t" \
    "v_accvgpr_read_b32 v68, a94\n\t" \
    "v_accvgpr_read_b32 v69, a95\n\t" \
    "s_branch .Lpopmul_%=\n"

#define PLK_ASM2_PUSH5 \
    ".Lpush5_%=:\n\t" \
    "v_accvgpr_write_b32 a80, v24\n\t" \
    "v_accvgpr_write_b32 a81, v25\n\t" \
    "v_accvgpr_write_b32 a82, v26\n\t" \
    "v_accvgpr_write_b32 a83, v27\n\t" \
    "v_accvgpr_write_b32 a84, v28\n\t" \
    "v_accvgpr_write_b32 a85, v29\n\t" \
    "v_accvgpr_write_b32 a86, v30\n\t" \
    "v_accvgpr_write_b32 a87, v31\n\t" \
    "v_accvgpr_write_b32 a88, v54\n\t" \
    "v_accvgpr_write_b32 a89, v55\n\t" \
    "v_accvgpr_write_b32 a90, v56\n\t" \
    "v_accvgpr_write_b32 a91, v57\n\t" \
    "v_accvgpr_write_b32 a92, v58\n\t" \
    "v_accvgpr_write_b32 a93, v59\n\t" \
    "v_accvgpr_write_b32 a94, v60\n\t" \
    "v_accvgpr_write_b32 a95, v61\n\t" \
    "s_setpc_b64 s[88:89]\n"

#define PLK_ASM2_POP6 \
    ".Lpop6_%=:\n\t" \
    "v_accvgpr_read_b32 v32, a96\n\t" \
    "v_accvgpr_read_b32 v33, a97\n\t" \
    "v_accvgpr_read_b32 v34, a98\n\t" \
    "v_accvgpr_read_b32 v35, a99\n\t" \
    "v_accvgpr_read_b32 v36, a100\n\t" \
    "v_accvgpr_read_b32 v37, a101\n\t" \
    "v_accvgpr_read_b32 v38, a102\n\t" \
    "v_accvgpr_read_b32 v39, a103\n\t" \
    "v_accvgpr_read_b32 v62, a104\n\t" \
    "v_accvgpr_read_b32 v63, a105\n\t" \
    "v_accvgpr_read_b32 v64, a106\n\t" \
    "v_accvgpr_read_b32 v65, a107\n\t" \
    "v_accvgpr_read_b32 v66, a108\n\t" \
    "v_accvgpr_read_b32 v67, a109\n\t" \
    "v_accvgpr_read_b32 v68, a110\n\t" \
    "v_accvgpr_read_b32 v69, a111\n\t" \
    "s_branch .Lpopmul_%=\n"

#define PLK_ASM2_PUSH6 \
    ".Lpush6_%=:\n\t" \
    "v_accvgpr_write_b32 a96, v24\n\t" \
    "v_accvgpr_write_b32 a97, v25\n\t" \
    "v_accvgpr_write_b32 a98, v26\n\t" \
    "v_accvgpr_write_b32 a99, v27\n\t" \
    "v_accvgpr_write_b32 a100, v28\n\t" \
    "v_accvgpr_write_b32 a101, v29\n\t" \
    "v_accvgpr_write_b32 a102, v30\n\t" \
    "v_accvgpr_write_b32 a103, v31\n\t" \
    "v_accvgpr_write_b32 a104, v54\n\t" \
    "v_accvgpr_write_b32 a105, v55\n\t" \
    "v_accvgpr_write_b32 a106, v56\n\t" \
    "v_accvgpr_write_b32 a107, v57\n\t" \
    "v_accvgpr_write_b32 a108, v58\n\t" \
    "v_accvgpr_write_b32 a109, v59\n\t" \
    "v_accvgpr_write_b32 a110, v60\n\t" \
    "v_accvgpr_write_b32 a111, v61\n\t" \
    "s_setpc_b64 s[88:89]\n"

#define PLK_ASM2_POP7 \
    ".Lpop7_%=:\n\t" \
    "v_accvgpr_read_b32 v32, a112\n\t" \
    "v_accvgpr_read_b32 v33, a113\n\t" \
    "v_accvgpr_read_b32 v34, a114\n\t" \
    "v_accvgpr_read_b32 v35, a115\n\t" \
    "v_accvgpr_read_b32 v36, a116\n\t" \
    "v_accvgpr_read_b32 v37, a117\n\t" \
    "v_accvgpr_read_b32 v38, a118\n\t" \
    "v_accvgpr_read_b32 v39, a119\n\t" \
    "v_accvgpr_read_b32 v62, a120\n\t" \
    "v_accvgpr_read_b32 v63, a121\n\t" \
    "v_accvgpr_read_b32 v64, a122\n\t" \
    "v_accvgpr_read_b32 v65, a123\n\t" \
    "v_accvgpr_read_b32 v66, a124\n\t" \
    "v_accvgpr_read_b32 v67, a125\n\t" \
    "v_accvgpr_read_b32 v68, a126\n\t" \
    "v_accvgpr_read_b32 v69, a127\n\t" \
    "s_branch .Lpopmul_%=\n"

#define PLK_ASM2_PUSH7 \
    ".Lpush7_%=:\n\t" \
    "v_accvgpr_write_b32 a112, v24\n\t" \
    "v_accvgpr_write_b32 a113, v25\n\t" \
    "v_accvgpr_write_b32 a114, v26\n\t" \
    "v_accvgpr_write_b32 a115, v27\n\t" \
    "v_accvgpr_write_b32 a116, v28\n\t" \
    "v_accvgpr_write_b32 a117, v29\n\t" \
    "v_accvgpr_write_b32 a118, v30\n\t" \
    "v_accvgpr_write_b32 a119, v31\n\t" \
    "v_accvgpr_write_b32 a120, v54\n\t" \
    "v_accvgpr_write_b32 a121, v55\n\t" \
    "v_accvgpr_write_b32 a122, v56\n\t" \
    "v_accvgpr_write_b32 a123, v57\n\t" \
    "v_accvgpr_write_b32 a124, v58\n\t" \
    "v_accvgpr_write_b32 a125, v59\n\t" \
    "v_accvgpr_write_b32 a126, v60\n\t" \
    "v_accvgpr_write_b32 a127, v61\n\t" \
    "s_setpc_b64 s[88:89]\n"

#define PLK_ASM2_SLOTS_D4_POP \
    "s_cmp_eq_u32 s97, 0\n\ts_cbranch_scc1 .Lpop0_%=\n\t" \
    "s_cmp_eq_u32 s97, 1\n\ts_cbranch_scc1 .Lpop1_%=\n\t" \
    "s_cmp_eq_u32 s97, 2\n\ts_cbranch_scc1 .Lpop2_%=\n\t" \
    "s_branch .Lpop3_%=\n" \
    PLK_ASM2_POP0 PLK_ASM2_POP1 PLK_ASM2_POP2 PLK_ASM2_POP3

#define PLK_ASM2_SLOTS_D4_PUSH \
    "s_cmp_eq_u32 s97, 0\n\ts_cbranch_scc1 .Lpush0_%=\n\t" \
    "s_cmp_eq_u32 s97, 1\n\ts_cbranch_scc1 .Lpush1_%=\n\t" \
    "s_cmp_eq_u32 s97, 2\n\ts_cbranch_scc1 .Lpush2_%=\n\t" \
    "s_branch .Lpush3_%=\n" \
    PLK_ASM2_PUSH0 PLK_ASM2_PUSH1 PLK_ASM2_PUSH2 PLK_ASM2_PUSH3

#define PLK_ASM2_SLOTS_D8_POP \
    "s_cmp_eq_u32 s97, 0\n\ts_cbranch_scc1 .Lpop0_%=\n\t" \
    "s_cmp_eq_u32 s97, 1\n\ts_cbranch_scc1 .Lpop1_%=\n\t" \
    "s_cmp_eq_u32 s97, 2\n\ts_cbranch_scc1 .Lpop2_%=\n\t" \
    "s_cmp_eq_u32 s97, 3\n\ts_cbranch_scc1 .Lpop3_%=\n\t" \
    "s_cmp_eq_u32 s97, 4\n\ts_cbranch_scc1 .Lpop4_%=\n\t" \
    "s_cmp_eq_u32 s97, 5\n\ts_cbranch_scc1 .Lpop5_%=\n\t" \
    "s_cmp_eq_u32 s97, 6\n\ts_cbranch_scc1 .Lpop6_%=\n\t" \
    "s_branch .Lpop7_%=\n" \
    PLK_ASM2_POP0 PLK_ASM2_POP1 PLK_ASM2_POP2 PLK_ASM2_POP3 PLK_ASM2_POP4 PLK_ASM2_POP5 PLK_ASM2_POP6 PLK_ASM2_POP7

#define PLK_ASM2_SLOTS_D8_PUSH \
    "s_cmp_eq_u32 s97, 0\n\ts_cbranch_scc1 .Lpush0_%=\n\t" \
    "s_cmp_eq_u32 s97, 1\n\ts_cbranch_scc1 .Lpush1_%=\n\t" \
    "s_cmp_eq_u32 s97, 2\n\ts_cbranch_scc1 .Lpush2_%=\n\t" \
    "s_cmp_eq_u32 s97, 3\n\ts_cbranch_scc1 .Lpush3_%=\n\t" \
    "s_cmp_eq_u32 s97, 4\n\ts_cbranch_scc1 .Lpush4_%=\n\t" \
    "s_cmp_eq_u32 s97, 5\n\ts_cbranch_scc1 .Lpush5_%=\n\t" \
    "s_cmp_eq_u32 s97, 6\n\ts_cbranch_scc1 .Lpush6_%=\n\t" \
    "s_branch .Lpush7_%=\n" \
    PLK_ASM2_PUSH0 PLK_ASM2_PUSH1 PLK_ASM2_PUSH2 PLK_ASM2_PUSH3 PLK_ASM2_PUSH4 PLK_ASM2_PUSH5 PLK_ASM2_PUSH6 PLK_ASM2_PUSH7

#define PLK_ASM2_PROGRAM(POP_SLOTS, PUSH_SLOTS) \
        "v_mov_b32 v24, %[x0lo]\n\t"                                                        \
        "v_mov_b32 v25, %[x0hi]\n\t"                                                        \
        "v_mov_b32 v26, %[x1lo]\n\t"                                                        \
        "v_mov_b32 v27, %[x1hi]\n\t"                                                        \
        "v_mov_b32 v28, %[x2lo]\n\t"                                                        \
        "v_mov_b32 v29, %[x2hi]\n\t"                                                        \
        "v_mov_b32 v30, %[x3lo]\n\t"                                                        \
        "v_mov_b32 v31, %[x3hi]\n\t"                                                        \
        "v_mov_b32 v54, %[y0lo]\n\t"                                                        \
        "v_mov_b32 v55, %[y0hi]\n\t"                                                        \
        "v_mov_b32 v56, %[y1lo]\n\t"                                                        \
        "v_mov_b32 v57, %[y1hi]\n\t"                                                        \
        "v_mov_b32 v58, %[y2lo]\n\t"                                                        \
        "v_mov_b32 v59, %[y2hi]\n\t"                                                        \
        "v_mov_b32 v60, %[y3lo]\n\t"                                                        \
        "v_mov_b32 v61, %[y3hi]\n\t"                                                        \
        "v_mov_b32 v42, 0\n\t"                                                              \
        "v_mov_b32 v72, 0\n\t"                                                              \
        "v_mov_b32 v44, %[clane]\n\t"                                                       \
        "v_mov_b32 v74, %[clane2]\n\t"                                                      \
        "v_mov_b32 v45, %[nshift]\n\t"                                                      \
        "s_mov_b64 s[84:85], %[ops]\n\t"                                                    \
        "s_mov_b64 s[86:87], %[mstream]\n\t"                                                \
        "s_mov_b32 s92, %[tipbase]\n\t"                                                     \
        "s_mov_b32 s93, %[nchar32]\n\t"                                                     \
        "s_mov_b32 s94, %[tile]\n\t"                                                        \
        "s_mov_b32 s35, %[cwidth]\n\t"                                                      \
        "s_movk_i32 s95, 0xfc02\n\t"                                                        \
        "s_load_dwordx8 s[68:75], s[84:85], 0x0\n\t"                                        \
        "s_load_dwordx16 s[36:51], s[86:87], 0x0\n\t"                                       \
        "s_load_dwordx16 s[52:67], s[86:87], 0x40\n\t"                                      \
        "v_bfe_u32 v43, %[ch], v45, s35\n\t"                                                \
        "v_lshl_add_u32 v40, v43, 5, %[firsttip]\n\t"                                       \
        "ds_read_b64 v[46:47], v40\n\t"                                                     \
        "ds_read_b64 v[48:49], v40 offset:8\n\t"                                            \
        "ds_read_b64 v[50:51], v40 offset:16\n\t"                                           \
        "ds_read_b64 v[52:53], v40 offset:24\n\t"                                           \
        "ds_read_u8 v41, %[secaddr]\n\t"                                                    \
        "v_bfe_u32 v73, %[ch2], v45, s35\n\t"                                               \
        "v_lshl_add_u32 v70, v73, 5, %[firsttip]\n\t"                                       \
        "ds_read_b64 v[76:77], v70\n\t"                                                     \
        "ds_read_b64 v[78:79], v70 offset:8\n\t"                                            \
        "ds_read_b64 v[80:81], v70 offset:16\n\t"                                           \
        "ds_read_b64 v[82:83], v70 offset:24\n\t"                                           \
        "ds_read_u8 v71, %[secaddr2]\n\t"                                                   \
        "s_getpc_b64 s[90:91]\n" \
        ".Lpcref_%=:\n\t" \
        "s_add_u32 s90, s90, .Ldispatch_%=-.Lpcref_%=\n\t"                                  \
        "s_addc_u32 s91, s91, 0\n\t"                                                        \
        "s_waitcnt lgkmcnt(0)\n" \
        ".Lblock_%=:\n\t" \
        "s_load_dwordx8 s[76:83], s[84:85], 0x20\n\t"                                       \
        "s_add_u32 s84, s84, 32\n\t"                                                        \
        "s_addc_u32 s85, s85, 0\n\t"                                                        \
        "s_mov_b32 s96, s68\n\t"                                                            \
        "s_swappc_b64 s[88:89], s[90:91]\n\t"                                               \
        "s_mov_b32 s96, s69\n\t"                                                            \
        "s_swappc_b64 s[88:89], s[90:91]\n\t"                                               \
        "s_mov_b32 s96, s70\n\t"                                                            \
        "s_swappc_b64 s[88:89], s[90:91]\n\t"                                               \
        "s_mov_b32 s96, s71\n\t"                                                            \
        "s_swappc_b64 s[88:89], s[90:91]\n\t"                                               \
        "s_mov_b32 s96, s72\n\t"                                                            \
        "s_swappc_b64 s[88:89], s[90:91]\n\t"                                               \
        "s_mov_b32 s96, s73\n\t"                                                            \
        "s_swappc_b64 s[88:89], s[90:91]\n\t"                                               \
        "s_mov_b32 s96, s74\n\t"                                                            \
        "s_swappc_b64 s[88:89], s[90:91]\n\t"                                               \
        "s_mov_b32 s96, s75\n\t"                                                            \
        "s_swappc_b64 s[88:89], s[90:91]\n\t"                                               \
        "s_waitcnt lgkmcnt(0)\n\t"                                                          \
        "s_mov_b64 s[68:69], s[76:77]\n\t"                                                  \
        "s_mov_b64 s[70:71], s[78:79]\n\t"                                                  \
        "s_mov_b64 s[72:73], s[80:81]\n\t"                                                  \
        "s_mov_b64 s[74:75], s[82:83]\n\t"                                                  \
        "s_branch .Lblock_%=\n" \
        ".Ldispatch_%=:\n\t" \
        "s_and_b32 s97, s96, 7\n\t"                                                         \
        "s_cmp_eq_u32 s97, 2\n\t"                                                           \
        "s_cbranch_scc1 .Lmatvec_%=\n\t"                                                    \
        "s_cmp_eq_u32 s97, 5\n\t"                                                           \
        "s_cbranch_scc1 .Ltipmul_nw_%=\n\t"                                                 \
        "s_cmp_lt_u32 s97, 2\n\t"                                                           \
        "s_cbranch_scc1 .Ltip_%=\n\t"                                                       \
        "s_cmp_eq_u32 s97, 4\n\t"                                                           \
        "s_cbranch_scc1 .Lpop_%=\n\t"                                                       \
        "s_cmp_eq_u32 s97, 3\n\t"                                                           \
        "s_cbranch_scc1 .Lpush_%=\n\t"                                                      \
        "s_cmp_eq_u32 s97, 6\n\t"                                                           \
        "s_cbranch_scc1 .Lscale_%=\n\t"                                                     \
        "s_branch .Ldone_%=\n" \
        ".Lmatvec_%=:\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t"                                                          \
        "v_mul_f64 v[32:33], s[36:37], v[24:25]\n\t"                                        \
        "v_mul_f64 v[62:63], s[36:37], v[54:55]\n\t"                                        \
        "v_mul_f64 v[34:35], s[38:39], v[24:25]\n\t"                                        \
        "v_mul_f64 v[64:65], s[38:39], v[54:55]\n\t"                                        \
        "v_mul_f64 v[36:37], s[40:41], v[24:25]\n\t"                                        \
        "v_mul_f64 v[66:67], s[40:41], v[54:55]\n\t"                                        \
        "v_mul_f64 v[38:39], s[42:43], v[24:25]\n\t"                                        \
        "v_mul_f64 v[68:69], s[42:43], v[54:55]\n\t"                                        \
        "v_fma_f64 v[32:33], s[44:45], v[26:27], v[32:33]\n\t"                              \
        "v_fma_f64 v[62:63], s[44:45], v[56:57], v[62:63]\n\t"                              \
        "v_fma_f64 v[34:35], s[46:47], v[26:27], v[34:35]\n\t"                              \
        "v_fma_f64 v[64:65], s[46:47], v[56:57], v[64:65]\n\t"                              \
        "v_fma_f64 v[36:37], s[48:49], v[26:27], v[36:37]\n\t"                              \
        "v_fma_f64 v[66:67], s[48:49], v[56:57], v[66:67]\n\t"                              \
        "v_fma_f64 v[38:39], s[50:51], v[26:27], v[38:39]\n\t"                              \
        "v_fma_f64 v[68:69], s[50:51], v[56:57], v[68:69]\n\t"                              \
        "v_fma_f64 v[32:33], s[52:53], v[28:29], v[32:33]\n\t"                              \
        "v_fma_f64 v[62:63], s[52:53], v[58:59], v[62:63]\n\t"                              \
        "v_fma_f64 v[34:35], s[54:55], v[28:29], v[34:35]\n\t"                              \
        "v_fma_f64 v[64:65], s[54:55], v[58:59], v[64:65]\n\t"                              \
        "v_fma_f64 v[36:37], s[56:57], v[28:29], v[36:37]\n\t"                              \
        "v_fma_f64 v[66:67], s[56:57], v[58:59], v[66:67]\n\t"                              \
        "v_fma_f64 v[38:39], s[58:59], v[28:29], v[38:39]\n\t"                              \
        "v_fma_f64 v[68:69], s[58:59], v[58:59], v[68:69]\n\t"                              \
        "v_fma_f64 v[24:25], s[60:61], v[30:31], v[32:33]\n\t"                              \
        "v_fma_f64 v[54:55], s[60:61], v[60:61], v[62:63]\n\t"                              \
        "v_fma_f64 v[26:27], s[62:63], v[30:31], v[34:35]\n\t"                              \
        "v_fma_f64 v[56:57], s[62:63], v[60:61], v[64:65]\n\t"                              \
        "v_fma_f64 v[28:29], s[64:65], v[30:31], v[36:37]\n\t"                              \
        "v_fma_f64 v[58:59], s[64:65], v[60:61], v[66:67]\n\t"                              \
        "v_fma_f64 v[30:31], s[66:67], v[30:31], v[38:39]\n\t"                              \
        "v_fma_f64 v[60:61], s[66:67], v[60:61], v[68:69]\n\t"                              \
        "s_add_u32 s86, s86, 0x80\n\t"                                                      \
        "s_addc_u32 s87, s87, 0\n\t"                                                        \
        "s_load_dwordx16 s[36:51], s[86:87], 0x0\n\t"                                       \
        "s_load_dwordx16 s[52:67], s[86:87], 0x40\n\t"                                      \
        "s_setpc_b64 s[88:89]\n" \
        ".Ltip_%=:\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t"                                                          \
        "s_cmp_eq_u32 s97, 0\n\t"                                                           \
        "s_cbranch_scc1 .Ltipset_%=\n" \
        ".Ltipmul_nw_%=:\n\t" \
        "v_mul_f64 v[24:25], v[24:25], v[46:47]\n\t"                                        \
        "v_mul_f64 v[54:55], v[54:55], v[76:77]\n\t"                                        \
        "v_mul_f64 v[26:27], v[26:27], v[48:49]\n\t"                                        \
        "v_mul_f64 v[56:57], v[56:57], v[78:79]\n\t"                                        \
        "v_mul_f64 v[28:29], v[28:29], v[50:51]\n\t"                                        \
        "v_mul_f64 v[58:59], v[58:59], v[80:81]\n\t"                                        \
        "v_mul_f64 v[30:31], v[30:31], v[52:53]\n\t"                                        \
        "v_mul_f64 v[60:61], v[60:61], v[82:83]\n\t"                                        \
        "s_branch .Ltipnext_%=\n" \
        ".Ltipset_%=:\n\t" \
        "v_mov_b64 v[24:25], v[46:47]\n\t"                                                  \
        "v_mov_b64 v[26:27], v[48:49]\n\t"                                                  \
        "v_mov_b64 v[28:29], v[50:51]\n\t"                                                  \
        "v_mov_b64 v[30:31], v[52:53]\n\t"                                                  \
        "v_mov_b64 v[54:55], v[76:77]\n\t"                                                  \
        "v_mov_b64 v[56:57], v[78:79]\n\t"                                                  \
        "v_mov_b64 v[58:59], v[80:81]\n\t"                                                  \
        "v_mov_b64 v[60:61], v[82:83]\n" \
        ".Ltipnext_%=:\n\t" \
        "s_bfe_u32 s98, s96, 0xd0003\n\t"                                                   \
        "s_mul_i32 s98, s98, s93\n\t"                                                       \
        "s_add_u32 s98, s98, s92\n\t"                                                       \
        "s_lshr_b32 s99, s96, 16\n\t"                                                       \
        "s_mul_i32 s99, s99, s94\n\t"                                                       \
        "v_bfe_u32 v43, v41, v45, s35\n\t"                                                  \
        "v_bfe_u32 v73, v71, v45, s35\n\t"                                                  \
        "v_lshl_add_u32 v40, v43, 5, s98\n\t"                                               \
        "v_lshl_add_u32 v70, v73, 5, s98\n\t"                                               \
        "ds_read_b64 v[46:47], v40\n\t"                                                     \
        "ds_read_b64 v[48:49], v40 offset:8\n\t"                                            \
        "ds_read_b64 v[50:51], v40 offset:16\n\t"                                           \
        "ds_read_b64 v[52:53], v40 offset:24\n\t"                                           \
        "ds_read_b64 v[76:77], v70\n\t"                                                     \
        "ds_read_b64 v[78:79], v70 offset:8\n\t"                                            \
        "ds_read_b64 v[80:81], v70 offset:16\n\t"                                           \
        "ds_read_b64 v[82:83], v70 offset:24\n\t"                                           \
        "v_add_u32 v43, s99, v44\n\t"                                                       \
        "v_add_u32 v73, s99, v74\n\t"                                                       \
        "ds_read_u8 v41, v43\n\t"                                                           \
        "ds_read_u8 v71, v73\n\t"                                                           \
        "s_setpc_b64 s[88:89]\n" \
        ".Lpop_%=:\n\t" \
        "s_bfe_u32 s97, s96, 0xd0003\n\t" \
        POP_SLOTS \
        ".Lpopmul_%=:\n\t" \
        "v_mul_f64 v[24:25], v[24:25], v[32:33]\n\t"                                        \
        "v_mul_f64 v[54:55], v[54:55], v[62:63]\n\t"                                        \
        "v_mul_f64 v[26:27], v[26:27], v[34:35]\n\t"                                        \
        "v_mul_f64 v[56:57], v[56:57], v[64:65]\n\t"                                        \
        "v_mul_f64 v[28:29], v[28:29], v[36:37]\n\t"                                        \
        "v_mul_f64 v[58:59], v[58:59], v[66:67]\n\t"                                        \
        "v_mul_f64 v[30:31], v[30:31], v[38:39]\n\t"                                        \
        "v_mul_f64 v[60:61], v[60:61], v[68:69]\n\t"                                        \
        "s_setpc_b64 s[88:89]\n" \
        ".Lpush_%=:\n\t" \
        "s_bfe_u32 s97, s96, 0xd0003\n\t" \
        PUSH_SLOTS \
        ".Lscale_%=:\n\t" \
        "v_max_u32 v43, v25, v27\n\t"                                                       \
        "v_max_u32 v73, v55, v57\n\t"                                                       \
        "v_max3_u32 v43, v29, v31, v43\n\t"                                                 \
        "v_max3_u32 v73, v59, v61, v73\n\t"                                                 \
        "v_lshrrev_b32 v43, 20, v43\n\t"                                                    \
        "v_lshrrev_b32 v73, 20, v73\n\t"                                                    \
        "v_sub_u32 v40, 0x3fe, v43\n\t"                                                     \
        "v_sub_u32 v70, 0x3fe, v73\n\t"                                                     \
        "v_ldexp_f64 v[24:25], v[24:25], v40\n\t"                                           \
        "v_ldexp_f64 v[54:55], v[54:55], v70\n\t"                                           \
        "v_ldexp_f64 v[26:27], v[26:27], v40\n\t"                                           \
        "v_ldexp_f64 v[56:57], v[56:57], v70\n\t"                                           \
        "v_ldexp_f64 v[28:29], v[28:29], v40\n\t"                                           \
        "v_ldexp_f64 v[58:59], v[58:59], v70\n\t"                                           \
        "v_ldexp_f64 v[30:31], v[30:31], v40\n\t"                                           \
        "v_ldexp_f64 v[60:61], v[60:61], v70\n\t"                                           \
        "v_add3_u32 v42, v42, v43, s95\n\t"                                                 \
        "v_add3_u32 v72, v72, v73, s95\n\t"                                                 \
        "s_setpc_b64 s[88:89]\n" \
        ".Ldone_%=:\n\t" \
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"                                                 \
        "v_mov_b32 %[x0lo], v24\n\t"                                                        \
        "v_mov_b32 %[x0hi], v25\n\t"                                                        \
        "v_mov_b32 %[x1lo], v26\n\t"                                                        \
        "v_mov_b32 %[x1hi], v27\n\t"                                                        \
        "v_mov_b32 %[x2lo], v28\n\t"                                                        \
        "v_mov_b32 %[x2hi], v29\n\t"                                                        \
        "v_mov_b32 %[x3lo], v30\n\t"                                                        \
        "v_mov_b32 %[x3hi], v31\n\t"                                                        \
        "v_mov_b32 %[y0lo], v54\n\t"                                                        \
        "v_mov_b32 %[y0hi], v55\n\t"                                                        \
        "v_mov_b32 %[y1lo], v56\n\t"                                                        \
        "v_mov_b32 %[y1hi], v57\n\t"                                                        \
        "v_mov_b32 %[y2lo], v58\n\t"                                                        \
        "v_mov_b32 %[y2hi], v59\n\t"                                                        \
        "v_mov_b32 %[y3lo], v60\n\t"                                                        \
        "v_mov_b32 %[y3hi], v61\n\t"                                                        \
        "v_mov_b32 %[esc], v42\n\t"                                                         \
        "v_mov_b32 %[esc2], v72\n\t"                                                        \
        "s_nop 1"


#define PLK_ASM2_OPERANDS                                                             \
        : [x0lo] "+v"(x0lo), [x0hi] "+v"(x0hi), [x1lo] "+v"(x1lo), [x1hi] "+v"(x1hi), \
          [x2lo] "+v"(x2lo), [x2hi] "+v"(x2hi), [x3lo] "+v"(x3lo), [x3hi] "+v"(x3hi), [esc] "=v"(esc), \
          [y0lo] "+v"(y0lo), [y0hi] "+v"(y0hi), [y1lo] "+v"(y1lo), [y1hi] "+v"(y1hi), \
          [y2lo] "+v"(y2lo), [y2hi] "+v"(y2hi), [y3lo] "+v"(y3lo), [y3hi] "+v"(y3hi), [esc2] "=v"(esc2) \
        : [ch] "v"(p.ch_first), [ch2] "v"(p.ch_first2), [clane] "v"(p.code_lane_addr), [clane2] "v"(p.code_lane_addr2), \
          [nshift] "v"(p.nibble_shift), [secaddr] "v"(p.second_code_addr), [secaddr2] "v"(p.second_code_addr2), \
          [ops] "s"(p.ops), [mstream] "s"(p.mstream),                                  \
          [tipbase] "s"(p.tip_lds_addr), [nchar32] "s"(p.nchar32), [tile] "s"(p.row_bytes),           \
          [cwidth] "s"(p.code_width), [firsttip] "s"(p.first_tip_addr)

#define PLK_ASM2_CLOBBERS_V                                                           \
          "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", \
          "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83"

struct FusedAsmParams2 {
    FusedAsmParams a;           /* site A and the shared fields */
    unsigned code_lane_addr2, second_code_addr2;
    int ch_first2;
};

template <int D>
__device__ __forceinline__ void fused_run_program_asm2(double (&x)[4], double (&y)[4], int &esc, int &esc2,
                                                        const FusedAsmParams2 &pp)
{
    struct { const void *ops, *mstream; unsigned tip_lds_addr, nchar32, row_bytes, code_width, first_tip_addr,
             code_lane_addr, nibble_shift, second_code_addr; int ch_first;
             unsigned code_lane_addr2, second_code_addr2; int ch_first2; } p;
    p.ops = pp.a.ops; p.mstream = pp.a.mstream; p.tip_lds_addr = pp.a.tip_lds_addr; p.nchar32 = pp.a.nchar32;
    p.row_bytes = pp.a.row_bytes; p.code_width = pp.a.code_width; p.first_tip_addr = pp.a.first_tip_addr;
    p.code_lane_addr = pp.a.code_lane_addr; p.nibble_shift = pp.a.nibble_shift; p.second_code_addr = pp.a.second_code_addr;
    p.ch_first = pp.a.ch_first; p.code_lane_addr2 = pp.code_lane_addr2; p.second_code_addr2 = pp.second_code_addr2;
    p.ch_first2 = pp.ch_first2;
    int x0lo = __double2loint(x[0]), x0hi = __double2hiint(x[0]), x1lo = __double2loint(x[1]), x1hi = __double2hiint(x[1]);
    int x2lo = __double2loint(x[2]), x2hi = __double2hiint(x[2]), x3lo = __double2loint(x[3]), x3hi = __double2hiint(x[3]);
    int y0lo = __double2loint(y[0]), y0hi = __double2hiint(y[0]), y1lo = __double2loint(y[1]), y1hi = __double2hiint(y[1]);
    int y2lo = __double2loint(y[2]), y2hi = __double2hiint(y[2]), y3lo = __double2loint(y[3]), y3hi = __double2hiint(y[3]);
    if constexpr (D <= 4) {
        asm volatile(PLK_ASM2_PROGRAM(PLK_ASM2_SLOTS_D4_POP, PLK_ASM2_SLOTS_D4_PUSH)
                     PLK_ASM2_OPERANDS : PLK_ASM_CLOBBERS_COMMON, PLK_ASM2_CLOBBERS_V, PLK_CLOBBER_A0_31, PLK_CLOBBER_A32_63);
    } else {
        asm volatile(PLK_ASM2_PROGRAM(PLK_ASM2_SLOTS_D8_POP, PLK_ASM2_SLOTS_D8_PUSH)
                     PLK_ASM2_OPERANDS : PLK_ASM_CLOBBERS_COMMON, PLK_ASM2_CLOBBERS_V, PLK_CLOBBER_A0_31, PLK_CLOBBER_A32_63,
                       PLK_CLOBBER_A64_127);
    }
    x[0] = __hiloint2double(x0hi, x0lo); x[1] = __hiloint2double(x1hi, x1lo);
    x[2] = __hiloint2double(x2hi, x2lo); x[3] = __hiloint2double(x3hi, x3lo);
    y[0] = __hiloint2double(y0hi, y0lo); y[1] = __hiloint2double(y1hi, y1lo);
    y[2] = __hiloint2double(y2hi, y2lo); y[3] = __hiloint2double(y3hi, y3lo);
}

#define PLK_TILE2 (2 * PLK_TILE)

/* staging, category loop and epilogue in C++, the program run in assembly; sites tid and tid + 256 of a 512-site tile */
template <int D>
__global__ __launch_bounds__(PLK_TILE) void k_ll_fused4_asm2(FusedAsmArgs aa)
{
    const FusedArgs &a = aa.f;
    extern __shared__ double lds_dyn[];
    double *tip_lds = lds_dyn;
    const int tip_doubles = a.ntips * a.nchar * 4;
    uint8_t *code_lds = reinterpret_cast<uint8_t *>(lds_dyn + tip_doubles);
    const long tile0 = (long)blockIdx.x * PLK_TILE2;
    const int tid = threadIdx.x;
    const int row_bytes = aa.pack4 ? PLK_TILE2 / 2 : PLK_TILE2;
    {
        const int ndw = a.nobs * (PLK_TILE2 / 4);
        for (int idx = tid; idx < ndw; idx += PLK_TILE) {
            const int row = idx >> 7, col = idx & 127;
            const uint32_t *src = reinterpret_cast<const uint32_t *>(a.codes + (size_t)a.obs_nodes[row] * a.Spad + tile0);
            const uint32_t q = src[col];
            if (aa.pack4) {
                const uint32_t packed = (q & 0xf) | ((q >> 4) & 0xf0) | ((q >> 8) & 0xf00) | ((q >> 12) & 0xf000);
                reinterpret_cast<uint16_t *>(code_lds)[row * (PLK_TILE2 / 4) + col] = (uint16_t)packed;
            } else {
                reinterpret_cast<uint32_t *>(code_lds)[idx] = q;
            }
        }
    }
    const PLK_AS4 double *prior = as_uniform(a.cat_prior);
    const PLK_AS4 double *rootw = as_uniform(a.root_w);

    FusedAsmParams2 p;
    p.a.ops = aa.words;
    p.a.tip_lds_addr = (unsigned)(size_t)tip_lds;
    p.a.nchar32 = (unsigned)a.nchar * 32u;
    p.a.row_bytes = (unsigned)row_bytes;
    p.a.code_width = aa.pack4 ? 4u : 8u;
    p.a.first_tip_addr = p.a.tip_lds_addr + (unsigned)aa.first_tip * p.a.nchar32;
    const unsigned col_a = aa.pack4 ? (unsigned)(tid >> 1) : (unsigned)tid;
    const unsigned col_b = col_a + (aa.pack4 ? PLK_TILE / 2 : PLK_TILE);
    p.a.code_lane_addr = (unsigned)(size_t)code_lds + col_a;
    p.code_lane_addr2 = (unsigned)(size_t)code_lds + col_b;
    p.a.nibble_shift = aa.pack4 ? (unsigned)(tid & 1) * 4u : 0u;
    p.a.second_code_addr = p.a.code_lane_addr + (unsigned)aa.second_row * (unsigned)row_bytes;
    p.second_code_addr2 = p.code_lane_addr2 + (unsigned)aa.second_row * (unsigned)row_bytes;

    double sum[2] = {0.0, 0.0};
    int Eexp[2] = {0, 0};
    bool have[2] = {false, false};
    for (int c = 0; c < a.C; c++) {
        __syncthreads();
        {
            const double2 *src = reinterpret_cast<const double2 *>(a.tip + (size_t)c * tip_doubles);
            double2 *dst = reinterpret_cast<double2 *>(tip_lds);
            for (int idx = tid; idx < tip_doubles / 2; idx += PLK_TILE) dst[idx] = src[idx];
        }
        __syncthreads();
        double x[4] = {1.0, 1.0, 1.0, 1.0}, y[4] = {1.0, 1.0, 1.0, 1.0};
        int esc[2] = {0, 0};
        p.a.mstream = a.PS + (size_t)c * (a.nmat + 1) * 16;
        p.a.ch_first = code_lds[col_a + aa.first_row * row_bytes];
        p.ch_first2 = code_lds[col_b + aa.first_row * row_bytes];
        fused_run_program_asm2<D>(x, y, esc[0], esc[1], p);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const double *v = j ? y : x;
            double lh;
            if (a.root_mode == PLK_ROOT_NONE) lh = ((v[0] + v[1]) + v[2]) + v[3];
            else if (a.root_mode == PLK_ROOT_UNIFORM) lh = (((v[0] + v[1]) + v[2]) + v[3]) * 0.25;
            else lh = fma(rootw[3], v[3], fma(rootw[2], v[2], fma(rootw[1], v[1], rootw[0] * v[0])));
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

#endif
