#!/usr/bin/env python3
"""Generates phyly_amd/csrc/plk_fused4_c4_asm.h: the k = 4 fused traversal interpreter that carries FOUR rate
categories of a site through one pass of the traversal program (CDNA4 assembly, one inline-asm statement).

Why: in k_ll_fused4_asm (one category per pass) the scalar pipe is as busy as the vector pipe -- every op costs a
dispatch (s_swappc, compare chain, s_setpc: three taken branches) for 4..16 vector instructions, the op words and
the prefetch-chain bookkeeping are repeated per category, and each MATVEC waits for a matrix that was requested only
one op earlier.  With the four categories of GTR+G4 side by side every dispatch, op word, LDS address computation and
code fetch is shared by four times the vector work, and the matrices of the four categories stream through two SGPR
banks: while one bank feeds 16 FMAs the other one is being loaded.

Register map (inside the asm statement):
  (B0 = 20; v0..v19 stay with the compiler for the 5 input and 12 output operands: 112 VGPRs + 128 AGPRs, 2 waves/SIMD)
  v[B0:B0+31]      x[c][i]: vector under construction, category c = 0..3, component i = 0..3 (pair B0 + 8c + 2i)
  v[B0+32:B0+39]   T: product temporaries          v[B0+84:B0+91] U: second temporary set (stack pops)
  v[B0+40:B0+71]   prefetched tip values of the next observation, [c][i]
  v[B0+72] temp, +73 raw code byte of the next observation, +74..+77 scale exponents per category, +78, +79 temps,
  +80 LDS address of this lane's code column, +81 nibble shift, +82 LDS address (per category)
  uniform parameters arrive in the lanes of ONE input VGPR (lane i = parameter i, v_readlane): program and matrix
  stream pointers, LDS addresses and strides, the root weights
  a[(d*4 + c)*8 + r]  stack slot d of category c (D = 4 slots: 128 accumulation registers)
  s[36:67] matrix bank A, s[4:35] matrix bank B
  s[68:75] current op block, s[76:83] next op block, s[84:85] program pointer, s[86:87] matrix stream pointer,
  s[88:89] return address, s[90:91] dispatcher, s92 LDS address of the tip tables, s93 nchar * 32, s94 bytes per staged
  code row, s95 = -1022, s96 op word, s97..s99 temps, s100 code field width, s101 bytes per category tip table
Matrix stream: [matrix][category][16 doubles] (transposed P), 512 bytes per MATVEC, one spare matrix at the end.
Program words: the format of plk_fused4_asm.h (plk_program.h builds and checks it).

    python tools/gen_fused4_c4.py > phyly_amd/csrc/plk_fused4_c4_asm.h
"""

L = []
NC = 4                  # categories per pass (set per generated variant)
B0 = 20                 # first fixed VGPR: v0 .. v19 are left to the compiler's operands (5 inputs, 3 NC outputs)
XB = TB_ = TVB = MISC = UB = V_TMP = V_CODE = V_EXP = V_T1 = V_T2 = V_CLANE = V_NSH = V_ADDR = NV = 0


def set_nc(nc):
    """register map for nc categories per pass: x (8 nc), T (8), tip values (8 nc), misc (12), U (8)"""
    global NC, XB, TB_, TVB, MISC, UB, V_TMP, V_CODE, V_EXP, V_T1, V_T2, V_CLANE, V_NSH, V_ADDR, NV
    NC = nc
    XB = B0
    TB_ = XB + 8 * nc
    TVB = TB_ + 8
    MISC = TVB + 8 * nc
    UB = MISC + 12
    NV = UB + 8 - B0
    V_TMP, V_CODE, V_EXP = MISC, MISC + 1, MISC + 2
    V_T1, V_T2, V_CLANE, V_NSH, V_ADDR = MISC + 6, MISC + 7, MISC + 8, MISC + 9, MISC + 10


def e(s):
    L.append(s)


def X(c, i):
    r = XB + 8 * c + 2 * i
    return "v[%d:%d]" % (r, r + 1)


def Xhi(c, i):
    return "v%d" % (XB + 8 * c + 2 * i + 1)


def T(i, base=None):
    base = TB_ if base is None else base
    return "v[%d:%d]" % (base + 2 * i, base + 2 * i + 1)


def TV(c, i):
    r = TVB + 8 * c + 2 * i
    return "v[%d:%d]" % (r, r + 1)


def S(b, n):
    return "s[%d:%d]" % (b + 2 * n, b + 2 * n + 1)


BANK = (36, 4)


def load_bank(bank, off):
    b = BANK[bank]
    e("s_load_dwordx16 s[%d:%d], s[86:87], 0x%x" % (b, b + 15, off))
    e("s_load_dwordx16 s[%d:%d], s[86:87], 0x%x" % (b + 16, b + 31, off + 0x40))


def matvec(c, bank):
    b = BANK[bank]
    for i in range(4):
        e("v_mul_f64 %s, %s, %s" % (T(i), S(b, i), X(c, 0)))
    for j in (1, 2):
        for i in range(4):
            e("v_fma_f64 %s, %s, %s, %s" % (T(i), S(b, 4 * j + i), X(c, j), T(i)))
    for i in range(4):
        e("v_fma_f64 %s, %s, %s, %s" % (X(c, i), S(b, 12 + i), X(c, 3), T(i)))


def tip_prefetch(first):
    """issue the LDS reads of the next observation's tip values (4 categories) and of the code after next"""
    if first:
        e("v_bfe_u32 v%d, %%[ch], v%d, s100" % (V_T1, V_NSH))
        e("v_lshl_add_u32 v%d, v%d, 5, s99" % (V_ADDR, V_T1))          # s99 = first_tip_addr
    else:
        e("s_bfe_u32 s98, s96, 0xd0003")
        e("s_mul_i32 s98, s98, s93")
        e("s_add_u32 s98, s98, s92")
        e("s_lshr_b32 s99, s96, 16")
        e("s_mul_i32 s99, s99, s94")
        e("v_bfe_u32 v%d, v%d, v%d, s100" % (V_T1, V_CODE, V_NSH))
        e("v_lshl_add_u32 v%d, v%d, 5, s98" % (V_ADDR, V_T1))
    for c in range(NC):
        r = TVB + 8 * c
        e("ds_read_b128 v[%d:%d], v%d" % (r, r + 3, V_ADDR))
        e("ds_read_b128 v[%d:%d], v%d offset:16" % (r + 4, r + 7, V_ADDR))
        if c < NC - 1:
            e("v_add_u32 v%d, s101, v%d" % (V_ADDR, V_ADDR))
    if first:
        e("ds_read_u8 v%d, %%[secaddr]" % V_CODE)
    else:
        e("v_add_u32 v%d, s99, v%d" % (V_T1, V_CLANE))
        e("ds_read_u8 v%d, v%d" % (V_CODE, V_T1))


def gen(D):
    global L
    L = []
    # ---- prologue
    for c in range(NC):
        for i in range(4):
            r = XB + 8 * c + 2 * i
            e("v_mov_b32 v%d, 0" % r)
            e("v_mov_b32 v%d, 0x3ff00000" % (r + 1))
    for c in range(NC):
        e("v_mov_b32 v%d, 0" % (V_EXP + c))
    e("v_mov_b32 v%d, %%[clane]" % V_CLANE)
    e("v_mov_b32 v%d, %%[nshift]" % V_NSH)
    # uniform parameters: lane i of %[pv] (FusedC4Params::lane order)
    for lane, sreg in enumerate((84, 85, 86, 87, 92, 93, 94, 100, 101, 99)):
        e("v_readlane_b32 s%d, %%[pv], %d" % (sreg, lane))
    e("s_movk_i32 s95, 0xfc02")
    e("s_load_dwordx8 s[68:75], s[84:85], 0x0")
    load_bank(0, 0x0)
    load_bank(1, 0x80)
    tip_prefetch(True)
    e("s_getpc_b64 s[90:91]")
    e(".Lpcref_%=:")
    e("s_add_u32 s90, s90, .Ldispatch_%=-.Lpcref_%=")
    e("s_addc_u32 s91, s91, 0")
    e("s_waitcnt lgkmcnt(0)")
    # ---- block loop
    e(".Lblock_%=:")
    e("s_load_dwordx8 s[76:83], s[84:85], 0x20")
    e("s_add_u32 s84, s84, 32")
    e("s_addc_u32 s85, s85, 0")
    for k in range(8):
        e("s_mov_b32 s96, s%d" % (68 + k))
        e("s_swappc_b64 s[88:89], s[90:91]")
    e("s_waitcnt lgkmcnt(0)")
    for k in range(0, 8, 2):
        e("s_mov_b64 s[%d:%d], s[%d:%d]" % (68 + k, 69 + k, 76 + k, 77 + k))
    e("s_branch .Lblock_%=")
    # ---- dispatcher
    e(".Ldispatch_%=:")
    e("s_and_b32 s97, s96, 7")
    for code, label in ((2, "matvec"), (5, "tipmul_nw")):
        e("s_cmp_eq_u32 s97, %d" % code)
        e("s_cbranch_scc1 .L%s_%%=" % label)
    e("s_cmp_lt_u32 s97, 2")
    e("s_cbranch_scc1 .Ltip_%=")
    for code, label in ((4, "pop"), (3, "push"), (6, "scale")):
        e("s_cmp_eq_u32 s97, %d" % code)
        e("s_cbranch_scc1 .L%s_%%=" % label)
    e("s_branch .Ldone_%=")
    # ---- MATVEC: banks A, B hold categories 0, 1 of this matrix on entry and of the next matrix on exit
    e(".Lmatvec_%=:")
    e("s_waitcnt lgkmcnt(0)")
    if NC == 4:
        matvec(0, 0)
        load_bank(0, 0x100)
        matvec(1, 1)
        e("s_waitcnt lgkmcnt(0)")
        load_bank(1, 0x180)
        matvec(2, 0)
        e("s_waitcnt lgkmcnt(0)")
        load_bank(0, 0x200)
        matvec(3, 1)
        load_bank(1, 0x280)
        e("s_add_u32 s86, s86, 0x200")
    else:
        # two categories: both banks are refilled for the NEXT matrix, a whole op ahead of their use
        matvec(0, 0)
        load_bank(0, 0x100)
        matvec(1, 1)
        load_bank(1, 0x180)
        e("s_add_u32 s86, s86, 0x100")
    e("s_addc_u32 s87, s87, 0")
    e("s_setpc_b64 s[88:89]")
    # ---- TIP_SET / TIP_MUL
    e(".Ltip_%=:")
    e("s_waitcnt lgkmcnt(0)")
    e("s_cmp_eq_u32 s97, 0")
    e("s_cbranch_scc1 .Ltipset_%=")
    e(".Ltipmul_nw_%=:")
    for c in range(NC):
        for i in range(4):
            e("v_mul_f64 %s, %s, %s" % (X(c, i), X(c, i), TV(c, i)))
    e("s_branch .Ltipnext_%=")
    e(".Ltipset_%=:")
    for c in range(NC):
        for i in range(4):
            e("v_mov_b64 %s, %s" % (X(c, i), TV(c, i)))
    e(".Ltipnext_%=:")
    tip_prefetch(False)
    e("s_setpc_b64 s[88:89]")
    # ---- POPMUL d / PUSH d
    e(".Lpop_%=:")
    e("s_bfe_u32 s97, s96, 0xd0003")
    for d in range(D - 1):
        e("s_cmp_eq_u32 s97, %d" % d)
        e("s_cbranch_scc1 .Lpop%d_%%=" % d)
    e("s_branch .Lpop%d_%%=" % (D - 1))
    for d in range(D):
        e(".Lpop%d_%%=:" % d)
        for c in range(NC):
            tb = TB_ if c % 2 == 0 else UB
            for r in range(8):
                e("v_accvgpr_read_b32 v%d, a%d" % (tb + r, (d * NC + c) * 8 + r))
            if c >= 1:      # multiply the previous category while this one's reads complete
                pb = TB_ if (c - 1) % 2 == 0 else UB
                for i in range(4):
                    e("v_mul_f64 %s, %s, %s" % (X(c - 1, i), X(c - 1, i), T(i, pb)))
        e("s_nop 1")
        for i in range(4):
            e("v_mul_f64 %s, %s, %s" % (X(NC - 1, i), X(NC - 1, i), T(i, UB)))     # NC even: the last category used U
        e("s_setpc_b64 s[88:89]")
    e(".Lpush_%=:")
    e("s_bfe_u32 s97, s96, 0xd0003")
    for d in range(D - 1):
        e("s_cmp_eq_u32 s97, %d" % d)
        e("s_cbranch_scc1 .Lpush%d_%%=" % d)
    e("s_branch .Lpush%d_%%=" % (D - 1))
    for d in range(D):
        e(".Lpush%d_%%=:" % d)
        for c in range(NC):
            for r in range(8):
                e("v_accvgpr_write_b32 a%d, v%d" % ((d * NC + c) * 8 + r, XB + 8 * c + r))
        e("s_setpc_b64 s[88:89]")
    # ---- SCALE
    e(".Lscale_%=:")
    for c in range(NC):
        e("v_max_u32 v%d, %s, %s" % (V_T1, Xhi(c, 0), Xhi(c, 1)))
        e("v_max3_u32 v%d, %s, %s, v%d" % (V_T1, Xhi(c, 2), Xhi(c, 3), V_T1))
        e("v_lshrrev_b32 v%d, 20, v%d" % (V_T1, V_T1))
        e("v_sub_u32 v%d, 0x3fe, v%d" % (V_T2, V_T1))
        for i in range(4):
            e("v_ldexp_f64 %s, %s, v%d" % (X(c, i), X(c, i), V_T2))
        e("v_add3_u32 v%d, v%d, v%d, s95" % (V_EXP + c, V_EXP + c, V_T1))
    e("s_setpc_b64 s[88:89]")
    # ---- epilogue
    e(".Ldone_%=:")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    # root expectation of every category (src/model.c:283-350): lh_c = sum_i w_i x_ci with the root weights from
    # lanes 10..17 of the parameter register (ones for "no prior", 1/4 for uniform)
    for k in range(8):
        e("v_readlane_b32 s%d, %%[pv], %d" % (36 + k, 10 + k))
    for c in range(NC):
        e("v_mul_f64 %%[lh%d], s[36:37], %s" % (c, X(c, 0)))
        for i in (1, 2, 3):
            e("v_fma_f64 %%[lh%d], s[%d:%d], %s, %%[lh%d]" % (c, 36 + 2 * i, 37 + 2 * i, X(c, i), c))
        e("v_mov_b32 %%[e%d], v%d" % (c, V_EXP + c))
    e("s_nop 1")
    text = "\n".join('        "%s\\n%s"' % (l, "" if l.endswith(":") else "\\t") for l in L[:-1]) + '\n        "%s"' % L[-1]
    outs = ", ".join('[lh%d] "=&v"(lh[%d])' % (c, c) for c in range(NC))
    outs += ", " + ", ".join('[e%d] "=&v"(esc[%d])' % (c, c) for c in range(NC))
    ins = ", ".join('[%s] "v"(p.%s)' % (n, n) for n in ("ch", "clane", "nshift", "secaddr", "pv"))
    clob = ['"memory"', '"scc"', '"vcc"'] + ['"v%d"' % r for r in range(B0, B0 + NV)] + ['"s%d"' % r for r in range(4, 102)] + \
           ['"a%d"' % r for r in range(8 * NC * D)]
    rows = []
    line = "          "
    for cbit in clob:
        if len(line) + len(cbit) + 2 > 118:
            rows.append(line.rstrip())
            line = "          "
        line += cbit + ", "
    rows.append(line.rstrip().rstrip(","))
    return text, outs, ins, "\n".join(rows)


print("""/* GENERATED by tools/gen_fused4_c4.py -- do not edit; the generator holds the commentary. */
#ifndef PLK_FUSED4_C4_ASM_H
#define PLK_FUSED4_C4_ASM_H

struct FusedC4Params {
    int ch;                       /* raw code byte of the first observation op */
    unsigned clane, nshift, secaddr;
    unsigned pv;                  /* lane i holds uniform parameter i: 0 ops lo, 1 ops hi, 2 matrix stream lo, 3 hi, 4 LDS address
                                     of the tip tables, 5 nchar * 32, 6 bytes per staged code row, 7 code field width, 8 bytes
                                     per category tip table, 9 LDS address of the first observation's tip slot, 10..17 the four
                                     root weights (lo, hi) */
};

template <int NC, int D>
__device__ __forceinline__ void fused_run_program_cn(double (&lh)[NC], int (&esc)[NC], const FusedC4Params &p);
""")
for nc, D in ((4, 4), (2, 4), (2, 8)):
    set_nc(nc)
    text, outs, ins, clob = gen(D)
    print("""/* %d categories per pass, %d stack slots: root expectations w . x_c and scale exponents out */
template <>
__device__ __forceinline__ void fused_run_program_cn<%d, %d>(double (&lh)[%d], int (&esc)[%d], const FusedC4Params &p)
{
    asm volatile(
%s
        : %s
        : %s
        : %s);
}
""" % (nc, D, nc, D, nc, nc, text, outs, ins, clob))
print("#endif")
