#!/usr/bin/env python3
"""Generates phyly_amd/csrc/plk_fused4_v4_asm.h: the CDNA4 assembly text of the k = 4 pair-table interpreter with TWO
sites per lane (k_ll_fused4_v4, plk_fused4_v4.h).

Why a generator: every handler exists once per stack slot and every vector instruction twice (site A, site B,
interleaved so that the two independent dependency chains alternate in the issue stream); written as C macros that is
~700 lines of hand-numbered registers.  The register map, the handler numbering and the op-word format are defined
here once and the header is emitted from them.  Run `python tools/gen_fused4_v4.py` after changing this file; the header
is committed, the build does not depend on Python.

Op format (built by plk_fused_v4_words in plk_program.h from the pair-table program): 64 bits per op, blocks of 8 ops
(7 ops of the program and one REFILL op, see below),
  lo dword  handler index * 512 (byte offset of the handler in the table; the table is 32 KB aligned)
  hi dword  observation ops: bits 31:16 = LDS offset / 32 of the table the NEXT observation op reads, bits 15:0 = LDS
            offset / 64 of the staged code row of the observation op AFTER the next (both relative to LDS address 0)
Handler indices: 0 TIP_SET, 1 TIP_MUL, 2 MATVEC, 3 TIP_MUL without wait, 4 MATVEC + TIP_MUL, 6 SCALE, 7 END,
8 + d PUSH slot d, 12 + d TIP_SET + POPMUL slot d, 16 + d POPMUL slot d, 20 + d TIP_SET + PUSH slot d,
24 + d MATVEC + PUSH slot d, 28 + d MATVEC + POPMUL slot d (d < 4), 32 REFILL_A, 33 REFILL_B.

Dispatch is THREADED: there is no interpreter loop.  Two op blocks (16 ops, 32 dwords) sit in s[68:99] as a ring that
M0 indexes; every handler ends in NEXT = { s_movrels_b64 op, ring[M0]; M0 += 2; target = table | op.lo; s_setpc target },
one taken jump per op instead of two (call + return).  The last op of a block is a REFILL: executed when its block has
been consumed, it waits for the other block (requested one block earlier), requests the block after that into its own
half of the ring (REFILL_B also wraps M0 to 0) and goes on.  The program buffer therefore ends with two spare blocks.

Registers (all named in the clobber list of the asm statement):
  v[24:31] xA  v[32:39] xB   vectors under construction (site A = lane's first site, site B = HALF sites further)
  v[40:47] tA  v[48:55] tB   product accumulators
  v[56:63] pA  v[64:71] pB   prefetched tip / pair-table value of the next observation op
  v72 v73 value addresses   v74 v75 code bytes of the next observation op   v76 code address   v77 LDS address of the
  lane's byte in row 0 (site A; site B at +HALF)   v78 v79 scale exponents   v80 v81 temporaries
  v[82:145] stack: slot d = A v[82+16d : 89+16d], B v[90+16d : 97+16d]
  s[36:67] current matrix (transposed)   s[68:99] ring of two op blocks, indexed by M0 (dwords)
  s[20:21] program pointer (next block to request)   s[22:23] the op being executed (s23 = its fields)
  s[24:25] jump target (s25 = high half of the table address, constant)   s26 low half of the table address
  s27 = -1022   s28 s29 temporaries   s[30:31] matrix stream pointer
"""
import os

HS = 512                      # bytes per handler slot
XA, XB, TA, TB, PA, PB = 24, 32, 40, 48, 56, 64
STACK0 = 82


def pair(r):
    return "v[%d:%d]" % (r, r + 1)


def slot_regs(d, site):
    base = STACK0 + 16 * d + 8 * site
    return [base + 2 * i for i in range(4)]


def matvec(dst_a=None, dst_b=None):
    """x = M x for both sites, rows interleaved A/B; dst_*: registers of the result (default in place)"""
    out = ["s_waitcnt lgkmcnt(0)"]
    for col in range(4):
        for row in range(4):
            s = 36 + 8 * col + 2 * row
            for (x, t, dst) in ((XA, TA, dst_a), (XB, TB, dst_b)):
                xin = pair(x + 2 * col)
                acc = pair(t + 2 * row)
                if col == 0:
                    out.append("v_mul_f64 %s, s[%d:%d], %s" % (acc, s, s + 1, xin))
                elif col < 3:
                    out.append("v_fma_f64 %s, s[%d:%d], %s, %s" % (acc, s, s + 1, xin, acc))
                else:
                    d = pair((dst[row]) if dst else x + 2 * row)
                    out.append("v_fma_f64 %s, s[%d:%d], %s, %s" % (d, s, s + 1, xin, acc))
    out += ["s_add_u32 s30, s30, 0x80", "s_addc_u32 s31, s31, 0",
            "s_load_dwordx16 s[36:51], s[30:31], 0x0", "s_load_dwordx16 s[52:67], s[30:31], 0x40"]
    return out


def tipmul():
    out = []
    for i in range(4):
        out.append("v_mul_f64 %s, %s, %s" % (pair(XA + 2 * i), pair(XA + 2 * i), pair(PA + 2 * i)))
        out.append("v_mul_f64 %s, %s, %s" % (pair(XB + 2 * i), pair(XB + 2 * i), pair(PB + 2 * i)))
    return out


def tipset():
    out = []
    for i in range(4):
        out.append("v_mov_b64 %s, %s" % (pair(XA + 2 * i), pair(PA + 2 * i)))
        out.append("v_mov_b64 %s, %s" % (pair(XB + 2 * i), pair(PB + 2 * i)))
    return out


def tipnext():
    return ["s_lshr_b32 s28, s23, 16", "s_and_b32 s29, s23, 0xffff",
            "v_add_lshl_u32 v72, v74, s28, 5", "v_add_lshl_u32 v73, v75, s28, 5",
            "ds_read_b128 v[56:59], v72", "ds_read_b128 v[60:63], v72 offset:16",
            "ds_read_b128 v[64:67], v73", "ds_read_b128 v[68:71], v73 offset:16",
            "v_lshl_add_u32 v76, s29, 6, v77",
            "ds_read_u8 v74, v76", "@HALF ds_read_u8 v75, v76"]


def push(d):
    out = []
    for i in range(4):
        out.append("v_mov_b64 %s, %s" % (pair(slot_regs(d, 0)[i]), pair(XA + 2 * i)))
        out.append("v_mov_b64 %s, %s" % (pair(slot_regs(d, 1)[i]), pair(XB + 2 * i)))
    return out


def popmul(d):
    out = []
    for i in range(4):
        out.append("v_mul_f64 %s, %s, %s" % (pair(XA + 2 * i), pair(XA + 2 * i), pair(slot_regs(d, 0)[i])))
        out.append("v_mul_f64 %s, %s, %s" % (pair(XB + 2 * i), pair(XB + 2 * i), pair(slot_regs(d, 1)[i])))
    return out


def setpop(d):
    """x = prefetched value o slot d (TIP_SET followed by POPMUL: no copy of the value)"""
    out = []
    for i in range(4):
        out.append("v_mul_f64 %s, %s, %s" % (pair(XA + 2 * i), pair(PA + 2 * i), pair(slot_regs(d, 0)[i])))
        out.append("v_mul_f64 %s, %s, %s" % (pair(XB + 2 * i), pair(PB + 2 * i), pair(slot_regs(d, 1)[i])))
    return out


def setpush(d):
    """slot d = prefetched value (TIP_SET followed by PUSH: the vector under construction is dead after a PUSH)"""
    out = []
    for i in range(4):
        out.append("v_mov_b64 %s, %s" % (pair(slot_regs(d, 0)[i]), pair(PA + 2 * i)))
        out.append("v_mov_b64 %s, %s" % (pair(slot_regs(d, 1)[i]), pair(PB + 2 * i)))
    return out


def scale():
    out = []
    for (x, tmp, esc) in ((XA, 80, 78), (XB, 81, 79)):
        out += ["v_max_u32 v%d, v%d, v%d" % (tmp, x + 1, x + 3), "v_max3_u32 v%d, v%d, v%d, v%d" % (tmp, x + 5, x + 7, tmp),
                "v_lshrrev_b32 v%d, 20, v%d" % (tmp, tmp), "v_sub_u32 v72, 0x3fe, v%d" % tmp]
        out += ["v_ldexp_f64 %s, %s, v72" % (pair(x + 2 * i), pair(x + 2 * i)) for i in range(4)]
        out.append("v_add3_u32 v%d, v%d, v%d, s27" % (esc, esc, tmp))
    return out


# NEXT: fetch the op the ring index points at and jump to its handler (s_movrels reads M0: M0 is written after it, never
# within one instruction before it)
RET = ["s_movrels_b64 s[22:23], s[68:69]", "s_addk_i32 m0, 0x2", "s_or_b32 s24, s26, s22", "s_setpc_b64 s[24:25]"]
NSLOTS = 64
REFILL_A, REFILL_B = 32, 33


def handlers():
    h = {}
    h[0] = ["s_waitcnt lgkmcnt(0)"] + tipset() + tipnext() + RET
    h[1] = ["s_waitcnt lgkmcnt(0)"] + tipmul() + tipnext() + RET
    h[2] = matvec() + RET
    h[3] = tipmul() + tipnext() + RET
    h[4] = matvec() + tipmul() + tipnext() + RET
    h[6] = scale() + RET
    h[7] = ["s_branch .Ldone_%="]
    for d in range(4):
        h[8 + d] = push(d) + RET
        h[12 + d] = ["s_waitcnt lgkmcnt(0)"] + setpop(d) + tipnext() + RET
        h[16 + d] = popmul(d) + RET
        h[20 + d] = ["s_waitcnt lgkmcnt(0)"] + setpush(d) + tipnext() + RET
        h[24 + d] = matvec(slot_regs(d, 0), slot_regs(d, 1)) + RET
        h[28 + d] = matvec() + popmul(d) + RET
    adv = ["s_add_u32 s20, s20, 0x40", "s_addc_u32 s21, s21, 0"]
    h[REFILL_A] = ["s_waitcnt lgkmcnt(0)", "s_load_dwordx16 s[68:83], s[20:21], 0x0"] + adv + RET
    h[REFILL_B] = ["s_mov_b32 m0, 0", "s_waitcnt lgkmcnt(0)", "s_load_dwordx16 s[84:99], s[20:21], 0x0"] + adv + RET
    return h


def emit():
    L = []           # (text, is_label)

    def ins(lines):
        for t in lines:
            L.append(t)

    # ---- prologue ----
    pro = []
    for i in range(4):
        pro += ["v_mov_b32 v%d, 0" % (XA + 2 * i), "v_mov_b32 v%d, 0x3ff00000" % (XA + 2 * i + 1),
                "v_mov_b32 v%d, 0" % (XB + 2 * i), "v_mov_b32 v%d, 0x3ff00000" % (XB + 2 * i + 1)]
    pro += ["v_mov_b32 v78, 0", "v_mov_b32 v79, 0", "v_mov_b32 v77, %[clane]",
            "s_mov_b64 s[20:21], %[ops]", "s_mov_b64 s[30:31], %[mstream]", "s_movk_i32 s27, 0xfc02", "s_mov_b32 m0, 0",
            "s_load_dwordx16 s[68:83], s[20:21], 0x0", "s_load_dwordx16 s[84:99], s[20:21], 0x40",
            "s_add_u32 s20, s20, 0x80", "s_addc_u32 s21, s21, 0",
            "s_load_dwordx16 s[36:51], s[30:31], 0x0", "s_load_dwordx16 s[52:67], s[30:31], 0x40",
            # prefetch chain start: codes of the first observation, its values, codes of the second
            "v_lshl_add_u32 v76, %[z0], 6, v77", "ds_read_u8 v74, v76", "@HALF ds_read_u8 v75, v76",
            "s_getpc_b64 s[24:25]", ".Lpcref_%=:", "s_add_u32 s26, s24, .Lh0_%=-.Lpcref_%=", "s_addc_u32 s25, s25, 0",
            "s_waitcnt lgkmcnt(0)",
            "v_add_lshl_u32 v72, v74, %[y0], 5", "v_add_lshl_u32 v73, v75, %[y0], 5",
            "ds_read_b128 v[56:59], v72", "ds_read_b128 v[60:63], v72 offset:16",
            "ds_read_b128 v[64:67], v73", "ds_read_b128 v[68:71], v73 offset:16",
            "v_lshl_add_u32 v76, %[z1], 6, v77", "ds_read_u8 v74, v76", "@HALF ds_read_u8 v75, v76",
            "s_waitcnt lgkmcnt(0)"]
    ins(pro)
    ins(RET)           # the first op
    # ---- handlers ----
    h = handlers()
    ins([".p2align 15", ".Lh0_%=:"])
    for idx in range(NSLOTS):
        if idx:
            ins([".p2align 9"])
        ins(h.get(idx, ["s_branch .Ldone_%="]))
    ins([".p2align 9", ".Ldone_%=:", "s_waitcnt vmcnt(0) lgkmcnt(0)"])
    # root dot product lh = w . x as one fma chain (weights 1 for no prior, 1/4 for the uniform prior: the same bits as the
    # plain sum and the scaled sum), so that only lh and the exponent leave the statement
    for (x, t) in ((XA, TA), (XB, TB)):
        ins(["v_mul_f64 %s, %%[w0], %s" % (pair(t), pair(x))])
        for i in range(1, 4):
            ins(["v_fma_f64 %s, %%[w%d], %s, %s" % (pair(t), i, pair(x + 2 * i), pair(t))])
    ins(["v_mov_b32 %%[al], v%d" % TA, "v_mov_b32 %%[ah], v%d" % (TA + 1), "v_mov_b32 %%[bl], v%d" % TB, "v_mov_b32 %%[bh], v%d" % (TB + 1)])
    ins(["v_mov_b32 %[ea], v78", "v_mov_b32 %[eb], v79", "s_nop 1"])
    return L


def c_string(lines):
    out = []
    for t in lines:
        if t.startswith("@HALF "):
            body = t[6:]
            out.append('        "%s offset:" #HALF "\\n\\t" \\' % body)
        elif t.endswith(":") or t.startswith(".p2align"):
            out.append('        "%s\\n" \\' % t)
        else:
            out.append('        "%s\\n\\t" \\' % t)
    return "\n".join(out)


def main():
    lines = emit()
    # size check: every handler must fit its slot (also checked on the object by tools/asm_layout_check.py)
    sizes = {"v_mul_f64": 8, "v_fma_f64": 8, "v_mov_b64": 4, "v_mov_b32": 4, "v_max_u32": 4, "v_max3_u32": 8, "v_lshrrev_b32": 4,
             "v_sub_u32": 8, "v_ldexp_f64": 8, "v_add3_u32": 8, "v_add_lshl_u32": 8, "v_lshl_add_u32": 8, "ds_read_b128": 8,
             "ds_read_u8": 8, "s_waitcnt": 4, "s_add_u32": 8, "s_addc_u32": 4, "s_load_dwordx16": 8, "s_setpc_b64": 4,
             "s_lshr_b32": 4, "s_and_b32": 8, "s_branch": 4, "s_movrels_b64": 4, "s_addk_i32": 4, "s_or_b32": 4, "s_mov_b32": 4}
    for idx, body in handlers().items():
        n = sum(sizes[b.replace("@HALF ", "").split()[0]] for b in body)
        assert n <= HS, (idx, n)
    vregs = ["v%d" % r for r in range(24, STACK0 + 64)]
    sregs = ["s%d" % r for r in range(20, 100)] + ["m0"]
    hdr = '''/* GENERATED by tools/gen_fused4_v4.py -- do not edit; change the generator and run it again.
 *
 * Assembly text of k_ll_fused4_v4 (plk_fused4_v4.h): the k = 4 pair-table interpreter with two sites per lane.  Op
 * format, handler numbering and the register map are documented in the generator. */
#ifndef PLK_FUSED4_V4_ASM_H
#define PLK_FUSED4_V4_ASM_H

#define PLK_V4_HANDLER_BYTES %d
#define PLK_V4_HANDLER_SLOTS %d
#define PLK_V4_REFILL_A %d
#define PLK_V4_REFILL_B %d

/* HALF: sites between a lane's two sites = workgroup size (an integer literal: it is pasted into ds_read offsets) */
#define PLK_V4_PROGRAM(HALF) \\
%s
        ""

#define PLK_V4_CLOBBERS "memory", "scc", "vcc", \\
        %s, \\
        %s

#endif
''' % (HS, NSLOTS, REFILL_A, REFILL_B, c_string(lines), ", ".join('"%s"' % v for v in vregs), ", ".join('"%s"' % s for s in sregs))
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "phyly_amd", "csrc", "plk_fused4_v4_asm.h")
    with open(path, "w") as f:
        f.write(hdr)
    print("wrote", path, "(%d asm lines)" % len(lines))


if __name__ == "__main__":
    main()
