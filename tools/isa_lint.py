#!/usr/bin/env python3
"""Static checks on the gfx950 ISA that hipcc produces for phyly_amd/csrc/plk_engine.hip.

The kernels contain two idioms whose safety the compiler cannot see, so it is checked on the generated code:

1. Scalar-cache pre-touch (plk_vec.h, plk_updown_vec.h): inline `s_load_dword sX, ...` whose result is never used and
   which is still in flight when the asm statement ends.  The load will write sX whenever it arrives.  Rule: between
   the end of such an asm block and the next `s_waitcnt ... lgkmcnt(0)` on every path, no instruction may write sX
   (otherwise a late load could replace a live value -- an op word, a pointer -- with matrix bytes).

2. AGPR stack (plk_fused4.h, plk_updown4.h): vectors parked in accumulation registers through inline
   v_accvgpr_write/read with fixed register numbers.  Rule: in those kernels the compiler itself must not touch any
   AGPR (no spills to AGPRs, no copies): every AGPR operand must sit inside an ASMSTART/ASMEND block.

Register spills and scratch use are reported as notes (performance, not safety).

usage: isa_lint.py file.s      exit status 0 = no violation; violations and notes are printed one per line
"""
import re
import sys

NO_DST = ("s_cmp", "s_branch", "s_cbranch", "s_waitcnt", "s_nop", "s_setpc", "s_endpgm", "s_barrier", "s_sleep",
          "s_bitcmp", "s_setprio", "s_sethalt", "s_trap", "s_dcache", "s_icache", "s_store", "s_buffer_store",
          "s_setreg", "s_set_gpr", "s_cbranch", "s_code_end", "s_waitcnt_depctr", "s_incperflevel", "s_decperflevel")
AGPR_KERNELS = re.compile(r"k_ll_fused4|k_down_fused4|k_ll_vec_rs|k_down_vec_rs")


def sgprs(tok):
    m = re.fullmatch(r"s(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def written(ins):
    """SGPRs an instruction writes (first operand of scalar ops, scalar destinations of vector ops)"""
    parts = ins.split(None, 1)
    if len(parts) < 2:
        return set()
    mn, ops = parts[0], [o.strip() for o in parts[1].split(",")]
    if mn.startswith(NO_DST):
        return set()
    if mn.startswith("s_") or mn.startswith(("v_readfirstlane", "v_readlane", "v_cmp", "v_cmpx")):
        out = sgprs(ops[0])
        if mn.startswith("s_swappc") or mn.startswith("s_getpc") or mn.startswith("s_call"):
            out |= sgprs(ops[0])
        return out
    # VOP3 carry-out / v_div_scale style second destination
    if mn.startswith(("v_add_co", "v_sub_co", "v_subrev_co", "v_addc_co", "v_subb_co", "v_div_scale", "v_mad_u64", "v_mad_i64")) and len(ops) > 1:
        return sgprs(ops[1])
    return set()


def split_kernels(text):
    kernels, name, body = {}, None, []
    for line in text.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
            kernels[name] = body
            continue
        if name is not None:
            if line.startswith(".Lfunc_end"):
                name = None
            else:
                body.append(line)
    return kernels


notes = []


def lint_kernel(name, lines):
    problems = []
    ins = []          # (text, in_asm)
    labels = {}
    in_asm = False
    for raw in lines:
        t = raw.split(";")[0].strip() if not raw.strip().startswith(";;#") else raw.strip()
        if raw.strip().startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if raw.strip().startswith(";;#ASMEND"):
            ins.append(("#asmend", False))
            in_asm = False
            continue
        if not t:
            continue
        m = re.match(r"^(\.L\w+):", t)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if t.startswith("."):
            continue
        ins.append((t, in_asm))
    # rule 2: AGPRs outside asm, scratch
    for t, ia in ins:
        if re.search(r"\bscratch_(load|store)", t) or re.search(r"buffer_(load|store)\w* .*\boffen\b.*s\[0:3\]", t):
            notes.append("note: %s: scratch access: %s" % (name, t))
            break
    if AGPR_KERNELS.search(name):
        for t, ia in ins:
            if not ia and (re.search(r"\ba\d+\b|\ba\[\d+:\d+\]", t) or "accvgpr" in t):
                problems.append("%s: compiler-generated AGPR use next to the inline AGPR stack: %s" % (name, t))
                break
    # rule 1: pending dummy scalar loads
    i = 0
    n = len(ins)
    while i < n:
        if not ins[i][1]:
            i += 1
            continue
        j = i
        pending = set()
        while j < n and ins[j][1]:
            t = ins[j][0]
            if t.startswith("s_load_") or t.startswith("s_buffer_load"):
                pending |= sgprs(t.split(None, 1)[1].split(",")[0].strip())
            elif t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
                pending = set()
            j += 1
        # j is at '#asmend' (or past the end)
        if pending:
            stack, seen = [j + 1], set()
            budget = 4000
            while stack and budget > 0:
                p = stack.pop()
                while p < n and budget > 0:
                    if p in seen:
                        break
                    seen.add(p)
                    budget -= 1
                    t, ia = ins[p]
                    if t == "#asmend":
                        p += 1
                        continue
                    if t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
                        break
                    if t.startswith("s_endpgm"):
                        break
                    w = written(t) & pending
                    if w and not (ia and t.startswith("s_load_")):
                        problems.append("%s: s%d is rewritten while an inline scalar load to it may still be in flight: %s"
                                        % (name, sorted(w)[0], t))
                        stack = []
                        break
                    m = re.match(r"s_c?branch\w*\s+(\.L\w+)", t)
                    if m and m.group(1) in labels:
                        stack.append(labels[m.group(1)])
                        if t.startswith("s_branch"):
                            break
                    p += 1
        i = j + 1
    return problems


def main(path):
    text = open(path).read()
    kernels = split_kernels(text)
    problems = []
    for name, lines in kernels.items():
        problems += lint_kernel(name, lines)
    # spills reported by the compiler's metadata
    for blk in re.finditer(r"- \.agpr_count:(?:.|\n)*?\.wavefront_size:\s+\d+", text):
        b = blk.group(0)
        nm = re.search(r"\.name:\s+(\S+)", b)
        sp = re.search(r"\.vgpr_spill_count:\s+(\d+)", b)
        ss = re.search(r"\.sgpr_spill_count:\s+(\d+)", b)
        ps = re.search(r"\.private_segment_fixed_size:\s+(\d+)", b)
        if nm and sp and int(sp.group(1)) > 0:
            notes.append("note: %s: %s VGPRs spilled, %s bytes of scratch" % (nm.group(1), sp.group(1), ps.group(1) if ps else "?"))
    for p in problems + notes:
        print(p)
    print("isa_lint: %d kernels, %d violations, %d notes" % (len(kernels), len(problems), len(notes)))
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
