#!/usr/bin/env python3
"""VGPR liveness of one kernel in a gfx950 .s file (hipcc --save-temps / -S): backward dataflow over the kernel's basic
blocks at 32-bit register granularity; prints the maximum number of live VGPRs, where it occurs, and the live count at
every label.  Used to find which vectors are live together when a register-resident kernel spills.

  python tools/vgpr_liveness.py build/plk_engine-hip-amdgcn-amd-amdhsa-gfx950.s _Z10k_up_nodesILi20EEv9UpVecArgs"""
import re
import sys

RE_V = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
STORE = ("global_store", "scratch_store", "ds_write", "buffer_store", "flat_store", "ds_store")
NODEF = ("v_cmp", "v_cmpx", "v_readlane", "v_readfirstlane", "s_", "global_store", "scratch_store", "ds_write", "buffer_store",
         "flat_store", "ds_store", "v_nop")


def regs(tok):
    out = []
    for m in RE_V.finditer(tok):
        if m.group(3) is not None:
            out.append(int(m.group(3)))
        else:
            out.extend(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def main(path, kernel):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.split(";")[0].strip() == kernel + ":")
    body = []
    for i in range(start + 1, len(lines)):
        body.append((i + 1, lines[i]))
        if "s_endpgm" in lines[i]:
            break
    # instructions and labels
    ins = []          # (lineno, text, defs, uses, label or None, branch target or None, falls through)
    labels = {}
    for ln, t in body:
        t = t.split(";")[0].rstrip()
        if not t.strip():
            continue
        if re.match(r"^\.?[A-Za-z_][\w.$]*:", t.strip()) and not t.startswith("\t"):
            labels[t.strip()[:-1]] = len(ins)
            continue
        s = t.strip()
        if s.startswith("."):
            continue
        op = s.split()[0]
        ops = s[len(op):].split(",")
        defs, uses = [], []
        if op.startswith(NODEF):
            for o in ops:
                uses += regs(o)
        else:
            defs = regs(ops[0]) if ops else []
            for o in ops[1:]:
                uses += regs(o)
            if op.startswith("v_writelane") or op.startswith("v_mac") or op.startswith("v_fmac") or "accvgpr" in op:
                uses += defs
        tgt = None
        fall = True
        if op.startswith("s_cbranch"):
            tgt = ops[0].strip()
        elif op == "s_branch":
            tgt = ops[0].strip()
            fall = False
        elif op == "s_endpgm":
            fall = False
        ins.append((ln, s, set(defs), set(uses), tgt, fall))
    n = len(ins)
    live_in = [set() for _ in range(n + 1)]
    changed = True
    while changed:
        changed = False
        for i in range(n - 1, -1, -1):
            ln, s, d, u, tgt, fall = ins[i]
            out = set()
            if fall and i + 1 < n:
                out |= live_in[i + 1]
            if tgt is not None and tgt in labels:
                out |= live_in[labels[tgt]]
            new = (out - d) | u
            if new != live_in[i]:
                live_in[i] = new
                changed = True
    best = max(range(n), key=lambda i: len(live_in[i]))
    print("max live VGPRs: %d at line %d: %s" % (len(live_in[best]), ins[best][0], ins[best][1]))
    def ranges(s):
        s = sorted(s); out = []; a = None
        for x in s:
            if a is None: a = b = x
            elif x == b + 1: b = x
            else: out.append((a, b)); a = b = x
        if a is not None: out.append((a, b))
        return " ".join("v%d" % a if a == b else "v[%d:%d]" % (a, b) for a, b in out)
    print("live there:", ranges(live_in[best]))
    inv = {v: k for k, v in labels.items()}
    for i in range(n):
        if i in inv:
            print("%-14s line %6d live %3d  %s" % (inv[i], ins[i][0], len(live_in[i]), ranges(live_in[i]) if len(sys.argv) > 3 else ""))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
