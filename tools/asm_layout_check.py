#!/usr/bin/env python3
"""Layout check of the k = 4 assembly interpreter's handler table (phyly_amd/csrc/plk_fused4_asm.h), part of the build.

The interpreter jumps to `table + 256 * handler_index`; the table starts on an 8 KB boundary and every handler must fit
its 256-byte slot (the MATVEC + TIP_MUL pair handler owns slots 4 and 5: it starts on a 512-byte boundary and the handler after it is aligned to 512 bytes).  Nothing in the source enforces that: a handler
that grows past its slot would still assemble, and the next index would land in the middle of it.  This script
disassembles the device object (llvm-objdump) and checks, for every k_ll_fused4_asm instantiation, that
  - the table (first instruction after the run of s_nop padding that ends on an 8 KB boundary) exists,
  - at every 256-byte boundary inside the table the previous instruction is padding (s_nop) or an unconditional
    transfer (s_setpc_b64 / s_branch), i.e. no handler runs across a boundary -- except between slots 4 and 5,
  - 32 slots are present (64 slots of 512 bytes on a 32 KB boundary for the two-sites-per-lane interpreter
    k_ll_fused4_v4, whose handlers end in a threaded dispatch: s_setpc_b64 again).

  python tools/asm_layout_check.py build/plk_engine-hip-amdgcn-amd-amdhsa-gfx950.o"""
import re
import subprocess
import sys

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
# kernel-name substring -> (bytes per handler slot, table alignment, slot that may run on into the next one or None, slots)
FAMILIES = {"k_ll_fused4_asm": (256, 0x2000, 4, 32), "k_ll_fused4_v4": (512, 0x8000, None, 64)}


def kernels(obj):
    out = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", obj], capture_output=True, text=True, check=True).stdout
    cur, body = None, {}
    for line in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            body[cur] = []
            continue
        m = re.match(r"^\s+(\S+)\s.*//\s*([0-9A-F]+):", line)
        if cur and m:
            body[cur].append((int(m.group(2), 16), m.group(1)))
    return {k: v for k, v in body.items() if any(f in k for f in FAMILIES)}


def check(name, ins, slot_bytes=256, align=0x2000, long_slot=4, NSLOTS=32):
    addr = {a: op for a, op in ins}
    order = [a for a, _ in ins]
    # table base: an aligned address whose predecessor is s_nop padding and which is followed by real code
    bases = [a for i, a in enumerate(order) if a % align == 0 and i > 0 and addr[order[i - 1]] == "s_nop" and addr[a] != "s_nop"]
    # (the code after the table -- the epilogue -- starts NSLOTS slots later and may match the same pattern)
    if not bases or len(bases) > 2 or (len(bases) == 2 and bases[1] != bases[0] + slot_bytes * NSLOTS):
        return ["%s: expected one handler table on a %d-byte boundary, found candidates at %s" % (name, align, [hex(b) for b in bases])]
    base, errs = bases[0], []
    prev = {order[i]: order[i - 1] for i in range(1, len(order))}
    for slot in range(1, NSLOTS + 1):
        b = base + slot_bytes * slot
        if long_slot is not None and slot == long_slot + 1:
            continue                                  # the pair handler of slot long_slot may continue into the next slot
        if b not in addr:
            if slot == NSLOTS and b > order[-1]:
                continue
            errs.append("%s: no instruction starts at slot boundary %d (+0x%x): an instruction straddles it" % (name, slot, slot_bytes * slot))
            continue
        if addr[prev[b]] not in ("s_nop", "s_setpc_b64", "s_branch"):
            errs.append("%s: handler of slot %d runs into slot %d (%s before +0x%x)" % (name, slot - 1, slot, addr[prev[b]], slot_bytes * slot))
    return errs


def main(obj):
    ks = kernels(obj)
    if not ks:
        print("asm_layout_check: no k_ll_fused4_asm kernel in", obj)
        return 1
    errs = []
    for name, ins in ks.items():
        fam = [f for f in FAMILIES if f in name][0]
        errs += check(name, ins, *FAMILIES[fam])
    for e in errs:
        print("asm_layout_check:", e)
    print("asm_layout_check: %d interpreter kernels, %d layout errors" % (len(ks), len(errs)))
    return 1 if errs else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
