#!/usr/bin/env python3
"""ISA audit of the NP = 256 attention backward kernel (build container).  Its next-head loads are inline assembly, i.e. the compiler believes
a register destination is valid at the asm statement.  Between such a load and the ONE tied wait (`s_waitcnt vmcnt(N)` with the
destinations as "+v" operands, first wait behind the loop header) NO instruction may touch those registers -- a compiler copy or spill
there would move data that has not landed (nothing interlocks a VGPR read against an outstanding VMEM load).  Also: no scratch, no
`vmcnt(0)` inside the head loop other than the head-0 one, at least as many store instructions as the waits assume.
usage: python tools/check_attn_bwd_isa.py          (exit code 1 on a violation; run after every change to those kernels)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "audiossl_amd", "csrc", "attention.hip")


def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def successors(body, labels, i):
    l = body[i]
    m = re.match(r"s_c?branch\w*\s+(\.\w+)", l)
    if l.startswith("s_branch"):
        return [labels[m.group(1)]]
    if l.startswith("s_cbranch"):
        return [labels[m.group(1)], i + 1]
    return [i + 1]


def audit(asm, kernel, min_stores):
    start = next(i for i, l in enumerate(asm) if re.match(r"^" + kernel + r".*:", l))
    end = next(i for i in range(start, len(asm)) if asm[i].strip().startswith("s_endpgm"))
    body = [l.strip() for l in asm[start:end]]
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"(\.\w+):", l)] if m}
    bad, in_asm, loads = [], False, []
    for i, l in enumerate(body):
        if l.startswith(";;#ASMSTART"):
            in_asm = True
        elif l.startswith(";;#ASMEND"):
            in_asm = False
        elif in_asm and re.match(r"global_load_dword(x4)?\s", l):          # register-destination asm load (LDS-DMA has its own mnemonic)
            loads.append((i, regs(l.split(",")[0])))
    dest = set().union(*[r for _, r in loads])
    last = max(i for i, _ in loads)
    tied = next(i for i, l in enumerate(body) if re.match(r"s_waitcnt vmcnt\(\d+\)$", l) and l != "s_waitcnt vmcnt(0)")
    # every instruction that can execute between the last load and the tied wait (CFG walk, stopping at the wait; the statically possible but
    # never taken exit path behind the loads is walked too: conservative)
    seen, todo, load_at = set(), [last + 1], {i for i, _ in loads}
    while todo:
        i = todo.pop()
        if i in seen or i == tied or i >= len(body) or i in load_at:      # (the loads again: the rotated iteration's path around the wait, h = -1 only)
            continue
        seen.add(i)
        l = body[i]
        if l and l[0] not in ";." and not l.startswith("s_") and regs(l) & dest:
            bad.append((i, l))
        todo.extend(successors(body, labels, i))
    if any(l.startswith("scratch_") for l in body):
        bad.append((-1, "scratch access"))
    zero_waits = [i for i, l in enumerate(body) if l == "s_waitcnt vmcnt(0)"]
    if len(zero_waits) > 1:
        bad.append((zero_waits[1], "more than one bare vmcnt(0) in the kernel (the head-0 one is the only one written)"))
    stores = [i for i, l in enumerate(body) if l.startswith("global_store")]
    if len(stores) < min_stores:
        bad.append((-1, f"{len(stores)} store instructions, the waits assume >= {min_stores}"))
    print(f"{kernel}: {len(body)} lines, {len(loads)} asm register loads into v{min(dest)}..v{max(dest)}, tied wait '{body[tied]}' at line "
          f"{tied}, {len(seen)} instructions between the loads and the wait, {len(stores)} store instructions, {'OK' if not bad else 'VIOLATIONS'}")
    for i, l in bad:
        print("  VIOLATION", i, l)
    return not bad


def main():
    asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-S", "--cuda-device-only",
                          "-o", "-", SRC], capture_output=True, text=True).stdout.split("\n")
    # MODE 0: bf16 dqkv (8 dK / dV + 4 dQ store instructions per head) ; MODE 2: e4m3 only (4 + 2)
    ok = all([audit(asm, "_ZN12_GLOBAL__N_118attn_bwd256_kernelILi%dEEE" % mode, n) for mode, n in ((0, 12), (2, 6))])
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
