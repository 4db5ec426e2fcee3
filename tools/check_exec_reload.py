#!/usr/bin/env python3
"""ISA audit (build container): a spill RELOAD executed while exec has been narrowed by a plain `s_and_b64 exec, exec, <mask>` (an inner
`if` that hipcc folded into the enclosing region without a save of its own) restores the register in the active lanes only; if all lanes
use it after the `s_or_b64 exec` that follows, the others continue with a stale value.  Found in round 4 (the fused row-dot epilogue of the
128-row GEMM instantiation: memory fault).  This scans every kernel of csrc/*.hip for a scratch_load between such an `s_and_b64 exec` and
the next write of exec, in the JOIN block (behind a label).  usage: python tools/check_exec_reload.py   (exit 1 if a candidate is found)

What counts as a write of exec (ends the narrowed region): any instruction whose destination is exec (`s_or_b64 exec, ...`, `s_mov_b64 exec, ...`,
`s_xor_b64 exec, ...`), and every `s_*_saveexec_b64` -- those write exec IMPLICITLY (their destination operand is the SGPR pair that receives the
old mask; round-4 ADVICE: the first version only looked at the first operand and never recognised them).  The state is also reset at `s_endpgm`
and at every kernel symbol, so a narrowing cannot leak from one kernel (or one exit path) into the text that follows it."""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

NARROW = re.compile(r"s_(and|andn2)_b64 exec, exec, ")
SAVEEXEC = re.compile(r"s_\w+_saveexec_b64\b")
EXEC_DST = re.compile(r"s_\w+_b64 exec\b")


def scan(asm_lines):
    """-> list of (kernel, line index of the reload, line index of the narrowing, text)"""
    out = []
    kernel, narrowed, joined = None, None, False
    for i, l in enumerate(asm_lines):
        t = l.strip()
        if not t or t.startswith(";"):
            continue
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kernel, narrowed, joined = m.group(1), None, False
            continue
        if t.startswith("s_endpgm"):
            narrowed, joined = None, False
        elif NARROW.match(t):
            narrowed, joined = i, False
        elif SAVEEXEC.match(t) or EXEC_DST.match(t):                # exec rewritten (explicitly, or implicitly by a saveexec form)
            narrowed, joined = None, False
        elif narrowed is not None and re.match(r"^\.LBB\w+:", t):
            joined = True                                           # past the body of the inner `if`: the join block, still under the narrowed mask
        elif narrowed is not None and joined and t.startswith("scratch_load"):   # (a reload INSIDE the inner body serves the active lanes only: fine)
            out.append((kernel, i, narrowed, t))
            narrowed, joined = None, False
    return out


def main():
    bad = 0
    for src in sorted(glob.glob(os.path.join(ROOT, "audiossl_amd", "csrc", "*.hip"))):
        asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-S", "--cuda-device-only", "-o", "-", src],
                             capture_output=True, text=True).stdout.split("\n")
        for kernel, i, n, t in scan(asm):
            print(f"{os.path.basename(src)}: {kernel}: reload under narrowed exec (line {i}, exec narrowed at {n}): {t}")
            bad += 1
    print("candidates:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
