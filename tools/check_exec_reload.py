#!/usr/bin/env python3
"""ISA audit (build container): a spill RELOAD executed while exec has been narrowed by a plain `s_and_b64 exec, exec, <mask>` (an inner
`if` that hipcc folded into the enclosing region without a save of its own) restores the register in the active lanes only; if all lanes
use it after the `s_or_b64 exec` that follows, the others continue with a stale value.  Found in round 4 (the fused row-dot epilogue of the
128-row GEMM instantiation: memory fault).  This scans every kernel of csrc/*.hip for a scratch_load between such an `s_and_b64 exec` and
the next write of exec, in the JOIN block (behind a label).  usage: python tools/check_exec_reload.py   (exit 1 if a candidate is found)"""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bad = 0
for src in sorted(glob.glob(os.path.join(ROOT, "audiossl_amd", "csrc", "*.hip"))):
    asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-S", "--cuda-device-only", "-o", "-", src],
                         capture_output=True, text=True).stdout.split("\n")
    kernel, narrowed, joined = None, None, False
    for i, l in enumerate(asm):
        t = l.strip()
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kernel, narrowed = m.group(1), None
        if re.match(r"s_and_b64 exec, exec, ", t) or re.match(r"s_andn2_b64 exec, exec, ", t):
            narrowed, joined = i, False
        elif narrowed is not None and re.match(r"^\.LBB\w+:", t):
            joined = True                                           # past the body of the inner `if`: the join block, still under the narrowed mask
        elif re.search(r"\bexec\b", t.split(",")[0]) and not t.startswith(";") and re.match(r"s_(or|mov|xor|and_saveexec|or_saveexec)\w*_b64 exec", t):
            narrowed = None
        elif narrowed is not None and joined and t.startswith("scratch_load"):   # (a reload INSIDE the inner body serves the active lanes only: fine)
            print(f"{os.path.basename(src)}: {kernel}: reload under narrowed exec (line {i}, exec narrowed at {narrowed}): {t}")
            bad += 1
            narrowed = None
print("candidates:", bad)
sys.exit(1 if bad else 0)
