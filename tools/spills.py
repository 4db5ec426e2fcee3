#!/usr/bin/env python3
"""Where does a kernel spill?  python tools/spills.py gemm.hip <mangled-name-substring>
Prints, for every matching kernel, the number of scratch instructions before / inside / after its MFMA range (main loop)."""
import os, re, subprocess, sys
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audiossl_amd", "csrc", sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
asm = "/tmp/_spills.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "--cuda-device-only",
                "-S", "-o", asm, src], check=True, capture_output=True)
lines = open(asm).read().split("\n")
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\S+):\s", l)
    if not m or sub not in m.group(1):
        continue
    end = next(j for j in range(i, len(lines)) if "s_endpgm" in lines[j])
    body = lines[i:end]
    mf = [k for k, x in enumerate(body) if "v_mfma" in x]
    sc = [k for k, x in enumerate(body) if "scratch_" in x]
    if not mf:
        continue
    pre = sum(1 for k in sc if k < mf[0]); mid = sum(1 for k in sc if mf[0] <= k <= mf[-1]); post = sum(1 for k in sc if k > mf[-1])
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
    print(f"{name[:70]:70s} scratch ops: before {pre:3d}  inside MFMA range {mid:3d}  after {post:3d}   ({len(body)} lines)")
