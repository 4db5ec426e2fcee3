#!/usr/bin/env python3
"""Compact kernel resource table for one csrc file: python tools/kres.py gemm.hip [name-substring]"""
import re, subprocess, sys, os
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audiossl_amd", "csrc", sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "--cuda-device-only",
                      "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null", src], capture_output=True, text=True).stderr
cur = {}
for line in out.splitlines():
    m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (.*?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
    cur[k.split(" ")[0]] = v
    if k.startswith("LDS"):
        n = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
        n = n.replace("(anonymous namespace)::", "").replace("void ", "")
        n = re.sub(r"\(.*", "", n)
        if sub in n:
            print(f"{n:44s} vgpr {cur.get('VGPRs'):>4s} agpr {cur.get('AGPRs'):>4s} scratch {cur.get('ScratchSize'):>5s} occ {cur.get('Occupancy')}")
