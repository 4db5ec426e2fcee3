"""Build libatst_hip.so (gfx950) in-tree with hipcc.  No torch, no cmake: one shared object, C ABI (include/atst_hip.h)."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
# ATST_LIB_TAG=x: an experiment build next to the product library (lib/libatst_hip_x.so, objects in lib/obj_x/), picked up by
# audiossl_amd.hip under the same variable -- several builds (ATST_EXTRA_FLAGS=...) can then be A/B-timed inside ONE gpurun call.
TAG = os.environ.get("ATST_LIB_TAG", "")
LIB = os.path.join(LIBDIR, f"libatst_hip_{TAG}.so" if TAG else "libatst_hip.so")
OBJDIR = os.path.join(LIBDIR, f"obj_{TAG}") if TAG else LIBDIR
SOURCES = ["api.hip", "engine.hip", "engine_hp.hip", "gemm.hip", "layernorm.hip", "attention.hip", "tokens.hip", "head.hip", "optim.hip",
           "frontend.hip", "profile.hip", "augment.hip", "gemm_tn8.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result"]
if os.environ.get("ATST_EXTRA_FLAGS"):      # experiment builds: e.g. ATST_EXTRA_FLAGS="-DATST_ABLATE_ATTN_STORE"
    FLAGS += os.environ["ATST_EXTRA_FLAGS"].split()
# Experiment builds for the stand-alone GEMM tools (tools/gemm_bench.py, trace_*.py, ablate*.sh): the measured-and-rejected
# round-2 GEMM variants live in tools/experiments/gemm_r02_variants.hip, which replaces csrc/gemm.hip when ATST_GEMM_VARIANTS=1
# or any of its switches is set.  That translation unit predates EPI_LNBWD: the encoder backward does not run on such a build.
_VARIANT_SWITCHES = ("ATST_TN_ILV", "ATST_TN_SPLIT", "ATST_TN_RM", "ATST_INTERLEAVE", "ATST_TALL_STAGES", "ATST_ABLATE", "ATST_EXPERIMENTS",
                     "ATST_TRACE_FINE", "ATST_TRACE", "ATST_NT_STORES", "ATST_TN_ISSUE")
GEMM_VARIANTS = os.environ.get("ATST_GEMM_VARIANTS") == "1" or any(os.environ.get(k) for k in _VARIANT_SWITCHES)
VARIANT_SRC = os.path.join(os.path.dirname(HERE), "tools", "experiments", "gemm_r02_variants.hip")
if GEMM_VARIANTS:
    FLAGS += ["-I", CSRC] + [f"-D{k}=" + os.environ[k] for k in _VARIANT_SWITCHES if os.environ.get(k)]



def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest():
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)) + ["../../include/atst_hip.h"]:
        p = os.path.join(CSRC, f)
        if os.path.isfile(p):
            h.update(open(p, "rb").read())
    if GEMM_VARIANTS:
        h.update(open(VARIANT_SRC, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    stamp = os.path.join(OBJDIR, "build.sha256")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    hipcc = _hipcc()
    objs, procs = [], []
    for s in SOURCES:
        o = os.path.join(OBJDIR, s.replace(".hip", ".o"))
        objs.append(o)
        src = VARIANT_SRC if (GEMM_VARIANTS and s == "gemm.hip") else os.path.join(CSRC, s)
        cmd = [hipcc, *FLAGS, "-c", src, "-o", o]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
        if verbose and out.strip():
            print(out)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB])
    open(stamp, "w").write(dig)
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
