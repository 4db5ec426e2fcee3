"""Build libatst_hip.so (gfx950) in-tree with hipcc.  No torch, no cmake: one shared object, C ABI (include/atst_hip.h)."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libatst_hip.so")
SOURCES = ["api.hip", "engine.hip", "gemm.hip", "layernorm.hip", "attention.hip", "tokens.hip", "head.hip", "optim.hip",
           "frontend.hip", "profile.hip", "augment.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result"]
if os.environ.get("ATST_EXTRA_FLAGS"):      # experiment builds: e.g. ATST_EXTRA_FLAGS="-DATST_ABLATE_ATTN_STORE"
    FLAGS += os.environ["ATST_EXTRA_FLAGS"].split()
for _k in ("ATST_TN_ILV", "ATST_TN_SPLIT", "ATST_TN_RM"):
    if os.environ.get(_k):
        FLAGS.append(f"-D{_k}=" + os.environ[_k])
if os.environ.get("ATST_INTERLEAVE"):
    FLAGS.append("-DATST_INTERLEAVE=" + os.environ["ATST_INTERLEAVE"])
if os.environ.get("ATST_TALL_STAGES"):
    FLAGS.append("-DATST_TALL_STAGES=" + os.environ["ATST_TALL_STAGES"])
if os.environ.get("ATST_ABLATE"):          # experiment builds only (tools/gemm_bench.py)
    FLAGS.append("-DATST_ABLATE=" + os.environ["ATST_ABLATE"])
if os.environ.get("ATST_EXPERIMENTS"):     # also compile the measured-and-rejected GEMM variants (tuning hooks 311 / 321 / 331 / 341)
    FLAGS.append("-DATST_EXPERIMENTS=" + os.environ["ATST_EXPERIMENTS"])
if os.environ.get("ATST_TRACE_FINE"):
    FLAGS.append("-DATST_TRACE_FINE=" + os.environ["ATST_TRACE_FINE"])
if os.environ.get("ATST_TRACE"):           # experiment builds only (tools/trace_gemm.py)
    FLAGS.append("-DATST_TRACE=" + os.environ["ATST_TRACE"])
if os.environ.get("ATST_NT_STORES"):
    FLAGS.append("-DATST_NT_STORES=" + os.environ["ATST_NT_STORES"])



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
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "build.sha256")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    hipcc = _hipcc()
    objs, procs = [], []
    for s in SOURCES:
        o = os.path.join(LIBDIR, s.replace(".hip", ".o"))
        objs.append(o)
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, s), "-o", o]
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
