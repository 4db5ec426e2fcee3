// In-library kernel timing with HIP events on the launch stream (bench.py's roofline figures).
#pragma once
#include <hip/hip_runtime.h>
enum ProfKind {
  PK_GEMM_NT0 = 0, PK_GEMM_NT1, PK_GEMM_NT2, PK_GEMM_NT3, PK_GEMM_NT4, PK_GEMM_NT5, PK_GEMM_TN,
  PK_ATTN_FWD, PK_ATTN_BWD_DKV, PK_ATTN_BWD_DQ, PK_LN_FWD, PK_LN_BWD, PK_MEL, PK_OPTIM, PK_GEMM_NT6, PK_COUNT
};
bool prof_on();
void prof_begin(int kind, double work, double bytes, hipStream_t st);   // work = algorithmic FLOPs (MFMA kinds) or bytes (HBM kinds); bytes = algorithmic HBM bytes
void prof_end(hipStream_t st);
struct ProfScope {
  hipStream_t st; bool on;
  ProfScope(int kind, double work, hipStream_t s, double bytes = -1.0) : st(s), on(prof_on()) { if (on) prof_begin(kind, work, bytes < 0 ? work : bytes, s); }
  ~ProfScope() { if (on) prof_end(st); }
};
