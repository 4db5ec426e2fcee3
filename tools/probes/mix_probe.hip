// DMA mix probe: A (HBM stream) : B (L2-resident panel) per k-tile for candidate GEMM block shapes, 1 or 2 blocks per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/mix_probe tools/probes/mix_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef const void __attribute__((address_space(1))) * gptr_t;
typedef void __attribute__((address_space(3))) * lptr_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr size_t B_BYTES = 384 * 1536 * 2;

template <int THREADS, int A_KT, int B_KT, int DEPTH>
__global__ __launch_bounds__(THREADS) void probe(const char* A, const char* B, size_t a_per_block, int ktiles, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int WAVES = THREADS / 64, A_PW = A_KT / 1024 / WAVES, B_PW = B_KT / 1024 / WAVES, PER = A_PW + B_PW;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const char* a = A + (size_t)blockIdx.x * a_per_block;
  size_t boff = 0;
  for (int kt = 0; kt < ktiles; ++kt) {
    char* stage = lds + (kt & 1) * (A_KT + B_KT);
#pragma unroll
    for (int j = 0; j < A_PW; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(a + (size_t)kt * A_KT + (wid * A_PW + j) * 1024 + lane * 16), (lptr_t)(stage + (wid * A_PW + j) * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < B_PW; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(B + boff + (wid * B_PW + j) * 1024 + lane * 16), (lptr_t)(stage + A_KT + (wid * B_PW + j) * 1024), 16, 0, 0);
    // keep DEPTH k-tiles in flight per wave
    constexpr int N = PER * DEPTH;
    if (N <= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (N <= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (N <= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (N <= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (N <= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    boff += B_KT; if (boff >= B_BYTES) boff = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (lds[tid] == 77) sink[tid] = lds[tid * 4];
}

template <int THREADS, int A_KT, int B_KT, int DEPTH>
void run(const char* name, int blocks_per_cu, const char* A, const char* B, float* sink, size_t a_alloc) {
  const int blocks = 256 * blocks_per_cu;
  const int ktiles = (int)((size_t)(32 << 20) / A_KT / blocks_per_cu);    // every CU streams 32 MB of A
  const size_t a_per_block = a_alloc / blocks;
  const int lds = 2 * (A_KT + B_KT);
  CK(hipFuncSetAttribute((const void*)probe<THREADS, A_KT, B_KT, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe<THREADS, A_KT, B_KT, DEPTH>), dim3(blocks), dim3(THREADS), lds, 0, A, B, a_per_block, ktiles, sink);
  CK(hipEventRecord(e0));
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<THREADS, A_KT, B_KT, DEPTH>), dim3(blocks), dim3(THREADS), lds, 0, A, B, a_per_block, ktiles, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double ab = (double)A_KT * ktiles * blocks, bb = (double)B_KT * ktiles * blocks;
  printf("%-58s %8.1f us  A/HBM %5.2f TB/s  B/L2 %5.2f TB/s  total %5.1f B/clk/CU @2.4GHz\n", name, ms * 1e3, ab / ms / 1e9, bb / ms / 1e9, (ab + bb) / 256 / (ms * 1e-3 * 2.4e9));
}

int main() {
  char *A, *B; float* sink;
  const size_t a_alloc = (size_t)8 << 30;
  CK(hipMalloc(&A, a_alloc + (64 << 20))); CK(hipMalloc(&B, B_BYTES + (1 << 20))); CK(hipMalloc(&sink, 4096));
  CK(hipMemset(A, 1, a_alloc)); CK(hipMemset(B, 1, B_BYTES));
  run<512, 16384, 24576, 2>("256x384 tile, 8 waves, 1 blk/CU, 2 tiles in flight", 1, A, B, sink, a_alloc);
  run<512, 16384, 24576, 3>("256x384 tile, 8 waves, 1 blk/CU, 3 tiles in flight", 1, A, B, sink, a_alloc);
  run<256, 8192, 24576, 2>("128x384 tile, 4 waves, 2 blk/CU, 2 tiles in flight", 2, A, B, sink, a_alloc);
  run<256, 8192, 24576, 3>("128x384 tile, 4 waves, 2 blk/CU, 3 tiles in flight", 2, A, B, sink, a_alloc);
  run<256, 16384, 12288, 2>("256x192 tile, 4 waves, 2 blk/CU, 2 tiles in flight", 2, A, B, sink, a_alloc);
  run<256, 16384, 24576, 2>("256x384 tile, 4 waves, 2 blk/CU, 2 tiles in flight", 2, A, B, sink, a_alloc);
  run<512, 32768, 24576, 2>("512x384 tile, 8 waves, 1 blk/CU, 2 tiles in flight", 1, A, B, sink, a_alloc);
  run<256, 8192, 12288, 3>("128x192 tile, 4 waves, 2 blk/CU, 3 tiles in flight", 2, A, B, sink, a_alloc);
  return 0;
}
