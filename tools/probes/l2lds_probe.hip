// Operand-delivery probe (run on the GPU box): how many bytes per clock per CU do the paths a GEMM main loop can use
// deliver, alone and mixed?  One 512-thread block per CU, no MFMA, no epilogue.
//   A = streamed-once operand (block-private slice of a 2 GB buffer -> HBM)
//   B = weight panel re-read by every block (1.18 MB -> L2 / MALL resident)
// modes: 0 DMA B | 1 DMA A | 2 DMA A+B (16 KB : 24 KB per k-tile, the 256x384 tile's mix) | 3 REG B | 4 REG B + ds_write
//        5 DMA A + REG B + ds_write | 6 REG A | 7 DMA A + REG B (no ds_write) | 8 DMA A+B at 32 KB : 24 KB (512-row tile)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/l2lds_probe tools/probes/l2lds_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const void __attribute__((address_space(1))) * gptr_t;
typedef void __attribute__((address_space(3))) * lptr_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr size_t B_BYTES = 384 * 1536 * 2;           // 1.18 MB panel
constexpr int KT_A = 16384, KT_B = 24576;            // bytes per k-tile (256 x 32 bf16, 384 x 32 bf16)

template <int MODE>
__global__ __launch_bounds__(512, 2) void probe(const char* A, const char* B, size_t a_per_block, int ktiles, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const char* a = A + (size_t)blockIdx.x * a_per_block;
  constexpr int A_KT = MODE == 8 ? 2 * KT_A : KT_A;
  constexpr int A_PW = A_KT / 1024 / 8, B_PW = KT_B / 1024 / 8;       // 1-KiB pieces per wave per k-tile: 2 (4) and 3
  constexpr bool dmaA = MODE == 1 || MODE == 2 || MODE == 5 || MODE == 7 || MODE == 8;
  constexpr bool dmaB = MODE == 0 || MODE == 2 || MODE == 8;
  constexpr bool regB = MODE == 3 || MODE == 4 || MODE == 5 || MODE == 7;
  constexpr bool regA = MODE == 6;
  constexpr bool dsw = MODE == 4 || MODE == 5;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  f32x4 r[2][3];
  size_t boff = 0;
  for (int kt = 0; kt < ktiles; ++kt) {
    char* stage = lds + (kt & 1) * (2 * KT_A + KT_B);
    if (dmaA) {
#pragma unroll
      for (int j = 0; j < A_PW; ++j)
        __builtin_amdgcn_global_load_lds((gptr_t)(a + (size_t)kt * A_KT + (wid * A_PW + j) * 1024 + lane * 16),
                                         (lptr_t)(stage + (wid * A_PW + j) * 1024), 16, 0, 0);
    }
    if (dmaB) {
#pragma unroll
      for (int j = 0; j < B_PW; ++j)
        __builtin_amdgcn_global_load_lds((gptr_t)(B + boff + (wid * B_PW + j) * 1024 + lane * 16),
                                         (lptr_t)(stage + 2 * KT_A + (wid * B_PW + j) * 1024), 16, 0, 0);
    }
    if (regB) {
#pragma unroll
      for (int j = 0; j < B_PW; ++j)
        r[kt & 1][j] = *reinterpret_cast<const f32x4*>(B + boff + (wid * B_PW + j) * 1024 + lane * 16);
    }
    if (regA) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        r[kt & 1][j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a + (size_t)kt * KT_A + (wid * 2 + j) * 1024 + lane * 16));
      r[kt & 1][2] = r[kt & 1][0];
    }
    // consume the PREVIOUS k-tile's registers (one tile of prefetch distance, like a software-pipelined main loop)
    if ((regB || regA) && kt > 0) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        if (dsw) *reinterpret_cast<f32x4*>(stage + 2 * KT_A + (wid * 3 + j) * 1024 + lane * 16) = r[(kt - 1) & 1][j];
        else acc += r[(kt - 1) & 1][j];
      }
    }
    if (dmaA || dmaB) {                      // keep two k-tiles of DMA in flight per wave
      constexpr int PER = (dmaA ? A_PW : 0) + (dmaB ? B_PW : 0);
      if (PER == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (PER == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (PER == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (PER == 5) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    }
    boff += KT_B; if (boff >= B_BYTES) boff = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f || lds[tid] == 77) sink[tid] = acc[0] + lds[tid * 4];
}

template <int MODE> void run(const char* name, const char* A, const char* B, float* sink, size_t a_total, double a_kt, double b_kt) {
  const int blocks = 256, ktiles = 2048;
  const size_t a_per_block = a_total / blocks;
  CK(hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (2 * KT_A + KT_B)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(512), 2 * (2 * KT_A + KT_B), 0, A, B, a_per_block, ktiles, sink);
  CK(hipEventRecord(e0));
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(512), 2 * (2 * KT_A + KT_B), 0, A, B, a_per_block, ktiles, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double bytes = (a_kt + b_kt) * ktiles * blocks;
  printf("%-44s %8.1f us  total %6.2f TB/s  (A/HBM %5.2f TB/s, B/L2 %5.2f TB/s)  %5.1f B/clk/CU @2.4GHz\n", name, ms * 1e3, bytes / ms / 1e9,
         a_kt * ktiles * blocks / ms / 1e9, b_kt * ktiles * blocks / ms / 1e9, bytes / 256 / (ms * 1e-3 * 2.4e9));
}

int main() {
  const size_t a_total = (size_t)256 * 2048 * 2 * KT_A;   // 16 GB would be too much: 256 blocks x 2048 ktiles x 32 KB = 17 GB -> wrap inside 4 GB
  char *A, *B; float* sink;
  const size_t a_alloc = (size_t)4 << 30;
  CK(hipMalloc(&A, a_alloc + (64 << 20))); CK(hipMalloc(&B, B_BYTES + (1 << 20))); CK(hipMalloc(&sink, 4096));
  CK(hipMemset(A, 1, a_alloc)); CK(hipMemset(B, 1, B_BYTES));
  (void)a_total;
  // each block streams 2048 x 16 KB = 32 MB (64 MB in mode 8): 256 blocks -> 8 (16) GB; wrap the per-block base into the 4 GB buffer
  const size_t per_block = a_alloc / 256;                 // 16 MB slices: a block wraps twice -> still far beyond L2/MALL (4 GB footprint)
  (void)per_block;
  run<0>("0 DMA B(L2) only", A, B, sink, a_alloc, 0, KT_B);
  run<1>("1 DMA A(HBM) only", A, B, sink, a_alloc, KT_A, 0);
  run<2>("2 DMA A+B  16:24", A, B, sink, a_alloc, KT_A, KT_B);
  run<8>("8 DMA A+B  32:24", A, B, sink, a_alloc, 2 * KT_A, KT_B);
  run<3>("3 REG B(L2) only", A, B, sink, a_alloc, 0, KT_B);
  run<4>("4 REG B + ds_write_b128", A, B, sink, a_alloc, 0, KT_B);
  run<6>("6 REG A(HBM) only (nt)", A, B, sink, a_alloc, KT_A, 0);
  run<7>("7 DMA A + REG B", A, B, sink, a_alloc, KT_A, KT_B);
  run<5>("5 DMA A + REG B + ds_write_b128", A, B, sink, a_alloc, KT_A, KT_B);
  return 0;
}
