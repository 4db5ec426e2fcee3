// e4m3 weight gradient: dW[n, k] += dq * sum_m dY8[m, n] X8[m, k]   (dq = 1 / (scale_y * scale_x), fp32 atomics, M split over blocks).
// "fp8 weight gradients through ds_read_b64_tr_b8".  Both operands are m-major in HBM (the e4m3 copies the fp8 forward / dgrad
// already write), and v_mfma_scale_f32_32x32x64_f8f6f4 wants 32 contraction bytes per lane: the fragments are TRANSPOSED reads of a row-major LDS image.
//
// ds_read_b64_tr_b8 (probed on gfx950, tools/probes/probe_tr8.hip): inside every 16-lane group lane a supplies the address of 8 contiguous bytes,
// B[a][0..7]; lane i of the group receives byte j = B[2 j + (i >> 3)][i & 7], j = 0..7.  With the even lanes pointing at rows m0 .. m0 + 7 of
// columns c .. c + 7 and the odd lanes at the same rows of columns c + 8 .. c + 15, lane i receives column c + i, rows m0 .. m0 + 7: one MFMA
// operand (32 bytes per lane = 32 values of m for one n / k) is four such reads.  A and B fragments are built the same way, so the
// (byte position -> m) map is the same on both sides, which is all the contraction needs.
//
// Structure: 256 (n) x 256 (k) output tile, 8 waves 2 x 4, wave tile 128 x 64 (8 accumulators); a stage is 64 rows of m = ONE MFMA k-step: 16 KB of dY8 +
// 16 KB of X8, four stages in a 128-KB ring filled by LDS-DMA (inline asm, scalar base + lane offset) three stages ahead behind counted vmcnt waits;
// the two wave rows run half a stage apart (one reads fragments while the other is in its MFMAs: see the main loop).  The
// 16-B chunks of a 256-B image row are XOR-permuted by 2 (m & 7) (on the DMA's source side and in the read address): the 32 lanes a transposing
// read serves together then fall on 32 distinct 8-byte bank pairs.
// Reference math: the weight gradient of nn.Linear (audiossl/modules/transformer.py:87-90,109,119) on the e4m3 copies of its two operands.
#include "common.h"
#include "kernels.h"
#include "profile.h"

namespace {
typedef int v2i_ __attribute__((ext_vector_type(2)));
typedef int v8i_ __attribute__((ext_vector_type(8)));
constexpr int TN8 = 256, TK8 = 256, RM8 = 64, IMG = RM8 * 256, STG = 2 * IMG, NSTG8 = 4;     // 16 KB per operand image, 32 KB per stage, 128 KB ring

struct Wgrad8Args {
  const uint8_t* dY; const uint8_t* X; int M, N, K, ldy, ldx; float* dW; int ldw;
  const float* sy; const float* sx;     // device scalars: the scales the two copies were quantised with
  int m_per_split;
};

DEVFN const char* sgpr_ptr8(const void* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi_ = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const char*>(((unsigned long long)hi_ << 32) | lo);
}
DEVFN void glds16_s8(unsigned voff, const void* sbase, unsigned lds_dst) {
  unsigned keep;                                     // M0 is compiler-reserved and not preserved around a statement: saved, set, restored inside it
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
template <int OFF> DEVFN v2i_ tr8_at(unsigned addr) {
  v2i_ v;
  asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
// one operand fragment: rows m = 32 (g >> 1) + 8 r + (a >> 1), r = 0..3 (+2048 B each), of the 32 columns the address was built for
DEVFN v8i_ frag8(unsigned addr) {
  const v2i_ a = tr8_at<0>(addr), b = tr8_at<2048>(addr), c = tr8_at<4096>(addr), d = tr8_at<6144>(addr);
  return v8i_{a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1]};
}

__global__ __launch_bounds__(512, 2) void gemm_tn8_kernel(Wgrad8Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wid >> 2, wk = wid & 3;
  const int ntk = p.K / TK8, ntn = p.N / TN8, tiles = ntn * ntk;
  // all tiles of one M-split run on one XCD (they stream the same dY8 / X8 rows)
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int tile = id % tiles, split = id / tiles;
  const int n0 = (tile / ntk) * TN8, k0 = (tile % ntk) * TK8;
  const int m_begin = split * p.m_per_split;
  int m_end = m_begin + p.m_per_split; if (m_end > p.M) m_end = p.M;
  if (m_begin >= m_end) return;
  const int nst = (m_end - m_begin) / RM8;                          // the launcher cuts M at multiples of 64

  // LDS-DMA: a stage is 32 pieces of 1 KiB (4 image rows of 256 B); wave w stages pieces 2 w, 2 w + 1 of each image.  lane -> (row, position) of
  // the linear image; position c holds chunk c ^ 2 (row & 7) of the row.
  unsigned vo[2][2];                                                // [operand][piece]
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (2 * wid + j) * 4 + (lane >> 4), ch = (lane & 15) ^ ((row & 7) << 1);
    vo[0][j] = (unsigned)(row * p.ldy + ch * 16);
    vo[1][j] = (unsigned)(row * p.ldx + ch * 16);
  }
  const char* by = sgpr_ptr8(p.dY + (size_t)m_begin * p.ldy + n0);
  const char* bx = sgpr_ptr8(p.X + (size_t)m_begin * p.ldx + k0);
  const size_t sty = (size_t)RM8 * p.ldy, stx = (size_t)RM8 * p.ldx;
  const unsigned lds0 = lds_addr(smem_raw) + (2 * wid) * 1024;
  auto stage = [&](int s) {
    const unsigned dst = lds0 + (s & (NSTG8 - 1)) * STG;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      glds16_s8(vo[0][j], by + (size_t)s * sty, dst + j * 1024);
      glds16_s8(vo[1][j], bx + (size_t)s * stx, dst + IMG + j * 1024);
    }
  };

  // fragment addresses inside stage 0 (stage and operand image are added as wave-uniform terms)
  const int g = lane >> 4, a = lane & 15;
  auto faddr = [&](int c0) {                                        // c0: first of the fragment's 32 columns (a multiple of 32)
    const int row = 32 * (g >> 1) + (a >> 1), chunk = (c0 >> 4) + (g & 1);
    return (unsigned)(row * 256 + ((chunk ^ (a & 14)) << 4) + 8 * (a & 1));
  };
  unsigned fa[4], fb[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) fa[i] = lds_addr(smem_raw) + faddr(wn * 128 + i * 32);
#pragma unroll
  for (int j = 0; j < 2; ++j) fb[j] = lds_addr(smem_raw) + IMG + faddr(wk * 64 + j * 32);

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
  for (int s = 0; s < NSTG8 - 1; ++s)
    if (s < nst) stage(s);
  // my 4 LDS-DMA instructions of stage t have landed; the `y` younger stages issued so far stay in flight
  auto wait_stage = [&](int y) {
    if (y >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (y == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto issued = [&](int upto) { return upto < nst - 1 ? upto : nst - 1; };   // the last stage requested once stage `upto` has had its turn
  // The two wave rows (waves 0-3 / 4-7: one wave of each on every SIMD) run HALF A STAGE apart -- row 1 passes one extra barrier first -- so that a
  // SIMD's matrix pipe works on one row's 8 MFMAs (512 cycles) while the other row reads its 24 transposed fragments (the eight waves' reads of a
  // stage are 96 KB = 768 LDS cycles: in step with the MFMAs they were a serial 40 % of the stage).  Two barriers per stage: B1 in front of the
  // reads, B2 in front of the MFMAs; row 0's B1(s) is row 1's B2(s - 1).  A row has waited for ITS pieces of the stage the other row reads next
  // before the barrier that releases those reads: row 0 for stage s before B1(s), row 1 for stage s + 1 before B2(s) (and for stage 0 before the
  // extra barrier).  A buffer is refilled (stage s + 3 into the buffer of s - 1) behind B1(s), when both rows' reads of s - 1 are complete.
  if (wn == 1) { wait_stage(issued(NSTG8 - 2)); asm volatile("s_barrier" ::: "memory"); }
  for (int s = 0; s < nst; ++s) {
    if (wn == 0) wait_stage(issued(s + NSTG8 - 2) - s);
    asm volatile("s_barrier" ::: "memory");                        // B1
    if (s + NSTG8 - 1 < nst) stage(s + NSTG8 - 1);
    const unsigned so = (s & (NSTG8 - 1)) * STG;
    v8i_ b8[2], a8[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) b8[j] = frag8(fb[j] + so);
#pragma unroll
    for (int i = 0; i < 4; ++i) a8[i] = frag8(fa[i] + so);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a8[0]), "+v"(a8[1]), "+v"(a8[2]), "+v"(a8[3]), "+v"(b8[0]), "+v"(b8[1]));
    if (wn == 1 && s + 1 < nst) wait_stage(issued(s + NSTG8 - 1) - (s + 1));
    asm volatile("s_barrier" ::: "memory");                        // B2
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i], b8[j], acc[i][j], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
  }
  if (wn == 0) asm volatile("s_barrier" ::: "memory");            // row 1's last B2
  // epilogue: fp32 atomics, consecutive lanes on consecutive k (C layout: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
  const float dq = 1.0f / ((p.sy ? *p.sy : 1.0f) * (p.sx ? *p.sx : 1.0f));
  const int hi = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 128 + i * 32 + crow32(r, hi), k = k0 + wk * 64 + j * 32 + l31;
        atomicAdd(p.dW + (size_t)n * p.ldw + k, acc[i][j][r] * dq);
      }
}
}  // namespace

// dW[N, K] (fp32, ldw) += (1 / (*scale_y * *scale_x)) dY8[M, N]^T X8[M, K]; N, K multiples of 256, M a multiple of 64, ldy / ldx multiples of 16.
int atst_gemm_tn8(const uint8_t* dY8, const uint8_t* X8, int M, int N, int K, int ldy, int ldx, float* dW, int ldw, const float* scale_y,
                  const float* scale_x, hipStream_t st) {
  if (!dY8 || !X8 || !dW || M <= 0 || M % RM8 || N % TN8 || K % TK8 || ldy % 16 || ldx % 16) return ATST_EINVAL;
  Wgrad8Args a{dY8, X8, M, N, K, ldy, ldx, dW, ldw, scale_y, scale_x, 0};
  const int tiles = (N / TN8) * (K / TK8);
  int splits = 512 / tiles; if (splits < 1) splits = 1;             // two rounds of one block per CU
  int mps = (M + splits - 1) / splits;
  mps = ((mps + RM8 - 1) / RM8) * RM8;
  if (mps < 4 * RM8) mps = 4 * RM8;
  a.m_per_split = mps;
  splits = (M + mps - 1) / mps;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tn8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NSTG8 * STG);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  ProfScope ps(PK_GEMM_TN, 2.0 * M * N * K, st, (double)M * ((double)N + K) + 4.0 * N * K);
  hipLaunchKernelGGL(gemm_tn8_kernel, dim3(tiles * splits), dim3(512), NSTG8 * STG, st, a);
  return (int)hipGetLastError();
}
