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

#ifndef ATST_TN8_ABL              // experiment builds (tools/tn8_ablate.sh): 1 = no fragment reads, 2 = no LDS-DMA, 4 = no MFMAs, 8 = no atomics
#define ATST_TN8_ABL 0
#endif
int g_tn8_splits = 0;         // tuning hook 1400 + s: M-splits of the grouped e4m3 weight gradient (0 = automatic)
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

// one (tile, M-split) of one problem.  `p` may be an element of a kernel-argument ARRAY picked by a block-uniform index: every field is made visibly
// uniform first (read through a lane-typed select of the element address they come back as VECTOR loads next to each use: gemm.hip tn_tall_body)
DEVFN void tn8_body(const Wgrad8Args& pa, int tile, int split, char* smem_raw) {
  Wgrad8Args p;
  p.dY = reinterpret_cast<const uint8_t*>(sgpr_ptr8(pa.dY)); p.X = reinterpret_cast<const uint8_t*>(sgpr_ptr8(pa.X));
  p.dW = reinterpret_cast<float*>(const_cast<char*>(sgpr_ptr8(pa.dW)));
  p.sy = reinterpret_cast<const float*>(sgpr_ptr8(pa.sy)); p.sx = reinterpret_cast<const float*>(sgpr_ptr8(pa.sx));
  p.M = __builtin_amdgcn_readfirstlane(pa.M); p.N = __builtin_amdgcn_readfirstlane(pa.N); p.K = __builtin_amdgcn_readfirstlane(pa.K);
  p.ldy = __builtin_amdgcn_readfirstlane(pa.ldy); p.ldx = __builtin_amdgcn_readfirstlane(pa.ldx); p.ldw = __builtin_amdgcn_readfirstlane(pa.ldw);
  p.m_per_split = __builtin_amdgcn_readfirstlane(pa.m_per_split);
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wid >> 2, wk = wid & 3;
  const int ntk = (p.K + TK8 - 1) / TK8;
  const int n0 = (tile / ntk) * TN8, k0 = (tile % ntk) * TK8;
  // N, K are multiples of 128 (round 6: d = 384 -- 384, 1152, 1536): the last tile of a dimension may be HALF valid.  Its upper 128 columns are staged
  // from the lower 128 again (in-range addresses, products never stored) -- 384 = 1.5 tiles costs 2; the four problems of an ATST-small block are 38 tiles
  // for 27 tiles' worth of output, and still run ~3x the bf16 grouped kernel.
  const bool half_n = n0 + TN8 > p.N, half_k = k0 + TK8 > p.K;
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
    vo[0][j] = (unsigned)(row * p.ldy + ((half_n ? ch & 7 : ch) * 16));
    vo[1][j] = (unsigned)(row * p.ldx + ((half_k ? ch & 7 : ch) * 16));
  }
  const char* by = sgpr_ptr8(p.dY + (size_t)m_begin * p.ldy + n0);
  const char* bx = sgpr_ptr8(p.X + (size_t)m_begin * p.ldx + k0);
  const size_t sty = (size_t)RM8 * p.ldy, stx = (size_t)RM8 * p.ldx;
  const unsigned lds0 = lds_addr(smem_raw) + (2 * wid) * 1024;
  auto stage = [&](int s) {
    const unsigned dst = lds0 + (s & (NSTG8 - 1)) * STG;
#pragma unroll
    for (int j = 0; j < ((ATST_TN8_ABL & 2) ? 0 : 2); ++j) {
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
#if ATST_TN8_ABL & 1
#pragma unroll
    for (int j = 0; j < 2; ++j) b8[j] = v8i_{(int)so, 1, 2, 3, 4, 5, 6, 7};
#pragma unroll
    for (int i = 0; i < 4; ++i) a8[i] = v8i_{(int)so, 1, 2, 3, 4, 5, 6, 7};
#else
#pragma unroll
    for (int j = 0; j < 2; ++j) b8[j] = frag8(fb[j] + so);
#pragma unroll
    for (int i = 0; i < 4; ++i) a8[i] = frag8(fa[i] + so);
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a8[0]), "+v"(a8[1]), "+v"(a8[2]), "+v"(a8[3]), "+v"(b8[0]), "+v"(b8[1]));
    if (wn == 1 && s + 1 < nst) wait_stage(issued(s + NSTG8 - 1) - (s + 1));
    asm volatile("s_barrier" ::: "memory");                        // B2
    __builtin_amdgcn_sched_barrier(0);
#if ATST_TN8_ABL & 4
    asm volatile("" :: "v"(a8[0]), "v"(a8[1]), "v"(a8[2]), "v"(a8[3]), "v"(b8[0]), "v"(b8[1]));
#else
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i], b8[j], acc[i][j], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
#endif
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
        if (n >= p.N || k >= p.K) continue;                            // (uniform per wave: the invalid half of an edge tile is a whole wave row / two wave columns)
        if (!(ATST_TN8_ABL & 8)) atomicAdd(p.dW + (size_t)n * p.ldw + k, acc[i][j][r] * dq); else asm volatile("" :: "v"(acc[i][j][r] * dq));
      }
}

__global__ __launch_bounds__(512, 2) void gemm_tn8_kernel(Wgrad8Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tiles = ((p.N + TN8 - 1) / TN8) * ((p.K + TK8 - 1) / TK8);
  const int id = xcd_remap(blockIdx.x, gridDim.x);                // all tiles of one M-split run on one XCD (they stream the same dY8 / X8 rows)
  tn8_body(p, id % tiles, id / tiles, smem_raw);
}

// The e4m3 weight gradients of ONE transformer block in one launch (the four Linears share M and the M-split): what a launch pays besides its MFMAs is
// the fp32 atomics that combine the M-splits -- splits x 4 N K bytes per problem, and a problem launched alone needs ~512 / tiles splits to fill two rounds
// of the chip: 132 MB of atomics for EVERY problem, 59-88 us each (tools/tn8_ablate.sh: 21 % of fc1's launch, 59 % of proj's).  Together the four
// problems are 108 tiles: 2 splits = 216 blocks fill ONE round (84 % of the CUs) with 57 MB of atomics for all four -- measured best of 1 / 2 / 3 / 7 / 14
// splits (872 / 473-499 / 658 / 495-520 / 555-580 us, profiles/r05_tn8_group_splits.txt); the launcher below picks it.
struct Wgrad8Group { Wgrad8Args it[4]; int first_tile[5]; int n; };
__global__ __launch_bounds__(512, 2) void gemm_tn8_group_kernel(Wgrad8Group g) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int ntiles = g.first_tile[g.n];
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int t = id % ntiles, split = id / ntiles;                 // split-major: the tiles of one M-split (all problems) are neighbours on an XCD
  int pi = 0;
#pragma unroll
  for (int q = 1; q < 4; ++q) if (q < g.n && t >= g.first_tile[q]) pi = q;
  pi = __builtin_amdgcn_readfirstlane(pi);
  tn8_body(g.it[pi], t - g.first_tile[pi], split, smem_raw);
}
}  // namespace

static int tn8_splits(int tiles) { const int s = (192 + tiles - 1) / tiles; return s < 1 ? 1 : s; }   // one round, >= 3/4 of the 256 CUs

// dW[N, K] (fp32, ldw) += (1 / (*scale_y * *scale_x)) dY8[M, N]^T X8[M, K]; N, K multiples of 128 (256 x 256 tiles, the last one of a dimension half valid), M a multiple of 64, ldy / ldx multiples of 16.
int atst_gemm_tn8(const uint8_t* dY8, const uint8_t* X8, int M, int N, int K, int ldy, int ldx, float* dW, int ldw, const float* scale_y,
                  const float* scale_x, hipStream_t st) {
  if (!dY8 || !X8 || !dW || M <= 0 || M % RM8 || N % 128 || K % 128 || ldy % 16 || ldx % 16) return ATST_EINVAL;
  Wgrad8Args a{dY8, X8, M, N, K, ldy, ldx, dW, ldw, scale_y, scale_x, 0};
  const int tiles = ((N + TN8 - 1) / TN8) * ((K + TK8 - 1) / TK8);
  int splits = tn8_splits(tiles);                                   // (two rounds -- 512 / tiles -- paid 132 MB of atomics per launch: tools/tn8_ablate.sh)
  int mps = (M + splits - 1) / splits;
  mps = ((mps + RM8 - 1) / RM8) * RM8;
  if (mps < 4 * RM8) mps = 4 * RM8;
  a.m_per_split = mps;
  splits = (M + mps - 1) / mps;
  static OncePerDevice attr_done; int attr_done_dev;
  if (attr_done.need(attr_done_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tn8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NSTG8 * STG);
    if (e != hipSuccess) return (int)e;
    attr_done.done(attr_done_dev);
  }
  ProfScope ps(PK_GEMM_TN, 2.0 * M * N * K, st, (double)M * ((double)N + K) + 4.0 * N * K);
  hipLaunchKernelGGL(gemm_tn8_kernel, dim3(tiles * splits), dim3(512), NSTG8 * STG, st, a);
  return (int)hipGetLastError();
}

// dW_i += dY8_i^T X8_i / (scale_y_i scale_x_i) for up to four problems that share M (one transformer block); every N_i, K_i a multiple of 128.
int atst_gemm_tn8_group(const Wgrad8Item* items, int n, int M, hipStream_t st) {
  if (n < 1 || n > 4 || M <= 0 || M % RM8) return ATST_EINVAL;
  Wgrad8Group g{}; g.n = n;
  double flops = 0.0, bytes = 0.0;
  for (int i = 0; i < n; ++i) {
    const Wgrad8Item& it = items[i];
    if (!it.dY8 || !it.X8 || !it.dW || it.N % 128 || it.K % 128 || it.ldy % 16 || it.ldx % 16) return ATST_EINVAL;
    g.it[i] = Wgrad8Args{it.dY8, it.X8, M, it.N, it.K, it.ldy, it.ldx, it.dW, it.ldw, it.scale_y, it.scale_x, 0};
    g.first_tile[i + 1] = g.first_tile[i] + ((it.N + TN8 - 1) / TN8) * ((it.K + TK8 - 1) / TK8);
    flops += 2.0 * M * it.N * it.K; bytes += (double)M * ((double)it.N + it.K) + 4.0 * it.N * it.K;
  }
  // M-splits: ONE round of blocks, the smallest count that occupies >= 3/4 of the chip.  Measured on the four problems of an ATST-base block (108 tiles,
  // M = 131072; us per launch): 1 split 872, **2 splits 499**, 3: 658, 5: 621, 6: 547, 7 (three full rounds): 495-520, 14: 555-580 -- every split adds
  // 28 MB of fp32 atomics, and a second, partly filled round costs more than the idle sixth of the chip
  const int tiles = g.first_tile[n];
  int best = tn8_splits(tiles);
  int splits;
  splits = g_tn8_splits > 0 ? g_tn8_splits : best;
  int mps = (M + splits - 1) / splits;
  mps = ((mps + RM8 - 1) / RM8) * RM8;
  if (mps < 4 * RM8) mps = 4 * RM8;
  splits = (M + mps - 1) / mps;
  for (int i = 0; i < n; ++i) g.it[i].m_per_split = mps;
  static OncePerDevice attr_done; int attr_done_dev;
  if (attr_done.need(attr_done_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tn8_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NSTG8 * STG);
    if (e != hipSuccess) return (int)e;
    attr_done.done(attr_done_dev);
  }
  ProfScope ps(PK_GEMM_TN, flops, st, bytes);
  hipLaunchKernelGGL(gemm_tn8_group_kernel, dim3(tiles * splits), dim3(512), NSTG8 * STG, st, g);
  return (int)hipGetLastError();
}
