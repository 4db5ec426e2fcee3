// High-precision twin of the encoder driver (engine.hip): the SAME sequence of operations -- patchify, token table, 12 x (LN,
// QKV, attention, proj + residual, LN, fc1 + GELU, fc2 + residual), final LN, and its backward -- with fp32 activations and
// gradients, every Linear evaluated by the production MFMA GEMMs on split-bf16 operands ([hi|lo|hi] x [hi|hi|lo] along the
// contraction axis: products exact, ~2^-17 relative), and attention / LayerNorm / GELU in plain fp32 kernels.
//
// Purpose: parity.  A bf16 encoder cannot be compared with the fp32 reference below ~1 % (bf16 rounding of every operand, then
// ReLU-gate flips behind the BatchNorm heads, DESIGN.md "Precision"), so the bf16 tests cannot tell a rounding difference from a
// wiring mistake.  This mode removes the rounding and keeps everything else -- the Python engine (view grouping, heads, loss,
// optimizer, EMA, ATST-Frame row gather), the token stage, DropPath / key-padding semantics, the parameter and gradient
// layout, and the production GEMM / wgrad kernels and their fp32 epilogues -- so that gradients can be pinned to the reference
// goldens at <= 2e-3 per tensor (tests/test_precise_gpu.py).  Measured 6.9x slower than the bf16 path at the same batch (662 vs 4582 clips/s, 2 views x 64 clips: bench.py --precise, profiles/r05_k_bench_precise_clip2_b64.json; round 4: 232 clips/s = 18-20x, when its attention / LayerNorm-backward / GELU' twins re-read every row from global memory).
// Reference math: audiossl/modules/transformer.py:95-159, audiossl/models/atst/audio_transformer.py:153-221.
#include "common.h"
#include "kernels.h"
#include "../../include/atst_hip.h"

namespace {
constexpr float LN_EPS = 1e-6f;

// ---- split-bf16 operand preparation ------------------------------------------------------------------------------------
// contraction along rows (weight gradients): y[3][n] = {hi, lo, hi} (which = 0, dY) or {hi, hi, lo} (which = 1, X)
__global__ void split3_rows_kernel(const float* __restrict__ x, size_t n, int which, bf16* __restrict__ y) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    const bf16 hi = f2bf(v), lo = f2bf(v - bf2f(hi));
    y[i] = hi; y[n + i] = which ? hi : lo; y[2 * n + i] = which ? lo : hi;
  }
}
// B operand of a dgrad GEMM: W [R, K] fp32 -> y [K, 3 R] = {hi | hi | lo} of W^T
__global__ void split3_t_kernel(const float* __restrict__ W, int R, int K, bf16* __restrict__ y) {
  const size_t total = (size_t)R * K;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / K; const int k = (int)(i % K);
    const float v = W[i];
    const bf16 hi = f2bf(v), lo = f2bf(v - bf2f(hi));
    bf16* o = y + (size_t)k * 3 * R + r;
    o[0] = hi; o[R] = hi; o[2 * (size_t)R] = lo;
  }
}

// ---- LayerNorm, one wave per row, any C that is a multiple of 64 ---------------------------------------------------------
__global__ __launch_bounds__(256) void ln_fwd_hp_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int M, int C) {
  const int lane = threadIdx.x & 63, row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= M) return;
  const float* xr = x + (size_t)row * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += xr[c];
  const float mu = wave_sum(s) / C;
  float q = 0.f;
  for (int c = lane; c < C; c += 64) { const float d = xr[c] - mu; q += d * d; }
  const float rs = rsqrtf(wave_sum(q) / C + LN_EPS);
  for (int c = lane; c < C; c += 64) y[(size_t)row * C + c] = (xr[c] - mu) * rs * gamma[c] + beta[c];
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}
// dx = dres + LN'(dy) ; g = row_scale dx (fp32, the upstream Linear's gradient operand) ; dgamma, dbeta, dbias_up += column sums.
// A wave walks LNB_ROWS consecutive rows and keeps the column sums of its lanes' columns in registers: one atomic per column per wave instead of three
// per ELEMENT (round 5: this kernel was 10 % of the parity-mode step).
constexpr int LNB_ROWS = 16, LNB_MAXC = 768 / 64;
__global__ __launch_bounds__(256) void ln_bwd_hp_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ dres,
                                                        float* __restrict__ dx, float* __restrict__ g, const float* __restrict__ row_scale, int rps,
                                                        float* dgamma, float* dbeta, float* dbias_up, int M, int C) {
  const int lane = threadIdx.x & 63, w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nc = C / 64;
  float sg[LNB_MAXC], sb[LNB_MAXC], su[LNB_MAXC];
#pragma unroll
  for (int k = 0; k < LNB_MAXC; ++k) sg[k] = sb[k] = su[k] = 0.f;
  for (int r = 0; r < LNB_ROWS; ++r) {
    const int row = w * LNB_ROWS + r;
    if (row >= M) break;
    const size_t base = (size_t)row * C;
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < LNB_MAXC; ++k) {
      if (k >= nc) break;
      const int c = lane + 64 * k;
      const float xh = (x[base + c] - mu) * rs, d = dy[base + c], dg = d * gamma[c];
      s1 += dg; s2 += dg * xh;
      sg[k] += d * xh; sb[k] += d;
    }
    const float c1 = wave_sum(s1) / C, c2 = wave_sum(s2) / C;
    const float sc = row_scale ? row_scale[row / rps] : 1.0f;
#pragma unroll
    for (int k = 0; k < LNB_MAXC; ++k) {
      if (k >= nc) break;
      const int c = lane + 64 * k;
      const float xh = (x[base + c] - mu) * rs;
      const float o = (dres ? dres[base + c] : 0.f) + rs * (dy[base + c] * gamma[c] - c1 - xh * c2);
      dx[base + c] = o;
      if (g) { const float gs = o * sc; g[base + c] = gs; su[k] += gs; }
    }
  }
#pragma unroll
  for (int k = 0; k < LNB_MAXC; ++k) {
    if (k >= nc) break;
    const int c = lane + 64 * k;
    atomicAdd(dgamma + c, sg[k]); atomicAdd(dbeta + c, sb[k]);
    if (g && dbias_up) atomicAdd(dbias_up + c, su[k]);
  }
}

// ---- exact erf-GELU (nn.GELU default, audiossl/modules/transformer.py:70-92) ----------------------------------------------
__global__ void gelu_hp_kernel(const float* __restrict__ u, size_t n, float* __restrict__ a) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = u[i];
    a[i] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
  }
}
// du = dA * gelu'(u) ; colsum[n] += sum_rows du  (fc1 bias gradient)
__global__ void dgelu_hp_kernel(const float* __restrict__ dA, const float* __restrict__ u, int M, int N, float* __restrict__ du, float* colsum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  float acc = 0.f;
  for (int m = blockIdx.y; m < M; m += gridDim.y) {
    const size_t i = (size_t)m * N + c;
    const float v = u[i];
    const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752f)), pdf = 0.3989422804014327f * expf(-0.5f * v * v);
    const float d = dA[i] * (cdf + v * pdf);
    du[i] = d; acc += d;
  }
  atomicAdd(colsum + c, acc);
}

// ---- attention in plain fp32: one block per (sequence, head), one thread per query / per key ------------------------------
// scores = q k^T / 8 over the `valid` keys (the reference adds -10000 to the others: their weight is exp(-10000 + ...) == 0 in
// fp32), softmax, o = p v.  lse saved for the backward.  ref: audiossl/modules/transformer.py:95-126.
// Round 5: the rows every thread walks (K, V ; in the backward also Q, dO) are staged in LDS once per block (2 x NP x 64 fp32 = 128 KB at NP = 256) and
// read back as broadcast 16-B reads -- before, all 256 threads re-read every row from global memory and the two kernels were 66 % of the
// parity-mode step (profiles/r05_precise_kernel_stats.csv).  Same arithmetic in the same order; the backward is one launch (dq, then dk / dv).
DEVFN void hp_stage_rows(const float* __restrict__ src /* row 0 of the head's 64 columns */, size_t ld, int NP, float* __restrict__ dst) {
  for (int idx = threadIdx.x; idx < NP * 16; idx += blockDim.x) {
    const int row = idx >> 4, c4 = (idx & 15) * 4;
    *reinterpret_cast<f32x4*>(dst + row * 64 + c4) = *reinterpret_cast<const f32x4*>(src + (size_t)row * ld + c4);
  }
}
__global__ __launch_bounds__(256) void attn_hp_fwd_kernel(const float* __restrict__ qkv, const int* __restrict__ valid, float* __restrict__ o,
                                                          float* __restrict__ lse, int H, int NP) {
  extern __shared__ __attribute__((aligned(16))) float hp_sm[];
  float* sK = hp_sm; float* sV = hp_sm + NP * 64;
  const int s = blockIdx.x / H, h = blockIdx.x % H, C = H * 64, i = threadIdx.x;
  const int nv = valid[s];
  const float* base = qkv + (size_t)s * NP * 3 * C;
  hp_stage_rows(base + C + h * 64, 3 * (size_t)C, NP, sK);
  hp_stage_rows(base + 2 * C + h * 64, 3 * (size_t)C, NP, sV);
  __syncthreads();
  if (i >= NP) return;
  float q[64];
  for (int d = 0; d < 64; ++d) q[d] = base[(size_t)i * 3 * C + h * 64 + d];
  float m = -INFINITY;
  for (int j = 0; j < nv; ++j) {
    const float* k = sK + j * 64;
    float sc = 0.f;
#pragma unroll
    for (int d = 0; d < 64; ++d) sc += q[d] * k[d];
    m = fmaxf(m, sc * 0.125f);
  }
  float l = 0.f, acc[64];
  for (int d = 0; d < 64; ++d) acc[d] = 0.f;
  for (int j = 0; j < nv; ++j) {
    const float* k = sK + j * 64;
    const float* v = sV + j * 64;
    float sc = 0.f;
#pragma unroll
    for (int d = 0; d < 64; ++d) sc += q[d] * k[d];
    const float p = expf(sc * 0.125f - m);
    l += p;
#pragma unroll
    for (int d = 0; d < 64; ++d) acc[d] += p * v[d];
  }
  const float inv = 1.0f / l;
  for (int d = 0; d < 64; ++d) o[((size_t)s * NP + i) * C + h * 64 + d] = acc[d] * inv;
  lse[((size_t)s * H + h) * NP + i] = m + logf(l);
}
// thread = query i: dq_i (K, V in LDS) ; then thread = key j: dk_j, dv_j (Q, dO in LDS ; D_i = rowsum(dO o) and lse_i from the first half).
// p_ij recomputed from q, k and the saved lse.
__global__ __launch_bounds__(256) void attn_hp_bwd_kernel(const float* __restrict__ qkv, const int* __restrict__ valid, const float* __restrict__ o,
                                                          const float* __restrict__ lse, const float* __restrict__ d_o, float* __restrict__ dqkv, int H, int NP) {
  extern __shared__ __attribute__((aligned(16))) float hp_sm[];
  float* sA = hp_sm; float* sB = hp_sm + NP * 64; float* sD = sB + NP * 64; float* sL = sD + NP;
  const int s = blockIdx.x / H, h = blockIdx.x % H, C = H * 64, t = threadIdx.x;
  const int nv = valid[s];
  const float* base = qkv + (size_t)s * NP * 3 * C;
  const float* dob = d_o + (size_t)s * NP * C + h * 64;
  hp_stage_rows(base + C + h * 64, 3 * (size_t)C, NP, sA);          // K
  hp_stage_rows(base + 2 * C + h * 64, 3 * (size_t)C, NP, sB);      // V
  __syncthreads();
  float* out = dqkv + ((size_t)s * NP + (t < NP ? t : 0)) * 3 * C + h * 64;
  if (t < NP) {
    const int i = t;
    float q[64], g[64], acc[64];
    float D = 0.f;
    const float* qi = base + (size_t)i * 3 * C + h * 64;
    const float* oi = o + ((size_t)s * NP + i) * C + h * 64;
    for (int d = 0; d < 64; ++d) { q[d] = qi[d]; g[d] = dob[(size_t)i * C + d]; D += g[d] * oi[d]; acc[d] = 0.f; }
    const float L = lse[((size_t)s * H + h) * NP + i];
    for (int j = 0; j < nv; ++j) {
      const float* kj = sA + j * 64; const float* vj = sB + j * 64;
      float sc = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < 64; ++d) { sc += q[d] * kj[d]; dp += g[d] * vj[d]; }
      const float ds = expf(sc * 0.125f - L) * (dp - D) * 0.125f;
#pragma unroll
      for (int d = 0; d < 64; ++d) acc[d] += ds * kj[d];
    }
    for (int d = 0; d < 64; ++d) out[d] = acc[d];
    sD[i] = D; sL[i] = L;
  }
  float k[64], v[64];
  const bool live = t < NP && t < nv;
  if (live) { for (int d = 0; d < 64; ++d) { k[d] = sA[t * 64 + d]; v[d] = sB[t * 64 + d]; } }
  __syncthreads();                                                  // every thread is done with K / V: the buffers take Q and dO
  hp_stage_rows(base + h * 64, 3 * (size_t)C, NP, sA);              // Q
  hp_stage_rows(dob, (size_t)C, NP, sB);                            // dO
  __syncthreads();
  if (t >= NP) return;
  float dk[64], dv[64];
  for (int d = 0; d < 64; ++d) dk[d] = dv[d] = 0.f;
  if (live) {
    for (int i = 0; i < NP; ++i) {                                  // every query row attends (pad queries carry zero upstream gradient)
      const float* qi = sA + i * 64; const float* gi = sB + i * 64;
      float sc = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < 64; ++d) { sc += qi[d] * k[d]; dp += gi[d] * v[d]; }
      const float p = expf(sc * 0.125f - sL[i]);
      const float ds = p * (dp - sD[i]) * 0.125f;
#pragma unroll
      for (int d = 0; d < 64; ++d) { dv[d] += p * gi[d]; dk[d] += ds * qi[d]; }
    }
  }
  for (int d = 0; d < 64; ++d) { out[C + d] = dk[d]; out[2 * C + d] = dv[d]; }
}
static int hp_attn_lds_attr(int NP) {                               // dynamic LDS beyond 64 KB needs the attribute (once per kernel)
  static OncePerDevice done; int done_dev;
  if (!done.need(done_dev)) return ATST_OK;
  hipError_t e = hipFuncSetAttribute((const void*)attn_hp_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * 64 * 4);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_hp_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (2 * 256 * 64 + 2 * 256) * 4);
  if (e != hipSuccess) return (int)e;
  done.done(done_dev); (void)NP;
  return ATST_OK;
}

// ---- token plumbing in fp32 ------------------------------------------------------------------------------------------------
// gradient of the token stage (token_grad_kernel of tokens.hip with an fp32 g0)
__global__ void token_grad_hp_kernel(const float* __restrict__ dx0, const uint8_t* __restrict__ rowflag, int S, int NP, int n_tok, int C, int use_cls,
                                     float* dcls, float* dpos, float* dbias, float* dmask, float* __restrict__ g0) {
  const int n = blockIdx.x;
  const bool is_cls = use_cls && n == 0;
  const bool is_patch = use_cls ? (n >= 1 && n <= n_tok) : (n < n_tok);
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float all = 0.f, un = 0.f, mk = 0.f;
    for (int s = 0; s < S; ++s) {
      const size_t row = (size_t)s * NP + n;
      const float v = dx0[row * C + c];
      const bool masked = rowflag && rowflag[row];
      all += v;
      if (masked) mk += v; else un += v;
      g0[row * C + c] = (is_patch && !masked) ? v : 0.f;
    }
    if (is_cls) atomicAdd(dcls + c, all);
    if (is_cls || is_patch) atomicAdd(dpos + (size_t)(use_cls ? n : n + 1) * C + c, all);
    if (is_patch) { atomicAdd(dbias + c, un); if (dmask && rowflag) atomicAdd(dmask + c, mk); }
  }
}

// ---- workspace ----------------------------------------------------------------------------------------------------------------
struct LayerHp { float *h1, *qkv, *o, *h2, *u, *a, *mean1, *rstd1, *mean2, *rstd2, *lse; };
struct WsHp {
  float* patches; float* table; float* x[2 * ATST_MAX_DEPTH + 1];
  LayerHp L[ATST_MAX_DEPTH];
  float *hN, *meanN, *rstdN;
  bf16 *a3, *w3, *r3a, *r3b;                     // split-bf16 scratch: A operand [M, 12 C], weight [4 C, 3 C], row-stacked dY / X [3 M, 4 C]
  float *dxA, *dxB, *g, *g2, *dh, *du, *dA, *dqkv, *d_o, *dout;
  size_t bytes;
};
struct Carver {
  char* base; size_t off;
  template <typename T> T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = reinterpret_cast<T*>(base + off);
    off += n * sizeof(T);
    return p;
  }
};
WsHp carve_hp(void* ws, int S, int NP, int C, int H, int depth, int PK) {
  WsHp w{};
  Carver c{reinterpret_cast<char*>(ws), 0};
  const size_t M = (size_t)S * NP;
  w.patches = c.take<float>(M * PK); w.table = c.take<float>((size_t)NP * C);
  for (int i = 0; i < 2 * depth + 1; ++i) w.x[i] = c.take<float>(M * C);
  for (int i = 0; i < depth; ++i) {
    LayerHp& l = w.L[i];
    l.h1 = c.take<float>(M * C); l.qkv = c.take<float>(M * 3 * C); l.o = c.take<float>(M * C); l.h2 = c.take<float>(M * C);
    l.u = c.take<float>(M * 4 * C); l.a = c.take<float>(M * 4 * C);
    l.mean1 = c.take<float>(M); l.rstd1 = c.take<float>(M); l.mean2 = c.take<float>(M); l.rstd2 = c.take<float>(M);
    l.lse = c.take<float>((size_t)S * H * NP);
  }
  w.hN = c.take<float>(M * C); w.meanN = c.take<float>(M); w.rstdN = c.take<float>(M);
  w.a3 = c.take<bf16>(M * (12 * C > 3 * PK ? 12 * C : 3 * PK)); w.w3 = c.take<bf16>((size_t)12 * C * C + 3 * (size_t)PK * C);
  w.r3a = c.take<bf16>(3 * M * 4 * C); w.r3b = c.take<bf16>(3 * M * (4 * C > PK ? 4 * C : PK));
  w.dxA = c.take<float>(M * C); w.dxB = c.take<float>(M * C); w.g = c.take<float>(M * C); w.g2 = c.take<float>(M * C);
  w.dh = c.take<float>(M * C); w.du = c.take<float>(M * 4 * C); w.dA = c.take<float>(M * 4 * C); w.dqkv = c.take<float>(M * 3 * C);
  w.d_o = c.take<float>(M * C); w.dout = c.take<float>(M * C);
  w.bytes = (c.off + 255) & ~(size_t)255;
  return w;
}

#define RUN(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)
#define LAUNCH_OK() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)
inline int grid_for(size_t n) { size_t b = (n + 255) / 256; return (int)(b < 4096 ? (b ? b : 1) : 4096); }

struct Hp {
  const atst_encoder_t* e; WsHp w; hipStream_t st; int M, C;
  // out[M, N] = epilogue(A[M, K] W[N, K]^T)  (forward Linear: W = the fp32 master at offset w_off)
  int linear(const float* A, int64_t w_off, int N, int K, int epi, float* out, const float* bias, const float* resid = nullptr,
             const float* row_scale = nullptr, int rps = 1) {
    RUN(atst_split3(A, M, K, 0, w.a3, st));
    RUN(atst_split3(e->p32 + w_off, N, K, 1, w.w3, st));
    return gemm3(N, K, epi, out, bias, resid, row_scale, rps);
  }
  // out[M, K] = dY[M, N] W[N, K]  (dgrad: the B operand is W^T)
  int linear_t(const float* dY, int64_t w_off, int N, int K, float* out) {
    RUN(atst_split3(dY, M, N, 0, w.a3, st));
    hipLaunchKernelGGL(split3_t_kernel, dim3(grid_for((size_t)N * K)), dim3(256), 0, st, e->p32 + w_off, N, K, w.w3);
    LAUNCH_OK();
    return gemm3(K, N, EPI_F32, out, nullptr, nullptr, nullptr, 1);
  }
  int gemm3(int N, int K, int epi, float* out, const float* bias, const float* resid, const float* row_scale, int rps) {
    GemmArgs a{};
    a.A = w.a3; a.B = w.w3; a.M = M; a.N = N; a.K = 3 * K; a.lda = 3 * K; a.ldb = 3 * K; a.epi = epi; a.C = out; a.ldc = N;
    a.bias = bias; a.resid = resid; a.row_scale = row_scale; a.rows_per_seq = rps;
    return atst_gemm_nt(a, st);
  }
  // dW[N, K] += dY[M, N]^T X[M, K]
  int wgrad(const float* dY, const float* X, int N, int K, float* dW) {
    hipLaunchKernelGGL(split3_rows_kernel, dim3(grid_for((size_t)M * N)), dim3(256), 0, st, dY, (size_t)M * N, 0, w.r3a);
    hipLaunchKernelGGL(split3_rows_kernel, dim3(grid_for((size_t)M * K)), dim3(256), 0, st, X, (size_t)M * K, 1, w.r3b);
    LAUNCH_OK();
    WgradArgs a{};
    a.dY = w.r3a; a.X = w.r3b; a.M = 3 * M; a.N = N; a.K = K; a.ldy = N; a.ldx = K; a.dW = dW; a.ldw = K; a.m_per_split = 0;
    return atst_gemm_tn(a, st);
  }
  int ln_fwd(const float* x, int64_t g_off, int64_t b_off, float* y, float* mean, float* rstd) {
    hipLaunchKernelGGL(ln_fwd_hp_kernel, dim3((M + 3) / 4), dim3(256), 0, st, x, e->p32 + g_off, e->p32 + b_off, y, mean, rstd, M, C);
    LAUNCH_OK();
    return ATST_OK;
  }
  int ln_bwd(const float* dy, const float* x, const float* mean, const float* rstd, int64_t g_off, const float* dres, float* dx, float* g,
             const float* row_scale, float* dgamma, float* dbeta, float* dbias_up) {
    hipLaunchKernelGGL(ln_bwd_hp_kernel, dim3((M + 4 * LNB_ROWS - 1) / (4 * LNB_ROWS)), dim3(256), 0, st, dy, x, mean, rstd, e->p32 + g_off, dres, dx, g, row_scale, e->NP,
                       dgamma, dbeta, dbias_up, M, C);
    LAUNCH_OK();
    return ATST_OK;
  }
};

inline int pk_of(const atst_encoder_t* e) { return (e->patch_h > 0 ? e->patch_h : 64) * (e->patch_w > 0 ? e->patch_w : 4); }
bool check_hp(const atst_encoder_t* e) {
  if (!e || e->depth < 1 || e->depth > ATST_MAX_DEPTH || e->C != e->H * 64 || (e->C != 384 && e->C != 768)) return false;
  if (e->NP != 32 && e->NP != 64 && e->NP != 128 && e->NP != 256) return false;
  if (e->n_tok + e->use_cls > e->NP || e->fp8) return false;
  return e->ws_bytes >= carve_hp(nullptr, e->S, e->NP, e->C, e->H, e->depth, pk_of(e)).bytes;
}
}  // namespace

extern "C" size_t atst_encoder_hp_ws_bytes(int S, int NP, int C, int H, int depth, int patch_h, int patch_w) {
  return carve_hp(nullptr, S, NP, C, H, depth, (patch_h > 0 ? patch_h : 64) * (patch_w > 0 ? patch_w : 4)).bytes;
}
extern "C" const float* atst_encoder_hp_out(const atst_encoder_t* e) { return carve_hp(e->ws, e->S, e->NP, e->C, e->H, e->depth, pk_of(e)).hN; }
extern "C" float* atst_encoder_hp_dout(const atst_encoder_t* e) { return carve_hp(e->ws, e->S, e->NP, e->C, e->H, e->depth, pk_of(e)).dout; }
extern "C" const float* atst_encoder_hp_block_out(const atst_encoder_t* e, int i) { return carve_hp(e->ws, e->S, e->NP, e->C, e->H, e->depth, pk_of(e)).x[2 * i + 2]; }

extern "C" int atst_encoder_hp_fwd(const atst_encoder_t* e, void* stream) {
  if (!check_hp(e)) return ATST_EINVAL;
  Hp hp{e, carve_hp(e->ws, e->S, e->NP, e->C, e->H, e->depth, pk_of(e)), reinterpret_cast<hipStream_t>(stream), e->S * e->NP, e->C};
  const WsHp& w = hp.w; hipStream_t st = hp.st;
  const int S = e->S, NP = e->NP, C = e->C, M = S * NP;
  const float* p = e->p32; const atst_enc_off_t& o = e->off;
  const int PK = pk_of(e);
  RUN(atst_patchify_f32(e->mel, S, e->width, NP, e->use_cls, w.patches, st, e->patch_h > 0 ? e->patch_h : 64, e->patch_w > 0 ? e->patch_w : 4));
  RUN(atst_token_table(e->use_cls ? p + o.cls_token : nullptr, p + o.pos_embed, p + o.patch_b, NP, e->n_tok, C, e->use_cls, w.table, st));
  {                                                     // x0 = (1 - m) (patch W^T + b) + m mask_embed + pos (+ CLS): the production EPI_PATCH epilogue
    RUN(atst_split3(w.patches, M, PK, 0, w.a3, st));
    RUN(atst_split3(p + o.patch_w, C, PK, 1, w.w3, st));
    GemmArgs a{};
    a.A = w.a3; a.B = w.w3; a.M = M; a.N = C; a.K = 3 * PK; a.lda = 3 * PK; a.ldb = 3 * PK; a.epi = EPI_PATCH; a.C = w.x[0]; a.ldc = C;
    a.bias = p + o.patch_b; a.rows_per_seq = NP; a.table = w.table; a.rowflag = e->rowflag; a.alt = p + o.mask_embed;
    RUN(atst_gemm_nt(a, st));
  }
  for (int i = 0; i < e->depth; ++i) {
    const atst_layer_off_t& lo = o.layer[i];
    const LayerHp& l = w.L[i];
    const float* s1 = e->dp_scale ? e->dp_scale + (size_t)(2 * i) * S : nullptr;
    const float* s2 = e->dp_scale ? e->dp_scale + (size_t)(2 * i + 1) * S : nullptr;
    RUN(hp.ln_fwd(w.x[2 * i], lo.ln1_w, lo.ln1_b, l.h1, l.mean1, l.rstd1));
    RUN(hp.linear(l.h1, lo.qkv_w, 3 * C, C, EPI_F32, l.qkv, nullptr));
    RUN(hp_attn_lds_attr(NP));
    hipLaunchKernelGGL(attn_hp_fwd_kernel, dim3(S * e->H), dim3(256), 2 * NP * 64 * 4, st, l.qkv, e->valid, l.o, l.lse, e->H, NP);
    LAUNCH_OK();
    RUN(hp.linear(l.o, lo.proj_w, C, C, EPI_RESID, w.x[2 * i + 1], p + lo.proj_b, w.x[2 * i], s1, NP));
    RUN(hp.ln_fwd(w.x[2 * i + 1], lo.ln2_w, lo.ln2_b, l.h2, l.mean2, l.rstd2));
    RUN(hp.linear(l.h2, lo.fc1_w, 4 * C, C, EPI_F32, l.u, p + lo.fc1_b));
    hipLaunchKernelGGL(gelu_hp_kernel, dim3(grid_for((size_t)M * 4 * C)), dim3(256), 0, st, l.u, (size_t)M * 4 * C, l.a);
    LAUNCH_OK();
    RUN(hp.linear(l.a, lo.fc2_w, C, 4 * C, EPI_RESID, w.x[2 * i + 2], p + lo.fc2_b, w.x[2 * i + 1], s2, NP));
    if (e->tap && i >= e->tap_first) {
      hipError_t rc = hipMemcpyAsync(e->tap + (size_t)(i - e->tap_first) * M * C, w.x[2 * i + 2], (size_t)M * C * sizeof(float),
                                     hipMemcpyDeviceToDevice, st);
      if (rc != hipSuccess) return (int)rc;
    }
  }
  return hp.ln_fwd(w.x[2 * e->depth], o.norm_w, o.norm_b, w.hN, w.meanN, w.rstdN);
}

extern "C" int atst_encoder_hp_bwd(const atst_encoder_t* e, void* stream) {
  if (!check_hp(e) || !e->g32) return ATST_EINVAL;
  Hp hp{e, carve_hp(e->ws, e->S, e->NP, e->C, e->H, e->depth, pk_of(e)), reinterpret_cast<hipStream_t>(stream), e->S * e->NP, e->C};
  const WsHp& w = hp.w; hipStream_t st = hp.st;
  const int S = e->S, NP = e->NP, C = e->C, M = S * NP, D = e->depth;
  float* G = e->g32; const atst_enc_off_t& o = e->off;
  auto dps = [&](int layer, int which) -> const float* { return e->dp_scale ? e->dp_scale + (size_t)(2 * layer + which) * S : nullptr; };
  float* cur = w.dxA; float* oth = w.dxB;
  // final LayerNorm: g = s2(D-1) * d(x_out of the last block), its column sums = fc2 bias gradient of the last block
  RUN(hp.ln_bwd(w.dout, w.x[2 * D], w.meanN, w.rstdN, o.norm_w, nullptr, cur, w.g, dps(D - 1, 1), G + o.norm_w, G + o.norm_b, G + o.layer[D - 1].fc2_b));
  for (int i = D - 1; i >= 0; --i) {
    const atst_layer_off_t& lo = o.layer[i];
    const LayerHp& l = w.L[i];
    // ---- MLP branch: x_out = x_mid + s2 (fc2(gelu(fc1(LN2(x_mid)))) + b2) ; w.g = s2 d(x_out)
    RUN(hp.linear_t(w.g, lo.fc2_w, C, 4 * C, w.dA));
    hipLaunchKernelGGL(dgelu_hp_kernel, dim3((4 * C + 255) / 256, M >= 16384 ? 256 : 16), dim3(256), 0, st, w.dA, l.u, M, 4 * C, w.du, G + lo.fc1_b);
    LAUNCH_OK();
    RUN(hp.linear_t(w.du, lo.fc1_w, 4 * C, C, w.dh));
    RUN(hp.ln_bwd(w.dh, w.x[2 * i + 1], l.mean2, l.rstd2, lo.ln2_w, cur, oth, w.g2, dps(i, 0), G + lo.ln2_w, G + lo.ln2_b, G + lo.proj_b));
    { float* t = cur; cur = oth; oth = t; }
    // ---- attention branch: x_mid = x_in + s1 (proj(attn(LN1(x_in))) + bp) ; w.g2 = s1 d(x_mid)
    RUN(hp.linear_t(w.g2, lo.proj_w, C, C, w.d_o));
    RUN(hp_attn_lds_attr(NP));
    hipLaunchKernelGGL(attn_hp_bwd_kernel, dim3(S * e->H), dim3(256), (2 * NP * 64 + 2 * NP) * 4, st, l.qkv, e->valid, l.o, l.lse, w.d_o, w.dqkv, e->H, NP);
    LAUNCH_OK();
    RUN(hp.wgrad(w.du, l.h2, 4 * C, C, G + lo.fc1_w));
    RUN(hp.wgrad(w.g, l.a, C, 4 * C, G + lo.fc2_w));
    RUN(hp.wgrad(w.dqkv, l.h1, 3 * C, C, G + lo.qkv_w));
    RUN(hp.wgrad(w.g2, l.o, C, C, G + lo.proj_w));
    RUN(hp.linear_t(w.dqkv, lo.qkv_w, 3 * C, C, w.dh));
    RUN(hp.ln_bwd(w.dh, w.x[2 * i], l.mean1, l.rstd1, lo.ln1_w, cur, oth, i > 0 ? w.g : nullptr, i > 0 ? dps(i - 1, 1) : nullptr,
                  G + lo.ln1_w, G + lo.ln1_b, i > 0 ? G + o.layer[i - 1].fc2_b : nullptr));
    { float* t = cur; cur = oth; oth = t; }
  }
  // ---- token stage: x0 = (1 - m) (patch W^T + b) + m mask_embed + pos (+ CLS)
  hipLaunchKernelGGL(token_grad_hp_kernel, dim3(NP), dim3(128), 0, st, cur, e->rowflag, S, NP, e->n_tok, C, e->use_cls,
                     e->use_cls ? G + o.cls_token : nullptr, G + o.pos_embed, G + o.patch_b, e->rowflag ? G + o.mask_embed : nullptr, w.g);
  LAUNCH_OK();
  return hp.wgrad(w.g, w.patches, C, pk_of(e), G + o.patch_w);
}
