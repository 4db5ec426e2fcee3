// BYOL projector / predictor support kernels and the cosine loss (the GEMMs themselves are gemm.hip).
// Reference: audiossl/models/atst/byol.py:6-22 (Linear -> BatchNorm1d(train) -> ReLU -> Linear), :24-78 (loss, monitors).
// Statistics are fp32; the cross-rank (SyncBatchNorm) combination of [mean, M2, count] and of [sum_dy, sum_dy_xhat] is
// done by the host between these kernels (tiny vectors), see audiossl_amd/engine.py.
#include <cstdlib>
#include "common.h"
#include "kernels.h"

namespace {

// part[y][n] = sum over the rows of row-block y of f(h[r,n])   block = 64 columns x 4 row groups.  The row-block partials
// are combined in a fixed order by bn_finalize_kernel: the batch statistics -- and with them the whole forward pass -- are
// bit-reproducible from run to run (fp32 atomics here let ~1e-7 noise flip ReLU gates behind the BatchNorm, which moved the
// gradients of identical inputs by 0.3 % between runs at 512 rows).
template <int MODE>  // 0: sum h ; 1: sum (h-mean)^2
__global__ __launch_bounds__(256) void bn_col_kernel(const float* __restrict__ h, int R, int N, const float* __restrict__ mean,
                                                     float* __restrict__ part) {
  // block = 256 columns (4 per thread: 16-B loads, 1 KB of a row per wave instruction) x 4 row groups; a column's rows are summed in the
  // same order as by the one-column-per-thread version of rounds 1-3 (bit-identical partials), at 5.3 instead of 3.7 TB/s
  __shared__ f32x4 red[4][64];
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int col = (blockIdx.x * 64 + c) * 4;
  const bool live = col < N;
  const f32x4 mu = (MODE == 1 && live) ? *reinterpret_cast<const f32x4*>(mean + col) : f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  if (live) {
    const int step = gridDim.y * 4;
    int r = blockIdx.y * 4 + rg;
    for (; r + 3 * step < R; r += 4 * step) {                       // four independent loads in flight, added in row order
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(h + (size_t)(r + u * step) * N + col);
#pragma unroll
      for (int u = 0; u < 4; ++u) { const f32x4 d = v[u] - mu; a += MODE == 1 ? d * d : d; }
    }
    for (; r < R; r += step) { const f32x4 d = *reinterpret_cast<const f32x4*>(h + (size_t)r * N + col) - mu; a += MODE == 1 ? d * d : d; }
  }
  red[rg][c] = a;
  __syncthreads();
  if (threadIdx.x < 64 && live) *reinterpret_cast<f32x4*>(part + (size_t)blockIdx.y * N + col) = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}
__global__ void bn_finalize_kernel(const float* __restrict__ part, int gy, int N, float scale, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float a = 0.f;
  for (int y = 0; y < gy; ++y) a += part[(size_t)y * N + i];
  out[i] = a * scale;
}
// what nn.BatchNorm1d(train) does with the batch statistics besides normalising, in one launch (it was eight element-wise torch kernels
// per head, 24 launches per step): rstd = rsqrt(M2 / count + eps) ; running_mean <- (1 - mom) running_mean + mom mean ;
// running_var <- (1 - mom) running_var + mom M2 / max(count - 1, 1) ; num_batches_tracked += 1.  count: the python float of a single
// rank, or a device scalar across ranks (never read back).  ref: torch.nn.BatchNorm1d inside build_mlp, audiossl/models/atst/byol.py:13-16.
__global__ void bn_finish_kernel(const float* __restrict__ mean, const float* __restrict__ m2, float count, const float* __restrict__ count_dev,
                                 float momentum, float eps, float* __restrict__ running_mean, float* __restrict__ running_var,
                                 long long* __restrict__ num_batches, float* __restrict__ rstd, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && num_batches) *num_batches += 1;
  if (i >= n) return;
  const float cnt = count_dev ? *count_dev : count;
  const float q = m2[i];
  rstd[i] = 1.0f / sqrtf(q / cnt + eps);
  if (running_mean) {
    running_mean[i] = running_mean[i] * (1.0f - momentum) + momentum * mean[i];
    running_var[i] = running_var[i] * (1.0f - momentum) + momentum * (q / fmaxf(cnt - 1.0f, 1.0f));
  }
}
__global__ void bn_apply_relu_kernel(const float* __restrict__ h, const float* __restrict__ mean, const float* __restrict__ rstd,
                                     const float* __restrict__ gamma, const float* __restrict__ beta, size_t total, int N,
                                     bf16* __restrict__ y) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % N);
    const float v = (h[i] - mean[c]) * rstd[c] * gamma[c] + beta[c];
    y[i] = f2bf(v > 0.f ? v : 0.f);
  }
}

// Round 6: the three element-wise BatchNorm kernels of the heads as COLUMN-FIXED row streams.  The flat-index forms above / below fetch the four (six)
// per-column parameter vectors again for every 16 B of data -- five to seven vector-memory instructions per useful one -- and ran at 2.5-2.8 TB/s on the
// ATST-Frame heads (41 k-83 k rows x 4096 columns: 7 % of the Frame step); here a thread owns four columns, keeps their parameters in registers and walks down
// the rows, two rows in flight.  Block = 64 column threads x 4 row lanes (1-KiB row segments per wave), grid (N / 256, row groups).  Same arithmetic per element.
#ifndef BN_CT
#define BN_CT 64
#endif
struct BnCols { f32x4 mu, rs, gm, bt; };
DEVFN BnCols bn_cols(const float* mean, const float* rstd, const float* gamma, const float* beta, int col) {
  return BnCols{*reinterpret_cast<const f32x4*>(mean + col), *reinterpret_cast<const f32x4*>(rstd + col), *reinterpret_cast<const f32x4*>(gamma + col),
                *reinterpret_cast<const f32x4*>(beta + col)};
}
template <bool SPLIT3>
__global__ __launch_bounds__(256) void bn_apply_relu_cols_kernel(const float* __restrict__ h, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta, int R, int N,
                                                                 bf16* __restrict__ y) {
  constexpr int CT = BN_CT, RL = 256 / CT;                         // column threads per block, row lanes
  const int c = threadIdx.x % CT, rg = threadIdx.x / CT;
  const int col = (blockIdx.x * CT + c) * 4;
  if (col >= N) return;
  const BnCols q = bn_cols(mean, rstd, gamma, beta, col);
  const int step = gridDim.y * RL;
  auto one = [&](int r, const f32x4& x) {
    bf16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = (x[e] - q.mu[e]) * q.rs[e] * q.gm[e] + q.bt[e];
      v = v > 0.f ? v : 0.f;
      hi[e] = f2bf(v);
      if constexpr (SPLIT3) lo[e] = f2bf(v - bf2f(hi[e]));
    }
    if constexpr (SPLIT3) {
      bf16* o = y + (size_t)r * 3 * (size_t)N + col;
      *reinterpret_cast<bf16x4*>(o) = hi; *reinterpret_cast<bf16x4*>(o + N) = lo; *reinterpret_cast<bf16x4*>(o + 2 * (size_t)N) = hi;
    } else {
      *reinterpret_cast<bf16x4*>(y + (size_t)r * N + col) = hi;
    }
  };
  int r = blockIdx.y * RL + rg;
  for (; r + 3 * step < R; r += 4 * step) {                        // four rows in flight
    f32x4 x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(h + (size_t)(r + k * step) * N + col));
#pragma unroll
    for (int k = 0; k < 4; ++k) one(r + k * step, x[k]);
  }
  for (; r < R; r += step) one(r, *reinterpret_cast<const f32x4*>(h + (size_t)r * N + col));
}
template <typename OUT>
__global__ __launch_bounds__(256) void bn_bwd_dx_cols_kernel(const float* __restrict__ dy, const float* __restrict__ h, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ sum_dy, const float* __restrict__ sum_dy_xhat, float inv_count,
                                                             int R, int N, OUT* __restrict__ dh) {
  typedef OUT out4 __attribute__((ext_vector_type(4)));
  constexpr int CT = BN_CT, RL = 256 / CT;
  const int c = threadIdx.x % CT, rg = threadIdx.x / CT;
  const int col = (blockIdx.x * CT + c) * 4;
  if (col >= N) return;
  const BnCols q = bn_cols(mean, rstd, gamma, beta, col);
  const f32x4 s1 = *reinterpret_cast<const f32x4*>(sum_dy + col), s2 = *reinterpret_cast<const f32x4*>(sum_dy_xhat + col);
  const int step = gridDim.y * RL;
  auto one = [&](int r, const f32x4& hv, const f32x4& dv) {
    out4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (hv[e] - q.mu[e]) * q.rs[e];
      const float d = (xh * q.gm[e] + q.bt[e] > 0.f) ? dv[e] : 0.f;
      o[e] = (OUT)(q.gm[e] * q.rs[e] * (d - s1[e] * inv_count - xh * s2[e] * inv_count));
    }
    *reinterpret_cast<out4*>(dh + (size_t)r * N + col) = o;
  };
  int r = blockIdx.y * RL + rg;
  for (; r + 3 * step < R; r += 4 * step) {                        // four rows = eight loads in flight
    f32x4 hv[4], dv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      hv[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(h + (size_t)(r + k * step) * N + col));
      dv[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dy + (size_t)(r + k * step) * N + col));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) one(r + k * step, hv[k], dv[k]);
  }
  for (; r < R; r += step) one(r, *reinterpret_cast<const f32x4*>(h + (size_t)r * N + col), *reinterpret_cast<const f32x4*>(dy + (size_t)r * N + col));
}
// grid of the column-fixed kernels: N / 256 column blocks x enough row groups for ~4096 blocks (at least 8 rows per thread when there are that many)
inline dim3 bn_cols_grid(int R, int N) {
  constexpr int CT = BN_CT, RL = 256 / CT;
  const int gx = (N + 4 * CT - 1) / (4 * CT);
  int gy = (R + 8 * RL - 1) / (8 * RL); const int cap = (4096 + gx - 1) / gx;
  if (gy > cap) gy = cap; if (gy < 1) gy = 1;
  return dim3(gx, gy);
}
inline bool bn_cols_on() { static const bool on = !(getenv("ATST_BN_COLS") && getenv("ATST_BN_COLS")[0] == '0'); return on; }   // ATST_BN_COLS=0: the flat-index kernels (A/B)

// same, but emits the split-bf16 operand [hi | lo | hi] (row stride 3N) for the following Linear (see split3_kernel)
__global__ void bn_apply_relu_split3_kernel(const float* __restrict__ h, const float* __restrict__ mean,
                                            const float* __restrict__ rstd, const float* __restrict__ gamma,
                                            const float* __restrict__ beta, size_t total, int N, bf16* __restrict__ y) {
  // four columns per thread: one 16-B load, three 8-B stores (N % 4 == 0)
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < total; i += (size_t)gridDim.x * blockDim.x * 4) {
    const int c = (int)(i % N);
    const size_t r = i / N;
    const f32x4 x = *reinterpret_cast<const f32x4*>(h + i);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), rs = *reinterpret_cast<const f32x4*>(rstd + c);
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c), bt = *reinterpret_cast<const f32x4*>(beta + c);
    bf16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = (x[e] - mu[e]) * rs[e] * gm[e] + bt[e];
      v = v > 0.f ? v : 0.f;
      hi[e] = f2bf(v); lo[e] = f2bf(v - bf2f(hi[e]));
    }
    bf16* o = y + r * 3 * (size_t)N + c;
    *reinterpret_cast<bf16x4*>(o) = hi; *reinterpret_cast<bf16x4*>(o + N) = lo; *reinterpret_cast<bf16x4*>(o + 2 * (size_t)N) = hi;
  }
}

__global__ __launch_bounds__(256) void bn_relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ h,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          int R, int N, float* __restrict__ part) {   // part[2][gridDim.y][N]
  // block = 256 columns (4 per thread, 16-B loads) x 4 row groups; same per-column row order as before (bit-identical partials)
  __shared__ f32x4 red[2][4][64];
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int col = (blockIdx.x * 64 + c) * 4;
  const bool live = col < N;
  f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
  if (live) {
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + col), rs = *reinterpret_cast<const f32x4*>(rstd + col);
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + col), bt = *reinterpret_cast<const f32x4*>(beta + col);
    const int step = gridDim.y * 4;
    auto acc = [&](const f32x4& hv, const f32x4& dv) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (hv[e] - mu[e]) * rs[e];
        const float d = (xh * gm[e] + bt[e] > 0.f) ? dv[e] : 0.f;
        a[e] += d; b[e] += d * xh;
      }
    };
    int r = blockIdx.y * 4 + rg;
    for (; r + 3 * step < R; r += 4 * step) {                       // four rows in flight (eight 16-B loads), accumulated in row order
      f32x4 hv[4], dv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        hv[k] = *reinterpret_cast<const f32x4*>(h + (size_t)(r + k * step) * N + col);
        dv[k] = *reinterpret_cast<const f32x4*>(dy + (size_t)(r + k * step) * N + col);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) acc(hv[k], dv[k]);
    }
    for (; r < R; r += step) acc(*reinterpret_cast<const f32x4*>(h + (size_t)r * N + col), *reinterpret_cast<const f32x4*>(dy + (size_t)r * N + col));
  }
  red[0][rg][c] = a; red[1][rg][c] = b;
  __syncthreads();
  if (threadIdx.x < 64 && live) {                                  // row-block partials, combined in fixed order (see bn_col_kernel):
    // the BatchNorm backward subtracts these batch sums from dy with ~100x cancellation, so fp32 atomic-order noise here
    // reached 2e-4 of the encoder's upstream gradient
    *reinterpret_cast<f32x4*>(part + (size_t)blockIdx.y * N + col) = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
    *reinterpret_cast<f32x4*>(part + ((size_t)gridDim.y + blockIdx.y) * N + col) = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
  }
}

template <typename OUT>   // bf16: operand of the next MFMA GEMMs ; float: parity mode (split-bf16 operands are made from it)
__global__ void bn_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ h, const float* __restrict__ mean,
                                 const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                 const float* __restrict__ sum_dy, const float* __restrict__ sum_dy_xhat, float inv_count,
                                 size_t total, int N, OUT* __restrict__ dh) {
  typedef OUT out4 __attribute__((ext_vector_type(4)));
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < total; i += (size_t)gridDim.x * blockDim.x * 4) {   // 4 columns per thread (N % 4 == 0)
    const int c = (int)(i % N);
    const f32x4 hv = *reinterpret_cast<const f32x4*>(h + i), dv = *reinterpret_cast<const f32x4*>(dy + i);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), rs = *reinterpret_cast<const f32x4*>(rstd + c);
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c), bt = *reinterpret_cast<const f32x4*>(beta + c);
    const f32x4 s1 = *reinterpret_cast<const f32x4*>(sum_dy + c), s2 = *reinterpret_cast<const f32x4*>(sum_dy_xhat + c);
    out4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (hv[e] - mu[e]) * rs[e];
      const float d = (xh * gm[e] + bt[e] > 0.f) ? dv[e] : 0.f;
      o[e] = (OUT)(gm[e] * rs[e] * (d - s1[e] * inv_count - xh * s2[e] * inv_count));
    }
    *reinterpret_cast<out4*>(dh + i) = o;
  }
}

// Split-bf16 operand for the Linear that feeds BatchNorm+ReLU.  x = hi + lo with hi = bf16(x), lo = bf16(x - hi);
// writing A' = [hi | lo | hi] and B' = [hi | hi | lo] side by side along K makes ONE bf16 MFMA GEMM over 3K compute
// x_hi W_hi + x_lo W_hi + x_hi W_lo, i.e. the product to ~2^-16 relative.  Needed because the ReLU gate behind the
// BatchNorm is discontinuous: a 2^-9 perturbation of h flips ~0.3 % of the gates and moves the gradient by ~sqrt of that
// (measured 4-11 %, DESIGN.md "Precision"); at 2^-16 the effect is below the rest of the bf16 pipeline.
__global__ void split3_kernel(const float* __restrict__ x, int R, int K, int b_layout, bf16* __restrict__ y) {
  const size_t total = (size_t)R * K;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / K; const int k = (int)(i % K);
    const float v = x[i];
    const bf16 hi = f2bf(v), lo = f2bf(v - bf2f(hi));
    bf16* o = y + r * 3 * (size_t)K + k;
    o[0] = hi; o[K] = b_layout ? hi : lo; o[2 * (size_t)K] = b_layout ? lo : hi;
  }
}

__global__ void cast_kernel(const float* __restrict__ x, size_t n, bf16* __restrict__ y) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = f2bf(x[i]);
}

// One wave per row, grid-stride.  Rows [0, ncrops*B) are student rows (view-major), rows [ncrops*B, ncrops*B + 2B)
// teacher rows.  acc[0] += sum over pairs <t_hat, s_hat> ; stats[0..D) student col sums, [D..2D) student col sq-sums,
// [2D..4D) teacher.  Monitor sums are kept in registers across the rows a wave visits and reduced once per block
// (ATST-Frame has ~170 k rows: one atomic per row-element serialised the whole kernel).
__global__ __launch_bounds__(256) void byol_loss_kernel(const float* __restrict__ student, const float* __restrict__ teacher,
                                                        int B, int ncrops, int nteach, float coef, float* __restrict__ acc,
                                                        float* __restrict__ dstudent, float* __restrict__ stats) {
  constexpr int D = 256;
  __shared__ float red[4][4 * D];
  __shared__ float racc[4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  // nteach == 2: cross-view pairs (iq != iv) ; nteach == 1: the asymmetric ATST-Frame loss, teacher view 0 vs student view 0
  const int ns = ncrops * B, total = ns + nteach * B;
  float st[4][4];                                          // [student sum, student sq, teacher sum, teacher sq][4 columns of this lane]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int e = 0; e < 4; ++e) st[a][e] = 0.f;
  float dots = 0.f;
  for (int row = wave; row < total; row += nwaves) {
    if (row >= ns) {                                       // teacher row: monitors only
      const f32x4 t = *reinterpret_cast<const f32x4*>(teacher + (size_t)(row - ns) * D + lane * 4);
      const float n = fmaxf(sqrtf(wave_sum(t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3])), 1e-12f);
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float v = t[e] / n; st[2][e] += v; st[3][e] += v * v; }
      continue;
    }
    const int iv = row / B, b = row % B;
    const f32x4 s = *reinterpret_cast<const f32x4*>(student + (size_t)row * D + lane * 4);
    const float ns_ = fmaxf(sqrtf(wave_sum(s[0] * s[0] + s[1] * s[1] + s[2] * s[2] + s[3] * s[3])), 1e-12f);
    float sh[4], T[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) sh[e] = s[e] / ns_;
    for (int iq = 0; iq < nteach; ++iq) {
      if (nteach == 2 && iq == iv) continue;
      const f32x4 t = *reinterpret_cast<const f32x4*>(teacher + (size_t)(iq * B + b) * D + lane * 4);
      const float n = fmaxf(sqrtf(wave_sum(t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3])), 1e-12f);
#pragma unroll
      for (int e = 0; e < 4; ++e) T[e] += t[e] / n;
    }
    const float dot = wave_sum(T[0] * sh[0] + T[1] * sh[1] + T[2] * sh[2] + T[3] * sh[3]);
    f32x4 g;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      g[e] = -coef * (T[e] - sh[e] * dot) / ns_;
      st[0][e] += sh[e]; st[1][e] += sh[e] * sh[e];
    }
    *reinterpret_cast<f32x4*>(dstudent + (size_t)row * D + lane * 4) = g;
    dots += dot;
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wid][a * D + lane * 4 + e] = st[a][e];
  if (lane == 0) racc[wid] = dots;
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * D; i += blockDim.x) atomicAdd(stats + i, red[0][i] + red[1][i] + red[2][i] + red[3][i]);
  if (threadIdx.x == 0) atomicAdd(acc, racc[0] + racc[1] + racc[2] + racc[3]);
}
}  // namespace

int atst_bn_stats(const float* h, int R, int N, float* mean, float* m2, float* scratch, hipStream_t st) {
  if (N % 64 || R <= 0 || !scratch) return ATST_EINVAL;
  int gy = (R + 63) / 64; if (gy > ATST_BN_ROW_BLOCKS) gy = ATST_BN_ROW_BLOCKS;
  hipLaunchKernelGGL(bn_col_kernel<0>, dim3((N + 255) / 256, gy), dim3(256), 0, st, h, R, N, (const float*)nullptr, scratch);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((N + 255) / 256), dim3(256), 0, st, (const float*)scratch, gy, N, 1.0f / R, mean);
  hipLaunchKernelGGL(bn_col_kernel<1>, dim3((N + 255) / 256, gy), dim3(256), 0, st, h, R, N, (const float*)mean, scratch);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((N + 255) / 256), dim3(256), 0, st, (const float*)scratch, gy, N, 1.0f, m2);
  return (int)hipGetLastError();
}
int atst_bn_finish(const float* mean, const float* m2, float count, const float* count_dev, float momentum, float eps, float* running_mean,
                   float* running_var, long long* num_batches, float* rstd, int n, hipStream_t st) {
  if (n <= 0 || !mean || !m2 || !rstd || (running_mean && !running_var)) return ATST_EINVAL;
  hipLaunchKernelGGL(bn_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, st, mean, m2, count, count_dev, momentum, eps, running_mean, running_var,
                     num_batches, rstd, n);
  return (int)hipGetLastError();
}
int atst_bn_apply_relu(const float* h, const float* mean, const float* rstd, const float* gamma, const float* beta,
                       int R, int N, bf16* y, hipStream_t st) {
  const size_t total = (size_t)R * N;
  if (N % 4 == 0 && bn_cols_on()) {
    hipLaunchKernelGGL(bn_apply_relu_cols_kernel<false>, bn_cols_grid(R, N), dim3(256), 0, st, h, mean, rstd, gamma, beta, R, N, y);
    return (int)hipGetLastError();
  }
  int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(bn_apply_relu_kernel, dim3(grid), dim3(256), 0, st, h, mean, rstd, gamma, beta, total, N, y);
  return (int)hipGetLastError();
}
int atst_bn_apply_relu_split3(const float* h, const float* mean, const float* rstd, const float* gamma, const float* beta,
                              int R, int N, bf16* y, hipStream_t st) {
  const size_t total = (size_t)R * N;
  if (N % 4) return ATST_EINVAL;
  if (bn_cols_on()) {
    hipLaunchKernelGGL(bn_apply_relu_cols_kernel<true>, bn_cols_grid(R, N), dim3(256), 0, st, h, mean, rstd, gamma, beta, R, N, y);
    return (int)hipGetLastError();
  }
  int grid = (int)((total / 4 + 255) / 256); if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(bn_apply_relu_split3_kernel, dim3(grid), dim3(256), 0, st, h, mean, rstd, gamma, beta, total, N, y);
  return (int)hipGetLastError();
}
int atst_bn_relu_bwd(const float* dy, const float* h, const float* mean, const float* rstd, const float* gamma,
                     const float* beta, int R, int N, float* sum_dy, float* sum_dy_xhat, float* scratch, hipStream_t st) {
  if (N % 64 || R <= 0 || !scratch) return ATST_EINVAL;
  int gy = (R + 63) / 64; if (gy > ATST_BN_ROW_BLOCKS) gy = ATST_BN_ROW_BLOCKS;
  hipLaunchKernelGGL(bn_relu_bwd_kernel, dim3((N + 255) / 256, gy), dim3(256), 0, st, dy, h, mean, rstd, gamma, beta, R, N, scratch);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((N + 255) / 256), dim3(256), 0, st, (const float*)scratch, gy, N, 1.0f, sum_dy);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((N + 255) / 256), dim3(256), 0, st, (const float*)(scratch + (size_t)gy * N), gy, N, 1.0f, sum_dy_xhat);
  return (int)hipGetLastError();
}
int atst_bn_bwd_dx(const float* dy, const float* h, const float* mean, const float* rstd, const float* gamma,
                   const float* beta, const float* sum_dy, const float* sum_dy_xhat, float inv_count, int R, int N,
                   bf16* dh, hipStream_t st) {
  const size_t total = (size_t)R * N;
  if (N % 4) return ATST_EINVAL;
  if (bn_cols_on()) {
    hipLaunchKernelGGL(bn_bwd_dx_cols_kernel<bf16>, bn_cols_grid(R, N), dim3(256), 0, st, dy, h, mean, rstd, gamma, beta, sum_dy, sum_dy_xhat, inv_count, R, N, dh);
    return (int)hipGetLastError();
  }
  int grid = (int)((total / 4 + 255) / 256); if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(bn_bwd_dx_kernel<bf16>, dim3(grid), dim3(256), 0, st, dy, h, mean, rstd, gamma, beta, sum_dy, sum_dy_xhat,
                     inv_count, total, N, dh);
  return (int)hipGetLastError();
}
int atst_bn_bwd_dx_fp32(const float* dy, const float* h, const float* mean, const float* rstd, const float* gamma,
                        const float* beta, const float* sum_dy, const float* sum_dy_xhat, float inv_count, int R, int N,
                        float* dh, hipStream_t st) {
  const size_t total = (size_t)R * N;
  if (N % 4) return ATST_EINVAL;
  if (bn_cols_on()) {
    hipLaunchKernelGGL(bn_bwd_dx_cols_kernel<float>, bn_cols_grid(R, N), dim3(256), 0, st, dy, h, mean, rstd, gamma, beta, sum_dy, sum_dy_xhat, inv_count, R, N, dh);
    return (int)hipGetLastError();
  }
  int grid = (int)((total / 4 + 255) / 256); if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(bn_bwd_dx_kernel<float>, dim3(grid), dim3(256), 0, st, dy, h, mean, rstd, gamma, beta, sum_dy, sum_dy_xhat,
                     inv_count, total, N, dh);
  return (int)hipGetLastError();
}
int atst_split3(const float* x, int R, int K, int b_layout, bf16* y, hipStream_t st) {
  const size_t total = (size_t)R * K;
  if (total == 0) return ATST_OK;
  int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(split3_kernel, dim3(grid), dim3(256), 0, st, x, R, K, b_layout, y);
  return (int)hipGetLastError();
}
int atst_cast_f32_bf16(const float* x, size_t n, bf16* y, hipStream_t st) {
  if (n == 0) return ATST_OK;
  int grid = (int)((n + 255) / 256); if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(cast_kernel, dim3(grid), dim3(256), 0, st, x, n, y);
  return (int)hipGetLastError();
}
int atst_byol_loss(const float* student, const float* teacher, int B, int ncrops, int D, float* loss, float* dstudent,
                   float* stats, hipStream_t st) {
  // ncrops == -1: asymmetric ATST-Frame loss (methods/atstframe/byol.py:83-84): one teacher view against one student view
  const bool asym = ncrops == -1;
  if (D != 256 || B <= 0 || (!asym && ncrops < 2)) return ATST_EINVAL;
  const int nstu = asym ? 1 : ncrops, nteach = asym ? 1 : 2;
  const int npairs = asym ? 1 : 2 * ncrops - 2;
  const float coef = 2.0f / ((float)npairs * (float)B);
  hipMemsetAsync(loss, 0, sizeof(float), st);
  hipMemsetAsync(stats, 0, 4 * D * sizeof(float), st);
  const int rows = (nstu + nteach) * B;
  int grid = (rows + 3) / 4; if (grid > 512) grid = 512;
  hipLaunchKernelGGL(byol_loss_kernel, dim3(grid), dim3(256), 0, st, student, teacher, B, nstu, nteach, coef, loss, dstudent, stats);
  return (int)hipGetLastError();
}
