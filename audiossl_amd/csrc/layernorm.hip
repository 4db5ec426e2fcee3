// LayerNorm(eps=1e-6) forward / backward for the pre-LN blocks (ref: audiossl/modules/transformer.py:128,132,144-146;
// audiossl/models/atst/audio_transformer.py:113,201).  HBM-bound: one wave per token row, 8-byte coalesced accesses,
// statistics in fp32.  The backward also folds in the residual-gradient add, the DropPath row scale + bf16 cast of the
// gradient handed to the upstream linear layer, and that layer's bias gradient (column sum), so dx is read once.
#include "common.h"
#include "kernels.h"
#include "profile.h"

namespace {
constexpr float LN_EPS = 1e-6f;

template <int VPT, typename OUT = bf16>   // values per lane = C / 64 ; OUT = bf16 (GEMM operand) or float (inference taps)
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, OUT* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out, int M,
                                                     uint8_t* __restrict__ y8 = nullptr, float s8 = 1.0f) {
  constexpr int C = VPT * 64;
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  float gm[VPT], bt[VPT];
#pragma unroll
  for (int i = 0; i < VPT / 2; ++i) {
    const f32x2 g2 = *reinterpret_cast<const f32x2*>(gamma + i * 128 + lane * 2);
    const f32x2 b2 = *reinterpret_cast<const f32x2*>(beta + i * 128 + lane * 2);
    gm[2 * i] = g2[0]; gm[2 * i + 1] = g2[1]; bt[2 * i] = b2[0]; bt[2 * i + 1] = b2[1];
  }
  for (int row = wave; row < M; row += nwaves) {
    const float* xr = x + (size_t)row * C;
    float v[VPT];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPT / 2; ++i) {
      const f32x2 t = *reinterpret_cast<const f32x2*>(xr + i * 128 + lane * 2);
      v[2 * i] = t[0]; v[2 * i + 1] = t[1]; s += t[0] + t[1];
    }
    const float mu = wave_sum(s) * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) { const float d = v[i] - mu; q += d * d; }
    const float rs = rsqrtf(wave_sum(q) * (1.0f / C) + LN_EPS);
    OUT* yr = y + (size_t)row * C;
#pragma unroll
    for (int i = 0; i < VPT / 2; ++i) {
      const float o0 = (v[2 * i] - mu) * rs * gm[2 * i] + bt[2 * i], o1 = (v[2 * i + 1] - mu) * rs * gm[2 * i + 1] + bt[2 * i + 1];
      if constexpr (sizeof(OUT) == 2) {
        bf16x2 o; o[0] = f2bf(o0); o[1] = f2bf(o1);
        *reinterpret_cast<bf16x2*>(yr + i * 128 + lane * 2) = o;
        if (y8) {                                                  // fp8 forward: e4m3 copy of the SAME bf16 values (the next GEMM's A operand)
          const int q = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(bf2f(o[0]) * s8, -448.f, 448.f),
                                                        __builtin_amdgcn_fmed3f(bf2f(o[1]) * s8, -448.f, 448.f), 0, false);
          *reinterpret_cast<unsigned short*>(y8 + (size_t)row * C + i * 128 + lane * 2) = (unsigned short)q;
        }
      } else {
        *reinterpret_cast<f32x2*>(yr + i * 128 + lane * 2) = f32x2{o0, o1};
      }
    }
    if (lane == 0 && mean_out) { mean_out[row] = mu; rstd_out[row] = rs; }
  }
}

template <int VPT>
__global__ __launch_bounds__(256) void ln_bwd_kernel(LnBwdArgs p) {
  constexpr int C = VPT * 64;
  __shared__ float red[3][4][C];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  float gm[VPT], dg[VPT], db[VPT], du[VPT];
#pragma unroll
  for (int i = 0; i < VPT; ++i) { gm[i] = p.gamma[(i >> 1) * 128 + lane * 2 + (i & 1)]; dg[i] = db[i] = du[i] = 0.f; }
  const float s8 = (p.g8 && p.g8_scale) ? *p.g8_scale : 1.0f;     // fp8 dgrad: delayed-scaling quantisation scale of g
  float gmax = 0.f;
  for (int row = wave; row < p.M; row += nwaves) {
    const size_t base = (size_t)row * C;
    const float mu = p.mean[row], rs = p.rstd[row];
    float xh[VPT], dyg[VPT];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPT / 2; ++i) {
      const f32x2 xv = *reinterpret_cast<const f32x2*>(p.x + base + i * 128 + lane * 2);
      const bf16x2 dv = *reinterpret_cast<const bf16x2*>(p.dy + base + i * 128 + lane * 2);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int k = 2 * i + e;
        const float d = bf2f(dv[e]);
        xh[k] = (xv[e] - mu) * rs;
        dyg[k] = d * gm[k];
        dg[k] += d * xh[k]; db[k] += d;
        s1 += dyg[k]; s2 += dyg[k] * xh[k];
      }
    }
    const float c1 = wave_sum(s1) * (1.0f / C), c2 = wave_sum(s2) * (1.0f / C);
    const float sc = p.row_scale ? p.row_scale[row / p.rows_per_seq] : 1.0f;
#pragma unroll
    for (int i = 0; i < VPT / 2; ++i) {
      f32x2 o; bf16x2 gq;
      f32x2 rv = {0.f, 0.f};
      if (p.dres) rv = *reinterpret_cast<const f32x2*>(p.dres + base + i * 128 + lane * 2);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int k = 2 * i + e;
        o[e] = rv[e] + rs * (dyg[k] - c1 - xh[k] * c2);
        const float gs = o[e] * sc;
        gq[e] = f2bf(gs);
        du[k] += gs;
        gmax = fmaxf(gmax, fabsf(gs));
      }
      *reinterpret_cast<f32x2*>(p.dx + base + i * 128 + lane * 2) = o;
      if (p.g) *reinterpret_cast<bf16x2*>(p.g + base + i * 128 + lane * 2) = gq;
      if (p.g8) {                                                  // e4m3 copy of the SAME bf16 values (A operand of the fp8 dgrad GEMM)
        const int q = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(bf2f(gq[0]) * s8, -448.f, 448.f),
                                                      __builtin_amdgcn_fmed3f(bf2f(gq[1]) * s8, -448.f, 448.f), 0, false);
        *reinterpret_cast<unsigned short*>(p.g8 + base + i * 128 + lane * 2) = (unsigned short)q;
      }
    }
  }
  if (p.g_amax) {
    gmax = wave_max(gmax);
    if (lane == 0 && gmax > 0.f) atomicMax(reinterpret_cast<unsigned*>(p.g_amax), __float_as_uint(gmax));
  }
  // block reduction of the three column accumulators, then one atomic per column per block
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int col = (i >> 1) * 128 + lane * 2 + (i & 1);
    red[0][wid][col] = dg[i]; red[1][wid][col] = db[i]; red[2][wid][col] = du[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float a = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
    const float b = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
    atomicAdd(p.dgamma + c, a);
    atomicAdd(p.dbeta + c, b);
    if (p.dbias_up) atomicAdd(p.dbias_up + c, red[2][0][c] + red[2][1][c] + red[2][2][c] + red[2][3][c]);
  }
}

int ln_grid(int M) { int b = (M + 3) / 4; return b < 2048 ? b : 2048; }
}  // namespace

int atst_ln_fwd(const float* x, const float* gamma, const float* beta, bf16* y, float* mean, float* rstd, int M, int C, hipStream_t st,
                uint8_t* y8, float s8) {
  if (M <= 0) return ATST_OK;
  ProfScope ps(PK_LN_FWD, (double)M * C * (y8 ? 7.0 : 6.0), st);      // read fp32, write bf16 (+ e4m3)
  if (C == 384) hipLaunchKernelGGL(ln_fwd_kernel<6>, dim3(ln_grid(M)), dim3(256), 0, st, x, gamma, beta, y, mean, rstd, M, y8, s8);
  else if (C == 768) hipLaunchKernelGGL(ln_fwd_kernel<12>, dim3(ln_grid(M)), dim3(256), 0, st, x, gamma, beta, y, mean, rstd, M, y8, s8);
  else return ATST_EINVAL;
  return (int)hipGetLastError();
}

int atst_ln_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, int M, int C, hipStream_t st) {
  if (M <= 0) return ATST_OK;
  ProfScope ps(PK_LN_FWD, (double)M * C * 8.0, st);
  if (C == 384) hipLaunchKernelGGL((ln_fwd_kernel<6, float>), dim3(ln_grid(M)), dim3(256), 0, st, x, gamma, beta, y, (float*)nullptr, (float*)nullptr, M);
  else if (C == 768) hipLaunchKernelGGL((ln_fwd_kernel<12, float>), dim3(ln_grid(M)), dim3(256), 0, st, x, gamma, beta, y, (float*)nullptr, (float*)nullptr, M);
  else return ATST_EINVAL;
  return (int)hipGetLastError();
}

int atst_ln_bwd(const LnBwdArgs& a, hipStream_t st) {
  if (a.M <= 0) return ATST_OK;
  int grid = (a.M + 63) / 64; if (grid > 1024) grid = 1024; if (grid < 1) grid = 1;
  ProfScope ps(PK_LN_BWD, (double)a.M * a.C * (2.0 + 4.0 + (a.dres ? 4.0 : 0.0) + 4.0 + (a.g ? 2.0 : 0.0)), st);
  if (a.C == 384) hipLaunchKernelGGL(ln_bwd_kernel<6>, dim3(grid), dim3(256), 0, st, a);
  else if (a.C == 768) hipLaunchKernelGGL(ln_bwd_kernel<12>, dim3(grid), dim3(256), 0, st, a);
  else return ATST_EINVAL;
  return (int)hipGetLastError();
}
