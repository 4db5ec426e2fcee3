// LayerNorm(eps=1e-6) forward / backward for the pre-LN blocks (ref: audiossl/modules/transformer.py:128,132,144-146;
// audiossl/models/atst/audio_transformer.py:113,201).  HBM-bound: one wave per token row, 8-byte coalesced accesses,
// statistics in fp32.  The backward also folds in the residual-gradient add, the DropPath row scale + bf16 cast of the
// gradient handed to the upstream linear layer, and that layer's bias gradient (column sum), so dx is read once.
#include "common.h"
#include "kernels.h"
#include "profile.h"

namespace {
constexpr float LN_EPS = 1e-6f;

// Element map of a row: lane l owns W consecutive columns of every chunk of 64 W columns (W = 4 when C is a multiple of 256:
// 16-B fp32 / 8-B bf16 / 4-B e4m3 accesses; W = 2 for C = 384).  Half as many memory instructions per row as the 8-byte version
// (ln_bwd at C = 768: 36 -> 18): 558 -> see profiles / DESIGN.md.
template <int VPT> struct LnMap {
  static constexpr int W = VPT % 4 == 0 ? 4 : 2, CH = VPT / W;
  static DEVFN int col(int chunk, int lane) { return chunk * 64 * W + lane * W; }
};
template <int W> struct VecOf;
template <> struct VecOf<2> { typedef f32x2 f; typedef bf16x2 h; };
template <> struct VecOf<4> { typedef f32x4 f; typedef bf16x4 h; };
template <int W> DEVFN void st_fp8(uint8_t* dst, const float* v, float s) {      // v: the bf16-rounded values
  int q = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v[0] * s, -448.f, 448.f), __builtin_amdgcn_fmed3f(v[1] * s, -448.f, 448.f), 0, false);
  if constexpr (W == 2) { *reinterpret_cast<unsigned short*>(dst) = (unsigned short)q; }
  else {
    q = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v[2] * s, -448.f, 448.f), __builtin_amdgcn_fmed3f(v[3] * s, -448.f, 448.f), q, true);
    *reinterpret_cast<int*>(dst) = q;
  }
}

template <int VPT, typename OUT = bf16, typename IN = float>   // values per lane = C / 64 ; OUT = bf16 (GEMM operand) or float (inference taps) ; IN = float, or bf16: the residual stream of an fp8 inference pass (round 6)
__global__ __launch_bounds__(256) void ln_fwd_kernel(const IN* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, OUT* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out, int M,
                                                     uint8_t* __restrict__ y8 = nullptr, float s8c = 1.0f, unsigned* __restrict__ sat = nullptr,
                                                     const float* __restrict__ s8p = nullptr, float* __restrict__ amax8 = nullptr) {
  const float s8 = s8p ? *s8p : s8c;                              // fp8 forward: running (delayed) activation scale, or the constant
  float rmax = 0.f;
  using MP = LnMap<VPT>; constexpr int C = VPT * 64, W = MP::W, CH = MP::CH;
  typedef typename VecOf<W>::f fvec; typedef typename VecOf<W>::h hvec;
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  float gm[VPT], bt[VPT];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const fvec g2 = *reinterpret_cast<const fvec*>(gamma + MP::col(i, lane));
    const fvec b2 = *reinterpret_cast<const fvec*>(beta + MP::col(i, lane));
#pragma unroll
    for (int e = 0; e < W; ++e) { gm[W * i + e] = g2[e]; bt[W * i + e] = b2[e]; }
  }
  unsigned nclip = 0;                                             // fp8 forward: elements of the e4m3 copy clipped at +-448
  for (int row = wave; row < M; row += nwaves) {
    const IN* xr = x + (size_t)row * C;
    float v[VPT];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      if constexpr (sizeof(IN) == 2) {
        const hvec t = *reinterpret_cast<const hvec*>(xr + MP::col(i, lane));
#pragma unroll
        for (int e = 0; e < W; ++e) { v[W * i + e] = bf2f(t[e]); s += v[W * i + e]; }
      } else {
        const fvec t = *reinterpret_cast<const fvec*>(xr + MP::col(i, lane));
#pragma unroll
        for (int e = 0; e < W; ++e) { v[W * i + e] = t[e]; s += t[e]; }
      }
    }
    const float mu = wave_sum(s) * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) { const float d = v[i] - mu; q += d * d; }
    const float rs = rsqrtf(wave_sum(q) * (1.0f / C) + LN_EPS);
    OUT* yr = y + (size_t)row * C;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      float o[W];
#pragma unroll
      for (int e = 0; e < W; ++e) o[e] = (v[W * i + e] - mu) * rs * gm[W * i + e] + bt[W * i + e];
      if constexpr (sizeof(OUT) == 2) {
        hvec ob; float r[W];
#pragma unroll
        for (int e = 0; e < W; ++e) { ob[e] = f2bf(o[e]); r[e] = bf2f(ob[e]); }
        if (y) *reinterpret_cast<hvec*>(yr + MP::col(i, lane)) = ob;   // (fp8: not written when every reader takes the e4m3 copy)
        if (y8) {
          st_fp8<W>(y8 + (size_t)row * C + MP::col(i, lane), r, s8);   // fp8 forward: e4m3 copy of the SAME bf16 values (the next GEMM's A operand)
          float cm = 0.f;
#pragma unroll
          for (int e = 0; e < W; ++e) cm = fmaxf(cm, fabsf(r[e]));
          rmax = fmaxf(rmax, cm);
          if (cm * s8 > 448.f) {                                     // rare: something of this chunk was clipped -- count exactly
#pragma unroll
            for (int e = 0; e < W; ++e) nclip += fabsf(r[e] * s8) > 448.f ? 1u : 0u;
          }
        }
      } else {
        fvec of;
#pragma unroll
        for (int e = 0; e < W; ++e) of[e] = o[e];
        *reinterpret_cast<fvec*>(yr + MP::col(i, lane)) = of;
      }
    }
    if (lane == 0 && mean_out) { mean_out[row] = mu; rstd_out[row] = rs; }
  }
  if constexpr (sizeof(OUT) == 2) {
    if (y8) {
      f8_sat_add(sat, nclip);
      if (amax8) amax_post(amax8, wave_max(rmax), lane, wave);
    }
  }
}

template <int VPT>
__global__ __launch_bounds__(256) void ln_bwd_kernel(LnBwdArgs p) {
  using MP = LnMap<VPT>; constexpr int C = VPT * 64, W = MP::W, CH = MP::CH;
  typedef typename VecOf<W>::f fvec; typedef typename VecOf<W>::h hvec;
  __shared__ float red[3][4][C];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  float gm[VPT], dg[VPT], db[VPT], du[VPT];
#pragma unroll
  for (int i = 0; i < VPT; ++i) { gm[i] = p.gamma[MP::col(i / W, lane) + i % W]; dg[i] = db[i] = du[i] = 0.f; }
  const float s8 = (p.g8 && p.g8_scale) ? *p.g8_scale : 1.0f;     // fp8 dgrad: delayed-scaling quantisation scale of g
  float gmax = 0.f;
  for (int row = wave; row < p.M; row += nwaves) {
    const size_t base = (size_t)row * C;
    const float mu = p.mean[row], rs = p.rstd[row];
    float xh[VPT], dyg[VPT];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const fvec xv = *reinterpret_cast<const fvec*>(p.x + base + MP::col(i, lane));
      const hvec dv = *reinterpret_cast<const hvec*>(p.dy + base + MP::col(i, lane));
#pragma unroll
      for (int e = 0; e < W; ++e) {
        const int k = W * i + e;
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
    for (int i = 0; i < CH; ++i) {
      fvec o; hvec gq; float r[W];
      fvec rv;
#pragma unroll
      for (int e = 0; e < W; ++e) rv[e] = 0.f;
      if (p.dres) rv = *reinterpret_cast<const fvec*>(p.dres + base + MP::col(i, lane));
#pragma unroll
      for (int e = 0; e < W; ++e) {
        const int k = W * i + e;
        o[e] = rv[e] + rs * (dyg[k] - c1 - xh[k] * c2);
        const float gs = o[e] * sc;
        gq[e] = f2bf(gs); r[e] = bf2f(gq[e]);
        du[k] += gs;
        gmax = fmaxf(gmax, fabsf(gs));
      }
      *reinterpret_cast<fvec*>(p.dx + base + MP::col(i, lane)) = o;
      if (p.g) *reinterpret_cast<hvec*>(p.g + base + MP::col(i, lane)) = gq;
      if (p.g8) st_fp8<W>(p.g8 + base + MP::col(i, lane), r, s8);   // e4m3 copy of the SAME bf16 values (A operand of the fp8 dgrad GEMM)
    }
  }
  if (p.g_amax) amax_post(p.g_amax, wave_max(gmax), lane, blockIdx.x * 4 + wid);
  // block reduction of the three column accumulators, then one atomic per column per block
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int col = MP::col(i / W, lane) + i % W;
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
                uint8_t* y8, float s8, unsigned* sat, const float* s8p, float* amax8) {
  if (M <= 0) return ATST_OK;
  if (!y && !y8) return ATST_EINVAL;
  ProfScope ps(PK_LN_FWD, (double)M * C * (4.0 + (y ? 2.0 : 0.0) + (y8 ? 1.0 : 0.0)), st);      // read fp32, write bf16 and / or e4m3
  if (C == 384) hipLaunchKernelGGL(ln_fwd_kernel<6>, dim3(ln_grid(M)), dim3(256), 0, st, x, gamma, beta, y, mean, rstd, M, y8, s8, sat, s8p, amax8);
  else if (C == 768) hipLaunchKernelGGL(ln_fwd_kernel<12>, dim3(ln_grid(M)), dim3(256), 0, st, x, gamma, beta, y, mean, rstd, M, y8, s8, sat, s8p, amax8);
  else return ATST_EINVAL;
  return (int)hipGetLastError();
}

// the same LayerNorm on a bf16 residual stream (fp8 inference / teacher passes, round 6): reads 2 B per element instead of 4
int atst_ln_fwd_b16in(const bf16* x, const float* gamma, const float* beta, bf16* y, float* mean, float* rstd, int M, int C, hipStream_t st,
                      uint8_t* y8, float s8, unsigned* sat, const float* s8p, float* amax8) {
  if (M <= 0) return ATST_OK;
  if (!y && !y8) return ATST_EINVAL;
  ProfScope ps(PK_LN_FWD, (double)M * C * (2.0 + (y ? 2.0 : 0.0) + (y8 ? 1.0 : 0.0)), st);
  if (C == 384) hipLaunchKernelGGL((ln_fwd_kernel<6, bf16, bf16>), dim3(ln_grid(M)), dim3(256), 0, st, x, gamma, beta, y, mean, rstd, M, y8, s8, sat, s8p, amax8);
  else if (C == 768) hipLaunchKernelGGL((ln_fwd_kernel<12, bf16, bf16>), dim3(ln_grid(M)), dim3(256), 0, st, x, gamma, beta, y, mean, rstd, M, y8, s8, sat, s8p, amax8);
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
  ProfScope ps(PK_LN_BWD, (double)a.M * a.C * (2.0 + 4.0 + (a.dres ? 4.0 : 0.0) + 4.0 + (a.g ? 2.0 : 0.0) + (a.g8 ? 1.0 : 0.0)), st);
  if (a.C == 384) hipLaunchKernelGGL(ln_bwd_kernel<6>, dim3(grid), dim3(256), 0, st, a);
  else if (a.C == 768) hipLaunchKernelGGL(ln_bwd_kernel<12>, dim3(grid), dim3(256), 0, st, a);
  else return ATST_EINVAL;
  return (int)hipGetLastError();
}
