// Fused HF-AdamW step + EMA teacher update + bf16 shadow-copy refresh over the flat parameter buffers, and the batched
// bf16 transpose that produces the dgrad operand (W^T) copies.
// Reference: transformers.optimization.AdamW as called from audiossl/methods/atst/model.py:44-48 (eps 1e-6 added to the
// un-corrected sqrt(v), bias correction folded into the step size, decoupled weight decay applied AFTER the Adam update,
// parameters whose grad is None skipped) and ATST.update_teacher, audiossl/models/atst/atst.py:29-34.
// The reference runs ~160 tensors x 6 elementwise kernels for AdamW plus ~150 x 2 for the EMA; here it is one HBM pass:
// read p,g,m,v,t  write p,m,v,t + two bf16 copies  = 46 B / parameter.
#include "common.h"
#include "kernels.h"
#include "profile.h"

namespace {
__global__ __launch_bounds__(256) void adamw_ema_kernel(OptimArgs a) {
  const size_t nchunk = (a.n + 255) / 256;
  const int sub = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (size_t chunk = (size_t)blockIdx.x * 4 + sub; chunk < nchunk; chunk += (size_t)gridDim.x * 4) {
    const uint8_t fl = a.chunk_flags[chunk];
    const size_t i = chunk * 256 + lane * 4;
    f32x4 p = *reinterpret_cast<const f32x4*>(a.p + i);
    if (fl & 2) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(a.g + i);
      f32x4 m = *reinterpret_cast<const f32x4*>(a.m + i);
      f32x4 v = *reinterpret_cast<const f32x4*>(a.v + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ge = g[e] * a.grad_scale;
        m[e] = a.beta1 * m[e] + a.om_beta1 * ge;
        v[e] = a.beta2 * v[e] + a.om_beta2 * (ge * ge);
        p[e] = p[e] - a.step_size * (m[e] / (sqrtf(v[e]) + a.eps));
        if (fl & 1) p[e] = p[e] - a.lr_wd * p[e];
      }
      *reinterpret_cast<f32x4*>(a.m + i) = m;
      *reinterpret_cast<f32x4*>(a.v + i) = v;
      *reinterpret_cast<f32x4*>(a.p + i) = p;
    }
    if (a.p_bf16) {
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = f2bf(p[e]);
      *reinterpret_cast<bf16x4*>(a.p_bf16 + i) = o;
    }
    if (a.t && (fl & 4) && i < a.n_teacher) {
      f32x4 t = *reinterpret_cast<const f32x4*>(a.t + i);
      const float om = a.om_ema;
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] = t[e] * a.ema_m + om * p[e];
      *reinterpret_cast<f32x4*>(a.t + i) = t;
      if (a.t_bf16) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(t[e]);
        *reinterpret_cast<bf16x4*>(a.t_bf16 + i) = o;
      }
    }
  }
}

__global__ __launch_bounds__(256) void transpose_kernel(const bf16* __restrict__ src, int rows, int cols, bf16* __restrict__ dst) {
  __shared__ bf16 tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (r0 + r < rows && c0 + c < cols) ? src[(size_t)(r0 + r) * cols + c0 + c] : f2bf(0.f);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (r0 + r < rows && c0 + c < cols) dst[(size_t)(c0 + c) * rows + r0 + r] = tile[r][c];
  }
}
// All 2-D parameters of the flat buffer in one launch: table[i] = {element offset, rows, cols, first tile}; a block finds
// its matrix by scanning the (<= 128-entry) table.  Replaces one 10-us launch per weight matrix per step (58 of them).
__global__ __launch_bounds__(256) void transpose_batch_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst,
                                                              const int* __restrict__ table, int n) {
  __shared__ bf16 tile[64][66];
  int m = 0;
  for (int i = 1; i < n; ++i) if ((int)blockIdx.x >= table[4 * i + 3]) m = i;
  const int off = table[4 * m], rows = table[4 * m + 1], cols = table[4 * m + 2], t = blockIdx.x - table[4 * m + 3];
  const int tc = (cols + 63) / 64;
  const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
  const bf16* s = src + off; bf16* d = dst + off;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (r0 + r < rows && c0 + c < cols) ? s[(size_t)(r0 + r) * cols + c0 + c] : f2bf(0.f);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (r0 + r < rows && c0 + c < cols) d[(size_t)(c0 + c) * rows + r0 + r] = tile[r][c];
  }
}
}  // namespace

int atst_transpose_bf16_batch(const bf16* src, bf16* dst, const int* table, int n, int total_tiles, hipStream_t st) {
  if (n <= 0 || total_tiles <= 0) return ATST_OK;
  hipLaunchKernelGGL(transpose_batch_kernel, dim3(total_tiles), dim3(256), 0, st, src, dst, table, n);
  return (int)hipGetLastError();
}

int atst_adamw_ema(const OptimArgs& a, hipStream_t st) {
  if (a.n == 0 || (a.n % 256)) return ATST_EINVAL;
  const size_t nchunk = a.n / 256;
  int grid = (int)((nchunk + 3) / 4); if (grid > 4096) grid = 4096;
  ProfScope ps(PK_OPTIM, (double)a.n * 34.0 + (double)a.n_teacher * 10.0, st);
  hipLaunchKernelGGL(adamw_ema_kernel, dim3(grid), dim3(256), 0, st, a);
  return (int)hipGetLastError();
}
int atst_transpose_bf16(const bf16* src, int rows, int cols, bf16* dst, hipStream_t st) {
  hipLaunchKernelGGL(transpose_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, st, src, rows, cols, dst);
  return (int)hipGetLastError();
}


// ---- fp8 (OCP e4m3) operand preparation ----------------------------------------------------------------------------------
namespace {
__global__ void quant_fp8_kernel(const bf16* __restrict__ x, size_t n8, float scale, unsigned* __restrict__ y, unsigned* __restrict__ sat) {
  unsigned nclip = 0;
  // 8 bf16 in (16 B), 8 e4m3 out (8 B) per thread per iteration
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + i * 8);
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float t = bf2f(v[e]) * scale; nclip += fabsf(t) > 448.f ? 1u : 0u; f[e] = __builtin_amdgcn_fmed3f(t, -448.f, 448.f); }
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false); lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false); hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
    y[i * 2] = (unsigned)lo; y[i * 2 + 1] = (unsigned)hi;
  }
  f8_sat_add(sat, nclip);
}
__global__ void amax_batch_kernel(const float* __restrict__ p, const int* __restrict__ table, float* __restrict__ amax) {
  const int t = blockIdx.y;
  const size_t off = (size_t)table[2 * t], n = (size_t)table[2 * t + 1];
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(p[off + i]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(amax + t), __float_as_uint(m));   // non-negative floats order like their bits
}
__global__ void quant_weights_kernel(const float* __restrict__ p, const int* __restrict__ table, const float* __restrict__ amax,
                                     unsigned* __restrict__ p8, float* __restrict__ dq) {
  const int t = blockIdx.y;
  const size_t off = (size_t)table[2 * t], n4 = (size_t)table[2 * t + 1] / 4;
  const float a = fmaxf(amax[t], 1e-30f), s = 448.f / a;
  if (blockIdx.x == 0 && threadIdx.x == 0) dq[t] = a / 448.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p + off + i * 4);
    int r = 0;
    r = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v[0] * s, -448.f, 448.f), __builtin_amdgcn_fmed3f(v[1] * s, -448.f, 448.f), r, false);
    r = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v[2] * s, -448.f, 448.f), __builtin_amdgcn_fmed3f(v[3] * s, -448.f, 448.f), r, true);
    p8[(off + i * 4) / 4] = (unsigned)r;
  }
}
}  // namespace

// Gradient operands of the fp8 dgrad GEMMs (delayed scaling): y = e4m3(clamp(x * *scale)) when y != null, and
// *amax = max(*amax, max |x|) (float bits of a non-negative value order like unsigned) for the NEXT step's scale.
__global__ void quant_fp8_dyn_kernel(const bf16* __restrict__ x, size_t n8, const float* __restrict__ scale, unsigned* __restrict__ y,
                                     float* __restrict__ amax, unsigned* __restrict__ sat) {
  const float sc = scale ? *scale : 1.0f;
  float m = 0.f;
  unsigned nclip = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + i * 8);
    float f[8];
    float cm = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float t = bf2f(v[e]); cm = fmaxf(cm, fabsf(t)); f[e] = __builtin_amdgcn_fmed3f(t * sc, -448.f, 448.f); }
    m = fmaxf(m, cm);
    if (cm * sc > 448.f) {                                           // rare: count what was clipped
#pragma unroll
      for (int e = 0; e < 8; ++e) nclip += fabsf(bf2f(v[e]) * sc) > 448.f ? 1u : 0u;
    }
    if (y) {
      int lo = 0, hi = 0;
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false); lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false); hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
      y[i * 2] = (unsigned)lo; y[i * 2 + 1] = (unsigned)hi;
    }
  }
  if (amax) amax_post(amax, wave_max(m), threadIdx.x & 63, blockIdx.x * 4 + (threadIdx.x >> 6));   // amax: a SITE (AMAX_SITE_STRIDE floats)
  if (y) f8_sat_add(sat, nclip);
}
// after a backward: scale[i] = 448 / (margin * amax[i]) for the next step (unchanged where nothing was observed), amax[i] = 0
__global__ void fp8_update_scales_kernel(float* __restrict__ amax, float* __restrict__ scale, int n, float margin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = amax[i];
  if (a > 0.f) scale[i] = 448.0f / (margin * a);
  amax[i] = 0.f;
}
// e4m3 copy of the TRANSPOSED bf16 weight shadows (dgrad B operands) with the per-tensor scales of the forward copies (dq = amax / 448)
__global__ void quant_bf16_table_kernel(const bf16* __restrict__ p, const int* __restrict__ table, const float* __restrict__ dq,
                                        unsigned char* __restrict__ p8) {
  const int t = blockIdx.y;
  const size_t off = (size_t)table[2 * t], n = (size_t)table[2 * t + 1];
  const float sc = dq[t] > 0.f ? 1.0f / dq[t] : 0.f;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; i < n; i += (size_t)gridDim.x * blockDim.x * 2) {
    const bf16x2 v = *reinterpret_cast<const bf16x2*>(p + off + i);
    const int q = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(bf2f(v[0]) * sc, -448.f, 448.f),
                                                  __builtin_amdgcn_fmed3f(bf2f(v[1]) * sc, -448.f, 448.f), 0, false);
    *reinterpret_cast<unsigned short*>(p8 + off + i) = (unsigned short)q;
  }
}

int atst_quant_fp8_dyn(const bf16* x, size_t n, const float* scale, uint8_t* y, float* amax, hipStream_t st, unsigned* sat) {
  if (n == 0) return ATST_OK;
  if (n % 8) return ATST_EINVAL;
  int grid = (int)((n / 8 + 255) / 256); if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(quant_fp8_dyn_kernel, dim3(grid), dim3(256), 0, st, x, n / 8, scale, reinterpret_cast<unsigned*>(y), amax, sat);
  return (int)hipGetLastError();
}
int atst_fp8_update_scales(float* amax, float* scale, int n, float margin, hipStream_t st) {
  if (n <= 0) return ATST_OK;
  hipLaunchKernelGGL(fp8_update_scales_kernel, dim3((n + 255) / 256), dim3(256), 0, st, amax, scale, n, margin);
  return (int)hipGetLastError();
}
int atst_quant_bf16_table_fp8(const bf16* p16, const int* table, int n, const float* dq, uint8_t* p8, hipStream_t st) {
  if (n <= 0) return ATST_OK;
  hipLaunchKernelGGL(quant_bf16_table_kernel, dim3(32, n), dim3(256), 0, st, p16, table, dq, p8);
  return (int)hipGetLastError();
}
int atst_quant_fp8(const bf16* x, size_t n, float scale, uint8_t* y, hipStream_t st, unsigned* sat) {
  if (n == 0) return ATST_OK;
  if (n % 8) return ATST_EINVAL;
  int grid = (int)((n / 8 + 255) / 256); if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(quant_fp8_kernel, dim3(grid), dim3(256), 0, st, x, n / 8, scale, reinterpret_cast<unsigned*>(y), sat);
  return (int)hipGetLastError();
}
int atst_quant_weights_fp8(const float* p32, const int* table, int n, uint8_t* p8, float* dq, float* amax, hipStream_t st) {
  if (n <= 0) return ATST_OK;
  hipError_t e = hipMemsetAsync(amax, 0, sizeof(float) * n, st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(amax_batch_kernel, dim3(32, n), dim3(256), 0, st, p32, table, amax);
  hipLaunchKernelGGL(quant_weights_kernel, dim3(32, n), dim3(256), 0, st, p32, table, amax, reinterpret_cast<unsigned*>(p8), dq);
  return (int)hipGetLastError();
}
