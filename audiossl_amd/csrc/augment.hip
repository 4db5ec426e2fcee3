// Batched device versions of the reference's host-side spectrogram augmentations, applied
// after collate to [B, 64, T] log-mel batches:
//   RandomResizeCrop  audiossl/transforms/byol_a.py:7-49  : zero "virtual canvas" with the input at its centre, crop
//                     [i:i+h, j:j+w], F.interpolate(bicubic, align_corners=True) back to the input size
//   Mixup             audiossl/transforms/byol_a.py:61-115 : log((1-a) exp(x) + a exp(z) + eps), z from a FIFO memory bank
// The random draws (i, j, h, w / bank index, window start, a) are made by the host layer and passed in, so the kernels
// are deterministic functions that are checked against goldens of the reference run with the same draws.
#include "common.h"
#include "kernels.h"

namespace {

// torch's cubic convolution coefficients (A = -0.75), aten/src/ATen/native/UpSample.h
DEVFN void cubic_coeffs(float t, float c[4]) {
  const float A = -0.75f;
  const float x0 = t + 1.0f, x1 = t, x2 = 1.0f - t, x3 = 2.0f - t;
  c[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
  c[1] = ((A + 2.0f) * x1 - (A + 3.0f)) * x1 * x1 + 1.0f;
  c[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
  c[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}

__global__ __launch_bounds__(256) void rrc_bicubic_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                         const int* __restrict__ params, int B, int H, int W, int CH, int CW) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * H * W) return;
  const int x = (int)(idx % W), y = (int)((idx / W) % H), b = (int)(idx / ((long)W * H));
  const int ci = params[4 * b], cj = params[4 * b + 1], h = params[4 * b + 2], w = params[4 * b + 3];
  const int y0 = (CH - H) / 2, x0 = (CW - W) / 2;                // input placed at the canvas centre
  const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  const float ry = sy * (float)y, rx = sx * (float)x;
  const int iy = (int)floorf(ry), ix = (int)floorf(rx);
  float cy[4], cx[4];
  cubic_coeffs(ry - (float)iy, cy);
  cubic_coeffs(rx - (float)ix, cx);
  const float* src = in + (size_t)b * H * W;
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    int yy = iy - 1 + a; yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);       // torch clamps taps to the (cropped) input
    const int gy = ci + yy - y0;
    float row = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      int xx = ix - 1 + c; xx = xx < 0 ? 0 : (xx > w - 1 ? w - 1 : xx);
      const int gx = cj + xx - x0;
      const float v = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? src[(size_t)gy * W + gx] : 0.f;   // canvas outside the input is zero
      row += cx[c] * v;
    }
    acc += cy[a] * row;
  }
  out[idx] = acc;
}

// Mixing window: Wm = min(W, Wz) frames; x frames [xstart, xstart + Wm) are mixed with z frames [zstart, zstart + Wm)
// (one of the two starts is 0); x frames outside the window only go through log(exp(.) + eps) (byol_a.py:72-77).
__global__ __launch_bounds__(256) void log_mixup_exp_kernel(const float* __restrict__ x, const float* __restrict__ bank,
                                                           const int* __restrict__ zidx, const int* __restrict__ zstart,
                                                           const int* __restrict__ xstart, const float* __restrict__ alpha,
                                                           float* __restrict__ out, int B, int H, int W, int Wz) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * H * W) return;
  const int t = (int)(idx % W), f = (int)((idx / W) % H), b = (int)(idx / ((long)W * H));
  const int Wm = W < Wz ? W : Wz, u = t - xstart[b];
  float e = expf(x[idx]);
  if (u >= 0 && u < Wm) {
    const float a = alpha[b];
    e = (1.0f - a) * e + a * expf(bank[((size_t)zidx[b] * H + f) * Wz + zstart[b] + u]);
  }
  out[idx] = logf(e + 1.1920928955078125e-07f);                    // torch.finfo(float32).eps
}
}  // namespace

int atst_rrc_bicubic(const float* in, float* out, const int* params, int B, int H, int W, int CH, int CW, hipStream_t st) {
  if (B <= 0) return ATST_OK;
  if (CH < H || CW < W) return ATST_EINVAL;
  const long n = (long)B * H * W;
  hipLaunchKernelGGL(rrc_bicubic_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, params, B, H, W, CH, CW);
  return (int)hipGetLastError();
}
int atst_log_mixup_exp(const float* x, const float* bank, const int* zidx, const int* zstart, const int* xstart, const float* alpha,
                       float* out, int B, int H, int W, int Wz, hipStream_t st) {
  if (B <= 0) return ATST_OK;
  const long n = (long)B * H * W;
  hipLaunchKernelGGL(log_mixup_exp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, bank, zidx, zstart, xstart, alpha, out, B, H, W, Wz);
  return (int)hipGetLastError();
}
