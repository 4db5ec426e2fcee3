// Token plumbing around the encoder: patch gather (bit-exact index map), per-token additive table (CLS / positional /
// bias), ragged row gather / scatter, column sums, and the gradient reduction of the token-embedding stage.
// Reference: audiossl/models/atst/audio_transformer.py:56-75 (PatchEmbed_v2), :153-186 (prepare_tokens);
//            audiossl/methods/atstframe/audio_transformer.py:161-207.
#include "common.h"
#include "kernels.h"

namespace {

// out[(s*NP + tok), k] , k = f*4 + t  <-  mel[s, 0, f, p*4 + t] , p = tok - use_cls ; rows without a patch are zero.
// One block = one sequence x 16 tokens: 64x64 fp32 tile through LDS so that both the mel reads (time-contiguous) and
// the patch-row writes (k-contiguous) are coalesced.
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ mel, int width, int NP, int use_cls,
                                                       int n_patch, bf16* __restrict__ out) {
  __shared__ float tile[64][65];
  const int s = blockIdx.y, tok0 = blockIdx.x * 16, tid = threadIdx.x;
  const int p0 = tok0 - use_cls;
  const float* m = mel + (size_t)s * 64 * width;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int idx = tid + 256 * i, f = idx >> 6, tt = idx & 63;
    const int p = p0 + (tt >> 2);
    float v = 0.f;
    if (p >= 0 && p < n_patch) v = m[(size_t)f * width + p * 4 + (tt & 3)];
    tile[f][tt] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i, pp = c >> 6, f = c & 63;
    const int tok = tok0 + pp;
    if (tok < NP) {
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = f2bf(tile[f][pp * 4 + e]);
      *reinterpret_cast<bf16x4*>(out + ((size_t)s * NP + tok) * 256 + f * 4) = o;
    }
  }
}

// Any single-row patch geometry (patch height = number of mel bands FH, patch width pw; K = FH pw, k = f pw + t): one block per
// (token, sequence), thread = k.  Used for everything but the shipped 64 x 4 patches (e.g. 128 bands x 8 frames, configs[4]).
template <typename OUT>
__global__ void patchify_generic_kernel(const float* __restrict__ mel, int FH, int pw, int width, int NP, int use_cls, int n_patch,
                                        OUT* __restrict__ out) {
  const int s = blockIdx.y, tok = blockIdx.x, p = tok - use_cls, K = FH * pw;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    float v = 0.f;
    if (p >= 0 && p < n_patch) v = mel[((size_t)s * FH + k / pw) * width + p * pw + k % pw];
    out[((size_t)s * NP + tok) * K + k] = (OUT)v;
  }
}

__global__ void token_table_kernel(const float* cls, const float* pos, const float* bias, int NP, int n_tok, int C,
                                   int use_cls, float* table) {
  const int n = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float v = 0.f;
    if (use_cls) {
      if (n == 0) v = cls[c] + pos[c];
      else if (n <= n_tok) v = bias[c] + pos[(size_t)n * C + c];
    } else if (n < n_tok) {
      v = bias[c] + pos[(size_t)(n + 1) * C + c];
    }
    table[(size_t)n * C + c] = v;
  }
}

__global__ void gather_rows_kernel(const bf16* __restrict__ src, const int* __restrict__ rows, int R, int C, float* __restrict__ dst) {
  const int r = blockIdx.x;
  const bf16* s = src + (size_t)rows[r] * C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) dst[(size_t)r * C + c] = bf2f(s[c]);
}
__global__ void scatter_rows_kernel(const float* __restrict__ src, const int* __restrict__ rows, int R, int C, bf16* __restrict__ dst) {
  const int r = blockIdx.x;
  bf16* d = dst + (size_t)rows[r] * C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) d[c] = f2bf(src[(size_t)r * C + c]);
}

// out[n] += sum_m x[m, n] ; block = 128 columns x (rows strided by gridDim.y * 4)
__global__ __launch_bounds__(256) void colsum_kernel(const bf16* __restrict__ x, int M, int N, int ld, float* __restrict__ out) {
  __shared__ float red[4][128];
  const int cp = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int col = blockIdx.x * 128 + cp * 2;
  float a0 = 0.f, a1 = 0.f;
  for (int m = blockIdx.y * 4 + rg; m < M; m += gridDim.y * 4) {
    const bf16x2 v = *reinterpret_cast<const bf16x2*>(x + (size_t)m * ld + col);
    a0 += bf2f(v[0]); a1 += bf2f(v[1]);
  }
  red[rg][cp * 2] = a0; red[rg][cp * 2 + 1] = a1;
  __syncthreads();
  if (threadIdx.x < 128) {
    const int c = threadIdx.x;
    atomicAdd(out + blockIdx.x * 128 + c, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
  }
}

// gradient of the token-embedding stage: g0 = bf16 gradient of the patch-embedding output (zero on masked / non-patch rows), and the sums over
// sequences for pos_embed (per token), cls_token, the patch bias (unmasked patch rows) and mask_embed (masked rows).
// block = TG_TOK consecutive tokens (x) a slice of sequences; threadIdx.y = one of 4 token slots (a slot walks tokens slot, slot + 4, ...: the four
// slots read four ADJACENT rows of a sequence), threadIdx.x = a 4-column group (16-B loads, 8-B stores), four sequences in flight per thread.
// Sums leave through LDS so that every atomic instruction touches 64 consecutive floats: fp32 atomics cost by the LINES an instruction
// touches (2 columns per lane and one token per block -- rounds 1-3 -- was 3.1 M atomics, 4 lines each, 1.5 M of them on the 12 lines of
// the patch bias: 236 us for 300 MB, 1.3 TB/s).
constexpr int TG_TOK = 8;
__global__ __launch_bounds__(768) void token_grad_kernel(const float* __restrict__ dx0, const uint8_t* __restrict__ rowflag, int S, int NP,
                                                         int n_tok, int C, int use_cls, float* dcls, float* dpos, float* dbias, float* dmask,
                                                         bf16* __restrict__ g0) {
  extern __shared__ float tg_red[];                                // [4][C]
  const int q = threadIdx.x, slot = threadIdx.y, c = q * 4, lin = slot * blockDim.x + q;   // lin in [0, C)
  const int s_per = (S + gridDim.y - 1) / gridDim.y;
  const int s_begin = blockIdx.y * s_per;
  const int s_end = s_begin + s_per < S ? s_begin + s_per : S;
  f32x4 un = {0.f, 0.f, 0.f, 0.f}, mk = un;
  constexpr int U = 4;
  for (int t = 0; t < TG_TOK / 4; ++t) {
    const int n = blockIdx.x * TG_TOK + 4 * t + slot;
    const bool tok_live = n < NP;
    const bool is_cls = use_cls && n == 0;
    const bool is_patch = use_cls ? (n >= 1 && n <= n_tok) : (n < n_tok);
    f32x4 all = {0.f, 0.f, 0.f, 0.f};
    if (tok_live) {
      for (int s0 = s_begin; s0 < s_end; s0 += U) {
        f32x4 v[U]; bool live[U], masked[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          live[u] = s0 + u < s_end;
          const size_t row = (size_t)(live[u] ? s0 + u : s_begin) * NP + n;
          v[u] = *reinterpret_cast<const f32x4*>(dx0 + row * C + c);
          masked[u] = rowflag && rowflag[row];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (!live[u]) continue;
          const size_t row = (size_t)(s0 + u) * NP + n;
          all += v[u];
          if (is_patch) { if (masked[u]) mk += v[u]; else un += v[u]; }
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = f2bf((masked[u] || !is_patch) ? 0.f : v[u][e]);
          *reinterpret_cast<bf16x4*>(g0 + row * C + c) = o;
        }
      }
    }
    // this slot's token sums -> LDS -> atomics on consecutive floats (a slot's row of tg_red is read back by the same slot's threads)
    __syncthreads();
    *reinterpret_cast<f32x4*>(tg_red + slot * C + c) = all;
    __syncthreads();
    if (tok_live && (is_cls || is_patch)) {
      const int pi = use_cls ? n : n + 1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int col = q + (int)blockDim.x * e;
        const float v = tg_red[slot * C + col];
        atomicAdd(dpos + (size_t)pi * C + col, v);
        if (is_cls) atomicAdd(dcls + col, v);
      }
    }
  }
  // patch-bias / mask-token sums: over the block's tokens and slots, one atomic per column per block
  __syncthreads();
  *reinterpret_cast<f32x4*>(tg_red + slot * C + c) = un;
  __syncthreads();
  atomicAdd(dbias + lin, tg_red[lin] + tg_red[C + lin] + tg_red[2 * C + lin] + tg_red[3 * C + lin]);
  if (dmask && rowflag) {
    __syncthreads();
    *reinterpret_cast<f32x4*>(tg_red + slot * C + c) = mk;
    __syncthreads();
    atomicAdd(dmask + lin, tg_red[lin] + tg_red[C + lin] + tg_red[2 * C + lin] + tg_red[3 * C + lin]);
  }
}
}  // namespace

int atst_patchify(const float* mel, int S, int width, int NP, int use_cls, bf16* out, hipStream_t st, int patch_h, int patch_w) {
  if (S <= 0) return ATST_OK;
  if (patch_h <= 0 || patch_w <= 0) return ATST_EINVAL;
  const int n_patch = (width - width % patch_w) / patch_w;
  if (n_patch + use_cls > NP) return ATST_EINVAL;
  if (patch_h == 64 && patch_w == 4)
    hipLaunchKernelGGL(patchify_kernel, dim3((NP + 15) / 16, S), dim3(256), 0, st, mel, width, NP, use_cls, n_patch, out);
  else
    hipLaunchKernelGGL(patchify_generic_kernel<bf16>, dim3(NP, S), dim3(256), 0, st, mel, patch_h, patch_w, width, NP, use_cls, n_patch, out);
  return (int)hipGetLastError();
}
int atst_patchify_f32(const float* mel, int S, int width, int NP, int use_cls, float* out, hipStream_t st, int patch_h, int patch_w) {
  if (S <= 0) return ATST_OK;
  if (patch_h <= 0 || patch_w <= 0) return ATST_EINVAL;
  const int n_patch = (width - width % patch_w) / patch_w;
  if (n_patch + use_cls > NP) return ATST_EINVAL;
  hipLaunchKernelGGL(patchify_generic_kernel<float>, dim3(NP, S), dim3(256), 0, st, mel, patch_h, patch_w, width, NP, use_cls, n_patch, out);
  return (int)hipGetLastError();
}
int atst_token_table(const float* cls, const float* pos, const float* bias, int NP, int n_tok, int C, int use_cls, float* table, hipStream_t st) {
  hipLaunchKernelGGL(token_table_kernel, dim3(NP), dim3(128), 0, st, cls, pos, bias, NP, n_tok, C, use_cls, table);
  return (int)hipGetLastError();
}
int atst_gather_rows(const bf16* src, const int* rows, int R, int C, float* dst, hipStream_t st) {
  if (R <= 0) return ATST_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(R), dim3(128), 0, st, src, rows, R, C, dst);
  return (int)hipGetLastError();
}
int atst_scatter_rows(const float* src, const int* rows, int R, int C, bf16* dst, hipStream_t st) {
  if (R <= 0) return ATST_OK;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(R), dim3(128), 0, st, src, rows, R, C, dst);
  return (int)hipGetLastError();
}
int atst_colsum_bf16(const bf16* x, int M, int N, int ld, float* out, hipStream_t st) {
  if (M <= 0) return ATST_OK;
  if (N % 128) return ATST_EINVAL;
  int gy = (M + 255) / 256; if (gy > 64) gy = 64;
  hipLaunchKernelGGL(colsum_kernel, dim3(N / 128, gy), dim3(256), 0, st, x, M, N, ld, out);
  return (int)hipGetLastError();
}
int atst_token_grad(const float* dx0, const uint8_t* rowflag, int S, int NP, int n_tok, int C, int use_cls,
                    float* dcls, float* dpos, float* dbias, float* dmask, bf16* g0, hipStream_t st) {
  if (S <= 0) return ATST_OK;
  if (C % 16 || C > 768) return ATST_EINVAL;
  const int gx = (NP + TG_TOK - 1) / TG_TOK;
  int gy = 1024 / gx; if (gy > S / 8) gy = S / 8; if (gy < 1) gy = 1;          // >= 8 sequences per block, ~1024 blocks
  hipLaunchKernelGGL(token_grad_kernel, dim3(gx, gy), dim3(C / 4, 4), 4 * C * sizeof(float), st, dx0, rowflag, S, NP, n_tok, C, use_cls,
                     dcls, dpos, dbias, dmask, g0);
  return (int)hipGetLastError();
}
