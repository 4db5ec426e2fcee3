// Log-mel front end on gfx950:  waveform -> reflect-pad/frame/Hann -> 1024-point real FFT -> |.|^2 -> 64-band HTK mel
// -> 10 log10 -> (per clip) clamp at max-80 dB -> min-max to [-1,1].
// Restates torchaudio MelSpectrogram(sr, n_fft=1024, win_length=W, hop=160, f_min=60, f_max=7800, n_mels) ->
// AmplitudeToDB("power", top_db=80) -> MinMax(-79.6482, 50.6842) as called from audiossl/methods/atst/transform.py:14-18
// and audiossl/methods/atstframe/transform.py:14-22  (the reference runs this on CPU dataloader workers, 1 thread each).
// n_mels = 64 (shipped recipes) or 128 (the reference's `n_mels` / `sr` parameters, transform.py:14-16: BASELINE.json
// configs[4] "high-res mel"); the sample rate only enters through the filterbank table the caller supplies.
//
// HBM-bound by design (0.64 MB read + 0.26 MB written per clip-view), so no GEMM reshaping: one wave computes one frame
// as a 512-point complex Stockham radix-8 FFT (3 passes through LDS) of the even/odd-packed real signal, un-packs the
// 513 bins, and reduces the triangular filters from LDS.  A block buffers 32 frames x 64 bands in LDS so that the
// [64, T] output rows are written in 128-B segments.  The per-clip maximum is an atomicMax on an order-preserving key.
#include <cmath>
#include "common.h"
#include "kernels.h"
#include "profile.h"

namespace {

constexpr int NFFT = 1024, HOPS = 160, NBIN = 513;
constexpr int FPB = 32;                       // frames per block (8 per wave)
constexpr float DB_MIN = -79.6482f, DB_MAX = 50.6842f, TOP_DB = 80.0f, DB_BIAS = 200.0f;

__device__ f32x2 g_tw512[512];               // exp(-2 pi i j / 512)
__device__ f32x2 g_tw1024[NBIN];             // exp(-2 pi i k / 1024), k <= 512

// Complex numbers are 2-vectors so that every operation is ONE packed instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 with the
// swizzles and sign flips in their op_sel / neg modifiers).  Written on HIP's float2 struct the same arithmetic went through LLVM's SLP
// vectoriser, which built each complex product from two packed FMAs (sum and difference) and stitched the halves together with v_mov:
// 177 moves in a kernel that is bound by VALU issue.
typedef f32x2 cpx;
DEVFN cpx mk(float re, float im) { return cpx{re, im}; }
DEVFN cpx cmul(cpx a, cpx b) { return __builtin_elementwise_fma(a.yy, cpx{-b.y, b.x}, a.xx * b); }   // (ax bx - ay by, ax by + ay bx)
DEVFN cpx rot90(cpx b) { return cpx{-b.y, b.x}; }                          // i b
DEVFN cpx cmul_r(cpx a, cpx b, cpx ib) { return __builtin_elementwise_fma(a.yy, ib, a.xx * b); }   // a * b with i b supplied (frame-invariant twiddles)
DEVFN cpx cadd(cpx a, cpx b) { return a + b; }
DEVFN cpx csub(cpx a, cpx b) { return a - b; }
DEVFN cpx mul_mi(cpx a) { return cpx{a.y, -a.x}; }                          // a * (-i)

// 8-point DFT, natural-order in / natural-order out (three radix-2 layers)
DEVFN void dft8(cpx* u) {
  const float s = 0.70710678118654752f;
  cpx a[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = cadd(u[i], u[i + 4]); a[i + 4] = csub(u[i], u[i + 4]); }
  a[5] = (a[5] + mul_mi(a[5])) * s;                                        // a5 * (s, -s)
  a[6] = mul_mi(a[6]);
  a[7] = (mul_mi(a[7]) - a[7]) * s;                                        // a7 * (-s, -s)
  cpx b[8];
#pragma unroll
  for (int h = 0; h < 8; h += 4) {
    b[h] = cadd(a[h], a[h + 2]); b[h + 2] = csub(a[h], a[h + 2]);
    b[h + 1] = cadd(a[h + 1], a[h + 3]); b[h + 3] = mul_mi(csub(a[h + 1], a[h + 3]));
  }
  cpx c[8];
#pragma unroll
  for (int h = 0; h < 8; h += 4) {
    c[h] = cadd(b[h], b[h + 1]); c[h + 1] = csub(b[h], b[h + 1]);
    c[h + 2] = cadd(b[h + 2], b[h + 3]); c[h + 3] = csub(b[h + 2], b[h + 3]);
  }
  u[0] = c[0]; u[1] = c[4]; u[2] = c[2]; u[3] = c[6]; u[4] = c[1]; u[5] = c[5]; u[6] = c[3]; u[7] = c[7];
}
// the same transform with the outputs left where the last layer produces them: v[j] is natural output BREV8(j); the callers store v[j] at
// the address of output BREV8(j), a compile-time index.
DEVFN constexpr int BREV8(int j) { return ((j & 1) << 2) | (j & 2) | ((j & 4) >> 2); }
DEVFN void dft8_brev(cpx* u) {
  const float s = 0.70710678118654752f;
  cpx a[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = cadd(u[i], u[i + 4]); a[i + 4] = csub(u[i], u[i + 4]); }
  a[5] = (a[5] + mul_mi(a[5])) * s;
  a[6] = mul_mi(a[6]);
  a[7] = (mul_mi(a[7]) - a[7]) * s;
  cpx b[8];
#pragma unroll
  for (int h = 0; h < 8; h += 4) {
    b[h] = cadd(a[h], a[h + 2]); b[h + 2] = csub(a[h], a[h + 2]);
    b[h + 1] = cadd(a[h + 1], a[h + 3]); b[h + 3] = mul_mi(csub(a[h + 1], a[h + 3]));
  }
#pragma unroll
  for (int h = 0; h < 8; h += 4) {
    u[h] = cadd(b[h], b[h + 1]); u[h + 1] = csub(b[h], b[h + 1]);
    u[h + 2] = cadd(b[h + 2], b[h + 3]); u[h + 3] = csub(b[h + 2], b[h + 3]);
  }
}

DEVFN int reflect(int i, int n) { i = i < 0 ? -i : i; return i >= n ? 2 * (n - 1) - i : i; }

// NB: mel bands per lane (n_mels = 64 NB).  MAXLEN > 0: the filter taps of this lane's bands live in REGISTERS (tap-major table
// [fb_maxlen][n_mels], fb_maxlen <= MAXLEN, zero beyond a band's length) and the triangular product is a fully unrolled chain of
// MAXLEN LDS reads at immediate offsets + FMAs; MAXLEN = 0: generic per-lane loop for longer filters.  Round 3 ran that loop for
// every frame -- 39 iterations of a dependent global load (a 64-line gather, band-major table) + LDS read + FMA that the compiler
// cannot pipeline across a per-lane trip count: ~40 % of the kernel (tools/mel_probe.py).
// Experiment builds (tools/mel_ablate.sh): ATST_MEL_ABL bit 0 = no waveform loads (constant input), 1 = no FFT passes 2 and 3, 2 = no real-transform
// un-pack / power, 3 = no filterbank products, 4 = no log10, 5 = no output stores.
#ifndef ATST_MEL_ABL
#define ATST_MEL_ABL 0
#endif
#ifndef ATST_MEL_TWO_BUFFERS
#define ATST_MEL_TWO_BUFFERS 0
#endif
#ifndef ATST_MEL_OCC
#define ATST_MEL_OCC 2                 // waves per SIMD the register allocation is held to.  3 (what the 54 KB of LDS would allow) spills 7 registers: 1170 vs 1069 us per 512 clips (profiles/r04_mel_probe.txt)
#endif
template <int NB, int MAXLEN>
__global__ __launch_bounds__(256, ATST_MEL_OCC) void stft_mel_db_kernel(const float* __restrict__ wave, int wave_ld, int n_samples, int T,
                                                          const float* __restrict__ window, const float* __restrict__ fbw,
                                                          const int* __restrict__ fb_start, const int* __restrict__ fb_len,
                                                          int fb_maxlen, float* __restrict__ out, unsigned int* __restrict__ clipmax) {
  // padded indices: pa(i) = i + i/8 (the radix-8 pass writes 8 consecutive points per lane: 64-B lane stride otherwise,
  // 8-way bank conflicts), pb(i) = i + 8 (i/64) (the second pass scatters groups of 8 lanes 512 B apart)
  // ONE transform buffer per wave, the three passes work in place: a wave's LDS instructions execute in order, so the reads of a pass (one
  // instruction for all 64 lanes) are served before the writes that follow them.  (Two buffers made the block 54 KB of LDS = two blocks per
  // CU whatever the register budget; 36 KB allows four.)
  __shared__ cpx fa[4][576];
#if ATST_MEL_TWO_BUFFERS
  __shared__ cpx fb[4][576];
#else
  cpx (*fb)[576] = fa;
#endif
  __shared__ float pw[4][NBIN + 7 + MAXLEN];                        // the tail is read (with zero weights) by bands whose filter is shorter than MAXLEN: kept zero
  constexpr int NMEL = 64 * NB;
  __shared__ float dbb[NMEL][FPB + 1];
  __shared__ float wmax[4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int clip = blockIdx.y, t0 = blockIdx.x * FPB;
  const float* w = wave + (size_t)clip * wave_ld;
  int m_start[NB], m_len[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) { m_start[b] = fb_start[lane + 64 * b]; m_len[b] = fb_len[lane + 64 * b]; }
  float vmax = -1e30f;
  float fw[NB][MAXLEN > 0 ? MAXLEN : 1];
  if constexpr (MAXLEN > 0) {
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int q = 0; q < MAXLEN; ++q) fw[b][q] = q < fb_maxlen ? fbw[q * NMEL + lane + 64 * b] : 0.f;
    for (int i = lane; i < 7 + MAXLEN; i += 64) pw[wid][NBIN + i] = 0.f;
  }
  // twiddles of the two twiddled passes depend on the lane only: fetched once per block, not once per frame
  cpx tw8[8], tw64[8], tw1k[9], tw8r[8], tw64r[8], tw1kr[9];      // ...r: i * twiddle, so that a complex product is two packed instructions with no sign flip
#pragma unroll
  for (int j = 0; j < 9; ++j) { tw1k[j] = g_tw1024[lane + 64 * j <= 512 ? lane + 64 * j : 512]; tw1kr[j] = rot90(tw1k[j]); asm volatile("" : "+v"(tw1kr[j])); }   // opaque: not re-derived (one v_xor) at every use
  f32x2 win[8];                                                    // this lane's 16 window taps
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    tw8[r] = g_tw512[(lane & 7) * r * 8]; tw64[r] = g_tw512[lane * r]; tw8r[r] = rot90(tw8[r]); tw64r[r] = rot90(tw64[r]);
    asm volatile("" : "+v"(tw8r[r]), "+v"(tw64r[r]));
    win[r] = *reinterpret_cast<const f32x2*>(window + 2 * (lane + 64 * r));
  }
  // every wave transforms its own frame in its own LDS arrays: the passes only need the wave's LDS writes to be visible
  // to its own lanes (in-order LDS pipe + lgkmcnt(0)), not a block barrier
  auto wave_sync = [] { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
  for (int fi = 0; fi < FPB / 4; ++fi) {
    const int fl = fi * 4 + wid;                       // frame within block
    int t = t0 + fl; if (t >= T) t = T - 1;             // duplicates are computed but never stored
    const int base = t * HOPS - NFFT / 2;
    cpx u[8];
#if ATST_MEL_ABL & 1
#pragma unroll
    for (int r = 0; r < 8; ++r) u[r] = mk(win[r][0] * (float)(t + r), win[r][1]);
    if (false) {
#else
    if (base >= 0 && base + NFFT <= n_samples && ((reinterpret_cast<size_t>(w + base) & 7) == 0)) {   // interior frame, 8-B aligned: vector loads
#endif
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const f32x2 x2 = *reinterpret_cast<const f32x2*>(w + base + 2 * (lane + 64 * r));
        u[r] = x2 * win[r];
      }
    } else {                                                       // frames that reach into the reflect padding
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int j0 = 2 * (lane + 64 * r);
        u[r] = mk(w[reflect(base + j0, n_samples)], w[reflect(base + j0 + 1, n_samples)]) * win[r];
      }
    }
    dft8_brev(u);                                       // pass p = 1 : no twiddles, out index lane*8 + r
#pragma unroll
    for (int r = 0; r < 8; ++r) fa[wid][lane * 9 + BREV8(r)] = u[r];      // pa(lane * 8 + r)
    wave_sync();
#if !(ATST_MEL_ABL & 2)
    {                                                   // pass p = 8
      const int k = lane & 7, j = (lane - k) * 8 + k;
#pragma unroll
      for (int r = 0; r < 8; ++r) u[r] = cmul_r(fa[wid][lane + (lane >> 3) + 72 * r], tw8[r], tw8r[r]);   // pa(lane + 64 r)
      dft8_brev(u);
#pragma unroll
      for (int r = 0; r < 8; ++r) fb[wid][j + BREV8(r) * 8 + (lane >> 3) * 8] = u[r];          // pb(j + 8 r), j + 8 r < 64 (lane >> 3) + 64
    }
    wave_sync();
    {                                                   // pass p = 64
#pragma unroll
      for (int r = 0; r < 8; ++r) u[r] = cmul_r(fb[wid][lane + 72 * r], tw64[r], tw64r[r]);                // pb(lane + 64 r)
      dft8_brev(u);
#pragma unroll
      for (int r = 0; r < 8; ++r) fa[wid][lane + 64 * BREV8(r)] = u[r];
    }
    wave_sync();
#endif
    // un-pack the real transform, power spectrum: bins k = lane + 64 j.  The twiddles of a lane's nine bins are frame-invariant and sit in
    // registers (tw1k): fetched inside the loop they were nine dependent global loads per frame, 300 of the kernel's 1075 us (profiles/r04_mel_ablate.txt)
#pragma unroll
    for (int j = 0; j < ((ATST_MEL_ABL & 4) ? 0 : 9); ++j) {
      const int k = lane + 64 * j;
      if (k <= 512) {
        const cpx zk = fa[wid][k & 511];
        cpx zc = fa[wid][(512 - k) & 511]; zc.y = -zc.y;
        const cpx e = (zk + zc) * 0.5f;
        const cpx o = mul_mi(zk - zc) * 0.5f;                       // d / (2i)
        const cpx x = cadd(e, cmul_r(o, tw1k[j], tw1kr[j]));
        pw[wid][k] = x.x * x.x + x.y * x.y;
      }
    }
    wave_sync();
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float mel = 0.f;
      if constexpr (MAXLEN > 0 && !(ATST_MEL_ABL & 8)) {
        const float* pp = &pw[wid][m_start[b]];
#pragma unroll
        for (int q = 0; q < MAXLEN; ++q) mel = fmaf(fw[b][q], pp[q], mel);     // same summation order as the loop
      } else if (!(ATST_MEL_ABL & 8)) {
        for (int q = 0; q < m_len[b]; ++q) mel += fbw[q * NMEL + lane + 64 * b] * pw[wid][m_start[b] + q];
      } else {
        mel = pw[wid][m_start[b]] + fw[b][0];
      }
#if ATST_MEL_ABL & 16
      const float db = mel;
#else
      const float db = 10.0f * log10f(fmaxf(mel, 1e-10f));
#endif
      dbb[lane + 64 * b][fl] = db;
      if (t0 + fl < T) vmax = fmaxf(vmax, db);
    }
  }
  vmax = wave_max(vmax);
  if (lane == 0) wmax[wid] = vmax;
  __syncthreads();
  if (tid == 0) {
    const float m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])) + DB_BIAS;   // > 0 : uint order == float order
    atomicMax(clipmax + clip, __float_as_uint(m));
  }
  float* o = out + (size_t)clip * NMEL * T;
  for (int i = tid; i < ((ATST_MEL_ABL & 32) ? 1 : NMEL * FPB); i += 256) {
    const int m = i / FPB, f = i % FPB;
    if (t0 + f < T) o[(size_t)m * T + t0 + f] = dbb[m][f];
  }
}

__global__ void db_finalize_kernel(float* __restrict__ x, const unsigned int* __restrict__ clipmax, int per_clip) {
  const int clip = blockIdx.y;
  const float floor_db = __uint_as_float(clipmax[clip]) - DB_BIAS - TOP_DB;
  float* p = x + (size_t)clip * per_clip;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < per_clip; i += gridDim.x * blockDim.x) {
    const float db = fmaxf(p[i], floor_db);
    p[i] = (db - DB_MIN) / (DB_MAX - DB_MIN) * 2.0f - 1.0f;
  }
}

// Twiddle tables are filled by a kernel on the caller's stream the first time the front end runs (double-precision
// sincos, rounded to fp32 exactly like the host tables used to be): no hipMemcpyToSymbol, so no entry point synchronises.
__global__ void twiddle_init_kernel() {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 512) { const double a = -2.0 * M_PI * i / 512.0; g_tw512[i] = f32x2{(float)cos(a), (float)sin(a)}; }
  if (i < NBIN) { const double a = -2.0 * M_PI * i / 1024.0; g_tw1024[i] = f32x2{(float)cos(a), (float)sin(a)}; }
}
// One fill per DEVICE (the __device__ tables are per device), and a launch on any OTHER stream first waits for the event
// recorded behind that fill -- a second front-end call on a side stream must not read tables that are still being written.
int init_twiddles(hipStream_t st) {
  constexpr int MAXDEV = 16;
  static bool done[MAXDEV] = {};
  static hipEvent_t ready[MAXDEV] = {};
  static hipStream_t first[MAXDEV] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return ATST_EINVAL;
  if (!done[dev]) {
    hipLaunchKernelGGL(twiddle_init_kernel, dim3((NBIN + 255) / 256), dim3(256), 0, st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    e = hipEventCreateWithFlags(&ready[dev], hipEventDisableTiming);
    if (e != hipSuccess) return (int)e;
    e = hipEventRecord(ready[dev], st);
    if (e != hipSuccess) return (int)e;
    first[dev] = st; done[dev] = true;
    return 0;
  }
  if (st != first[dev]) {                              // completed events cost nothing to wait on
    const hipError_t e = hipStreamWaitEvent(st, ready[dev], 0);
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}
}  // namespace

int atst_mel_frontend(const float* wave, int n_clips, int n_samples, int wave_ld, int n_mels, int win_length, const float* window,
                      const float* fb_weights, const int* fb_start, const int* fb_len, int fb_maxlen,
                      float* out, unsigned int* clipmax, hipStream_t st) {
  (void)win_length;                  // the zero-padded 1024-tap window is supplied by the caller ; fb_weights is [fb_maxlen][n_mels] (tap-major)
  if (wave_ld <= 0) wave_ld = n_samples;
  if (n_clips <= 0 || n_samples < NFFT / 2 + 1 || wave_ld < n_samples || (n_mels != 64 && n_mels != 128)) return ATST_EINVAL;
  int rc = init_twiddles(st);
  if (rc) return rc;
  const int T = 1 + n_samples / HOPS;
  hipMemsetAsync(clipmax, 0, n_clips * sizeof(unsigned int), st);
  ProfScope ps(PK_MEL, (double)n_clips * (4.0 * n_samples + 4.0 * n_mels * T), st);   // wave read once + dB written once
#define MEL_LAUNCH(NB_, ML_) hipLaunchKernelGGL((stft_mel_db_kernel<NB_, ML_>), dim3((T + FPB - 1) / FPB, n_clips), dim3(256), 0, st, wave, wave_ld, \
                                                n_samples, T, window, fb_weights, fb_start, fb_len, fb_maxlen, out, clipmax)
  if (n_mels == 64) {                                 // 16 kHz / 64 bands: longest filter 39 bins
#ifdef ATST_MEL_FORCE_LOOP
    MEL_LAUNCH(1, 0);
#else
    if (fb_maxlen <= 40) MEL_LAUNCH(1, 40); else MEL_LAUNCH(1, 0);
#endif
  } else {                                            // 32 kHz / 128 bands: 10 bins ; 16 kHz / 128 bands: ~20
    if (fb_maxlen <= 16) MEL_LAUNCH(2, 16); else if (fb_maxlen <= 40) MEL_LAUNCH(2, 40); else MEL_LAUNCH(2, 0);
  }
#undef MEL_LAUNCH
  hipLaunchKernelGGL(db_finalize_kernel, dim3(16, n_clips), dim3(256), 0, st, out, clipmax, n_mels * T);
  return (int)hipGetLastError();
}
