// bf16 (and e4m3) MFMA GEMMs for the ATST encoder / heads on gfx950.
//
//   gemm_nt  : C[M,N] = A[M,K] * B[N,K]^T  (+ fused epilogue)      forward GEMMs (weights are [out,in] row-major, torch
//              convention) and dgrad GEMMs (B = pre-transposed bf16 weight copy)
//   gemm_tn  : dW[N,K] += dY[M,N]^T * X[M,K]  (fp32 atomic accumulate, split over M)   wgrad GEMMs; both operands are
//              m-major in HBM, so fragments come from row-major LDS tiles through ds_read_b64_tr_b16.
//
// Kernels in this file (DESIGN.md section 3 has the measurements behind each choice):
//   gemm_nt_row384_kernel   256x384 (M >= 8192) or 128x384 output tile, 8 waves, operands HBM/L2 -> LDS by global_load_lds into
//                           a 3-stage (2-stage) ring of 32-deep k-tiles, XOR-swizzled source addresses, LDS-DMA issue spread
//                           between the MFMA groups; every N of the encoder is a multiple of 384, and for N == 384 a block owns
//                           whole rows: the residual epilogue also produces the next LayerNorm (forward) and the dgrad
//                           epilogue runs the LayerNorm backward (EPI_LNBWD).  e4m3 operands: same ring, MX-scaled MFMA.
//   gemm_nt_w4_kernel       the same wave tile (128x96) in 4-wave blocks, two per CU, for launches of <= 1.5 rounds
//   gemm_nt_kernel          128x128 / 256x128 tiles for N % 384 != 0 (heads) and the K = 384 dGELU GEMM; <.., KS = true>: split-K for fp32
//                           outputs with a handful of tiles and a long K (head Linears), partial tiles summed in a fixed order by
//                           splitk_reduce_kernel
//   gemm_tn_tall[_group]_kernel  192x384 weight-gradient tile, the four gradients of a block in one launch
//   gemm_tn_kernel          128x128 weight-gradient tile for the remaining shapes
// Measured-and-rejected variants (ping-pong main loop, register epilogue, 64-deep stages, phase skew, split loaders, phase
// tracers, ablation switches) live in tools/experiments/ (gemm_r02_variants.hip ; wgrad_schedule_variants.patch), not here.  One
// rejected variant of round 4 stays, behind hook 371, because a test keeps it bit-identical to the shipped path: the store-only bf16
// epilogue from transposed accumulators (template parameter TR of gemm_nt_row384_kernel; 1-10 % slower).
// Reference math being accelerated: nn.Linear in audiossl/modules/transformer.py:109,119,87-90 and
// audiossl/models/atst/audio_transformer.py:63,69 ; audiossl/models/atst/byol.py:13 ; LayerNorm backward of
// audiossl/modules/transformer.py:128,132,144-146.
#include "common.h"
#include "kernels.h"
#include "profile.h"

#include <mutex>
#include <type_traits>
#include <unordered_map>
#include <utility>

namespace {

constexpr int BN = 128, BK = 32;               // BK = 32: 64-B LDS rows (4 x 16-B chunks), XOR-swizzled
constexpr int C_LD = BN + 4;                   // fp32 epilogue staging tile [128][132] = 67,584 B (re-uses the operand LDS)

// Epilogue on 8 consecutive columns of one row.  The accumulator tile is staged through LDS so that every global access
// is a 16-B piece of a contiguous row segment, and the epilogue runs in two phases per staged part: (1) epi_fetch8 issues
// every global LOAD the part needs (residual / saved pre-activation / token table) into registers, (2) epilogue8 computes
// and stores.  On gfx9 loads and stores retire through one in-order counter (vmcnt): a load issued after a store cannot be
// waited on without also waiting for that store to be acknowledged by L2, so a load -> store -> load -> store sequence
// costs one full memory round trip per store (measured: that was ~half of every GEMM's time).
struct EpiAux { f32x4 a0, a1; float s; };
// Cache policy of the epilogue traffic (build-time A/B: ATST_EXTRA_FLAGS=-DATST_NT=<mask>): bit 0 bf16 stores of epilogue8, bit 1 its fp32
// stores, bit 2 the fp32 row stores of the row-wise epilogues, bit 3 the epilogue loads, bit 4 the bf16 rows of the row-wise epilogues -- non-temporal when set.
#ifndef ATST_NT
#define ATST_NT 127            // bits 5 / 6: the GELU activation a / the plain bf16 GEMM outputs on their own (A/B builds: profiles/r04_cache_policy_ab.txt)
#endif
template <int BIT, class T> DEVFN void st_pol(const T& v, T* dst) { if constexpr ((ATST_NT >> BIT) & 1) __builtin_nontemporal_store(v, dst); else *dst = v; }
template <int BIT, class T> DEVFN T ld_pol(const T* src) { if constexpr ((ATST_NT >> BIT) & 1) return __builtin_nontemporal_load(src); else return *src; }

// Block barrier for LDS hand-offs inside the epilogues.  __syncthreads() also drains vmcnt (workgroup-scope fence): every part
// would then wait for the acknowledgement of the global stores it has just issued and for the LDS-DMA refills in flight.
DEVFN void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#ifndef ATST_EPI_ABL              // experiment builds (tools/epi_ablate.sh): 1 = no GELU / dGELU arithmetic, 2 = no e4m3 copy, 4 = no u store / load
#define ATST_EPI_ABL 0
#endif
template <int EPI, bool SCALE = true>
DEVFN void epi_fetch8(const GemmArgs& p, int row, int col, EpiAux& x) {
  const size_t idx = (size_t)row * p.ldc + col;
  if constexpr (EPI == EPI_RESID) {
    if (p.resid_bf16) {                                            // bf16 residual stream (fp8 inference passes): one 16-B load of 8 values
      const bf16x8 r = ld_pol<3>(reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(p.resid) + idx));
      x.a0 = f32x4{bf2f(r[0]), bf2f(r[1]), bf2f(r[2]), bf2f(r[3])}; x.a1 = f32x4{bf2f(r[4]), bf2f(r[5]), bf2f(r[6]), bf2f(r[7])};
    } else {
    x.a0 = ld_pol<3>(reinterpret_cast<const f32x4*>(p.resid + idx));
    x.a1 = ld_pol<3>(reinterpret_cast<const f32x4*>(p.resid + idx + 4));
    }
    if constexpr (SCALE) x.s = p.row_scale ? p.row_scale[row / p.rows_per_seq] : 1.0f;   // else: the caller supplies it
  } else if constexpr (EPI == EPI_DGELU) {
#if ATST_EPI_ABL & 4
    x.a0 = f32x4{1.f, 1.f, 1.f, 1.f};
#else
    x.a0 = ld_pol<3>(reinterpret_cast<const f32x4*>(p.U + idx));       // 8 bf16 pre-activations
#endif
  } else if constexpr (EPI == EPI_PATCH) {
    const int tok = row % p.rows_per_seq;
    x.a0 = *reinterpret_cast<const f32x4*>(p.table + (size_t)tok * p.N + col);
    x.a1 = *reinterpret_cast<const f32x4*>(p.table + (size_t)tok * p.N + col + 4);
    x.s = (p.rowflag && p.rowflag[row]) ? 1.0f : 0.0f;
  }
}

// bias of the 8 columns starting at col (zeros when the GEMM has none)
DEVFN void epi_bias8(const GemmArgs& p, int col, f32x4& b0, f32x4& b1) {
  b0 = f32x4{0.f, 0.f, 0.f, 0.f}; b1 = b0;
  if (p.bias) { b0 = *reinterpret_cast<const f32x4*>(p.bias + col); b1 = *reinterpret_cast<const f32x4*>(p.bias + col + 4); }
}

template <int EPI>
DEVFN void epilogue8(const GemmArgs& p, int row, int col, f32x4 v0, f32x4 v1, const f32x4& b0, const f32x4& b1, const EpiAux& x,
                     f32x4& w0, f32x4& w1) {
  const size_t idx = (size_t)row * p.ldc + col;
  auto st_bf16 = [](void* base, size_t i, const f32x4& lo, const f32x4& hi4) {
    const float t[8] = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    const bf16x8 o = pack8(t);
    st_pol<0>(o, reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(base) + i));                      // streamed once: keep L2 for operands
  };
  auto st_bf16_b = [](void* base, size_t i, const f32x4& lo, const f32x4& hi4, auto bit) {           // the same with its own policy bit (A/B builds)
    const float t[8] = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    const bf16x8 o = pack8(t);
    st_pol<decltype(bit)::value>(o, reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(base) + i));
  };
  auto st_f32 = [](void* base, size_t i, const f32x4& lo, const f32x4& hi4) {
    st_pol<1>(lo, reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + i));
    st_pol<1>(hi4, reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + i + 4));
  };
  if constexpr (EPI == EPI_BF16) {
    st_bf16_b(p.C, idx, v0 + b0, v1 + b1, std::integral_constant<int, 6>{});    // bit 6: plain-bf16 GEMM outputs (qkv, dgrads)
  } else if constexpr (EPI == EPI_F32) {
    st_f32(p.C, idx, v0 + b0, v1 + b1);
  } else if constexpr (EPI == EPI_BIAS_GELU) {
    v0 += b0; v1 += b1;
#if !(ATST_EPI_ABL & 4)
    if (p.C) st_bf16(p.C, idx, v0, v1);                            // pre-activation u (saved for backward; skipped in inference)
#endif
    f32x4 g0, g1;
#if ATST_EPI_ABL & 1
    g0 = v0; g1 = v1;
#else
#pragma unroll
    for (int e = 0; e < 4; e += 2) {                                // pairs: packed FMAs (common.h gelu_bf16dst2)
      const f32x2 a0 = gelu_bf16dst2(f32x2{v0[e], v0[e + 1]}), a1 = gelu_bf16dst2(f32x2{v1[e], v1[e + 1]});
      g0[e] = a0[0]; g0[e + 1] = a0[1]; g1[e] = a1[0]; g1[e + 1] = a1[1];
    }
#endif
    if (p.C2) st_bf16_b(p.C2, idx, g0, g1, std::integral_constant<int, 5>{});    // activation a (bit 5: the next GEMM reads it whole); not written when every reader takes the e4m3 copy
#if ATST_EPI_ABL & 2
    if (p.q8) { w0 = g0; w1 = g1; asm volatile("" :: "v"(g0), "v"(g1)); } else
#endif
    if (p.q8) {                                                    // fp8 forward: e4m3 copy of the SAME bf16 values for the fc2 GEMM
      const float s = x.s;                                         // running (delayed) activation scale, or the constant: the caller read it once
      auto c = [&](float a_) { return __builtin_amdgcn_fmed3f(bf2f(f2bf(a_)) * s, -448.f, 448.f); };
      int lo = __builtin_amdgcn_cvt_pk_fp8_f32(c(g0[0]), c(g0[1]), 0, false); lo = __builtin_amdgcn_cvt_pk_fp8_f32(c(g0[2]), c(g0[3]), lo, true);
      int hi_w = __builtin_amdgcn_cvt_pk_fp8_f32(c(g1[0]), c(g1[1]), 0, false); hi_w = __builtin_amdgcn_cvt_pk_fp8_f32(c(g1[2]), c(g1[3]), hi_w, true);
      typedef int v2i_ __attribute__((ext_vector_type(2)));
      *reinterpret_cast<v2i_*>(p.q8 + idx) = v2i_{lo, hi_w};
      // elements beyond +-448 / scale are clipped: counted (f8_sat), but only on the rare path where the slot's largest value is one of them
      const float m8 = fmaxf(fmaxf(fmaxf(fabsf(g0[0]), fabsf(g0[1])), fmaxf(fabsf(g0[2]), fabsf(g0[3]))), fmaxf(fmaxf(fabsf(g1[0]), fabsf(g1[1])), fmaxf(fabsf(g1[2]), fabsf(g1[3]))));
      if (m8 * s > 447.f && p.q8_sat) {                            // (bf16 rounding of g can only move a value across 448 / s by 2^-8 relative: 447 is a safe trigger)
        unsigned nclip = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) nclip += (fabsf(bf2f(f2bf(g0[e])) * s) > 448.f ? 1u : 0u) + (fabsf(bf2f(f2bf(g1[e])) * s) > 448.f ? 1u : 0u);
        if (nclip) atomicAdd(p.q8_sat, nclip);
      }
      w0 = f32x4{m8, 0.f, 0.f, 0.f}; w1 = f32x4{0.f, 0.f, 0.f, 0.f};   // for the running amax of this site (row-384 kernel)
    }
  } else if constexpr (EPI == EPI_RESID) {
    if (p.out_bf16) st_bf16_b(p.C, idx, x.a0 + x.s * (v0 + b0), x.a1 + x.s * (v1 + b1), std::integral_constant<int, 4>{});   // (read back by the next LayerNorm: policy of the row-wise bf16 rows)
    else st_f32(p.C, idx, x.a0 + x.s * (v0 + b0), x.a1 + x.s * (v1 + b1));
  } else if constexpr (EPI == EPI_DGELU) {
    const bf16x8 u = __builtin_bit_cast(bf16x8, x.a0);
#if !(ATST_EPI_ABL & 1)
#pragma unroll
#endif
    for (int e = 0; e < ((ATST_EPI_ABL & 1) ? 0 : 4); e += 2) {     // pairs: packed FMAs (common.h gelu_grad_bf16dst2)
      const f32x2 d0 = gelu_grad_bf16dst2(f32x2{bf2f(u[e]), bf2f(u[e + 1])}), d1 = gelu_grad_bf16dst2(f32x2{bf2f(u[4 + e]), bf2f(u[5 + e])});
      v0[e] *= d0[0]; v0[e + 1] *= d0[1]; v1[e] *= d1[0]; v1[e + 1] *= d1[1];
    }
    if (p.C) st_bf16(p.C, idx, v0, v1);                             // du (bf16): not written when every reader takes the e4m3 copy (fp8 dgrad + e4m3 weight gradients)
    w0 = v0; w1 = v1;
#if ATST_EPI_ABL & 2
    if (p.q8) { asm volatile("" :: "v"(v0), "v"(v1)); } else
#endif
    if (p.q8) {                                                    // fp8 dgrad: e4m3 copy of the SAME bf16 values for the fc1 dgrad GEMM (scale in x.s)
      auto c = [&](float a_) { return __builtin_amdgcn_fmed3f(bf2f(f2bf(a_)) * x.s, -448.f, 448.f); };
      int lo = __builtin_amdgcn_cvt_pk_fp8_f32(c(v0[0]), c(v0[1]), 0, false); lo = __builtin_amdgcn_cvt_pk_fp8_f32(c(v0[2]), c(v0[3]), lo, true);
      int hi_w = __builtin_amdgcn_cvt_pk_fp8_f32(c(v1[0]), c(v1[1]), 0, false); hi_w = __builtin_amdgcn_cvt_pk_fp8_f32(c(v1[2]), c(v1[3]), hi_w, true);
      typedef int v2i_ __attribute__((ext_vector_type(2)));
      *reinterpret_cast<v2i_*>(p.q8 + idx) = v2i_{lo, hi_w};
    }
  } else if constexpr (EPI == EPI_PATCH) {
    f32x4 o0 = v0 + x.a0, o1 = v1 + x.a1;
    if (x.s != 0.f) {                                              // mask-token substitution (ATST-Frame)
      o0 = x.a0 - b0 + *reinterpret_cast<const f32x4*>(p.alt + col);
      o1 = x.a1 - b1 + *reinterpret_cast<const f32x4*>(p.alt + col + 4);
    }
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + idx) = o0;
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + idx + 4) = o1;
  }
}

// Geometry of one instantiation: BMT x 128 output tile, waves in a (BMT/WTM) x 2 grid, each wave WTM x 64
// (WTM = 128 halves the LDS fragment traffic per MFMA: 6 fragment reads feed 8 MFMAs instead of 4 feeding 4).
template <int BMT, int NSTG, int WTM> struct NtGeo {
  static constexpr int WAVES = (BMT / WTM) * 2, THREADS = WAVES * 64, MI = WTM / 32;
  static constexpr int A_BYTES = BMT * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  static constexpr int EPI_BYTES = 64 * C_LD * 4;                       // 64 rows of the fp32 tile at a time
  static constexpr int LDS = NSTG * STAGE > EPI_BYTES ? NSTG * STAGE : EPI_BYTES;
  static constexpr int A_IPW = (BMT / 16) / WAVES, B_IPW_NUM = BN / 16;  // 1-KiB load instructions (16 rows) per wave
  static constexpr int BLOCKS_PER_CU = 163840 / LDS > 4 ? 4 : 163840 / LDS;
  static constexpr int WPS = BLOCKS_PER_CU * WAVES / 4 > 8 ? 8 : BLOCKS_PER_CU * WAVES / 4;   // waves per SIMD to plan for
};

// KS: split-K instantiation (EPI_F32 only) -- a template parameter so that every other instantiation compiles to exactly what it was without it
// (as a run-time branch it cost the 256 x 128 tile of the ATST-Frame heads 10 %).
template <int EPI, int BMT, int NSTG, int WTM, bool KS = false>
__global__ __launch_bounds__((NtGeo<BMT, NSTG, WTM>::THREADS), (NtGeo<BMT, NSTG, WTM>::WPS)) void gemm_nt_kernel(GemmArgs p) {
  static_assert(!KS || EPI == EPI_F32, "split-K: fp32 output only");
  using G = NtGeo<BMT, NSTG, WTM>;
  constexpr int MI = G::MI;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1, hi = lane >> 5, l31 = lane & 31;

  const int ntn = p.N / BN;
  const int ntm = (p.M + BMT - 1) / BMT;
  // split-K (EPI_F32 only, p.ksplit > 1): blocks [s ntm ntn, (s + 1) ntm ntn) work on k-tiles [s kchunk, (s + 1) kchunk) of every output tile
  const int ksp = KS ? p.ksplit : 1;
  const int ks_id = ksp > 1 ? (int)blockIdx.x / (ntm * ntn) : 0;
  const int id = xcd_remap(ksp > 1 ? (int)blockIdx.x % (ntm * ntn) : (int)blockIdx.x, ntm * ntn);
  const int m0 = (id / ntn) * BMT, n0 = (id % ntn) * BN;
  const int nk_all = p.K / BK, kchunk = (nk_all + ksp - 1) / ksp, kt_first = ks_id * kchunk;

  // Operand tiles go HBM/L2 -> LDS directly (global_load_lds, 16 B per lane, 1 KiB = 16 rows per wave-instruction, no
  // staging registers), NSTG-1 K-tiles ahead of the MFMAs.  LDS rows are 64 B; chunk c of row r is stored at chunk
  // c ^ ((r >> 2) & 3) so that the 16-lane groups of ds_read_b128 hit 16 distinct 16-B bank slots; the permutation is
  // applied to the per-lane *source* address (the LDS destination of global_load_lds is lane-linear).
  typedef const void __attribute__((address_space(1))) * gptr_t;
  typedef void __attribute__((address_space(3))) * lptr_t;
  char* lds = smem_raw;
  constexpr int NB_PER_WAVE = (G::B_IPW_NUM + G::WAVES - 1) / G::WAVES;     // 2 (4 waves) or 1 (8 waves)
  const bf16* srcA[G::A_IPW]; const bf16* srcB[NB_PER_WAVE];
  const int lrow = lane >> 2, lchunk = lane & 3;
#pragma unroll
  for (int j = 0; j < G::A_IPW; ++j) {
    const int row = (wid * G::A_IPW + j) * 16 + lrow;
    int ra = m0 + row; ra = ra < p.M ? ra : p.M - 1;              // clamp: rows >= M are never stored
    srcA[j] = p.A + (size_t)ra * p.lda + (lchunk ^ ((row >> 2) & 3)) * 8 + kt_first * BK;
  }
#pragma unroll
  for (int j = 0; j < NB_PER_WAVE; ++j) {
    const int row = (wid * NB_PER_WAVE + j) * 16 + lrow;
    srcB[j] = p.B + (size_t)(n0 + row) * p.ldb + (lchunk ^ ((row >> 2) & 3)) * 8 + kt_first * BK;
  }
  constexpr int LOADS_PER_TILE = G::A_IPW + NB_PER_WAVE;          // per wave, in issue order
  auto issue = [&](int kt) {
    char* st = lds + (kt % NSTG) * G::STAGE;
#pragma unroll
    for (int j = 0; j < G::A_IPW; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(srcA[j] + kt * BK), (lptr_t)(st + (wid * G::A_IPW + j) * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < NB_PER_WAVE; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(srcB[j] + kt * BK), (lptr_t)(st + G::A_BYTES + (wid * NB_PER_WAVE + j) * 1024), 16, 0, 0);
  };

  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = nk_all - kt_first < kchunk ? (nk_all - kt_first > 0 ? nk_all - kt_first : 0) : kchunk;
  const int xr = (l31 >> 2) & 3;                                  // swizzle key of this lane's fragment rows
  const int offA = (wm * WTM + l31) * 64, offB = G::A_BYTES + (wn * 64 + l31) * 64;
#pragma unroll
  for (int t = 0; t < NSTG - 1; ++t)
    if (t < nk) issue(t);
  for (int kt = 0; kt < nk; ++kt) {
    // my share of tile kt has landed (loads complete in order; the younger NSTG-2 tiles may still be in flight);
    // barrier => everyone's share has, and everyone is done reading stage (kt-1) % NSTG, which the next issue overwrites
    const int younger = nk - 1 - kt < NSTG - 2 ? nk - 1 - kt : NSTG - 2;
    switch ((younger > 0 ? younger : 0) * LOADS_PER_TILE) {       // vmcnt needs an immediate
      case 0: asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory"); break;
    }
    if (kt + NSTG - 1 < nk) issue(kt + NSTG - 1);
    const char* st = lds + (kt % NSTG) * G::STAGE;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int co = ((ks * 2 + hi) ^ xr) << 4;
      bf16x8 af[MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[mi] = *reinterpret_cast<const bf16x8*>(st + offA + mi * 32 * 64 + co);
      bf16x8 b0 = *reinterpret_cast<const bf16x8*>(st + offB + co), b1 = *reinterpret_cast<const bf16x8*>(st + offB + 32 * 64 + co);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        acc[mi][0] = mfma32(af[mi], b0, acc[mi][0]);
        acc[mi][1] = mfma32(af[mi], b1, acc[mi][1]);
      }
    }
  }
  asm volatile("s_barrier" ::: "memory");                         // all operand reads done before the tile is staged

  // stage the fp32 tile through LDS, 64 rows at a time (part p = rows [64p, 64p+64) of the block tile)
  float* sC = reinterpret_cast<float*>(smem_raw);
  constexpr int RPP = G::THREADS / 16;                            // rows stored per pass (8 columns per thread)
  constexpr int NPASS = 64 / RPP;
  const int c8 = (tid & 15) * 8, rr = tid >> 4;
  f32x4 bias0, bias1;
  epi_bias8(p, n0 + c8, bias0, bias1);
  f32x4 csum0 = {0.f, 0.f, 0.f, 0.f}, csum1 = csum0;              // EPI_DGELU: column sums of du = fc1 bias gradient
  float omax = 0.f;                                               // EPI_DGELU: running max |du| (fp8 backward, recording step)
#pragma unroll
  for (int part = 0; part < BMT / 64; ++part) {
    EpiAux aux[NPASS];
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {                    // phase 1: every global load of this part (in flight over the staging)
      const int row = m0 + part * 64 + pass * RPP + rr;
      if (row < p.M) epi_fetch8<EPI>(p, row, n0 + c8, aux[pass]);
    }
    if (wm == part / (WTM / 64)) {
      constexpr int SUBS = WTM / 64;
#pragma unroll
      for (int sub = 0; sub < SUBS; ++sub) {
        if (sub != part % SUBS) continue;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              sC[(mi * 32 + crow32(r, hi)) * C_LD + wn * 64 + ni * 32 + l31] = acc[sub * 2 + mi][ni][r];
      }
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {                    // phase 2: compute + stores only
      const int rl = pass * RPP + rr, row = m0 + part * 64 + rl;
      if (row < p.M) {
        f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = w0;
        if constexpr (KS) {                                           // partial sums of this K range -> workspace; splitk_reduce_kernel adds them up in a fixed order
          float* dst = p.ks_ws + ((size_t)ks_id * p.M + row) * p.N + n0 + c8;
          *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(sC + rl * C_LD + c8);
          *reinterpret_cast<f32x4*>(dst + 4) = *reinterpret_cast<const f32x4*>(sC + rl * C_LD + c8 + 4);
          continue;
        }
        epilogue8<EPI>(p, row, n0 + c8, *reinterpret_cast<const f32x4*>(sC + rl * C_LD + c8),
                       *reinterpret_cast<const f32x4*>(sC + rl * C_LD + c8 + 4), bias0, bias1, aux[pass], w0, w1);
        if constexpr (EPI == EPI_DGELU) {
          csum0 += w0; csum1 += w1;
          if (p.q8_amax) {                                         // recording step of the fp8 backward (delayed scaling): max |du| of this launch -- launches below 8192 rows at K = 384 land on this
#pragma unroll                                                     // kernel, and without the post the du site kept its start-up scale (found by the 2-rank d = 384 fp8 test, round 6)
            for (int e = 0; e < 4; ++e) omax = fmaxf(omax, fmaxf(fabsf(w0[e]), fabsf(w1[e])));
          }
        }
      }
    }
    if (part + 1 < BMT / 64) __syncthreads();
  }
  if constexpr (EPI == EPI_DGELU) {
    if (p.q8_amax) amax_post(p.q8_amax, wave_max(omax), lane, blockIdx.x * (G::THREADS / 64) + (tid >> 6));
  }
  if constexpr (EPI == EPI_DGELU) {
    if (p.colsum) {                                               // block-reduce over the RPP row groups, one atomic per column
      __syncthreads();
      *reinterpret_cast<f32x4*>(sC + rr * BN + c8) = csum0;
      *reinterpret_cast<f32x4*>(sC + rr * BN + c8 + 4) = csum1;
      __syncthreads();
      if (tid < BN) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < RPP; ++g) t += sC[g * BN + tid];
        atomicAdd(p.colsum + n0 + tid, t);
      }
    }
  }
}

// Split-K reduction in a FIXED order, as its own launch (stream order makes the partial tiles visible; an in-kernel "last block reduces" needs an
// agent-scope release, i.e. an L2 write-back per block on this 8-XCD part: measured 68-87 us against 36-42 us for atomics).  Fixed order
// because the head Linears feed BatchNorm + ReLU gates: with fp32 atomics the order varies from run to run and flips gates of near-zero
// pre-activations.  C[row, col .. col+3] = bias + sum_s ws[s][row][col .. col+3].
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int ks, int M, int N, const float* __restrict__ bias,
                                                            float* __restrict__ C, int ldc) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, n4 = (size_t)(N / 4);
  if (i >= (size_t)M * n4) return;
  const int row = (int)(i / n4), col = (int)(i % n4) * 4;
  f32x4 acc = bias ? *reinterpret_cast<const f32x4*>(bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
  const float* src = ws + (size_t)row * N + col;
  for (int s = 0; s < ks; ++s) acc += *reinterpret_cast<const f32x4*>(src + (size_t)s * M * N);
  *reinterpret_cast<f32x4*>(C + (size_t)row * ldc + col) = acc;
}

// ---- helpers of the phased main loops (gemm_nt_row384_kernel<.., PH>, gemm_p8.h) ----
template <int N> DEVFN void p8_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
// LDS-DMA of 16 B per lane as inline assembly in the scalar-base form: address = SGPR pair + 32-bit lane offset.  Through the builtin hipcc turned the
// (wave-uniform) k-tile advance into per-lane 64-bit pointers that it then spilled -- and every spill reload is a `s_waitcnt vmcnt(0)`, which drains
// the five units the loop keeps in flight.  As with attention.hip's glds16_asm the compiler does not count these loads: the waits are the kernel's own.
// a pointer the compiler cannot prove wave-uniform (tile origins come out of an integer division done on the VALU) made visibly so: the "s" constraint of
// the asm below does not insert the v_readfirstlane itself -- it hands the assembler a VGPR pair
DEVFN const char* sgpr_ptr(const void* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi_ = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const char*>(((unsigned long long)hi_ << 32) | lo);
}
DEVFN void p8_glds16(unsigned voff, const void* sbase /* wave-uniform */, unsigned lds_dst /* wave-uniform LDS byte address */) {
  unsigned keep;                                     // M0 is compiler-reserved and not preserved around a statement: saved, set, restored inside it
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}


// ---- 384-column tiles: 8 waves (2 x 4) ---------------------------------------------------------------------------------
// Every N on the encoder path (384, 1152, 1536, 768...) is a multiple of 384.  MI = 32-row accumulator blocks per wave:
//   MI = 2: 128 x 384 tile, wave 64 x 96, 2-stage ring of 32 KB, two blocks per CU (M < 8192)
//   MI = 4: 256 x 384 tile, wave 128 x 96 = 12 accumulators, 3-stage ring of 40 KB, one block per CU.  Per 32-deep k-tile the
//           block stages 16 KB of A and 24 KB of B for six 128x128 units of output: 6.7 KB / unit (10.7 for MI = 2, 16 for the
//           square tile) -- operand delivery L2 -> LDS is what bounds these GEMMs.
// For N = 384 one block owns whole output rows, which is what the row-wise epilogues below need.
namespace row384 {
constexpr int BNR = 384, WAVES = 8, THREADS = 512, CLD = BNR + 4;
constexpr int CLD2 = 448, PLANE1 = 208;       // staging layout of the 8-column-slot epilogues (see the staging loop)
constexpr int TR_PITCH = 208;                  // transposed-accumulator epilogue: bytes per staged row of a wave's strip (96 bf16 + 16 B)
template <int MI> struct Geo {
  static constexpr int BMR = 64 * MI, ROWB = BK * 2, A_BYTES = BMR * ROWB, BB = BNR * ROWB, STAGE = A_BYTES + BB;   // 32 / 40 KB
  static constexpr int NSTG = MI == 4 ? 3 : 2;
  static constexpr int RPI = 1024 / ROWB;                                                        // rows per 1-KiB load instruction (16)
  static constexpr int A_IPW = (BMR / RPI) / WAVES, B_IPW = (BNR / RPI) / WAVES;                 // load instructions per wave: 1-2 + 3
};
// XOR key of a row's 16-B chunks: the 16 rows of a ds_read_b128 service group must land on 16 distinct bank slots
DEVFN int swz_key(int row) { return (row >> 2) & 3; }
// LDS layout of the epilogue (floats): staging tile | bias | gamma, beta | per-row scale | row-wise: per-row {mean, rstd}, input slots of NT streamed
// row tensors, [wave][pair][tensor][2 rows]
template <int MI, bool ROWWISE, int NT> struct EpiLds {
  static constexpr int SC = 32 * (ROWWISE ? CLD : CLD2), SBIAS = ROWWISE ? BNR : CLD2;
  static constexpr int BYTES = (SC + SBIAS + 2 * BNR + 64 * MI * (ROWWISE ? 3 : 1)) * 4 + (ROWWISE ? WAVES * 2 * NT * 2 * 384 * 4 : 0);
};
template <int MI, int EPI, bool LN> constexpr int lds_bytes() {
  constexpr bool rowwise = (LN && EPI == EPI_RESID) || EPI == EPI_LNBWD;
  constexpr int epi = EpiLds<MI, rowwise, EPI == EPI_LNBWD ? 2 : 1>::BYTES, ring = Geo<MI>::NSTG * Geo<MI>::STAGE;
  return epi > ring ? epi : ring;                                 // 64 / 120 KB ; row-wise epilogues 110 KB (residual + LN) / 158 KB (LN backward)
}
}

// Row-wise epilogues (N == 384: the block owns whole rows).  A wave works on TWO rows at a time, one per half-wave: lane
// (hh = lane >> 5, li = lane & 31) owns columns 128 j + 4 li .. + 3 (j = 0..2) of row hh of the pair, so every LDS / global
// access is 16 B per lane (fp32) or 8 B (bf16) -- the epilogue's time is set by the number of vector-memory instructions it
// issues, not only by its bytes -- and the row reductions are 5-step butterflies inside a half-wave.
//
// The fp32 row tensors such an epilogue READS (residual stream ; LayerNorm input and residual gradient) do not go through
// registers: with 192 accumulator registers live a wave cannot hold more than one part's rows, so every part exposed a full
// HBM round trip and the CU had <= 12 KB per wave in flight (measured: the fused LayerNorm-backward GEMM was no faster than
// GEMM + ln_bwd_kernel).  Instead each wave owns a private LDS slot per pair of rows and fills it by LDS-DMA (global_load_lds,
// 3 x 1 KiB per tensor: the two rows of a pair are adjacent in HBM); the slot is re-issued for the NEXT part as soon as the
// wave has copied this part's rows out of it, i.e. one part (compute, stores, barrier, staging) ahead of its use, with no
// registers held meanwhile.  The only wait is `s_waitcnt vmcnt(0)` in front of the staging barrier.
typedef const void __attribute__((address_space(1))) * gptr_t_;
typedef void __attribute__((address_space(3))) * lptr_t_;
constexpr int ROWIN_PAIR_BYTES = 2 * 384 * 4;                      // 3072 B: two adjacent fp32 rows
// 64 floats (one per lane, any per-lane source address) into 64 consecutive LDS floats, asynchronously.  The row-wise epilogues fill
// their small tables (bias, gamma, beta, per-row DropPath scale, LayerNorm statistics) this way: fetched into registers they cost
// two exposed round trips (load -> s_waitcnt -> ds_write, twice) at the head of every tile's epilogue, with no register free to
// start them earlier; as LDS-DMA they land under the staging of part 0, whose vmcnt(0) + barrier covers them.
DEVFN void lds_fill64(const float* src_lane, float* dst_wave) {
  __builtin_amdgcn_global_load_lds((gptr_t_)src_lane, (lptr_t_)dst_wave, 4, 0, 0);
}
template <int NT>
DEVFN void rowin_issue(const float* t0, const float* t1, int M, int row0, char* slot, int lane) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const float* base = t == 0 ? t0 : t1;
    if (!base) continue;                                           // optional tensor (residual gradient of the first LayerNorm backward)
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) {
      const int boff = pc * 1024 + lane * 16, second = boff >= 1536 ? 1 : 0;
      int row = row0 + second; row = row < M ? row : M - 1;        // rows >= M: clamped, never used
      const char* src = reinterpret_cast<const char*>(base + (size_t)row * 384) + (boff - second * 1536);
      __builtin_amdgcn_global_load_lds((gptr_t_)src, (lptr_t_)(slot + t * ROWIN_PAIR_BYTES + pc * 1024), 16, 0, 0);
    }
  }
}
// this lane's 12 values of tensor t of its row of the pair (lane = (hh, li): row hh, columns 128 j + 4 li .. + 3)
DEVFN void rowin_read(const char* slot, int t, int hh, int li, f32x4* v) {
#pragma unroll
  for (int j = 0; j < 3; ++j) v[j] = *reinterpret_cast<const f32x4*>(slot + t * ROWIN_PAIR_BYTES + hh * 1536 + j * 512 + li * 16);
}
// Waiting for a slot.  vmcnt retires in order, so "the refill of pair q has landed" = "at most the VMEM operations issued after
// it are outstanding".  Between the refill of a pair and its use one part later a wave issues, in program order, the stores of
// the pair, then the refill and the stores of the other pair (row-wise LayerNorm: refill, stores, refill, stores) -- NVM
// operations when every row of the tile is live and every optional output exists (`exact`); the count is taken two short, and
// anything the compiler adds (scratch traffic) only makes the wait stricter.  Otherwise: vmcnt(0).
template <int NVM> DEVFN void rowin_wait(bool exact) {
  if (exact) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NVM) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// A row-wise epilogue is bound by the NUMBER of vector-memory instructions it issues (tools/trace_rowwise.py: a pair of rows
// took ~4 k cycles for 8 stores + 3 LDS-DMA pieces per wave, ~50 cycles per store instruction per CU whatever its width), so
// the bf16 row goes out in two instructions instead of three: neighbouring lanes swap 4-column groups (one DPP move per
// dword) so that even lanes hold 8 consecutive columns of the row's first 128 and odd lanes 8 of its second 128 -- one
// 16-B store for both -- and the last 128 columns follow as one 8-B store.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
DEVFN u32x2 pack_bf16x4(const float* v) {
  bf16x4 o; o[0] = f2bf(v[0]); o[1] = f2bf(v[1]); o[2] = f2bf(v[2]); o[3] = f2bf(v[3]);
  return __builtin_bit_cast(u32x2, o);
}
DEVFN void st_bf16_row(bf16* rowp, const float* v /* [12]: columns 128 j + 4 li + e */, int li) {
  const u32x2 c0 = pack_bf16x4(v), c1 = pack_bf16x4(v + 4), c2 = pack_bf16x4(v + 8);
  const bool odd = li & 1;
  const u32x2 send = odd ? c0 : c1;
  u32x2 recv;                                                      // quad_perm [1,0,3,2]: the neighbouring lane's dword
  recv[0] = __builtin_amdgcn_update_dpp(0u, send[0], 0xB1, 0xF, 0xF, true);
  recv[1] = __builtin_amdgcn_update_dpp(0u, send[1], 0xB1, 0xF, 0xF, true);
  const u32x4 wide = odd ? u32x4{recv[0], recv[1], c1[0], c1[1]} : u32x4{c0[0], c0[1], recv[0], recv[1]};
  st_pol<4>(wide, reinterpret_cast<u32x4*>(rowp + (odd ? 128 + 4 * (li - 1) : 4 * li)));
  st_pol<4>(c2, reinterpret_cast<u32x2*>(rowp + 256 + 4 * li));
}
// e4m3 copy of a row-wise epilogue's bf16 row (round 6: the all-e4m3 step at d = 384 keeps its LayerNorms inside the GEMM epilogues).  384 B per
// row = 12 B per lane: a pair of lanes swaps one dword, so that even lanes hold 8 consecutive columns of the first 128 and odd lanes 8 of the second
// -- one 8-B store for both -- and the last 128 columns follow as one 4-B store (two instructions, like the bf16 row).
struct RowQ8 { float s, amax; unsigned nclip; };                  // scale of the copy ; running max |value| (the site's next scale) ; elements clipped at +-448
DEVFN unsigned pack_e4m3x4(const float* v, float s) {
  int q = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v[0] * s, -448.f, 448.f), __builtin_amdgcn_fmed3f(v[1] * s, -448.f, 448.f), 0, false);
  q = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v[2] * s, -448.f, 448.f), __builtin_amdgcn_fmed3f(v[3] * s, -448.f, 448.f), q, true);
  return (unsigned)q;
}
DEVFN void st_e4m3_row(uint8_t* rowp, const float* v /* [12]: bf16-rounded values of columns 128 j + 4 li + e */, float s, int li, bool live) {
  const unsigned c0 = pack_e4m3x4(v, s), c1 = pack_e4m3x4(v + 4, s), c2 = pack_e4m3x4(v + 8, s);
  const bool odd = li & 1;
  const unsigned recv = __builtin_amdgcn_update_dpp(0u, odd ? c0 : c1, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]: every lane takes part
  if (!live) return;
  const u32x2 wide = odd ? u32x2{recv, c1} : u32x2{c0, recv};
  st_pol<4>(wide, reinterpret_cast<u32x2*>(rowp + (odd ? 128 + 4 * (li - 1) : 4 * li)));
  st_pol<4>(c2, reinterpret_cast<unsigned*>(rowp + 256 + 4 * li));
}
// the two row stores from already-packed groups (lnb_row<Q8>: a group is packed as soon as its four values exist, so that 9 dwords instead of 12 floats live across the loop)
DEVFN void st_bf16_row_packed(bf16* rowp, u32x2 c0, u32x2 c1, u32x2 c2, int li, bool live) {
  const bool odd = li & 1;
  const u32x2 send = odd ? c0 : c1;
  u32x2 recv;
  recv[0] = __builtin_amdgcn_update_dpp(0u, send[0], 0xB1, 0xF, 0xF, true);
  recv[1] = __builtin_amdgcn_update_dpp(0u, send[1], 0xB1, 0xF, 0xF, true);
  if (!live) return;
  const u32x4 wide = odd ? u32x4{recv[0], recv[1], c1[0], c1[1]} : u32x4{c0[0], c0[1], recv[0], recv[1]};
  st_pol<4>(wide, reinterpret_cast<u32x4*>(rowp + (odd ? 128 + 4 * (li - 1) : 4 * li)));
  st_pol<4>(c2, reinterpret_cast<u32x2*>(rowp + 256 + 4 * li));
}
DEVFN void st_e4m3_row_packed(uint8_t* rowp, unsigned c0, unsigned c1, unsigned c2, int li, bool live) {
  const bool odd = li & 1;
  const unsigned recv = __builtin_amdgcn_update_dpp(0u, odd ? c0 : c1, 0xB1, 0xF, 0xF, true);
  if (!live) return;
  const u32x2 wide = odd ? u32x2{recv, c1} : u32x2{c0, recv};
  st_pol<4>(wide, reinterpret_cast<u32x2*>(rowp + (odd ? 128 + 4 * (li - 1) : 4 * li)));
  st_pol<4>(c2, reinterpret_cast<unsigned*>(rowp + 256 + 4 * li));
}
// residual + LayerNorm forward of the new row: x_new = resid + s (acc + bias) -> fp32 stream ; h = LN(x_new) -> bf16 operand
// of the next GEMM ; row statistics saved for the LayerNorm backward.  ref: Block.forward, audiossl/modules/transformer.py:136-150.
template <bool Q8>
DEVFN void lnf_row(const GemmArgs& p, int row, const float* sRow, const float* sBias, const float* sGamma, const float* sBeta,
                   float sc, const f32x4* rv, int li, float* sStatRow /* LDS: {mean, rstd} of this row, written out once per tile */, RowQ8& q8) {
  float v[12];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int col = j * 128 + li * 4;
    const f32x4 a4 = *reinterpret_cast<const f32x4*>(sRow + col), b4 = *reinterpret_cast<const f32x4*>(sBias + col);
    const f32x4 o = rv[j] + sc * (a4 + b4);
    st_pol<2>(o, reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (size_t)row * 384 + col));
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[4 * j + e] = o[e]; sum += o[e]; }
  }
  const float mu = half_sum(sum) * (1.0f / 384.0f);
  float qd = 0.f;
#pragma unroll
  for (int k = 0; k < 12; ++k) { const float d = v[k] - mu; qd += d * d; }
  const float rs = rsqrtf(half_sum(qd) * (1.0f / 384.0f) + 1e-6f);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int col = j * 128 + li * 4;
    const f32x4 g4 = *reinterpret_cast<const f32x4*>(sGamma + col), e4 = *reinterpret_cast<const f32x4*>(sBeta + col);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[4 * j + e] = (v[4 * j + e] - mu) * rs * g4[e] + e4[e];
  }
  if constexpr (Q8) {
    if (p.q8) {                                                    // e4m3 copy of the SAME bf16 values (A operand of the next e4m3 GEMM), as ln_fwd_kernel writes it
      float cm = 0.f;
#pragma unroll
      for (int k = 0; k < 12; ++k) { v[k] = bf2f(f2bf(v[k])); cm = fmaxf(cm, fabsf(v[k])); }
      st_e4m3_row(p.q8 + (size_t)row * 384, v, q8.s, li, true);
      q8.amax = fmaxf(q8.amax, cm);
      asm volatile("" : "+v"(q8.amax));                            // pinned: see lnb_fold4
      if (cm * q8.s > 448.f) {                                     // rare: count exactly
#pragma unroll
        for (int k = 0; k < 12; ++k) q8.nclip += fabsf(v[k] * q8.s) > 448.f ? 1u : 0u;
      }
    }
    if (p.ln_out) st_bf16_row(p.ln_out + (size_t)row * 384, v, li);   // (not written when every reader takes the e4m3 copy: fp8_lean)
  } else {
    st_bf16_row(p.ln_out + (size_t)row * 384, v, li);
  }
  if (li == 0) { sStatRow[0] = mu; sStatRow[1] = rs; }
}
// LayerNorm backward of one row, fused into the N == 384 dgrad epilogues (EPI_LNBWD).  dy comes from the staged fp32
// accumulators (never rounded to bf16, never written to HBM), x / dres were loaded before the staging barrier.  Math as in
// ln_bwd_kernel (layernorm.hip):
//   dx = dres + rstd (dy g - mean(dy g) - xhat mean(dy g xhat)) ; g_out = bf16(row_scale dx) ; column sums for dgamma,
//   dbeta and the upstream bias gradient stay in registers until the block has finished its tile.
// ref: the autograd of nn.LayerNorm(eps=1e-6) + the residual add + DropPath in Block.forward, audiossl/modules/transformer.py:136-150.
// Column sums: the two half-waves work on different rows of the SAME columns; the contributions of a pair of rows are
// folded with v_permlane32_swap as soon as they exist, so that the lower half-wave keeps the sums of columns e = 0, 1 of each
// of its 4-column groups and the upper half those of e = 2, 3 (18 registers per lane instead of 36).
struct LnbCols { float dg[6], db[6], du[6]; };
DEVFN void lnb_fold4(float* acc2, const float* d4) {               // acc2[e] += d4[e] + partner's d4[e]  (lower: e = 0, 1 ; upper: e = 2, 3)
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const float fa = d4[e], fb = d4[e + 2];
    auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, fa), __builtin_bit_cast(unsigned, fb), false, false);
    const unsigned r0 = r[0], r1 = r[1];                           // lower half: d[e] of both rows ; upper half: d[e + 2] of both rows
    acc2[e] += __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
    // pin the update here: the sums are only read after the last part, and without this LLVM sinks the whole chain of adds and
    // swaps down to that point -- keeping every contribution of every part alive (measured: 190 spilled registers)
    asm volatile("" : "+v"(acc2[e]));
  }
}
// slot: this pair's input slot (tensor 0 = residual gradient, tensor 1 = LayerNorm input x); everything is read from LDS in
// both passes (the staged dy row, gamma, x, dres), so that nothing but the two row sums and the column sums lives across the
// reduction -- with 168+ accumulator registers still live in the first parts, every value kept in a register there is a
// spill (measured with x and dres copied to registers so that the slot could be refilled before the arithmetic: 116 B of
// scratch per lane and 375 / 306 us instead of 307 / 237 us for the fc1 / qkv dgrad GEMMs at M = 131072).
template <bool Q8>
DEVFN void lnb_row(const GemmArgs& p, int row, bool live, const float* sRow, const float* sGamma, float sc, float mu, float rs,
                   const char* slot, bool has_res, int hh, int li, LnbCols& cs, RowQ8& q8) {
  float s1 = 0.f, s2 = 0.f;                                        // live == false (row beyond M): every contribution is zero
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int col = j * 128 + li * 4;
    const f32x4 d4 = live ? *reinterpret_cast<const f32x4*>(sRow + col) : f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 g4 = *reinterpret_cast<const f32x4*>(sGamma + col);
    const f32x4 x4 = *reinterpret_cast<const f32x4*>(slot + ROWIN_PAIR_BYTES + hh * 1536 + j * 512 + li * 16);
    float ddg[4], ddb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = live ? (x4[e] - mu) * rs : 0.f, dyg = d4[e] * g4[e];
      ddg[e] = d4[e] * xh; ddb[e] = d4[e];
      s1 += dyg; s2 += dyg * xh;
    }
    lnb_fold4(cs.dg + 2 * j, ddg); lnb_fold4(cs.db + 2 * j, ddb);  // both half-waves get here: no early exit for dead rows
  }
  const float c1 = half_sum(s1) * (1.0f / 384.0f), c2 = half_sum(s2) * (1.0f / 384.0f);
  if constexpr (Q8) {                                              // the same arithmetic; every 4-column group is packed (bf16 and e4m3) as soon as it exists
    u32x2 pb[3]; unsigned p8[3]; float cm = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int col = j * 128 + li * 4;
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(sRow + col), g4 = *reinterpret_cast<const f32x4*>(sGamma + col);
      const f32x4 x4 = *reinterpret_cast<const f32x4*>(slot + ROWIN_PAIR_BYTES + hh * 1536 + j * 512 + li * 16);
      f32x4 r4 = {0.f, 0.f, 0.f, 0.f};
      if (has_res) r4 = *reinterpret_cast<const f32x4*>(slot + hh * 1536 + j * 512 + li * 16);
      f32x4 o; float g4s[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (x4[e] - mu) * rs;
        o[e] = live ? r4[e] + rs * (d4[e] * g4[e] - c1 - xh * c2) : 0.f;
        g4s[e] = o[e] * sc;
        cm = fmaxf(cm, fabsf(g4s[e]));
      }
      if (live) st_pol<2>(o, reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (size_t)row * 384 + col));
      lnb_fold4(cs.du + 2 * j, g4s);
      pb[j] = pack_bf16x4(g4s);
#pragma unroll
      for (int e = 0; e < 4; ++e) g4s[e] = bf2f(f2bf(g4s[e]));
      p8[j] = pack_e4m3x4(g4s, q8.s);                             // e4m3 copy of the bf16 values (A operand of the e4m3 proj / fc2 dgrad and weight-gradient GEMMs)
    }
    if (p.q8_amax) {                                               // max |g| of the site (dead rows: zeros) -- the next step's scale
      q8.amax = fmaxf(q8.amax, cm);
      asm volatile("" : "+v"(q8.amax));                            // pinned: see lnb_fold4
    }
    if (p.lnb_g) st_bf16_row_packed(p.lnb_g + (size_t)row * 384, pb[0], pb[1], pb[2], li, live);   // every lane takes part in the DPP exchange; dead rows do not store
    if (p.q8) st_e4m3_row_packed(p.q8 + (size_t)row * 384, p8[0], p8[1], p8[2], li, live);
    return;
  }
  float gs[12];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int col = j * 128 + li * 4;
    const f32x4 d4 = *reinterpret_cast<const f32x4*>(sRow + col), g4 = *reinterpret_cast<const f32x4*>(sGamma + col);
    const f32x4 x4 = *reinterpret_cast<const f32x4*>(slot + ROWIN_PAIR_BYTES + hh * 1536 + j * 512 + li * 16);
    f32x4 r4 = {0.f, 0.f, 0.f, 0.f};
    if (has_res) r4 = *reinterpret_cast<const f32x4*>(slot + hh * 1536 + j * 512 + li * 16);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (x4[e] - mu) * rs;
      o[e] = live ? r4[e] + rs * (d4[e] * g4[e] - c1 - xh * c2) : 0.f;
      gs[4 * j + e] = o[e] * sc;
    }
    if (live) st_pol<2>(o, reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (size_t)row * 384 + col));
    lnb_fold4(cs.du + 2 * j, gs + 4 * j);
  }
  if (p.lnb_g) {                                                   // every lane takes part in the DPP exchange; dead rows do not store
    if (live) st_bf16_row(p.lnb_g + (size_t)row * 384, gs, li);
  }
}
// Block reduction of the column sums of lnb_row over the NW waves of a block (red: >= 3 * NW * 384 floats of LDS that no
// wave reads any more), one atomic per column per block.
template <int NW>
DEVFN void lnb_flush(const GemmArgs& p, const LnbCols& cs, float* red, int tid) {
  const int li = tid & 31, hh = (tid >> 5) & 1, wid = tid >> 6;
  lds_barrier();
#pragma unroll
  for (int t = 0; t < 6; ++t) {
    const int col = (t >> 1) * 128 + li * 4 + (t & 1) + 2 * hh;
    red[(0 * NW + wid) * 384 + col] = cs.dg[t]; red[(1 * NW + wid) * 384 + col] = cs.db[t]; red[(2 * NW + wid) * 384 + col] = cs.du[t];
  }
  lds_barrier();
  for (int c = tid; c < 384; c += NW * 64) {
    float a = 0.f, b = 0.f, u = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { a += red[(0 * NW + w) * 384 + c]; b += red[(1 * NW + w) * 384 + c]; u += red[(2 * NW + w) * 384 + c]; }
    atomicAdd(p.lnb_dgamma + c, a);
    atomicAdd(p.lnb_dbeta + c, b);
    if (p.lnb_dbias_up) atomicAdd(p.lnb_dbias_up + c, u);
  }
}

// F8: the operands are OCP e4m3 bytes.  The host passes them as pairs ("bf16" elements: K, lda, ldb halved), so the whole
// LDS-DMA ring -- 64-byte rows, XOR swizzle, stage geometry -- is byte-identical; a k-tile then covers K = 64 and feeds ONE
// v_mfma_scale_f32_32x32x64_f8f6f4 per accumulator (unit block scales; twice the bf16 MFMA rate) instead of two
// 32x32x16 bf16 MFMAs: half the matrix-pipe time AND half the operand bytes per FLOP.  p.dq undoes the per-tensor scales.
// LN (EPI_RESID only): the epilogue also produces the LayerNorm of the new residual row (the next sub-layer's pre-LN).
// TR (EPI_BF16 only): the MFMA operands are swapped, so an accumulator block holds C^T: a lane owns ONE row of the block and four consecutive
// columns per register quad.  The epilogue then needs no fp32 staging and no block barrier: a quad is packed to 8 B of bf16, a wave writes its
// 32 x 96 block row to a PRIVATE staging strip (12 ds_write_b64 instead of 96 ds_write_b32), reads it back as 16-B pieces of 192-B row segments and
// stores them; the eight waves drift apart and overlap each other's LDS and store phases.
#ifndef ATST_MI2_WPS           // waves per SIMD the 128-row instantiation is compiled for: 4 = two blocks per CU (128 registers), 2 = one (256, no spills)
#define ATST_MI2_WPS 4
#endif
// PH (round 5, MI = 4, bf16 operands): the PHASED main loop -- the structure of gemm_p8.h inside this kernel's tile, so that every epilogue below
// (in particular the row-wise LayerNorm ones, which need the 384-column tile) is reused unchanged.  A 32-deep k-tile is four phases (ks, mh): 6 MFMAs
// of the wave's row blocks 2 mh, 2 mh + 1 against its three column blocks at k-step ks; fragments in two register sets per operand (A by mh, B by ks:
// 40 registers beside the 192 accumulators); the two wave rows run one barrier apart, so that each SIMD's two waves alternate between their MFMA
// cluster and their fragment reads / LDS-DMA; k-tile t + 2 is staged during phases 1-3 of k-tile t (inline-asm LDS-DMA, scalar base + lane offset)
// and the only vector-memory wait is a counted `vmcnt(5)` in the last phase of a k-tile.
// PH == 2 (round 6): the same phased loop on 64-DEEP k-tiles of whole 128-B rows (two 80-KB slots = all of the LDS): the two-team experiment measured the
// same bytes streaming at 44 B/clk/CU as whole lines and at 28 as 64-B rows (DESIGN.md section 3 "Round 6").  Eight phases (ks, mh) per k-tile, k-tile t + 1 staged in
// phases 1 .. 5 of k-tile t (10 pieces per wave), `vmcnt(0)` in the last phase.  Chunk c of row r at c ^ ((r >> 1) & 7) as in gemm_p8.h.
// Q8: the row-wise epilogues also write the e4m3 copy of their bf16 row (and its amax); a compile-time switch so that the bf16 step's instantiations keep
// their register allocation (with the copy as a run-time branch the LayerNorm-backward epilogue spilled 19-25 registers where it had spilled none).
template <int EPI, int MI, bool LN = false, bool F8 = false, bool TR = false, int PH = 0, bool Q8 = false>
__global__ __launch_bounds__(512, (MI == 2 ? ATST_MI2_WPS : 2)) void gemm_nt_row384_kernel(GemmArgs p) {
  static_assert(!TR || (EPI == EPI_BF16 && !LN), "transposed accumulators: store-only bf16 epilogue");
  static_assert(!PH || (MI == 4 && !F8 && !TR), "phased main loop: 256-row tile, bf16 operands");
  using namespace row384;
  using RG = row384::Geo<MI>;
  constexpr int BMR = RG::BMR, A_BYTES = RG::A_BYTES, STAGE = RG::STAGE, NSTG = RG::NSTG, ROWB = RG::ROWB;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  typedef const void __attribute__((address_space(1))) * gptr_t;
  typedef void __attribute__((address_space(3))) * lptr_t;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 2, wn = wid & 3, hi = lane >> 5, l31 = lane & 31;
  const int ntn = p.N / BNR;
  const int ntm = (p.M + BMR - 1) / BMR;
  const int id = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (id / ntn) * BMR, n0 = (id % ntn) * BNR;

  // Operand tiles go HBM/L2 -> LDS directly (global_load_lds, 16 B per lane, 1 KiB = 16 rows per wave-instruction, no
  // staging registers), NSTG-1 k-tiles ahead of the MFMAs.  LDS rows are 64 B; chunk c of row r is stored at chunk
  // c ^ swz_key(r); the permutation is applied to the per-lane *source* address (the LDS destination is lane-linear).
  char* lds = smem_raw;
  const int lrow = lane >> 2, lchunk = lane & 3;
  const bf16* srcA[RG::A_IPW]; const bf16* srcB[RG::B_IPW];
#pragma unroll
  for (int j = 0; j < RG::A_IPW; ++j) {
    const int row = (wid * RG::A_IPW + j) * RG::RPI + lrow;
    int ra = m0 + row; ra = ra < p.M ? ra : p.M - 1;              // clamp: rows >= M are never stored
    srcA[j] = p.A + (size_t)ra * p.lda + (lchunk ^ swz_key(row)) * 8;
  }
#pragma unroll
  for (int j = 0; j < RG::B_IPW; ++j) {
    const int row = (wid * RG::B_IPW + j) * RG::RPI + lrow;
    srcB[j] = p.B + (size_t)(n0 + row) * p.ldb + (lchunk ^ swz_key(row)) * 8;
  }
  auto issue = [&](int kt) {
    char* st = lds + (kt % NSTG) * STAGE;
#pragma unroll
    for (int j = 0; j < RG::A_IPW; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(srcA[j] + kt * BK), (lptr_t)(st + (wid * RG::A_IPW + j) * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < RG::B_IPW; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(srcB[j] + kt * BK), (lptr_t)(st + A_BYTES + (wid * RG::B_IPW + j) * 1024), 16, 0, 0);
  };
  f32x16 acc[MI][3];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float* sBiasT = reinterpret_cast<float*>(smem_raw + NSTG * STAGE);   // TR: bias of the block's 384 columns, behind the ring
  if constexpr (TR) {                                              // the oldest load of every wave: landed (and barrier-published) long before the epilogue
    if (wid < BNR / 64) {
      if (p.bias) lds_fill64(p.bias + n0 + wid * 64 + lane, sBiasT + wid * 64); else sBiasT[wid * 64 + lane] = 0.f;
    }
  }

  if constexpr (PH == 2) {
    constexpr int BKL = 64, ROWL = 128, A_BY = BMR * ROWL, STG = A_BY + BNR * ROWL;   // 32 KB + 48 KB
    const int nk = p.K / BKL;                                      // >= 2 (launcher)
    const int wu = __builtin_amdgcn_readfirstlane(wid);
    // LDS-DMA: a piece = 8 rows x 128 B; wave w stages A pieces 4 w .. 4 w + 3 and B pieces 6 w .. 6 w + 5; lane (r8, c) fetches chunk c ^ key(row)
    const int r8 = lane >> 3, c8 = lane & 7;
    unsigned voA[4], voB[6];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (wu * 4 + j) * 8 + r8;
      int ra = m0 + row; ra = ra < p.M ? ra : p.M - 1;              // clamp: rows >= M are never stored
      voA[j] = (unsigned)((ra - m0) * p.lda + (c8 ^ ((row >> 1) & 7)) * 8) * 2u;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int row = (wu * 6 + j) * 8 + r8;
      voB[j] = (unsigned)(row * p.ldb + (c8 ^ ((row >> 1) & 7)) * 8) * 2u;
    }
    const char* baseA = sgpr_ptr(p.A + (size_t)m0 * p.lda);
    const char* baseB = sgpr_ptr(p.B + (size_t)n0 * p.ldb);
    const unsigned ldsA = lds_addr(smem_raw) + wu * 4 * 1024, ldsB = lds_addr(smem_raw) + A_BY + wu * 6 * 1024;
    auto stage_piece = [&](int kt, int slot, int j) {               // j: 0 .. 3 = A pieces ; 4 .. 9 = B pieces
      const size_t kofs = (size_t)kt * (BKL * 2);
      if (j < 4) p8_glds16(voA[j < 4 ? j : 0], baseA + kofs, ldsA + slot * STG + j * 1024);
      else p8_glds16(voB[j >= 4 ? j - 4 : 0], baseB + kofs, ldsB + slot * STG + (j - 4) * 1024);
    };
    const int xk = (l31 >> 1) & 7;
    const int fA0 = (wm * 128 + l31) * ROWL + ((hi ^ xk) << 4), fB0 = A_BY + (wn * 96 + l31) * ROWL + ((hi ^ xk) << 4);
    bf16x8 fa[2][2], fb[2][3];                                      // A: [mh][rb] ; B: [ks & 1][ni]
    auto read_a = [&](int slot, int ks, int mh) {
      int b0 = fA0; asm volatile("" : "+v"(b0));
      const char* s = lds + slot * STG + mh * 64 * ROWL;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) fa[mh][rb] = *reinterpret_cast<const bf16x8*>(s + (b0 ^ (ks << 5)) + rb * 32 * ROWL);
    };
    auto read_b = [&](int slot, int ks) {
      int b0 = fB0; asm volatile("" : "+v"(b0));
      const char* s = lds + slot * STG;
#pragma unroll
      for (int ni = 0; ni < 3; ++ni) fb[ks & 1][ni] = *reinterpret_cast<const bf16x8*>(s + (b0 ^ (ks << 5)) + ni * 32 * ROWL);
    };
#pragma unroll
    for (int j = 0; j < 10; ++j) stage_piece(0, 0, j);
    p8_wait_vm<0>();
    asm volatile("s_barrier" ::: "memory");
    if ((wu >> 2) == 1) asm volatile("s_barrier" ::: "memory");     // the upper wave row runs one barrier behind the lower one
    for (int t = 0; t < nk; ++t) {
      const int slot = t & 1, slot1 = slot ^ 1;
      const bool more = t + 1 < nk;
      auto phase = [&](auto kstag, auto mhtag) {
        constexpr int ks = decltype(kstag)::value, mh = decltype(mhtag)::value, ph = 2 * ks + mh;
        if constexpr (mh == 0) read_b(slot, ks);
        read_a(slot, ks, mh);
        if (more) {                                                 // k-tile t + 1 into the slot k-tile t - 1 was read from (its last reads: phase 7, both wave rows past them from phase 1 on)
          if constexpr (ph >= 1 && ph <= 5) { stage_piece(t + 1, slot1, 2 * (ph - 1)); stage_piece(t + 1, slot1, 2 * (ph - 1) + 1); }
          if constexpr (ph == 7) p8_wait_vm<0>();                   // k-tile t + 1 has landed -- mine; the barrier makes it everyone's
        }
        asm volatile("s_barrier" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int ni = 0; ni < 3; ++ni) acc[2 * mh + rb][ni] = mfma32(fa[mh][rb], fb[ks & 1][ni], acc[2 * mh + rb][ni]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
      };
      static_for<8>([&](auto phtag) { constexpr int q = decltype(phtag)::value; phase(std::integral_constant<int, (q >> 1)>{}, std::integral_constant<int, (q & 1)>{}); });
    }
    if ((wu >> 2) == 0) asm volatile("s_barrier" ::: "memory");     // re-align the wave rows: every operand read is done, the ring is free
  } else
  if constexpr (PH) {
    const int nk = p.K / BK;                                       // >= 2 (checked by the launcher)
    const int wu = __builtin_amdgcn_readfirstlane(wid);
    // LDS-DMA sources: scalar base (tile origin + k-tile) + 32-bit lane offsets; wave w stages pieces 2 w, 2 w + 1 of A and 3 w .. 3 w + 2 of B
    unsigned voA[2], voB[3];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = (wu * 2 + j) * 16 + lrow;
      int ra = m0 + row; ra = ra < p.M ? ra : p.M - 1;              // clamp: rows >= M are never stored
      voA[j] = (unsigned)((ra - m0) * p.lda + (lchunk ^ swz_key(row)) * 8) * 2u;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int row = (wu * 3 + j) * 16 + lrow;
      voB[j] = (unsigned)(row * p.ldb + (lchunk ^ swz_key(row)) * 8) * 2u;
    }
    const char* baseA = sgpr_ptr(p.A + (size_t)m0 * p.lda);
    const char* baseB = sgpr_ptr(p.B + (size_t)n0 * p.ldb);
    const unsigned ldsA = lds_addr(smem_raw) + wu * 2 * 1024, ldsB = lds_addr(smem_raw) + A_BYTES + wu * 3 * 1024;
    auto stage_piece = [&](int kt, int slot, int j) {               // j: 0, 1 = A pieces ; 2 .. 4 = B pieces
      const size_t kofs = (size_t)kt * (BK * 2);
      if (j < 2) p8_glds16(voA[j < 2 ? j : 0], baseA + kofs, ldsA + slot * STAGE + j * 1024);
      else p8_glds16(voB[j >= 2 ? j - 2 : 0], baseB + kofs, ldsB + slot * STAGE + (j - 2) * 1024);
    };
    // fragment addresses: (row base | c0) ^ (ks << 5), c0 = (hi ^ key) << 4 ; re-derived from a laundered copy at every read (see gemm_p8.h)
    const int xk = swz_key(l31);
    const int fA0 = (wm * 128 + l31) * ROWB + ((hi ^ xk) << 4), fB0 = A_BYTES + (wn * 96 + l31) * ROWB + ((hi ^ xk) << 4);
    bf16x8 fa[2][2], fb[2][3];                                      // A: [mh][rb] ; B: [ks][ni]
    auto read_a = [&](int slot, int ks, int mh) {
      int b0 = fA0; asm volatile("" : "+v"(b0));
      const char* s = lds + slot * STAGE + mh * 64 * ROWB;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) fa[mh][rb] = *reinterpret_cast<const bf16x8*>(s + (b0 ^ (ks << 5)) + rb * 32 * ROWB);
    };
    auto read_b = [&](int slot, int ks) {
      int b0 = fB0; asm volatile("" : "+v"(b0));
      const char* s = lds + slot * STAGE;
#pragma unroll
      for (int ni = 0; ni < 3; ++ni) fb[ks][ni] = *reinterpret_cast<const bf16x8*>(s + (b0 ^ (ks << 5)) + ni * 32 * ROWB);
    };
    // prologue: k-tiles 0 and 1 ; wait for k-tile 0 ; the upper wave row runs one barrier behind the lower one
#pragma unroll
    for (int j = 0; j < 5; ++j) stage_piece(0, 0, j);
#pragma unroll
    for (int j = 0; j < 5; ++j) stage_piece(1, 1, j);
    p8_wait_vm<5>();
    asm volatile("s_barrier" ::: "memory");
    if ((wu >> 2) == 1) asm volatile("s_barrier" ::: "memory");
    int slot = 0;                                                   // ring slot of k-tile t (t % 3), slot2 = slot of k-tile t + 2
    for (int t = 0; t < nk; ++t) {
      const int slot2 = slot == 0 ? 2 : slot - 1;
      const bool more = t + 2 < nk;
      auto phase = [&](auto kstag, auto mhtag) {
        constexpr int ks = decltype(kstag)::value, mh = decltype(mhtag)::value, ph = 2 * ks + mh;
        if constexpr (mh == 0) read_b(slot, ks);
        read_a(slot, ks, mh);
        if (more) {                                                 // k-tile t + 2 into the slot k-tile t - 1 was read from (last read two phases ago or more)
          if constexpr (ph == 1) { stage_piece(t + 2, slot2, 0); stage_piece(t + 2, slot2, 2); }
          if constexpr (ph == 2) { stage_piece(t + 2, slot2, 1); stage_piece(t + 2, slot2, 3); }
          if constexpr (ph == 3) stage_piece(t + 2, slot2, 4);
        }
        if constexpr (ph == 3) { if (more) p8_wait_vm<5>(); else p8_wait_vm<0>(); }   // k-tile t + 1 has landed -- mine; read in the next phase
        asm volatile("s_barrier" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int ni = 0; ni < 3; ++ni) acc[2 * mh + rb][ni] = mfma32(fa[mh][rb], fb[ks][ni], acc[2 * mh + rb][ni]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
      };
      phase(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      phase(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
      phase(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
      phase(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
      slot = slot == 2 ? 0 : slot + 1;
    }
    if ((wu >> 2) == 0) asm volatile("s_barrier" ::: "memory");     // re-align the wave rows: every operand read is done, the ring is free
  } else {
  const int nk = p.K / BK;
  const int xr = swz_key(l31);                                     // every fragment row is l31 plus a multiple of 32
  const int offA = (wm * 32 * MI + l31) * ROWB, offB = A_BYTES + (wn * 96 + l31) * ROWB;
  constexpr int LOADS_PER_TILE = RG::A_IPW + RG::B_IPW;           // per wave, in issue order
  constexpr bool ILV = MI == 4;                                   // the 128-row tile has no registers to spare for the pinned order; a 2-stage ring needs its loads early
  auto issue_one = [&](int kt, int j) {                           // j-th load instruction of tile kt
    char* st = lds + (kt % NSTG) * STAGE;
    if (j < RG::A_IPW)
      __builtin_amdgcn_global_load_lds((gptr_t)(srcA[j < RG::A_IPW ? j : 0] + kt * BK), (lptr_t)(st + (wid * RG::A_IPW + j) * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((gptr_t)(srcB[j >= RG::A_IPW ? j - RG::A_IPW : 0] + kt * BK), (lptr_t)(st + A_BYTES + (wid * RG::B_IPW + j - RG::A_IPW) * 1024), 16, 0, 0);
  };
  // One k-tile of MFMAs.  ISSUE: the loads of tile kt + NSTG - 1 are spread BETWEEN the MFMA groups instead of in front
  // of them: every wave leaves the barrier at the same moment, and eight waves x 4-5 LDS-DMA instructions queue on the
  // CU's one address path for longer than the tile's MFMAs take -- issued up front, each (in-order) wave reaches its
  // MFMAs only after that queue drains and the matrix pipes idle (measured: loads-only 104 us + MFMA-only 91 us gave
  // 162 us); behind an already-issued MFMA group the same wait is hidden.
  auto tile = [&](int kt, auto issue_tag) {
    constexpr bool ISSUE = decltype(issue_tag)::value;
    const char* st = lds + (kt % NSTG) * STAGE;
    int slot = 0;
    if constexpr (F8) {
      // lane (row l31, half hi) owns bytes [32 hi, 32 hi + 32) of its 64-byte row: chunks 2 hi and 2 hi + 1 (swizzled); A and B
      // use the same (half, byte) -> k map, which is all the contraction needs
      typedef int v4i_ __attribute__((ext_vector_type(4)));
      typedef int v8i_ __attribute__((ext_vector_type(8)));
      const int c0 = ((2 * hi) ^ xr) << 4, c1 = ((2 * hi + 1) ^ xr) << 4;
      auto frag = [&](const char* rowp) {
        const v4i_ lo = *reinterpret_cast<const v4i_*>(rowp + c0), hi4 = *reinterpret_cast<const v4i_*>(rowp + c1);
        return v8i_{lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
      };
      v8i_ b8[3];
#pragma unroll
      for (int ni = 0; ni < 3; ++ni) b8[ni] = frag(st + offB + ni * 32 * ROWB);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const v8i_ a8 = frag(st + offA + mi * 32 * ROWB);
#pragma unroll
        for (int ni = 0; ni < 3; ++ni)
          acc[mi][ni] = TR ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8[ni], a8, acc[mi][ni], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F)
                           : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8[ni], acc[mi][ni], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        if (ILV && ISSUE) {                                      // 5 loads over 4 slots
          __builtin_amdgcn_sched_barrier(0);
          issue_one(kt + NSTG - 1, slot); ++slot;
          if (mi == 0) { issue_one(kt + NSTG - 1, slot); ++slot; }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      return;
    }
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int co = ((ks * 2 + hi) ^ xr) << 4;
      bf16x8 af[MI], bf[3];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[mi] = *reinterpret_cast<const bf16x8*>(st + offA + mi * 32 * ROWB + co);
#pragma unroll
      for (int ni = 0; ni < 3; ++ni) bf[ni] = *reinterpret_cast<const bf16x8*>(st + offB + ni * 32 * ROWB + co);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 3; ++ni) acc[mi][ni] = TR ? mfma32(bf[ni], af[mi], acc[mi][ni]) : mfma32(af[mi], bf[ni], acc[mi][ni]);
        if (ILV && ISSUE && slot < LOADS_PER_TILE) {
          __builtin_amdgcn_sched_barrier(0);
          issue_one(kt + NSTG - 1, slot);
          __builtin_amdgcn_sched_barrier(0);
          ++slot;
        }
      }
    }
  };
#pragma unroll
  for (int t = 0; t < NSTG - 1; ++t)
    if (t < nk) issue(t);
  const int nfull = nk - (NSTG - 1) > 0 ? nk - (NSTG - 1) : 0;    // tiles that still have a successor to fetch
  for (int kt = 0; kt < nfull; ++kt) {
    // my share of tile kt has landed (loads retire in order; the younger NSTG-2 tiles may still be in flight); barrier =>
    // everyone's has, and everyone is done reading stage (kt-1) % NSTG, which the next issue overwrites
    if constexpr (NSTG == 2) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    else if constexpr (LOADS_PER_TILE == 5) asm volatile("s_waitcnt vmcnt(5)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (!ILV) issue(kt + NSTG - 1);
    tile(kt, std::true_type{});
  }
  for (int kt = nfull; kt < nk; ++kt) {                           // drain: nothing left to fetch
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    tile(kt, std::false_type{});
  }
  asm volatile("s_barrier" ::: "memory");

  }
  if constexpr (TR) {
    constexpr int TP = row384::TR_PITCH;
    char* stg = smem_raw + wid * (32 * TP);                       // this wave's strip: 32 rows x 96 bf16 (+ pad)
    const float dq = F8 ? (p.dq ? *p.dq : 1.0f) * (p.dq_mul != 0.f ? p.dq_mul : 1.0f) / (p.dq_div ? *p.dq_div : 1.0f) : 1.0f;
    bf16* cw = reinterpret_cast<bf16*>(p.C) + n0 + wn * 96;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int ni = 0; ni < 3; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c4 = ni * 32 + 8 * g + 4 * hi;                // my four columns of this quad inside the wave's 96
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(sBiasT + wn * 96 + c4);
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (F8 ? acc[mi][ni][4 * g + e] * dq : acc[mi][ni][4 * g + e]) + b4[e];
          *reinterpret_cast<u32x2*>(stg + l31 * TP + c4 * 2) = pack_bf16x4(v);
        }
      // LDS operations of one wave execute in order: the reads below see the writes above, the next block row's writes follow the reads
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int idx = j * 64 + lane, r = idx / 12, c = idx - r * 12;
        const u32x4 w = *reinterpret_cast<const u32x4*>(stg + r * TP + c * 16);
        const int row = m0 + wm * 32 * MI + mi * 32 + r;
        if (row < p.M) __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(cw + (size_t)row * p.ldc + c * 8));
      }
    }
    return;
  }

  // Epilogue: the fp32 tile goes through LDS 32 rows at a time so that every global access is a 16-B piece of a full
  // 384-column row.  Part (mi, h) takes 16 rows of accumulator block mi from EVERY wave (two 16-row groups, one per
  // wave row), so all waves retire the same 24 accumulator registers per part.  8-column-slot epilogues: the registers
  // freed by the dump hold that part's global loads (saved activations / residual), issued as one batch before the staging
  // barrier while the stores follow it (see epi_fetch8).  Row-wise epilogues: see rowin_issue.
  constexpr int RP = 32, SLOTS = RP * 48 / THREADS, LNROWS = RP / WAVES, NPAIR = LNROWS / 2, NPART = 2 * MI;
  constexpr bool fused_ln = LN && EPI == EPI_RESID;               // residual + LayerNorm forward of the new row
  constexpr bool lnbwd = EPI == EPI_LNBWD;                        // LayerNorm backward of the dgrad row
  constexpr bool rowwise = fused_ln || lnbwd;                     // half a wave per row: plain staging rows, 16 B per lane
  constexpr int NT = lnbwd ? 2 : 1;                               // streamed row tensors: residual | residual gradient + LayerNorm input
  // VMEM operations of one pair: 5 stores (3 x 16 B fp32 row + 2 bf16 row) + 6 / 3 LDS-DMA refill pieces (LayerNorm backward /
  // residual + LayerNorm).  Outstanding behind a refill when it is needed: see rowin_wait.
  constexpr int NVM = (lnbwd ? 5 + 6 : 5 + 3 + 5) - 2;
  const bool exact_vm = m0 + BMR <= p.M && (!lnbwd || ((p.lnb_g || p.q8) && p.resid));   // (a bf16 row AND its e4m3 copy: more stores than NVM counts -- a stricter wait, never a shorter one)
  using EL = row384::EpiLds<MI, rowwise, NT>;
  float* sC = reinterpret_cast<float*>(smem_raw);
  float* sBias = sC + EL::SC;                                     // bias of this block's 384 columns (zeros when absent)
  float* sGamma = sBias + EL::SBIAS; float* sBeta = sGamma + BNR; // LayerNorm affine parameters (row-wise epilogues)
  float* sScale = sBeta + BNR;                                    // per-row DropPath scale of this block's rows
  float* sCol = sGamma;                                           // EPI_DGELU: column sums of du (fc1 bias gradient); no LN there
  // the row-wise epilogues' index math starts from a laundered copy of the thread id: computed from `tid` it is hoisted above
  // the main loop, where there is not a single register to spare (measured: three scratch reloads per k-tile)
  int tid_e = tid;
  asm volatile("" : "+v"(tid_e));
  const int lane_e = tid_e & 63, wid_e = tid_e >> 6;
  float* sStat = sScale + BMR;                                    // row-wise: {mean, rstd} of the block's rows (forward: written out once per tile ; backward: fetched once)
  char* sIn = reinterpret_cast<char*>(sStat + 2 * BMR) + (wid_e * NPAIR) * NT * ROWIN_PAIR_BYTES;   // this wave's input slots [pair][tensor]
  const int hh = lane_e >> 5, li = lane_e & 31;
  auto tile_row_of = [&](int part, int rl) { return (rl >> 4) * (32 * MI) + (part >> 1) * 32 + (part & 1) * 16 + (rl & 15); };   // staged row -> row of the block tile
  auto rowin_part = [&](int part, int q) {                        // DMA of pair q of `part`
    const int row0 = m0 + tile_row_of(part, wid_e * LNROWS + 2 * q);
    rowin_issue<NT>(p.resid, lnbwd ? p.lnb_x : nullptr, p.M, row0, sIn + q * NT * ROWIN_PAIR_BYTES, lane_e);
  };
  if constexpr (rowwise) {                                        // part 0's rows arrive under its staging
#pragma unroll
    for (int q = 0; q < NPAIR; ++q) rowin_part(0, q);
  }
  const float dqv = F8 ? (p.dq ? *p.dq : 1.0f) * (p.dq_mul != 0.f ? p.dq_mul : 1.0f) / (p.dq_div ? *p.dq_div : 1.0f) : 1.0f;
  if constexpr (rowwise) {                                        // tables by LDS-DMA (lds_fill64): landed by the vmcnt(0) + barrier of part 0
    const int c = wid_e * 64 + lane_e;                            // column (waves 0-5) / row of the tile (waves 0-3)
    if (wid_e < BNR / 64) {
      if (p.bias) lds_fill64(p.bias + n0 + c, sBias + wid_e * 64); else sBias[c] = 0.f;
      lds_fill64(p.ln_gamma + c, sGamma + wid_e * 64);
      if (fused_ln) lds_fill64(p.ln_beta + c, sBeta + wid_e * 64);
    }
    if (wid_e < BMR / 64) {
      const int r = m0 + c < p.M ? m0 + c : p.M - 1;              // rows >= M: clamped, never used
      if (p.row_scale) lds_fill64(p.row_scale + r / p.rows_per_seq, sScale + wid_e * 64); else sScale[c] = 1.0f;
      if constexpr (lnbwd) { lds_fill64(p.ln_mean + r, sStat + wid_e * 64); lds_fill64(p.ln_rstd + r, sStat + BMR + wid_e * 64); }   // backward: [mean | rstd]
    }
  } else {
    // bias in the same two-plane layout as the tile (position 4 g + e of plane h <- column 8 g + 4 h + e), also by LDS-DMA: the
    // gather is on the source side.  Landed by the vmcnt(0) in front of part 0's staging barrier.
    if (wid_e < 6) {
      const int pos = (wid_e % 3) * 64 + lane_e, plane = wid_e / 3;
      float* dst = sBias + plane * PLANE1 + (wid_e % 3) * 64;
      if (p.bias) lds_fill64(p.bias + n0 + 8 * (pos >> 2) + 4 * plane + (pos & 3), dst); else dst[lane_e] = 0.f;
    }
    if (EPI == EPI_DGELU && tid < BNR) sCol[tid] = 0.f;
    if constexpr (EPI == EPI_RESID) {
      if (tid < BMR) {
        const int row = m0 + tid;
        sScale[tid] = (p.row_scale && row < p.M) ? p.row_scale[row / p.rows_per_seq] : 1.0f;
      }
    }
  }
  // EPI_DGELU: column sums of du (fc1 bias gradient).  A thread's slot i covers the same 8 columns in every part, so it keeps 8 running
  // sums per slot in registers (the accumulator blocks retired by the staging make room) and the block folds them ONCE, after the
  // last part.  (Writing each part's products back to the staging tile and summing columns behind two more barriers per part cost
  // the base-geometry dGELU GEMM ~25 %; one LDS atomic per element 2.7x.)
  f32x4 csr[EPI == EPI_DGELU ? SLOTS : 1][2];
#pragma unroll
  for (int i = 0; i < (EPI == EPI_DGELU ? SLOTS : 1); ++i) { csr[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; csr[i][1] = csr[i][0]; }
  float q8s = 1.0f, omax = 0.f;                                   // EPI_DGELU: scale of the e4m3 copy of du ; running max |du| (next step's scale)
  if constexpr (EPI == EPI_DGELU) { if (p.q8 && p.q8_scale_ptr) q8s = *p.q8_scale_ptr; }
  // fc1 + GELU (fp8 forward): the running scale of the e4m3 copy is read ONCE here -- read per slot inside epilogue8 it is a load in the
  // middle of a part's stores (loads and stores retire through one counter)
  if constexpr (EPI == EPI_BIAS_GELU) { if (p.q8) q8s = p.q8_scale_ptr ? *p.q8_scale_ptr : p.q8_scale; }
  LnbCols lcs;
  if constexpr (lnbwd) {
#pragma unroll
    for (int i = 0; i < 6; ++i) lcs.dg[i] = lcs.db[i] = lcs.du[i] = 0.f;
  }
  RowQ8 rq8{1.0f, 0.f, 0u};                                       // row-wise epilogues: optional e4m3 copy of the bf16 row they write (LayerNorm output / row_scale dx)
  if constexpr (rowwise && Q8) { if (p.q8) rq8.s = p.q8_scale_ptr ? *p.q8_scale_ptr : (p.q8_scale != 0.f ? p.q8_scale : 1.0f); }
#pragma unroll
  for (int part = 0; part < NPART; ++part) {
    const int mi = part >> 1, h = part & 1;
    auto tile_row = [&](int rl) { return tile_row_of(part, rl); };
#pragma unroll
    for (int ni = 0; ni < 3; ++ni)
#pragma unroll
      for (int r8 = 0; r8 < 8; ++r8) {
        const int lrow_ = wm * 16 + (r8 & 3) + 8 * (r8 >> 2) + 4 * hi;
        // Staging layout.  Row-wise epilogues: plain rows (pitch CLD), read back 16 B per lane -- conflict-free.  Other
        // epilogues read 8 consecutive columns per thread as two f32x4; in plain rows the 16-lane service groups of
        // ds_read_b128 then stride 32 B and hit every bank twice (SQ counters, round 1: 15-18 % of the LDS cycles of the bf16 /
        // GELU epilogues were bank conflicts).  There the low and the high four columns of every 8-column slot live in two
        // planes (columns 0-191 / 208-399 of a 448-float row): a service group reads 16 consecutive 16-B pieces of one plane.
        const int ccol = wn * 96 + ni * 32 + l31;
        const int pos = rowwise ? lrow_ * CLD + ccol : lrow_ * CLD2 + ((ccol >> 3) << 2) + (ccol & 3) + ((ccol & 4) ? PLANE1 : 0);
        sC[pos] = F8 ? acc[mi][ni][h * 8 + r8] * dqv : acc[mi][ni][h * 8 + r8];
      }
    // (Issuing these loads one part ahead was measured: no gain -- 219 vs 218 us on fc2+residual -- and 17 spilled
    // registers; a part's time is set by the CU's memory throughput, not by the exposed round trip.)
    EpiAux aux[rowwise ? 1 : SLOTS];
    if constexpr (rowwise) {
      rowin_wait<NVM>(exact_vm && part > 0);                       // pair 0's rows (and statistics) of this part have landed in my slot (part 0: everything issued so far)
    } else {
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) {
        const int idx = tid + THREADS * i, row = m0 + tile_row(idx / 48);
        if (row < p.M) epi_fetch8<EPI, false>(p, row, n0 + (idx % 48) * 8, aux[i]);
      }
      if (part == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the bias table (LDS-DMA) has landed -- mine; the barrier makes it everyone's
    }
    lds_barrier();
    if constexpr (rowwise) {
#pragma unroll
      for (int q = 0; q < NPAIR; ++q) {
        const int rl = wid_e * LNROWS + 2 * q + hh, trow = tile_row(rl), row = m0 + trow;
        const char* slot = sIn + q * NT * ROWIN_PAIR_BYTES;
        // this pair's rows have landed (pair 0: in front of the staging barrier; part 0: both).  In the LAST part pair 0 issues no
        // refill, so fewer operations follow this pair's refill than NVM assumes: wait for everything there.
        if (q > 0 && part > 0) rowin_wait<NVM>(exact_vm && part + 1 < NPART);
        if constexpr (lnbwd) {
          const float mu = sStat[trow], rs = sStat[BMR + trow];
          lnb_row<Q8>(p, row, row < p.M, sC + rl * CLD, sGamma, sScale[trow], mu, rs, slot, p.resid != nullptr, hh, li, lcs, rq8);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the slot has been read: refill it with the next part's rows
          if (part + 1 < NPART) rowin_part(part + 1, q);
        } else {
          f32x4 rres[3];
          rowin_read(slot, 0, hh, li, rres);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the slot has been copied out: refill it with the next part's rows
          if (part + 1 < NPART) rowin_part(part + 1, q);
          // Fused residual + LayerNorm of the NEXT sub-layer: replaces a separate HBM pass (ln_fwd_kernel) over x.
          if (row < p.M) lnf_row<Q8>(p, row, sC + rl * CLD, sBias, sGamma, sBeta, sScale[trow], rres, li, sStat + 2 * trow, rq8);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) {
        const int idx = tid + THREADS * i, rl = idx / 48, c8 = (idx % 48) * 8;
        const int trow = tile_row(rl), row = m0 + trow;
        f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = w0;
        if (row < p.M) {
          if constexpr (EPI == EPI_RESID) aux[i].s = sScale[trow];
          if constexpr (EPI == EPI_DGELU || EPI == EPI_BIAS_GELU) aux[i].s = q8s;
          epilogue8<EPI>(p, row, n0 + c8, *reinterpret_cast<const f32x4*>(sC + rl * CLD2 + (c8 >> 1)), *reinterpret_cast<const f32x4*>(sC + rl * CLD2 + PLANE1 + (c8 >> 1)),
                         *reinterpret_cast<const f32x4*>(sBias + (c8 >> 1)), *reinterpret_cast<const f32x4*>(sBias + PLANE1 + (c8 >> 1)), aux[i], w0, w1);
        }
        if constexpr (EPI == EPI_DGELU || EPI == EPI_BIAS_GELU) {
          if (p.q8_amax) {                                         // dGELU: max |du| ; fc1 + GELU (fp8 forward): max |a| -- the next step's scale of this site
#pragma unroll
            for (int e = 0; e < 4; ++e) omax = fmaxf(omax, fmaxf(fabsf(w0[e]), fabsf(w1[e])));
          }
        }
        if constexpr (EPI == EPI_DGELU) { csr[i][0] += w0; csr[i][1] += w1; }   // rows beyond M contribute zeros
      }
    }
    if (part < NPART - 1) lds_barrier();
  }
  if constexpr (EPI == EPI_DGELU) {
    if (p.colsum) {                                               // fold the 32 row slots of every column: the slots' sums go to their staging positions once
      lds_barrier();
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) {
        const int idx = tid + THREADS * i, rl = idx / 48, c8 = (idx % 48) * 8;
        *reinterpret_cast<f32x4*>(sC + rl * CLD2 + (c8 >> 1)) = csr[i][0];
        *reinterpret_cast<f32x4*>(sC + rl * CLD2 + PLANE1 + (c8 >> 1)) = csr[i][1];
      }
      lds_barrier();
      if (tid < BNR) {
        const int ph = ((tid >> 3) << 2) + (tid & 3) + ((tid & 4) ? PLANE1 : 0);
        float t = 0.f;
#pragma unroll 8
        for (int r = 0; r < RP; ++r) t += sC[r * CLD2 + ph];
        atomicAdd(p.colsum + n0 + tid, t);
      }
    }
    if (p.q8_amax) amax_post(p.q8_amax, wave_max(omax), lane, blockIdx.x * WAVES + wid);
  }
  if constexpr (EPI == EPI_BIAS_GELU) {
    if (p.q8 && p.q8_amax) amax_post(p.q8_amax, wave_max(omax), lane, blockIdx.x * WAVES + wid);
  }
  if constexpr (rowwise && Q8) {
    if (p.q8_amax) amax_post(p.q8_amax, wave_max(rq8.amax), lane, blockIdx.x * WAVES + wid);
    if constexpr (fused_ln) { if (p.q8) f8_sat_add(p.q8_sat, rq8.nclip); }
  }
  if constexpr (lnbwd) lnb_flush<WAVES>(p, lcs, sC, tid);        // 9,216 floats of the staging area, behind a barrier
  if constexpr (fused_ln) {                                       // row statistics of the whole tile: two store instructions per wave instead of two per row
    lds_barrier();
    if (tid < BMR && m0 + tid < p.M) { p.ln_mean[m0 + tid] = sStat[2 * tid]; p.ln_rstd[m0 + tid] = sStat[2 * tid + 1]; }
  }
}

// ---- 4-wave blocks, two per CU -----------------------------------------------------------------------------------------
// Same wave tile as the 256 x 384 kernel (128 x 96 = 12 accumulators, 256 registers), but FOUR waves per block so that two
// independent blocks share a CU (one wave of each per SIMD): while one block is in its epilogue (VALU + stores) or
// blocked on LDS-DMA issue, the other runs its MFMAs -- the phases that the one-block-per-CU kernel serialises
// (profiles/r02_trace_gemm.txt: 2608 cycles per k-tile against 1536 of MFMA time, then an epilogue with idle matrix pipes).
//   WM = 2: waves 2 x 2, block tile 256 x 192.  A row panel is staged by the blocks of both column halves (the second read
//           is an L2 hit), B half as often per row: the best operand mix measured (tools/probes/mix_probe: A 6.1 TB/s).
//   WM = 1: waves 1 x 4, block tile 128 x 384: whole rows, for the row-wise epilogues (residual + LayerNorm, LayerNorm backward).
// Operand rings are split: A (the HBM stream) NA stages deep, B (L2-resident weights) two.  vmcnt retires in order per
// wave, so the operands are issued by DIFFERENT waves (A: wave 0 [and 1]; B: the others): the A loaders keep NA - 2 tiles
// in flight behind the one being waited for, which a wave that also issued B could not (waiting for B(t+1) would retire
// every older A load first).  ~48 KB of HBM reads in flight per CU is what the memory system needs to stream at full rate
// (tools/probes/l2lds_probe: 10 B/clk/CU x ~4800 clk).
namespace w4 {
template <int WM> struct Geo {
  static constexpr int WN = 4 / WM, BM = 128 * WM, BNB = 96 * WN, THREADS = 256;
  static constexpr int A_STAGE = BM * 64, B_STAGE = BNB * 64;                 // 64-B rows (BK = 32): 8 / 16 KB ; 24 / 12 KB
  static constexpr int NA = WM == 1 ? 4 : 3, NB = 2;
  static constexpr int A_BYTES = NA * A_STAGE, RING = A_BYTES + NB * B_STAGE;  // 32 + 48 = 80 KB ; 48 + 24 = 72 KB
  static constexpr int CLD = BNB + 4, RP = 16 * WM;                           // staged rows per epilogue part
  static constexpr int PL1 = BNB / 2 + 16, CLD2 = BNB == 192 ? 224 : 448;     // two-plane staging of the epilogues without LayerNorm (conflict-free f32x4 read-back)
  static constexpr int EPI_BYTES = RP * CLD2 * 4 + (CLD2 + 2 * BNB) * 4 + BM * 4;
  // row-wise epilogues (WM == 1): staging rows of pitch CLD, bias / gamma / beta / scale, NT streamed row tensors [wave][pair][tensor][2 rows]
  template <int NT> static constexpr int rowwise_bytes() { return (RP * CLD + 3 * BNB + 3 * BM) * 4 + 4 * 2 * NT * 2 * 384 * 4; }   // 55,552 ; 80,128 B
  template <int EPI, bool LN> static constexpr int lds_bytes() {
    constexpr bool rowwise = (LN && EPI == EPI_RESID) || EPI == EPI_LNBWD;
    constexpr int epi = rowwise ? (EPI == EPI_LNBWD ? rowwise_bytes<2>() : rowwise_bytes<1>()) : EPI_BYTES;
    return RING > epi ? RING : epi;
  }
  static constexpr int A_WAVES = WM == 1 ? 1 : 2, B_WAVES = 4 - A_WAVES;
  static constexpr int PA = (BM / 16) / A_WAVES, PB = (BNB / 16) / B_WAVES;   // 1-KiB pieces per loader wave per k-tile: 8, 8 ; 8, 6
  static constexpr int SPR = BNB / 8;                                        // 8-column slots per row
};
}

// F8 (round 6): e4m3 operands as in gemm_nt_row384_kernel -- the rings are byte-identical (64-B rows = 64 e4m3 values of K), one
// v_mfma_scale_f32_32x32x64_f8f6f4 per accumulator and k-tile; the 1 s local views of the e4m3 step at d = 384 (M = 26624: 104 tiles of 256 rows on 256 CUs).
template <int EPI, int WM, bool LN = false, bool Q8 = false, bool F8 = false>
__global__ __launch_bounds__(256, 2) void gemm_nt_w4_kernel(GemmArgs p) {
  using G = w4::Geo<WM>;
  constexpr int WN = G::WN, BM = G::BM, BNB = G::BNB, CLD = G::CLD, NA = G::NA, NB = G::NB, MI = 4;
  static_assert(!LN || WM == 1, "the fused LayerNorm needs whole rows");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  typedef const void __attribute__((address_space(1))) * gptr_t;
  typedef void __attribute__((address_space(3))) * lptr_t;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN, hi = lane >> 5, l31 = lane & 31;
  const int ntn = p.N / BNB;
  const int ntm = (p.M + BM - 1) / BM;
  const int id = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (id / ntn) * BM, n0 = (id % ntn) * BNB;

  char* ldsA = smem_raw; char* ldsB = smem_raw + G::A_BYTES;
  const bool a_loader = wid < G::A_WAVES;
  const int lrow = lane >> 2, lchunk = lane & 3;
  // per-lane source of piece 0 of this wave's share; piece j is 16 rows further down
  const bf16* src;
  int piece0;
  if (a_loader) {
    piece0 = wid * G::PA;
    const int row = piece0 * 16 + lrow;
    int ra = m0 + row; ra = ra < p.M ? ra : p.M - 1;               // clamp: rows >= M are never stored
    src = p.A + (size_t)ra * p.lda + (lchunk ^ ((row >> 2) & 3)) * 8;
  } else {
    piece0 = (wid - G::A_WAVES) * G::PB;
    const int row = piece0 * 16 + lrow;
    src = p.B + (size_t)(n0 + row) * p.ldb + (lchunk ^ ((row >> 2) & 3)) * 8;
  }
  // rows >= M inside a piece: only the last M-tile can be ragged; clamp per piece there
  const bool ragged = m0 + BM > p.M;
  auto issue = [&](int kt) {
    if (a_loader) {
      char* st = ldsA + (kt % NA) * G::A_STAGE + piece0 * 1024;
#pragma unroll
      for (int j = 0; j < G::PA; ++j) {
        const bf16* s_ = src + (size_t)j * 16 * p.lda;
        if (ragged) {
          const int row = (piece0 + j) * 16 + lrow;
          int ra = m0 + row; ra = ra < p.M ? ra : p.M - 1;
          s_ = p.A + (size_t)ra * p.lda + (lchunk ^ ((row >> 2) & 3)) * 8;
        }
        __builtin_amdgcn_global_load_lds((gptr_t)(s_ + kt * BK), (lptr_t)(st + j * 1024), 16, 0, 0);
      }
    } else {
      char* st = ldsB + (kt % NB) * G::B_STAGE + piece0 * 1024;
#pragma unroll
      for (int j = 0; j < G::PB; ++j)
        __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)j * 16 * p.ldb + kt * BK), (lptr_t)(st + j * 1024), 16, 0, 0);
    }
  };
  f32x16 acc[MI][3];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / BK;
  const int xr = (l31 >> 2) & 3;
  const int offA = (wm * 128 + l31) * 64, offB = (wn * 96 + l31) * 64;
  // prologue: A loaders NA - 1 tiles ahead, B loaders NB - 1
  if (a_loader) {
#pragma unroll
    for (int t = 0; t < NA - 1; ++t) if (t < nk) issue(t);
  } else {
#pragma unroll
    for (int t = 0; t < NB - 1; ++t) if (t < nk) issue(t);
  }
  for (int kt = 0; kt < nk; ++kt) {
    // my share of tile kt has landed (my younger tiles may still be in flight); barrier => everyone's has, and everyone
    // is done reading tile kt - 1, whose stages the next issue overwrites
    if (a_loader) {
      const int younger = nk - 1 - kt < NA - 2 ? nk - 1 - kt : NA - 2;
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");       // PA == 8 in both geometries
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // NB == 2: nothing younger
    }
    asm volatile("s_barrier" ::: "memory");
    if (a_loader) { if (kt + NA - 1 < nk) issue(kt + NA - 1); }
    else { if (kt + NB - 1 < nk) issue(kt + NB - 1); }
    const char* sa = ldsA + (kt % NA) * G::A_STAGE; const char* sb = ldsB + (kt % NB) * G::B_STAGE;
    if constexpr (F8) {                                            // lane (row l31, half hi): bytes [32 hi, 32 hi + 32) of its 64-byte row = chunks 2 hi, 2 hi + 1 (swizzled)
      typedef int v4i_ __attribute__((ext_vector_type(4)));
      typedef int v8i_ __attribute__((ext_vector_type(8)));
      const int c0 = ((2 * hi) ^ xr) << 4, c1 = ((2 * hi + 1) ^ xr) << 4;
      auto frag = [&](const char* rowp) {
        const v4i_ lo = *reinterpret_cast<const v4i_*>(rowp + c0), hi4 = *reinterpret_cast<const v4i_*>(rowp + c1);
        return v8i_{lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
      };
      v8i_ b8[3];
#pragma unroll
      for (int ni = 0; ni < 3; ++ni) b8[ni] = frag(sb + offB + ni * 32 * 64);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const v8i_ a8 = frag(sa + offA + mi * 32 * 64);
#pragma unroll
        for (int ni = 0; ni < 3; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8[ni], acc[mi][ni], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
      }
    } else
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int co = ((ks * 2 + hi) ^ xr) << 4;
      bf16x8 af[MI], bf[3];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[mi] = *reinterpret_cast<const bf16x8*>(sa + offA + mi * 32 * 64 + co);
#pragma unroll
      for (int ni = 0; ni < 3; ++ni) bf[ni] = *reinterpret_cast<const bf16x8*>(sb + offB + ni * 32 * 64 + co);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 3; ++ni) acc[mi][ni] = mfma32(af[mi], bf[ni], acc[mi][ni]);
    }
  }
  asm volatile("s_barrier" ::: "memory");

  // Epilogue: as in the 8-wave kernel, the fp32 tile goes through LDS so that every global access is a 16-B piece of a
  // contiguous row segment.  Part (mi, h) stages 16 rows of accumulator block mi from every wave: 16 * WM rows x BNB columns.
  constexpr int NPART = 2 * MI;
  constexpr bool fused_ln = LN && EPI == EPI_RESID;
  constexpr bool lnbwd = EPI == EPI_LNBWD;
  constexpr bool rowwise = fused_ln || lnbwd;                     // (WM == 1) half a wave per row, inputs through per-wave LDS slots: see rowin_issue
  constexpr int NT = lnbwd ? 2 : 1;
  constexpr int NVM = (lnbwd ? 5 + 6 : 5 + 3 + 5) - 2;            // see the 8-wave kernel
  const bool exact_vm = m0 + BM <= p.M && (!lnbwd || ((p.lnb_g || p.q8) && p.resid));
  static_assert(!lnbwd || WM == 1, "the LayerNorm backward needs whole rows");
  float* sC = reinterpret_cast<float*>(smem_raw);
  float* sBias = sC + G::RP * (rowwise ? CLD : G::CLD2);
  float* sGamma = sBias + (rowwise ? BNB : G::CLD2); float* sBeta = sGamma + BNB;
  float* sScale = sBeta + BNB;
  float* sCol = sGamma;                                           // EPI_DGELU column sums (no LN there)
  float* sStat = sScale + BM;
  char* sIn = reinterpret_cast<char*>(sStat + 2 * BM) + (wid * 2) * NT * ROWIN_PAIR_BYTES;   // this wave's input slots [pair][tensor]
  const int hh = lane >> 5, li = lane & 31;
  auto tile_row_of = [&](int part, int rl) { return (rl >> 4) * 128 + (part >> 1) * 32 + (part & 1) * 16 + (rl & 15); };   // staged row -> row of the block tile
  auto rowin_part = [&](int part, int q) {
    const int row0 = m0 + tile_row_of(part, wid * 4 + 2 * q);
    rowin_issue<NT>(p.resid, lnbwd ? p.lnb_x : nullptr, p.M, row0, sIn + q * NT * ROWIN_PAIR_BYTES, lane);
  };
  if constexpr (rowwise) {
#pragma unroll
    for (int q = 0; q < 2; ++q) rowin_part(0, q);
  }
  if constexpr (rowwise) {                                        // tables by LDS-DMA (lds_fill64), see the 8-wave kernel
#pragma unroll
    for (int c0 = wid * 64; c0 < BNB; c0 += 256) {
      const int c = c0 + lane;
      if (p.bias) lds_fill64(p.bias + n0 + c, sBias + c0); else sBias[c] = 0.f;
      lds_fill64(p.ln_gamma + c, sGamma + c0);
      if (fused_ln) lds_fill64(p.ln_beta + c, sBeta + c0);
    }
  } else {
    for (int c = tid; c < BNB; c += 256) {
      sBias[((c >> 3) << 2) + (c & 3) + ((c & 4) ? G::PL1 : 0)] = p.bias ? p.bias[n0 + c] : 0.f;
      if (EPI == EPI_DGELU) sCol[c] = 0.f;
    }
  }
  LnbCols lcs;
  if constexpr (lnbwd) {
#pragma unroll
    for (int i = 0; i < 6; ++i) lcs.dg[i] = lcs.db[i] = lcs.du[i] = 0.f;
  }
  const float dqv = F8 ? (p.dq ? *p.dq : 1.0f) * (p.dq_mul != 0.f ? p.dq_mul : 1.0f) / (p.dq_div ? *p.dq_div : 1.0f) : 1.0f;
  RowQ8 rq8{1.0f, 0.f, 0u};                                       // see the 8-wave kernel
  if constexpr (rowwise && Q8) { if (p.q8) rq8.s = p.q8_scale_ptr ? *p.q8_scale_ptr : (p.q8_scale != 0.f ? p.q8_scale : 1.0f); }
  if constexpr (rowwise) {
    if (wid < BM / 64) {
      const int t = wid * 64 + lane, r = m0 + t < p.M ? m0 + t : p.M - 1;
      if (p.row_scale) lds_fill64(p.row_scale + r / p.rows_per_seq, sScale + wid * 64); else sScale[t] = 1.0f;
      if constexpr (lnbwd) { lds_fill64(p.ln_mean + r, sStat + wid * 64); lds_fill64(p.ln_rstd + r, sStat + BM + wid * 64); }   // backward: [mean | rstd]
    }
  } else if constexpr (EPI == EPI_RESID) {
    if (tid < BM) {
      const int row = m0 + tid;
      sScale[tid] = (p.row_scale && row < p.M) ? p.row_scale[row / p.rows_per_seq] : 1.0f;
    }
  }
#pragma unroll
  for (int part = 0; part < NPART; ++part) {
    const int mi = part >> 1, h = part & 1;
    auto tile_row = [&](int rl) { return tile_row_of(part, rl); };
#pragma unroll
    for (int ni = 0; ni < 3; ++ni)
#pragma unroll
      for (int r8 = 0; r8 < 8; ++r8) {
        const int lr = wm * 16 + (r8 & 3) + 8 * (r8 >> 2) + 4 * hi;
        const int ccol = wn * 96 + ni * 32 + l31;
        sC[rowwise ? lr * CLD + ccol : lr * G::CLD2 + ((ccol >> 3) << 2) + (ccol & 3) + ((ccol & 4) ? G::PL1 : 0)] = F8 ? acc[mi][ni][h * 8 + r8] * dqv : acc[mi][ni][h * 8 + r8];
      }
    EpiAux aux[rowwise ? 1 : 3];
    if constexpr (rowwise) {
      rowin_wait<NVM>(exact_vm && part > 0);                       // pair 0's rows (and statistics) of this part have landed in my slot (part 0: everything issued so far)
    } else {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int idx = tid + 256 * i, row = m0 + tile_row(idx / G::SPR);
        if (row < p.M) epi_fetch8<EPI, false>(p, row, n0 + (idx % G::SPR) * 8, aux[i]);
      }
    }
    lds_barrier();
    if constexpr (rowwise) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int rl = wid * 4 + 2 * q + hh, trow = tile_row(rl), row = m0 + trow;
        const char* slot = sIn + q * NT * ROWIN_PAIR_BYTES;
        // this pair's rows have landed (pair 0: in front of the staging barrier; part 0: both).  In the LAST part pair 0 issues no
        // refill, so fewer operations follow this pair's refill than NVM assumes: wait for everything there.
        if (q > 0 && part > 0) rowin_wait<NVM>(exact_vm && part + 1 < NPART);
        if constexpr (lnbwd) {
          const float mu = sStat[trow], rs = sStat[BM + trow];
          lnb_row<Q8>(p, row, row < p.M, sC + rl * CLD, sGamma, sScale[trow], mu, rs, slot, p.resid != nullptr, hh, li, lcs, rq8);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the slot has been read: refill it with the next part's rows
          if (part + 1 < NPART) rowin_part(part + 1, q);
        } else {
          f32x4 rres[3];
          rowin_read(slot, 0, hh, li, rres);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the slot has been copied out: refill it with the next part's rows
          if (part + 1 < NPART) rowin_part(part + 1, q);
          if (row < p.M) lnf_row<Q8>(p, row, sC + rl * CLD, sBias, sGamma, sBeta, sScale[trow], rres, li, sStat + 2 * trow, rq8);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int idx = tid + 256 * i, rl = idx / G::SPR, c8 = (idx % G::SPR) * 8;
        const int trow = tile_row(rl), row = m0 + trow;
        if (row < p.M) {
          f32x4 w0, w1;
          if constexpr (EPI == EPI_RESID) aux[i].s = sScale[trow];
          epilogue8<EPI>(p, row, n0 + c8, *reinterpret_cast<const f32x4*>(sC + rl * G::CLD2 + (c8 >> 1)), *reinterpret_cast<const f32x4*>(sC + rl * G::CLD2 + G::PL1 + (c8 >> 1)),
                         *reinterpret_cast<const f32x4*>(sBias + (c8 >> 1)), *reinterpret_cast<const f32x4*>(sBias + G::PL1 + (c8 >> 1)), aux[i], w0, w1);
          if constexpr (EPI == EPI_DGELU) {
            if (p.colsum) {
#pragma unroll
              for (int e = 0; e < 4; ++e) { atomicAdd(sCol + c8 + e, w0[e]); atomicAdd(sCol + c8 + 4 + e, w1[e]); }
            }
          }
        }
      }
    }
    if (part < NPART - 1) lds_barrier();
  }
  if constexpr (EPI == EPI_DGELU) {
    if (p.colsum) {
      lds_barrier();
      for (int c = tid; c < BNB; c += 256) atomicAdd(p.colsum + n0 + c, sCol[c]);
    }
  }
  if constexpr (rowwise && Q8) {
    if (p.q8_amax) amax_post(p.q8_amax, wave_max(rq8.amax), lane, blockIdx.x * 4 + wid);
    if constexpr (fused_ln) { if (p.q8) f8_sat_add(p.q8_sat, rq8.nclip); }
  }
  if constexpr (lnbwd) lnb_flush<4>(p, lcs, sC, tid);
  if constexpr (fused_ln) {
    lds_barrier();
    if (tid < BM && m0 + tid < p.M) { p.ln_mean[m0 + tid] = sStat[2 * tid]; p.ln_rstd[m0 + tid] = sStat[2 * tid + 1]; }
  }
}

#include "gemm_p8.h"
#include "gemm_tt.h"

// ---- wgrad: dW[n,k] += sum_m dY[m,n] X[m,k] -------------------------------------------------------------------------
constexpr int WM = 64;                          // contraction rows per stage
constexpr int W_LD = 128 + 32;                  // 160 bf16 = 320 B row stride: 4 consecutive rows x 64 B (one ds_read_b64_tr_b16 half) on 4 distinct bank windows
constexpr int W_TILE = WM * W_LD;
constexpr int WGRAD_LDS_BYTES = 2 * 2 * W_TILE * 2;   // 81,920 B: two blocks fill the CU's 160 KB exactly

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(WgradArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16* smem = reinterpret_cast<bf16*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wn = wid >> 1, wk = wid & 1, hi = lane >> 5, l31 = lane & 31;
  const int ntn = p.N / 128, ntk = p.K / 128;
  // XCD-aware order: all output tiles of one M-split (they stream the SAME dY / X rows) run on one XCD, so each
  // row is fetched from HBM once per XCD-resident split instead of once per tile (PMC: 3.3x over-fetch without this)
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int tile = id % (ntn * ntk), split = id / (ntn * ntk);
  const int n0 = (tile / ntk) * 128, k0 = (tile % ntk) * 128;
  const int m_begin = split * p.m_per_split;
  int m_end = m_begin + p.m_per_split; if (m_end > p.M) m_end = p.M;
  if (m_begin >= m_end) return;
  const int nst = (m_end - m_begin + WM - 1) / WM;

  int s_row[4], s_col[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int c = tid + 256 * i; s_row[i] = c >> 4; s_col[i] = (c & 15) * 8; }
  bf16x8 ry[4], rx[4];
  auto gload = [&](int st) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m_begin + st * WM + s_row[i];
      if (m < m_end) {
        ry[i] = ld_frag(p.dY + (size_t)m * p.ldy + n0 + s_col[i]);
        rx[i] = ld_frag(p.X + (size_t)m * p.ldx + k0 + s_col[i]);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) { ry[i][e] = f2bf(0.f); rx[i][e] = f2bf(0.f); }
      }
    }
  };
  auto swrite = [&](int buf) {
    bf16* sY = smem + buf * 2 * W_TILE; bf16* sX = sY + W_TILE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<bf16x8*>(sY + s_row[i] * W_LD + s_col[i]) = ry[i];
      *reinterpret_cast<bf16x8*>(sX + s_row[i] * W_LD + s_col[i]) = rx[i];
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  gload(0);
  swrite(0);
  __syncthreads();
  for (int st = 0; st < nst; ++st) {
    const int buf = st & 1;
    if (st + 1 < nst) gload(st + 1);
    const bf16* sY = smem + buf * 2 * W_TILE; const bf16* sX = sY + W_TILE;
#pragma unroll
    for (int ms = 0; ms < WM / 16; ++ms) {
      // both operands use the same (transposed-read) k order, so the contraction is consistent
      bf16x8 a0 = ld_frag_tr(sY, W_LD, ms * 16, wn * 64, lane), a1 = ld_frag_tr(sY, W_LD, ms * 16, wn * 64 + 32, lane);
      bf16x8 b0 = ld_frag_tr(sX, W_LD, ms * 16, wk * 64, lane), b1 = ld_frag_tr(sX, W_LD, ms * 16, wk * 64 + 32, lane);
      acc[0][0] = mfma32(a0, b0, acc[0][0]);
      acc[0][1] = mfma32(a0, b1, acc[0][1]);
      acc[1][0] = mfma32(a1, b0, acc[1][0]);
      acc[1][1] = mfma32(a1, b1, acc[1][1]);
    }
    if (st + 1 < nst) swrite(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + wn * 64 + ni * 32 + crow32(r, hi);
#pragma unroll
      for (int ki = 0; ki < 2; ++ki)
        atomicAdd(p.dW + (size_t)n * p.ldw + k0 + wk * 64 + ki * 32 + l31, acc[ni][ki][r]);
    }
}

// ---- wgrad, 192 x 384 output tile, LDS-DMA ring ----------------------------------------------------------------------
// dW[n0:+192, k0:+384] += dY[m, n]^T X[m, k] over one M-split.  8 waves (2 x 4), each 96 x 96 = 9 accumulators.  A stage
// is 64 contraction rows of both operands, row-major as they lie in HBM ([64][192] and [64][384] bf16 = 24 + 48 KB),
// brought in by global_load_lds (9 x 1 KiB per wave) into a 2-stage ring (144 KB, one block per CU); the 16-B chunks of a
// row are XOR-permuted (tr_swz, applied to the per-lane source address) so that the 4-row x 32-column blocks fetched by
// ds_read_b64_tr_b16 -- the hardware transpose that turns the m-major image into MFMA fragments -- fall on distinct banks.
// 36.9 KB staged per 64 rows of a 192x384 tile = 8.2 KB per 128x128 unit, against 16 KB for the square tile.
// Round 3: the transposing reads are inline assembly and the loop is software-pipelined by hand (tn_tall_body): through the
// builtin the compiler put `s_waitcnt vmcnt(0)` in front of the first fragment read after every LDS-DMA issue, i.e. the ring never
// prefetched (706 -> 471 us for the four gradients of a block at M = 131072; profiles/r03_wgrad_schedule.txt).
namespace tnt {
constexpr int TN = 192, TK = 384;
constexpr int PY = TN * 2, PX = TK * 2;                               // row pitches in bytes
constexpr int RM_ALIGN = 64;                                          // M-splits are cut at multiples of this (every configuration's stage divides it)
// NW waves (NW/4 x 4), RM contraction rows per ring stage, NST stages, STAG: the wave rows issue their LDS-DMA share at different
// points of the stage instead of all together behind the barrier.
template <int NW_, int RM_, int NST_, int STAG_> struct Cfg {
  static constexpr int NW = NW_, RM = RM_, NST = NST_, STAG = STAG_, THREADS = NW * 64;
  static constexpr int WN = NW / 4, WI = 6 / WN;                      // wave rows over the 192 dY columns ; 32-column MFMA tiles per wave along n
  static constexpr int Y_BYTES = RM * PY, X_BYTES = RM * PX, STAGE = Y_BYTES + X_BYTES, LDS = NST * STAGE;
  static constexpr int Y_PIECES = Y_BYTES / 1024, PIECES = STAGE / 1024;      // 1-KiB LDS-DMA pieces per stage
  static constexpr int PPW = (PIECES + NW - 1) / NW, REM = PIECES % NW;       // pieces per wave per stage: PPW (waves < REM when REM != 0) or PPW - 1
  static_assert(NST >= 2 && NST <= 4 && LDS <= 160 * 1024 && RM_ALIGN % RM == 0 && 6 % WN == 0, "configuration");
};
}

// ds_read_b64_tr_b16 is serviced 32 lanes at a time = 4 rows x 64 B of the image, over 64 banks (256 B): the four rows
// must land in four different 64-B windows.  768-B pitch (== 0 mod 256): window index ^= row & 3; 384-B pitch (rows
// alternate between offsets 0 and 128): window index ^= (row >> 1) & 1.
template <int P> DEVFN int tr_swz(int row) { return P % 256 == 0 ? (row & 3) << 2 : ((row >> 1) & 1) << 2; }
// Transposing reads as inline assembly with explicit waits (common.h: lds_tr4_at): the ring prefetches only with them -- through
// the builtin every read waited for the stage just requested (5.3 k cycles per 64-row stage for 2.3 k of MFMA).
template <int OFF, int P> DEVFN bf16x8 ld_frag_tr_at(unsigned addr) {       // rows r and r + 8 of the 16-row group: same swizzle key
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = lds_tr4_at<OFF>(addr); u.s.b = lds_tr4_at<OFF + 8 * P>(addr);
  return u.v;
}
// per-lane byte offset of a fragment's first read inside an operand image (row group 0); later row groups are immediate offsets
template <int P> DEVFN unsigned tr_frag_off(int c0, int lane) {
  const int a = lane & 15, g = lane >> 4;
  const int row = 4 * (g >> 1) + (a >> 2);
  const int col = c0 + (g & 1) * 16 + 4 * (a & 3);                 // element column; 8 elements per 16-B chunk
  const int pch = (col >> 3) ^ tr_swz<P>(row);                     // the key depends on row bits 0-1 only: unchanged by + 8, + 16 rows
  return row * P + pch * 16 + (col & 7) * 2;
}

// Waits for the inline-assembly fragment reads.  The fragments are "+v" operands so that no MFMA that consumes them moves above the wait.
template <int N> DEVFN void tn_wait_frags(bf16x8 (&f)[5]) {           // N younger reads may stay outstanding
  asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]) : "n"(N));
}
template <int N> DEVFN void tn_wait_frags(bf16x8 (&f)[6]) {
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]) : "n"(N));
}
// Stage hand-off: my LDS-DMA pieces up to the stage wanted have landed (N younger ones may be in flight), all my fragment reads are
// complete; barrier.
template <int N> DEVFN void tn_handoff(bf16x8 (&f)[5]) {
  asm volatile("s_waitcnt vmcnt(%5) lgkmcnt(0)\n\ts_barrier" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]) : "n"(N) : "memory");
}
template <int N> DEVFN void tn_handoff(bf16x8 (&f)[6]) {
  asm volatile("s_waitcnt vmcnt(%6) lgkmcnt(0)\n\ts_barrier" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]) : "n"(N) : "memory");
}

// one (tile, split) of one problem
template <class C>
DEVFN void tn_tall_body(const WgradArgs& pa, int tile, int split, char* smem_raw) {
  using namespace tnt;
  constexpr int RM = C::RM, NST = C::NST, PPW = C::PPW, WI = C::WI, STAGE = C::STAGE, Y_BYTES = C::Y_BYTES, Y_PIECES = C::Y_PIECES;
  constexpr int NMS = RM / 16, NF = WI + 3, NR = 2 * NF;            // 16-row steps per stage ; fragments / transposing reads per step
  static_assert(NMS % 2 == 0 && NR <= 15 && (NST - 1) * PPW <= 63, "fragment double buffer parity ; lgkmcnt / vmcnt field widths");
  typedef const void __attribute__((address_space(1))) * gptr_t;
  typedef void __attribute__((address_space(3))) * lptr_t;
  typedef char __attribute__((address_space(3))) * lchar_t;
  // Everything that is uniform over the wave is made visibly so (SGPRs): the wave index, and the problem's strides -- read through
  // `pa` (a kernel-argument array indexed by the tile) they were re-loaded from memory next to each use, by a VECTOR load behind a
  // lane-typed select of the field address, and the s_waitcnt for that load drained every LDS-DMA piece issued before it.
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wid >> 2, wk = wid & 3, hi = lane >> 5, l31 = lane & 31;
  const int M = __builtin_amdgcn_readfirstlane(pa.M), K = __builtin_amdgcn_readfirstlane(pa.K);
  const int ldy = __builtin_amdgcn_readfirstlane(pa.ldy), ldx = __builtin_amdgcn_readfirstlane(pa.ldx), ldw = __builtin_amdgcn_readfirstlane(pa.ldw);
  const int mps = __builtin_amdgcn_readfirstlane(pa.m_per_split);
  // Tile order inside a problem: the operand with FEWER panels varies fastest, so that neighbouring tiles -- which is where an M-split
  // gets cut between two XCDs (30 blocks per XCD, 24 tiles per split) -- differ in the cheaper operand: fc2 (2 dY panels x 4 X panels)
  // cut in the middle re-fetches its 768 B/row of dY on the second XCD instead of 3072 B/row of X (measured: -0.6 %).
  const int ntk = K / TK, ntn = __builtin_amdgcn_readfirstlane(pa.N) / TN;
  const int n0 = (ntn < ntk ? tile % ntn : tile / ntk) * TN, k0 = (ntn < ntk ? tile / ntn : tile % ntk) * TK;
  const int m_begin = split * mps;
  int m_end = m_begin + mps; if (m_end > M) m_end = M;
  if (m_begin >= m_end) return;
  const int nst = (m_end - m_begin) / RM;

  // lane -> (row, chunk) of the linear stage image.  Piece q (1 KiB) of a stage: q < Y_PIECES -> dY image, else X image;
  // wave w issues pieces w, w + NW, ...  One source pointer per piece and lane, advanced by RM rows per issued stage.  (Wave-uniform
  // bases + constant 32-bit lane offsets save registers but put a dependent v_lshl_add_u64 in front of every issue: 4-8 % slower.)
  const bool hiw = C::REM == 0 || wid < C::REM;                    // this wave issues PPW pieces per stage (else PPW - 1)
  const bf16* src[PPW]; int step[PPW], ldsoff[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int q = wid + C::NW * j < C::PIECES ? wid + C::NW * j : 0;
    if (q < Y_PIECES) {
      const int c_ = q * 64 + lane, row = c_ / 24, c = (c_ % 24) ^ tr_swz<PY>(row);
      src[j] = pa.dY + (size_t)(m_begin + row) * ldy + n0 + c * 8; step[j] = RM * ldy; ldsoff[j] = q * 1024;
    } else {
      const int c_ = (q - Y_PIECES) * 64 + lane, row = c_ / 48, c = (c_ % 48) ^ tr_swz<PX>(row);
      src[j] = pa.X + (size_t)(m_begin + row) * ldx + k0 + c * 8; step[j] = RM * ldx; ldsoff[j] = Y_BYTES + (q - Y_PIECES) * 1024;
    }
  }
  int issued = 0;                                                  // stages this wave has issued (in order)
  auto issue_stage = [&]() {
    char* buf = smem_raw + (issued % NST) * STAGE;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      if (C::REM != 0 && j == PPW - 1 && !hiw) break;
      __builtin_amdgcn_global_load_lds((gptr_t)src[j], (lptr_t)(buf + ldsoff[j]), 16, 0, 0);
      src[j] += step[j];
    }
    ++issued;
  };
  f32x16 acc[WI][3];
#pragma unroll
  for (int i = 0; i < WI; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const unsigned lds0 = (unsigned)(size_t)(lchar_t)smem_raw;
  unsigned fa[WI], fb[3];                                          // fragment read addresses inside stage 0
#pragma unroll
  for (int i = 0; i < WI; ++i) fa[i] = lds0 + tr_frag_off<PY>(wn * (WI * 32) + i * 32, lane);
#pragma unroll
  for (int j = 0; j < 3; ++j) fb[j] = lds0 + Y_BYTES + tr_frag_off<PX>(wk * 96 + j * 32, lane);

  // Software pipeline over 16-row steps.  The fragments of step t + 1 are requested before the MFMAs of step t (two register
  // sets); the reads are inline assembly, so every wait is written here and the fragments are tied to it ("+v") to keep the MFMAs
  // behind it.  The hand-off of a stage sits in the LAST step of the stage before it: wait for my pieces of stage st + 1 (vmcnt; the
  // younger stages stay in flight) and for my outstanding fragment reads (all of stage st), barrier -- now stage st + 1 is complete
  // and nobody reads stage st any more, so its buffer takes stage st + NST at once, and step 0 of stage st + 1 is requested under
  // the last MFMAs of stage st.
  // STAG: the wave rows issue their pieces at different points, so that the LDS-DMA issue of one wave per SIMD (which blocks that
  // wave for as long as the memory pipeline is backed up: ~1-2 k cycles per stage) overlaps the MFMAs of the other: row 0 right behind
  // the hand-off barrier, row 1 in front of the MFMAs of step 0 of the following stage -- before the next hand-off, so the vmcnt counts
  // there are the same for all.  (Measured, round-robin medians at M = 131072: 471 us ; un-staggered 487 ; 32-row stages in a 4-deep
  // ring 498, with the issue spread between the MFMAs 485 ; 12 waves of 64 x 96 spill at 168 registers: profiles/r03_wgrad_schedule.txt.)
  bf16x8 fr[2][NF];
  auto rd = [&](auto ms_tag, auto set_tag, unsigned so) {
    constexpr int MS = decltype(ms_tag)::value, SET = decltype(set_tag)::value;
#pragma unroll
    for (int i = 0; i < WI; ++i) fr[SET][i] = ld_frag_tr_at<MS * 16 * PY, PY>(fa[i] + so);
#pragma unroll
    for (int j = 0; j < 3; ++j) fr[SET][WI + j] = ld_frag_tr_at<MS * 16 * PX, PX>(fb[j] + so);
  };
  auto mm = [&](auto set_tag) {
    constexpr int SET = decltype(set_tag)::value;
#pragma unroll
    for (int i = 0; i < WI; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[i][j] = mfma32(fr[SET][i], fr[SET][WI + j], acc[i][j]);
    // MFMAs are ordinary instructions to the scheduler: without this it floats them across the (volatile) reads and waits of the
    // following steps -- e.g. two steps' MFMAs merged behind the third wait -- and the overlap becomes an accident of code shape
    __builtin_amdgcn_sched_barrier(0);
  };
  auto wait_frags = [&](auto set_tag, auto n_tag) { tn_wait_frags<decltype(n_tag)::value>(fr[decltype(set_tag)::value]); };
  auto handoff = [&](auto set_tag, auto n_tag) { tn_handoff<decltype(n_tag)::value>(fr[decltype(set_tag)::value]); };
  using I0 = std::integral_constant<int, 0>;

#pragma unroll
  for (int s_ = 0; s_ < NST; ++s_)
    if (s_ < nst) issue_stage();
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");     // once per block: everything issued so far has landed
  rd(I0{}, I0{}, 0u);
  for (int st = 0; st < nst; ++st) {
    const unsigned so = (unsigned)(st % NST) * STAGE, so_next = (unsigned)((st + 1) % NST) * STAGE;
    const int want = st + NST < nst ? st + NST : nst;              // stages that may have been issued while stage st is being read
    static_for<NMS>([&](auto ms_tag) {
      constexpr int MS = decltype(ms_tag)::value;
      using CUR = std::integral_constant<int, MS & 1>; using NXT = std::integral_constant<int, (MS & 1) ^ 1>;
      if constexpr (MS < NMS - 1) {
        rd(std::integral_constant<int, MS + 1>{}, NXT{}, so);
        if (C::STAG && MS == 0 && wn >= 1 && issued < want) issue_stage();
        wait_frags(CUR{}, std::integral_constant<int, NR>{});
        mm(CUR{});
      } else {
        const int want1 = st + 1 + NST < nst ? st + 1 + NST : nst; // ... once stage st's buffer is free
        if (st + 1 < nst) {
          const int younger = issued - (st + 2);                   // every wave row has issued `want` stages by now
          if (NST >= 4 && younger >= 2) { if (hiw) handoff(CUR{}, std::integral_constant<int, 2 * PPW>{}); else handoff(CUR{}, std::integral_constant<int, 2 * (PPW - 1)>{}); }
          else if (NST >= 3 && younger == 1) { if (hiw) handoff(CUR{}, std::integral_constant<int, PPW>{}); else handoff(CUR{}, std::integral_constant<int, PPW - 1>{}); }
          else handoff(CUR{}, I0{});
          if ((!C::STAG || wn == 0) && issued < want1) issue_stage();
          rd(I0{}, NXT{}, so_next);
        } else {
          wait_frags(CUR{}, I0{});
        }
        mm(CUR{});
      }
    });
  }
#pragma unroll
  for (int i = 0; i < WI; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + wn * (WI * 32) + i * 32 + crow32(r, hi);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        atomicAdd(pa.dW + (size_t)n * ldw + k0 + wk * 96 + j * 32 + l31, acc[i][j][r]);
      }
    }
}

template <class C>
__global__ __launch_bounds__(C::THREADS, C::NW / 4) void gemm_tn_tall_kernel(WgradArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int ntiles = (p.N / tnt::TN) * (p.K / tnt::TK);
  const int id = xcd_remap(blockIdx.x, gridDim.x);                // all tiles of one M-split on one XCD (they stream the same rows)
  tn_tall_body<C>(p, id % ntiles, id / ntiles, smem_raw);
}

// Several weight gradients in ONE launch (the four of a transformer block): the grid is one round of the chip however
// many problems share it, so each problem needs 1/n-th of the M-splits it would need alone -- and the fp32 atomics that
// combine the splits (measured ~1.5 TB/s, 45-50 us per GEMM when each is launched alone) shrink by the same factor.
// Block -> (problem, tile, M-split) map (round 4).  Blocks run on XCD blockIdx % 8 and xcd_remap gives every XCD a contiguous chunk of
// logical ids; the tiles of ONE problem inside ONE M-split stream the same rows of dY / X (fc1: 8 tiles share the 768 B/row of h2, ...),
// so a chunk boundary that falls inside such a group makes the second L2 fetch the shared panel again: with 24 tiles x 10 splits = 30
// blocks per XCD every split straddled two L2s and the launch fetched 1.09x its algorithmic bytes (PMC, round 3).  The host packs whole
// groups into the per-XCD chunks instead (first fit, largest first; a group is cut only when nothing else fits) and hands the kernel the
// resulting table: 10 x {8, 8, 6, 2} blocks pack into 8 chunks of 30 without a single cut.
constexpr int WG_MAP_MAX = 1024;
struct WgradGroup {
  WgradArgs it[ATST_WGRAD_GROUP_MAX]; int first_tile[ATST_WGRAD_GROUP_MAX + 1]; int n;
  int use_map;                                  // 0: logical id = tile + ntiles * split (round-3 order)
  uint16_t map[WG_MAP_MAX];                     // [logical id] = problem << 14 | tile in problem << 8 | split   (<= 4 problems, <= 64 tiles, <= 256 splits)
};
template <class C>
__global__ __launch_bounds__(C::THREADS, C::NW / 4) void gemm_tn_tall_group_kernel(WgradGroup g) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int ntiles = g.first_tile[g.n];
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  if (g.use_map) {
    const int e = __builtin_amdgcn_readfirstlane((int)g.map[id]);
    tn_tall_body<C>(g.it[e >> 14], (e >> 8) & 63, e & 255, smem_raw);
    return;
  }
  const int t = id % ntiles, split = id / ntiles;
  int k = 0;
#pragma unroll
  for (int i = 1; i < ATST_WGRAD_GROUP_MAX; ++i) if (i < g.n && t >= g.first_tile[i]) k = i;
  tn_tall_body<C>(g.it[k], t - g.first_tile[k], split, smem_raw);
}
// host side of the map: groups = (problem, split) with tiles(problem) members each; bins = the chunks xcd_remap gives the 8 XCDs
static bool build_wgrad_map(WgradGroup& g, int splits, int nblk) {
  const int ntiles = g.first_tile[g.n];
  if (nblk != ntiles * splits || nblk > WG_MAP_MAX || splits > 256) return false;
  int gsz[ATST_WGRAD_GROUP_MAX];
  for (int k = 0; k < g.n; ++k) { gsz[k] = g.first_tile[k + 1] - g.first_tile[k]; if (gsz[k] > 64) return false; }
  // remaining members of every group: left[k][split] tiles not yet placed (placed from tile 0 upwards)
  static thread_local int left[ATST_WGRAD_GROUP_MAX][256];
  for (int k = 0; k < g.n; ++k) for (int sp = 0; sp < splits; ++sp) left[k][sp] = gsz[k];
  const int q = nblk >> 3, r = nblk & 7;
  int pos = 0;
  for (int x = 0; x < 8; ++x) {
    int cap = q + (x < r ? 1 : 0);
    while (cap > 0) {
      // largest untouched group that fits; else the largest remainder (cut it)
      int bk = -1, bs = -1, best = 0; bool fits = false;
      for (int k = 0; k < g.n; ++k)
        for (int sp = 0; sp < splits; ++sp) {
          const int l = left[k][sp];
          if (l == 0) continue;
          const bool f = l <= cap;
          if ((f && !fits) || (f == fits && l > best)) { bk = k; bs = sp; best = l; fits = f; }
        }
      if (bk < 0) return false;
      const int take = best <= cap ? best : cap, t0 = gsz[bk] - left[bk][bs];
      for (int t = 0; t < take; ++t) g.map[pos++] = (uint16_t)((bk << 14) | ((t0 + t) << 8) | bs);
      left[bk][bs] -= take; cap -= take;
    }
  }
  return pos == nblk;
}

int g_tn_map = 1;           // 130/131: grouped wgrad: (problem, split) groups packed onto XCDs (131, default) / round-3 order tile + ntiles * split (130)
int g_tn_cfg = 1;           // 120 + c: wgrad schedule: 0 = every wave issues behind the hand-off ; 1 = wave rows staggered (default) ; 2 = 32-row stages, 4-deep ring
template <class C> int launch_tn_tall(const WgradArgs& p, int nblk, hipStream_t st) {
  static OncePerDevice attr; int attr_dev;
  if (attr.need(attr_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_tall_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    if (e != hipSuccess) return (int)e;
    attr.done(attr_dev);
  }
  hipLaunchKernelGGL(gemm_tn_tall_kernel<C>, dim3(nblk), dim3(C::THREADS), C::LDS, st, p);
  return (int)hipGetLastError();
}
template <class C> int launch_tn_tall_group(const WgradGroup& g, int nblk, hipStream_t st) {
  static OncePerDevice attr; int attr_dev;
  if (attr.need(attr_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_tall_group_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    if (e != hipSuccess) return (int)e;
    attr.done(attr_dev);
  }
  hipLaunchKernelGGL(gemm_tn_tall_group_kernel<C>, dim3(nblk), dim3(C::THREADS), C::LDS, st, g);
  return (int)hipGetLastError();
}
#define TN_CFG_DISPATCH(FN, ...)                                              \
  switch (g_tn_cfg) {                                                         \
    case 0: return FN<tnt::Cfg<8, 64, 2, 0>>(__VA_ARGS__);                    \
    case 2: return FN<tnt::Cfg<8, 32, 4, 1>>(__VA_ARGS__);                    \
    default: return FN<tnt::Cfg<8, 64, 2, 1>>(__VA_ARGS__);                   \
  }
int dispatch_tn_tall(const WgradArgs& p, int nblk, hipStream_t st) { TN_CFG_DISPATCH(launch_tn_tall, p, nblk, st) }
int dispatch_tn_tall_group(const WgradGroup& g, int nblk, hipStream_t st) { TN_CFG_DISPATCH(launch_tn_tall_group, g, nblk, st) }

// Tuning hooks (atst_tune_gemm_variant, include/atst_hip.h); the defaults are the measured best.
int g_nt_variant = -1;      // -1 auto ; 0: 128x128 2-stage ; 1: 128x128 3-stage ; 3: 256x128 4 waves of 128x64 ; 4: force the row-384 tile
int g_tn_tall = 1;          // 105/106: wgrad 192 x 384 LDS-DMA tile when N % 192 == 0, K % 384 == 0, M % 64 == 0
int g_tn_rounds = 1;        // 110 + r: 128x128 wgrad grid = r rounds of 512 resident blocks; 1 measured best (-20 %)
int g_row384_auto = 1;      // 300/301: row-384 tile whenever N % 384 == 0
int g_row384_tall = 2;      // 302/303/304: 256 x 384 tiles for M >= 8192: never / plain bf16 GEMMs only / every epilogue
int g_dgelu_row384 = 2;     // 306/307/308: dGELU GEMM on the 256x384 tile never / always / when K >= 768.  With the register column sums (tools/dgelu_probe.py): base (K = 768) 991 -> 889 us, small (K = 384) 314-342 -> 339 us
int g_w4_mode = 0;          // 330 + m: 4-wave two-blocks-per-CU kernels: 0 only for small grids ; 2 = 256x192 (plain epilogues) + 128x384 (row-wise) everywhere ; 3 = 256x192 for the plain epilogues only
int g_w4_min_m = 8192;      // 350/351: apply the tall / 4-wave kernels from M = 8192 / from any M (parity tests run small shapes)
int g_w4_auto = 1;          // 360/361: 4-wave kernels for launches of <= 1.5 rounds of 256 x 384 tiles
int g_f8_w4 = 0;            // 2120/2121: ... for e4m3 operands too (round 6: N = 384 / 1152 GEMMs of the 1 s local views in the e4m3 step at d = 384).  Measured -0.5 % in the
                            // step (6337-6350 -> 6306-6319 clips/s, same box, alternating): the local groups run beside the teacher / global-view chain on the second stream, which
                            // already fills the CUs their 104-block grids leave idle -- off by default
int g_ph = 1;               // 396/397/398: phased main loop (template parameter PH) of the 256 x 384 tile: off / on (32-deep, three 40-KB slots) / 64-deep whole-line stages (two 80-KB slots, round 6)
int g_p8_skew = 0;          // 1000 + c: start-up skew of every other first-round block of the phased kernel, c x 1024 cycles (experiment)
int g_p8 = 3;               // 390/391/392/393: 256 x 256 phased kernel (gemm_p8.h) for N % 256 == 0, K % 128 == 0, M % 256 == 0, M >= 8192: off / bf16 operands only / also e4m3
                            // except fc1 + GELU / every e4m3 GEMM (default since the lean outputs: fc1 + GELU 606 -> 592 us, base fp8 +0.8 ... 1.8 %)
                            // operands (default; except fc1 + GELU, whose e4m3 epilogue -- u, a, the e4m3 copy of a, amax -- measured 640 vs 647 us on the 256 x 384 tile).
                            // Same box, same call, base fp8 step: 2405-2407 (391) -> 2453-2476 clips/s (392), profiles/r05_p8_fp8_step_ab.txt
int g_tt = 0;               // 2000 + v: persistent overlap kernels (gemm_tt.h): 0 off / 1 two teams of four waves / 2 one stream, 4 waves x 512 registers -- for the plain bf16 + fc1-GELU launches they take
int g_f32_splitk = 1;       // 380/381: split-K for fp32-output GEMMs with <= 64 output tiles and K >= 2048
// Split-K workspace: per (device, stream) -- kernels of one stream run in order, so one buffer per stream is race-free; the null stream is the same
// handle on every device, hence the device in the key (round-4 ADVICE).  Sized on first use for the largest head shape of the path
// (24 splits x 2048 rows x 4096 columns fp32 = 768 MiB would be the ATST-Frame extreme; the clip heads need <= 16 MB), grown on demand (a
// stream synchronise + hipFree: once, in warm-up), never shrunk, never freed: the ONE piece of device memory the library owns.
struct SplitKWs { float* ws = nullptr; size_t cap = 0; };
struct SplitKKey { int dev; hipStream_t st; bool operator==(const SplitKKey& o) const { return dev == o.dev && st == o.st; } };
struct SplitKHash { size_t operator()(const SplitKKey& k) const { return std::hash<const void*>()((const void*)k.st) ^ ((size_t)k.dev * 0x9E3779B97F4A7C15ull); } };
int splitk_workspace(hipStream_t st, size_t floats, float** ws) {
  static std::mutex mu;
  static std::unordered_map<SplitKKey, SplitKWs, SplitKHash> table;
  int dev = 0;
  { hipError_t e = hipGetDevice(&dev); if (e != hipSuccess) return (int)e; }
  std::lock_guard<std::mutex> lk(mu);
  SplitKWs& w = table[SplitKKey{dev, st}];
  if (floats > w.cap) {
    if (w.ws) { hipError_t e = hipStreamSynchronize(st); if (e != hipSuccess) return (int)e; (void)hipFree(w.ws); w.ws = nullptr; w.cap = 0; }
    size_t want = floats + floats / 4;
    if (want < (size_t)4 << 20) want = (size_t)4 << 20;           // 16 MB floor: covers every clip-head shape of the bench from the first call
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&w.ws), want * sizeof(float));
    if (e != hipSuccess) return (int)e;
    w.cap = want;
  }
  *ws = w.ws;
  return ATST_OK;
}
int g_bf16_tr = 0;          // 370/371: store-only bf16 epilogue from transposed accumulators (wave-private staging, no block barrier): measured 1-10 % slower

// Algorithmic HBM bytes of one nt GEMM: both operands once, every epilogue input once, every output once.
template <int EPI>
double nt_bytes(const GemmArgs& a) {
  const double mn = (double)a.M * a.N;
  double b = 2.0 * a.K * ((double)a.M + a.N);
  if (EPI == EPI_BF16) b += 2.0 * mn;
  if (EPI == EPI_F32) b += 4.0 * mn;
  if (EPI == EPI_BIAS_GELU) b += (a.C2 ? 2.0 * mn : 0.0) + (a.C ? 2.0 * mn : 0.0) + (a.q8 ? mn : 0.0);   // a (bf16 and / or e4m3), u only when it is saved (training)
  if (EPI == EPI_RESID) b += (a.resid_bf16 ? 2.0 : 4.0) * mn + (a.out_bf16 ? 2.0 : 4.0) * mn + (a.ln_out ? 2.0 * mn : 0.0) + (a.q8 ? mn : 0.0);
  if (EPI == EPI_DGELU) b += 2.0 * mn + (a.C ? 2.0 * mn : 0.0) + (a.q8 ? mn : 0.0);      // u in ; du out (bf16 and / or e4m3)
  if (EPI == EPI_PATCH) b += 4.0 * mn;
  if (EPI == EPI_LNBWD) b += 8.0 * mn + (a.resid ? 4.0 * mn : 0.0) + (a.lnb_g ? 2.0 * mn : 0.0) + (a.q8 ? mn : 0.0);   // x in, dx out, dres in, g out (bf16 and / or e4m3)
  return b;
}
// EPI_RESID whose epilogue is also the LayerNorm of the new row: a bf16 LayerNorm output, or (e4m3 step at d = 384, lean) only its e4m3 copy
inline bool ln_fused(const GemmArgs& a) { return a.ln_out || (a.fp8 && a.q8 && a.ln_gamma && a.epi == EPI_RESID); }
template <int EPI> constexpr int prof_kind() { return EPI == EPI_LNBWD ? PK_GEMM_NT6 : PK_GEMM_NT0 + EPI; }

template <int EPI, int BMT, int NSTG, int WTM, bool KS = false>
int launch_nt_cfg(const GemmArgs& a, hipStream_t st) {
  using G = NtGeo<BMT, NSTG, WTM>;
  static OncePerDevice attr_done; int attr_done_dev;
  if (attr_done.need(attr_done_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_kernel<EPI, BMT, NSTG, WTM, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    if (e != hipSuccess) return (int)e;
    attr_done.done(attr_done_dev);
  }
  const int nblk = ((a.M + BMT - 1) / BMT) * (a.N / BN) * (KS ? a.ksplit : 1);
  ProfScope ps(prof_kind<EPI>(), 2.0 * a.M * a.N * a.K, st, nt_bytes<EPI>(a));
  hipLaunchKernelGGL((gemm_nt_kernel<EPI, BMT, NSTG, WTM, KS>), dim3(nblk), dim3(G::THREADS), G::LDS, st, a);
  return (int)hipGetLastError();
}
template <int EPI, int MI, bool LN, bool F8 = false, bool TR = false, int PH = 0, bool Q8 = false>
int launch_nt_row384_cfg(const GemmArgs& a, hipStream_t st) {
  using RG = row384::Geo<MI>;
  constexpr int LDS = PH == 2 ? 163840 : TR ? RG::NSTG * RG::STAGE + row384::BNR * 4 : row384::lds_bytes<MI, EPI, LN>();   // PH == 2: two 80-KB slots
  static OncePerDevice attr_done; int attr_done_dev;
  if (attr_done.need(attr_done_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_row384_kernel<EPI, MI, LN, F8, TR, PH, Q8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    attr_done.done(attr_done_dev);
  }
  const int nblk = ((a.M + RG::BMR - 1) / RG::BMR) * (a.N / row384::BNR);
  hipLaunchKernelGGL((gemm_nt_row384_kernel<EPI, MI, LN, F8, TR, PH, Q8>), dim3(nblk), dim3(row384::THREADS), LDS, st, a);
  return (int)hipGetLastError();
}
template <int EPI, int WM, bool LN, bool Q8 = false, bool F8 = false>
int launch_nt_w4_cfg(const GemmArgs& a, hipStream_t st) {
  using G = w4::Geo<WM>;
  constexpr int LDS = G::template lds_bytes<EPI, LN>();
  static_assert(LDS <= 81920, "two blocks per CU");
  static OncePerDevice attr_done; int attr_done_dev;
  if (attr_done.need(attr_done_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_w4_kernel<EPI, WM, LN, Q8, F8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    attr_done.done(attr_done_dev);
  }
  const int nblk = ((a.M + G::BM - 1) / G::BM) * (a.N / G::BNB);
  hipLaunchKernelGGL((gemm_nt_w4_kernel<EPI, WM, LN, Q8, F8>), dim3(nblk), dim3(256), LDS, st, a);
  return (int)hipGetLastError();
}
template <int EPI, bool F8 = false>
int launch_nt_p8(const GemmArgs& a, hipStream_t st) {
  static OncePerDevice attr_done; int attr_done_dev;
  constexpr int lds = p8::RING;
  if (attr_done.need(attr_done_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_p8_kernel<EPI, F8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr_done.done(attr_done_dev);
  }
  ProfScope ps(prof_kind<EPI>(), (F8 ? 4.0 : 2.0) * a.M * a.N * a.K, st, nt_bytes<EPI>(a));      // F8: K counts byte pairs here
  GemmArgs b = a; b.skew = g_p8_skew;
  hipLaunchKernelGGL((gemm_nt_p8_kernel<EPI, F8>), dim3((a.M / p8::BM) * (a.N / p8::BNP)), dim3(p8::THREADS), lds, st, b);
  return (int)hipGetLastError();
}
// shapes the phased kernel takes (a.K, a.lda, a.ldb in bf16 elements -- for e4m3 operands the launcher has already halved them)
template <int EPI>
bool p8_ok(const GemmArgs& a) {
  constexpr bool epi_ok = EPI == EPI_BF16 || EPI == EPI_F32 || EPI == EPI_BIAS_GELU || EPI == EPI_RESID || EPI == EPI_DGELU;
  return epi_ok && g_p8 && !ln_fused(a) && (a.fp8 || !a.q8) && a.N % p8::BNP == 0 && a.K % (2 * p8::BKP) == 0 && a.M % p8::BM == 0 &&
         a.M >= (g_w4_min_m < 8192 ? 256 : 8192) && a.lda % 8 == 0 && a.ldb % 8 == 0;
}
// two-team persistent kernel (gemm_tt.h): one block per CU, the epilogue of a tile under the main loop of the next
template <int EPI>
bool tt_ok(const GemmArgs& a) {
  constexpr bool epi_ok = EPI == EPI_BF16 || EPI == EPI_BIAS_GELU;
  return epi_ok && !a.fp8 && !a.q8 && !a.ln_out && a.M % tt::BM == 0 && a.N % tt::BNT == 0 && a.K % tt::BKT == 0 && a.K >= tt::BKT * tt::MIN_STAGES &&
         a.M >= (g_w4_min_m < 8192 ? 256 : 8192) && a.lda % 8 == 0 && a.ldb % 8 == 0 && (EPI != EPI_BIAS_GELU || a.C2);
}
int tt_num_cus() {
  static int cus[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
  if (!cus[dev]) { int n = 0; if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256; cus[dev] = n; }
  return cus[dev];
}
template <int EPI, bool SU>
int launch_nt_tt_cfg(const GemmArgs& a, hipStream_t st) {
  static OncePerDevice attr_done; int dev;
  if (attr_done.need(dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_tt_kernel<EPI, SU>, hipFuncAttributeMaxDynamicSharedMemorySize, tt::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_done.done(dev);
  }
  const int T = (a.M / tt::BM) * (a.N / tt::BNT);
  ProfScope ps(prof_kind<EPI>(), 2.0 * a.M * a.N * a.K, st, nt_bytes<EPI>(a));
  hipLaunchKernelGGL((gemm_nt_tt_kernel<EPI, SU>), dim3(tt_num_cus()), dim3(tt::THREADS), tt::LDS_BYTES, st, a, T);
  return (int)hipGetLastError();
}
template <int EPI, bool SU>
int launch_nt_t1_cfg(const GemmArgs& a, hipStream_t st) {         // the same overlap as ONE instruction stream: 4 waves, one per SIMD, 512 registers (hook 2002)
  static OncePerDevice attr_done; int dev;
  if (attr_done.need(dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_t1_kernel<EPI, SU>, hipFuncAttributeMaxDynamicSharedMemorySize, tt::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_done.done(dev);
  }
  const int T = (a.M / tt::BM) * (a.N / tt::BNT);
  ProfScope ps(prof_kind<EPI>(), 2.0 * a.M * a.N * a.K, st, nt_bytes<EPI>(a));
  hipLaunchKernelGGL((gemm_nt_t1_kernel<EPI, SU>), dim3(tt_num_cus()), dim3(256), tt::LDS_BYTES, st, a, T);
  return (int)hipGetLastError();
}
template <int EPI>
int launch_nt_tt(const GemmArgs& a, hipStream_t st) {
  if (g_tt == 2) {
    if constexpr (EPI == EPI_BIAS_GELU) { if (!a.C) return launch_nt_t1_cfg<EPI, false>(a, st); }
    if constexpr (EPI == EPI_BF16 || EPI == EPI_BIAS_GELU) return launch_nt_t1_cfg<EPI, true>(a, st);
  }
  if constexpr (EPI == EPI_BIAS_GELU) { if (!a.C) return launch_nt_tt_cfg<EPI, false>(a, st); }
  if constexpr (EPI == EPI_BF16 || EPI == EPI_BIAS_GELU) return launch_nt_tt_cfg<EPI, true>(a, st);
  return ATST_EINVAL;
}
template <int EPI>
int launch_nt_row384(const GemmArgs& a, hipStream_t st) {
  if (a.fp8) {                                                    // e4m3 operands seen as byte pairs: K, lda, ldb are already halved
    // small grids (the 1 s local views: M = 26624 -> 104 tiles of 256 rows per 384 columns): the 4-wave kernels, two blocks per CU, as for bf16 operands below
    const bool small8 = g_f8_w4 && g_w4_auto && a.M >= 2048 && (long)((a.M + 255) / 256) * (a.N / 384) <= 384;
    if constexpr (EPI == EPI_LNBWD) {                             // round 6: the LayerNorm backward in the epilogue of an e4m3 dgrad GEMM (d = 384)
      ProfScope ps(prof_kind<EPI>(), 4.0 * a.M * a.N * a.K, st, nt_bytes<EPI>(a));
      if (small8) return launch_nt_w4_cfg<EPI, 1, false, true, true>(a, st);
      return launch_nt_row384_cfg<EPI, 4, false, true, false, 0, true>(a, st);
    } else
    if constexpr (EPI == EPI_BF16 || EPI == EPI_BIAS_GELU || EPI == EPI_RESID || EPI == EPI_F32 || EPI == EPI_DGELU) {
      if constexpr (EPI == EPI_RESID) {
        if (ln_fused(a)) {                                        // ... and the LayerNorm forward of the new residual row
          ProfScope ps(prof_kind<EPI>(), 4.0 * a.M * a.N * a.K, st, nt_bytes<EPI>(a));
          if (small8) return launch_nt_w4_cfg<EPI, 1, true, true, true>(a, st);
          return launch_nt_row384_cfg<EPI, 4, true, true, false, 0, true>(a, st);
        }
      }
      if constexpr (EPI == EPI_BF16) {
        if (small8 && !a.q8) {
          ProfScope ps(prof_kind<EPI>(), 4.0 * a.M * a.N * a.K, st, nt_bytes<EPI>(a));
          return launch_nt_w4_cfg<EPI, 2, false, false, true>(a, st);
        }
      }
      if (g_nt_variant < 0 && g_p8 >= 2 && (EPI != EPI_BIAS_GELU || g_p8 >= 3) && p8_ok<EPI>(a)) return launch_nt_p8<EPI, true>(a, st);
      ProfScope ps(prof_kind<EPI>(), 4.0 * a.M * a.N * a.K, st, nt_bytes<EPI>(a));
      if constexpr (EPI == EPI_BF16) { if (g_bf16_tr) return launch_nt_row384_cfg<EPI, 4, false, true, true>(a, st); }
      return launch_nt_row384_cfg<EPI, 4, false, true>(a, st);
    } else {
      return ATST_EINVAL;
    }
  }
  ProfScope ps(prof_kind<EPI>(), 2.0 * a.M * a.N * a.K, st, nt_bytes<EPI>(a));
  constexpr bool rowwise_only = EPI == EPI_LNBWD;                 // epilogues that exist only on whole-row tiles
  if constexpr (EPI == EPI_LNBWD) {
    // bf16 operands, but the epilogue also writes the e4m3 copy of g / posts its amax (e4m3 step at d = 384, groups whose attention backward writes a bf16 dqkv:
    // the 1 s local views): the Q8 instantiations -- 4-wave blocks for the small grids, like the plain ones below
    if (a.q8 || a.q8_amax) {
      const long tiles8q = (long)((a.M + 255) / 256);
      if (g_w4_auto && a.M >= 2048 && tiles8q <= 384) return launch_nt_w4_cfg<EPI, 1, false, true>(a, st);
      return launch_nt_row384_cfg<EPI, 4, false, false, false, 0, true>(a, st);
    }
  }

  // Fewer than 1.5 rounds of 256 x 384 tiles (the 1 s local views: M = 32768, N = 384 -> 128 blocks on 256 CUs): the 4-wave
  // kernels launch twice as many, half as large blocks, two per CU.  Measured at M = 32768 (profiles/r02_gemm_bench.txt):
  // proj+resid 49 -> 36 us, fc2+resid 96 -> 75, fc1 dgrad 67 -> 53, qkv dgrad 48 -> 39, qkv fwd 47 -> 43; at M = 131072 the
  // 8-wave tile is 10-25 % faster (B is staged once per 256 rows), and fc1+GELU (N = 1536) is never better on 4 waves.
  const long tiles8 = (long)((a.M + 255) / 256) * (a.N / 384);
  const bool small_grid = g_w4_auto && a.M >= 2048 && tiles8 <= 384 && EPI != EPI_BIAS_GELU && EPI != EPI_DGELU && EPI != EPI_PATCH;
  if ((g_w4_mode && a.M >= g_w4_min_m) || small_grid) {
    const int w4m = g_w4_mode ? g_w4_mode : 2;
    if constexpr (EPI == EPI_RESID) {
      if (ln_fused(a)) { if (w4m != 3) return launch_nt_w4_cfg<EPI, 1, true>(a, st); }
    }
    if constexpr (rowwise_only) {
      if (w4m != 3) return launch_nt_w4_cfg<EPI, 1, false>(a, st);
    } else {
      if (!ln_fused(a)) return launch_nt_w4_cfg<EPI, 2, false>(a, st);
    }
  }
  // 256-row tiles: the operand ring of one block covers twice the output (6.7 vs 10.7 KB staged per 128x128 unit)
  const bool tall = a.M >= (g_w4_min_m < 8192 ? g_w4_min_m : 8192) && (g_row384_tall == 2 || (g_row384_tall == 1 && EPI == EPI_BF16 && (a.K >= 768 || a.N >= 768)));
  const int ph = (g_ph && tall && a.K >= 2 * BK) ? ((g_ph == 2 && a.K % 64 == 0 && a.K >= 128) ? 2 : 1) : 0;   // phased main loop of the 256-row tile (2: 64-deep whole-line stages, hook 398)
  if constexpr (EPI == EPI_RESID) {
    if (ln_fused(a)) return tall ? (ph == 2 ? launch_nt_row384_cfg<EPI, 4, true, false, false, 2>(a, st) : ph ? launch_nt_row384_cfg<EPI, 4, true, false, false, 1>(a, st) : launch_nt_row384_cfg<EPI, 4, true>(a, st)) : launch_nt_row384_cfg<EPI, 2, true>(a, st);
  }
  if constexpr (EPI == EPI_BF16) {
    if (g_bf16_tr) return tall ? launch_nt_row384_cfg<EPI, 4, false, false, true>(a, st) : launch_nt_row384_cfg<EPI, 2, false, false, true>(a, st);
  }
  return tall ? (ph == 2 ? launch_nt_row384_cfg<EPI, 4, false, false, false, 2>(a, st) : ph ? launch_nt_row384_cfg<EPI, 4, false, false, false, 1>(a, st) : launch_nt_row384_cfg<EPI, 4, false>(a, st)) : launch_nt_row384_cfg<EPI, 2, false>(a, st);
}
template <int EPI>
int launch_nt(const GemmArgs& a, hipStream_t st) {
  if constexpr (EPI == EPI_LNBWD) {                               // LayerNorm backward in the dgrad epilogue: whole rows only
    if (a.N != 384 || a.ldc != 384 || !a.ln_gamma || !a.ln_mean || !a.ln_rstd || !a.lnb_x || !a.lnb_dgamma || !a.lnb_dbeta || !a.C) return ATST_EINVAL;
    return launch_nt_row384<EPI>(a, st);
  } else {
    if (ln_fused(a)) {                                            // fused LayerNorm needs the block to own whole rows
      if (EPI != EPI_RESID || a.N != 384 || a.ldc != 384 || !a.ln_gamma || !a.ln_beta || !a.ln_mean || !a.ln_rstd) return ATST_EINVAL;
      if constexpr (EPI == EPI_RESID) return launch_nt_row384<EPI>(a, st);
    }
    int v = g_nt_variant;
    if constexpr (EPI == EPI_BF16 || EPI == EPI_BIAS_GELU) {
      if (v < 0 && g_tt && tt_ok<EPI>(a)) return launch_nt_tt<EPI>(a, st);
    }
    if constexpr (EPI != EPI_PATCH && EPI != EPI_LNBWD) {
      if (v < 0 && !a.fp8 && p8_ok<EPI>(a)) return launch_nt_p8<EPI>(a, st);
    }
    if constexpr (EPI == EPI_F32) {
      // A handful of output tiles with a long K (the head Linears: 1536 x 256 x 12288 = 24 tiles of 384 k-tiles, 103 us on 24 CUs): split K
      // over the idle CUs; partial tiles go to a library-owned workspace and splitk_reduce_kernel sums them in a fixed order.  The split
      // count is a function of K ALONE (16 k-tiles per split, at most 24 splits): the summation order of an output element must not depend on
      // how many rows the launch has -- 8 ranks with 4 clips each and 1 rank with 32 clips have to produce the same head outputs bit for bit,
      // or ReLU gates of near-zero pre-activations differ between them (tests/test_dist_gpu.py).
      const int tiles = ((a.M + 127) / 128) * (a.N / BN), nk = a.K / BK;
      if (g_f32_splitk && v < 0 && tiles <= 64 && nk >= 64) {
        const int ks = nk / 16 < 24 ? nk / 16 : 24;
        GemmArgs b = a; b.ksplit = ks;
        const int rc = splitk_workspace(st, (size_t)ks * a.M * a.N, &b.ks_ws);
        if (rc != ATST_OK) return rc;
        const int rc2 = launch_nt_cfg<EPI, 128, 3, 64, true>(b, st);
        if (rc2 != ATST_OK) return rc2;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(((size_t)a.M * (a.N / 4) + 255) / 256)), dim3(256), 0, st, b.ks_ws, ks, a.M, a.N, a.bias,
                           reinterpret_cast<float*>(a.C), a.ldc);
        return (int)hipGetLastError();
      }
    }
    if ((v == 4 || (v < 0 && g_row384_auto)) && a.N % 384 == 0 && (EPI != EPI_DGELU || g_dgelu_row384 == 1 || (g_dgelu_row384 == 2 && a.K >= 768))) return launch_nt_row384<EPI>(a, st);
    if (v < 0) v = (EPI == EPI_F32 && a.M >= 16384) ? 3          // ATST-Frame head Linears (83 k rows): 256x128 tile, -13 %
                 : a.K <= 512 ? 0 : 1;
    if (v == 3) return launch_nt_cfg<EPI, 256, 3, 128>(a, st);
    if (v == 1) return launch_nt_cfg<EPI, 128, 3, 64>(a, st);
    return launch_nt_cfg<EPI, 128, 2, 64>(a, st);
  }
}

}  // namespace

int g_f8_resid16 = 1;        // tuning hook 2100 / 2101: fp8 inference / teacher passes keep their residual stream in bf16 (read by engine.hip): off / on
int g_f8_fuse_ln = 1;        // tuning hook 2110 / 2111: the e4m3 step at d = 384 runs its LayerNorms (forward and backward) inside the GEMM epilogues (read by engine.hip): off / on
int g_tn_group_splits = 0;   // tuning hook 1500 + s: M-splits of the grouped bf16 weight gradient (0 = cost model)
void atst_gemm_nt_set_variant(int v) {
  if (v >= 2120 && v < 2122) g_f8_w4 = v - 2120;
  else if (v >= 2110 && v < 2112) g_f8_fuse_ln = v - 2110;
  else if (v >= 2100 && v < 2102) g_f8_resid16 = v - 2100;
  else if (v >= 2000 && v < 2100) g_tt = v - 2000;
  else if (v >= 1000 && v < 2000) g_p8_skew = v - 1000;
  else if (v >= 396 && v <= 398) g_ph = v - 396;
  else if (v >= 390) g_p8 = v - 390;
  else if (v >= 380) g_f32_splitk = v - 380;
  else if (v >= 370) g_bf16_tr = v - 370;
  else if (v >= 360) g_w4_auto = v - 360;
  else if (v >= 350) g_w4_min_m = v == 351 ? 1 : 8192;
  else if (v >= 330) g_w4_mode = v - 330;
  else if (v >= 306) g_dgelu_row384 = v - 306;
  else if (v >= 302) g_row384_tall = v - 302;
  else if (v >= 300) g_row384_auto = v - 300;
  else if (v >= 130 && v < 132) g_tn_map = v - 130;
  else if (v >= 120 && v < 130) g_tn_cfg = v - 120;
  else if (v >= 110) g_tn_rounds = v - 110;
  else if (v >= 105) g_tn_tall = v - 105;
  else if (v < 100) g_nt_variant = v;
}

int atst_gemm_nt(const GemmArgs& a0, hipStream_t st) {
  GemmArgs a = a0;
  // outputs that may be left out: the bf16 activation of fc1 + GELU and the bf16 du of the dGELU dgrad, when an e4m3 copy is written instead
  if ((a.epi == EPI_BIAS_GELU && !a.C2 && !a.q8) || (a.epi == EPI_DGELU && !a.C && !a.q8) || (a.epi != EPI_BIAS_GELU && a.epi != EPI_DGELU && !a.C)) return ATST_EINVAL;
  if (a.fp8) {                                                    // e4m3: K a multiple of 64 bytes, row-384 tile only; row-wise epilogues (round 6): N == 384, fp32 residual stream
    if (a.M <= 0 || a.N % 384 || a.K % 64 || a.lda % 16 || a.ldb % 16) return ATST_EINVAL;
    if (a.ln_out && a.epi != EPI_RESID) return ATST_EINVAL;
    if (ln_fused(a) && (a.N != 384 || a.ldc != 384 || !a.ln_gamma || !a.ln_beta || !a.ln_mean || !a.ln_rstd || a.resid_bf16 || a.out_bf16 || !a.resid)) return ATST_EINVAL;
    if (a.epi == EPI_LNBWD && (a.N != 384 || a.ldc != 384 || !a.ln_gamma || !a.ln_mean || !a.ln_rstd || !a.lnb_x || !a.lnb_dgamma || !a.lnb_dbeta)) return ATST_EINVAL;
    a.K /= 2; a.lda /= 2; a.ldb /= 2;
    switch (a.epi) {
      case EPI_LNBWD: return launch_nt_row384<EPI_LNBWD>(a, st);
      case EPI_BF16: return launch_nt_row384<EPI_BF16>(a, st);
      case EPI_F32: return launch_nt_row384<EPI_F32>(a, st);
      case EPI_BIAS_GELU: return launch_nt_row384<EPI_BIAS_GELU>(a, st);
      case EPI_RESID: return launch_nt_row384<EPI_RESID>(a, st);
      case EPI_DGELU: return launch_nt_row384<EPI_DGELU>(a, st);
    }
    return ATST_EINVAL;
  }
  if (a.M <= 0 || a.N % BN || a.K % BK || a.lda % 8 || a.ldb % 8) return ATST_EINVAL;
  if (a.q8 && a.epi != EPI_LNBWD) return ATST_EINVAL;             // e4m3 copies of an output: the e4m3 GEMMs, and the LayerNorm-backward epilogue of a bf16 dgrad GEMM
  switch (a.epi) {
    case EPI_BF16: return launch_nt<EPI_BF16>(a, st);
    case EPI_F32: return launch_nt<EPI_F32>(a, st);
    case EPI_BIAS_GELU: return launch_nt<EPI_BIAS_GELU>(a, st);
    case EPI_RESID: return launch_nt<EPI_RESID>(a, st);
    case EPI_DGELU: return launch_nt<EPI_DGELU>(a, st);
    case EPI_PATCH: return launch_nt<EPI_PATCH>(a, st);
    case EPI_LNBWD: return launch_nt<EPI_LNBWD>(a, st);
  }
  return ATST_EINVAL;
}

bool tn_tall_ok(const WgradArgs& a);

int atst_gemm_tn(const WgradArgs& a, hipStream_t st) {
  if (a.M <= 0 || a.N % 128 || a.K % 128 || a.ldy % 8 || a.ldx % 8) return ATST_EINVAL;
  WgradArgs p = a;
  if (tn_tall_ok(a) && a.m_per_split <= 0) {
    const int tiles = (a.N / tnt::TN) * (a.K / tnt::TK);
    int splits = 256 / tiles; if (splits < 1) splits = 1;        // one block per CU, one round
    int mps = (a.M + splits - 1) / splits;
    mps = ((mps + tnt::RM_ALIGN - 1) / tnt::RM_ALIGN) * tnt::RM_ALIGN;
    p.m_per_split = mps;
    const int nblk = tiles * ((a.M + mps - 1) / mps);
    ProfScope ps(PK_GEMM_TN, 2.0 * p.M * p.N * p.K, st, 2.0 * p.M * ((double)p.N + p.K) + 4.0 * p.N * p.K);
    return dispatch_tn_tall(p, nblk, st);
  }
  const int tiles = (a.N / 128) * (a.K / 128);
  if (p.m_per_split <= 0) {
    // 2 blocks / CU are resident (512 slots): pick the split count so that the grid is just under a whole number of
    // rounds (a 1044-block grid costs three rounds for two rounds of work)
    int splits = (g_tn_rounds * 512) / tiles;
    if (splits < 1) splits = 1;
    int mps = (a.M + splits - 1) / splits;
    mps = ((mps + WM - 1) / WM) * WM;
    if (mps < 4 * WM) mps = 4 * WM;
    p.m_per_split = mps;
  }
  const int splits = (p.M + p.m_per_split - 1) / p.m_per_split;
  const int nblk = tiles * splits;
  ProfScope ps(PK_GEMM_TN, 2.0 * p.M * p.N * p.K, st, 2.0 * p.M * ((double)p.N + p.K) + 4.0 * p.N * p.K);
  static OncePerDevice attr_done; int attr_done_dev;
  if (attr_done.need(attr_done_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WGRAD_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_done.done(attr_done_dev);
  }
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(nblk), dim3(256), WGRAD_LDS_BYTES, st, p);
  return (int)hipGetLastError();
}

bool tn_tall_ok(const WgradArgs& a) {
  return g_tn_tall && a.N % tnt::TN == 0 && a.K % tnt::TK == 0 && a.M % tnt::RM_ALIGN == 0 && a.M >= 8192 && a.ldy % 8 == 0 && a.ldx % 8 == 0;
}

int atst_gemm_tn_group(const WgradArgs* items, int n, hipStream_t st) {
  if (n < 1 || n > ATST_WGRAD_GROUP_MAX) return ATST_EINVAL;
  bool tall = true;
  for (int i = 0; i < n; ++i) tall = tall && tn_tall_ok(items[i]);
  if (!tall || n == 1) {                                          // shapes outside the tall tile: one launch each
    for (int i = 0; i < n; ++i) { const int rc = atst_gemm_tn(items[i], st); if (rc) return rc; }
    return ATST_OK;
  }
  WgradGroup g{};
  g.n = n;
  int tiles = 0; double flops = 0, bytes = 0;
  for (int i = 0; i < n; ++i) {
    g.it[i] = items[i];
    g.first_tile[i] = tiles;
    tiles += (items[i].N / tnt::TN) * (items[i].K / tnt::TK);
    flops += 2.0 * items[i].M * items[i].N * items[i].K;
    bytes += 2.0 * items[i].M * ((double)items[i].N + items[i].K) + 4.0 * items[i].N * items[i].K;
  }
  for (int i = n; i <= ATST_WGRAD_GROUP_MAX; ++i) g.first_tile[i] = tiles;
  // M-splits.  One block per CU is resident; a grid of tiles x splits blocks runs in ceil(tiles splits / 256) rounds of M / splits rows
  // each, and every split adds one pass of fp32 atomics over the outputs (measured ~1.5 TB/s).  With few tiles (d = 384: 24) one
  // round of 256 / tiles splits is best; with many (d = 768: 96 tiles -> 2 splits = 192 blocks, a quarter of the chip idle for the
  // whole launch) several full rounds of shorter blocks win: 96 x 8 = 768 blocks = 3 full rounds.  Cost model per candidate:
  // rounds / splits x T_M + splits x T_atom, T_M = 41 ns per row of one tile (measured), T_atom = output bytes / 1.5 TB/s.
  int splits = 1;
  {
    double out_bytes = 0; int Mmax = 0;
    for (int i = 0; i < n; ++i) { out_bytes += 4.0 * items[i].N * items[i].K; Mmax = items[i].M > Mmax ? items[i].M : Mmax; }
    const double t_m = Mmax * 0.041, t_atom = out_bytes / 1.5e6;          // microseconds
    double best = 1e30;
    for (int sp = 1; sp <= 32; ++sp) {
      if ((long)Mmax / sp < 4 * tnt::RM_ALIGN) break;
      const double cost = (double)((tiles * sp + 255) / 256) / sp * t_m + sp * t_atom;
      if (cost < best) { best = cost; splits = sp; }
    }
    if (g_tn_group_splits > 0) splits = g_tn_group_splits;            // tuning hook 1500 + s
  }
  int max_splits = 1, min_splits = 1 << 30;
  for (int i = 0; i < n; ++i) {
    int mps = (items[i].M + splits - 1) / splits;
    mps = ((mps + tnt::RM_ALIGN - 1) / tnt::RM_ALIGN) * tnt::RM_ALIGN;
    g.it[i].m_per_split = mps;
    const int sp = (items[i].M + mps - 1) / mps;
    if (sp > max_splits) max_splits = sp;
    if (sp < min_splits) min_splits = sp;
  }
  // every problem cut into the same number of splits (always, on the encoder path: one M): groups packed onto XCDs, see build_wgrad_map
  g.use_map = (g_tn_map && min_splits == max_splits && build_wgrad_map(g, max_splits, tiles * max_splits)) ? 1 : 0;
  ProfScope ps(PK_GEMM_TN, flops, st, bytes);
  return dispatch_tn_tall_group(g, tiles * max_splits, st);
}

