// FROZEN COPY of audiossl_amd/csrc/gemm.hip as of round 2 (commit 623845e): every measured-and-rejected GEMM variant behind its
// switch -- ping-pong main loop (hook 321), epilogue straight from the registers (341), 64-deep ring stages (311), 128x384
// 4-wave tile for every epilogue (331), start-up phase skew (100000+c), wgrad interleaved issue / split loaders / 32-row
// stages (ATST_TN_ILV / ATST_TN_SPLIT / ATST_TN_RM), ablation builds (ATST_ABLATE) and the s_memtime phase tracers
// (ATST_TRACE, ATST_TRACE_FINE).  Results: profiles/r02_trace_epi.txt, profiles/r02_trace_gemm.txt, DESIGN.md section 3.
// Built INSTEAD of csrc/gemm.hip by `ATST_GEMM_VARIANTS=1 python audiossl_amd/build.py` (or with any of the switches set) for
// the stand-alone GEMM tools only; it has no EPI_LNBWD, so the encoder backward does not run on such a build.
// bf16 MFMA GEMMs for the ATST encoder / heads on gfx950.
//
//   gemm_nt  : C[M,N] = A[M,K] * B[N,K]^T  (+ fused epilogue)      forward GEMMs (weights are [out,in] row-major, torch
//              convention) and dgrad GEMMs (B = pre-transposed bf16 weight copy)
//   gemm_tn  : dW[N,K] += dY[M,N]^T * X[M,K]  (fp32 atomic accumulate, split over M)   wgrad GEMMs; both operands are
//              m-major in HBM, so fragments come from row-major LDS tiles through ds_read_b64_tr_b16.
//
// Tile 128x128x64, 4 waves (2x2), each wave 64x64 = 2x2 v_mfma_f32_32x32x16_bf16 accumulators; register-staged
// global->LDS double buffering (one barrier per K-tile); XCD-aware block->tile map so blocks that share an A row panel
// share an L2.  Reference math being accelerated: nn.Linear in audiossl/modules/transformer.py:109,119,87-90 and
// audiossl/models/atst/audio_transformer.py:63,69 ; audiossl/models/atst/byol.py:13.
#include "common.h"
#include "kernels.h"
#include "profile.h"
#ifndef ATST_NT_STORES
#define ATST_NT_STORES 1    // epilogue outputs / residual reads are streamed once: non-temporal, so they do not evict operand panels from L2
#endif
#ifndef ATST_ABLATE
#define ATST_ABLATE 0      // experiment switch (tools only), all without stores: 1 full, 3 loads+ds_read, 4 ds_read+MFMA, 5 loads only, 6 MFMA only; 7 (row384): epilogue only
#endif

#include <type_traits>
#ifndef ATST_EXPERIMENTS
#define ATST_EXPERIMENTS 0  // 1: also compile the measured-and-rejected variants behind the tuning hooks (ping-pong main loop 321,
                            // register epilogue 341, 64-deep ring 311, 128x384 4-wave tile for every epilogue 331): profiles/r02_trace_epi.txt
#endif
#ifndef ATST_TRACE_FINE
#define ATST_TRACE_FINE 0   // 1: also stamp every k-tile of the main loop (tools/trace_gemm.py; perturbs the loop)
#endif
#ifndef ATST_TRACE
#define ATST_TRACE 0       // experiment builds (tools/trace_gemm.py): block ATST_TRACE-1 of the row-384 kernel stamps s_memtime at its phase boundaries into p.colsum
#endif

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

// Blocks of one launch start together and every tile costs the same, so all CUs (and both blocks of a CU) run their
// main loops -- HBM reads only -- and then their epilogues -- HBM writes only -- in lock step.  Delaying every other
// first-round block by about half a tile time puts half of the chip in each phase at any moment.
DEVFN void phase_skew(int cycles) {
  if (cycles > 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)cycles) __builtin_amdgcn_s_sleep(16);
  }
}

template <int EPI, bool SCALE = true>
DEVFN void epi_fetch8(const GemmArgs& p, int row, int col, EpiAux& x) {
  const size_t idx = (size_t)row * p.ldc + col;
  if constexpr (EPI == EPI_RESID) {
    x.a0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.resid + idx));
    x.a1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.resid + idx + 4));
    if constexpr (SCALE) x.s = p.row_scale ? p.row_scale[row / p.rows_per_seq] : 1.0f;   // else: the caller supplies it
  } else if constexpr (EPI == EPI_DGELU) {
    x.a0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.U + idx));       // 8 bf16 pre-activations
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
#if ATST_NT_STORES
    __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(base) + i));   // streamed once: keep L2 for operands
#else
    *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(base) + i) = o;
#endif
  };
  auto st_f32 = [](void* base, size_t i, const f32x4& lo, const f32x4& hi4) {
#if ATST_NT_STORES
    __builtin_nontemporal_store(lo, reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + i));
    __builtin_nontemporal_store(hi4, reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + i + 4));
#else
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + i) = lo;
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + i + 4) = hi4;
#endif
  };
  if constexpr (EPI == EPI_BF16) {
    st_bf16(p.C, idx, v0 + b0, v1 + b1);
  } else if constexpr (EPI == EPI_F32) {
    st_f32(p.C, idx, v0 + b0, v1 + b1);
  } else if constexpr (EPI == EPI_BIAS_GELU) {
    v0 += b0; v1 += b1;
    if (p.C) st_bf16(p.C, idx, v0, v1);                            // pre-activation u (saved for backward; skipped in inference)
    f32x4 g0, g1;
#pragma unroll
    for (int e = 0; e < 4; ++e) { g0[e] = gelu_f(v0[e]); g1[e] = gelu_f(v1[e]); }
    st_bf16(p.C2, idx, g0, g1);                                    // activation a
    if (p.q8) {                                                    // fp8 forward: e4m3 copy of the SAME bf16 values for the fc2 GEMM
      const float s = p.q8_scale;
      auto c = [&](float a_) { return __builtin_amdgcn_fmed3f(bf2f(f2bf(a_)) * s, -448.f, 448.f); };
      int lo = __builtin_amdgcn_cvt_pk_fp8_f32(c(g0[0]), c(g0[1]), 0, false); lo = __builtin_amdgcn_cvt_pk_fp8_f32(c(g0[2]), c(g0[3]), lo, true);
      int hi_w = __builtin_amdgcn_cvt_pk_fp8_f32(c(g1[0]), c(g1[1]), 0, false); hi_w = __builtin_amdgcn_cvt_pk_fp8_f32(c(g1[2]), c(g1[3]), hi_w, true);
      typedef int v2i_ __attribute__((ext_vector_type(2)));
      *reinterpret_cast<v2i_*>(p.q8 + idx) = v2i_{lo, hi_w};
    }
  } else if constexpr (EPI == EPI_RESID) {
    st_f32(p.C, idx, x.a0 + x.s * (v0 + b0), x.a1 + x.s * (v1 + b1));
  } else if constexpr (EPI == EPI_DGELU) {
    const bf16x8 u = __builtin_bit_cast(bf16x8, x.a0);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v0[e] *= gelu_grad_f(bf2f(u[e])); v1[e] *= gelu_grad_f(bf2f(u[4 + e])); }
    st_bf16(p.C, idx, v0, v1);
    w0 = v0; w1 = v1;
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

template <int EPI, int BMT, int NSTG, int WTM>
__global__ __launch_bounds__((NtGeo<BMT, NSTG, WTM>::THREADS), (NtGeo<BMT, NSTG, WTM>::WPS)) void gemm_nt_kernel(GemmArgs p) {
  using G = NtGeo<BMT, NSTG, WTM>;
  constexpr int MI = G::MI;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1, hi = lane >> 5, l31 = lane & 31;

  const int ntn = p.N / BN;
  const int ntm = (p.M + BMT - 1) / BMT;
  const int id = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (id / ntn) * BMT, n0 = (id % ntn) * BN;

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
    srcA[j] = p.A + (size_t)ra * p.lda + (lchunk ^ ((row >> 2) & 3)) * 8;
  }
#pragma unroll
  for (int j = 0; j < NB_PER_WAVE; ++j) {
    const int row = (wid * NB_PER_WAVE + j) * 16 + lrow;
    srcB[j] = p.B + (size_t)(n0 + row) * p.ldb + (lchunk ^ ((row >> 2) & 3)) * 8;
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

  const int nk = p.K / BK;
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
#if ATST_ABLATE != 4 && ATST_ABLATE != 6
    if (kt + NSTG - 1 < nk) issue(kt + NSTG - 1);
#endif
    const char* st = lds + (kt % NSTG) * G::STAGE;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int co = ((ks * 2 + hi) ^ xr) << 4;
      bf16x8 af[MI];
#if ATST_ABLATE == 5 || ATST_ABLATE == 6
      bf16x8 b0, b1;                                               // no LDS reads: operands are whatever is in registers
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) asm volatile("" : "=v"(af[mi]));
      asm volatile("" : "=v"(b0), "=v"(b1));
      (void)st; (void)co;
#else
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[mi] = *reinterpret_cast<const bf16x8*>(st + offA + mi * 32 * 64 + co);
      bf16x8 b0 = *reinterpret_cast<const bf16x8*>(st + offB + co), b1 = *reinterpret_cast<const bf16x8*>(st + offB + 32 * 64 + co);
#endif
#if ATST_ABLATE == 3 || ATST_ABLATE == 5
      asm volatile("" :: "v"(af[0]), "v"(af[MI - 1]), "v"(b0), "v"(b1));
#else
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        acc[mi][0] = mfma32(af[mi], b0, acc[mi][0]);
        acc[mi][1] = mfma32(af[mi], b1, acc[mi][1]);
      }
#endif
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
#if ATST_ABLATE
      if (sC[rl * C_LD + c8] != 12345.678f) continue;                // experiment builds: no epilogue stores
#endif
      if (row < p.M) {
        f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = w0;
        epilogue8<EPI>(p, row, n0 + c8, *reinterpret_cast<const f32x4*>(sC + rl * C_LD + c8),
                       *reinterpret_cast<const f32x4*>(sC + rl * C_LD + c8 + 4), bias0, bias1, aux[pass], w0, w1);
        if constexpr (EPI == EPI_DGELU) { csum0 += w0; csum1 += w1; }
      }
    }
    if (part + 1 < BMT / 64) __syncthreads();
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

// ---- 128 x 384 tile: 8 waves (2 x 4), each 64 x 96 = 2 x 3 accumulators ---------------------------------------------
// Every N on this path (384, 1152, 1536, 768...) is a multiple of 384.  Per 32-deep K step the block stages 8 KB of A
// and 24 KB of B for three 128x128 units of output: 10.7 KB / unit instead of 16 KB for the square tile -- operand
// staging (L2 -> LDS, ~14 TB/s chip-wide) is what bounds these GEMMs, see DESIGN.md.  For N = 384 one block owns whole
// output rows.
namespace row384 {
constexpr int BNR = 384, WAVES = 8, THREADS = 512, CLD = BNR + 4;
#ifndef ATST_TALL_STAGES
#define ATST_TALL_STAGES 3
#endif
#ifndef ATST_TN_ISSUE
#define ATST_TN_ISSUE 1        // tall wgrad: 1 next stage's LDS-DMA in front of the MFMAs (2-stage ring: they need the whole stage to land; measured best), 0 one per MFMA group, 2 two per group
#endif
#ifndef ATST_INTERLEAVE
#define ATST_INTERLEAVE 1      // LDS-DMA issue spread between the MFMA groups (0: in front of them; experiment builds)
#endif
constexpr int CLD2 = 448, PLANE1 = 208;       // staging layout of the epilogues WITHOUT the fused LayerNorm (see the staging loop)
constexpr int B_BYTES = BNR * BK * 2, EPI_BYTES = 32 * CLD2 * 4;                                  // 24 KB ; 57,344 B
// MI = 32-row accumulator blocks per wave: 2 -> 128-row tile, 4 -> 256-row tile.  BKT = K depth of one ring stage:
// 32 (64-B LDS rows, 4 chunks) or 64 (128-B rows: every LDS-DMA lane group fetches a whole 128-B line, half as many
// barriers per K; two stages then fill the CU's 160 KB).
template <int MI, int BKT = BK> struct Geo {
  static constexpr int BMR = 64 * MI, ROWB = BKT * 2, A_BYTES = BMR * ROWB, BB = BNR * ROWB, STAGE = A_BYTES + BB;   // 32 / 40 KB (BKT 32), 80 KB (256 rows, BKT 64)
  static constexpr int NSTG = BKT == 64 ? 2 : (MI == 4 ? ATST_TALL_STAGES : 2);
  static constexpr int LDS = NSTG * STAGE > EPI_BYTES + 8192 ? NSTG * STAGE : EPI_BYTES + 8192;  // 64 KB (2 blocks / CU) ; 120 KB ; 160 KB
  static constexpr int RPI = 1024 / ROWB;                                                        // rows per 1-KiB load instruction
  static constexpr int A_IPW = (BMR / RPI) / WAVES, B_IPW = (BNR / RPI) / WAVES;                 // load instructions per wave
  static constexpr int CPR = ROWB / 16;                                                          // 16-B chunks per row
};
// XOR key of a row's 16-B chunks: the 16 rows of a ds_read_b128 service group must land on 16 distinct bank slots
template <int BKT> DEVFN int swz_key(int row) { return BKT == 64 ? (row >> 1) & 7 : (row >> 2) & 3; }
}

// MI = 4 (256 x 384 tile, each wave 128 x 96 = 12 accumulators): 40 KB staged per 6 units of output = 6.7 KB / unit.
// DIR: epilogue straight from the accumulator registers (below).  The MFMA operands are then swapped (C^T = B A^T), so
// that a lane owns ONE output row (m = lane & 31) and its registers run along n.
// F8: the operands are OCP e4m3 bytes.  The host passes them as pairs ("bf16" elements: K, lda, ldb halved), so the whole
// LDS-DMA ring -- 64-byte rows, XOR swizzle, stage geometry -- is byte-identical; a k-tile then covers K = 64 and feeds ONE
// v_mfma_scale_f32_32x32x64_f8f6f4 per accumulator (unit block scales; twice the bf16 MFMA rate) instead of two
// 32x32x16 bf16 MFMAs: half the matrix-pipe time AND half the operand bytes per FLOP.  p.dq undoes the per-tensor scales.
template <int EPI, int MI, bool LN = false, int BKT = BK, bool PP = false, bool DIR = false, bool F8 = false>
__global__ __launch_bounds__(512, (MI == 2 ? 4 : 2)) void gemm_nt_row384_kernel(GemmArgs p) {
  using namespace row384;
  using RG = row384::Geo<MI, BKT>;
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
  if (MI == 4 && blockIdx.x < 256 && ((blockIdx.x >> 3) & 1)) phase_skew(p.skew);     // one block per CU: every other CU of each XCD

#if ATST_TRACE
  if (EPI != EPI_DGELU && p.colsum && (int)blockIdx.x == ATST_TRACE - 1 && lane == 0)
    (reinterpret_cast<unsigned long long*>(p.colsum) + (size_t)wid * 64 * 8)[7] = __builtin_amdgcn_s_memtime();
#endif
  char* lds = smem_raw;
  const int lrow = lane / RG::CPR, lchunk = lane % RG::CPR;
  const bf16* srcA[RG::A_IPW]; const bf16* srcB[RG::B_IPW];
#pragma unroll
  for (int j = 0; j < RG::A_IPW; ++j) {
    const int row = (wid * RG::A_IPW + j) * RG::RPI + lrow;
    int ra = m0 + row; ra = ra < p.M ? ra : p.M - 1;
    srcA[j] = p.A + (size_t)ra * p.lda + (lchunk ^ swz_key<BKT>(row)) * 8;
  }
#pragma unroll
  for (int j = 0; j < RG::B_IPW; ++j) {
    const int row = (wid * RG::B_IPW + j) * RG::RPI + lrow;
    srcB[j] = p.B + (size_t)(n0 + row) * p.ldb + (lchunk ^ swz_key<BKT>(row)) * 8;
  }
  auto issue = [&](int kt) {
    char* st = lds + (kt % NSTG) * STAGE;
#pragma unroll
    for (int j = 0; j < RG::A_IPW; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(srcA[j] + kt * BKT), (lptr_t)(st + (wid * RG::A_IPW + j) * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < RG::B_IPW; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(srcB[j] + kt * BKT), (lptr_t)(st + A_BYTES + (wid * RG::B_IPW + j) * 1024), 16, 0, 0);
  };
  f32x16 acc[MI][3];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#if ATST_ABLATE == 7
  const int nk = p.K < 0 ? 1 : 0;                                // experiment builds: epilogue only
#else
  const int nk = p.K / BKT;
#endif
  const int xr = swz_key<BKT>(l31);                                // every fragment row is l31 plus a multiple of 32
  const int offA = (wm * 32 * MI + l31) * ROWB, offB = A_BYTES + (wn * 96 + l31) * ROWB;
  constexpr int LOADS_PER_TILE = RG::A_IPW + RG::B_IPW;           // per wave, in issue order
  constexpr bool ILV = ATST_INTERLEAVE && MI == 4 && NSTG >= 3;   // the 128-row tile has no registers to spare for the pinned order; a 2-stage ring needs its loads early
  auto issue_one = [&](int kt, int j) {                           // j-th load instruction of tile kt
    char* st = lds + (kt % NSTG) * STAGE;
    if (j < RG::A_IPW)
      __builtin_amdgcn_global_load_lds((gptr_t)(srcA[j < RG::A_IPW ? j : 0] + kt * BKT), (lptr_t)(st + (wid * RG::A_IPW + j) * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((gptr_t)(srcB[j >= RG::A_IPW ? j - RG::A_IPW : 0] + kt * BKT), (lptr_t)(st + A_BYTES + (wid * RG::B_IPW + j - RG::A_IPW) * 1024), 16, 0, 0);
  };
  // One K-tile of MFMAs.  ISSUE: the loads of tile kt + NSTG - 1 are spread BETWEEN the MFMA groups instead of in front
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
          acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8[ni], acc[mi][ni], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
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
    for (int ks = 0; ks < BKT / 16; ++ks) {
      const int co = ((ks * 2 + hi) ^ xr) << 4;
      bf16x8 af[MI], bf[3];
#if ATST_ABLATE == 5 || ATST_ABLATE == 6
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) asm volatile("" : "=v"(af[mi]));   // no LDS reads: operands are whatever is in registers
#pragma unroll
      for (int ni = 0; ni < 3; ++ni) asm volatile("" : "=v"(bf[ni]));
      (void)st; (void)co;
#else
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[mi] = *reinterpret_cast<const bf16x8*>(st + offA + mi * 32 * ROWB + co);
#pragma unroll
      for (int ni = 0; ni < 3; ++ni) bf[ni] = *reinterpret_cast<const bf16x8*>(st + offB + ni * 32 * ROWB + co);
#endif
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
#if ATST_ABLATE == 3 || ATST_ABLATE == 5
        asm volatile("" :: "v"(af[mi]), "v"(bf[0]), "v"(bf[1]), "v"(bf[2]));
#else
#pragma unroll
        for (int ni = 0; ni < 3; ++ni) acc[mi][ni] = DIR ? mfma32(bf[ni], af[mi], acc[mi][ni]) : mfma32(af[mi], bf[ni], acc[mi][ni]);
#endif
#if ATST_ABLATE != 4 && ATST_ABLATE != 6
        if (ILV && ISSUE && slot < LOADS_PER_TILE) {
          __builtin_amdgcn_sched_barrier(0);
          issue_one(kt + NSTG - 1, slot);
          __builtin_amdgcn_sched_barrier(0);
          ++slot;
        }
#endif
      }
    }
  };
  if constexpr (PP) {
    // Ping-pong schedule.  The two waves that share a SIMD (wid and wid + 4: wave row 0 and wave row 1 of the tile)
    // alternate roles every half k-tile: while one runs its 24 MFMAs of tile t, the other issues its LDS-DMA share of
    // tile t + 2 -- so the block never has all eight waves queueing on the CU's one vector-memory address path while
    // the matrix pipes idle, nor all eight contending for the matrix pipes while the address path idles (the
    // lock-step schedule below does exactly that: loads-only 104 us + MFMA-only 91 us = 147 us together).
    //   phase 2t  : row 0 computes tile t        | row 1 issues tile t+2, waits for its share of tile t+1
    //   phase 2t+1: row 0 issues tile t+2, waits | row 1 computes tile t
    // Every wave executes two barriers per k-tile.  Stage (t+2) % 3 == (t-1) % 3 was last read in phases 2t-2 / 2t-1.
    static_assert(!PP || (NSTG == 3 && LOADS_PER_TILE == 5), "ping-pong schedule is written for the 3-stage, 5-loads-per-wave ring");
#if ATST_TRACE
    unsigned long long* trc = reinterpret_cast<unsigned long long*>(p.colsum) + (size_t)wid * 64 * 8;
    const bool trace = ATST_TRACE_FINE && EPI == EPI_BF16 && p.colsum && (int)blockIdx.x == ATST_TRACE - 1 && lane == 0;
#define STAMP(k, i) do { if (trace && (k) < 64) trc[(k) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(k, i) do { } while (0)
#endif
    STAMP(0, 6);
    issue(0);
    if (nk > 1) { issue(1); asm volatile("s_waitcnt vmcnt(5)\n\ts_barrier" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = kt + 2 < nk;
      STAMP(kt, 0);
      if (wm == 0) {
        tile(kt, std::false_type{});
        STAMP(kt, 1);
        asm volatile("s_barrier" ::: "memory");
        STAMP(kt, 2);
        if (more) { issue(kt + 2); STAMP(kt, 3); asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(kt, 4);
        asm volatile("s_barrier" ::: "memory");
      } else {
        if (more) { issue(kt + 2); STAMP(kt, 3); asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(kt, 4);
        asm volatile("s_barrier" ::: "memory");
        STAMP(kt, 2);
        tile(kt, std::false_type{});
        STAMP(kt, 1);
        asm volatile("s_barrier" ::: "memory");
      }
      STAMP(kt, 5);
    }
  } else {
#pragma unroll
  for (int t = 0; t < NSTG - 1; ++t)
    if (t < nk) issue(t);
  const int nfull = nk - (NSTG - 1) > 0 ? nk - (NSTG - 1) : 0;    // tiles that still have a successor to fetch
  for (int kt = 0; kt < nfull; ++kt) {
    // my share of tile kt has landed (loads retire in order; the younger NSTG-2 tiles may still be in flight); barrier =>
    // everyone's has, and everyone is done reading stage (kt-1) % NSTG, which the next issue overwrites
    if constexpr (NSTG == 2) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    else if constexpr ((NSTG - 2) * LOADS_PER_TILE == 4) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    else if constexpr ((NSTG - 2) * LOADS_PER_TILE == 5) asm volatile("s_waitcnt vmcnt(5)\n\ts_barrier" ::: "memory");
    else if constexpr ((NSTG - 2) * LOADS_PER_TILE == 8) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    else if constexpr ((NSTG - 2) * LOADS_PER_TILE == 10) asm volatile("s_waitcnt vmcnt(10)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#if ATST_ABLATE != 4 && ATST_ABLATE != 6
    if (!ILV) issue(kt + NSTG - 1);
#endif
#if ATST_TRACE
    { unsigned long long* trc = reinterpret_cast<unsigned long long*>(p.colsum) + (size_t)wid * 64 * 8;
      if (ATST_TRACE_FINE && EPI == EPI_BF16 && p.colsum && (int)blockIdx.x == ATST_TRACE - 1 && lane == 0 && kt < 64) trc[kt * 8 + 0] = __builtin_amdgcn_s_memtime(); }
#endif
    tile(kt, std::true_type{});
#if ATST_TRACE
    { unsigned long long* trc = reinterpret_cast<unsigned long long*>(p.colsum) + (size_t)wid * 64 * 8;
      if (ATST_TRACE_FINE && EPI == EPI_BF16 && p.colsum && (int)blockIdx.x == ATST_TRACE - 1 && lane == 0 && kt < 64) trc[kt * 8 + 1] = __builtin_amdgcn_s_memtime(); }
#endif
  }
  for (int kt = nfull; kt < nk; ++kt) {                           // drain: nothing left to fetch
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    tile(kt, std::false_type{});
  }
  asm volatile("s_barrier" ::: "memory");
  }

#if ATST_TRACE
  unsigned long long* trc2 = reinterpret_cast<unsigned long long*>(p.colsum) + (size_t)wid * 64 * 8;
  const bool trace2 = EPI != EPI_DGELU && p.colsum && (int)blockIdx.x == ATST_TRACE - 1 && lane == 0;
#define STAMP2(i) do { if (trace2) trc2[(i) * 8 + 7] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP2(i) do { } while (0)
#endif
  STAMP2(1);
  if constexpr (DIR) {
    // ---- epilogue straight from the accumulator registers --------------------------------------------------------------
    // Measured (tools/trace_epi.py, profiles/r02_trace_epi.txt): staging the tile through LDS costs ~1.9 k cycles for each
    // of its 8 parts whatever the epilogue does (24 ds_write_b32 per lane, two block barriers, conflicted read-back) =
    // 15 k cycles per tile against a 29 k main loop at K = 384; and an epilogue that loads (residual) pays one exposed HBM
    // round trip per part (8.8 k cycles each).  With the swapped MFMA orientation a lane owns row m = lane & 31 of each
    // 32 x 32 accumulator block and 4-column groups {8 g + 4 hi .. + 3}; one v_permlane32_swap per register pair trades
    // groups between the half-waves so that every lane holds two runs of 8 consecutive columns (16 hi + 8 k .. + 7).  Those
    // runs are exactly what epi_fetch8 / epilogue8 work on: 16-B (bf16) / 2 x 16-B (fp32) global accesses per run, no LDS
    // round trip, no barrier (the fused LayerNorm needs one per 32-row block for the cross-wave row statistics), and the
    // residual loads of a whole accumulator row block are in flight at once (12 KB per wave).
    float* sBias = reinterpret_cast<float*>(smem_raw);
    float* sGamma = sBias + BNR; float* sBeta = sGamma + BNR;
    float* sScale = sBeta + BNR;                                  // [BMR]
    float* sStat = sScale + BMR;                                  // [4 wave columns][BMR rows][2]: (mean, M2) of 96 columns
    constexpr bool fused_ln = LN && EPI == EPI_RESID;
    if (tid < BNR) {
      sBias[tid] = p.bias ? p.bias[n0 + tid] : 0.f;
      if (fused_ln) { sGamma[tid] = p.ln_gamma[tid]; sBeta[tid] = p.ln_beta[tid]; }
    }
    if constexpr (EPI == EPI_RESID) {
      if (tid < BMR) { const int row = m0 + tid; sScale[tid] = (p.row_scale && row < p.M) ? p.row_scale[row / p.rows_per_seq] : 1.0f; }
    }
    __syncthreads();
    auto swap_runs = [&](f32x16& v) {                             // afterwards run k = {v[4k..4k+3], v[4k+8..4k+11]}: columns 16 hi + 8 k .. + 7
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float fa = v[4 * k + e], fb = v[4 * k + 8 + e];    // (bit_cast straight on a vector-element lvalue reads element 0: clang quirk)
          auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, fa), __builtin_bit_cast(unsigned, fb), false, false);
          const unsigned r0 = r[0], r1 = r[1];
          v[4 * k + e] = __builtin_bit_cast(float, r0); v[4 * k + 8 + e] = __builtin_bit_cast(float, r1);
        }
    };
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int trow = wm * 32 * MI + mi * 32 + l31, row = m0 + trow;
      const bool live = row < p.M;
      const int cbase = wn * 96 + 16 * hi;                        // + ni * 32 + 8 * k
      if constexpr (!fused_ln) {
        EpiAux aux[3][2];
#pragma unroll
        for (int ni = 0; ni < 3; ++ni)
#pragma unroll
          for (int k = 0; k < 2; ++k)
            if (live) epi_fetch8<EPI, false>(p, row, n0 + cbase + ni * 32 + 8 * k, aux[ni][k]);
#pragma unroll
        for (int ni = 0; ni < 3; ++ni) {
          swap_runs(acc[mi][ni]);
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int c = cbase + ni * 32 + 8 * k;
            const f32x16& v = acc[mi][ni];
            f32x4 v0 = {v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]}, v1 = {v[4 * k + 8], v[4 * k + 9], v[4 * k + 10], v[4 * k + 11]};
            if (live) {
              f32x4 w0, w1;
              if constexpr (EPI == EPI_RESID) aux[ni][k].s = sScale[trow];
              epilogue8<EPI>(p, row, n0 + c, v0, v1, *reinterpret_cast<const f32x4*>(sBias + c), *reinterpret_cast<const f32x4*>(sBias + c + 4), aux[ni][k], w0, w1);
            }
          }
        }
      } else {
        // residual + LayerNorm of the next sub-layer.  Row statistics: the lane pair (hi = 0, 1) of every wave column holds
        // 96 of the row's 384 values -> (mean, M2) of those 96, exchanged through LDS, combined exactly (equal counts).
        f32x4 rr[3][2][2];
#pragma unroll
        for (int ni = 0; ni < 3; ++ni)
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const float* src = p.resid + (size_t)row * p.ldc + cbase + ni * 32 + 8 * k;
            if (live) { rr[ni][k][0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src)); rr[ni][k][1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + 4)); }
          }
        const float sc = sScale[trow];
        float s1 = 0.f;
#pragma unroll
        for (int ni = 0; ni < 3; ++ni) {
          swap_runs(acc[mi][ni]);
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int c = cbase + ni * 32 + 8 * k;
            f32x16& v = acc[mi][ni];
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(sBias + c), b1 = *reinterpret_cast<const f32x4*>(sBias + c + 4);
            f32x4 o0, o1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              o0[e] = rr[ni][k][0][e] + sc * (v[4 * k + e] + b0[e]);
              o1[e] = rr[ni][k][1][e] + sc * (v[4 * k + 8 + e] + b1[e]);
              v[4 * k + e] = o0[e]; v[4 * k + 8 + e] = o1[e];
              s1 += o0[e] + o1[e];
            }
            if (live) {
              float* dst = reinterpret_cast<float*>(p.C) + (size_t)row * p.ldc + c;
              __builtin_nontemporal_store(o0, reinterpret_cast<f32x4*>(dst));
              __builtin_nontemporal_store(o1, reinterpret_cast<f32x4*>(dst + 4));
            }
          }
        }
        s1 += __shfl_xor(s1, 32, 64);
        const float mw = s1 * (1.0f / 96.0f);
        float q = 0.f;
#pragma unroll
        for (int ni = 0; ni < 3; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) { const float d = acc[mi][ni][r] - mw; q += d * d; }
        q += __shfl_xor(q, 32, 64);
        if (hi == 0) { sStat[(wn * BMR + trow) * 2] = mw; sStat[(wn * BMR + trow) * 2 + 1] = q; }
        __syncthreads();
        float mu = 0.f, m2 = 0.f, mws[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) { mws[w] = sStat[(w * BMR + trow) * 2]; mu += mws[w]; m2 += sStat[(w * BMR + trow) * 2 + 1]; }
        mu *= 0.25f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const float d = mws[w] - mu; m2 += 96.0f * d * d; }
        const float rs = rsqrtf(m2 * (1.0f / 384.0f) + 1e-6f);
#pragma unroll
        for (int ni = 0; ni < 3; ++ni)
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int c = cbase + ni * 32 + 8 * k;
            const f32x16& v = acc[mi][ni];
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(sGamma + c), g1 = *reinterpret_cast<const f32x4*>(sGamma + c + 4);
            const f32x4 e0 = *reinterpret_cast<const f32x4*>(sBeta + c), e1 = *reinterpret_cast<const f32x4*>(sBeta + c + 4);
            float t[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              t[e] = (v[4 * k + e] - mu) * rs * g0[e] + e0[e];
              t[4 + e] = (v[4 * k + 8 + e] - mu) * rs * g1[e] + e1[e];
            }
            if (live) *reinterpret_cast<bf16x8*>(p.ln_out + (size_t)row * 384 + c) = pack8(t);
          }
        if (live && wn == 0 && hi == 0) { p.ln_mean[row] = mu; p.ln_rstd[row] = rs; }
      }
    }
#if ATST_TRACE
    STAMP2(9);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP2(10);
#endif
    return;
  }
  // Epilogue: the fp32 tile goes through LDS 32 rows at a time so that every global access is a 16-B piece of a full
  // 384-column row.  Part (mi, h) takes 16 rows of accumulator block mi from EVERY wave (two 16-row groups, one per
  // wave row), so all waves retire the same 24 accumulator registers per part and the registers freed by the dump hold
  // that part's global loads (residual / saved activations), which are issued as one batch before the staging barrier
  // while the stores follow it (see epi_fetch8).
  float* sC = reinterpret_cast<float*>(smem_raw);
  // rows staged per part: 32 (16 from each wave row).  64 (HPP = 2; the 256-row tile's 120 KB ring holds them: half as many
  // staging barriers) was measured: qkv 166 -> 160 us but proj/fc2+residual 127 -> 130 / 215 -> 220, step 57.98 -> 58.89 ms.
  constexpr int HPP = 1, RP = 32 * HPP, SLOTS = RP * 48 / THREADS, LNROWS = RP / WAVES;
  float* sBias = sC + RP * CLD2;                                  // bias of this block's 384 columns (zeros when absent)
  float* sGamma = sBias + CLD2; float* sBeta = sGamma + BNR;      // fused LayerNorm affine parameters
  float* sScale = sBeta + BNR;                                    // per-row DropPath scale of this block's rows
  float* sCol = sGamma;                                           // EPI_DGELU: column sums of du (fc1 bias gradient); no LN there
  constexpr int NPART = 2 * MI / HPP;
  constexpr bool fused_ln = LN && EPI == EPI_RESID;
  const float dqv = F8 ? (p.dq ? *p.dq : 1.0f) * (p.dq_mul != 0.f ? p.dq_mul : 1.0f) : 1.0f;
  if (tid < BNR) {                                                // visible after the first staging barrier; bias in the same two-plane layout as the tile
    sBias[fused_ln ? tid : ((tid >> 3) << 2) + (tid & 3) + ((tid & 4) ? PLANE1 : 0)] = p.bias ? p.bias[n0 + tid] : 0.f;
    if (fused_ln) { sGamma[tid] = p.ln_gamma[tid]; sBeta[tid] = p.ln_beta[tid]; }
    if (EPI == EPI_DGELU) sCol[tid] = 0.f;
  }
  if constexpr (EPI == EPI_RESID) {
    if (tid < BMR) { const int row = m0 + tid; sScale[tid] = (p.row_scale && row < p.M) ? p.row_scale[row / p.rows_per_seq] : 1.0f; }
  }
  float dg_col = 0.f;                                             // EPI_DGELU: this thread's column of the fc1 bias gradient
#pragma unroll
  for (int part = 0; part < NPART; ++part) {
    const int mi = HPP == 2 ? part : part >> 1, h = HPP == 2 ? 0 : part & 1;
    auto tile_row = [&](int rl) {                                 // staged row -> row of the block tile
      return HPP == 2 ? (rl >> 5) * (32 * MI) + mi * 32 + (rl & 31) : (rl >> 4) * (32 * MI) + mi * 32 + h * 16 + (rl & 15);
    };
#pragma unroll
    for (int ni = 0; ni < 3; ++ni)
#pragma unroll
      for (int r8 = 0; r8 < 8 * HPP; ++r8) {
        const int lrow = wm * (16 * HPP) + (r8 & 3) + 8 * (r8 >> 2) + 4 * hi;
        // Staging layout.  Fused LayerNorm: plain rows (pitch CLD), read back as f32x2 per lane -- conflict-free.  Other
        // epilogues read 8 consecutive columns per thread as two f32x4; in plain rows the 16-lane service groups of
        // ds_read_b128 then stride 32 B and hit every bank twice (SQ counters, round 1: 15-18 % of the LDS cycles of the bf16 /
        // GELU epilogues were bank conflicts).  There the low and the high four columns of every 8-column slot live in two
        // planes (columns 0-191 / 208-399 of a 448-float row): a service group reads 16 consecutive 16-B pieces of one plane.
        const int ccol = wn * 96 + ni * 32 + l31;
        const int pos = fused_ln ? lrow * CLD + ccol : lrow * CLD2 + ((ccol >> 3) << 2) + (ccol & 3) + ((ccol & 4) ? PLANE1 : 0);
        sC[pos] = F8 ? acc[mi][ni][h * 8 + r8] * dqv : acc[mi][ni][h * 8 + r8];
      }
    if (part == NPART / 2 - 1) STAMP2(20);
    // (Issuing these loads one part ahead was measured: no gain -- 219 vs 218 us on fc2+residual -- and 17 spilled
    // registers; a part's time is set by the CU's memory throughput, not by the exposed round trip.)
    EpiAux aux[fused_ln ? 1 : SLOTS];
    f32x2 rres[fused_ln ? LNROWS : 1][3];
    if constexpr (fused_ln) {
#pragma unroll
      for (int q = 0; q < LNROWS; ++q) {
        const int row = m0 + tile_row(wid * LNROWS + q);
        if (row < p.M) {
#pragma unroll
          for (int k = 0; k < 3; ++k)
            rres[q][k] = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(p.resid + (size_t)row * p.ldc + k * 128 + lane * 2));
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) {
        const int idx = tid + THREADS * i, row = m0 + tile_row(idx / 48);
        if (row < p.M) epi_fetch8<EPI, false>(p, row, n0 + (idx % 48) * 8, aux[i]);
      }
    }
    if (part == NPART / 2 - 1) STAMP2(21);
    __syncthreads();
    if (part == NPART / 2 - 1) STAMP2(22);
    if constexpr (fused_ln) {
      {
        // Fused residual + LayerNorm of the NEXT sub-layer (N == 384: the block owns whole rows): one wave per row,
        // x_new = resid + s (acc + bias) -> fp32 stream ; h = LN(x_new) -> bf16 operand of the next GEMM ; row statistics
        // saved for the LayerNorm backward.  Replaces a separate HBM pass (ln_fwd_kernel) over x.
#pragma unroll
        for (int q = 0; q < LNROWS; ++q) {
          const int rl = wid * LNROWS + q, trow = tile_row(rl), row = m0 + trow;
          if (row >= p.M) continue;
          const float sc = sScale[trow];
          float v[6];
          float sum = 0.f;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const int col = k * 128 + lane * 2;
            const f32x2 a2 = *reinterpret_cast<const f32x2*>(sC + rl * CLD + col);
            const f32x2 b2 = *reinterpret_cast<const f32x2*>(sBias + col);
            f32x2 o = rres[q][k] + sc * (a2 + b2);
            __builtin_nontemporal_store(o, reinterpret_cast<f32x2*>(reinterpret_cast<float*>(p.C) + (size_t)row * p.ldc + col));
            v[2 * k] = o[0]; v[2 * k + 1] = o[1]; sum += o[0] + o[1];
          }
          const float mu = wave_sum(sum) * (1.0f / 384.0f);
          float qd = 0.f;
#pragma unroll
          for (int k = 0; k < 6; ++k) { const float d = v[k] - mu; qd += d * d; }
          const float rs = rsqrtf(wave_sum(qd) * (1.0f / 384.0f) + 1e-6f);
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const int col = k * 128 + lane * 2;
            const f32x2 g2 = *reinterpret_cast<const f32x2*>(sGamma + col), be2 = *reinterpret_cast<const f32x2*>(sBeta + col);
            bf16x2 hv;
            hv[0] = f2bf((v[2 * k] - mu) * rs * g2[0] + be2[0]);
            hv[1] = f2bf((v[2 * k + 1] - mu) * rs * g2[1] + be2[1]);
            *reinterpret_cast<bf16x2*>(p.ln_out + (size_t)row * 384 + col) = hv;
          }
          if (lane == 0) { p.ln_mean[row] = mu; p.ln_rstd[row] = rs; }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) {
        const int idx = tid + THREADS * i, rl = idx / 48, c8 = (idx % 48) * 8;
        const int trow = tile_row(rl), row = m0 + trow;
#if ATST_ABLATE != 0 && ATST_ABLATE != 7
        if (sC[rl * CLD2 + (c8 >> 1)] != 12345.678f) continue;        // experiment builds: no epilogue stores
#endif
        f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = w0;
        if (row < p.M) {
          if constexpr (EPI == EPI_RESID) aux[i].s = sScale[trow];
          epilogue8<EPI>(p, row, n0 + c8, *reinterpret_cast<const f32x4*>(sC + rl * CLD2 + (c8 >> 1)), *reinterpret_cast<const f32x4*>(sC + rl * CLD2 + PLANE1 + (c8 >> 1)),
                         *reinterpret_cast<const f32x4*>(sBias + (c8 >> 1)), *reinterpret_cast<const f32x4*>(sBias + PLANE1 + (c8 >> 1)), aux[i], w0, w1);
        }
        if constexpr (EPI == EPI_DGELU) {
          // fc1 bias gradient = column sums of du.  The products go back into this thread's own staging slot (zeros for rows
          // beyond M); after a barrier 384 threads add up one column each over the 32 staged rows -- conflict-free in the
          // two-plane layout -- and keep the running sum in a register.  (One LDS atomic per element made this tile 2.7x
          // slower than the 128x128 kernel: 2624 vs 950 us at the base geometry.)
          if (p.colsum) {
            *reinterpret_cast<f32x4*>(sC + rl * CLD2 + (c8 >> 1)) = w0;
            *reinterpret_cast<f32x4*>(sC + rl * CLD2 + PLANE1 + (c8 >> 1)) = w1;
          }
        }
      }
      if constexpr (EPI == EPI_DGELU) {
        if (p.colsum) {
          __syncthreads();
          if (tid < BNR) {
            const int ph = ((tid >> 3) << 2) + (tid & 3) + ((tid & 4) ? PLANE1 : 0);
#pragma unroll 8
            for (int r = 0; r < RP; ++r) dg_col += sC[r * CLD2 + ph];
          }
        }
      }
    }
    if (part == NPART / 2 - 1) STAMP2(23);
    if (part < NPART - 1) __syncthreads();
    STAMP2(2 + part);
  }
  if constexpr (EPI == EPI_DGELU) {
    if (p.colsum && tid < BNR) atomicAdd(p.colsum + n0 + tid, dg_col);
  }
#if ATST_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP2(10);
#endif
}

// ---- 4-wave blocks, two per CU -----------------------------------------------------------------------------------------
// Same wave tile as the 256 x 384 kernel (128 x 96 = 12 accumulators, 256 registers), but FOUR waves per block so that two
// independent blocks share a CU (one wave of each per SIMD): while one block is in its epilogue (VALU + stores) or
// blocked on LDS-DMA issue, the other runs its MFMAs -- the phases that the one-block-per-CU kernel serialises
// (profiles/r02_trace_gemm.txt: 2608 cycles per k-tile against 1536 of MFMA time, then an epilogue with idle matrix pipes).
//   WM = 2: waves 2 x 2, block tile 256 x 192.  A row panel is staged by the blocks of both column halves (the second read
//           is an L2 hit), B half as often per row: the best operand mix measured (tools/probes/mix_probe: A 6.1 TB/s).
//   WM = 1: waves 1 x 4, block tile 128 x 384: whole rows, for the residual + LayerNorm epilogue.
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
  static constexpr int LDS = RING > EPI_BYTES ? RING : EPI_BYTES;
  static constexpr int A_WAVES = WM == 1 ? 1 : 2, B_WAVES = 4 - A_WAVES;
  static constexpr int PA = (BM / 16) / A_WAVES, PB = (BNB / 16) / B_WAVES;   // 1-KiB pieces per loader wave per k-tile: 8, 8 ; 8, 6
  static constexpr int SPR = BNB / 8;                                        // 8-column slots per row
};
}

template <int EPI, int WM, bool LN = false>
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
  if (blockIdx.x >= 256 && blockIdx.x < 512) phase_skew(p.skew);                      // two blocks per CU: the second of each pair

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
  float* sC = reinterpret_cast<float*>(smem_raw);
  float* sBias = sC + G::RP * G::CLD2;
  float* sGamma = sBias + G::CLD2; float* sBeta = sGamma + BNB;
  float* sScale = sBeta + BNB;
  float* sCol = sGamma;                                           // EPI_DGELU column sums (no LN there)
  constexpr int NPART = 2 * MI;
  constexpr bool fused_ln = LN && EPI == EPI_RESID;
  for (int c = tid; c < BNB; c += 256) {
    sBias[fused_ln ? c : ((c >> 3) << 2) + (c & 3) + ((c & 4) ? G::PL1 : 0)] = p.bias ? p.bias[n0 + c] : 0.f;
    if (fused_ln) { sGamma[c] = p.ln_gamma[c]; sBeta[c] = p.ln_beta[c]; }
    if (EPI == EPI_DGELU) sCol[c] = 0.f;
  }
  if constexpr (EPI == EPI_RESID) {
    if (tid < BM) { const int row = m0 + tid; sScale[tid] = (p.row_scale && row < p.M) ? p.row_scale[row / p.rows_per_seq] : 1.0f; }
  }
#pragma unroll
  for (int part = 0; part < NPART; ++part) {
    const int mi = part >> 1, h = part & 1;
    auto tile_row = [&](int rl) { return (rl >> 4) * 128 + mi * 32 + h * 16 + (rl & 15); };   // staged row -> row of the block tile
#pragma unroll
    for (int ni = 0; ni < 3; ++ni)
#pragma unroll
      for (int r8 = 0; r8 < 8; ++r8) {
        const int lr = wm * 16 + (r8 & 3) + 8 * (r8 >> 2) + 4 * hi;
        const int ccol = wn * 96 + ni * 32 + l31;
        sC[fused_ln ? lr * CLD + ccol : lr * G::CLD2 + ((ccol >> 3) << 2) + (ccol & 3) + ((ccol & 4) ? G::PL1 : 0)] = acc[mi][ni][h * 8 + r8];
      }
    EpiAux aux[fused_ln ? 1 : 3];
    f32x2 rres[fused_ln ? 4 : 1][3];
    if constexpr (fused_ln) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = m0 + tile_row(wid * 4 + q);
        if (row < p.M) {
#pragma unroll
          for (int k = 0; k < 3; ++k)
            rres[q][k] = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(p.resid + (size_t)row * p.ldc + k * 128 + lane * 2));
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int idx = tid + 256 * i, row = m0 + tile_row(idx / G::SPR);
        if (row < p.M) epi_fetch8<EPI, false>(p, row, n0 + (idx % G::SPR) * 8, aux[i]);
      }
    }
    __syncthreads();
    if constexpr (fused_ln) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rl = wid * 4 + q, trow = tile_row(rl), row = m0 + trow;
        if (row >= p.M) continue;
        const float sc = sScale[trow];
        float v[6];
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int col = k * 128 + lane * 2;
          const f32x2 a2 = *reinterpret_cast<const f32x2*>(sC + rl * CLD + col);
          const f32x2 b2 = *reinterpret_cast<const f32x2*>(sBias + col);
          f32x2 o = rres[q][k] + sc * (a2 + b2);
          __builtin_nontemporal_store(o, reinterpret_cast<f32x2*>(reinterpret_cast<float*>(p.C) + (size_t)row * p.ldc + col));
          v[2 * k] = o[0]; v[2 * k + 1] = o[1]; sum += o[0] + o[1];
        }
        const float mu = wave_sum(sum) * (1.0f / 384.0f);
        float qd = 0.f;
#pragma unroll
        for (int k = 0; k < 6; ++k) { const float d = v[k] - mu; qd += d * d; }
        const float rs = rsqrtf(wave_sum(qd) * (1.0f / 384.0f) + 1e-6f);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int col = k * 128 + lane * 2;
          const f32x2 g2 = *reinterpret_cast<const f32x2*>(sGamma + col), be2 = *reinterpret_cast<const f32x2*>(sBeta + col);
          bf16x2 hv;
          hv[0] = f2bf((v[2 * k] - mu) * rs * g2[0] + be2[0]);
          hv[1] = f2bf((v[2 * k + 1] - mu) * rs * g2[1] + be2[1]);
          *reinterpret_cast<bf16x2*>(p.ln_out + (size_t)row * 384 + col) = hv;
        }
        if (lane == 0) { p.ln_mean[row] = mu; p.ln_rstd[row] = rs; }
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
    if (part < NPART - 1) __syncthreads();
  }
  if constexpr (EPI == EPI_DGELU) {
    if (p.colsum) {
      __syncthreads();
      for (int c = tid; c < BNB; c += 256) atomicAdd(p.colsum + n0 + c, sCol[c]);
    }
  }
}

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
// ds_read_b64_tr_b16 -- the hardware transpose that turns the m-major image into MFMA fragments -- fall on distinct banks.  36.9 KB staged per 64 rows of a 192x384 tile = 8.2 KB per 128x128 unit, against
// 16 KB for the square tile: the wgrad GEMMs are bound by that L2 -> LDS traffic.
#ifndef ATST_TN_RM
#define ATST_TN_RM 64          // contraction rows per ring stage: 64 (2-stage ring) | 32 (4-stage ring: measured SLOWER, 268 vs 225 us on fc2 wgrad)
#endif
namespace tnt {
constexpr int TN = 192, TK = 384, RM = ATST_TN_RM;
constexpr int PY = TN * 2, PX = TK * 2;                               // row pitches in bytes
constexpr int Y_BYTES = RM * PY, X_BYTES = RM * PX, STAGE = Y_BYTES + X_BYTES;   // RM 64: 24,576 + 49,152 ; RM 32: half
// Both operands of a weight gradient are streamed from HBM; with 64-row stages only two fit (144 KB), i.e. one stage of
// prefetch.  Tried (round 2): 32-row stages in a 4-deep ring (three stages = 108 KB in flight, same LDS): SLOWER -- fc2 wgrad
// 225 -> 268 us, the grouped launch 320 -> 376 us: twice the barriers and 4-5 instead of 9 LDS-DMA pieces per wave per stage
// cost more than the deeper prefetch returns.  Kept selectable (ATST_TN_RM=32).
constexpr int NST = RM == 64 ? 2 : 4;
constexpr int LDS = NST * STAGE;                                      // 147,456 B either way
constexpr int Y_PIECES = Y_BYTES / 1024, PIECES = STAGE / 1024;       // 1-KiB LDS-DMA pieces per stage: 12 + 24 (RM 32) ; 24 + 48
#ifndef ATST_TN_ILV
#define ATST_TN_ILV 0          // 1: the next stage's LDS-DMA pieces are issued one per MFMA group instead of all after the barrier
#endif
#ifndef ATST_TN_SPLIT
#define ATST_TN_SPLIT 0        // 1: only waves 4-7 issue the LDS-DMA of the next stage while waves 0-3 already run the stage's MFMAs (measured: no gain, see below)
#endif
constexpr int LOADERS = ATST_TN_SPLIT ? 4 : 8, LOADER0 = ATST_TN_SPLIT ? 4 : 0;
constexpr int PPW = (PIECES + LOADERS - 1) / LOADERS;                 // pieces per loader wave per stage: 18 (split) / 9 (RM 64)
}

// ds_read_b64_tr_b16 is serviced 32 lanes at a time = 4 rows x 64 B of the image, over 64 banks (256 B): the four rows
// must land in four different 64-B windows.  768-B pitch (== 0 mod 256): window index ^= row & 3; 384-B pitch (rows
// alternate between offsets 0 and 128): window index ^= (row >> 1) & 1.
template <int P> DEVFN int tr_swz(int row) { return P % 256 == 0 ? (row & 3) << 2 : ((row >> 1) & 1) << 2; }
template <int P>
DEVFN bf16x8 ld_frag_tr_p(const char* X, int r0, int c0, int lane) {
  const int a = lane & 15, g = lane >> 4;
  const int row = r0 + 4 * (g >> 1) + (a >> 2);
  const int col = c0 + (g & 1) * 16 + 4 * (a & 3);                 // element column; 8 elements per 16-B chunk
  const int pch = (col >> 3) ^ tr_swz<P>(row);
  const bf16* ptr = reinterpret_cast<const bf16*>(X + row * P + pch * 16) + (col & 7);
  s16x4 lo = lds_tr4(ptr);
  s16x4 hi = lds_tr4(ptr + 8 * (P / 2));                           // +8 rows: same swizzle key
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = lo; u.s.b = hi;
  return u.v;
}

#if ATST_TRACE
__device__ unsigned long long g_tn_trc[8 * 260 * 4];            // [wave][stage][0: loop top, 1: after wait, 2: after barrier + issue, 3: after MFMAs]
#define TSTAMP(i) do { if (ttrace && st < 260) g_tn_trc[(wid * 260 + st) * 4 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TSTAMP(i) do { } while (0)
#endif
// one (tile, split) of one problem
DEVFN void tn_tall_body(const WgradArgs& p, int tile, int split, char* smem_raw) {
  using namespace tnt;
  typedef const void __attribute__((address_space(1))) * gptr_t;
  typedef void __attribute__((address_space(3))) * lptr_t;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wn = wid >> 2, wk = wid & 3, hi = lane >> 5, l31 = lane & 31;
  const int ntk = p.K / TK;
  const int n0 = (tile / ntk) * TN, k0 = (tile % ntk) * TK;
  const int m_begin = split * p.m_per_split;
  int m_end = m_begin + p.m_per_split; if (m_end > p.M) m_end = p.M;
  if (m_begin >= m_end) return;
  const int nst = (m_end - m_begin) / RM;

  // lane -> (row, chunk) of the linear stage image.  Piece q (1 KiB) of a stage: q < Y_PIECES -> dY image, else X image;
  // loader wave w issues pieces w', w' + LOADERS, ...
  // Stage timeline measured with every wave issuing its 9 pieces right after the barrier (tools/trace_tn.py): 1.5-2.0 k cycles
  // of LDS-DMA issue with idle matrix pipes, then 2.6-3.3 k cycles for the 2 x 36 MFMAs of a SIMD: 5.3 k per 64 rows.  With the
  // issue moved to waves 4-7 alone, waves 0-3 compute meanwhile and the two waves of a SIMD no longer contend for its pipe.
  const bool loader = wid >= LOADER0;
  int off[PPW]; bool isx[PPW]; bool have[PPW]; int ldsoff[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int q = (wid - LOADER0) + LOADERS * j;
    have[j] = loader && q < PIECES;
    isx[j] = q >= Y_PIECES;
    if (!isx[j]) {
      const int c_ = q * 64 + lane, row = c_ / 24, c = (c_ % 24) ^ tr_swz<PY>(row);
      off[j] = row * p.ldy + n0 + c * 8; ldsoff[j] = q * 1024;
    } else {
      const int c_ = (q - Y_PIECES) * 64 + lane, row = c_ / 48, c = (c_ % 48) ^ tr_swz<PX>(row);
      off[j] = row * p.ldx + k0 + c * 8; ldsoff[j] = Y_BYTES + (q - Y_PIECES) * 1024;
    }
  }
  const bf16* baseY = p.dY + (size_t)m_begin * p.ldy;
  const bf16* baseX = p.X + (size_t)m_begin * p.ldx;
  auto issue_piece = [&](int st, int j) {
    char* buf = smem_raw + (st % NST) * STAGE;
    if (!have[j]) return;
    const bf16* src = isx[j] ? baseX + (size_t)st * RM * p.ldx + off[j] : baseY + (size_t)st * RM * p.ldy + off[j];
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(buf + ldsoff[j]), 16, 0, 0);
  };
  auto issue_stage = [&](int st) {
#pragma unroll
    for (int j = 0; j < PPW; ++j) issue_piece(st, j);
  };
  f32x16 acc[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
  for (int s_ = 0; s_ < NST - 1; ++s_)
    if (s_ < nst) issue_stage(s_);
  constexpr int MINPW = PIECES / LOADERS;                         // loads per stage of the loader waves that issue the fewest
#if ATST_TRACE
  const bool ttrace = (int)blockIdx.x == 100 && lane == 0;
#endif
  for (int st = 0; st < nst; ++st) {
    TSTAMP(0);
    // stage st has landed (my loads retire in order; up to NST - 2 younger stages stay in flight; waves that issue one piece
    // more per stage simply wait for a little of stage st + 1 as well); barrier => for everyone, and stage st - 1's buffer is free
    const int younger = nst - 1 - st < NST - 2 ? nst - 1 - st : NST - 2;
    if (younger >= 2) {
      if constexpr (MINPW * 2 == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (younger == 1) {
      if constexpr (MINPW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    TSTAMP(1);
    asm volatile("s_barrier" ::: "memory");
    const bool more = st + NST - 1 < nst;
#if !ATST_TN_ILV
    if (more) issue_stage(st + NST - 1);
#endif
    TSTAMP(2);
    const char* sY = smem_raw + (st % NST) * STAGE; const char* sX = sY + Y_BYTES;
    int slot = 0;
#pragma unroll
    for (int ms = 0; ms < RM / 16; ++ms) {
      bf16x8 a[3], b[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) a[i] = ld_frag_tr_p<PY>(sY, ms * 16, wn * 96 + i * 32, lane);
#pragma unroll
      for (int i = 0; i < 3; ++i) b[i] = ld_frag_tr_p<PX>(sX, ms * 16, wk * 96 + i * 32, lane);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
#if ATST_TN_ILV
        if (slot < PPW) {                                         // next stage's LDS-DMA pieces spread between the MFMA groups
          __builtin_amdgcn_sched_barrier(0);
          if (more) issue_piece(st + NST - 1, slot);
          __builtin_amdgcn_sched_barrier(0);
          ++slot;
        }
#endif
      }
    }
    TSTAMP(3);
  }
#if ATST_TRACE
  { const int st = 259; TSTAMP(0); }
#endif
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + wn * 96 + i * 32 + crow32(r, hi);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
#if ATST_ABLATE == 8
        if (acc[i][j][r] != 12345.678f) continue;                   // experiment builds: no atomics
#endif
        atomicAdd(p.dW + (size_t)n * p.ldw + k0 + wk * 96 + j * 32 + l31, acc[i][j][r]);
      }
    }
}


__global__ __launch_bounds__(512, 2) void gemm_tn_tall_kernel(WgradArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int ntiles = (p.N / tnt::TN) * (p.K / tnt::TK);
  const int id = xcd_remap(blockIdx.x, gridDim.x);                // all tiles of one M-split on one XCD (they stream the same rows)
  tn_tall_body(p, id % ntiles, id / ntiles, smem_raw);
}

// Several weight gradients in ONE launch (the four of a transformer block): the grid is one round of the chip however
// many problems share it, so each problem needs 1/n-th of the M-splits it would need alone -- and the fp32 atomics that
// combine the splits (measured ~1.5 TB/s, 45-50 us per GEMM when each is launched alone) shrink by the same factor.
struct WgradGroup { WgradArgs it[ATST_WGRAD_GROUP_MAX]; int first_tile[ATST_WGRAD_GROUP_MAX + 1]; int n; };
__global__ __launch_bounds__(512, 2) void gemm_tn_tall_group_kernel(WgradGroup g) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int ntiles = g.first_tile[g.n];
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int t = id % ntiles, split = id / ntiles;
  int k = 0;
#pragma unroll
  for (int i = 1; i < ATST_WGRAD_GROUP_MAX; ++i) if (i < g.n && t >= g.first_tile[i]) k = i;
  tn_tall_body(g.it[k], t - g.first_tile[k], split, smem_raw);
}

int g_tn_tall = 1;        // wgrad: 192 x 384 LDS-DMA tile when N % 192 == 0, K % 384 == 0, M % 64 == 0 (tuning hook 105 = off, 106 = on)
int g_row384_auto = 1;      // use the 128x384 tile whenever N % 384 == 0 (tuning hook 300 turns it off)
int g_row384_tall = 2;      // 256 x 384 tiles for M >= 8192: 2 = every epilogue, 1 = plain bf16 GEMMs only, 0 = never (tuning hooks 304 / 303 / 302)
int g_tn_rounds = 1;      // wgrad grid = this many rounds of 512 resident blocks (tuning hook 110 + r); 1 measured best (-20 %)
int g_row384_dir = 0;       // 256-row tile: epilogue straight from the accumulator registers (tuning hook 341 = on)
int g_row384_pp = 0;        // ping-pong main loop of the 256-row tile (tuning hook 321 = on)
int g_row384_bk64 = 0;      // 256-row tile with 64-deep ring stages, whole 128-B lines per LDS-DMA lane group, 2 stages (tuning hook 311 = on): measured 2-5 % slower than 3 x 32-deep
int g_dgelu_row384 = 2;   // dGELU GEMM on the 256x384 tile: 0 never / 1 always / 2 when K >= 768 (hooks 306 / 307 / 308).  Measured with the staged column sums: base (K = 768) 968 -> 935 us, small (K = 384) 206 -> 227 us; with per-element LDS atomics it was 2.7x slower
int g_nt_variant = -1;    // -1 auto ; 0: 128x128 2-stage ; 1: 128x128 3-stage ; 2: 256x128 8 waves ; 3: 256x128 4 waves of 128x64

// Algorithmic HBM bytes of one nt GEMM: both operands once, every epilogue input once, every output once.
template <int EPI>
double nt_bytes(const GemmArgs& a) {
  const double mn = (double)a.M * a.N;
  double b = 2.0 * a.K * ((double)a.M + a.N);
  if (EPI == EPI_BF16) b += 2.0 * mn;
  if (EPI == EPI_F32) b += 4.0 * mn;
  if (EPI == EPI_BIAS_GELU) b += 2.0 * mn + (a.C2 ? 2.0 * mn : 0.0);
  if (EPI == EPI_RESID) b += 8.0 * mn + (a.ln_out ? 2.0 * mn : 0.0);
  if (EPI == EPI_DGELU) b += 4.0 * mn;
  if (EPI == EPI_PATCH) b += 4.0 * mn;
  return b;
}

template <int EPI, int BMT, int NSTG, int WTM>
int launch_nt_cfg(const GemmArgs& a, hipStream_t st) {
  using G = NtGeo<BMT, NSTG, WTM>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_kernel<EPI, BMT, NSTG, WTM>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  const int nblk = ((a.M + BMT - 1) / BMT) * (a.N / BN);
  ProfScope ps(PK_GEMM_NT0 + EPI, 2.0 * a.M * a.N * a.K, st, nt_bytes<EPI>(a));
  hipLaunchKernelGGL((gemm_nt_kernel<EPI, BMT, NSTG, WTM>), dim3(nblk), dim3(G::THREADS), G::LDS, st, a);
  return (int)hipGetLastError();
}
template <int EPI, int MI, bool LN, int BKT = BK, bool PP = false, bool DIR = false, bool F8 = false>
int launch_nt_row384_cfg(const GemmArgs& a, hipStream_t st) {
  using RG = row384::Geo<MI, BKT>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_row384_kernel<EPI, MI, LN, BKT, PP, DIR, F8>, hipFuncAttributeMaxDynamicSharedMemorySize, RG::LDS);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  const int nblk = ((a.M + RG::BMR - 1) / RG::BMR) * (a.N / row384::BNR);
  hipLaunchKernelGGL((gemm_nt_row384_kernel<EPI, MI, LN, BKT, PP, DIR, F8>), dim3(nblk), dim3(row384::THREADS), RG::LDS, st, a);
  return (int)hipGetLastError();
}
int g_w4_auto = 1;          // hook 360 / 361: 4-wave kernels for launches of fewer than 1.5 rounds of 256 x 384 tiles
int g_skew = 0;             // hook 100000 + c: phase skew in shader cycles per k-tile of the main loop (0 = off)
int g_w4_min_m = 8192;     // hook 351: use the 4-wave kernels for every M (parity tests run small shapes)
int g_w4_mode = 0;          // 4-wave two-blocks-per-CU kernels (tuning hook 330 + m): 0 off ; 1 = 128x384 for everything ; 2 = 256x192 (plain epilogues) + 128x384 (fused LayerNorm) ; 3 = 256x192 for the plain epilogues only
template <int EPI, int WM, bool LN>
int launch_nt_w4_cfg(const GemmArgs& a, hipStream_t st) {
  using G = w4::Geo<WM>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_w4_kernel<EPI, WM, LN>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  const int nblk = ((a.M + G::BM - 1) / G::BM) * (a.N / G::BNB);
  hipLaunchKernelGGL((gemm_nt_w4_kernel<EPI, WM, LN>), dim3(nblk), dim3(256), G::LDS, st, a);
  return (int)hipGetLastError();
}
template <int EPI>
int launch_nt_row384(const GemmArgs& a0, hipStream_t st) {
  GemmArgs a = a0;
  a.skew = g_skew * (a.K / BK);
  if (a.fp8) {                                                    // e4m3 operands seen as byte pairs: K, lda, ldb are already halved
    if constexpr (EPI == EPI_BF16 || EPI == EPI_BIAS_GELU || EPI == EPI_RESID || EPI == EPI_F32) {
      ProfScope ps(PK_GEMM_NT0 + EPI, 4.0 * a.M * a.N * a.K, st, nt_bytes<EPI>(a));
      return launch_nt_row384_cfg<EPI, 4, false, BK, false, false, true>(a, st);
    } else {
      return ATST_EINVAL;
    }
  }
  ProfScope ps(PK_GEMM_NT0 + EPI, 2.0 * a.M * a.N * a.K, st, nt_bytes<EPI>(a));

  // Fewer than 1.5 rounds of 256 x 384 tiles (the 1 s local views: M = 32768, N = 384 -> 128 blocks on 256 CUs): the 4-wave
  // kernels launch twice as many, half as large blocks, two per CU.  Measured at M = 32768 (profiles/r02_gemm_bench.txt):
  // proj+resid 49 -> 36 us, fc2+resid 96 -> 75, fc1 dgrad 67 -> 53, qkv dgrad 48 -> 39, qkv fwd 47 -> 43; at M = 131072 the
  // 8-wave tile is 10-25 % faster (B is staged once per 256 rows), and fc1+GELU (N = 1536) is never better on 4 waves.
  const long tiles8 = (long)((a.M + 255) / 256) * (a.N / 384);
  const bool small_grid = g_w4_auto && a.M >= 2048 && tiles8 <= 384 && EPI != EPI_BIAS_GELU && EPI != EPI_DGELU && EPI != EPI_PATCH;
  if ((g_w4_mode && a.M >= g_w4_min_m) || small_grid) {
    const int w4m = g_w4_mode ? g_w4_mode : 2;
    if constexpr (EPI == EPI_RESID) {
      if (a.ln_out) { if (w4m != 3) return launch_nt_w4_cfg<EPI, 1, true>(a, st); }
    }
    if constexpr (ATST_EXPERIMENTS) {
      if (w4m == 1 && !a.ln_out) return launch_nt_w4_cfg<EPI, 1, false>(a, st);
    }
    if (!a.ln_out) return launch_nt_w4_cfg<EPI, 2, false>(a, st);
  }
  // 256-row tiles: the operand ring of one block covers twice the output (6.7 vs 10.7 KB staged per 128x128 unit)
  const bool tall = a.M >= (g_w4_min_m < 8192 ? g_w4_min_m : 8192) && (g_row384_tall == 2 || (g_row384_tall == 1 && EPI == EPI_BF16 && (a.K >= 768 || a.N >= 768)));
  const bool deep = tall && g_row384_bk64 && a.K % 64 == 0;        // 64-deep ring stages
  if constexpr (ATST_EXPERIMENTS && (EPI == EPI_BF16 || EPI == EPI_F32 || EPI == EPI_BIAS_GELU || EPI == EPI_RESID)) {
    if (tall && g_row384_dir && ATST_TALL_STAGES == 3) {            // epilogue straight from the registers (tuning hook 340 = off / 341 = on)
      if constexpr (EPI == EPI_RESID) {
        if (a.ln_out) return launch_nt_row384_cfg<EPI, 4, true, BK, false, true>(a, st);
      }
      return launch_nt_row384_cfg<EPI, 4, false, BK, false, true>(a, st);
    }
  }
  if constexpr (ATST_EXPERIMENTS) {
    if (tall && g_row384_pp && ATST_TALL_STAGES == 3) {             // ping-pong main loop (tuning hook 320 / 321)
      if constexpr (EPI == EPI_RESID) {
        if (a.ln_out) return launch_nt_row384_cfg<EPI, 4, true, BK, true>(a, st);
      }
      return launch_nt_row384_cfg<EPI, 4, false, BK, true>(a, st);
    }
    if (deep) {
      if constexpr (EPI == EPI_RESID) {
        if (a.ln_out) return launch_nt_row384_cfg<EPI, 4, true, 64>(a, st);
      }
      return launch_nt_row384_cfg<EPI, 4, false, 64>(a, st);
    }
  }
  if constexpr (EPI == EPI_RESID) {
    if (a.ln_out) return tall ? launch_nt_row384_cfg<EPI, 4, true>(a, st) : launch_nt_row384_cfg<EPI, 2, true>(a, st);
  }
  return tall ? launch_nt_row384_cfg<EPI, 4, false>(a, st) : launch_nt_row384_cfg<EPI, 2, false>(a, st);
}
template <int EPI>
int launch_nt(const GemmArgs& a, hipStream_t st) {
  if (a.ln_out) {                                                 // fused LayerNorm needs the block to own whole rows
    if (EPI != EPI_RESID || a.N != 384 || !a.ln_gamma || !a.ln_beta || !a.ln_mean || !a.ln_rstd) return ATST_EINVAL;
    if constexpr (EPI == EPI_RESID) return launch_nt_row384<EPI>(a, st);
  }
  int v = g_nt_variant;
  if ((v == 4 || (v < 0 && g_row384_auto)) && a.N % 384 == 0 && (EPI != EPI_DGELU || g_dgelu_row384 == 1 || (g_dgelu_row384 == 2 && a.K >= 768))) return launch_nt_row384<EPI>(a, st);
  if (v < 0) v = (EPI == EPI_F32 && a.M >= 16384) ? 3            // ATST-Frame head Linears (83 k rows): 256x128 tile, -13 %
               : a.K <= 512 ? 0 : 1;
  if (v == 3) return launch_nt_cfg<EPI, 256, 3, 128>(a, st);
  if constexpr (ATST_EXPERIMENTS) {
    if (v == 2) return launch_nt_cfg<EPI, 256, 3, 64>(a, st);
  }
  if (v == 1) return launch_nt_cfg<EPI, 128, 3, 64>(a, st);
  return launch_nt_cfg<EPI, 128, 2, 64>(a, st);
}

}  // namespace

void atst_gemm_nt_set_variant(int v) { if (v >= 100000) g_skew = v - 100000; else if (v >= 360) g_w4_auto = v - 360; else if (v >= 350) g_w4_min_m = v == 351 ? 1 : 8192; else if (v >= 340) g_row384_dir = v - 340; else if (v >= 330) g_w4_mode = v - 330; else if (v >= 320) g_row384_pp = v - 320; else if (v >= 310) g_row384_bk64 = v - 310; else if (v >= 306) g_dgelu_row384 = v - 306; else if (v >= 302) g_row384_tall = v - 302; else if (v >= 300) g_row384_auto = v - 300; else if (v >= 110) g_tn_rounds = v - 110; else if (v >= 105) g_tn_tall = v - 105; else if (v < 100) g_nt_variant = v; }

int atst_gemm_nt(const GemmArgs& a0, hipStream_t st) {
  GemmArgs a = a0;
  if (a.fp8) {                                                    // e4m3: K a multiple of 64 bytes, row-384 tile only, no fused LayerNorm
    if (a.M <= 0 || a.N % 384 || a.K % 64 || a.lda % 16 || a.ldb % 16 || a.ln_out) return ATST_EINVAL;
    a.K /= 2; a.lda /= 2; a.ldb /= 2;
    switch (a.epi) {
      case EPI_BF16: return launch_nt_row384<EPI_BF16>(a, st);
      case EPI_F32: return launch_nt_row384<EPI_F32>(a, st);
      case EPI_BIAS_GELU: return launch_nt_row384<EPI_BIAS_GELU>(a, st);
      case EPI_RESID: return launch_nt_row384<EPI_RESID>(a, st);
    }
    return ATST_EINVAL;
  }
  if (a.M <= 0 || a.N % BN || a.K % BK || a.lda % 8 || a.ldb % 8) return ATST_EINVAL;
  switch (a.epi) {
    case EPI_BF16: return launch_nt<EPI_BF16>(a, st);
    case EPI_F32: return launch_nt<EPI_F32>(a, st);
    case EPI_BIAS_GELU: return launch_nt<EPI_BIAS_GELU>(a, st);
    case EPI_RESID: return launch_nt<EPI_RESID>(a, st);
    case EPI_DGELU: return launch_nt<EPI_DGELU>(a, st);
    case EPI_PATCH: return launch_nt<EPI_PATCH>(a, st);
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
    mps = ((mps + tnt::RM - 1) / tnt::RM) * tnt::RM;
    p.m_per_split = mps;
    const int nblk = tiles * ((a.M + mps - 1) / mps);
    ProfScope ps(PK_GEMM_TN, 2.0 * p.M * p.N * p.K, st, 2.0 * p.M * ((double)p.N + p.K) + 4.0 * p.N * p.K);
    static bool tall_attr = false;
    if (!tall_attr) {
      hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_tall_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, tnt::LDS);
      if (e != hipSuccess) return (int)e;
      tall_attr = true;
    }
    hipLaunchKernelGGL(gemm_tn_tall_kernel, dim3(nblk), dim3(512), tnt::LDS, st, p);
    return (int)hipGetLastError();
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
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WGRAD_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(nblk), dim3(256), WGRAD_LDS_BYTES, st, p);
  return (int)hipGetLastError();
}

bool tn_tall_ok(const WgradArgs& a) {
  return g_tn_tall && a.N % tnt::TN == 0 && a.K % tnt::TK == 0 && a.M % tnt::RM == 0 && a.M >= 8192 && a.ldy % 8 == 0 && a.ldx % 8 == 0;
}

#if ATST_TRACE
extern "C" int atst_debug_tn_trace(unsigned long long* host, int n) {   // trace builds only (tools/trace_tn.py)
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tn_trc), sizeof(unsigned long long) * (size_t)n);
}
#endif
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
  int splits = 256 / tiles; if (splits < 1) splits = 1;          // one block per CU, one round
  int max_splits = 1;
  for (int i = 0; i < n; ++i) {
    int mps = (items[i].M + splits - 1) / splits;
    mps = ((mps + tnt::RM - 1) / tnt::RM) * tnt::RM;
    g.it[i].m_per_split = mps;
    const int sp = (items[i].M + mps - 1) / mps;
    if (sp > max_splits) max_splits = sp;
  }
  ProfScope ps(PK_GEMM_TN, flops, st, bytes);
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_tall_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, tnt::LDS);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  hipLaunchKernelGGL(gemm_tn_tall_group_kernel, dim3(tiles * max_splits), dim3(512), tnt::LDS, st, g);
  return (int)hipGetLastError();
}

