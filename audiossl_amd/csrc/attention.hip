// Fused multi-head self-attention (head_dim 64, <= 256 tokens / sequence) forward + backward for gfx950.
//
// Reference math (audiossl/modules/transformer.py:107-121, 152-159):
//     attn = softmax(q k^T * hd^-0.5 + mask) ; mask[b,.,.,j] = -10000 * (j >= length[b]) ; y = attn v
// The [S,H,N,N] probability tensor and the [S,1,N,N] mask of the reference are never materialised: K and V of one
// (sequence, head) live in LDS (2 x 36 KB at N=256), scores stay in MFMA accumulators, the key-padding bias is computed
// from `valid[s]` in registers.  Key tiles that lie completely beyond valid[s] are skipped: their reference weight is
// exp(-10000 + ...) == 0 exactly in fp32.
//
// All three kernels use the "swapped" orientation S^T = K Q^T so that one lane owns one query column: row max / sum are
// in-lane over 16 registers + one cross-half shuffle, and the probability registers are re-used directly as the B
// operand of the second MFMA (no LDS round trip).  Transposed operands (V^T, K^T, Q^T, dO^T) are produced by
// ds_read_b64_tr_b16 from the row-major LDS images.
//
//   forward            : wave = 32 queries ; loop over key tiles of 32 ; online softmax ; O^T += V^T P^T
//   backward, dK/dV    : wave = 32 keys    ; loop over query tiles  ; dV^T += dO^T P ; dK^T += Q^T dS
//   backward, dQ       : wave = 32 queries ; loop over key tiles    ; dQ^T += K^T dS^T
// (S and dP are recomputed in both backward kernels instead of exchanging dS through LDS or atomics.)
#include "common.h"
#include "kernels.h"
#include "profile.h"

namespace {

constexpr int HD = 64;
constexpr int A_LD = HD + 8;          // 72 bf16 = 144 B LDS row stride
constexpr float MASK_NEG = -10000.0f;
constexpr float LOG2E = 1.4426950408889634f;

template <int NP> struct Geo {
  static constexpr int CHUNKS = NP / 32;                          // 32-row tiles per sequence
  static constexpr int G = CHUNKS >= 4 ? 1 : 4 / CHUNKS;          // (sequence, head) pairs per 4-wave block
  static constexpr int WPP = 4 / G;                               // waves per pair
  static constexpr int CPW = CHUNKS / WPP;                        // tiles per wave
  static constexpr int LDS_MAT = NP * A_LD;                       // elements of one staged [NP][72] matrix
};

// stage rows [0,NP) x 64 columns (col0..) of a [*, ld] bf16 matrix into sm[NP][A_LD] with `nthr` threads; rows >= rs (packed
// sequences: they belong to the next sequence) are staged as zeros, i.e. exactly what a padded layout holds there
template <int NP>
DEVFN void stage_rows(bf16* sm, const bf16* g, size_t ld, int t, int nthr, int rs) {
  const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int c = t; c < NP * 8; c += nthr) {
    const int r = c >> 3, k = (c & 7) * 8;
    *reinterpret_cast<bf16x8*>(sm + r * A_LD + k) = r < rs ? ld_frag(g + (size_t)r * ld + k) : z;
  }
}
DEVFN bf16x8 ld_frag_if(bool live, const bf16* p) { const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0}; return live ? ld_frag(p) : z; }

// v_exp_f32 as is.  exp2f() adds a compare / two selects / ldexp around it for results in the denormal range; every use
// here has an argument <= ~0 whose underflow to 0 is the wanted result.
DEVFN float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Store one 64-wide bf16 row per lane pair from two C-layout accumulator tiles (t0: columns 0..31, t1: 32..63 of the row
// owned by lane & 31).  A lane holds 4-element groups {8g + 4 hi .. +3}; v_permlane32_swap trades groups between the
// two half-waves so that every lane ends up with two complete 8-element (16-B) pieces per tile: 4 dwordx4 stores per
// row instead of 8 dwordx2 (the row-per-lane store tail is issue-bound: -18 % on the forward kernel).
#ifndef ATST_ATTN_NT              // build-time A/B of the store cache policy: bit 0 row-per-lane stores (store_row64), bit 1 full-line stores (store_tile64_staged)
#define ATST_ATTN_NT 0
#endif
DEVFN unsigned pack2(float a, float b) { bf16x2 t; t[0] = f2bf(a); t[1] = f2bf(b); return __builtin_bit_cast(unsigned, t); }
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
DEVFN void store_row64(bf16* row, const f32x16& t0, const f32x16& t1, float mul, int hi) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const f32x16& v = t == 0 ? t0 : t1;
    unsigned P[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      P[g][0] = pack2(v[4 * g] * mul, v[4 * g + 1] * mul);
      P[g][1] = pack2(v[4 * g + 2] * mul, v[4 * g + 3] * mul);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        auto r = __builtin_amdgcn_permlane32_swap(P[k][w], P[k + 2][w], false, false);   // P[k].upper <-> P[k+2].lower
        P[k][w] = r[0]; P[k + 2][w] = r[1];
      }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      u32x4 o = {P[k][0], P[k][1], P[k + 2][0], P[k + 2][1]};
      if constexpr (ATST_ATTN_NT & 1) __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(row + t * 32 + 16 * hi + 8 * k));
      else *reinterpret_cast<u32x4*>(row + t * 32 + 16 * hi + 8 * k) = o;
    }
  }
}

// Full-line stores of a 32-row x 64-column bf16 tile held as two C-layout accumulators (row = lane & 31; t0: columns 0..31, t1: 32..63):
// the tile is transposed through a wave-private 4 KB LDS region (16-B chunks XOR-permuted by the row: conflict-free both ways) so that
// every store instruction writes 8 complete 128-B lines instead of 32-B pieces of 32 different lines (store_row64: the row-per-lane
// pattern costs the NP = 256 kernels the full store bandwidth of the chip at that granularity, profiles/r04_attn_ablate.txt).
DEVFN void store_tile64_staged(bf16* g00 /* global address of (row 0, column 0) */, size_t ld /* elements between rows */, const f32x16& t0,
                               const f32x16& t1, float mul, char* stage /* 4 KB, this wave's */, int lane) {
  const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const f32x16& v = t == 0 ? t0 : t1;
    unsigned P[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      P[g][0] = pack2(v[4 * g] * mul, v[4 * g + 1] * mul);
      P[g][1] = pack2(v[4 * g + 2] * mul, v[4 * g + 3] * mul);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        auto r = __builtin_amdgcn_permlane32_swap(P[k][w], P[k + 2][w], false, false);   // P[k].upper <-> P[k+2].lower
        P[k][w] = r[0]; P[k + 2][w] = r[1];
      }
#pragma unroll
    for (int k = 0; k < 2; ++k) {                                 // columns t * 32 + 16 hi + 8 k .. + 7 = chunk 4 t + 2 hi + k of row l31
      const u32x4 o = {P[k][0], P[k][1], P[k + 2][0], P[k + 2][1]};
      *reinterpret_cast<u32x4*>(stage + l31 * 128 + (((4 * t + 2 * hi + k) ^ (l31 & 7)) << 4)) = o;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {                                   // instruction j: rows 8 j .. 8 j + 7, lane = (row, chunk): 8 lanes per 128-B line
    const int row = 8 * j + (lane >> 3), c = lane & 7;
    const u32x4 o = *reinterpret_cast<const u32x4*>(stage + row * 128 + ((c ^ (row & 7)) << 4));
    if constexpr (ATST_ATTN_NT & 2) __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(g00 + (size_t)row * ld + c * 8));
    else *reinterpret_cast<u32x4*>(g00 + (size_t)row * ld + c * 8) = o;
  }
}

// e4m3 copy of a gradient tile (fp8 qkv dgrad / weight gradient, round 5): the SAME values the bf16 store would write -- rounded to bf16 first --
// times the delayed scale, clamped to +-448.  One dword = the lane's 4-element group.
DEVFN unsigned pack4_e4m3(float a, float b, float c, float d, float s8) {
  auto q = [&](float x) { return __builtin_amdgcn_fmed3f(bf2f(f2bf(x)) * s8, -448.f, 448.f); };
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(q(a), q(b), 0, false);
  return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(q(c), q(d), w, true);
}
DEVFN float amax16(const f32x16& v, float m) {
#pragma unroll
  for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(v[r]));
  return m;
}
// 64-wide e4m3 row per lane pair: after two half-wave swaps per tile a lane holds 16 contiguous columns (t * 32 + 16 hi ..): 2 dwordx4 stores per row
DEVFN void store_row64_e4m3(uint8_t* row, const f32x16& t0, const f32x16& t1, float mul, float s8, int hi) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const f32x16& v = t == 0 ? t0 : t1;
    unsigned X[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) X[g] = pack4_e4m3(v[4 * g] * mul, v[4 * g + 1] * mul, v[4 * g + 2] * mul, v[4 * g + 3] * mul, s8);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      auto r = __builtin_amdgcn_permlane32_swap(X[k], X[k + 2], false, false);   // lower lanes: {D_k, E_k}, upper: {D_k+2, E_k+2}
      X[k] = r[0]; X[k + 2] = r[1];
    }
    const u32x4 o = {X[0], X[2], X[1], X[3]};
    *reinterpret_cast<u32x4*>(row + t * 32 + 16 * hi) = o;
  }
}
// the same for kernels whose tile may hold rows that are not theirs to write (packed sequences, NP < 256): every lane takes part in the swaps, only live rows store
DEVFN void store_row64_e4m3_if(bool live, uint8_t* row, const f32x16& t0, const f32x16& t1, float mul, float s8, int hi) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const f32x16& v = t == 0 ? t0 : t1;
    unsigned X[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) X[g] = pack4_e4m3(v[4 * g] * mul, v[4 * g + 1] * mul, v[4 * g + 2] * mul, v[4 * g + 3] * mul, s8);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      auto r = __builtin_amdgcn_permlane32_swap(X[k], X[k + 2], false, false);
      X[k] = r[0]; X[k + 2] = r[1];
    }
    const u32x4 o = {X[0], X[2], X[1], X[3]};
    if (live) *reinterpret_cast<u32x4*>(row + t * 32 + 16 * hi) = o;
  }
}
// the e4m3 twin of store_tile64_staged: 32 rows x 64 B through 2 KB of the wave's LDS region (16-B chunks XOR-permuted by the row pair), then
// 2 store instructions of 16 complete 64-B rows each
DEVFN void store_tile64_staged_e4m3(uint8_t* g00, size_t ld /* bytes between rows */, const f32x16& t0, const f32x16& t1, float mul, float s8,
                                    char* stage, int lane) {
  const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const f32x16& v = t == 0 ? t0 : t1;
    unsigned X[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) X[g] = pack4_e4m3(v[4 * g] * mul, v[4 * g + 1] * mul, v[4 * g + 2] * mul, v[4 * g + 3] * mul, s8);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      auto r = __builtin_amdgcn_permlane32_swap(X[k], X[k + 2], false, false);
      X[k] = r[0]; X[k + 2] = r[1];
    }
    const u32x4 o = {X[0], X[2], X[1], X[3]};                      // columns t * 32 + 16 hi .. + 15 = chunk 2 t + hi of row l31
    *reinterpret_cast<u32x4*>(stage + l31 * 64 + (((2 * t + hi) ^ ((l31 >> 1) & 3)) << 4)) = o;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = 16 * j + (lane >> 2), c = lane & 3;
    const u32x4 o = *reinterpret_cast<const u32x4*>(stage + row * 64 + ((c ^ ((row >> 1) & 3)) << 4));
    *reinterpret_cast<u32x4*>(g00 + (size_t)row * ld + c * 16) = o;
  }
}

DEVFN void zero16(f32x16& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// ---------------------------------------------------------------------------------------------------------------------
// MODE (round 6, NP = 32: the 1 s local views of the e4m3 step): as in attn_fwd256v2_kernel -- 0 = bf16 output ; 1 = bf16 + the e4m3 copy p.o8 (amax -> p.o8_amax, clipped
// elements -> p.o8_sat) ; 2 = e4m3 only
template <int NP, int MODE = 0>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs p) {
  using GE = Geo<NP>;
  float s8 = 1.0f, rmax = 0.f;
  if constexpr (MODE != 0) s8 = p.o8_scale ? *p.o8_scale : p.o8_scale_k;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, hi = lane >> 5, l31 = lane & 31;
  const int C = p.H * HD;
  const size_t ld = 3 * (size_t)C;
  const int pair_in_blk = wid / GE::WPP, wave_in_pair = wid % GE::WPP;
  const int pair = blockIdx.x * GE::G + pair_in_blk;
  const int npairs = p.S * p.H;
  const bool active = pair < npairs;
  const int s = active ? pair / p.H : 0, h = active ? pair % p.H : 0;
  bf16* sK = reinterpret_cast<bf16*>(smem_raw) + pair_in_blk * 2 * GE::LDS_MAT;
  bf16* sV = sK + GE::LDS_MAT;
  const int RS = p.stride > 0 ? p.stride : NP;                    // rows between sequences (packed when < NP)
  const bf16* base = p.qkv + (size_t)s * RS * ld + h * HD;
  if (active) {
    const int t = wave_in_pair * 64 + lane, nthr = GE::WPP * 64;
    stage_rows<NP>(sK, base + C, ld, t, nthr, RS);
    stage_rows<NP>(sV, base + 2 * C, ld, t, nthr, RS);
  }
  __syncthreads();
  if (!active) return;
  const int valid = p.valid[s];
  const int ntile = (valid + 31) / 32 < GE::CHUNKS ? (valid + 31) / 32 : GE::CHUNKS;
  const float scale = 0.125f;

  for (int cw = 0; cw < GE::CPW; ++cw) {
    const int q0 = (wave_in_pair * GE::CPW + cw) * 32;
    const bool qlive = q0 + l31 < RS;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = ld_frag_if(qlive, base + (size_t)(q0 + l31) * ld + ks * 16 + hi * 8);
    float m_run = -1e30f, l_run = 0.f;
    f32x16 o0, o1; zero16(o0); zero16(o1);
    for (int j = 0; j < ntile; ++j) {
      f32x16 sc; zero16(sc);
      const bf16* kr = sK + (j * 32 + l31) * A_LD + hi * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) sc = mfma32(ld_frag(kr + ks * 16), qf[ks], sc);
      float pv[16];
      float mx = -1e30f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kv = j * 32 + crow32(r, hi);
        pv[r] = sc[r] * scale + (kv >= valid ? MASK_NEG : 0.f);
        mx = fmaxf(mx, pv[r]);
      }
      mx = pair32_max(mx);
      const float m_new = fmaxf(m_run, mx);
      const float alpha = fast_exp2((m_run - m_new) * LOG2E);
      float rs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { pv[r] = fast_exp2((pv[r] - m_new) * LOG2E); rs += pv[r]; }
      rs = pair32_sum(rs);
      l_run = l_run * alpha + rs;
      m_run = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x8 pf = pack8(pv + 8 * t);
        o0 = mfma32(ld_frag_tr(sV, A_LD, j * 32 + 16 * t, 0, lane), pf, o0);
        o1 = mfma32(ld_frag_tr(sV, A_LD, j * 32 + 16 * t, 32, lane), pf, o1);
      }
    }
    const float inv = 1.0f / l_run;
    if constexpr (MODE != 2) {
    bf16* orow = p.o + ((size_t)s * RS + q0 + l31) * C + h * HD;
    if (qlive) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 a, b;
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = f2bf(o0[4 * g + e] * inv); b[e] = f2bf(o1[4 * g + e] * inv); }
        *reinterpret_cast<bf16x4*>(orow + 8 * g + 4 * hi) = a;
        *reinterpret_cast<bf16x4*>(orow + 32 + 8 * g + 4 * hi) = b;
      }
    }
    }
    if constexpr (MODE != 0) {                                     // what atst_quant_fp8_dyn did over the stored bf16 values
      store_row64_e4m3_if(qlive, p.o8 + ((size_t)s * RS + q0 + l31) * C + h * HD, o0, o1, inv, s8, hi);
      const float cm = qlive ? bf2f(f2bf(amax16(o0, amax16(o1, 0.f)) * inv)) : 0.f;
      rmax = fmaxf(rmax, cm);
      if (cm * s8 > 448.f && p.o8_sat) {
        unsigned n = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) n += (fabsf(bf2f(f2bf(o0[r] * inv)) * s8) > 448.f ? 1u : 0u) + (fabsf(bf2f(f2bf(o1[r] * inv)) * s8) > 448.f ? 1u : 0u);
        if (n) atomicAdd(p.o8_sat, n);
      }
    }
    if (hi == 0) p.lse[((size_t)s * p.H + h) * NP + q0 + l31] = m_run + __logf(l_run);   // lse is [S, H, NP] whatever the row stride
  }
  if constexpr (MODE != 0) {
    if (p.o8_amax) amax_post(p.o8_amax, wave_max(rmax), lane, blockIdx.x * 4 + wid);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Forward, NP = 256, one block (8 waves) per SEQUENCE looping over its heads: K/V of head h+1 (and the next Q fragments)
// are fetched into registers while head h is computed and are written to the other LDS buffer afterwards, so the HBM
// latency of the 64 KB K/V panel is hidden behind the MFMAs of the previous head (the per-(sequence, head) kernel above
// exposes it once per block).  Wave w owns the 32-query chunk w.
constexpr int F256_BUF = 2 * 256 * A_LD * 2;            // bytes of one K+V buffer (73,728)

__global__ __launch_bounds__(512, 2) void attn_fwd256_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int NP = 256;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, hi = lane >> 5, l31 = lane & 31;
  const int H = p.H, C = H * HD;
  const size_t ld = 3 * (size_t)C;
  const int s = blockIdx.x;
  const bf16* base = p.qkv + (size_t)s * NP * ld;
  const int valid = p.valid[s];
  const int ntile = (valid + 31) / 32 < 8 ? (valid + 31) / 32 : 8;
  const float scale = 0.125f;
  const int q0 = wid * 32;

  bf16x8 stg[8], qn[4];
  auto gload = [&](int h) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = tid + 512 * i, mat = c >> 11, r = (c >> 3) & 255, k = (c & 7) * 8;
      stg[i] = ld_frag(base + (size_t)r * ld + (1 + mat) * C + h * HD + k);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qn[ks] = ld_frag(base + (size_t)(q0 + l31) * ld + h * HD + ks * 16 + hi * 8);
  };
  auto swrite = [&](int buf) {
    bf16* dst = reinterpret_cast<bf16*>(smem_raw + buf * F256_BUF);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = tid + 512 * i, mat = c >> 11, r = (c >> 3) & 255, k = (c & 7) * 8;
      *reinterpret_cast<bf16x8*>(dst + mat * (NP * A_LD) + r * A_LD + k) = stg[i];
    }
  };
  gload(0);
  swrite(0);
  __syncthreads();
  for (int h = 0; h < H; ++h) {
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = qn[ks];
    if (h + 1 < H) gload(h + 1);                                   // in flight during this head's MFMAs
    const bf16* sK = reinterpret_cast<const bf16*>(smem_raw + (h & 1) * F256_BUF);
    const bf16* sV = sK + NP * A_LD;
    float m_run = -1e30f, l_run = 0.f;
    f32x16 o0, o1; zero16(o0); zero16(o1);
    for (int j = 0; j < ntile; ++j) {
      f32x16 sc; zero16(sc);
      const bf16* kr = sK + (j * 32 + l31) * A_LD + hi * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) sc = mfma32(ld_frag(kr + ks * 16), qf[ks], sc);
      float pv[16];
      float mx = -1e30f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kv = j * 32 + crow32(r, hi);
        pv[r] = sc[r] * scale + (kv >= valid ? MASK_NEG : 0.f);
        mx = fmaxf(mx, pv[r]);
      }
      mx = pair32_max(mx);
      const float m_new = fmaxf(m_run, mx);
      const float alpha = fast_exp2((m_run - m_new) * LOG2E);
      float rs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { pv[r] = fast_exp2((pv[r] - m_new) * LOG2E); rs += pv[r]; }
      rs = pair32_sum(rs);
      l_run = l_run * alpha + rs;
      m_run = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x8 pf = pack8(pv + 8 * t);
        o0 = mfma32(ld_frag_tr(sV, A_LD, j * 32 + 16 * t, 0, lane), pf, o0);
        o1 = mfma32(ld_frag_tr(sV, A_LD, j * 32 + 16 * t, 32, lane), pf, o1);
      }
    }
    const float inv = 1.0f / l_run;
    bf16* orow = p.o + ((size_t)s * NP + q0 + l31) * C + h * HD;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 a, b;
#pragma unroll
      for (int e = 0; e < 4; ++e) { a[e] = f2bf(o0[4 * g + e] * inv); b[e] = f2bf(o1[4 * g + e] * inv); }
      *reinterpret_cast<bf16x4*>(orow + 8 * g + 4 * hi) = a;
      *reinterpret_cast<bf16x4*>(orow + 32 + 8 * g + 4 * hi) = b;
    }
    if (hi == 0) p.lse[((size_t)s * H + h) * NP + q0 + l31] = m_run + __logf(l_run);
    if (h + 1 < H) swrite((h + 1) & 1);                            // buffer (h+1)&1 was last read for head h-1
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Forward, NP = 256, second version: same block-per-sequence head loop, but
//  * K and V of head h+1 go HBM -> LDS by global_load_lds while head h is computed (no staging registers, no ds_write
//    pass): [256][64] bf16 images with 128-B rows, double buffered (2 x 64 KB).  16-B chunk c of row r sits at chunk
//    c ^ ((r >> 1) & 7) for K (ds_read_b128 fragments: the 16 rows of a service group hit 16 distinct 16-B bank slots)
//    and at c ^ (4 ((r >> 1) & 1)) for V (ds_read_b64_tr_b16: 4 rows x 64 B per half-wave on 4 distinct bank windows);
//  * the whole 32 x 256 score strip of a wave stays in registers (8 accumulator tiles), so the softmax is the plain
//    two-pass one: no running-max rescale of the output accumulators, one exp2 and one FMA per score, and the key mask
//    is applied only in the tile that straddles valid[s].  The online version spent ~3.3x the MFMA time in VALU work.
constexpr int F2_MAT = 256 * 128;                       // bytes of one [256][64] bf16 image
constexpr int F2_BUF = 2 * F2_MAT;                      // K + V of one head

DEVFN bf16x8 ld_frag_tr_v(const char* X, int r0, int c0, int lane) {
  const int a = lane & 15, g = lane >> 4;
  const int row = r0 + 4 * (g >> 1) + (a >> 2);
  const int col = c0 + (g & 1) * 16 + 4 * (a & 3);
  const int pch = (col >> 3) ^ (((row >> 1) & 1) << 2);
  const bf16* ptr = reinterpret_cast<const bf16*>(X + row * 128 + pch * 16) + (col & 7);
  s16x4 lo = lds_tr4(ptr);
  s16x4 hi = lds_tr4(ptr + 8 * 64);                                // +8 rows: same key
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = lo; u.s.b = hi;
  return u.v;
}

// Asynchronous transfers issued as inline assembly, i.e. outside the compiler's vmcnt bookkeeping (it then inserts no wait of its own: with
// the builtin / plain loads it put `s_waitcnt vmcnt(0)` in front of every later use while an LDS-DMA was in flight, which also waited
// for the output stores just issued).  The kernels wait by hand with exact counts; see attn_bwd256_kernel for the rules.
DEVFN void glds16_asm(const void* gsrc, unsigned lds_dst /* wave-uniform LDS byte address */) {
  unsigned keep;                                     // M0 is compiler-reserved and not preserved around a statement: saved, set, restored inside it
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
DEVFN void gload16_asm(bf16x8& dst, const void* src) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src) : "memory"); }
DEVFN void gload4_asm(float& dst, const void* src) { asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(src) : "memory"); }
// MODE (fp8 forward, round 5): 0 = bf16 output ; 1 = bf16 + the e4m3 copy p.o8 = e4m3(bf16(o) * scale) the proj GEMM reads (training: the backward
// needs the bf16 one) ; 2 = e4m3 only (inference / teacher passes).  With p.o8: max |bf16(o)| goes to the amax site p.o8_amax and the clipped
// elements to the counter p.o8_sat, exactly what the separate quantisation pass (atst_quant_fp8_dyn) did over the stored bf16 values.  (MODE 1 keeps
// 44 B of loop invariants -- the next head's LDS-DMA source offsets -- in scratch; they are reloaded right behind the head's own vmcnt(0) + barrier.)
template <int MODE>
__global__ __launch_bounds__(512, 2) void attn_fwd256v2_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  typedef const void __attribute__((address_space(1))) * gptr_t;
  float s8 = 1.0f, rmax = 0.f;
  if constexpr (MODE != 0) s8 = p.o8_scale ? *p.o8_scale : p.o8_scale_k;
  typedef void __attribute__((address_space(3))) * lptr_t;
  constexpr int NP = 256;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, hi = lane >> 5, l31 = lane & 31;
  const int H = p.H, C = H * HD;
  const size_t ld = 3 * (size_t)C;
  const int s = blockIdx.x;
  const bf16* base = p.qkv + (size_t)s * NP * ld;
  const int valid = p.valid[s];
  const int ntile = (valid + 31) / 32 < 8 ? (valid + 31) / 32 : 8;
  const int q0 = wid * 32;
  const float c1 = 0.125f * LOG2E;                                 // softmax scale folded into the exp2 argument

  // LDS-DMA: instruction i of a wave moves 8 rows x 128 B; per head each wave moves rows [32 wid, 32 wid + 32) of K and V
  int offK[4], offV[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wid * 32 + i * 8 + (lane >> 3), pc = lane & 7;
    offK[i] = row * (int)ld + C + (pc ^ ((row >> 1) & 7)) * 8;
    offV[i] = row * (int)ld + 2 * C + (pc ^ (((row >> 1) & 1) << 2)) * 8;
  }
  auto issue = [&](int h) {
    char* buf = smem_raw + (h & 1) * F2_BUF;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((gptr_t)(base + offK[i] + h * HD), (lptr_t)(buf + (wid * 4 + i) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(base + offV[i] + h * HD), (lptr_t)(buf + F2_MAT + (wid * 4 + i) * 1024), 16, 0, 0);
    }
  };
  bf16x8 qn[4];
  auto qload = [&](int h) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qn[ks] = ld_frag(base + (size_t)(q0 + l31) * ld + h * HD + ks * 16 + hi * 8);
  };
  // lane-only parts of the swizzled fragment addresses (tile offsets are compile-time constants added by the reads)
  int kofs[4], vofs[2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) kofs[ks] = l31 * 128 + (((ks * 2 + hi) ^ ((l31 >> 1) & 7)) << 4);
  {
    const int a = lane & 15, g = lane >> 4;
    const int lrow = 4 * (g >> 1) + (a >> 2), lcol = (g & 1) * 16 + 4 * (a & 3), swz = ((lrow >> 1) & 1) << 2;
#pragma unroll
    for (int c = 0; c < 2; ++c) vofs[c] = lrow * 128 + (((c * 4 + (lcol >> 3)) ^ swz) << 4) + (lcol & 7) * 2;
  }
  auto vfrag = [&](const char* sV, int r0, int c) {                // V^T fragment: keys r0..r0+15, head dims 32c..32c+31
    const bf16* ptr = reinterpret_cast<const bf16*>(sV + vofs[c] + r0 * 128);
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lds_tr4(ptr); u.s.b = lds_tr4(ptr + 8 * 64);
    return u.v;
  };
  // bit r of mbits: accumulator register r of the LAST key tile is a padded key.  One per-lane word instead of 16
  // compare masks per tile: those are loop-invariant, get hoisted out of the head loop and spill the scalar registers.
  unsigned mbits = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) mbits |= ((ntile - 1) * 32 + crow32(r, hi) >= valid ? 1u : 0u) << r;
  asm volatile("" : "+v"(mbits));
  // (Round 4 also ran this kernel with its transfers as inline assembly and an exact vmcnt(4) at the top of every head, like the backward:
  // +1.2 ... 1.6 us per launch inside the step, profiles/r04_ab_*: the compiler-tracked form stays.)
  issue(0);
  qload(0);
  for (int h = 0; h < H; ++h) {
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = qn[ks];
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");   // head h landed everywhere; buffer (h+1)&1 no longer read
    if (h + 1 < H) { issue(h + 1); qload(h + 1); }
    const char* sK = smem_raw + (h & 1) * F2_BUF;
    const char* sV = sK + F2_MAT;
    // pass 1: the wave's 32 x 256 score strip (raw q.k, scale folded into the exponent) and its row maximum
    f32x16 sc[8];
    float mx = -3.0e38f;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < ntile) {
        sc[j] = mfma32(*reinterpret_cast<const bf16x8*>(sK + kofs[0] + j * 4096), qf[0], zero);   // C = inline 0: no accumulator clears
#pragma unroll
        for (int ks = 1; ks < 4; ++ks)
          sc[j] = mfma32(*reinterpret_cast<const bf16x8*>(sK + kofs[ks] + j * 4096), qf[ks], sc[j]);
        if (j == ntile - 1) {                                      // the only tile that can hold padded keys
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if ((mbits >> r) & 1u) sc[j][r] = -3.0e38f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[j][r]);
      }
    }
    mx = pair32_max(mx);
    const float c0 = -mx * c1;
    // pass 2: p = exp2(c1 s - c1 max) ; O^T += V^T P^T
    float rs = 0.f;
    f32x16 o0, o1; zero16(o0); zero16(o1);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < ntile) {
        float pv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { pv[r] = __builtin_amdgcn_exp2f(fmaf(sc[j][r], c1, c0)); rs += pv[r]; }   // raw v_exp_f32: argument <= 0, underflow to 0 is the wanted result (exp2f adds a 4-instruction denormal-range fix-up)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const bf16x8 pf = pack8(pv + 8 * t);
          o0 = mfma32(vfrag(sV, j * 32 + 16 * t, 0), pf, o0);
          o1 = mfma32(vfrag(sV, j * 32 + 16 * t, 1), pf, o1);
        }
      }
    }
    rs = pair32_sum(rs);
    const float inv = 1.0f / rs;
    bf16* orow = p.o + ((size_t)s * NP + q0 + l31) * C + h * HD;
#ifdef ATST_ABLATE_ATTN_STORE
    if (rs == 12345.678f)                                           // experiment builds: no output stores
#endif
    if constexpr (MODE != 2) store_row64(orow, o0, o1, inv, hi);   // (full-line stores through an LDS transposition buffer, as in the backward: 60 B of scratch at 256 registers -- the score strip is the register file)
    if constexpr (MODE != 0) {
      store_row64_e4m3(p.o8 + ((size_t)s * NP + q0 + l31) * C + h * HD, o0, o1, inv, s8, hi);
      const float cm = bf2f(f2bf(amax16(o0, amax16(o1, 0.f)) * inv));      // rounding is monotonic: the largest |bf16(x)| of this lane's 32 values
      rmax = fmaxf(rmax, cm);
      if (cm * s8 > 448.f && p.o8_sat) {                          // rare: something of this row piece was clipped -- count exactly, post at once (no counter kept live)
        unsigned n = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) n += (fabsf(bf2f(f2bf(o0[r] * inv)) * s8) > 448.f ? 1u : 0u) + (fabsf(bf2f(f2bf(o1[r] * inv)) * s8) > 448.f ? 1u : 0u);
        if (n) atomicAdd(p.o8_sat, n);
      }
    }
    if (p.lse && hi == 0) p.lse[((size_t)s * H + h) * NP + q0 + l31] = mx * 0.125f + __logf(rs);
  }
  if constexpr (MODE != 0) {
    if (p.o8_amax) amax_post(p.o8_amax, wave_max(rmax), lane, blockIdx.x * 8 + wid);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward part 1: dK, dV.  LDS: Q [NP][72], dO [NP][72], lse[NP], D[NP] per pair.
template <int NP>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnArgs p) {
  using GE = Geo<NP>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, hi = lane >> 5, l31 = lane & 31;
  const int C = p.H * HD;
  const size_t ld = 3 * (size_t)C;
  const int pair_in_blk = wid / GE::WPP, wave_in_pair = wid % GE::WPP;
  const int pair = blockIdx.x * GE::G + pair_in_blk;
  const bool active = pair < p.S * p.H;
  const int s = active ? pair / p.H : 0, h = active ? pair % p.H : 0;
  constexpr int PAIR_BYTES = 2 * GE::LDS_MAT * 2 + 2 * NP * 4;
  char* pb = smem_raw + pair_in_blk * PAIR_BYTES;
  bf16* sQ = reinterpret_cast<bf16*>(pb);
  bf16* sDO = sQ + GE::LDS_MAT;
  float* sLse = reinterpret_cast<float*>(sDO + GE::LDS_MAT);
  float* sD = sLse + NP;
  const int RS = p.stride > 0 ? p.stride : NP;                    // rows between sequences (packed when < NP)
  const bf16* base = p.qkv + (size_t)s * RS * ld + h * HD;
  const bf16* dobase = p.d_o + (size_t)s * RS * C + h * HD;
  const bf16* obase = p.o + (size_t)s * RS * C + h * HD;
  if (active) {
    const int t = wave_in_pair * 64 + lane, nthr = GE::WPP * 64;
    stage_rows<NP>(sQ, base, ld, t, nthr, RS);
    stage_rows<NP>(sDO, dobase, (size_t)C, t, nthr, RS);
    for (int q = t; q < NP; q += nthr) {
      float d = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const bf16x8 a = ld_frag_if(q < RS, dobase + (size_t)q * C + k * 8), b = ld_frag_if(q < RS, obase + (size_t)q * C + k * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) d += bf2f(a[e]) * bf2f(b[e]);
      }
      sD[q] = d;
      sLse[q] = p.lse[((size_t)s * p.H + h) * NP + q];
    }
  }
  __syncthreads();
  if (!active) return;
  const int valid = p.valid[s];
  const float scale = 0.125f;
  // queries beyond the valid range still exist in the reference (their dO is exactly zero), so all query tiles run
  for (int cw = 0; cw < GE::CPW; ++cw) {
    const int k0 = (wave_in_pair * GE::CPW + cw) * 32;
    bf16* dkrow = p.dqkv + ((size_t)s * RS + k0 + l31) * ld + C + h * HD;
    bf16* dvrow = dkrow + C;
    const bool klive = k0 + l31 < RS;
    f32x16 dk0, dk1, dv0, dv1; zero16(dk0); zero16(dk1); zero16(dv0); zero16(dv1);
    if (k0 < valid) {
      bf16x8 kf[4], vf[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        kf[ks] = ld_frag_if(klive, base + C + (size_t)(k0 + l31) * ld + ks * 16 + hi * 8);
        vf[ks] = ld_frag_if(klive, base + 2 * C + (size_t)(k0 + l31) * ld + ks * 16 + hi * 8);
      }
      const float kbias = (k0 + l31 >= valid) ? MASK_NEG : 0.f;
      for (int i = 0; i < GE::CHUNKS; ++i) {
        f32x16 sc, dp; zero16(sc); zero16(dp);
        const bf16* qr = sQ + (i * 32 + l31) * A_LD + hi * 8;
        const bf16* dr = sDO + (i * 32 + l31) * A_LD + hi * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          sc = mfma32(ld_frag(qr + ks * 16), kf[ks], sc);       // S[q, kv]
          dp = mfma32(ld_frag(dr + ks * 16), vf[ks], dp);       // dP[q, kv]
        }
        float pv[16], ds[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(sLse + i * 32 + 8 * g + 4 * hi);
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(sD + i * 32 + 8 * g + 4 * hi);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            const float pr = fast_exp2((sc[r] * scale + kbias - l4[e]) * LOG2E);
            pv[r] = pr;
            ds[r] = pr * (dp[r] - d4[e]) * scale;
          }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const bf16x8 pf = pack8(pv + 8 * t), dsf = pack8(ds + 8 * t);
          dv0 = mfma32(ld_frag_tr(sDO, A_LD, i * 32 + 16 * t, 0, lane), pf, dv0);
          dv1 = mfma32(ld_frag_tr(sDO, A_LD, i * 32 + 16 * t, 32, lane), pf, dv1);
          dk0 = mfma32(ld_frag_tr(sQ, A_LD, i * 32 + 16 * t, 0, lane), dsf, dk0);
          dk1 = mfma32(ld_frag_tr(sQ, A_LD, i * 32 + 16 * t, 32, lane), dsf, dk1);
        }
      }
    }
    if (klive) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 a, b, c, d;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a[e] = f2bf(dk0[4 * g + e]); b[e] = f2bf(dk1[4 * g + e]);
          c[e] = f2bf(dv0[4 * g + e]); d[e] = f2bf(dv1[4 * g + e]);
        }
        *reinterpret_cast<bf16x4*>(dkrow + 8 * g + 4 * hi) = a;
        *reinterpret_cast<bf16x4*>(dkrow + 32 + 8 * g + 4 * hi) = b;
        *reinterpret_cast<bf16x4*>(dvrow + 8 * g + 4 * hi) = c;
        *reinterpret_cast<bf16x4*>(dvrow + 32 + 8 * g + 4 * hi) = d;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward part 2: dQ.  LDS: K [NP][72], V [NP][72] per pair.
template <int NP>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnArgs p) {
  using GE = Geo<NP>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, hi = lane >> 5, l31 = lane & 31;
  const int C = p.H * HD;
  const size_t ld = 3 * (size_t)C;
  const int pair_in_blk = wid / GE::WPP, wave_in_pair = wid % GE::WPP;
  const int pair = blockIdx.x * GE::G + pair_in_blk;
  const bool active = pair < p.S * p.H;
  const int s = active ? pair / p.H : 0, h = active ? pair % p.H : 0;
  bf16* sK = reinterpret_cast<bf16*>(smem_raw) + pair_in_blk * 2 * GE::LDS_MAT;
  bf16* sV = sK + GE::LDS_MAT;
  const int RS = p.stride > 0 ? p.stride : NP;                    // rows between sequences (packed when < NP)
  const bf16* base = p.qkv + (size_t)s * RS * ld + h * HD;
  if (active) {
    const int t = wave_in_pair * 64 + lane, nthr = GE::WPP * 64;
    stage_rows<NP>(sK, base + C, ld, t, nthr, RS);
    stage_rows<NP>(sV, base + 2 * C, ld, t, nthr, RS);
  }
  __syncthreads();
  if (!active) return;
  const int valid = p.valid[s];
  const int ntile = (valid + 31) / 32 < GE::CHUNKS ? (valid + 31) / 32 : GE::CHUNKS;
  const float scale = 0.125f;
  for (int cw = 0; cw < GE::CPW; ++cw) {
    const int q0 = (wave_in_pair * GE::CPW + cw) * 32;
    const size_t qrow = (size_t)s * RS + q0 + l31;
    const bool qlive = q0 + l31 < RS;
    bf16x8 qf[4], dof[4];
    float dpart = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = ld_frag_if(qlive, base + (size_t)(q0 + l31) * ld + ks * 16 + hi * 8);
      dof[ks] = ld_frag_if(qlive, p.d_o + qrow * C + h * HD + ks * 16 + hi * 8);
      const bf16x8 of = ld_frag_if(qlive, p.o + qrow * C + h * HD + ks * 16 + hi * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) dpart += bf2f(dof[ks][e]) * bf2f(of[e]);
    }
    const float D = pair32_sum(dpart);
    const float lse = p.lse[((size_t)s * p.H + h) * NP + q0 + l31];
    f32x16 dq0, dq1; zero16(dq0); zero16(dq1);
    for (int j = 0; j < ntile; ++j) {
      f32x16 sc, dp; zero16(sc); zero16(dp);
      const bf16* kr = sK + (j * 32 + l31) * A_LD + hi * 8;
      const bf16* vr = sV + (j * 32 + l31) * A_LD + hi * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        sc = mfma32(ld_frag(kr + ks * 16), qf[ks], sc);        // S^T[kv, q]
        dp = mfma32(ld_frag(vr + ks * 16), dof[ks], dp);       // dP^T[kv, q]
      }
      float ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kv = j * 32 + crow32(r, hi);
        const float pr = fast_exp2((sc[r] * scale + (kv >= valid ? MASK_NEG : 0.f) - lse) * LOG2E);
        ds[r] = pr * (dp[r] - D) * scale;
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x8 dsf = pack8(ds + 8 * t);
        dq0 = mfma32(ld_frag_tr(sK, A_LD, j * 32 + 16 * t, 0, lane), dsf, dq0);
        dq1 = mfma32(ld_frag_tr(sK, A_LD, j * 32 + 16 * t, 32, lane), dsf, dq1);
      }
    }
    bf16* dqrow = p.dqkv + qrow * ld + h * HD;
    if (qlive) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 a, b;
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = f2bf(dq0[4 * g + e]); b[e] = f2bf(dq1[4 * g + e]); }
        *reinterpret_cast<bf16x4*>(dqrow + 8 * g + 4 * hi) = a;
        *reinterpret_cast<bf16x4*>(dqrow + 32 + 8 * g + 4 * hi) = b;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward, NP = 32 (the 1 s local views: 26 tokens, packed), ONE wave per (sequence, head) pair doing both halves.  The two-kernel path reads
// Q, K, V twice, dO three times and O twice from memory (each kernel stages its own pair of matrices and fetches the other pair as fragments);
// here Q, K, V, dO are staged once (18 KB per pair, wave-private), O is read once for D = rowsum(dO * O), and the 32 x 32 score tile is simply
// computed in both orientations (S for dK / dV: lane = key ; S^T for dQ: lane = query) -- 16 extra MFMAs cost nothing next to a second pass
// over memory and a second launch.
// MODE (round 6): 0 = bf16 dqkv ; 2 = ONLY the e4m3 copy p.dqkv8 + max |bf16(dqkv)| over the live rows posted to p.q8_amax, as attn_bwd256_kernel<2>: the qkv dgrad
// and weight gradient of the local views then read e4m3 like those of the 10 s views.
template <int MODE = 0>
__global__ __launch_bounds__(256) void attn_bwd32_kernel(AttnArgs p) {
  constexpr int NP = 32, MAT = NP * A_LD;
  float s8 = 1.0f, gmax = 0.f;
  if constexpr (MODE == 2) s8 = *p.q8_scale;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, hi = lane >> 5, l31 = lane & 31;
  const int C = p.H * HD;
  const size_t ld = 3 * (size_t)C;
  const int pair = blockIdx.x * 4 + wid;
  if (pair >= p.S * p.H) return;                                  // wave-private LDS, no block barrier below (MODE 2: nothing to post either)
  const int s = pair / p.H, h = pair % p.H;
  constexpr int PAIR_BYTES = 4 * MAT * 2 + 2 * NP * 4;
  char* pb = smem_raw + wid * PAIR_BYTES;
  bf16* sQ = reinterpret_cast<bf16*>(pb);
  bf16* sDO = sQ + MAT; bf16* sK = sDO + MAT; bf16* sV = sK + MAT;
  float* sLse = reinterpret_cast<float*>(sV + MAT);
  float* sD = sLse + NP;
  const int RS = p.stride > 0 ? p.stride : NP;
  const bf16* base = p.qkv + (size_t)s * RS * ld + h * HD;
  const bf16* dobase = p.d_o + (size_t)s * RS * C + h * HD;
  const bf16* obase = p.o + (size_t)s * RS * C + h * HD;
  stage_rows<NP>(sQ, base, ld, lane, 64, RS);
  stage_rows<NP>(sK, base + C, ld, lane, 64, RS);
  stage_rows<NP>(sV, base + 2 * C, ld, lane, 64, RS);
  stage_rows<NP>(sDO, dobase, (size_t)C, lane, 64, RS);
  {                                                               // D[q]: half a row per lane (lane = q + 32 half), folded by one swap
    const int q = l31;
    float d = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bf16x8 a = ld_frag_if(q < RS, dobase + (size_t)q * C + (hi * 4 + k) * 8), b = ld_frag_if(q < RS, obase + (size_t)q * C + (hi * 4 + k) * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) d += bf2f(a[e]) * bf2f(b[e]);
    }
    d = pair32_sum(d);
    if (hi == 0) { sD[q] = d; sLse[q] = p.lse[((size_t)s * p.H + h) * NP + q]; }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // this wave's staging is complete (LDS operations of a wave execute in order)
  __builtin_amdgcn_wave_barrier();
  const int valid = p.valid[s];
  const float scale = 0.125f;
  const bool live = l31 < RS;
  // ---- dK, dV: lane = key l31
  {
    bf16* dkrow = p.dqkv + ((size_t)s * RS + l31) * ld + C + h * HD;
    bf16* dvrow = dkrow + C;
    f32x16 dk0, dk1, dv0, dv1; zero16(dk0); zero16(dk1); zero16(dv0); zero16(dv1);
    if (valid > 0) {
      const float kbias = (l31 >= valid) ? MASK_NEG : 0.f;
      f32x16 sc, dp; zero16(sc); zero16(dp);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        sc = mfma32(ld_frag(sQ + l31 * A_LD + hi * 8 + ks * 16), ld_frag(sK + l31 * A_LD + hi * 8 + ks * 16), sc);      // S[q, kv]
        dp = mfma32(ld_frag(sDO + l31 * A_LD + hi * 8 + ks * 16), ld_frag(sV + l31 * A_LD + hi * 8 + ks * 16), dp);     // dP[q, kv]
      }
      float pv[16], ds[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(sLse + 8 * g + 4 * hi);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(sD + 8 * g + 4 * hi);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const float pr = fast_exp2((sc[r] * scale + kbias - l4[e]) * LOG2E);
          pv[r] = pr;
          ds[r] = pr * (dp[r] - d4[e]) * scale;
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x8 pf = pack8(pv + 8 * t), dsf = pack8(ds + 8 * t);
        dv0 = mfma32(ld_frag_tr(sDO, A_LD, 16 * t, 0, lane), pf, dv0);
        dv1 = mfma32(ld_frag_tr(sDO, A_LD, 16 * t, 32, lane), pf, dv1);
        dk0 = mfma32(ld_frag_tr(sQ, A_LD, 16 * t, 0, lane), dsf, dk0);
        dk1 = mfma32(ld_frag_tr(sQ, A_LD, 16 * t, 32, lane), dsf, dk1);
      }
    }
    if constexpr (MODE == 2) {
      uint8_t* dk8 = p.dqkv8 + ((size_t)s * RS + l31) * ld + C + h * HD;
      store_row64_e4m3_if(live, dk8, dk0, dk1, 1.0f, s8, hi);
      store_row64_e4m3_if(live, dk8 + C, dv0, dv1, 1.0f, s8, hi);
      if (live) gmax = amax16(dk0, amax16(dk1, amax16(dv0, amax16(dv1, gmax))));
    } else
    if (live) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 a, b, c, d;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a[e] = f2bf(dk0[4 * g + e]); b[e] = f2bf(dk1[4 * g + e]);
          c[e] = f2bf(dv0[4 * g + e]); d[e] = f2bf(dv1[4 * g + e]);
        }
        *reinterpret_cast<bf16x4*>(dkrow + 8 * g + 4 * hi) = a;
        *reinterpret_cast<bf16x4*>(dkrow + 32 + 8 * g + 4 * hi) = b;
        *reinterpret_cast<bf16x4*>(dvrow + 8 * g + 4 * hi) = c;
        *reinterpret_cast<bf16x4*>(dvrow + 32 + 8 * g + 4 * hi) = d;
      }
    }
  }
  // ---- dQ: lane = query l31 (S^T[kv, q], as attn_bwd_dq_kernel)
  {
    const float D = sD[l31], lse = sLse[l31];
    f32x16 dq0, dq1; zero16(dq0); zero16(dq1);
    if (valid > 0) {
      f32x16 sc, dp; zero16(sc); zero16(dp);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        sc = mfma32(ld_frag(sK + l31 * A_LD + hi * 8 + ks * 16), ld_frag(sQ + l31 * A_LD + hi * 8 + ks * 16), sc);      // S^T[kv, q]
        dp = mfma32(ld_frag(sV + l31 * A_LD + hi * 8 + ks * 16), ld_frag(sDO + l31 * A_LD + hi * 8 + ks * 16), dp);     // dP^T[kv, q]
      }
      float ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kv = crow32(r, hi);
        const float pr = fast_exp2((sc[r] * scale + (kv >= valid ? MASK_NEG : 0.f) - lse) * LOG2E);
        ds[r] = pr * (dp[r] - D) * scale;
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x8 dsf = pack8(ds + 8 * t);
        dq0 = mfma32(ld_frag_tr(sK, A_LD, 16 * t, 0, lane), dsf, dq0);
        dq1 = mfma32(ld_frag_tr(sK, A_LD, 16 * t, 32, lane), dsf, dq1);
      }
    }
    if constexpr (MODE == 2) {
      store_row64_e4m3_if(live, p.dqkv8 + ((size_t)s * RS + l31) * ld + h * HD, dq0, dq1, 1.0f, s8, hi);
      if (live) gmax = amax16(dq0, amax16(dq1, gmax));
    } else {
    bf16* dqrow = p.dqkv + ((size_t)s * RS + l31) * ld + h * HD;
    if (live) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 a, b;
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = f2bf(dq0[4 * g + e]); b[e] = f2bf(dq1[4 * g + e]); }
        *reinterpret_cast<bf16x4*>(dqrow + 8 * g + 4 * hi) = a;
        *reinterpret_cast<bf16x4*>(dqrow + 32 + 8 * g + 4 * hi) = b;
      }
    }
    }
  }
  if constexpr (MODE == 2) amax_post(p.q8_amax, bf2f(f2bf(wave_max(gmax))), lane, blockIdx.x * 4 + wid);
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward, NP = 256, one block (8 waves) per SEQUENCE looping over its heads.  Q, K, V, dO of the current head are all
// LDS-resident (4 x 36 KB) so that both halves of the backward -- dK/dV (wave = 32 keys) and dQ (wave = 32 queries) --
// read every operand on-chip after ONE staging pass (the two-kernel version stages Q,dO and K,V separately and re-reads
// the per-wave fragments from HBM).  The next head's panels are prefetched into registers during the dQ phase.  D = rowsum(dO * O) comes from attn_rowdot_kernel.
constexpr int B256_MAT = 256 * HD;                                 // elements of one staged matrix: [256][64], 128-B rows
constexpr int B256_LDS = 4 * B256_MAT * 2 + 2 * 256 * 4;           // 133,120 B
// The four images are read both by rows (ds_read_b128 fragments) and transposed (ds_read_b64_tr_b16).  16-B chunk c of
// row r is stored at chunk c ^ key(r), key = rotate-right of ((r >> 1) & 7): a bijection of (r >> 1) & 7, so the 16 rows
// of a ds_read_b128 service group hit 16 distinct bank slots, and its top bit is bit 1 of r, so the 4 rows x 64 B of a
// transpose-read half-wave hit 4 distinct 64-B windows (the padded 144-B pitch had 2-way conflicts there: 22 % of the
// kernel's LDS cycles).
DEVFN int b256_key(int row) { const int t = (row >> 1) & 7; return ((t & 1) << 2) | (t >> 1); }
DEVFN const bf16* b256_ptr(const bf16* base, int row, int chunk) { return base + row * HD + ((chunk ^ b256_key(row)) << 3); }
DEVFN bf16x8 b256_frag_tr(const bf16* X, int r0, int c0, int lane) {
  const int a = lane & 15, g = lane >> 4;
  const int row = r0 + 4 * (g >> 1) + (a >> 2);
  const int col = c0 + (g & 1) * 16 + 4 * (a & 3);
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = lds_tr4(b256_ptr(X, row, col >> 3) + (col & 7));
  u.s.b = lds_tr4(b256_ptr(X, row + 8, col >> 3) + (col & 7));
  return u.v;
}

__global__ __launch_bounds__(256) void attn_rowdot_kernel(const bf16* __restrict__ d_o, const bf16* __restrict__ o, float* __restrict__ D,
                                                          int S, int H, int NP) {
  // D[s,h,q] = sum_d dO[s,q,h,d] * O[s,q,h,d].  One thread per 16-B chunk (8 elements) of the [rows, C] tensors, flat over rows: every lane of every
  // wave loads (the round-1 form gave a wave one row -- 48 of 64 lanes at C = 384); a head is 8 consecutive chunks = 8 consecutive lanes (C / 8 and the
  // wave size are multiples of 8, so a group never straddles a wave or a row)
  const int cpr = H * (HD / 8);                                    // chunks per row
  const size_t total = (size_t)S * NP * cpr;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ((total + 7) & ~(size_t)7); i += (size_t)gridDim.x * blockDim.x) {
    float d = 0.f;
    if (i < total) {
      const bf16x8 a = ld_frag(d_o + i * 8), b = ld_frag(o + i * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) d += bf2f(a[e]) * bf2f(b[e]);
    }
    d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
    if ((threadIdx.x & 7) == 0 && i < total) {
      const size_t row = i / cpr; const int h = (int)(i % cpr) >> 3;
      D[((size_t)(row / NP) * H + h) * NP + row % NP] = d;
    }
  }
}

// Experiment builds (tools/attn_ablate.sh): ATST_ATTN_ABL bit 0 = no softmax arithmetic (P := S), bit 1 = no dK / dV products (transposed fragment
// reads + 8 MFMAs per query block), bit 2 = no dQ phase, bit 3 = no global stores, bit 4 = no S / dP products in the dK / dV phase.
#ifndef ATST_ATTN_ABL
#define ATST_ATTN_ABL 0
#endif
// ---- round 4: every asynchronous transfer issued and waited for BY HAND ---------------------------------------------------------
// What the ablation of the round-3 kernel showed (tools/attn_ablate.sh, profiles/r04_attn_ablate.txt): removing the global stores alone
// took 67 of 300 us off, the softmax arithmetic 32, each group of products 50-65 -- the parts add up, little overlaps.  One suspect was
// in the ISA: hipcc's waitcnt insertion had put `s_waitcnt vmcnt(0)` (a) between the register prefetch loads of the next head, right
// behind the dK / dV stores and the LDS-DMA issue, (b) in front of the first transposing LDS read of the dQ loop (it treats the
// builtin as a reader of every LDS-DMA in flight) and (c) at the top of every head -- vmcnt retires in order, so each of them also
// waited for the stores just issued to be acknowledged, and for the next head's operands to land before the current head went on.
// This version removes all of them (checked in the ISA: tools/check_attn_bwd_isa.py) and frees the 56 B of scratch the round-3 kernel
// spilled (245 registers instead of 256) -- and measures 307 vs 310 us (profiles/r04_attn_time.txt): the drains were not the cost.  What
// is: the row-per-lane stores themselves (32-B pieces of 128-B lines, ~60 us = the store bandwidth of the chip at that
// granularity) and, in the product loops, the LDS-read latency in front of every MFMA (no register left to read a block ahead).
// Kept because it is the cleaner schedule (no spills, exact waits), not because it is faster.  Here:
//   * every load is inline assembly (LDS-DMA: glds16_asm ; register loads: gload16_asm), invisible to the compiler's scoreboard, so it
//     inserts no wait; the waits are written below as exact `vmcnt(N)` counts, N = the number of STORE instructions issued behind the
//     loads being waited for (vmcnt counts loads and stores in issue order; more stores than counted only makes a wait stricter);
//   * K / V no longer pass through registers into LDS: their LDS images are only needed by the dQ phase, so they are LDS-DMA'd at the
//     top of the head and land under the dK / dV phase, which needs just this wave's own 32 key rows -- those arrive as MFMA
//     fragments straight from global memory (prefetched during the previous head's dQ phase: the 32 registers the raw rows used to
//     occupy), no LDS write pass, no LDS read for them;
//   * per head the queue of a wave is  [K,V DMA 16] dKdV-phase [dK,dV stores 8] wait(8) | [Q,dO DMA 8][lse,D 2][K,V fragments 8]
//     dQ-phase [dQ stores 4] | next head: wait(4).
// everything this wave loaded for the head that starts has landed; N younger operations (stores) may still be in flight.  The
// destinations are "+v" operands: no consumer moves above the wait, and the compiler cannot assume their old contents.
template <int N> DEVFN void bwd_wait_head(bf16x8 (&k)[4], bf16x8 (&v)[4], float& a, float& b) {
  asm volatile("s_waitcnt vmcnt(%10)" : "+v"(k[0]), "+v"(k[1]), "+v"(k[2]), "+v"(k[3]), "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(a), "+v"(b)
               : "n"(N) : "memory");
}
#if ATST_ATTN_ABL & 8
constexpr int BWD_ST_DKV = 0, BWD_ST_DQ = 0, BWD_ST_DQ8 = 0;         // experiment build without stores
#else
constexpr int BWD_ST_DKV = 8, BWD_ST_DQ = 4;         // store instructions of store_row64 x 2 / x 1 (16 B per lane each: cannot be fewer)
constexpr int BWD_ST_DQ8 = 2;                        // ... of store_row64_e4m3 (MODE 2)
#endif

// MODE (fp8 qkv gradient path, round 5): 0 = bf16 dqkv ; 2 = ONLY the e4m3 copy p.dqkv8 = e4m3(bf16(dqkv) * *p.q8_scale) + max |bf16(dqkv)| posted
// to p.q8_amax (delayed scaling): both consumers (qkv dgrad, qkv weight gradient) then read e4m3, and the kernel stores 1 byte per element
// instead of 2 (6 store instructions per head instead of 12).  (A bf16 + amax variant -- the recording step -- spilled 112 B next to the
// asm-loaded registers: the engine records that one step's amax with a pass over the bf16 dqkv instead.)
template <int MODE>
__global__ __launch_bounds__(512, 2) void attn_bwd256_kernel(AttnArgs p, const float* __restrict__ Dg) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int NP = 256;
  constexpr int ST_DQ = MODE == 2 ? BWD_ST_DQ8 : BWD_ST_DQ;
  float s8 = 1.0f, gmax = 0.f;
  if constexpr (MODE == 2) s8 = *p.q8_scale;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), hi = lane >> 5, l31 = lane & 31;
  const int H = p.H, C = H * HD;
  const size_t ld = 3 * (size_t)C;
  const int s = blockIdx.x;
  bf16* sQ = reinterpret_cast<bf16*>(smem_raw);
  bf16* sK = sQ + B256_MAT; bf16* sV = sK + B256_MAT; bf16* sDO = sV + B256_MAT;
  float* sLse = reinterpret_cast<float*>(sDO + B256_MAT);
  float* sD = sLse + NP;
  const bf16* qkv = p.qkv + (size_t)s * NP * ld;
  const bf16* dob = p.d_o + (size_t)s * NP * C;
  const int valid = p.valid[s];
  const int ntile = (valid + 31) / 32 < 8 ? (valid + 31) / 32 : 8;
  const float scale = 0.125f;
  const int k0 = wid * 32;                                          // this wave's key rows (dK / dV phase) = its query rows (dQ phase)

  // LDS-DMA map: instruction i of a wave covers rows 32 wid + 8 i .. + 8 (128-B rows), lane = (row, physical chunk); the same map
  // serves Q, K, V (columns 0, C, 2C of the qkv row) and dO
  int dq_off[4], ddo_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wid * 32 + i * 8 + (lane >> 3), c = (lane & 7) ^ b256_key(row);
    dq_off[i] = row * (int)ld + c * 8;
    ddo_off[i] = row * C + c * 8;
  }
  const unsigned lQ = lds_addr(sQ) + wid * 4096, lK = lds_addr(sK) + wid * 4096, lV = lds_addr(sV) + wid * 4096, lDO = lds_addr(sDO) + wid * 4096;
  auto dma_qdo = [&](int h) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16_asm(qkv + dq_off[i] + h * HD, lQ + i * 1024);
      glds16_asm(dob + ddo_off[i] + h * HD, lDO + i * 1024);
    }
  };
  auto dma_kv = [&](int h) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16_asm(qkv + dq_off[i] + C + h * HD, lK + i * 1024);
      glds16_asm(qkv + dq_off[i] + 2 * C + h * HD, lV + i * 1024);
    }
  };
  // next head's operands that live in registers: this wave's own K / V rows as MFMA fragments (row k0 + l31, logical chunk 2 ks + hi)
  // and one lse / D value per thread (threads >= 256 re-load row tid - 256: every wave issues the same number of loads)
  bf16x8 kfn[4], vfn[4];
  float plse, pd;
  const bf16* kv_src = qkv + (size_t)(k0 + l31) * ld + C + hi * 8;
  auto load_regs = [&](int h) {
    gload4_asm(plse, p.lse + ((size_t)s * H + h) * NP + (tid & 255));
    gload4_asm(pd, Dg + ((size_t)s * H + h) * NP + (tid & 255));
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      gload16_asm(kfn[ks], kv_src + h * HD + ks * 16);
      gload16_asm(vfn[ks], kv_src + C + h * HD + ks * 16);
    }
  };
  // lane-only parts of the swizzled fragment addresses (row blocks are multiples of 16 rows, so the permutation key of a
  // fragment row depends on the lane alone); the block / tile offsets are added as constants by the reads
  int rofs[4], tofs[2][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) rofs[ks] = l31 * HD + (((ks * 2 + hi) ^ b256_key(l31)) << 3);
  {
    const int a = lane & 15, g = lane >> 4;
    const int lrow = 4 * (g >> 1) + (a >> 2), lcol = (g & 1) * 16 + 4 * (a & 3);
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int half = 0; half < 2; ++half)
        tofs[cc][half] = (lrow + 8 * half) * HD + (((cc * 4 + (lcol >> 3)) ^ b256_key(lrow + 8 * half)) << 3) + (lcol & 7);
  }
  auto rfrag = [&](const bf16* X, int row0, int ks) { return ld_frag(X + row0 * HD + rofs[ks]); };          // rows row0 + l31
  auto tfrag = [&](const bf16* X, int r0, int cc) {                                                          // X^T fragment
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lds_tr4(X + r0 * HD + tofs[cc][0]); u.s.b = lds_tr4(X + r0 * HD + tofs[cc][1]);
    return u.v;
  };
  const float c1 = scale * LOG2E;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  unsigned mbits = 0;                                               // padded keys of the last key tile (dQ phase), one bit per accumulator register
#pragma unroll
  for (int r = 0; r < 16; ++r) mbits |= ((ntile - 1) * 32 + crow32(r, hi) >= valid ? 1u : 0u) << r;
  asm volatile("" : "+v"(mbits));
  // The loop is ROTATED: iteration h = -1 only issues the loads of head 0.  The register-destination loads (kfn, vfn, plse, pd) then have
  // ONE definition site (load_regs below) and ONE wait site (bwd_wait_head): with a peeled prologue the compiler merged the two
  // definitions through register copies at the loop's back edge -- copies that read the destination registers while the loads may
  // still be in flight (nothing interlocks a VGPR read against an outstanding VMEM load; an asm load counts as complete at its
  // statement).  Checked in the ISA after every change to this kernel (tools/check_attn_bwd_isa.py).
  for (int h = -1; h < H; ++h) {
    bf16x8 kf[4], vf[4];
    f32x16 dk0, dk1, dv0, dv1; zero16(dk0); zero16(dk1); zero16(dv0); zero16(dv1);
    if (h >= 0) {
    // Q, dO images (LDS-DMA) and this wave's K / V fragments, lse, D of head h have landed; only the previous head's dQ stores are
    // younger (head 0: nothing is, so everything is waited for -- by a statement without register operands)
    if (h == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bwd_wait_head<ST_DQ>(kfn, vfn, plse, pd);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { kf[ks] = kfn[ks]; vf[ks] = vfn[ks]; }
    if (tid < NP) { sLse[tid] = -plse * LOG2E; sD[tid] = pd; }         // exponent offset of P = exp2(c1 s - lse log2 e)
    __syncthreads();                                               // everyone's share of Q / dO is in place, nobody reads the previous head's K / V any more
    dma_kv(h);                                                     // K, V images of this head: needed by the dQ phase only
    // ---------------- dK, dV : this wave owns keys [32 wid, 32 wid + 32)
    {
      if (k0 < valid) {
        const float kbias = (k0 + l31 >= valid) ? -3.0e38f : 0.f;  // padded key: P = 0 (reference: exp(-10000 + ...) == 0 in fp32)
        const bool padded = k0 + 32 > valid;                       // wave-uniform: only the block that straddles `valid` needs the bias
        // (Software-pipelining this loop -- S / dP of block i + 1, or dV / dK of block i - 1, next to the softmax arithmetic of block i -- was
        // measured in round 4: the extra live operands spill (104 / 36 B of scratch at 256 registers) and hipcc keeps the 16 MFMAs of a
        // block together whatever sched_group_barrier asks for; the loop is bound by exposed LDS-read latency -- every MFMA waits for
        // a fragment read issued right in front of it, there is no register left to read a block ahead.  profiles/r04_attn_ablate.txt)
        for (int i = 0; i < 8; ++i) {
          f32x16 sc, dp;
#if ATST_ATTN_ABL & 16
          sc = zero; dp = zero; asm volatile("" : "+v"(sc), "+v"(dp));
#else
          sc = mfma32(rfrag(sQ, i * 32, 0), kf[0], zero);
          dp = mfma32(rfrag(sDO, i * 32, 0), vf[0], zero);
#pragma unroll
          for (int ks = 1; ks < 4; ++ks) {
            sc = mfma32(rfrag(sQ, i * 32, ks), kf[ks], sc);
            dp = mfma32(rfrag(sDO, i * 32, ks), vf[ks], dp);
          }
#endif
          float pv[16], ds[16];                                    // ds without the softmax scale: applied once to dK at the end
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(sLse + i * 32 + 8 * g + 4 * hi);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(sD + i * 32 + 8 * g + 4 * hi);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int r = 4 * g + e;
#if ATST_ATTN_ABL & 1
              const float pr = sc[r];
#else
              float arg = fmaf(sc[r], c1, l4[e]);
              if (padded) arg += kbias;
              const float pr = fast_exp2(arg);
#endif
              pv[r] = pr;
              ds[r] = pr * (dp[r] - d4[e]);
            }
          }
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            bf16x8 pf = pack8(pv + 8 * t), dsf = pack8(ds + 8 * t);
#if ATST_ATTN_ABL & 2
            asm volatile("" :: "v"(pf), "v"(dsf));
#else
            dv0 = mfma32(tfrag(sDO, i * 32 + 16 * t, 0), pf, dv0);
            dv1 = mfma32(tfrag(sDO, i * 32 + 16 * t, 1), pf, dv1);
            dk0 = mfma32(tfrag(sQ, i * 32 + 16 * t, 0), dsf, dk0);
            dk1 = mfma32(tfrag(sQ, i * 32 + 16 * t, 1), dsf, dk1);
#endif
          }
        }
      }
    }
    }                                                              // h >= 0
    // ---------------- dQ : this wave owns queries [32 wid, 32 wid + 32)
    {
      const int q0 = k0;
      bf16x8 qf[4], dof[4];
      float Dq = 0.f, nlse = 0.f;
      if (h >= 0) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          qf[ks] = rfrag(sQ, q0, ks);
          dof[ks] = rfrag(sDO, q0, ks);
        }
        Dq = sD[q0 + l31]; nlse = sLse[q0 + l31];
      }
      // my K / V pieces have landed (issued a whole dK / dV phase ago: nothing younger is in flight); with the barrier: everyone's have,
      // and every wave holds its query rows, so the Q / dO images are dead: this wave's 32 rows of each first serve as the transposition
      // buffers of its dK / dV tiles (full-line stores), then take the next head's Q / dO (this wave's own LDS-DMA share: wave-private)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (h >= 0) {
#if ATST_ATTN_ABL & 8
        asm volatile("" :: "v"(dk0), "v"(dk1), "v"(dv0), "v"(dv1));
#else
        if constexpr (MODE != 0) gmax = fmaxf(amax16(dk0, amax16(dk1, 0.f)) * scale, amax16(dv0, amax16(dv1, gmax)));
        bf16* dk00 = p.dqkv + ((size_t)s * NP + k0) * ld + C + h * HD;
        if constexpr (MODE == 2) {
          uint8_t* dk8 = p.dqkv8 + ((size_t)s * NP + k0) * ld + C + h * HD;
          store_tile64_staged_e4m3(dk8, ld, dk0, dk1, scale, s8, reinterpret_cast<char*>(sQ) + wid * 4096, lane);
          store_tile64_staged_e4m3(dk8 + C, ld, dv0, dv1, 1.0f, s8, reinterpret_cast<char*>(sDO) + wid * 4096, lane);
        } else if (p.row_stores) {                                        // tuning hook 406: the round-3 row-per-lane stores (A/B only)
          store_row64(dk00 + (size_t)l31 * ld, dk0, dk1, scale, hi);
          store_row64(dk00 + (size_t)l31 * ld + C, dv0, dv1, 1.0f, hi);
        } else {
          store_tile64_staged(dk00, ld, dk0, dk1, scale, reinterpret_cast<char*>(sQ) + wid * 4096, lane);
          store_tile64_staged(dk00 + C, ld, dv0, dv1, 1.0f, reinterpret_cast<char*>(sDO) + wid * 4096, lane);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the buffers have been read back: the DMA below may overwrite them
#endif
      }
      if (h + 1 < H) {
        dma_qdo(h + 1);
        load_regs(h + 1);
      }
      if (h < 0) continue;                                         // the rotated iteration: loads of head 0 issued, nothing to compute
      f32x16 dq0, dq1; zero16(dq0); zero16(dq1);
      for (int j = 0; j < ((ATST_ATTN_ABL & 4) ? 0 : ntile); ++j) {
        f32x16 sc, dp;
        sc = mfma32(rfrag(sK, j * 32, 0), qf[0], zero);
        dp = mfma32(rfrag(sV, j * 32, 0), dof[0], zero);
#pragma unroll
        for (int ks = 1; ks < 4; ++ks) {
          sc = mfma32(rfrag(sK, j * 32, ks), qf[ks], sc);
          dp = mfma32(rfrag(sV, j * 32, ks), dof[ks], dp);
        }
        float ds[16];
        const unsigned mb = j == ntile - 1 ? mbits : 0u;           // only the last key tile can hold padded keys
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#if ATST_ATTN_ABL & 1
          float pr = sc[r];
#else
          float pr = fast_exp2(fmaf(sc[r], c1, nlse));
#endif
          if ((mb >> r) & 1u) pr = 0.f;
          ds[r] = pr * (dp[r] - Dq);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const bf16x8 dsf = pack8(ds + 8 * t);
          dq0 = mfma32(tfrag(sK, j * 32 + 16 * t, 0), dsf, dq0);
          dq1 = mfma32(tfrag(sK, j * 32 + 16 * t, 1), dsf, dq1);
        }
      }
      bf16* dqrow = p.dqkv + ((size_t)s * NP + q0 + l31) * ld + h * HD;
#if ATST_ATTN_ABL & 8
      asm volatile("" :: "v"(dq0), "v"(dq1));
#else
      if constexpr (MODE != 0) gmax = fmaxf(gmax, amax16(dq0, amax16(dq1, 0.f)) * scale);
      if constexpr (MODE == 2) store_row64_e4m3(p.dqkv8 + ((size_t)s * NP + q0 + l31) * ld + h * HD, dq0, dq1, scale, s8, hi);
      else store_row64(dqrow, dq0, dq1, scale, hi);
#endif
    }
    // (the barrier at the top of the next head separates this head's last K / V reads from the next K / V DMA)
  }
  // the site takes max |bf16(x)|: rounding is monotonic, so it is the rounded maximum
  if constexpr (MODE != 0) amax_post(p.q8_amax, bf2f(f2bf(wave_max(gmax))), lane, blockIdx.x * 8 + wid);
}

template <int NP> int fwd_lds() { return Geo<NP>::G * 2 * Geo<NP>::LDS_MAT * 2; }
template <int NP> int dkv_lds() { return Geo<NP>::G * (2 * Geo<NP>::LDS_MAT * 2 + 2 * NP * 4); }

template <int NP>
int launch_fwd(const AttnArgs& a, hipStream_t st) {
  static OncePerDevice done; int done_dev;
  const int lds = fwd_lds<NP>();
  if (done.need(done_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_kernel<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if constexpr (NP == 32) {
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_fwd_kernel<NP, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_fwd_kernel<NP, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
    if (e != hipSuccess) return (int)e;
    done.done(done_dev);
  }
  const int nblk = (a.S * a.H + Geo<NP>::G - 1) / Geo<NP>::G;
  ProfScope ps(PK_ATTN_FWD, 4.0 * a.S * a.H * (double)NP * NP * HD, st, (6.0 + (a.o ? 2.0 : 0.0) + (a.o8 ? 1.0 : 0.0)) * a.S * a.H * (double)NP * HD);     // QK^T + PV on the padded geometry
  if constexpr (NP == 32) {
    if (a.o8 && !a.o) { hipLaunchKernelGGL((attn_fwd_kernel<NP, 2>), dim3(nblk), dim3(256), lds, st, a); return (int)hipGetLastError(); }
    if (a.o8) { hipLaunchKernelGGL((attn_fwd_kernel<NP, 1>), dim3(nblk), dim3(256), lds, st, a); return (int)hipGetLastError(); }
  }
  hipLaunchKernelGGL(attn_fwd_kernel<NP>, dim3(nblk), dim3(256), lds, st, a);
  return (int)hipGetLastError();
}
template <int NP>
int launch_bwd(const AttnArgs& a, hipStream_t st) {
  static OncePerDevice done; int done_dev;
  const int lds1 = dkv_lds<NP>(), lds2 = fwd_lds<NP>();
  if (done.need(done_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds1);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
    if (e != hipSuccess) return (int)e;
    done.done(done_dev);
  }
  const int nblk = (a.S * a.H + Geo<NP>::G - 1) / Geo<NP>::G;
  {
    ProfScope ps(PK_ATTN_BWD_DKV, 8.0 * a.S * a.H * (double)NP * NP * HD, st, 12.0 * a.S * a.H * (double)NP * HD);   // S, dP, dV, dK
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<NP>, dim3(nblk), dim3(256), lds1, st, a);
  }
  {
    ProfScope ps(PK_ATTN_BWD_DQ, 6.0 * a.S * a.H * (double)NP * NP * HD, st, 10.0 * a.S * a.H * (double)NP * HD);    // S, dP (recomputed), dQ
    hipLaunchKernelGGL(attn_bwd_dq_kernel<NP>, dim3(nblk), dim3(256), lds2, st, a);
  }
  return (int)hipGetLastError();
}
}  // namespace

int g_bwd256 = 1;          // merged per-sequence backward kernel for NP = 256 (tuning hook 402/403); the round-3 version of it (compiler-scheduled transfers) is in git history: 310 vs 307 us, profiles/r04_attn_time.txt
int g_fwd256 = 2;          // NP = 256 forward: 2 = two-pass softmax + LDS-DMA head loop, 1 = online-softmax head loop, 0 = per-(sequence, head) kernel (tuning hooks 404 / 401 / 400)
int g_bwd32 = 1;           // 408 / 409: NP = 32 backward as two kernels (A/B) / one fused kernel (default)
int g_bwd_row_stores = 0;  // 406 / 407: NP = 256 backward dK / dV stores row-per-lane (A/B) / LDS-transposed full lines (default)
int g_q8_np32 = 1;         // 412 / 413: the NP = 32 kernels (forward: e4m3 copy of the output ; fused backward: e4m3-only dqkv) take part in the e4m3 step: off / on (round 6)
int g_fwd_q8 = 1;          // 410 / 411: fp8 forward, e4m3 copy of the attention output by a separate pass (A/B) / by the NP = 256 forward kernel itself (default)
void atst_attn_set_variant(int v) { if (v == 12 || v == 13) g_q8_np32 = v == 13; else if (v == 10 || v == 11) g_fwd_q8 = v == 11; else if (v == 8 || v == 9) g_bwd32 = v == 9; else if (v == 6 || v == 7) g_bwd_row_stores = v == 6; else if (v == 4) g_fwd256 = 2; else if (v >= 2) g_bwd256 = v - 2; else g_fwd256 = v; }

// the e4m3 output of the forward exists in the two-pass NP = 256 kernel only (engine.hip asks; otherwise it quantises the bf16 output in a pass)
bool atst_attn_fwd_q8_ok(int NP, int H) { return (NP == 256 && g_fwd_q8 && g_fwd256 == 2 && (size_t)NP * 3 * H * HD * 2 < (1u << 30)) || (NP == 32 && g_q8_np32 && g_fwd_q8); }
int atst_attn_fwd(const AttnArgs& a, hipStream_t st) {
  if (a.S <= 0) return ATST_OK;
  if (a.o8 ? !atst_attn_fwd_q8_ok(a.NP, a.H) : !a.o) return ATST_EINVAL;
  if (a.stride < 0 || a.stride > a.NP || (a.NP == 256 && a.stride != 0 && a.stride != 256)) return ATST_EINVAL;   // packed sequences: NP < 256 kernels only
  if (a.NP == 256 && g_fwd256 == 2 && (size_t)a.NP * 3 * a.H * HD * 2 < (1u << 30)) {
    static OncePerDevice done2; int done2_dev;
    if (done2.need(done2_dev)) {
      hipError_t e = hipFuncSetAttribute((const void*)attn_fwd256v2_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * F2_BUF);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_fwd256v2_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * F2_BUF);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_fwd256v2_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * F2_BUF);
      if (e != hipSuccess) return (int)e;
      done2.done(done2_dev);
    }
    ProfScope ps(PK_ATTN_FWD, 4.0 * a.S * a.H * 256.0 * 256.0 * HD, st, (6.0 + (a.o ? 2.0 : 0.0) + (a.o8 ? 1.0 : 0.0)) * a.S * a.H * 256.0 * HD);
    if (a.o8 && !a.o) hipLaunchKernelGGL(attn_fwd256v2_kernel<2>, dim3(a.S), dim3(512), 2 * F2_BUF, st, a);
    else if (a.o8) hipLaunchKernelGGL(attn_fwd256v2_kernel<1>, dim3(a.S), dim3(512), 2 * F2_BUF, st, a);
    else hipLaunchKernelGGL(attn_fwd256v2_kernel<0>, dim3(a.S), dim3(512), 2 * F2_BUF, st, a);
    return (int)hipGetLastError();
  }
  if (a.NP == 256 && g_fwd256) {
    static OncePerDevice done; int done_dev;
    if (done.need(done_dev)) {
      hipError_t e = hipFuncSetAttribute((const void*)attn_fwd256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * F256_BUF);
      if (e != hipSuccess) return (int)e;
      done.done(done_dev);
    }
    ProfScope ps(PK_ATTN_FWD, 4.0 * a.S * a.H * 256.0 * 256.0 * HD, st, 8.0 * a.S * a.H * 256.0 * HD);
    hipLaunchKernelGGL(attn_fwd256_kernel, dim3(a.S), dim3(512), 2 * F256_BUF, st, a);
    return (int)hipGetLastError();
  }
  switch (a.NP) {
    case 256: return launch_fwd<256>(a, st);
    case 128: return launch_fwd<128>(a, st);
    case 64: return launch_fwd<64>(a, st);
    case 32: return launch_fwd<32>(a, st);
  }
  return ATST_EINVAL;
}
// the e4m3 output of the backward exists in the merged NP = 256 kernel only (engine.hip asks before it plans a block's qkv gradient GEMMs)
bool atst_attn_bwd_q8_ok(int NP) { return (NP == 256 && g_bwd256) || (NP == 32 && g_bwd32 && g_q8_np32); }
int atst_attn_bwd(const AttnArgs& a, hipStream_t st) {
  if (a.S <= 0) return ATST_OK;
  if (a.dqkv8 && !(atst_attn_bwd_q8_ok(a.NP) && (a.dscratch || a.NP == 32) && a.q8_scale && a.q8_amax)) return ATST_EINVAL;
  if (a.stride < 0 || a.stride > a.NP || (a.NP == 256 && a.stride != 0 && a.stride != 256)) return ATST_EINVAL;
  if (a.NP == 256 && g_bwd256 && a.dscratch) {
    static OncePerDevice done; int done_dev;
    if (done.need(done_dev)) {
      hipError_t e = hipFuncSetAttribute((const void*)attn_bwd256_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, B256_LDS);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_bwd256_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, B256_LDS);
      if (e != hipSuccess) return (int)e;
      done.done(done_dev);
    }
    const long rows = (long)a.S * 256;
    {
      const size_t chunks = (size_t)rows * a.H * (HD / 8);
      size_t nb = (chunks + 255) / 256; if (nb > 65536) nb = 65536;
      hipLaunchKernelGGL(attn_rowdot_kernel, dim3((unsigned)nb), dim3(256), 0, st, a.d_o, (const bf16*)a.o, a.dscratch, a.S, a.H, 256);
    }
    ProfScope ps(PK_ATTN_BWD_DKV, 14.0 * a.S * a.H * 256.0 * 256.0 * HD, st, 16.0 * a.S * a.H * 256.0 * HD);
    AttnArgs a2 = a; a2.row_stores = g_bwd_row_stores;
    if (a.dqkv8) {
      if (!a.q8_scale || !a.q8_amax) return ATST_EINVAL;
      hipLaunchKernelGGL(attn_bwd256_kernel<2>, dim3(a.S), dim3(512), B256_LDS, st, a2, (const float*)a.dscratch);
    } else hipLaunchKernelGGL(attn_bwd256_kernel<0>, dim3(a.S), dim3(512), B256_LDS, st, a2, (const float*)a.dscratch);
    return (int)hipGetLastError();
  }
  if (a.NP == 32 && g_bwd32) {                                     // one wave per (sequence, head): both halves from one staging pass
    constexpr int LDS32 = 4 * (4 * 32 * A_LD * 2 + 2 * 32 * 4);
    static OncePerDevice done32; int done32_dev;
    if (done32.need(done32_dev)) {
      hipError_t e = hipFuncSetAttribute((const void*)attn_bwd32_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS32);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_bwd32_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS32);
      if (e != hipSuccess) return (int)e;
      done32.done(done32_dev);
    }
    ProfScope ps(PK_ATTN_BWD_DKV, 18.0 * a.S * a.H * 32.0 * 32.0 * HD, st, (a.dqkv8 ? 13.0 : 16.0) * a.S * a.H * 32.0 * HD);
    if (a.dqkv8) hipLaunchKernelGGL(attn_bwd32_kernel<2>, dim3((a.S * a.H + 3) / 4), dim3(256), LDS32, st, a);
    else hipLaunchKernelGGL(attn_bwd32_kernel<0>, dim3((a.S * a.H + 3) / 4), dim3(256), LDS32, st, a);
    return (int)hipGetLastError();
  }
  switch (a.NP) {
    case 256: return launch_bwd<256>(a, st);
    case 128: return launch_bwd<128>(a, st);
    case 64: return launch_bwd<64>(a, st);
    case 32: return launch_bwd<32>(a, st);
  }
  return ATST_EINVAL;
}
