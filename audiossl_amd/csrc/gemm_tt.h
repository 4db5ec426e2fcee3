// "Two teams" persistent bf16 GEMM (C = A B^T + epilogue): the epilogue of tile i runs UNDER the main loop of tile i + 1, inside one CU.
// Included by gemm.hip (inside its anonymous namespace, behind the epilogue helpers and gemm_p8.h).
//
// Why.  Every other nt kernel of this library holds ONE output tile per CU and runs its main loop and its epilogue one after the other: the
// matrix pipes idle while the tile is stored, the vector-memory path idles while it is multiplied (rounds 2-5: at
// K = 384 the epilogue is half of the launch).  Two tiles per CU as two independent blocks did not help (gemm_nt_w4_kernel): co-resident
// blocks start together and stay in phase.  Here the phase opposition is built in:
//   * one persistent block per CU (grid = number of CUs), 8 waves = two TEAMS of four (waves w and w + 4 share a SIMD, so every SIMD
//     holds one wave of each team), block tile 256 x 128, wave tile 128 x 64 = 8 accumulators of 32 x 32 (128 registers) -- ONE
//     accumulator set per wave, 256 registers per wave as before; the two tiles in flight belong to different waves;
//   * the block walks its tiles in ROUNDS.  In round r team (r & 1) multiplies tile r (role ML) while the other team stores tile r - 1
//     (role EP) from the accumulators it filled in round r - 1; then the roles swap.  The SIMD's issue arbiter interleaves the two
//     instruction streams; nothing is interleaved by hand;
//   * ONE operand stream for both teams: a round is K / 32 STAGES (A 256 x 32 + B 128 x 32 bf16 = 24 KB, rows of 64 B, chunk c of row r at
//     c ^ ((r >> 2) & 3) as in gemm_nt_row384_kernel), a ring of 5 stages (120 KB) filled by LDS-DMA four stages ahead -- across tile
//     boundaries, so no tile starts on a cold ring.  Every wave of both teams issues 3 of a stage's 24 one-KiB pieces per iteration,
//     whatever its role: the vector-memory queue of every wave has the same static shape, and the only operand wait is a counted
//     `s_waitcnt vmcnt(N)` (vmcnt retires loads AND stores in issue order) -- N = 6 for a wave that stores nothing, 6 + the stores and
//     epilogue loads it has issued since, known at compile time per stage, for the EP role;
//   * one `s_barrier` per stage for all eight waves: it publishes stage s + 1, frees the slot of stage s - 1, and keeps the EP team's
//     slices in step with the ML team's stages.  The ML wave is software-pipelined over k-steps (fragments of the next k-step are read
//     while the 8 MFMAs of this one execute; two register sets of 6 fragments);
//   * EP role: wave-private staging (32 rows x 64 columns fp32 per wave, no block barrier in the epilogue): accumulator block row mb is
//     written to LDS in stage 2 mb, read back as 16-B pieces of full 128-B output lines and stored in stages 2 mb + 1 and 2 mb + 2; the
//     last store leaves in stage 8 of at least 12, so a wave that turns ML again has no store in its wait window.
// LDS: 120 KB ring + 4 x 8.5 KB staging (the teams alternate in the EP role and share it) = 154 KB, one block per CU.
// Shapes: M % 256 == 0, N % 128 == 0, K % 32 == 0, K >= 384.  Reference math: nn.Linear / GELU of audiossl/modules/transformer.py:70-92,
// 109,119 and their input gradients.
#ifndef ATST_TT_ABL            // experiment builds (tools/tt_ablate.sh): 1 = the EP role does nothing but keep step ; 4 = no fragment reads / MFMAs ; 8 = whole-line stream sources
#define ATST_TT_ABL 0
#endif
#ifndef ATST_TT_PRIO           // s_setprio: 0 none ; 1 the ML role runs at priority 2 ; 2 the EP role runs at priority 2
#define ATST_TT_PRIO 1
#endif
#ifndef ATST_TT_ISS            // who issues the operand stream: 0 = every wave 3 pieces per stage ; 1 = the four waves of the ML team 6 each, the other team none
#define ATST_TT_ISS 1
#endif
namespace tt {
constexpr int BM = 256, BNT = 128, BKT = 32, ROWB = 64, A_BYTES = BM * ROWB, B_BYTES = BNT * ROWB, STAGE = A_BYTES + B_BYTES, NS = 5, RING = NS * STAGE;
constexpr int SP = 68, STG_WAVE = 32 * SP * 4, LDS_BYTES = RING + 4 * STG_WAVE;     // staging rows of 64 floats + 4 (16-B aligned)
constexpr int THREADS = 512, EP_STAGES = 12;
constexpr int ISS = ATST_TT_ISS, NPW = ISS ? 6 : 3;                 // pieces per issuing wave per stage
static_assert(LDS_BYTES <= 163840, "one block per CU");
// vector-memory operations the EP role issues in stage s besides LDS-DMA pieces: P = register loads (in front of its pieces), S = stores (behind them)
template <int EPI, bool SU> constexpr int ep_stores(int s) { return (ATST_TT_ABL & 1) ? 0 : (s >= 1 && s <= 8) ? ((EPI == EPI_BIAS_GELU && SU ? 2 : 1) * 2) : 0; }   // SU: fc1 + GELU also saves u
template <int EPI> constexpr int ep_loads(int s) { return s == 0 ? 2 : 0; }          // bias: two 16-B loads per lane
// Operations that may stay outstanding at the head of EP stage s, i.e. everything this wave has issued behind the pieces of stage s + 1 (which it
// issued three stages earlier).  -1: the wave has no piece of that stage in flight -- no wait.
//   ISS 0 (every wave issues): stores of s - 3, then [loads, 3 pieces, stores] of s - 2 and s - 1; stages < 0 are the previous round (ML / idle role).
//   ISS 1 (ML team issues): the wave was ML in the previous round, so only stages 1 .. 3 of this round are its own: behind them the pieces of the
//   previous round's later stages (6 each) and whatever the EP role has issued so far.  Its stores are never IN FRONT of a piece it waits for:
//   vmcnt retires in order, and a store's acknowledgement takes several stage times -- with ISS 0 every operand wait of the EP role sat behind one.
template <int EPI, bool SU> constexpr int ep_vmcnt(int s) {
  if (ISS) {
    if (s >= 3) return -1;
    int n = 6 * (2 - s);
    for (int i = 0; i < s; ++i) n += ep_loads<EPI>(i) + ep_stores<EPI, SU>(i);
    return n;
  }
  int n = 6;
  for (int d = 1; d <= 3; ++d) n += (s - d >= 0 ? ep_stores<EPI, SU>(s - d) : 0);
  for (int d = 1; d <= 2; ++d) n += (s - d >= 0 ? ep_loads<EPI>(s - d) : 0);
  return n;
}
}

template <int N> DEVFN void tt_wait_tied(f32x4& a, f32x4& b) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory"); }
DEVFN void tt_gload16x2(const void* src, f32x4& a, f32x4& b) {
  asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16" : "=&v"(a), "=&v"(b) : "v"(src) : "memory");
}

#ifdef ATST_TT_TRACE
// s_memtime stamps of one block (tools/tt_trace.py): scalar stores (they do not touch vmcnt); [wave][round][stage][4] 64-bit, rounds TR0 .. TR0 + 3
DEVFN void tt_stamp(unsigned long long* base, unsigned off_bytes) {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\ts_store_dwordx2 %0, %1, %2" : "=&s"(t) : "s"(base), "s"(off_bytes) : "memory");
}
#endif
template <int EPI, bool SU>
__global__ __launch_bounds__(512, 2) void gemm_nt_tt_kernel(GemmArgs p, int T) {
  using namespace tt;
  static_assert(EPI == EPI_BF16 || EPI == EPI_BIAS_GELU, "epilogues carried so far");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wid >> 2, tw = wid & 3, wr = tw >> 1, wc = tw & 1, hi = lane >> 5, l31 = lane & 31;
  const int G = gridDim.x, rb = xcd_remap(blockIdx.x, G);
  const int ntn = p.N / BNT;
  const int nt = rb < T ? (T - rb + G - 1) / G : 0;                 // tiles of this block: ids rb, rb + G, ...
  if (nt == 0) return;
  const int nk = p.K / BKT;                                         // stages per tile, >= EP_STAGES (launcher)
#ifdef ATST_TT_TRACE
  int tr_r = -1, tr_s = 0;                                          // round / stage of the stamps (set by the round loop / advanced by other_stream, ml_stage)
  auto stamp = [&](int k) {
    if (blockIdx.x == ATST_TT_TRACE - 1 && tr_r >= 2 && tr_r < 6 && tr_s < 48)
      tt_stamp(reinterpret_cast<unsigned long long*>(p.colsum), (unsigned)(((((wid * 4 + (tr_r - 2)) * 48 + tr_s) * 4) + k) * 8));
  };
#else
  auto stamp = [&](int) {};
#endif

  // ---- operand stream (LDS-DMA): a stage is 16 A pieces (16 rows each) + 8 B pieces of 1 KiB; lane (lrow, chunk) of a piece fetches 16 B.
  //   ISS 0: wave w issues A pieces w, w + 8 and B piece w.   ISS 1: wave tw of the ML team issues A pieces 4 tw .. 4 tw + 3 and B pieces 2 tw, 2 tw + 1.
  const int lrow = lane >> 2, lch = (lane & 3) ^ ((lrow >> 2) & 3);
  const unsigned voA = (unsigned)((lrow * p.lda + lch * 8) * 2), voB = (unsigned)((lrow * p.ldb + lch * 8) * 2);
  const size_t pa = (size_t)16 * p.lda * 2, pb = (size_t)16 * p.ldb * 2;   // bytes between pieces
  const unsigned lds0 = lds_addr(smem_raw);
  int cur_j = 0, cur_s = 0;
  const char* curA; const char* curB;
  auto tile_mn = [&](int j, int& m0, int& n0) { const int id = j * G + rb, mi = id / ntn; m0 = mi * BM; n0 = (id - mi * ntn) * BNT; };
  auto set_tile = [&](int j) {
    int m0, n0; tile_mn(j, m0, n0);
    curA = sgpr_ptr(reinterpret_cast<const char*>(p.A) + (size_t)m0 * p.lda * 2);
    curB = sgpr_ptr(reinterpret_cast<const char*>(p.B) + (size_t)n0 * p.ldb * 2);
  };
  auto dma_piece = [&](int i, int slot) {                             // i-th piece of this wave (0 .. NPW - 1) of the stage the cursor points at
    const bool is_b = ISS ? i >= 4 : i == 2;
    const int idx = ISS ? (is_b ? 2 * tw + (i - 4) : 4 * tw + i) : (is_b ? wid : wid + 8 * i);
    const unsigned dst = lds0 + slot * STAGE + (is_b ? A_BYTES : 0) + idx * 1024;
#if ATST_TT_ABL & 8                                                   // whole-line sources (8 rows x 128 B per piece; stages 2 k / 2 k + 1 = upper / lower rows of k-tile k): same bytes, garbage results
    const size_t odd = cur_s & 1;
    const unsigned vfA = (unsigned)(((lane >> 3) * p.lda + (lane & 7) * 8) * 2), vfB = (unsigned)(((lane >> 3) * p.ldb + (lane & 7) * 8) * 2);
    if (is_b) p8_glds16(vfB, curB + (idx + 8 * odd) * (pb / 2) - odd * 64, dst); else p8_glds16(vfA, curA + (idx + 16 * odd) * (pa / 2) - odd * 64, dst);
    return;
#endif
    if (is_b) p8_glds16(voB, curB + idx * pb, dst); else p8_glds16(voA, curA + idx * pa, dst);
  };
  auto dma_advance = [&]() {                                          // the stream continues into the block's next tile; past the last one it re-reads it (never consumed)
    curA += ROWB; curB += ROWB;
    if (++cur_s == nk) { cur_s = 0; if (cur_j + 1 < nt) ++cur_j; set_tile(cur_j); }
  };
  int slot_c = 0;                                                     // ring slot of the stage being multiplied
  auto slot_next = [&]() { return slot_c == NS - 1 ? 0 : slot_c + 1; };
  auto slot_issue = [&]() { return slot_c == 0 ? NS - 1 : slot_c - 1; };   // stage s + 4 goes where stage s - 1 was

  // ---- ML role: fragments (row l31 of a 32-row block, chunk (2 ks + hi) ^ key) and MFMAs
  const int keyf = (l31 >> 2) & 3;
  const int fA0 = (wr * 128 + l31) * ROWB + ((hi ^ keyf) << 4), fB0 = A_BYTES + (wc * 64 + l31) * ROWB + ((hi ^ keyf) << 4);
  f32x16 acc[4][2];
  auto rd = [&](int slot, int ks, bf16x8 (&fa)[4], bf16x8 (&fb)[2]) {
#if ATST_TT_ABL & 4
    return;
#endif
    const char* s = smem_raw + slot * STAGE;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) fb[nb] = *reinterpret_cast<const bf16x8*>(s + ((fB0 ^ (ks << 5)) + nb * 32 * ROWB));
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) fa[mb] = *reinterpret_cast<const bf16x8*>(s + ((fA0 ^ (ks << 5)) + mb * 32 * ROWB));
  };
  auto mm2 = [&](auto first, int mb, bf16x8 (&fa)[4], bf16x8 (&fb)[2]) {           // two MFMAs: block row mb
    constexpr bool FIRST = decltype(first)::value;
#if ATST_TT_ABL & 4
    return;
#endif
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      if constexpr (FIRST) {
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[mb][nb] = mfma32(fa[mb], fb[nb], z);
      } else {
        acc[mb][nb] = mfma32(fa[mb], fb[nb], acc[mb][nb]);
      }
    }
  };
  auto head = [&](auto ntag) {
    constexpr int N = decltype(ntag)::value;
    stamp(0);
    if constexpr (N >= 0) p8_wait_vm<(N >= 0 ? N : 0)>();
    stamp(1);
    asm volatile("s_barrier" ::: "memory");
    stamp(2);
  };
#define TT_I(x) std::integral_constant<int, x>{}
#define TT_B(x) std::integral_constant<bool, x>{}
#define TT_FENCE() __builtin_amdgcn_sched_barrier(0)

  bf16x8 xa[4], xb[2], ya[4], yb[2];
  // One stage of the ML role: [head] fragments of k-step 1 | 8 MFMAs of k-step 0 | fragments of the next stage's k-step 0 | 8 MFMAs of k-step 1, the wave's
  // pieces of stage s + 4 between the MFMA pairs (a piece issued behind an MFMA pair blocks the wave while the matrix pipe is busy anyway).
  auto ml_stage = [&](auto first, bool more) {
    head(TT_I(2 * NPW));
    const int si = slot_issue();
    auto pc = [&](int i) { TT_FENCE(); dma_piece(i, si); TT_FENCE(); };
    rd(slot_c, 1, ya, yb);
    TT_FENCE();
    mm2(first, 0, xa, xb); if (ISS) pc(0);
    mm2(first, 1, xa, xb); pc(ISS ? 1 : 0);
    mm2(first, 2, xa, xb); if (ISS) pc(2);
    mm2(first, 3, xa, xb); if (!ISS) pc(1);
    TT_FENCE();
    if (more) rd(slot_next(), 0, xa, xb);                            // (stage s + 1 is visible since this stage's barrier)
    TT_FENCE();
    mm2(TT_B(false), 0, ya, yb); if (ISS) pc(3);
    mm2(TT_B(false), 1, ya, yb); pc(ISS ? 4 : 2);
    mm2(TT_B(false), 2, ya, yb); if (ISS) pc(5);
    mm2(TT_B(false), 3, ya, yb);
    TT_FENCE();
    dma_advance();
    slot_c = slot_next();
    stamp(3);
#ifdef ATST_TT_TRACE
    ++tr_s;
#endif
  };
  auto other_stream = [&]() {                                         // what a wave that is NOT multiplying does for the stream in one stage
    if constexpr (!ISS) { const int si = slot_issue(); dma_piece(0, si); dma_piece(1, si); dma_piece(2, si); }
    dma_advance();
    slot_c = slot_next();
  };
  auto idle_stage = [&]() {
    if constexpr (ISS) head(TT_I(-1)); else head(TT_I(6));
    other_stream();
    stamp(3);
#ifdef ATST_TT_TRACE
    ++tr_s;
#endif
  };

  // ---- EP role
  float* stg = reinterpret_cast<float*>(smem_raw + RING + tw * STG_WAVE);
  int ep_m0 = 0, ep_n0 = 0;
  f32x4 bia0, bia1;
  auto ep_stage = [&](auto stag, auto fintag) {
    constexpr int S = decltype(stag)::value;
    constexpr bool FIN = decltype(fintag)::value;                   // the block's last round: nothing is multiplied any more -- no stream, no barrier
    if constexpr (!FIN) head(TT_I((ep_vmcnt<EPI, SU>(S))));
    if constexpr (S == 0) {                                          // bias of this lane's 8 columns (a valid dummy address when the GEMM has none)
      const int col = ep_n0 + wc * 64 + (lane & 7) * 8;
      const float* src = p.bias ? p.bias + col : reinterpret_cast<const float*>(p.B) + (lane & 7) * 8;
      tt_gload16x2(src, bia0, bia1);
    }
    if constexpr (!FIN) other_stream();
    if constexpr (S == 1) {                                          // behind the bias loads: (ISS 0) 3 pieces of stage 0, 3 of stage 1
      if constexpr (FIN || ISS) tt_wait_tied<0>(bia0, bia1); else tt_wait_tied<6>(bia0, bia1);
      if (!p.bias) { bia0 = f32x4{0.f, 0.f, 0.f, 0.f}; bia1 = bia0; }
    }
#if ATST_TT_ABL & 1
    return;
#endif
    if constexpr (S >= 1 && S <= 8) {                                // read back + store: block row q, local rows 8 j + (lane >> 3), 8 columns per lane
      constexpr int q = (S - 1) / 2, jb = ((S - 1) & 1) * 2;
#pragma unroll
      for (int j = jb; j < jb + 2; ++j) {
        const int rl = j * 8 + (lane >> 3), c8 = (lane & 7) * 8;
        f32x4 v0 = *reinterpret_cast<const f32x4*>(stg + rl * SP + c8) + bia0;
        f32x4 v1 = *reinterpret_cast<const f32x4*>(stg + rl * SP + c8 + 4) + bia1;
        const size_t idx = (size_t)(ep_m0 + wr * 128 + q * 32 + rl) * p.ldc + ep_n0 + wc * 64 + c8;
        if constexpr (EPI == EPI_BF16) {
          const float t[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          st_pol<6>(pack8(t), reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.C) + idx));
        } else if constexpr (EPI == EPI_BIAS_GELU) {
          const float t[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          if constexpr (SU) st_pol<0>(pack8(t), reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.C) + idx));        // pre-activation u (training)
          float g[8];
#pragma unroll
          for (int e = 0; e < 8; e += 2) { const f32x2 a = gelu_bf16dst2(f32x2{t[e], t[e + 1]}); g[e] = a[0]; g[e + 1] = a[1]; }
          st_pol<5>(pack8(g), reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.C2) + idx));
        }
      }
    }
    if constexpr (S % 2 == 0 && S <= 6) {                            // stage the next block row (LDS operations of a wave execute in order: behind the reads above)
      constexpr int q = S / 2;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) stg[crow32(r, hi) * SP + nb * 32 + l31] = acc[q][nb][r];
    }
  };
  auto ep_round = [&](auto fintag) {
#ifdef ATST_TT_TRACE
    static_for<EP_STAGES>([&](auto s) { ep_stage(s, fintag); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamp(3); ++tr_s; });
#else
    static_for<EP_STAGES>([&](auto s) { ep_stage(s, fintag); });
#endif
    if constexpr (!decltype(fintag)::value) { for (int s = EP_STAGES; s < nk; ++s) idle_stage(); }
  };

  // ---- prologue: stages 0 .. 3 of the stream (ISS 1: issued by team 0, which multiplies first); stage 0 visible
  set_tile(0);
#pragma unroll
  for (int st = 0; st < NS - 1; ++st) {
    if (!ISS || team == 0) {
#pragma unroll
      for (int i = 0; i < NPW; ++i) dma_piece(i, st);
    }
    dma_advance();
  }
  p8_wait_vm<3 * NPW>();
  asm volatile("s_barrier" ::: "memory");

  for (int r = 0; r <= nt; ++r) {
#ifdef ATST_TT_TRACE
    tr_r = r; tr_s = 0;
#endif
    if (team == (r & 1)) {
      if (r < nt) {                                                  // ML: tile r
        if (ATST_TT_PRIO == 1) __builtin_amdgcn_s_setprio(2);
        rd(slot_c, 0, xa, xb);
        ml_stage(TT_B(true), true);
        for (int s = 1; s < nk; ++s) ml_stage(TT_B(false), s + 1 < nk);
        if (ATST_TT_PRIO == 1) __builtin_amdgcn_s_setprio(0);
      }                                                              // (r == nt: this team has nothing left -- the other one stores the last tile alone)
    } else {
      if (r >= 1) {                                                  // EP: tile r - 1
        tile_mn(r - 1, ep_m0, ep_n0);
        if (ATST_TT_PRIO == 2) __builtin_amdgcn_s_setprio(2);
        if (r < nt) ep_round(TT_B(false)); else ep_round(TT_B(true));
        if (ATST_TT_PRIO == 2) __builtin_amdgcn_s_setprio(0);
      } else {
        for (int s = 0; s < nk; ++s) idle_stage();                   // round 0: nothing to store yet
      }
    }
  }
#ifdef ATST_TT_TRACE
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
#endif
#undef TT_I
#undef TT_B
#undef TT_FENCE
}
