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
//   * ONE operand stream for both teams, issued by the ML team: a round is K / 64 STAGES (A 256 x 64 + B 128 x 64 bf16 = 48 KB; rows of
//     128 B = whole cache lines -- with 64-B rows the same bytes streamed at 28 B/clk/CU, with whole lines at 44, profiles/r06_tt_v1_*;
//     chunk c of row r at c ^ ((r >> 1) & 7) as in gemm_p8.h), a ring of 3 stages (144 KB) filled by LDS-DMA two stages ahead -- across
//     tile boundaries, so no tile starts on a cold ring.  The stream's waits are `vmcnt(0)` of the ML role (whose queue holds nothing but
//     its 12 pieces per stage); the EP role's stores are never in front of a piece anyone waits for (vmcnt retires in issue order and a
//     store's acknowledgement takes several stage times);
//   * one `s_barrier` per stage for all eight waves: it publishes stage s + 1, frees the slot of stage s - 1, and keeps the EP team's
//     slices in step with the ML team's stages.  The ML wave is software-pipelined over k-steps (fragments of the next k-step are read
//     while the 8 MFMAs of this one execute; two register sets of 6 fragments);
//   * TRANSPOSED accumulators (the weight fragment is the MFMA's A operand): a lane owns ONE output row and four consecutive columns per
//     register quad, so bias / GELU run on the registers, four values are packed to bf16 and staged with one ds_write_b64 (v1 staged fp32
//     with 128 ds_write_b32 per tile and wave: 20 cycles each next to the other team's fragment reads) into a wave-private 4-KB buffer
//     (32 rows x 128 B, 16-B chunks XOR-permuted by the row), read back as 16-B pieces of whole 128-B output lines.  No block barrier in
//     the epilogue.  Block row q of the tile is staged in stage q + 1, read back while block row q + 1 is converted, stored in stage q + 2.
// LDS: 144 KB ring + 4 x 4 KB staging (the teams alternate in the EP role and share it) = 160 KB, one block per CU.
// Shapes: M % 256 == 0, N % 128 == 0, K % 64 == 0, K >= 384.  Reference math: nn.Linear / GELU of audiossl/modules/transformer.py:70-92,
// 109,119 and their input gradients.
#ifndef ATST_TT_ABL            // experiment builds (tools/tt_ablate.sh): 1 = the EP role does nothing but keep step ; 4 = no fragment reads / MFMAs
#define ATST_TT_ABL 0
#endif
#ifndef ATST_TT_PRIO           // s_setprio: 0 none ; 1 the ML role runs at priority 2 ; 2 the EP role runs at priority 2
#define ATST_TT_PRIO 1
#endif
namespace tt {
constexpr int BM = 256, BNT = 128, BKT = 64, ROWB = 128, A_BYTES = BM * ROWB, B_BYTES = BNT * ROWB, STAGE = A_BYTES + B_BYTES, NS = 3, RING = NS * STAGE;
constexpr int STG_WAVE = 32 * 128, LDS_BYTES = RING + 4 * STG_WAVE;
constexpr int THREADS = 512, MIN_STAGES = 6;
constexpr int NPW = 12;                                             // pieces per ML wave per stage: 8 of A (8 rows each), 4 of B
static_assert(LDS_BYTES <= 163840, "one block per CU");
}
typedef unsigned int tt_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int tt_u32x4 __attribute__((ext_vector_type(4)));

DEVFN void tt_gload16(const void* src, f32x4& a) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a) : "v"(src) : "memory"); }
DEVFN void tt_wait_all(f32x4 (&b)[2][4]) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[0][3]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2]), "+v"(b[1][3]) :: "memory");
}
#ifdef ATST_TT_TRACE
// s_memtime stamps of one block (tools/tt_trace.py): scalar stores (they do not touch vmcnt); [wave][round][stage][4] 64-bit, rounds 2 .. 5
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
  const int nk = p.K / BKT;                                         // stages per tile, >= MIN_STAGES (launcher)
#ifdef ATST_TT_TRACE
  int tr_r = -1, tr_s = 0;
  auto stamp = [&](int k) {
    if (blockIdx.x == ATST_TT_TRACE - 1 && tr_r >= 2 && tr_r < 6 && tr_s < 48)
      tt_stamp(reinterpret_cast<unsigned long long*>(p.colsum), (unsigned)(((((wid * 4 + (tr_r - 2)) * 48 + tr_s) * 4) + k) * 8));
  };
#define TT_NEXT_STAGE() ++tr_s
#else
  auto stamp = [&](int) {};
#define TT_NEXT_STAGE()
#endif

  // ---- operand stream (LDS-DMA), issued by the ML team.  A stage = 32 A pieces + 16 B pieces of 1 KiB = 8 rows x 128 B; lane (r8, c) of a piece fetches
  // chunk c ^ key(row) of its row, key = (row >> 1) & 7 = 4 (piece & 1) | (r8 >> 1).  Wave tw issues A pieces 8 tw .. 8 tw + 7 and B pieces 4 tw .. 4 tw + 3.
  const int r8 = lane >> 3, c8l = lane & 7;
  unsigned voA[2], voB[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int ch = c8l ^ (4 * par + (r8 >> 1));
    voA[par] = (unsigned)((r8 * p.lda + ch * 8) * 2); voB[par] = (unsigned)((r8 * p.ldb + ch * 8) * 2);
  }
  const size_t pa = (size_t)8 * p.lda * 2, pb = (size_t)8 * p.ldb * 2;     // bytes between pieces
  const unsigned lds0 = lds_addr(smem_raw);
  int cur_j = 0, cur_s = 0;
  const char* curA; const char* curB;
  auto tile_mn = [&](int j, int& m0, int& n0) { const int id = j * G + rb, mi = id / ntn; m0 = mi * BM; n0 = (id - mi * ntn) * BNT; };
  auto set_tile = [&](int j) {
    int m0, n0; tile_mn(j, m0, n0);
    curA = sgpr_ptr(reinterpret_cast<const char*>(p.A) + (size_t)m0 * p.lda * 2);
    curB = sgpr_ptr(reinterpret_cast<const char*>(p.B) + (size_t)n0 * p.ldb * 2);
  };
  auto dma_piece = [&](int i, int slot) {                             // i-th piece of this wave (0 .. 11) of the stage the cursor points at
    const bool is_b = i >= 8;
    const int idx = is_b ? 4 * tw + (i - 8) : 8 * tw + i;
    const unsigned dst = lds0 + slot * STAGE + (is_b ? A_BYTES : 0) + idx * 1024;
    if (is_b) p8_glds16(voB[i & 1], curB + idx * pb, dst); else p8_glds16(voA[i & 1], curA + idx * pa, dst);     // (piece parity = i & 1: 8 tw and 4 tw are even)
  };
  auto dma_advance = [&]() {                                          // the stream continues into the block's next tile; past the last one it re-reads it (never consumed)
    curA += ROWB; curB += ROWB;
    if (++cur_s == nk) { cur_s = 0; if (cur_j + 1 < nt) ++cur_j; set_tile(cur_j); }
  };
  int slot_c = 0;                                                     // ring slot of the stage being multiplied
  auto slot_next = [&]() { return slot_c == NS - 1 ? 0 : slot_c + 1; };
  auto slot_issue = [&]() { return slot_c == 0 ? NS - 1 : slot_c - 1; };   // stage s + 2 goes where stage s - 1 was

  // ---- ML role: fragments (row l31 of a 32-row block, chunk (2 ks + hi) ^ key, key = (l31 >> 1) & 7) and MFMAs with the WEIGHT fragment as the A operand:
  // acc[mb][nb] is the TRANSPOSE of the 32 x 32 output block -- lane l holds output row l31 (of block row mb), register r column 8 (r >> 2) + 4 hi + (r & 3).
  const int keyf = (l31 >> 1) & 7;
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
        acc[mb][nb] = mfma32(fb[nb], fa[mb], z);
      } else {
        acc[mb][nb] = mfma32(fb[nb], fa[mb], acc[mb][nb]);
      }
    }
  };
  auto head = [&](bool wait) {
    stamp(0);
    if (wait) p8_wait_vm<0>();
    stamp(1);
    asm volatile("s_barrier" ::: "memory");
    stamp(2);
  };
#define TT_B(x) std::integral_constant<bool, x>{}
#define TT_FENCE() __builtin_amdgcn_sched_barrier(0)

  bf16x8 xa[4], xb[2], ya[4], yb[2];
  // One stage (k-tile of 64) of the ML role: [head] then four k-steps of 8 MFMAs; the fragments of k-step ks + 1 (k-step 0 of the next stage after the third) are
  // requested in front of the MFMAs of k-step ks; the wave's 12 pieces of stage s + 2 go between the MFMA pairs (three per k-step): a piece issued behind an
  // MFMA pair blocks the wave while the matrix pipe is busy anyway.
  auto ml_stage = [&](auto first, bool wait, bool more) {
    head(wait);
    const int si = slot_issue();
    auto pc = [&](int i) { TT_FENCE(); dma_piece(i, si); TT_FENCE(); };
    rd(slot_c, 1, ya, yb);
    TT_FENCE();
    mm2(first, 0, xa, xb); pc(0); mm2(first, 1, xa, xb); pc(1); mm2(first, 2, xa, xb); pc(2); mm2(first, 3, xa, xb);
    TT_FENCE();
    rd(slot_c, 2, xa, xb);
    TT_FENCE();
    mm2(TT_B(false), 0, ya, yb); pc(3); mm2(TT_B(false), 1, ya, yb); pc(4); mm2(TT_B(false), 2, ya, yb); pc(5); mm2(TT_B(false), 3, ya, yb);
    TT_FENCE();
    rd(slot_c, 3, ya, yb);
    TT_FENCE();
    mm2(TT_B(false), 0, xa, xb); pc(6); mm2(TT_B(false), 1, xa, xb); pc(7); mm2(TT_B(false), 2, xa, xb); pc(8); mm2(TT_B(false), 3, xa, xb);
    TT_FENCE();
    if (more) rd(slot_next(), 0, xa, xb);                            // (stage s + 1 is visible since this stage's barrier)
    TT_FENCE();
    mm2(TT_B(false), 0, ya, yb); pc(9); mm2(TT_B(false), 1, ya, yb); pc(10); mm2(TT_B(false), 2, ya, yb); pc(11); mm2(TT_B(false), 3, ya, yb);
    TT_FENCE();
    dma_advance();
    slot_c = slot_next();
    stamp(3);
    TT_NEXT_STAGE();
  };
  auto other_stream = [&]() { dma_advance(); slot_c = slot_next(); };      // a wave that is not multiplying only keeps its cursor in step
  auto idle_stage = [&]() { head(false); other_stream(); stamp(3); TT_NEXT_STAGE(); };

  // ---- EP role
  char* stg = smem_raw + RING + tw * STG_WAVE;
  const int wb0 = l31 * 128 + hi * 8 + ((l31 & 7) << 4);              // staging write: chunk ch of row l31 at (this ^ (ch << 4))
  const int rb0 = r8 * 128 + ((c8l ^ r8) << 4);                      // staging read-back: row 8 j + r8, chunk c8l
  const unsigned vst = (unsigned)((r8 * p.ldc + c8l * 8) * 2);         // byte offset of this lane's 16-B piece inside an 8-row group of the output
  int ep_m0 = 0, ep_n0 = 0;
  // L2 prefetch of the NEXT tile's A rows (round 6, after the in-step profile): with two stages of lead the stream hides an L2 hit, not an HBM miss --
  // stand-alone (operands resident in the 256-MB last-level cache after the first repetition) qkv ran 137 us, inside the training step 175.  In stage 1
  // of its round every EP wave touches one dword of every 128-B line of 64 rows of the tile the stream turns to at the end of this round: K / 64 loads per
  // wave; their common destination register is released in stage 5 behind a counted wait (the stores of stages 1 .. 4 were issued after them and may
  // stay in flight: vmcnt retires in order), four stages = several HBM latencies later.
  int touch_m0 = -1;
  float touch_reg = 0.f;
  auto touch_next = [&]() {
#ifndef ATST_TT_NO_TOUCH
    if (touch_m0 < 0) return;
    const char* a = reinterpret_cast<const char*>(p.A) + (size_t)(touch_m0 + tw * 64 + lane) * p.lda * 2;
    for (int j = 0; j < nk; ++j) { asm volatile("global_load_dword %0, %1, off" : "+v"(touch_reg) : "v"(a) : "memory"); a += 128; }
#endif
  };
  f32x4 bb[2][4];                                                     // bias of this lane's columns: [nb][g] = columns nb * 32 + 8 g + 4 hi .. + 3 of the wave's 64
  // Block row q of the tile as an ITEM: values -> bf16 -> staging (C) ... read back (R) ... whole-line stores (S).  PASS 0: acc + bias (plain output /
  // pre-activation u) ; PASS 1: GELU of it.  The three steps of an item sit in different places of the instruction stream (v2a: with C, R, S of one item
  // back to back a slice took 2.1-2.5 k cycles -- three exposed round trips through an LDS / vector-memory system the other team keeps saturated):
  // an item is read back while the next one is computed, and stored a stage after it was staged.
  auto ep_C = [&](auto qtag, auto passtag) __attribute__((always_inline)) {
    constexpr int q = decltype(qtag)::value, PASS = decltype(passtag)::value;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[q][nb][4 * g + e] + bb[nb][g][e];
        if constexpr (PASS == 1) {
          const f32x2 a0 = gelu_bf16dst2(f32x2{v[0], v[1]}), a1 = gelu_bf16dst2(f32x2{v[2], v[3]});
          v[0] = a0[0]; v[1] = a0[1]; v[2] = a1[0]; v[3] = a1[1];
        }
        *reinterpret_cast<tt_u32x2*>(stg + (wb0 ^ ((nb * 4 + g) << 4))) = __builtin_bit_cast(tt_u32x2, pack_bf16x4(v));
      }
  };
  auto ep_R = [&](tt_u32x4 (&w)[4]) __attribute__((always_inline)) {   // (LDS operations of a wave execute in order: the reads see the item staged before them, the next item's writes follow them)
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const tt_u32x4*>(stg + rb0 + j * 1024);
  };
  auto ep_S = [&](int q, const tt_u32x4 (&w)[4], void* out) __attribute__((always_inline)) {
    char* obase = reinterpret_cast<char*>(out) + ((size_t)(ep_m0 + wr * 128 + q * 32) * p.ldc + ep_n0 + wc * 64) * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(w[j], reinterpret_cast<tt_u32x4*>(obase + (size_t)j * 8 * p.ldc * 2 + vst));
  };
  tt_u32x4 wq[4];                                                     // an item between its read-back and its stores
  constexpr std::integral_constant<int, 0> P0{}; constexpr std::integral_constant<int, 1> P1{};
  auto ep_stage = [&](auto stag, bool fin) __attribute__((always_inline)) {   // fin: the block's last round -- nothing is multiplied any more: no barrier
    constexpr int S = decltype(stag)::value;
    if (!fin) head(S == 0);                                          // S == 0: this wave multiplied in the previous round and its pieces of stage 1 are still in flight
    if constexpr (S == 0) {
      const float* src = p.bias ? p.bias + ep_n0 + wc * 64 + 4 * hi : reinterpret_cast<const float*>(p.B) + 4 * hi;     // (a valid dummy address when the GEMM has no bias)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) tt_gload16(src + nb * 32 + 8 * g, bb[nb][g]);
    }
    if (!fin) other_stream();
    if constexpr (S == 1) {
      tt_wait_all(bb);
      if (!p.bias) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int g = 0; g < 4; ++g) bb[nb][g] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (!fin) touch_next();
    }
    if constexpr (S == 5) {                                          // the prefetch loads of stage 1 have landed: behind them only the 12 (u and a: 28) stores of stages 1 .. 4
      constexpr int TOUCH_N = (ATST_TT_ABL & 1) ? 0 : ((EPI == EPI_BIAS_GELU && SU) ? 28 : 12);
      if (!fin) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(touch_reg) : "n"(TOUCH_N) : "memory");
    }
#if !(ATST_TT_ABL & 1)
    constexpr std::integral_constant<int, (S >= 1 && S <= 4) ? S - 1 : 0> qc{};      // the block row staged in this stage
    if constexpr (EPI == EPI_BIAS_GELU && SU) {
      // items u0 a0 u1 a1 ... : stage S stages u(q), a(q) of q = S - 1, reads back a(q - 1), u(q), stores a(q - 1), u(q); stage 5: a(3)
      if constexpr (S >= 2 && S <= 5) ep_R(wq);                      // a(q - 1)
      if constexpr (S >= 1 && S <= 4) ep_C(qc, P0);
      if constexpr (S >= 2 && S <= 5) ep_S(S - 2, wq, p.C2);
      if constexpr (S >= 1 && S <= 4) { ep_R(wq); ep_C(qc, P1); ep_S(S - 1, wq, p.C); }
    } else {
      constexpr bool gelu = EPI == EPI_BIAS_GELU;
      if constexpr (S >= 2 && S <= 5) ep_R(wq);
      if constexpr (S >= 1 && S <= 4) { if constexpr (gelu) ep_C(qc, P1); else ep_C(qc, P0); }
      if constexpr (S >= 2 && S <= 5) ep_S(S - 2, wq, gelu ? p.C2 : p.C);
    }
#endif
#ifdef ATST_TT_TRACE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    stamp(3);
    TT_NEXT_STAGE();
  };
  auto ep_round = [&](bool fin) __attribute__((always_inline)) {
    static_for<6>([&](auto s) __attribute__((always_inline)) { ep_stage(s, fin); });
    if (!fin) { for (int s = 6; s < nk; ++s) idle_stage(); }
  };

  // ---- prologue: stages 0 and 1 of the stream, issued by team 0 (which multiplies first); stage 0 visible
  set_tile(0);
#pragma unroll
  for (int st = 0; st < NS - 1; ++st) {
    if (team == 0) {
#pragma unroll
      for (int i = 0; i < NPW; ++i) dma_piece(i, st);
    }
    dma_advance();
  }
  p8_wait_vm<NPW>();
  asm volatile("s_barrier" ::: "memory");

  for (int r = 0; r <= nt; ++r) {
#ifdef ATST_TT_TRACE
    tr_r = r; tr_s = 0;
#endif
    if (team == (r & 1)) {
      if (r < nt) {                                                  // ML: tile r
        if (ATST_TT_PRIO == 1) __builtin_amdgcn_s_setprio(2);
        rd(slot_c, 0, xa, xb);
        ml_stage(TT_B(true), r == 0, true);                          // (stage 1 of a later round was issued -- and is waited for -- by the other team)
        for (int s = 1; s < nk; ++s) ml_stage(TT_B(false), true, s + 1 < nk);
        if (ATST_TT_PRIO == 1) __builtin_amdgcn_s_setprio(0);
      }                                                              // (r == nt: this team has nothing left -- the other one stores the last tile alone)
    } else {
      if (r >= 1) {                                                  // EP: tile r - 1
        tile_mn(r - 1, ep_m0, ep_n0);
        { int tn0; touch_m0 = -1; if (r + 1 < nt) tile_mn(r + 1, touch_m0, tn0); }
        if (ATST_TT_PRIO == 2) __builtin_amdgcn_s_setprio(2);
        ep_round(r == nt);
        if (ATST_TT_PRIO == 2) __builtin_amdgcn_s_setprio(0);
      } else {
        for (int s = 0; s < nk; ++s) idle_stage();                   // round 0: nothing to store yet
      }
    }
  }
#ifdef ATST_TT_TRACE
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
#endif
#undef TT_B
#undef TT_FENCE
#undef TT_NEXT_STAGE
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------
// The SAME overlap as one instruction stream: the 4-wave, one-wave-per-SIMD, 512-register form VERDICT r5 item 1 sketched (tuning hook 2002).  A wave keeps
// TWO accumulator sets (2 x 128 registers): in round r it multiplies tile r into set r & 1 and, between the MFMA pairs of that main loop, runs the epilogue
// items of tile r - 1 out of the other set -- staged, read back and stored by the same micro-ops as the two-team kernel above, placed by hand: a stage is 16
// MFMA pairs = 16 slots; slots 0 .. 11 carry the wave's 12 LDS-DMA pieces of stage s + 2, slot 0 the read-back of the item staged a stage earlier, slots
// 1 .. 8 (u and a: 1 .. 4 and 6 .. 9) the conversion of this stage's item, slots 12 .. 15 the stores -- BEHIND the stage's pieces, so that the operand wait
// of the next stage (`vmcnt(N)`, N = the stores just issued) does not sit behind them.  Same tile (256 x 128), same ring, same staging, same stream.
template <int EPI, bool SU>
__global__ __launch_bounds__(256, 1) void gemm_nt_t1_kernel(GemmArgs p, int T) {
  using namespace tt;
  static_assert(EPI == EPI_BF16 || EPI == EPI_BIAS_GELU, "epilogues carried so far");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, tw = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = tw >> 1, wc = tw & 1, hi = lane >> 5, l31 = lane & 31;
  const int G = gridDim.x, rb = xcd_remap(blockIdx.x, G);
  const int ntn = p.N / BNT;
  const int nt = rb < T ? (T - rb + G - 1) / G : 0;
  if (nt == 0) return;
  const int nk = p.K / BKT;
  const int r8 = lane >> 3, c8l = lane & 7;
  unsigned voA[2], voB[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int ch = c8l ^ (4 * par + (r8 >> 1));
    voA[par] = (unsigned)((r8 * p.lda + ch * 8) * 2); voB[par] = (unsigned)((r8 * p.ldb + ch * 8) * 2);
  }
  const size_t pa = (size_t)8 * p.lda * 2, pb = (size_t)8 * p.ldb * 2;
  const unsigned lds0 = lds_addr(smem_raw);
  int cur_j = 0, cur_s = 0;
  const char* curA; const char* curB;
  auto tile_mn = [&](int j, int& m0, int& n0) { const int id = j * G + rb, mi = id / ntn; m0 = mi * BM; n0 = (id - mi * ntn) * BNT; };
  auto set_tile = [&](int j) {
    int m0, n0; tile_mn(j, m0, n0);
    curA = sgpr_ptr(reinterpret_cast<const char*>(p.A) + (size_t)m0 * p.lda * 2);
    curB = sgpr_ptr(reinterpret_cast<const char*>(p.B) + (size_t)n0 * p.ldb * 2);
  };
  auto dma_piece = [&](int i, int slot) {
    const bool is_b = i >= 8;
    const int idx = is_b ? 4 * tw + (i - 8) : 8 * tw + i;
    const unsigned dst = lds0 + slot * STAGE + (is_b ? A_BYTES : 0) + idx * 1024;
    if (is_b) p8_glds16(voB[i & 1], curB + idx * pb, dst); else p8_glds16(voA[i & 1], curA + idx * pa, dst);
  };
  auto dma_advance = [&]() {
    curA += ROWB; curB += ROWB;
    if (++cur_s == nk) { cur_s = 0; if (cur_j + 1 < nt) ++cur_j; set_tile(cur_j); }
  };
  int slot_c = 0;
  auto slot_next = [&]() { return slot_c == NS - 1 ? 0 : slot_c + 1; };
  auto slot_issue = [&]() { return slot_c == 0 ? NS - 1 : slot_c - 1; };
  const int keyf = (l31 >> 1) & 7;
  const int fA0 = (wr * 128 + l31) * ROWB + ((hi ^ keyf) << 4), fB0 = A_BYTES + (wc * 64 + l31) * ROWB + ((hi ^ keyf) << 4);
  f32x16 acc[2][4][2];
  bf16x8 xa[4], xb[2], ya[4], yb[2];
  // (every lane-dependent address below starts from a LAUNDERED copy of its base: derived from the base directly, hipcc hoists all the variants out of the
  // round loop -- dozens of registers beside 256 accumulators -- and spills them; a reload is a vector-memory operation inside the counted waits)
  auto rd = [&](int slot, int ks, bf16x8 (&fa)[4], bf16x8 (&fb)[2]) {
    const char* s = smem_raw + slot * STAGE;
    int a0 = fA0, b0 = fB0;
    asm volatile("" : "+v"(a0), "+v"(b0));
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) fb[nb] = *reinterpret_cast<const bf16x8*>(s + ((b0 ^ (ks << 5)) + nb * 32 * ROWB));
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) fa[mb] = *reinterpret_cast<const bf16x8*>(s + ((a0 ^ (ks << 5)) + mb * 32 * ROWB));
  };
  // ---- epilogue micro-ops (tile r - 1, accumulator set Q)
  char* stg = smem_raw + RING + tw * STG_WAVE;
  const int wb0 = l31 * 128 + hi * 8 + ((l31 & 7) << 4);
  const int rb0 = r8 * 128 + ((c8l ^ r8) << 4);
  const unsigned vst = (unsigned)((r8 * p.ldc + c8l * 8) * 2);
  int ep_m0 = 0, ep_n0 = 0;
  f32x4 bb[2][4];
  tt_u32x4 wq[4], wu[4];
  auto ep_c1 = [&](auto qtag, auto passtag, auto qstag, int k) __attribute__((always_inline)) {      // one (nb, g) group of item (block row q, PASS) out of set QS
    constexpr int q = decltype(qtag)::value, PASS = decltype(passtag)::value, QS = decltype(qstag)::value;
    const int nb = k >> 2, g = k & 3;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      // BOTH accumulator sets have to live in the accumulator half of the register file (2 x 128 of the 256 AGPRs): read through an "a"-constrained
      // v_accvgpr_read, or hipcc keeps the set the epilogue reads in arch VGPRs -- 128 of the 256 that fragments, bias, staging data and addresses share --
      // and spills it in 30-register clumps inside the round (each reload a vector-memory operation inside the counted waits).  The set was written a round
      // ago: no MFMA -> accvgpr_read hazard to pad.
      float av;
      asm("v_accvgpr_read_b32 %0, %1" : "=v"(av) : "a"(acc[QS][q][nb][4 * g + e]));
      v[e] = av + bb[nb][g][e];
    }
    if constexpr (PASS == 1) {
      const f32x2 a0 = gelu_bf16dst2(f32x2{v[0], v[1]}), a1 = gelu_bf16dst2(f32x2{v[2], v[3]});
      v[0] = a0[0]; v[1] = a0[1]; v[2] = a1[0]; v[3] = a1[1];
    }
    int w0 = wb0; asm volatile("" : "+v"(w0));
    *reinterpret_cast<tt_u32x2*>(stg + (w0 ^ ((nb * 4 + g) << 4))) = __builtin_bit_cast(tt_u32x2, pack_bf16x4(v));
  };
  auto ep_R = [&](tt_u32x4 (&w)[4]) __attribute__((always_inline)) {
    int r0 = rb0; asm volatile("" : "+v"(r0));
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const tt_u32x4*>(stg + r0 + j * 1024);
  };
  auto ep_S1 = [&](int q, int j, const tt_u32x4 (&w)[4], void* out) __attribute__((always_inline)) {
    char* obase = reinterpret_cast<char*>(out) + ((size_t)(ep_m0 + wr * 128 + q * 32) * p.ldc + ep_n0 + wc * 64) * 2;
    unsigned v0 = vst; asm volatile("" : "+v"(v0));
    __builtin_nontemporal_store(w[j], reinterpret_cast<tt_u32x4*>(obase + (size_t)j * 8 * p.ldc * 2 + v0));
  };
  constexpr std::integral_constant<int, 0> P0{}; constexpr std::integral_constant<int, 1> P1{};
  // the epilogue work of slot I of stage S (see the header): static indices everywhere
  auto ep_slot = [&](auto stag, auto itag, auto qstag) __attribute__((always_inline)) {
    constexpr int S = decltype(stag)::value, I = decltype(itag)::value;
    constexpr std::integral_constant<int, (S >= 1 && S <= 4) ? S - 1 : 0> qc{};
    constexpr bool stage_c = S >= 1 && S <= 4, stage_s = S >= 2 && S <= 5;
    if constexpr (EPI == EPI_BIAS_GELU && SU) {
      if constexpr (I == 0 && stage_s) ep_R(wq);                                          // a(q - 1)
      if constexpr (I >= 1 && I <= 4 && stage_c) { ep_c1(qc, P0, qstag, 2 * (I - 1)); ep_c1(qc, P0, qstag, 2 * (I - 1) + 1); }
      if constexpr (I == 5 && stage_c) ep_R(wu);                                          // u(q)
      if constexpr (I >= 6 && I <= 9 && stage_c) { ep_c1(qc, P1, qstag, 2 * (I - 6)); ep_c1(qc, P1, qstag, 2 * (I - 6) + 1); }
      if constexpr (I >= 12 && stage_s) ep_S1(S - 2, I - 12, wq, p.C2);
      if constexpr (I >= 12 && stage_c) ep_S1(S - 1, I - 12, wu, p.C);
    } else {
      constexpr bool gelu = EPI == EPI_BIAS_GELU;
      if constexpr (I == 0 && stage_s) ep_R(wq);
      if constexpr (I >= 1 && I <= 8 && stage_c) { if constexpr (gelu) ep_c1(qc, P1, qstag, I - 1); else ep_c1(qc, P0, qstag, I - 1); }
      if constexpr (I >= 12 && stage_s) ep_S1(S - 2, I - 12, wq, gelu ? p.C2 : p.C);
    }
  };
  // stores a stage issues behind its pieces = what may stay in flight at the head of the next stage
  auto stores_of = [](int s) constexpr { return (EPI == EPI_BIAS_GELU && SU) ? ((s >= 2 && s <= 4) ? 8 : (s == 1 || s == 5) ? 4 : 0) : ((s >= 2 && s <= 5) ? 4 : 0); };
#define TT_FENCE() __builtin_amdgcn_sched_barrier(0)
  // one stage: S < 0 = a stage without epilogue work (rolled loop) ; P = accumulator set of the main loop
  auto stage = [&](auto stag, auto ptag, auto first, auto mltag, auto eptag, bool more, int nwait) __attribute__((always_inline)) {
    constexpr int S = decltype(stag)::value, P = decltype(ptag)::value;
    constexpr bool do_ml = decltype(mltag)::value, do_ep = decltype(eptag)::value;
    constexpr bool FIRST = decltype(first)::value;
    if constexpr (do_ml) {
      if (nwait == 0) p8_wait_vm<0>(); else if (nwait == 4) p8_wait_vm<4>(); else p8_wait_vm<8>();
      asm volatile("s_barrier" ::: "memory");
    }
    if constexpr (S == 0) {
      if constexpr (do_ep) {                                        // bias of this lane's columns: the oldest vector-memory operations of the stage -- landed by the next head
        const float* src = p.bias ? p.bias + ep_n0 + wc * 64 + 4 * hi : reinterpret_cast<const float*>(p.B) + 4 * hi;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int g = 0; g < 4; ++g) tt_gload16(src + nb * 32 + 8 * g, bb[nb][g]);
      }
    }
    if constexpr (S == 1) {
      if constexpr (do_ep) {
        if constexpr (!do_ml) tt_wait_all(bb); else asm volatile("" : "+v"(bb[0][0]), "+v"(bb[0][1]), "+v"(bb[0][2]), "+v"(bb[0][3]), "+v"(bb[1][0]), "+v"(bb[1][1]), "+v"(bb[1][2]), "+v"(bb[1][3]));   // (with a main loop: this stage's head waited vmcnt(0))
        if (!p.bias) {
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g) bb[nb][g] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
    const int si = slot_issue();
    auto mm2 = [&](auto f, int mb, bf16x8 (&fa)[4], bf16x8 (&fb)[2]) __attribute__((always_inline)) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        if constexpr (decltype(f)::value) {
          const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc[P][mb][nb] = mfma32(fb[nb], fa[mb], z);
        } else {
          acc[P][mb][nb] = mfma32(fb[nb], fa[mb], acc[P][mb][nb]);
        }
      }
    };
    auto slot = [&](auto itag) __attribute__((always_inline)) {     // what follows MFMA pair I
      constexpr int I = decltype(itag)::value;
      TT_FENCE();
      if constexpr (I < 12 && do_ml) dma_piece(I, si);
      if constexpr (S >= 0 && do_ep) ep_slot(stag, itag, std::integral_constant<int, P ^ 1>{});
      TT_FENCE();
    };
    constexpr std::integral_constant<bool, false> NF{};
    if constexpr (do_ml) rd(slot_c, 1, ya, yb);
    TT_FENCE();
    if constexpr (do_ml) mm2(first, 0, xa, xb); slot(std::integral_constant<int, 0>{});
    if constexpr (do_ml) mm2(first, 1, xa, xb); slot(std::integral_constant<int, 1>{});
    if constexpr (do_ml) mm2(first, 2, xa, xb); slot(std::integral_constant<int, 2>{});
    if constexpr (do_ml) mm2(first, 3, xa, xb); slot(std::integral_constant<int, 3>{});
    if constexpr (do_ml) rd(slot_c, 2, xa, xb);
    TT_FENCE();
    if constexpr (do_ml) mm2(NF, 0, ya, yb); slot(std::integral_constant<int, 4>{});
    if constexpr (do_ml) mm2(NF, 1, ya, yb); slot(std::integral_constant<int, 5>{});
    if constexpr (do_ml) mm2(NF, 2, ya, yb); slot(std::integral_constant<int, 6>{});
    if constexpr (do_ml) mm2(NF, 3, ya, yb); slot(std::integral_constant<int, 7>{});
    if constexpr (do_ml) rd(slot_c, 3, ya, yb);
    TT_FENCE();
    if constexpr (do_ml) mm2(NF, 0, xa, xb); slot(std::integral_constant<int, 8>{});
    if constexpr (do_ml) mm2(NF, 1, xa, xb); slot(std::integral_constant<int, 9>{});
    if constexpr (do_ml) mm2(NF, 2, xa, xb); slot(std::integral_constant<int, 10>{});
    if constexpr (do_ml) mm2(NF, 3, xa, xb); slot(std::integral_constant<int, 11>{});
    if constexpr (do_ml) { if (more) rd(slot_next(), 0, xa, xb); }
    TT_FENCE();
    if constexpr (do_ml) mm2(NF, 0, ya, yb); slot(std::integral_constant<int, 12>{});
    if constexpr (do_ml) mm2(NF, 1, ya, yb); slot(std::integral_constant<int, 13>{});
    if constexpr (do_ml) mm2(NF, 2, ya, yb); slot(std::integral_constant<int, 14>{});
    if constexpr (do_ml) mm2(NF, 3, ya, yb); slot(std::integral_constant<int, 15>{});
    TT_FENCE();
    if constexpr (do_ml) { dma_advance(); slot_c = slot_next(); }
    (void)FIRST;
  };
  auto round = [&](auto ptag, auto mltag, auto eptag, int nwait0) __attribute__((always_inline)) {
    constexpr std::integral_constant<bool, true> TF{}; constexpr std::integral_constant<bool, false> NF{};
    constexpr bool do_ml = decltype(mltag)::value, do_ep = decltype(eptag)::value;
    if constexpr (do_ml) rd(slot_c, 0, xa, xb);
    stage(std::integral_constant<int, 0>{}, ptag, TF, mltag, eptag, true, nwait0);
    stage(std::integral_constant<int, 1>{}, ptag, NF, mltag, eptag, true, do_ep ? stores_of(0) : 0);
    stage(std::integral_constant<int, 2>{}, ptag, NF, mltag, eptag, true, do_ep ? stores_of(1) : 0);
    stage(std::integral_constant<int, 3>{}, ptag, NF, mltag, eptag, true, do_ep ? stores_of(2) : 0);
    stage(std::integral_constant<int, 4>{}, ptag, NF, mltag, eptag, true, do_ep ? stores_of(3) : 0);
    stage(std::integral_constant<int, 5>{}, ptag, NF, mltag, eptag, nk > 6, do_ep ? stores_of(4) : 0);
    if constexpr (do_ml) {
      for (int s = 6; s < nk; ++s) stage(std::integral_constant<int, -1>{}, ptag, NF, TF, NF, s + 1 < nk, (do_ep && s == 6) ? stores_of(5) : 0);
    }
  };
  // ---- prologue: stages 0 and 1 of the stream
  set_tile(0);
#pragma unroll
  for (int st = 0; st < NS - 1; ++st) {
#pragma unroll
    for (int i = 0; i < NPW; ++i) dma_piece(i, st);
    dma_advance();
  }
  p8_wait_vm<NPW>();
  asm volatile("s_barrier" ::: "memory");
  constexpr std::integral_constant<bool, true> YES{}; constexpr std::integral_constant<bool, false> NO{};
  constexpr std::integral_constant<int, 0> E0{}; constexpr std::integral_constant<int, 1> E1{};
  round(E0, YES, NO, 0);                                             // round 0: nothing to store yet
  // rounds 1 .. nt - 1: main loop + the epilogue of the previous tile, in one instruction stream.  Two rounds per loop iteration, so that both accumulator-set
  // assignments are straight-line code (an if / else on the round's parity made hipcc reconcile 256 accumulator registers at the join: ~60 spills per round).
  // nw: what the head of stage 0 may leave in flight = the stores of the previous round's last stage (only when that round had six stages and an epilogue)
  int r = 1;
  for (; r + 1 < nt; r += 2) {
    tile_mn(r - 1, ep_m0, ep_n0);
    round(E1, YES, YES, (r >= 2 && nk == 6) ? stores_of(5) : 0);
    tile_mn(r, ep_m0, ep_n0);
    round(E0, YES, YES, nk == 6 ? stores_of(5) : 0);
  }
  if (r < nt) { tile_mn(r - 1, ep_m0, ep_n0); round(E1, YES, YES, (r >= 2 && nk == 6) ? stores_of(5) : 0); }
  tile_mn(nt - 1, ep_m0, ep_n0);                                     // the last tile's epilogue alone
  if (nt & 1) round(E1, NO, YES, 0); else round(E0, NO, YES, 0);
#undef TT_FENCE
}
