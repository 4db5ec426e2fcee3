// 256 x 256 "phased" bf16 GEMM (C = A B^T + epilogue) for N % 256 == 0, K % 128 == 0, M % 256 == 0 -- the ATST-base geometry (d = 768: N = 768 /
// 2304 / 3072) and the N = 1536 launches of ATST-small.  Included by gemm.hip (inside its anonymous namespace, behind the epilogue helpers).
//
// Why a second main loop.  The 256 x 384 tile of gemm_nt_row384_kernel spends 2.3-2.4 k cycles per 32-deep k-tile for 1.5 k cycles of MFMA
// (DESIGN.md section 3): all eight waves leave one barrier together, all wait for their fragment reads together, all issue LDS-DMA
// together -- and its 192 accumulator registers leave no room to read fragments ahead.  Measured on the same box (round 5,
// tools/gemm_bench_base.py): the library GEMM reaches 1.2-1.4 PFLOP/s on these shapes where that loop reaches 0.95-1.1.  This kernel is the
// structure the CDNA4 guide calls the 8-phase template, restated for 32x32x16 MFMAs and this repo's LDS-DMA idiom:
//   * 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 = 8 accumulators of 32 x 32 (128 registers) -- which leaves 96 registers for TWO sets of
//     fragments per operand, so the reads of one phase are issued while the MFMAs of the previous one execute;
//   * BK = 64; a k-tile is four 16-KB UNITS (A rows of the lower / upper 64-row halves of both wave rows, B columns of the lower / upper 32-column
//     halves of all four wave columns); one unit is read per phase, one unit is staged per phase (2 LDS-DMA instructions per wave), four phases
//     per k-tile: phase q multiplies one 64 x 32 quadrant of the wave tile over the whole k-tile (8 MFMAs = 256 cycles);
//   * the two wave rows run STAGGERED by one barrier: while one row's waves are in their MFMA cluster, the other row's (their SIMD partners:
//     waves w and w + 4 share a SIMD) read fragments and issue LDS-DMA, then they swap -- the matrix pipe always has a cluster to run;
//   * every unit is staged SIX phases before it is read (two k-tile buffers = 8 unit slots, each slot re-staged two phases after its last
//     read), and the only vector-memory wait of the loop is a counted `s_waitcnt vmcnt(10)`: five units (80 KB per CU) stay in flight
//     across every barrier; a unit is read one phase after the wait that retires it.
// LDS image: rows of 128 B (64 bf16); 16-B chunk c of unit row u is stored at chunk c ^ ((u >> 1) & 7) -- the 16 lanes of a ds_read_b128
// service group then fall on 16 distinct bank slots (lane groups per MI355X_MICROARCH.md section LDS); the permutation is applied to the per-lane
// SOURCE address of the LDS-DMA (its destination is lane-linear) and again on the read.
// Reference math being accelerated: nn.Linear of audiossl/modules/transformer.py:87-90,109,119 and its input gradient.
namespace p8 {
constexpr int BM = 256, BNP = 256, BKP = 64, ROWB = 128, UNIT = 128 * ROWB, TILEB = 4 * UNIT, RING = 2 * TILEB;   // 16 KB / 64 KB / 128 KB
constexpr int THREADS = 512;
constexpr int PL1 = 144, CLD2 = 288;            // epilogue staging: two planes (low / high four columns of every 8-column slot) of a 288-float row
constexpr int RP = 32, SLOTS = RP * (BNP / 8) / THREADS;          // 32 staged rows per part, 2 slots of 8 columns per thread
constexpr int EPI_FLOATS = RP * CLD2 + CLD2 + BM + BNP;           // staging | bias (two planes) | per-row scale | column sums
static_assert(EPI_FLOATS * 4 <= RING, "the epilogue re-uses the operand ring");
// dGELU on e4m3 operands: the saved pre-activations u of the tile arrive by LDS-DMA, 16 KB per 32-row part, three buffers in the ring behind the
// staging area; part p + 2 is requested when part p starts.  (Measured alternatives, same A/B: parts 0 / 1 requested when the block starts, into
// buffers behind the ring: -33 us instead of -44 ; the whole tile requested the moment the main loop is done, 80 KB in flight: -4 us.  What the u
// read costs is its bytes next to everybody's operand fetches, not an exposed round trip: tools/epi_ablate.sh.)
constexpr int U_BUF = RP * BNP * 2, U_OFF2 = 40960;
static_assert(EPI_FLOATS * 4 <= U_OFF2 && U_OFF2 + 3 * U_BUF <= RING, "u buffers");
template <int EPI, bool F8> constexpr bool udma() { return EPI == EPI_DGELU && F8; }
constexpr int u_buf_off(int part) { return U_OFF2 + (part % 3) * U_BUF; }
}

#ifndef ATST_P8_ABL            // experiment builds (tools/p8_ablate.sh): 1 = no epilogue at all ; 2 = staging + read-back but no global loads / stores
#define ATST_P8_ABL 0
#endif
// F8: the operands are OCP e4m3 bytes, passed as byte PAIRS (K, lda, ldb halved by the launcher, as for gemm_nt_row384_kernel<.., F8>): the ring, the
// units, the swizzle and the LDS-DMA are byte-identical; a 128-byte row is two 64-byte k-steps of ONE v_mfma_scale_f32_32x32x64_f8f6f4 each (unit block
// scales), so a phase is 4 MFMAs over K = 128 -- twice the FLOPs of a bf16 phase in the same matrix-pipe time.  p.dq / dq_mul / dq_div undo the
// per-tensor scales when the accumulators are staged; the GELU / dGELU epilogues also write the e4m3 copy of their output and post its amax.
typedef int p8_v4i __attribute__((ext_vector_type(4)));
typedef int p8_v8i __attribute__((ext_vector_type(8)));
// SA (single A set): the e4m3 operands are 8-register tuples, and with two sets of A fragments hipcc could not place them (spills inside the loop).
// There the A fragments have ONE register set, read in the phase that first uses them (q0: A lower + B lower = 12 reads, q2: A upper, q3: none), the
// units are staged five phases ahead instead of six and four of them (64 KB) stay in flight behind `vmcnt(8)`.
template <int EPI, bool F8 = false, bool SA = false>
__global__ __launch_bounds__(512, 2) void gemm_nt_p8_kernel(GemmArgs p) {
  using namespace p8;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  typedef const void __attribute__((address_space(1))) * gptr_t;
  typedef void __attribute__((address_space(3))) * lptr_t;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3, hi = lane >> 5, l31 = lane & 31;
  const int ntn = p.N / BNP, ntm = p.M / BM;
  const int id = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (id / ntn) * BM, n0 = (id % ntn) * BNP;
  const int nk = p.K / BKP;                                       // even, >= 2 (checked by the launcher)
#ifndef ATST_P8_UDMA
#define ATST_P8_UDMA 1
#endif
  constexpr bool UDMA_ON = ATST_P8_UDMA && !(ATST_EPI_ABL & 4) && !(ATST_P8_ABL & 3);
  auto tile_row_of0 = [&](int part, int rl) { return (rl >> 4) * 128 + (part >> 1) * 32 + (part & 1) * 16 + (rl & 15); };
  auto u_dma = [&](int part, int lane) {                          // (dGELU, e4m3 operands) part `part` of u -> its LDS buffer: 2 x 1 KB per wave
    const void* ubase = sgpr_ptr(p.U + (size_t)m0 * p.ldc + n0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kb = wid + 8 * j, rl = kb * 2 + (lane >> 5);
      p8_glds16((unsigned)(tile_row_of0(part, rl) * p.ldc + (lane & 31) * 8) * 2u, ubase, lds_addr(smem_raw) + u_buf_off(part) + kb * 1024);
    }
  };
  if (p.skew > 0 && blockIdx.x < 256 && ((blockIdx.x >> 3) & 1)) {   // experiment (hook 1000 + c): every other block of the first round starts c x 1024 cycles late
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (long long)p.skew * 1024) __builtin_amdgcn_s_sleep(8);
  }

  // ---- LDS-DMA sources.  Instruction ii = 2 wid + j of a unit covers unit rows 8 ii .. 8 ii + 7 (8 lanes per 128-B row).
  //   A units: unit row u <-> tile row (u >> 6) * 128 + (u & 63) (+ 64 for the upper-half unit)   B units: u <-> column (u >> 5) * 64 + (u & 31) (+ 32)
  // Address = wave-uniform base (SGPRs: tile origin + k-tile + unit half) + a 32-bit per-lane byte offset: four offset registers instead of
  // four 64-bit pointers -- with 224 registers of accumulators and fragments live, every register of addressing counts.
  unsigned voA[2], voB[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int u = (2 * wid + j) * 8 + (lane >> 3), key = (u >> 1) & 7, ch = (lane & 7) ^ key;
    voA[j] = (unsigned)(((u >> 6) * 128 + (u & 63)) * p.lda + ch * 8) * 2u;
    voB[j] = (unsigned)(((u >> 5) * 64 + (u & 31)) * p.ldb + ch * 8) * 2u;
  }
  const char* baseA = sgpr_ptr(p.A + (size_t)m0 * p.lda);
  const char* baseB = sgpr_ptr(p.B + (size_t)n0 * p.ldb);
  const size_t a_half = (size_t)64 * p.lda * 2, b_half = (size_t)32 * p.ldb * 2;
  char* lds = smem_raw;
  const unsigned lds0 = lds_addr(smem_raw) + (2 * wid) * 1024;    // this wave's two pieces of a unit (LDS byte address)
  // unit slots of a k-tile buffer: 0 = A lower rows, 1 = A upper rows, 2 = B lower columns, 3 = B upper columns
  auto stage = [&](int kt, int unit) {                            // kt: k-tile, unit: 0..3
    if constexpr (F8) kt = kt < nk ? kt : nk - 2 + (kt & 1);      // (see the loop: no separate last pair)
    const unsigned dst = lds0 + (kt & 1) * TILEB + unit * UNIT;
    const char* base = (unit < 2 ? baseA + (unit & 1) * a_half : baseB + (unit & 1) * b_half) + (size_t)kt * (BKP * 2);
#pragma unroll
    for (int j = 0; j < 2; ++j) p8_glds16(unit < 2 ? voA[j] : voB[j], base, dst + j * 1024);
  };

  // ---- fragment addresses: row (32 rb + l31) of the wave's 64 / 32 unit rows, chunk (2 ks + hi) ^ key, key = (l31 >> 1) & 7
  // The four k-step addresses of a row differ by an XOR: (row base | c0) ^ (ks << 5), c0 = ((hi ^ key) << 4) -- row bases are multiples of 128 B.
  // Only the two ks = 0 addresses live across the loop; the other three are re-derived (one v_xor each) from a laundered copy at every read,
  // or hipcc hoists all eight out of the loop and spills one of them (its reload is a `s_waitcnt vmcnt(0)` in front of the last k-tiles).
  const int xr = (l31 >> 1) & 7;
  // bf16: chunk 2 ks + hi of k-step ks ; e4m3: chunks 4 kk + 2 hi + e (e = 0, 1) of k-step kk.  jx(i): what the i-th 16-B read of a row XORs into the address.
  const int c0 = F8 ? (((2 * hi) ^ xr) << 4) : ((hi ^ xr) << 4);
  auto jx = [](int i) { return F8 ? (((i >> 1) << 6) | ((i & 1) << 4)) : (i << 5); };
  const int baseA0 = (wr * 64 + l31) * ROWB + c0, baseB0 = 2 * UNIT + (wc * 32 + l31) * ROWB + c0;
  p8_v4i fa[SA ? 1 : 2][2][F8 ? 1 : 4], fb[2][F8 ? 1 : 4];                 // bf16: [half mh][row block rb][k-step] ; [half nh][k-step]   (16 B = one MFMA operand)
  p8_v8i fa8[SA ? 1 : 2][2][F8 ? 2 : 1], fb8[2][F8 ? 2 : 1];               // e4m3: one 32-B operand per k-step, its halves read straight into the tuple
  auto read_a = [&](int buf, int mh) {
    int b0 = baseA0;
    asm volatile("" : "+v"(b0));
    const char* s = lds + buf * TILEB + mh * UNIT;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const p8_v4i v = *reinterpret_cast<const p8_v4i*>(s + (b0 ^ jx(i)) + rb * 32 * ROWB);
        if constexpr (F8) { if (i & 1) fa8[SA ? 0 : mh][rb][i >> 1].hi = v; else fa8[SA ? 0 : mh][rb][i >> 1].lo = v; }
        else fa[SA ? 0 : mh][rb][F8 ? 0 : i] = v;
      }
  };
  auto read_b = [&](int buf, int nh) {
    int b0 = baseB0;
    asm volatile("" : "+v"(b0));
    const char* s = lds + buf * TILEB + nh * UNIT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const p8_v4i v = *reinterpret_cast<const p8_v4i*>(s + (b0 ^ jx(i)));
      if constexpr (F8) { if (i & 1) fb8[nh][i >> 1].hi = v; else fb8[nh][i >> 1].lo = v; }
      else fb[nh][F8 ? 0 : i] = v;
    }
  };
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  auto quadrant = [&](int mh, int nh) {                           // 64 x 32 of the wave tile over the whole k-tile: 8 (bf16) / 4 (e4m3) MFMAs, 256 cycles
    __builtin_amdgcn_s_setprio(1);
    if constexpr (F8) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
          acc[mh * 2 + rb][nh] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa8[SA ? 0 : mh][rb][F8 ? kk : 0], fb8[nh][F8 ? kk : 0], acc[mh * 2 + rb][nh], 0, 0, 0,
                                                                                 0x7F7F7F7F, 0, 0x7F7F7F7F);
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
          acc[mh * 2 + rb][nh] = mfma32(__builtin_bit_cast(bf16x8, fa[SA ? 0 : mh][rb][F8 ? 0 : ks]), __builtin_bit_cast(bf16x8, fb[nh][F8 ? 0 : ks]), acc[mh * 2 + rb][nh]);
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- schedule.  Absolute phase f = 4 t + q of k-tile t:
  //   reads   q0: B lower(t)   q1: B upper(t)   q2: A upper(t)   q3: A lower(t + 1)
  //   MFMAs   q0: (lo, lo)     q1: (lo, up)     q2: (up, up)     q3: (up, lo)
  //   stages  q0: A upper(t+1) q1: A lower(t+2) q2: B lower(t+2) q3: B upper(t+2)       (each slot two phases after its last read, six before its next)
  // Prologue = phases -7 .. -1 of that schedule without MFMAs: seven units.
  if constexpr (!SA) {
    stage(0, 0); stage(0, 2); stage(0, 3); stage(0, 1); stage(1, 0); stage(1, 2); stage(1, 3);
    p8_wait_vm<10>();                                             // the first two units (A lower(0), B lower(0)) have landed -- mine
    asm volatile("s_barrier" ::: "memory");                       // ... everyone's
    read_a(0, 0);
  } else {                                                        // SA schedule: phases -6 .. -1 (see below)
    stage(0, 0); stage(0, 2); stage(0, 3); stage(0, 1); stage(1, 0); stage(1, 2);
    p8_wait_vm<8>();
    asm volatile("s_barrier" ::: "memory");
  }
  if (wr == 1) asm volatile("s_barrier" ::: "memory");            // the upper wave row runs one barrier behind the lower one

  // One phase of the pair (t, t + 1) of k-tiles.  LAST: this is the last pair -- nothing beyond k-tile nk - 1 is staged or read, and the waits count
  // down with the units still in flight (VML = instructions allowed in flight behind this phase's wait in the last pair, -1 = nothing outstanding;
  // 10 (SA: 8) everywhere else: two per unit staged in the last five (four) phases).  The last pair is a second, straight-line copy of the body.
  auto phase = [&](int t, auto ltag, auto qtag, auto btag, auto ftag, auto vtag) {
    constexpr int q = decltype(qtag)::value, buf = decltype(btag)::value, VML = decltype(vtag)::value;   // buf = tile & 1
    constexpr bool FIRST = decltype(ftag)::value;                 // first k-tile of the pair
    constexpr bool last = decltype(ltag)::value;
    // L segment: fragment reads of this phase's unit, LDS-DMA of the unit six (SA: five) phases ahead, counted wait, barrier
    if constexpr (!SA) {
      //   reads   q0: B lower(t)   q1: B upper(t)   q2: A upper(t)   q3: A lower(t + 1)
      //   stages  q0: A upper(t+1) q1: A lower(t+2) q2: B lower(t+2) q3: B upper(t+2)     (each slot two phases after its last read, six before its next)
      if constexpr (q == 0) read_b(buf, 0);
      if constexpr (q == 1) read_b(buf, 1);
      if constexpr (q == 2) read_a(buf, 1);
      if constexpr (q == 3) { if (FIRST || !last) read_a(buf ^ 1, 0); }   // (F8, last k-tile: reads the other buffer's lower A rows, unused)
      if ((FIRST && q == 0) || !last) {
        if constexpr (q == 0) stage(t + 1, 1);
        if constexpr (q == 1) stage(t + 2, 0);
        if constexpr (q == 2) stage(t + 2, 2);
        if constexpr (q == 3) stage(t + 2, 3);
      }
      if (!last) p8_wait_vm<10>();
      else if constexpr (VML >= 0) p8_wait_vm<(VML >= 0 ? VML : 0)>();
    } else {
      //   reads   q0: A lower(t) + B lower(t)   q1: B upper(t)   q2: A upper(t)   q3: --
      //   stages  q0: B upper(t+1)   q1: A upper(t+1)   q2: A lower(t+2)   q3: B lower(t+2)     (two / three phases after the slot's last read, five before its next)
      if constexpr (q == 0) { read_b(buf, 0); read_a(buf, 0); }
      if constexpr (q == 1) read_b(buf, 1);
      if constexpr (q == 2) read_a(buf, 1);
      if ((FIRST && q <= 1) || !last) {
        if constexpr (q == 0) stage(t + 1, 3);
        if constexpr (q == 1) stage(t + 1, 1);
        if constexpr (q == 2) stage(t + 2, 0);
        if constexpr (q == 3) stage(t + 2, 2);
      }
      constexpr int VS = (FIRST && q == 0 && VML == 10) ? 8 : VML;      // last pair: 8 8 6 4 | 2 0 - -
      if (!last) p8_wait_vm<8>();
      else if constexpr (VS >= 0) p8_wait_vm<(VS >= 0 ? VS : 0)>();
    }
    asm volatile("s_barrier" ::: "memory");
    // M segment
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    quadrant(q >= 2 ? 1 : 0, (q == 1 || q == 2) ? 1 : 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
  };
#define P8_I(x) std::integral_constant<int, x>{}
#define P8_B(x) std::integral_constant<bool, x>{}
  int t = 0;
#define P8_PAIR(L)                                                          \
    phase(t, P8_B(L), P8_I(0), P8_I(0), P8_B(true), P8_I(10));              \
    phase(t, P8_B(L), P8_I(1), P8_I(0), P8_B(true), P8_I(8));               \
    phase(t, P8_B(L), P8_I(2), P8_I(0), P8_B(true), P8_I(6));               \
    phase(t, P8_B(L), P8_I(3), P8_I(0), P8_B(true), P8_I(4));               \
    phase(t + 1, P8_B(L), P8_I(0), P8_I(1), P8_B(false), P8_I(2));          \
    phase(t + 1, P8_B(L), P8_I(1), P8_I(1), P8_B(false), P8_I(0));          \
    phase(t + 1, P8_B(L), P8_I(2), P8_I(1), P8_B(false), P8_I(-1));         \
    phase(t + 1, P8_B(L), P8_I(3), P8_I(1), P8_B(false), P8_I(-1));
  if constexpr (F8) {
    // e4m3: K = 768 is six k-tiles, a third of them in the last pair -- whose straight-line copy hipcc register-allocates badly (8-register operand
    // tuples: freshly read fragments spilled).  Every pair runs the steady body instead; stage() clamps a target beyond the last k-tile to the last
    // k-tile of the same parity, i.e. re-writes a slot with the bytes it already holds (harmless to concurrent readers; 7 units of extra L2 -> LDS
    // traffic per tile), and the waits stay `vmcnt(10)`.
    for (; t < nk; t += 2) { P8_PAIR(false) }
    p8_wait_vm<0>();
  } else {
    for (; t + 2 < nk; t += 2) { P8_PAIR(false) }
    P8_PAIR(true)
  }
#undef P8_PAIR
#undef P8_I
#undef P8_B
  if (wr == 0) asm volatile("s_barrier" ::: "memory");            // re-align the two wave rows: every operand read is done, the ring is free

  // ---- epilogue: the fp32 tile goes through LDS 32 rows at a time (16 rows of accumulator block mb from both wave rows), so that every global
  // access is a 16-B piece of a 256-column row segment; all global loads of a part are issued before the staging barrier, all stores after it
  // (see epi_fetch8 / epilogue8).
  // the epilogue's index math starts from a laundered copy of the thread id: computed from `tid` it is hoisted above the main loop, where there
  // is not a register to spare
  int tid_e = threadIdx.x;
  asm volatile("" : "+v"(tid_e));
#if ATST_P8_ABL & 1
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) asm volatile("" :: "v"(acc[i][j]));
  return;
#endif
  {
  const int tid = tid_e, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
  float* sC = reinterpret_cast<float*>(smem_raw);
  const float dqv = F8 ? (p.dq ? *p.dq : 1.0f) * (p.dq_mul != 0.f ? p.dq_mul : 1.0f) / (p.dq_div ? *p.dq_div : 1.0f) : 1.0f;
  auto tile_row_of = [&](int part, int rl) { return (rl >> 4) * 128 + (part >> 1) * 32 + (part & 1) * 16 + (rl & 15); };
  if constexpr (EPI == EPI_RESID || EPI == EPI_F32) {
    // fp32 tensors: ONE 16-B access per lane with consecutive lanes on consecutive 16-B pieces -- a wave instruction covers one whole 1-KB row segment
    // (8 full lines).  The 8-column slots of epilogue8 are right for bf16 outputs (16 B = 8 columns) but make two strided half-line accesses of an
    // fp32 row (16 lines touched per instruction), and a CU's epilogue rate is set by the lines its vector-memory instructions touch.
    constexpr int CLD4 = 260, ITEMS = RP * (BNP / 4) / THREADS;    // plain staging rows ; 4 items of 4 columns per thread per part
    float* sB4 = sC + RP * CLD4; float* sS4 = sB4 + BNP;
    if (wid < 4) { if (p.bias) lds_fill64(p.bias + n0 + wid * 64 + lane, sB4 + wid * 64); else sB4[wid * 64 + lane] = 0.f; }
    if constexpr (EPI == EPI_RESID) {
      if (tid < BM) sS4[tid] = p.row_scale ? p.row_scale[(m0 + tid) / p.rows_per_seq] : 1.0f;
    }
    constexpr int PFR = EPI == EPI_RESID ? 2 : 0;                 // residual rows of part + 2 are requested while part is processed
    f32x4 rres[PFR + 1][ITEMS];
    auto fetch4 = [&](int part, f32x4 (&r)[ITEMS]) {
      if constexpr (EPI == EPI_RESID) {
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
          const int idx = tid + THREADS * k;
#if !(ATST_P8_ABL & 2)
          const size_t ri = (size_t)(m0 + tile_row_of(part, idx >> 6)) * p.ldc + n0 + (idx & 63) * 4;
          if (p.resid_bf16) {                                     // bf16 residual stream (fp8 inference passes): 8 B per lane, a wave instruction = one 512-B row segment
            const bf16x4 rb = ld_pol<3>(reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(p.resid) + ri));
            r[k] = f32x4{bf2f(rb[0]), bf2f(rb[1]), bf2f(rb[2]), bf2f(rb[3])};
          } else {
            r[k] = ld_pol<3>(reinterpret_cast<const f32x4*>(p.resid + ri));
          }
#endif
        }
      }
    };
#pragma unroll
    for (int pp = 0; pp < PFR; ++pp) fetch4(pp, rres[pp]);
#pragma unroll
    for (int part = 0; part < 8; ++part) {
      const int mb = part >> 1, h = part & 1;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8)
          sC[(wr * 16 + (r8 & 3) + 8 * (r8 >> 2) + 4 * hi) * CLD4 + wc * 64 + nb * 32 + l31] = F8 ? acc[mb][nb][h * 8 + r8] * dqv : acc[mb][nb][h * 8 + r8];
      if (part + PFR < 8) fetch4(part + PFR, rres[(part + PFR) % (PFR + 1)]);
      if (part == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the bias table (LDS-DMA) has landed -- mine; the barrier makes it everyone's
      lds_barrier();
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) {
        const int idx = tid + THREADS * k, rl = idx >> 6, c4 = (idx & 63) * 4;
        const int trow = tile_row_of(part, rl);
        const f32x4 v = *reinterpret_cast<const f32x4*>(sC + rl * CLD4 + c4) + *reinterpret_cast<const f32x4*>(sB4 + c4);
#if ATST_P8_ABL & 2
        asm volatile("" :: "v"(v)); continue;
#endif
        f32x4 o = v;
        if constexpr (EPI == EPI_RESID) o = rres[part % (PFR + 1)][k] + sS4[trow] * v;
        if (EPI == EPI_RESID && p.out_bf16) {
          bf16x4 ob; ob[0] = f2bf(o[0]); ob[1] = f2bf(o[1]); ob[2] = f2bf(o[2]); ob[3] = f2bf(o[3]);
          st_pol<4>(ob, reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p.C) + (size_t)(m0 + trow) * p.ldc + n0 + c4));
        } else
        st_pol<1>(o, reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (size_t)(m0 + trow) * p.ldc + n0 + c4));
      }
      if (part < 7) lds_barrier();
    }
    return;
  }
  float* sBias = sC + RP * CLD2;
  // e4m3 copies of the output (fp8 forward: GELU activation ; fp8 dgrad: du): scale read ONCE (a load inside a part's stores waits for them), running max |x|
  float q8s = 1.0f, omax = 0.f;
  if constexpr (EPI == EPI_DGELU) { if (p.q8 && p.q8_scale_ptr) q8s = *p.q8_scale_ptr; }
  if constexpr (EPI == EPI_BIAS_GELU) { if (p.q8) q8s = p.q8_scale_ptr ? *p.q8_scale_ptr : p.q8_scale; }
  if (wid < 4) {                                                  // bias in the two-plane layout of the tile, by LDS-DMA (gather on the source side)
    const int pos = (wid & 1) * 64 + lane, plane = wid >> 1;      // position 4 g + e of plane h <- column 8 g + 4 h + e
    float* dst = sBias + plane * PL1 + (wid & 1) * 64;
    if (p.bias) lds_fill64(p.bias + n0 + 8 * (pos >> 2) + 4 * plane + (pos & 3), dst); else dst[lane] = 0.f;
  }
  f32x4 csr[EPI == EPI_DGELU ? SLOTS : 1][2];
#pragma unroll
  for (int i = 0; i < (EPI == EPI_DGELU ? SLOTS : 1); ++i) { csr[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; csr[i][1] = csr[i][0]; }
  // Epilogues that READ (residual stream ; saved pre-activation): the loads of part + PF are issued while part is processed -- (PF + 1) x 64 KB in
  // flight per CU instead of one part's.  A CU's epilogue rate is bytes in flight / memory latency; with one part in flight it is ~22 GB/s, which
  // only equals a fair share of HBM when all 256 CUs run their epilogues at the same moment (profiles/r05_p8_ablate.txt).  The fragment registers of
  // the main loop are free here, so the deeper queue costs nothing.
  // dGELU, e4m3 operands: u arrives by LDS-DMA (UDMA, see p8::udma): no register holds a load in flight, so this instantiation -- which had no
  // registers for a prefetch -- is two parts ahead like the bf16 one (638 -> 594 us at M = 131072, same box).  A wave's 1-KB instructions are exactly
  // the pieces its own threads read back (slot idx = 16-B piece idx), so its own counted vmcnt is the only synchronisation.  (The bf16 instantiation
  // keeps its register prefetch: 770 vs 786 us with the DMA form.)
  constexpr bool UDMA = udma<EPI, F8>() && UDMA_ON;
  if constexpr (UDMA) { u_dma(0, lane); u_dma(1, lane); }
  constexpr int PF = (EPI == EPI_DGELU && !F8 && !UDMA) ? 2 : 0;  // register prefetch (the e4m3 instantiation has no registers for it: spills)
  EpiAux aux[PF + 1][SLOTS];
  auto fetch_part = [&](int part, EpiAux (&a)[SLOTS]) {
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const int idx = tid + THREADS * i;
#if !(ATST_P8_ABL & 2)
      epi_fetch8<EPI, false>(p, m0 + tile_row_of(part, idx >> 5), n0 + (idx & 31) * 8, a[i]);
#endif
    }
  };
#pragma unroll
  for (int pp = 0; pp < PF; ++pp) fetch_part(pp, aux[pp]);
#pragma unroll
  for (int part = 0; part < 8; ++part) {
    const int mb = part >> 1, h = part & 1;
    if constexpr (UDMA) { if (part + 2 < 8) u_dma(part + 2, lane); }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r8 = 0; r8 < 8; ++r8) {
        const int lrow = wr * 16 + (r8 & 3) + 8 * (r8 >> 2) + 4 * hi, ccol = wc * 64 + nb * 32 + l31;
        sC[lrow * CLD2 + ((ccol >> 3) << 2) + (ccol & 3) + ((ccol & 4) ? PL1 : 0)] = F8 ? acc[mb][nb][h * 8 + r8] * dqv : acc[mb][nb][h * 8 + r8];
      }
    if constexpr (!UDMA) { if (part + PF < 8) fetch_part(part + PF, aux[(part + PF) % (PF + 1)]); }
    if (part == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the bias table (LDS-DMA) has landed -- mine; the barrier makes it everyone's
    lds_barrier();
    if constexpr (UDMA) {
      // u of this part has landed (part 0: the vmcnt(0) above).  Behind its 2 DMA instructions this wave has issued the DMAs of the parts requested since
      // (2 each) and the stores of the parts processed since -- SLOTS per part (du or its e4m3 copy; both: more stores, the wait is only stricter):
      // part 1: 2 * 2 + 2 ; parts 2..5: 2 * 2 + 2 * 2 ; part 6: 2 + 4 ; part 7: 4
      static_assert(SLOTS == 2, "the counts below");
      if (part == 1 || part == 6) p8_wait_vm<6>(); else if (part == 7) p8_wait_vm<4>(); else if (part >= 2) p8_wait_vm<8>();
    }
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const int idx = tid + THREADS * i, rl = idx >> 5, c8 = (idx & 31) * 8;
      const int trow = tile_row_of(part, rl), row = m0 + trow;
      f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = w0;
      EpiAux& ax = aux[part % (PF + 1)][i];
      if constexpr (UDMA) ax.a0 = *reinterpret_cast<const f32x4*>(smem_raw + u_buf_off(part) + idx * 16);
      if constexpr (EPI == EPI_DGELU || EPI == EPI_BIAS_GELU) ax.s = q8s;
#if ATST_P8_ABL & 2
      { f32x4 z0 = *reinterpret_cast<const f32x4*>(sC + rl * CLD2 + (c8 >> 1)), z1 = *reinterpret_cast<const f32x4*>(sC + rl * CLD2 + PL1 + (c8 >> 1));
        asm volatile("" :: "v"(z0), "v"(z1)); (void)row; continue; }
#endif
      epilogue8<EPI>(p, row, n0 + c8, *reinterpret_cast<const f32x4*>(sC + rl * CLD2 + (c8 >> 1)), *reinterpret_cast<const f32x4*>(sC + rl * CLD2 + PL1 + (c8 >> 1)),
                     *reinterpret_cast<const f32x4*>(sBias + (c8 >> 1)), *reinterpret_cast<const f32x4*>(sBias + PL1 + (c8 >> 1)), ax, w0, w1);
      if constexpr (EPI == EPI_DGELU || EPI == EPI_BIAS_GELU) {
        if (p.q8_amax) {
#pragma unroll
          for (int e = 0; e < 4; ++e) omax = fmaxf(omax, fmaxf(fabsf(w0[e]), fabsf(w1[e])));
        }
      }
      if constexpr (EPI == EPI_DGELU) { csr[i][0] += w0; csr[i][1] += w1; }
    }
    if (part < 7) lds_barrier();
  }
  if constexpr (EPI == EPI_DGELU) {
    if (p.colsum) {                                               // fold the 32 row slots of every column once, after the last part
      lds_barrier();
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) {
        const int idx = tid + THREADS * i, rl = idx >> 5, c8 = (idx & 31) * 8;
        *reinterpret_cast<f32x4*>(sC + rl * CLD2 + (c8 >> 1)) = csr[i][0];
        *reinterpret_cast<f32x4*>(sC + rl * CLD2 + PL1 + (c8 >> 1)) = csr[i][1];
      }
      lds_barrier();
      if (tid < BNP) {
        const int ph = ((tid >> 3) << 2) + (tid & 3) + ((tid & 4) ? PL1 : 0);
        float t = 0.f;
#pragma unroll 8
        for (int r = 0; r < RP; ++r) t += sC[r * CLD2 + ph];
        atomicAdd(p.colsum + n0 + tid, t);
      }
    }
    if (p.q8_amax) amax_post(p.q8_amax, wave_max(omax), lane, blockIdx.x * 8 + wid);
  }
  if constexpr (EPI == EPI_BIAS_GELU) {
    if (p.q8 && p.q8_amax) amax_post(p.q8_amax, wave_max(omax), lane, blockIdx.x * 8 + wid);
  }
  }
}
