// Whole-encoder forward / backward driver: one host call enqueues every kernel of AST.forward / FrameAST.forward
// (audiossl/models/atst/audio_transformer.py:188-221, audiossl/methods/atstframe/audio_transformer.py:183-207,
// 12 x Block.forward audiossl/modules/transformer.py:136-150) or of its backward on a HIP stream.  No Python, no
// autograd graph, no per-op dispatch: the activation tape is a fixed carve of one caller-owned workspace.
#include "common.h"
#include "kernels.h"
#include "../../include/atst_hip.h"

extern int g_f8_resid16;          // gemm.hip: tuning hook 2100 / 2101
extern int g_f8_fuse_ln;          // gemm.hip: tuning hook 2110 / 2111
namespace {

struct LayerWs {
  bf16 *h1, *qkv, *o, *h2, *u, *a;
  float *mean1, *rstd1, *mean2, *rstd2, *lse;
  uint8_t *h18, *o8, *h28, *a8;       // fp8 training: the e4m3 copies of the qkv / proj / fc1 / fc2 inputs are KEPT per layer (operands of the e4m3 weight gradients)
};
struct Ws {
  bf16* patches; float* table;
  float* x[2 * ATST_MAX_DEPTH + 1];
  LayerWs L[ATST_MAX_DEPTH];
  bf16* hN; float *meanN, *rstdN;
  uint8_t *q8a, *q8b;                 // fp8 forward: e4m3 copies of the current GEMM's A operand ([M,C] and [M,4C])
  // backward scratch
  float *dxA, *dxB;
  bf16 *g, *g2, *dh, *du, *dqkv, *d_o, *dout;
  uint8_t *g8, *g28, *du8, *dqkv8;    // fp8 dgrad: e4m3 copies of g, g2, du ; dqkv8: what the NP = 256 attention backward writes INSTEAD of dqkv
  float* dscr;
  size_t bytes;
};

struct Carver {
  char* base; size_t off;
  template <typename T> T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = reinterpret_cast<T*>(base + off);
    off += n * sizeof(T);
    return p;
  }
};

Ws carve(void* ws, int S, int NP, int C, int H, int depth, int train, int fp8, int PK = 256) {
  Ws w{};
  Carver c{reinterpret_cast<char*>(ws), 0};
  const size_t M = (size_t)S * NP;
  w.patches = c.take<bf16>(M * PK);                                // PK = patch_h * patch_w (256 for the shipped 64 x 4 patches)
  w.table = c.take<float>((size_t)NP * C);
  const int nx = train ? 2 * depth + 1 : 1;
  for (int i = 0; i < nx; ++i) w.x[i] = c.take<float>(M * C);
  for (int i = nx; i < 2 * depth + 1; ++i) w.x[i] = w.x[0];
  const int nl = train ? depth : 1;
  for (int i = 0; i < nl; ++i) {
    LayerWs& l = w.L[i];
    l.h1 = c.take<bf16>(M * C); l.qkv = c.take<bf16>(M * 3 * C); l.o = c.take<bf16>(M * C);
    l.h2 = c.take<bf16>(M * C); l.u = c.take<bf16>(M * 4 * C); l.a = c.take<bf16>(M * 4 * C);
    l.mean1 = c.take<float>(M); l.rstd1 = c.take<float>(M); l.mean2 = c.take<float>(M); l.rstd2 = c.take<float>(M);
    l.lse = c.take<float>((size_t)S * H * NP);
    if (fp8 && train) { l.h18 = c.take<uint8_t>(M * C); l.o8 = c.take<uint8_t>(M * C); l.h28 = c.take<uint8_t>(M * C); l.a8 = c.take<uint8_t>(M * 4 * C); }
  }
  for (int i = nl; i < depth; ++i) w.L[i] = w.L[0];
  w.hN = c.take<bf16>(M * C); w.meanN = c.take<float>(M); w.rstdN = c.take<float>(M);
  if (fp8 && !train) {                                             // inference: one transient pair of e4m3 operand copies shared by all layers
    w.q8a = c.take<uint8_t>(M * C); w.q8b = c.take<uint8_t>(M * 4 * C);
    for (int i = 0; i < depth; ++i) { w.L[i].h18 = w.L[i].o8 = w.L[i].h28 = w.q8a; w.L[i].a8 = w.q8b; }
  }
  if (train) {
    w.dxA = c.take<float>(M * C); w.dxB = c.take<float>(M * C);
    w.g = c.take<bf16>(M * C); w.g2 = c.take<bf16>(M * C); w.dh = c.take<bf16>(M * C); w.du = c.take<bf16>(M * 4 * C);
    w.dqkv = c.take<bf16>(M * 3 * C); w.d_o = c.take<bf16>(M * C); w.dout = c.take<bf16>(M * C);
    w.dscr = c.take<float>((size_t)S * H * NP);
    if (fp8) { w.g8 = c.take<uint8_t>(M * C); w.g28 = c.take<uint8_t>(M * C); w.du8 = c.take<uint8_t>(M * 4 * C); w.dqkv8 = c.take<uint8_t>(M * 3 * C); }
  }
  w.bytes = (c.off + 255) & ~(size_t)255;
  return w;
}

inline const bf16* B16(const uint16_t* p) { return reinterpret_cast<const bf16*>(p); }

#define RUN(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)

int gemm(const bf16* A, const bf16* B, int M, int N, int K, int epi, void* C, hipStream_t st, const float* bias = nullptr,
         const float* resid = nullptr, const float* row_scale = nullptr, int rps = 1, void* C2 = nullptr,
         const bf16* U = nullptr, float* colsum = nullptr, const float* ln_g = nullptr, const float* ln_b = nullptr,
         bf16* ln_out = nullptr, float* ln_mean = nullptr, float* ln_rstd = nullptr) {
  GemmArgs a{};
  a.A = A; a.B = B; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldb = K; a.epi = epi; a.C = C; a.ldc = N; a.C2 = C2;
  a.bias = bias; a.resid = resid; a.row_scale = row_scale; a.rows_per_seq = rps; a.U = U; a.colsum = colsum;
  a.ln_gamma = ln_g; a.ln_beta = ln_b; a.ln_out = ln_out; a.ln_mean = ln_mean; a.ln_rstd = ln_rstd;
  return atst_gemm_nt(a, st);
}
// e4m3 forward GEMM: A8 [M,K] (activation copy, scale act_scale), B8 [N,K] (weight shadow, per-tensor scale in *w_dq)
constexpr float ACT_SCALE = 8.0f, ACT_SCALE_GELU = 4.0f;   // activation scales when the caller passes no running ones (f8_act_scale == NULL): +-56 / +-112 fit
int gemm8(const uint8_t* A8, const uint8_t* B8, int M, int N, int K, int epi, void* C, hipStream_t st, const float* w_dq, float act_scale,
          const float* bias = nullptr, const float* resid = nullptr, const float* row_scale = nullptr, int rps = 1, void* C2 = nullptr,
          uint8_t* q8 = nullptr, float q8_scale = 1.0f, unsigned* q8_sat = nullptr, const float* act_scale_dev = nullptr,
          const float* q8_scale_dev = nullptr, float* q8_amax = nullptr, int resid_bf16 = 0, int out_bf16 = 0,
          const float* ln_g = nullptr, const float* ln_b = nullptr, bf16* ln_out = nullptr, float* ln_mean = nullptr, float* ln_rstd = nullptr) {
  GemmArgs a{};
  a.resid_bf16 = resid_bf16; a.out_bf16 = out_bf16;
  a.ln_gamma = ln_g; a.ln_beta = ln_b; a.ln_out = ln_out; a.ln_mean = ln_mean; a.ln_rstd = ln_rstd;   // EPI_RESID, N == 384: + LayerNorm of the new row (q8: its e4m3 copy)
  a.A = reinterpret_cast<const bf16*>(A8); a.B = reinterpret_cast<const bf16*>(B8); a.M = M; a.N = N; a.K = K; a.lda = K; a.ldb = K;
  a.epi = epi; a.C = C; a.ldc = N; a.C2 = C2; a.bias = bias; a.resid = resid; a.row_scale = row_scale; a.rows_per_seq = rps;
  a.fp8 = 1; a.dq = w_dq; a.dq_mul = 1.0f / act_scale; a.q8 = q8; a.q8_scale = q8_scale; a.q8_sat = q8_sat;
  if (act_scale_dev) { a.dq_mul = 1.0f; a.dq_div = act_scale_dev; }   // running activation scale of the A operand (device scalar)
  a.q8_scale_ptr = q8_scale_dev; a.q8_amax = q8_amax;                // scale / amax of the e4m3 copy this epilogue writes (GELU output)
  return atst_gemm_nt(a, st);
}
// dgrad GEMM (N == 384) whose epilogue is the backward of the LayerNorm in front of the differentiated Linear:
// dx = dres + LN'(dY W) ; g = bf16(row_scale dx) ; dgamma, dbeta, dbias_up += column sums (EPI_LNBWD, gemm.hip)
int gemm_lnbwd(const bf16* dY, const bf16* Wt, int M, int K, const float* x, const float* mean, const float* rstd, const float* gamma,
               const float* dres, float* dx, bf16* g, const float* row_scale, int rps, float* dgamma, float* dbeta, float* dbias_up,
               hipStream_t st) {
  GemmArgs a{};
  a.A = dY; a.B = Wt; a.M = M; a.N = 384; a.K = K; a.lda = K; a.ldb = K; a.epi = EPI_LNBWD; a.C = dx; a.ldc = 384;
  a.resid = dres; a.row_scale = row_scale; a.rows_per_seq = rps; a.ln_gamma = gamma; a.ln_mean = const_cast<float*>(mean);
  a.ln_rstd = const_cast<float*>(rstd); a.lnb_x = x; a.lnb_g = g; a.lnb_dgamma = dgamma; a.lnb_dbeta = dbeta; a.lnb_dbias_up = dbias_up;
  return atst_gemm_nt(a, st);
}
// the same with e4m3 bookkeeping (round 6, d = 384): dY / Wt are e4m3 bytes when `w_dq` is given (per-tensor factor *w_dq, gradient scale *a_scale),
// and the epilogue also writes g8 = e4m3(bf16(row_scale dx) * *g8_scale) and posts max |row_scale dx| to g8_amax -- what ln_bwd_kernel does in the unfused step
int gemm_lnbwd8(const void* dY, const void* Wt, int M, int K, const float* w_dq, const float* a_scale, const float* x, const float* mean, const float* rstd,
                const float* gamma, const float* dres, float* dx, bf16* g, uint8_t* g8, const float* g8_scale, float* g8_amax, const float* row_scale, int rps,
                float* dgamma, float* dbeta, float* dbias_up, hipStream_t st) {
  GemmArgs a{};
  a.A = reinterpret_cast<const bf16*>(dY); a.B = reinterpret_cast<const bf16*>(Wt); a.M = M; a.N = 384; a.K = K; a.lda = K; a.ldb = K; a.epi = EPI_LNBWD; a.C = dx; a.ldc = 384;
  if (w_dq) { a.fp8 = 1; a.dq = w_dq; a.dq_mul = 1.0f; a.dq_div = a_scale; }
  a.resid = dres; a.row_scale = row_scale; a.rows_per_seq = rps; a.ln_gamma = gamma; a.ln_mean = const_cast<float*>(mean);
  a.ln_rstd = const_cast<float*>(rstd); a.lnb_x = x; a.lnb_g = g; a.lnb_dgamma = dgamma; a.lnb_dbeta = dbeta; a.lnb_dbias_up = dbias_up;
  a.q8 = g8; a.q8_scale_ptr = g8_scale; a.q8_amax = g8_amax;
  return atst_gemm_nt(a, st);
}
// e4m3 dgrad GEMM: dY8 [M,K] (gradient operand, quantised with the device scale *a_scale), Wt8 [N,K] (transposed weight shadow, per-tensor
// factor *w_dq); out = acc * *w_dq / *a_scale.  EPI_DGELU: also the e4m3 copy of du for the next dgrad GEMM + its amax.
int gemm8_bwd(const uint8_t* dY8, const uint8_t* Wt8, int M, int N, int K, int epi, void* C, hipStream_t st, const float* w_dq, const float* a_scale,
              const bf16* U = nullptr, float* colsum = nullptr, uint8_t* q8 = nullptr, const float* q8_scale = nullptr, float* q8_amax = nullptr) {
  GemmArgs a{};
  a.A = reinterpret_cast<const bf16*>(dY8); a.B = reinterpret_cast<const bf16*>(Wt8); a.M = M; a.N = N; a.K = K; a.lda = K; a.ldb = K;
  a.epi = epi; a.C = C; a.ldc = N; a.fp8 = 1; a.dq = w_dq; a.dq_mul = 1.0f; a.dq_div = a_scale; a.U = U; a.colsum = colsum;
  a.q8 = q8; a.q8_scale_ptr = q8_scale; a.q8_amax = q8_amax;
  return atst_gemm_nt(a, st);
}
int wgrad(const bf16* dY, const bf16* X, int M, int N, int K, float* dW, hipStream_t st) {
  WgradArgs a{};
  a.dY = dY; a.X = X; a.M = M; a.N = N; a.K = K; a.ldy = N; a.ldx = K; a.dW = dW; a.ldw = K; a.m_per_split = 0;
  return atst_gemm_tn(a, st);
}
}  // namespace

inline int patch_k(const atst_encoder_t* e) { return (e->patch_h > 0 ? e->patch_h : 64) * (e->patch_w > 0 ? e->patch_w : 4); }
// Rows between consecutive sequences in every [tokens, *] tensor of a pass.  Default NP (the 32-row-tile-aligned token count the
// attention kernels work on); a smaller value packs the sequences -- 1 s local views are 26 tokens in tiles of 32: the GEMM,
// LayerNorm and weight-gradient kernels then run over 19 % fewer rows, and the NP < 256 attention kernels treat the rows that
// complete a sequence's last tile as absent (attention.hip: stride).  The workspace is sized for NP rows per sequence either way.
static inline int row_stride(const atst_encoder_t* e) { return e->row_stride > 0 ? e->row_stride : e->NP; }

extern "C" size_t atst_encoder_ws_bytes(int S, int NP, int C, int H, int depth, int train, int fp8) {
  return carve(nullptr, S, NP, C, H, depth, train, fp8).bytes;
}
extern "C" size_t atst_encoder_ws_bytes_geo(int S, int NP, int C, int H, int depth, int train, int fp8, int patch_h, int patch_w) {
  return carve(nullptr, S, NP, C, H, depth, train, fp8, patch_h * patch_w).bytes;
}

static bool check(const atst_encoder_t* e) {
  if (!e || e->depth < 1 || e->depth > ATST_MAX_DEPTH) return false;
  if (e->C != e->H * 64 || (e->C != 384 && e->C != 768)) return false;
  if (e->NP != 32 && e->NP != 64 && e->NP != 128 && e->NP != 256) return false;
  if (e->n_tok + e->use_cls > e->NP) return false;
  if (e->row_stride != 0 && (e->row_stride < e->n_tok + e->use_cls || e->row_stride > e->NP || (e->NP == 256 && e->row_stride != 256))) return false;
  if (patch_k(e) % 256 || patch_k(e) > 1024) return false;         // patch GEMM / weight gradient tiles
  if (e->ws_bytes < carve(nullptr, e->S, e->NP, e->C, e->H, e->depth, e->train, e->fp8, patch_k(e)).bytes) return false;
  return true;
}

extern "C" const uint16_t* atst_encoder_out(const atst_encoder_t* e) {
  return reinterpret_cast<const uint16_t*>(carve(e->ws, e->S, e->NP, e->C, e->H, e->depth, e->train, e->fp8, patch_k(e)).hN);
}
extern "C" uint16_t* atst_encoder_dout(const atst_encoder_t* e) {
  return reinterpret_cast<uint16_t*>(carve(e->ws, e->S, e->NP, e->C, e->H, e->depth, e->train, e->fp8, patch_k(e)).dout);
}
extern "C" const float* atst_encoder_block_out(const atst_encoder_t* e, int i) {
  return carve(e->ws, e->S, e->NP, e->C, e->H, e->depth, e->train, e->fp8, patch_k(e)).x[2 * i + 2];
}
extern "C" const float* atst_encoder_tokens(const atst_encoder_t* e) {
  return carve(e->ws, e->S, e->NP, e->C, e->H, e->depth, e->train, e->fp8, patch_k(e)).x[0];
}

extern "C" int atst_encoder_fwd(const atst_encoder_t* e, void* stream) {
  if (!check(e)) return ATST_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const Ws w = carve(e->ws, e->S, e->NP, e->C, e->H, e->depth, e->train, e->fp8, patch_k(e));
  const int S = e->S, NP = e->NP, C = e->C, RS = row_stride(e), M = S * RS;   // RS < NP: packed sequences (see row_stride)
  const float* p = e->p32; const bf16* q = B16(e->p16);
  const atst_enc_off_t& o = e->off;

  const bool f8 = e->fp8 != 0;                      // forward GEMMs on e4m3 copies of the operands (ATST-base recipe, configs[4])
  if (f8 && (!e->p8 || !e->w_dq || C % 384)) return ATST_EINVAL;
  const bool fuse_ln = C == 384 && !f8;             // N == 384: the residual GEMM blocks own whole rows
  const int PK = patch_k(e);
  RUN(atst_patchify(e->mel, S, e->width, RS, e->use_cls, w.patches, st, e->patch_h > 0 ? e->patch_h : 64, e->patch_w > 0 ? e->patch_w : 4));
  RUN(atst_token_table(e->use_cls ? p + o.cls_token : nullptr, p + o.pos_embed, p + o.patch_b, RS, e->n_tok, C, e->use_cls, w.table, st));
  {
    GemmArgs a{};
    a.A = w.patches; a.B = q + o.patch_w; a.M = M; a.N = C; a.K = PK; a.lda = PK; a.ldb = PK; a.epi = EPI_PATCH;
    a.C = w.x[0]; a.ldc = C; a.bias = p + o.patch_b; a.rows_per_seq = RS; a.table = w.table; a.rowflag = e->rowflag;
    a.alt = p + o.mask_embed;
    RUN(atst_gemm_nt(a, st));
  }
  for (int i = 0; i < e->depth; ++i) {
    const atst_layer_off_t& lo = o.layer[i];
    const LayerWs& l = w.L[i];
    const float* s1 = e->dp_scale ? e->dp_scale + (size_t)(2 * i) * S : nullptr;
    const float* s2 = e->dp_scale ? e->dp_scale + (size_t)(2 * i + 1) * S : nullptr;
    if (f8) {
      const uint8_t* q8 = e->p8; const float* dq = e->w_dq + 4 * i;
      const size_t MC = (size_t)M * C;
      // the e4m3 operand copies come out of the producing kernels (LayerNorm, GELU epilogue); only the attention output
      // is quantised by a pass of its own
      unsigned* sat = e->f8_sat;                      // clipped-element counter (or null)
      // activation scales: site k of block i (0: LN1 out -> qkv, 1: attention out -> proj, 2: LN2 out -> fc1, 3: GELU out -> fc2).  With
      // f8_act_scale / f8_act_amax ([depth][4] device floats) the scale is the caller's running (delayed) one and this pass records max |x|
      // of every site; without them the constants ACT_SCALE / ACT_SCALE_GELU.
      auto scp = [&](int k) -> const float* { return e->f8_act_scale ? e->f8_act_scale + 4 * i + k : nullptr; };
      auto amp = [&](int k) -> float* { return e->f8_act_amax ? e->f8_act_amax + (size_t)(4 * i + k) * AMAX_SITE_STRIDE : nullptr; };
      // bf16 copies nobody will read are not written (fp8_lean; an inference pass never reads them): every GEMM of this forward takes the e4m3
      // copy, and the backward's only readers are the bf16 weight gradients -- lean >= 1: fc1 / fc2 take h28 / a8, lean >= 2: qkv takes h18
      const int lean = e->train ? e->fp8_lean : 2;
      bf16* const h1o = lean >= 2 ? nullptr : l.h1; bf16* const h2o = lean >= 1 ? nullptr : l.h2; bf16* const ao = lean >= 1 ? nullptr : l.a;
      // fp8 INFERENCE / TEACHER passes keep their residual stream in bf16 (round 6; VERDICT r5 item 4, candidate (i)): no gradient flows through them and
      // their features already carry the e4m3 staircase (7.9e-2 against fp32), while the fp32 stream is 12 of the ~17 bytes per element and sub-layer that
      // the residual GEMM epilogues and the LayerNorm kernels move.  The stream lives in the pass's (unused: lean) bf16 LayerNorm-1 buffer; the patch stage
      // and the taps of the inference API stay fp32 (block 0 reads the fp32 tokens).  Hook 2100 switches it off.
      // d = 384 (round 6): the proj / fc2 GEMM blocks own whole rows, so -- as in the bf16 step -- their epilogue is also the LayerNorm of the new residual
      // row: the e4m3 operand of the next GEMM (and its amax / clip count), the row statistics and, unless lean, the bf16 copy come out of the GEMM
      // and the two ln_fwd passes of a block go away.  The row-wise epilogue streams an fp32 residual row through LDS-DMA slots; it takes precedence over
      // the bf16 stream of the inference passes: measured per sub-layer at M = 131072, fused + fp32 stream 98 us, bf16 stream 109 us + LayerNorm pass 42 us.
      // (An inference pass shares ONE pair of e4m3 operand buffers between its layers, so proj reads o8 from the buffer its epilogue writes h28 to: a block
      // reads and writes the SAME 256 rows, and its last operand read has landed -- vmcnt(0) + barrier at the end of the main loop -- before its first store.)
      const bool fl8 = C == 384 && g_f8_fuse_ln;
      const bool xb16 = !e->train && !e->tap && g_f8_resid16 && !fl8;
      bf16* const xb = w.L[0].h1;
      if (fl8 && i > 0) {}                          // LN1 of this block came out of the previous block's fc2 epilogue
      else
      if (xb16 && i > 0) RUN(atst_ln_fwd_b16in(xb, p + lo.ln1_w, p + lo.ln1_b, h1o, l.mean1, l.rstd1, M, C, st, l.h18, ACT_SCALE, sat, scp(0), amp(0)));
      else
      RUN(atst_ln_fwd(w.x[2 * i], p + lo.ln1_w, p + lo.ln1_b, h1o, l.mean1, l.rstd1, M, C, st, l.h18, ACT_SCALE, sat, scp(0), amp(0)));
      RUN(gemm8(l.h18, q8 + lo.qkv_w, M, 3 * C, C, EPI_BF16, l.qkv, st, dq + 0, ACT_SCALE, nullptr, nullptr, nullptr, 1, nullptr, nullptr, 1.0f, nullptr, scp(0)));
      AttnArgs at{};
      at.qkv = l.qkv; at.valid = e->valid; at.o = l.o; at.lse = l.lse; at.S = S; at.H = e->H; at.NP = NP; at.stride = RS;
      // the e4m3 copy of the attention output (the proj GEMM's operand) comes out of the NP = 256 forward kernel itself -- next to the bf16 output the
      // backward reads, or instead of it in an inference pass; the other geometries quantise the bf16 output in a pass of their own
      const bool o8fused = atst_attn_fwd_q8_ok(NP, e->H);
      if (o8fused) {
        at.o8 = l.o8; at.o8_scale = scp(1); at.o8_scale_k = ACT_SCALE; at.o8_amax = scp(1) ? amp(1) : nullptr; at.o8_sat = sat;
        if (!e->train) at.o = nullptr;
      }
      RUN(atst_attn_fwd(at, st));
      if (!o8fused) {
        if (scp(1)) RUN(atst_quant_fp8_dyn(l.o, MC, scp(1), l.o8, amp(1), st, sat));
        else RUN(atst_quant_fp8(l.o, MC, ACT_SCALE, l.o8, st, sat));
      }
      if (xb16) {
        RUN(gemm8(l.o8, q8 + lo.proj_w, M, C, C, EPI_RESID, xb, st, dq + 1, ACT_SCALE, p + lo.proj_b, i > 0 ? reinterpret_cast<const float*>(xb) : w.x[0], s1, RS, nullptr, nullptr,
                  1.0f, nullptr, scp(1), nullptr, nullptr, i > 0 ? 1 : 0, 1));
        RUN(atst_ln_fwd_b16in(xb, p + lo.ln2_w, p + lo.ln2_b, h2o, l.mean2, l.rstd2, M, C, st, l.h28, ACT_SCALE, sat, scp(2), amp(2)));
      } else if (fl8) {
        RUN(gemm8(l.o8, q8 + lo.proj_w, M, C, C, EPI_RESID, w.x[2 * i + 1], st, dq + 1, ACT_SCALE, p + lo.proj_b, w.x[2 * i], s1, RS, nullptr, l.h28, ACT_SCALE, sat, scp(1),
                  scp(2), amp(2), 0, 0, p + lo.ln2_w, p + lo.ln2_b, h2o, l.mean2, l.rstd2));
      } else {
      RUN(gemm8(l.o8, q8 + lo.proj_w, M, C, C, EPI_RESID, w.x[2 * i + 1], st, dq + 1, ACT_SCALE, p + lo.proj_b, w.x[2 * i], s1, RS, nullptr, nullptr, 1.0f, nullptr, scp(1)));
      RUN(atst_ln_fwd(w.x[2 * i + 1], p + lo.ln2_w, p + lo.ln2_b, h2o, l.mean2, l.rstd2, M, C, st, l.h28, ACT_SCALE, sat, scp(2), amp(2)));
      }
      RUN(gemm8(l.h28, q8 + lo.fc1_w, M, 4 * C, C, EPI_BIAS_GELU, e->train ? l.u : nullptr, st, dq + 2, ACT_SCALE, p + lo.fc1_b, nullptr, nullptr, 1, ao,
                l.a8, ACT_SCALE_GELU, sat, scp(2), scp(3), amp(3)));
      if (xb16) RUN(gemm8(l.a8, q8 + lo.fc2_w, M, C, 4 * C, EPI_RESID, xb, st, dq + 3, ACT_SCALE_GELU, p + lo.fc2_b, reinterpret_cast<const float*>(xb), s2, RS, nullptr, nullptr,
                          1.0f, nullptr, scp(3), nullptr, nullptr, 1, 1));
      else if (fl8) {                                 // fc2 + residual + (LN1 of the next block -> its e4m3 qkv operand | final norm -> bf16)
        const bool last = i + 1 == e->depth;
        const LayerWs& nl = w.L[last ? i : i + 1];
        const atst_layer_off_t& no = o.layer[last ? i : i + 1];
        const float* nsc = (!last && e->f8_act_scale) ? e->f8_act_scale + 4 * (i + 1) : nullptr;
        float* nam = (!last && e->f8_act_amax) ? e->f8_act_amax + (size_t)(4 * (i + 1)) * AMAX_SITE_STRIDE : nullptr;
        RUN(gemm8(l.a8, q8 + lo.fc2_w, M, C, 4 * C, EPI_RESID, w.x[2 * i + 2], st, dq + 3, ACT_SCALE_GELU, p + lo.fc2_b, w.x[2 * i + 1], s2, RS, nullptr,
                  last ? nullptr : nl.h18, ACT_SCALE, last ? nullptr : sat, scp(3), nsc, nam, 0, 0,
                  p + (last ? o.norm_w : no.ln1_w), p + (last ? o.norm_b : no.ln1_b), last ? w.hN : (lean >= 2 ? nullptr : nl.h1), last ? w.meanN : nl.mean1, last ? w.rstdN : nl.rstd1));
      } else
      RUN(gemm8(l.a8, q8 + lo.fc2_w, M, C, 4 * C, EPI_RESID, w.x[2 * i + 2], st, dq + 3, ACT_SCALE_GELU, p + lo.fc2_b, w.x[2 * i + 1], s2, RS, nullptr, nullptr, 1.0f, nullptr, scp(3)));
    } else {
      if (i == 0 || !fuse_ln) RUN(atst_ln_fwd(w.x[2 * i], p + lo.ln1_w, p + lo.ln1_b, l.h1, l.mean1, l.rstd1, M, C, st));
      RUN(gemm(l.h1, q + lo.qkv_w, M, 3 * C, C, EPI_BF16, l.qkv, st));
      AttnArgs at{};
      at.qkv = l.qkv; at.valid = e->valid; at.o = l.o; at.lse = l.lse; at.S = S; at.H = e->H; at.NP = NP; at.stride = RS;
      RUN(atst_attn_fwd(at, st));
      if (fuse_ln) {                                  // proj + residual + LN2 in one kernel
        RUN(gemm(l.o, q + lo.proj_w, M, C, C, EPI_RESID, w.x[2 * i + 1], st, p + lo.proj_b, w.x[2 * i], s1, RS, nullptr, nullptr,
                 nullptr, p + lo.ln2_w, p + lo.ln2_b, l.h2, l.mean2, l.rstd2));
      } else {
        RUN(gemm(l.o, q + lo.proj_w, M, C, C, EPI_RESID, w.x[2 * i + 1], st, p + lo.proj_b, w.x[2 * i], s1, RS));
        RUN(atst_ln_fwd(w.x[2 * i + 1], p + lo.ln2_w, p + lo.ln2_b, l.h2, l.mean2, l.rstd2, M, C, st));
      }
      RUN(gemm(l.h2, q + lo.fc1_w, M, 4 * C, C, EPI_BIAS_GELU, e->train ? l.u : nullptr, st, p + lo.fc1_b, nullptr, nullptr, 1, l.a));
      if (fuse_ln) {                                  // fc2 + residual + (LN1 of the next block | final norm)
        const bool last = i + 1 == e->depth;
        const LayerWs& nl = w.L[last ? i : i + 1];
        RUN(gemm(l.a, q + lo.fc2_w, M, C, 4 * C, EPI_RESID, w.x[2 * i + 2], st, p + lo.fc2_b, w.x[2 * i + 1], s2, RS, nullptr, nullptr,
                 nullptr, p + (last ? o.norm_w : o.layer[i + 1].ln1_w), p + (last ? o.norm_b : o.layer[i + 1].ln1_b),
                 last ? w.hN : nl.h1, last ? w.meanN : nl.mean1, last ? w.rstdN : nl.rstd1));
      } else {
        RUN(gemm(l.a, q + lo.fc2_w, M, C, 4 * C, EPI_RESID, w.x[2 * i + 2], st, p + lo.fc2_b, w.x[2 * i + 1], s2, RS));
      }
    }
    if (e->tap && i >= e->tap_first) {
      hipError_t rc = hipMemcpyAsync(e->tap + (size_t)(i - e->tap_first) * M * C, w.x[2 * i + 2], (size_t)M * C * sizeof(float),
                                     hipMemcpyDeviceToDevice, st);
      if (rc != hipSuccess) return (int)rc;
    }
  }
  if (f8 && !e->train && !e->tap && g_f8_resid16 && e->depth > 0 && !(C == 384 && g_f8_fuse_ln)) RUN(atst_ln_fwd_b16in(w.L[0].h1, p + o.norm_w, p + o.norm_b, w.hN, w.meanN, w.rstdN, M, C, st));
  else
  if (f8 && C == 384 && g_f8_fuse_ln) {}            // the final norm came out of the last fc2 epilogue
  else
  if (!fuse_ln) RUN(atst_ln_fwd(w.x[2 * e->depth], p + o.norm_w, p + o.norm_b, w.hN, w.meanN, w.rstdN, M, C, st));
  return ATST_OK;
}

// Backward over blocks [lo, hi) (descending); head: also the final LayerNorm backward (needs hi == depth); tail: also the
// token stage (needs lo == 0).  At every block boundary the residual gradient is back in dxA and the bf16 operand of the
// next block's MLP branch is in w.g, so a split needs no state besides the workspace.
static int encoder_bwd_range(const atst_encoder_t* e, int lo, int hi, bool head, bool tail, void* stream);

extern "C" int atst_encoder_bwd(const atst_encoder_t* e, void* stream) {
  if (!check(e)) return ATST_EINVAL;
  return encoder_bwd_range(e, 0, e->depth, true, true, stream);
}
/* part 0: final LayerNorm + blocks [split, depth) ; part 1: blocks [0, split) + token stage.  Lets the caller start the
 * gradient all-reduce of the upper blocks while the lower ones are still being differentiated.                          */
extern "C" int atst_encoder_bwd_part(const atst_encoder_t* e, int part, int split, void* stream) {
  if (!check(e) || split < 0 || split > e->depth || (part != 0 && part != 1)) return ATST_EINVAL;
  return part == 0 ? encoder_bwd_range(e, split, e->depth, true, false, stream) : encoder_bwd_range(e, 0, split, false, true, stream);
}

extern "C" int atst_encoder_bwd_range(const atst_encoder_t* e, int lo, int hi, void* stream) {
  if (!check(e) || lo < 0 || hi > e->depth || lo >= hi) return ATST_EINVAL;
  return encoder_bwd_range(e, lo, hi, hi == e->depth, lo == 0, stream);
}

static int encoder_bwd_range(const atst_encoder_t* e, int lo, int hi, bool head, bool tail, void* stream) {
  if (!check(e) || !e->train || !e->p16t || !e->g32) return ATST_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const Ws w = carve(e->ws, e->S, e->NP, e->C, e->H, e->depth, e->train, e->fp8, patch_k(e));
  const int S = e->S, NP = e->NP, C = e->C, RS = row_stride(e), M = S * RS, D = e->depth;
  const float* p = e->p32; const bf16* qt = B16(e->p16t);
  float* G = e->g32;
  const atst_enc_off_t& o = e->off;
  auto dps = [&](int layer, int which) -> const float* {
    return e->dp_scale ? e->dp_scale + (size_t)(2 * layer + which) * S : nullptr;
  };

  // C == 384: the two N = 384 dgrad GEMMs of a block own whole rows, so their epilogue runs the LayerNorm backward itself
  // (no dh round trip through HBM, no separate pass over x and the residual gradient)
  const bool fuse_lnb = C == 384 && !(e->fp8 && e->fp8_bwd >= 1);   // (an fp8 backward -- recording or on -- runs the unfused LayerNorm backward, whose kernel writes the e4m3 gradient copies and their amax: round 6, d = 384)
  // fp8 dgrad (BASELINE.json configs[4]): the fc2 / fc1 / proj dgrad GEMMs of a block on e4m3 operands; qkv dgrad and weight gradients stay bf16.  Gradient
  // operands use DELAYED scaling: site (block i, k) -- k = 0: g (into fc2), 1: du (into fc1), 2: g2 (into proj), 3: dqkv (into qkv: fp8_wgrad >= 2) -- is
  // quantised with g8_scale[4 i + k], the scale derived from the amax seen in the previous step, and records this step's amax in
  // g8_amax[4 i + k]; fp8_bwd == 1 only records (first step: bf16 dgrad), == 2 also computes in fp8.
  const bool rec8 = e->fp8 && e->fp8_bwd >= 1 && e->g8_amax && !fuse_lnb;   // (!fuse_lnb: implied)
  const bool use8 = rec8 && e->fp8_bwd >= 2 && e->p8t && e->g8_scale && e->w_dq;
  // d = 384 with the e4m3 dgrads ON (round 6): the LayerNorm backward is back inside the fc1 / qkv dgrad epilogues, which now also write the e4m3 gradient
  // copies and post their amax (the recording step, fp8_bwd == 1, keeps the separate pass)
  const bool fuse_lnb8 = C == 384 && use8 && g_f8_fuse_ln;
  const bool use8w = use8 && e->fp8_wgrad && e->f8_act_scale_bwd && M % 64 == 0;   // e4m3 weight gradients of fc1 / fc2 / proj (N, K multiples of 128: C = 768 and, round 6, C = 384)
  // fp8_wgrad == 2: the qkv Linear too -- the NP = 256 attention backward then writes dqkv as e4m3 ONLY (site 3) and both the qkv weight gradient
  // and the qkv dgrad read that copy.  fp8_wgrad == 3 (or 2 while the dgrad itself is still recording): bf16 qkv gradient, site 3's amax taken by
  // a pass over the bf16 dqkv -- the step that gives the site its first scale.
  const bool q8ok = atst_attn_bwd_q8_ok(NP) && w.dscr && w.dqkv8;
  const bool use8q = use8w && e->fp8_wgrad == 2 && q8ok;
  const bool rec8q = rec8 && !use8q && e->fp8_wgrad >= 2 && q8ok && e->g8_scale;
  // a lean forward (fp8_lean) did not write the bf16 activations the bf16 weight gradients would read: the backward must be the e4m3 one it announced
  if (e->fp8 && ((e->fp8_lean >= 1 && !use8w) || (e->fp8_lean >= 2 && !use8q))) return ATST_EINVAL;
  // ... and with e4m3 dgrads + weight gradients the bf16 copies of g, g2 and du have no reader either (backward-internal: no contract with the forward)
  const bool skip16 = use8w;
  auto gs8 = [&](int layer, int k) -> const float* { return use8 ? e->g8_scale + 4 * layer + k : nullptr; };
  auto ga8 = [&](int layer, int k) -> float* { return rec8 ? e->g8_amax + (size_t)(4 * layer + k) * AMAX_SITE_STRIDE : nullptr; };
  float* cur = w.dxA; float* oth = w.dxB;
  if (head) {
    LnBwdArgs a{};
    a.dy = w.dout; a.x = w.x[2 * D]; a.mean = w.meanN; a.rstd = w.rstdN; a.gamma = p + o.norm_w; a.dres = nullptr;
    a.dx = cur; a.g = skip16 ? nullptr : w.g; a.row_scale = dps(D - 1, 1); a.rows_per_seq = RS;
    a.dgamma = G + o.norm_w; a.dbeta = G + o.norm_b; a.dbias_up = G + o.layer[D - 1].fc2_b; a.M = M; a.C = C;
    a.g8 = use8 ? w.g8 : nullptr; a.g8_scale = gs8(D - 1, 0); a.g_amax = ga8(D - 1, 0);
    RUN(atst_ln_bwd(a, st));
  }
  for (int i = hi - 1; i >= lo; --i) {
    const atst_layer_off_t& lo_ = o.layer[i];
    const LayerWs& l = w.L[i];
    // ---- MLP branch: x_out = x_mid + s2 * (fc2(gelu(fc1(LN2(x_mid)))) + b2) ; w.g = s2 * d(x_out)
    // The block's four weight gradients are independent of everything downstream: they are launched together after the
    // attention backward (atst_gemm_tn_group), which is why the two residual-branch gradients live in separate buffers.
    if (use8) {
      RUN(gemm8_bwd(w.g8, e->p8t + lo_.fc2_w, M, 4 * C, C, EPI_DGELU, skip16 ? nullptr : w.du, st, e->w_dq + 4 * i + 3, gs8(i, 0), l.u, G + lo_.fc1_b,
                    w.du8, gs8(i, 1), ga8(i, 1)));
    } else {
      GemmArgs a{};
      a.A = w.g; a.B = qt + lo_.fc2_w; a.M = M; a.N = 4 * C; a.K = C; a.lda = C; a.ldb = C; a.epi = EPI_DGELU; a.C = w.du; a.ldc = 4 * C;
      a.U = l.u; a.colsum = G + lo_.fc1_b; a.rows_per_seq = 1; a.q8_amax = ga8(i, 1);
      RUN(atst_gemm_nt(a, st));
    }
    if (fuse_lnb) {
      RUN(gemm_lnbwd(w.du, qt + lo_.fc1_w, M, 4 * C, w.x[2 * i + 1], l.mean2, l.rstd2, p + lo_.ln2_w, cur, oth, w.g2, dps(i, 0), RS,
                     G + lo_.ln2_w, G + lo_.ln2_b, G + lo_.proj_b, st));
      float* t = cur; cur = oth; oth = t;
    } else if (fuse_lnb8) {
      RUN(gemm_lnbwd8(w.du8, e->p8t + lo_.fc1_w, M, 4 * C, e->w_dq + 4 * i + 2, gs8(i, 1), w.x[2 * i + 1], l.mean2, l.rstd2, p + lo_.ln2_w, cur, oth,
                      skip16 ? nullptr : w.g2, w.g28, gs8(i, 2), ga8(i, 2), dps(i, 0), RS, G + lo_.ln2_w, G + lo_.ln2_b, G + lo_.proj_b, st));
      float* t = cur; cur = oth; oth = t;
    } else {
      if (use8) RUN(gemm8_bwd(w.du8, e->p8t + lo_.fc1_w, M, C, 4 * C, EPI_BF16, w.dh, st, e->w_dq + 4 * i + 2, gs8(i, 1)));
      else RUN(gemm(w.du, qt + lo_.fc1_w, M, C, 4 * C, EPI_BF16, w.dh, st));
      LnBwdArgs a{};
      a.g8 = use8 ? w.g28 : nullptr; a.g8_scale = gs8(i, 2); a.g_amax = ga8(i, 2);
      a.dy = w.dh; a.x = w.x[2 * i + 1]; a.mean = l.mean2; a.rstd = l.rstd2; a.gamma = p + lo_.ln2_w; a.dres = cur;
      a.dx = oth; a.g = skip16 ? nullptr : w.g2; a.row_scale = dps(i, 0); a.rows_per_seq = RS;
      a.dgamma = G + lo_.ln2_w; a.dbeta = G + lo_.ln2_b; a.dbias_up = G + lo_.proj_b; a.M = M; a.C = C;
      RUN(atst_ln_bwd(a, st));
      float* t = cur; cur = oth; oth = t;
    }
    // ---- attention branch: x_mid = x_in + s1 * (proj(attn(LN1(x_in))) + bp) ; w.g = s1 * d(x_mid)
    if (use8) RUN(gemm8_bwd(w.g28, e->p8t + lo_.proj_w, M, C, C, EPI_BF16, w.d_o, st, e->w_dq + 4 * i + 1, gs8(i, 2)));
    else RUN(gemm(w.g2, qt + lo_.proj_w, M, C, C, EPI_BF16, w.d_o, st));
    AttnArgs at{};
    at.qkv = l.qkv; at.valid = e->valid; at.o = l.o; at.lse = l.lse; at.d_o = w.d_o; at.dqkv = w.dqkv; at.dscratch = w.dscr;
    at.S = S; at.H = e->H; at.NP = NP; at.stride = RS;
    if (use8q) { at.dqkv = nullptr; at.dqkv8 = w.dqkv8; at.q8_scale = gs8(i, 3); at.q8_amax = ga8(i, 3); }
    RUN(atst_attn_bwd(at, st));
    if (rec8q) RUN(atst_quant_fp8_dyn(w.dqkv, (size_t)M * 3 * C, e->g8_scale + 4 * i + 3, nullptr, ga8(i, 3), st, nullptr));   // amax only
    if (use8w) {
      // e4m3 weight gradients of fc1 / fc2 / proj (round 5): dY8 = the e4m3 gradient operands the dgrad GEMMs of this block have just used (scales
      // g8_scale[4 i + k]), X8 = the e4m3 activation copies the forward kept (scales f8_act_scale_bwd[4 i + k]: what the forward quantised WITH, the
      // caller's snapshot -- its running scales have moved on since).  The qkv gradient: e4m3 as well when the attention backward wrote dqkv8.
      const float* sxb = e->f8_act_scale_bwd + 4 * i;
      // one launch for the block's e4m3 weight gradients: they share M, and together they need a fraction of the M-splits (= fp32 atomics) each needs alone
      const Wgrad8Item items[4] = {
        {w.du8, l.h28, 4 * C, C, 4 * C, C, G + lo_.fc1_w, C, gs8(i, 1), sxb + 2},
        {w.g8, l.a8, C, 4 * C, C, 4 * C, G + lo_.fc2_w, 4 * C, gs8(i, 0), sxb + 3},
        {w.g28, l.o8, C, C, C, C, G + lo_.proj_w, C, gs8(i, 2), sxb + 1},
        {w.dqkv8, l.h18, 3 * C, C, 3 * C, C, G + lo_.qkv_w, C, gs8(i, 3), sxb + 0}};
      RUN(atst_gemm_tn8_group(items, use8q ? 4 : 3, M, st));
      if (!use8q) RUN(wgrad(w.dqkv, l.h1, M, 3 * C, C, G + lo_.qkv_w, st));
    } else {
      WgradArgs wg[4] = {};
      auto set = [&](int k, const bf16* dY, const bf16* X, int N, int K, float* dW) {
        wg[k].dY = dY; wg[k].X = X; wg[k].M = M; wg[k].N = N; wg[k].K = K; wg[k].ldy = N; wg[k].ldx = K; wg[k].dW = dW; wg[k].ldw = K;
      };
      set(0, w.du, l.h2, 4 * C, C, G + lo_.fc1_w);
      set(1, w.g, l.a, C, 4 * C, G + lo_.fc2_w);
      set(2, w.dqkv, l.h1, 3 * C, C, G + lo_.qkv_w);
      set(3, w.g2, l.o, C, C, G + lo_.proj_w);
      RUN(atst_gemm_tn_group(wg, 4, st));
    }
    if (fuse_lnb) {
      RUN(gemm_lnbwd(w.dqkv, qt + lo_.qkv_w, M, 3 * C, w.x[2 * i], l.mean1, l.rstd1, p + lo_.ln1_w, cur, oth, i > 0 ? w.g : nullptr,
                     i > 0 ? dps(i - 1, 1) : nullptr, RS, G + lo_.ln1_w, G + lo_.ln1_b, i > 0 ? G + o.layer[i - 1].fc2_b : nullptr, st));
      float* t = cur; cur = oth; oth = t;
    } else if (fuse_lnb8) {                           // e4m3 operands when the attention backward wrote dqkv8, bf16 otherwise (local views) -- the e4m3 copy of g either way
      RUN(gemm_lnbwd8(use8q ? (const void*)w.dqkv8 : (const void*)w.dqkv, use8q ? (const void*)(e->p8t + lo_.qkv_w) : (const void*)(qt + lo_.qkv_w), M, 3 * C,
                      use8q ? e->w_dq + 4 * i + 0 : nullptr, use8q ? gs8(i, 3) : nullptr, w.x[2 * i], l.mean1, l.rstd1, p + lo_.ln1_w, cur, oth,
                      (i > 0 && !skip16) ? w.g : nullptr, i > 0 ? w.g8 : nullptr, i > 0 ? gs8(i - 1, 0) : nullptr, i > 0 ? ga8(i - 1, 0) : nullptr,
                      i > 0 ? dps(i - 1, 1) : nullptr, RS, G + lo_.ln1_w, G + lo_.ln1_b, i > 0 ? G + o.layer[i - 1].fc2_b : nullptr, st));
      float* t = cur; cur = oth; oth = t;
    } else {
      // the qkv dgrad: e4m3 when the attention backward wrote the e4m3 dqkv itself (a quantisation PASS over a bf16 dqkv -- 906 MB, 255 us at
      // M = 131072 -- costs more than the e4m3 GEMM saves: measured 125.4 vs 124.x ms per step in round 4)
      if (use8q) RUN(gemm8_bwd(w.dqkv8, e->p8t + lo_.qkv_w, M, C, 3 * C, EPI_BF16, w.dh, st, e->w_dq + 4 * i + 0, gs8(i, 3)));
      else RUN(gemm(w.dqkv, qt + lo_.qkv_w, M, C, 3 * C, EPI_BF16, w.dh, st));
      LnBwdArgs a{};
      if (i > 0) { a.g8 = use8 ? w.g8 : nullptr; a.g8_scale = gs8(i - 1, 0); a.g_amax = ga8(i - 1, 0); }
      a.dy = w.dh; a.x = w.x[2 * i]; a.mean = l.mean1; a.rstd = l.rstd1; a.gamma = p + lo_.ln1_w; a.dres = cur;
      a.dx = oth; a.g = (i > 0 && !skip16) ? w.g : nullptr; a.row_scale = i > 0 ? dps(i - 1, 1) : nullptr; a.rows_per_seq = RS;
      a.dgamma = G + lo_.ln1_w; a.dbeta = G + lo_.ln1_b; a.dbias_up = i > 0 ? G + o.layer[i - 1].fc2_b : nullptr;
      a.M = M; a.C = C;
      RUN(atst_ln_bwd(a, st));
      float* t = cur; cur = oth; oth = t;
    }
  }
  if (!tail) return ATST_OK;
  // ---- token stage: x0 = (1-m) (patch W^T + b) + m mask_embed + pos  (+ CLS)
  RUN(atst_token_grad(cur, e->rowflag, S, RS, e->n_tok, C, e->use_cls, e->use_cls ? G + o.cls_token : nullptr,
                      G + o.pos_embed, G + o.patch_b, e->rowflag ? G + o.mask_embed : nullptr, w.g, st));
  RUN(wgrad(w.g, w.patches, M, C, patch_k(e), G + o.patch_w, st));
  return ATST_OK;
}
