// Internal launch interface between the kernel translation units and the C-ABI / engine (api.hip, engine.hip).
#pragma once
#include "common.h"

enum GemmEpilogue { EPI_BF16 = 0, EPI_F32 = 1, EPI_BIAS_GELU = 2, EPI_RESID = 3, EPI_DGELU = 4, EPI_PATCH = 5, EPI_LNBWD = 6 };

struct GemmArgs {
  const bf16* A; const bf16* B;      // A [M,K] lda ; B [N,K] ldb   (C = A * B^T)
  int M, N, K, lda, ldb;
  int epi;
  void* C; int ldc;                  // primary output (dtype by epilogue)
  void* C2;                          // EPI_BIAS_GELU: activation output (bf16)
  const float* bias;                 // [N] or null
  const float* resid;                // EPI_RESID: fp32 [M,ldc]
  const float* row_scale;            // EPI_RESID: per-sequence DropPath scale [M / rows_per_seq] or null
  int rows_per_seq;
  const bf16* U;                     // EPI_DGELU: saved pre-activation
  const float* table;                // EPI_PATCH: [rows_per_seq, N] per-token additive table
  const uint8_t* rowflag;            // EPI_PATCH: [M] 1 = replace by mask token (or null)
  const float* alt;                  // EPI_PATCH: mask_embed [N]
  // EPI_RESID with N == 384: optional fused LayerNorm(eps 1e-6) of the output row (the next sub-layer's pre-LN)
  const float* ln_gamma; const float* ln_beta; bf16* ln_out; float* ln_mean; float* ln_rstd;
  float* colsum;                     // EPI_DGELU: optional fp32 [N] accumulator of the column sums of the output (bias gradient)
  // EPI_LNBWD (N == 384, dgrad GEMMs in front of a LayerNorm): the accumulator row is dy of LayerNorm(x) and the epilogue is
  // that LayerNorm's backward (LnBwdArgs semantics): C = dx fp32 [M,384] = resid (residual gradient, or null) + LN backward;
  // lnb_g = bf16(row_scale * dx) or null; column sums into lnb_dgamma / lnb_dbeta / lnb_dbias_up (null = skip);
  // statistics of the forward in ln_mean / ln_rstd, affine weight in ln_gamma.
  const float* lnb_x; bf16* lnb_g; float* lnb_dgamma; float* lnb_dbeta; float* lnb_dbias_up;
  uint8_t* q8; float q8_scale;       // EPI_BIAS_GELU: optional e4m3 copy of the activation * q8_scale (A operand of the fp8 fc2 GEMM)
  unsigned* q8_sat;                  // EPI_BIAS_GELU: counter of the elements of that copy clipped at +-448 (or null)
  float dq_mul;                      // fp8 GEMMs: host factor on top of *dq (1 / activation scale); 0 is read as 1
  const float* dq_div;               // fp8 GEMMs: optional device scalar the accumulators are DIVIDED by (delayed-scaling quantisation scale of A)
  const float* q8_scale_ptr;         // EPI_DGELU: device scale of the optional e4m3 copy (q8) of the output; amax of the output -> q8_amax
  float* q8_amax;
  const float* dq;                   // fp8 GEMMs: device scalar multiplied into the accumulators (1 / (scale_A * scale_B)); null = 1
  int fp8;                           // operands are OCP e4m3 bytes (K, lda, ldb in elements = bytes); MX-scaled MFMA, unit block scales
  int skew;                          // only read by tools/experiments/gemm_r02_variants.hip (start-up skew experiment)
  int resid_bf16, out_bf16;          // EPI_RESID, e4m3 operands (fp8 inference / teacher passes, round 6): the residual stream is read / written as bf16 [M, ldc] instead of fp32
  int ksplit;                        // set by the launcher (EPI_F32, tiny grids, long K): K is split over `ksplit` blocks per output tile
  float* ks_ws;                      // split-K workspace [ksplit][M][N] of partial sums (library-owned, per stream)
};
int atst_gemm_nt(const GemmArgs& a, hipStream_t st);
void atst_gemm_nt_set_variant(int v);     // tuning hook: -1 auto, 0/1/2 fixed tile configuration

struct WgradArgs {
  const bf16* dY; const bf16* X;     // dY [M, >=N] ldy ; X [M, >=K] ldx
  int M, N, K, ldy, ldx;
  float* dW; int ldw;                // fp32 [N, K], accumulated with atomics
  int m_per_split;                   // 0 = auto
};
int atst_gemm_tn(const WgradArgs& a, hipStream_t st);
// e4m3 weight gradient (gemm_tn8.hip): dW += dY8^T X8 / (*scale_y * *scale_x); N, K % 128 == 0 (256 x 256 tiles, the last of a dimension half valid), M % 64 == 0
int atst_gemm_tn8(const uint8_t* dY8, const uint8_t* X8, int M, int N, int K, int ldy, int ldx, float* dW, int ldw, const float* scale_y,
                  const float* scale_x, hipStream_t st);
struct Wgrad8Item { const uint8_t* dY8; const uint8_t* X8; int N, K, ldy, ldx; float* dW; int ldw; const float* scale_y; const float* scale_x; };
int atst_gemm_tn8_group(const Wgrad8Item* items, int n, int M, hipStream_t st);   // up to four problems sharing M in one launch (one transformer block)
extern int g_tn8_splits;
#define ATST_WGRAD_GROUP_MAX 4
int atst_gemm_tn_group(const WgradArgs* items, int n, hipStream_t st);   // independent weight gradients sharing one launch

// LayerNorm (eps 1e-6), C in {384, 768}
int atst_ln_fwd(const float* x, const float* gamma, const float* beta, bf16* y, float* mean, float* rstd, int M, int C, hipStream_t st,
                uint8_t* y8 = nullptr, float s8 = 1.0f, unsigned* sat = nullptr, const float* s8p = nullptr, float* amax8 = nullptr);
                // y8: optional e4m3 copy of y * scale (fp8 forward), scale = *s8p (device, delayed scaling) or s8 ; sat: clipped-element counter ; amax8: max |y| (atomicMax)
int atst_ln_fwd_b16in(const bf16* x, const float* gamma, const float* beta, bf16* y, float* mean, float* rstd, int M, int C, hipStream_t st,
                      uint8_t* y8 = nullptr, float s8 = 1.0f, unsigned* sat = nullptr, const float* s8p = nullptr, float* amax8 = nullptr);   // bf16 residual stream in (fp8 inference passes)
int atst_ln_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, int M, int C, hipStream_t st);   // fp32 output, no statistics (inference taps)
struct LnBwdArgs {
  const bf16* dy;                    // [M,C] gradient wrt the LN output
  const float* x; const float* mean; const float* rstd; const float* gamma;
  const float* dres;                 // [M,C] fp32 gradient arriving through the residual (or null = 0)
  float* dx;                         // [M,C] fp32 out: dres + LN backward
  bf16* g;                           // [M,C] bf16 out: row_scale * dx   (operand of the upstream proj / fc2 backward), or null
  const float* row_scale; int rows_per_seq;
  uint8_t* g8; const float* g8_scale; float* g_amax;   // optional e4m3 copy of g * *g8_scale (fp8 dgrad operand) ; amax of |g| (atomicMax)
  float* dgamma; float* dbeta;       // [C] fp32 accumulated
  float* dbias_up;                   // [C] fp32 accumulated: column sum of g (bias gradient of the upstream linear) or null
  int M, C;
};
int atst_ln_bwd(const LnBwdArgs& a, hipStream_t st);

// fused attention, head_dim 64, NP (padded tokens / sequence) in {32, 64, 128, 256}
struct AttnArgs {
  const bf16* qkv;                   // [S*NP, 3*C]
  const int* valid;                  // [S] number of valid key tokens per sequence
  bf16* o;                           // fwd out [S*NP, C]
  float* lse;                        // fwd out [S, H, NP]
  const bf16* d_o;                   // bwd in  [S*NP, C]
  bf16* dqkv;                        // bwd out [S*NP, 3*C]
  float* dscratch;                   // bwd scratch [S, H, NP] fp32 (rowsum(dO*O)); null -> two-kernel backward
  int S, H, NP;
  int row_stores;                    // tuning hook (NP = 256 backward): 1 = row-per-lane dK / dV stores instead of the LDS-transposed full-line ones
  uint8_t* o8; const float* o8_scale; float o8_scale_k; float* o8_amax; unsigned* o8_sat;   // NP = 256 forward (fp8): o8 -> e4m3(bf16(o) * scale) [S*NP, C] next to
                                     // o (or instead of it: o == null) ; scale = *o8_scale, or the constant o8_scale_k ; optional amax site / clipped-element counter
  uint8_t* dqkv8; const float* q8_scale; float* q8_amax;   // NP = 256 backward (fp8 qkv gradient path): dqkv8 -> e4m3(bf16(dqkv) * *q8_scale) [S*NP, 3*C]
                                     // written INSTEAD of dqkv, max |bf16(dqkv)| posted to the amax site q8_amax (all three needed)
  int stride;                        // rows between consecutive sequences in qkv / o / d_o / dqkv (0 = NP).  stride < NP: sequences are PACKED --
                                     // the NP - stride rows that complete a sequence's last 32-row tile belong to the next sequence and are treated
                                     // as absent (read as zeros, never written).  NP < 256 kernels only.
};
int atst_attn_fwd(const AttnArgs& a, hipStream_t st);
bool atst_attn_bwd_q8_ok(int NP);
bool atst_attn_fwd_q8_ok(int NP, int H);
int atst_attn_bwd(const AttnArgs& a, hipStream_t st);
void atst_attn_set_variant(int v);

// token plumbing
int atst_patchify(const float* mel, int S, int width, int NP, int use_cls, bf16* out, hipStream_t st, int patch_h = 64, int patch_w = 4);
int atst_patchify_f32(const float* mel, int S, int width, int NP, int use_cls, float* out, hipStream_t st, int patch_h = 64, int patch_w = 4);
int atst_token_table(const float* cls, const float* pos, const float* bias, int NP, int n_tok, int C, int use_cls, float* table, hipStream_t st);
int atst_gather_rows(const bf16* src, const int* rows, int R, int C, float* dst, hipStream_t st);
int atst_scatter_rows(const float* src, const int* rows, int R, int C, bf16* dst, hipStream_t st);
int atst_colsum_bf16(const bf16* x, int M, int N, int ld, float* out, hipStream_t st);
int atst_token_grad(const float* dx0, const uint8_t* rowflag, int S, int NP, int n_tok, int C, int use_cls,
                    float* dcls, float* dpos, float* dbias, float* dmask, bf16* g0, hipStream_t st);

// heads
#define ATST_BN_ROW_BLOCKS 32
int atst_bn_stats(const float* h, int R, int N, float* mean, float* m2, float* scratch /* [ATST_BN_ROW_BLOCKS * N] */, hipStream_t st);
int atst_bn_finish(const float* mean, const float* m2, float count, const float* count_dev, float momentum, float eps, float* running_mean,
                   float* running_var, long long* num_batches, float* rstd, int n, hipStream_t st);
int atst_bn_apply_relu(const float* h, const float* mean, const float* rstd, const float* gamma, const float* beta,
                       int R, int N, bf16* y, hipStream_t st);
int atst_bn_apply_relu_split3(const float* h, const float* mean, const float* rstd, const float* gamma, const float* beta,
                              int R, int N, bf16* y, hipStream_t st);
int atst_bn_relu_bwd(const float* dy, const float* h, const float* mean, const float* rstd, const float* gamma,
                     const float* beta, int R, int N, float* sum_dy, float* sum_dy_xhat, float* scratch /* [2 * ATST_BN_ROW_BLOCKS * N] */, hipStream_t st);
int atst_bn_bwd_dx(const float* dy, const float* h, const float* mean, const float* rstd, const float* gamma,
                   const float* beta, const float* sum_dy, const float* sum_dy_xhat, float inv_count, int R, int N,
                   bf16* dh, hipStream_t st);
int atst_bn_bwd_dx_fp32(const float* dy, const float* h, const float* mean, const float* rstd, const float* gamma,
                        const float* beta, const float* sum_dy, const float* sum_dy_xhat, float inv_count, int R, int N,
                        float* dh, hipStream_t st);
int atst_cast_f32_bf16(const float* x, size_t n, bf16* y, hipStream_t st);
int atst_split3(const float* x, int R, int K, int b_layout, bf16* y, hipStream_t st);
int atst_byol_loss(const float* student, const float* teacher, int B, int ncrops, int D, float* loss, float* dstudent,
                   float* stats, hipStream_t st);

// fp8 (OCP e4m3) operand preparation for the ATST-base forward GEMMs (BASELINE.json configs[4])
int atst_quant_fp8(const bf16* x, size_t n, float scale, uint8_t* y, hipStream_t st, unsigned* sat = nullptr);   // y = e4m3(clamp(x * scale, +-448)) ; sat += clipped elements
// every tensor of `table` (device int32 [n][2] = {element offset, numel}, 256-aligned) of a flat fp32 buffer -> e4m3 at the same
// offsets with a per-tensor scale 448 / amax; dq[t] = amax / 448 (the factor that undoes it).  amax: device scratch [n].
int atst_quant_fp8_dyn(const bf16* x, size_t n, const float* scale /* device */, uint8_t* y /* or null: amax only */, float* amax /* device, atomicMax */, hipStream_t st,
                       unsigned* sat = nullptr /* clipped-element counter */);
int atst_fp8_update_scales(float* amax, float* scale, int n, float margin, hipStream_t st);
int atst_quant_bf16_table_fp8(const bf16* p16, const int* table, int n, const float* dq, uint8_t* p8, hipStream_t st);
int atst_quant_weights_fp8(const float* p32, const int* table, int n, uint8_t* p8, float* dq, float* amax, hipStream_t st);

// optimizer
struct OptimArgs {
  float* p; const float* g; float* m; float* v;       // student flat buffers [n]
  float* t;                                           // teacher flat buffer [n_teacher] (same offsets) or null
  bf16* p_bf16; bf16* t_bf16;                         // bf16 shadow copies (or null)
  const uint8_t* chunk_flags;                         // one byte per 256-element chunk: bit0 decay, bit1 update, bit2 ema
  size_t n; size_t n_teacher;
  float lr_wd, beta1, beta2, om_beta1, om_beta2, eps, step_size, ema_m, om_ema;   // host-rounded from doubles
  float grad_scale;                                   // multiplies g before use (1/world for DDP-sum)
};
int atst_adamw_ema(const OptimArgs& a, hipStream_t st);
int atst_transpose_bf16(const bf16* src, int rows, int cols, bf16* dst, hipStream_t st);
int atst_transpose_bf16_batch(const bf16* src, bf16* dst, const int* table, int n, int total_tiles, hipStream_t st);

// front end
int atst_mel_frontend(const float* wave, int n_clips, int n_samples, int wave_ld, int n_mels, int win_length, const float* window,
                      const float* fb_weights, const int* fb_start, const int* fb_len, int fb_maxlen,
                      float* out, unsigned int* clipmax, hipStream_t st);

// batched spectrogram augmentations (augment.hip)
int atst_rrc_bicubic(const float* in, float* out, const int* params, int B, int H, int W, int CH, int CW, hipStream_t st);
int atst_log_mixup_exp(const float* x, const float* bank, const int* zidx, const int* zstart, const int* xstart, const float* alpha,
                       float* out, int B, int H, int W, int Wz, hipStream_t st);
