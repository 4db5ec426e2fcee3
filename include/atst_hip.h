/* C ABI of libatst_hip.so -- the MI355X (gfx950) kernels behind the ATST / ATST-Frame pre-training hot path.
 *
 * The reference (Audio-WestlakeU/audiossl) has no native code and no FFI: its hot path is eager torch.nn calls.
 * Each entry point below therefore cites the reference *Python* call it replaces (path:line relative to the upstream
 * repository root).  Conventions for every function:
 *   - all pointers are DEVICE pointers owned by the caller (nothing is allocated or freed here), `stream` is a
 *     hipStream_t passed as void*; work is enqueued asynchronously on it;
 *   - bf16 tensors are passed as uint16_t* (raw bfloat16 bits), row-major, torch [out,in] convention for weights;
 *   - the return value is 0 on success, a hipError_t value, or ATST_EINVAL (1001) for an unsupported shape;
 *   - no entry point synchronises the device.
 * Binding sketches for the reference side are in INTEGRATION.md.
 */
#ifndef ATST_HIP_H
#define ATST_HIP_H
#include <stddef.h>
#include <stdint.h>

/* A running-amax SITE (delayed fp8 scaling) is not one float: same-address atomics serialise at L2 and a launch posts one per wave.  Every
 * `amax` argument below that is filled by atomicMax points at ATST_AMAX_SITE_STRIDE floats per site, of which ATST_AMAX_SLOTS (256 B apart) are
 * used; the site's value is the max over its slots (AtstEngine reduces them once per step), and the caller clears all of them.          */
#ifndef ATST_AMAX_SLOTS
#define ATST_AMAX_SLOTS 16
#define ATST_AMAX_SLOT_STRIDE 64
#define ATST_AMAX_SITE_STRIDE (ATST_AMAX_SLOTS * ATST_AMAX_SLOT_STRIDE)
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define ATST_MAX_DEPTH 24

/* GEMM epilogues (gemm_nt) */
enum { ATST_EPI_BF16 = 0, ATST_EPI_F32 = 1, ATST_EPI_BIAS_GELU = 2, ATST_EPI_RESID = 3, ATST_EPI_DGELU = 4, ATST_EPI_PATCH = 5,
       ATST_EPI_LNBWD = 6 /* only through atst_gemm_nt_lnbwd_bf16 */ };

/* ABI version of this header: bumped whenever a struct layout, a buffer contract or an argument meaning changes.  A caller built against another
 * version must not proceed (audiossl_amd/hip.py checks it at load time).
 *   100  rounds 1-3
 *   110  round 4: atst_encoder_t grew f8_sat / f8_act_scale / f8_act_amax ; every amax argument (atst_quant_fp8_dyn_bf16, g8_amax, f8_act_amax) is an
 *        ATST_AMAX_SITE_STRIDE-float SITE, not one float ; fb_weights of atst_mel_frontend_f32 is tap-major [fb_maxlen][n_mels]
 *   120  round 5: atst_encoder_t grew fp8_wgrad / f8_act_scale_bwd ; an fp8 training workspace also holds per-layer e4m3 activation copies
 *        and an e4m3 dqkv ; fp8_lean ; atst_gemm_tn_fp8, atst_gemm_tn_group_fp8, atst_attention_fwd_fp8, atst_attention_bwd_fp8
 *        (round 6 ADDED entry points without changing an existing one -- atst_gemm_nt_resid_ln_fp8, atst_gemm_nt_lnbwd_q8, atst_attention_fp8_ok -- and widened two
 *        preconditions: atst_gemm_tn_fp8 takes N, K multiples of 128, the fp8 attention entries take NP = 32: the version stays 120)                        */
#define ATST_ABI_VERSION 120
int atst_version(void);   /* = ATST_ABI_VERSION of the header the library was built from */
/* Tuning hooks for A/B measurements (tools/gemm_bench.py, env ATST_TUNE=a,b,... read by audiossl_amd/hip.py); defaults are the
 * measured best.  These are PROCESS-GLOBAL test knobs (plain ints inside the library, read at launch time, no locking): set them from one
 * thread, before the launches they are meant for, and restore the default afterwards (the tests do).  Nothing in the product path calls this.
 *   -1 auto | 0..3 force a 128x128 / 256x128 nt tile config | 4 force the row-384 tile
 *   105/106 wgrad 192x384 LDS-DMA tile off/on        110+r wgrad grid = r rounds of resident blocks (128x128 tile)
 *   120/121/122 192x384 wgrad schedule: all waves issue behind the hand-off / wave rows staggered (default) / 32-row stages in a 4-deep ring
 *   300/301 row-384 tile off/on      302/303/304 256-row tile: never / plain bf16 GEMMs / every epilogue
 *   306/307/308 dGELU GEMM on the 256x384 tile: never / always / when K >= 768 (default)
 *   330+m 4-wave two-blocks-per-CU kernels: 0 only for small grids (default, see 360/361) ; 2 everywhere (256x192 + 128x384/LN)
 *   350/351 apply the tall / 4-wave kernels from M = 8192 (default) / from any M (parity tests of those kernels at small M)
 *   360/361 4-wave kernels for launches of <= 1.5 rounds of 256x384 tiles off/on
 *   400/401/404 NP=256 attention forward: per-head / online / two-pass      402/403 merged NP=256 attention backward off/on
 *   130/131 grouped wgrad: round-3 block order / (problem, split) groups packed onto XCDs (default)
 *   406/407 NP=256 attention backward dK / dV stores: row-per-lane / LDS-transposed full lines (default)
 *   370/371 store-only bf16 GEMM epilogue: fp32 staging (default) / transposed accumulators + wave-private bf16 staging      380/381 split-K of fp32-output GEMMs with <= 64 tiles off/on
 *   390/391/392/393 256 x 256 phased GEMM kernel (csrc/gemm_p8.h) for N % 256 == 0, K % 128 == 0, M % 256 == 0: off / bf16 operands only / also e4m3
 *               operands except fc1 + GELU / every e4m3 GEMM (default)
 *   1000+c start-up skew of every other first-round block of that kernel, c x 1024 cycles (experiment: no effect; c < 400)
 *   1400+s M-splits of the grouped e4m3 weight gradient (0 = automatic: one round of blocks on >= 3/4 of the CUs)
 *   408/409 NP=32 attention backward: dK,dV kernel + dQ kernel / one fused kernel, one wave per (sequence, head) (default)
 *   410/411 fp8 forward, e4m3 copy of the attention output: a quantisation pass over the bf16 output / written by the NP=256 forward kernel (default)
 *   412/413 NP = 32 attention kernels in the e4m3 step (forward: e4m3 copy of the output ; fused backward: e4m3-only dqkv): off / on (default)
 *   396/397/398 phased main loop of the 256 x 384 tile: off / 32-deep k-tiles (default) / 64-deep k-tiles of whole 128-B rows
 *   1500+s M-splits of the grouped bf16 weight gradient (0 = cost model)
 *   2000/2001/2002 persistent GEMM kernels with a tile's epilogue under the next tile's main loop (csrc/gemm_tt.h): off (default) / two teams of 4 waves /
 *               one stream of 4 waves x 512 registers -- built, measured, rejected (DESIGN.md section 3 "Round 6")
 *   2100/2101 fp8 inference / teacher passes keep their residual stream in bf16: off / on (default; d = 384 takes 2111's fused epilogues on the fp32 stream instead)
 *   2110/2111 e4m3 step at d = 384: LayerNorm forward / backward as separate passes / inside the GEMM epilogues (default)
 *   2120/2121 e4m3 GEMMs of at most 1.5 rounds of 256 x 384 tiles: 8-wave kernel (default) / 4-wave kernels, two blocks per CU (measured -0.5 % in the step)
 * The measured-and-rejected GEMM variants of round 2 (64-deep ring stages, ping-pong main loop, register epilogue, start-up
 * skew, phase tracers) are not part of this library: tools/experiments/gemm_r02_variants.hip (ATST_GEMM_VARIANTS=1 build).          */
int atst_tune_gemm_variant(int v);

/* ---- front end: torchaudio MelSpectrogram -> AmplitudeToDB(top_db=80) -> MinMax ---------------------------------
 * replaces audiossl/methods/atst/transform.py:14-33 (`self.mel_feature`) / methods/atstframe/transform.py:16-41.
 * wave: n_clips rows of n_samples f32, row stride wave_ld (0 = n_samples; lets a view be a slice of a longer buffer) ->
 * out [n_clips, n_mels, 1 + n_samples/160] f32, n_mels = 64 or 128 (the reference's `n_mels` parameter, atstframe/transform.py:14-16;
 * the sample rate `sr` only enters through the filterbank table).  window: 1024 taps (Hann(win_length) zero-padded centred);
 * fb_*: compact triangular filterbank (n_mels bands: start bin, length, weights [fb_maxlen, n_mels] = tap-major, zero beyond a band's length).  clipmax: n_clips uint32 scratch. */
int atst_mel_frontend_f32(const float* wave, int n_clips, int n_samples, int wave_ld, int n_mels, int win_length, const float* window,
                          const float* fb_weights, const int* fb_start, const int* fb_len, int fb_maxlen,
                          float* out, uint32_t* clipmax, void* stream);

/* ---- single operators (unit-tested against the oracle through this ABI) ------------------------------------------ */
/* C = A[M,K] * B[N,K]^T with fused epilogue; replaces nn.Linear at audiossl/modules/transformer.py:87-90,109,119.   */
int atst_gemm_nt_bf16(const uint16_t* A, const uint16_t* B, int M, int N, int K, int lda, int ldb, int epi,
                      void* C, int ldc, void* C2, const float* bias, const float* resid, const float* row_scale,
                      int rows_per_seq, const uint16_t* U, const float* table, const uint8_t* rowflag, const float* alt,
                      float* colsum /* EPI_DGELU: optional [N] += column sums of the output */, void* stream);
/* Residual GEMM (N = 384 = the whole row) that also produces the LayerNorm of the new residual row:
 *   x_out = resid + row_scale[row / rows_per_seq] * (A B^T + bias) fp32 [M,384] ; ln_out = bf16(LayerNorm(x_out; gamma, beta, eps 1e-6)) ;
 *   ln_mean / ln_rstd = the row statistics (saved for the backward).  `x = x + drop_path(attn(norm1(x)))` followed by `norm2(x)`,
 *   audiossl/modules/transformer.py:136-150 ; what atst_encoder_fwd launches for proj and fc2 when C = 384.                   */
int atst_gemm_nt_resid_ln_bf16(const uint16_t* A, const uint16_t* B, int M, int K, const float* bias, const float* resid, const float* row_scale,
                               int rows_per_seq, float* x_out, const float* ln_gamma, const float* ln_beta, uint16_t* ln_out, float* ln_mean,
                               float* ln_rstd, void* stream);
/* dgrad GEMM in front of a LayerNorm with that LayerNorm's backward as its epilogue (N = 384 = the whole row):
 *   dy = dY[M,K] Wt[384,K]^T (never written) ; dx = dres + LayerNorm'(dy | x, mean, rstd, gamma) fp32 [M,384] ;
 *   g = bf16(row_scale[row / rows_per_seq] * dx) (or null) ; dgamma += sum_rows dy xhat ; dbeta += sum_rows dy ;
 *   dbias_up += sum_rows g (or null).  Replaces atst_gemm_nt_bf16(EPI_BF16) + atst_layernorm_bwd: autograd of
 *   `x + drop_path(f(norm(x)))`, audiossl/modules/transformer.py:136-150 (Block.forward).                                     */
int atst_gemm_nt_lnbwd_bf16(const uint16_t* dY, const uint16_t* Wt, int M, int K, const float* x, const float* mean, const float* rstd,
                            const float* gamma, const float* dres, float* dx, uint16_t* g, const float* row_scale, int rows_per_seq,
                            float* dgamma, float* dbeta, float* dbias_up, void* stream);
/* Round 6 -- the two row-wise epilogues above for the all-e4m3 step at d = 384 (what atst_encoder_fwd / _bwd launch for proj, fc2 and for the fc1 / qkv
 * dgrads when C = 384 and fp8 is on; Block.forward and its autograd, audiossl/modules/transformer.py:136-150):
 *   atst_gemm_nt_resid_ln_fp8: A8 [M,K], B8 [384,K] e4m3 bytes (K % 64 == 0), acc * *dq / *a_scale (a_scale: device scale of A8, or null = 1) ;
 *     x_out / ln_mean / ln_rstd as atst_gemm_nt_resid_ln_bf16 ; ln_out (bf16) and / or ln_out8 = e4m3(bf16(LayerNorm(x_out)) * *out8_scale) -- the operand of
 *     the next e4m3 GEMM ; out8_amax (an ATST_AMAX_SITE_STRIDE-float site, or null) receives max |bf16 LayerNorm output| ; out8_sat (or null) counts
 *     the elements clipped at +-448.
 *   atst_gemm_nt_lnbwd_q8: the LayerNorm-backward dgrad epilogue with dY / Wt either e4m3 bytes (operands_fp8 = 1: acc * *dq / *dy_scale) or bf16
 *     (operands_fp8 = 0: dq, dy_scale unused), which also writes g8 = e4m3(bf16(row_scale dx) * *g8_scale) (or null) and posts max |row_scale dx| to the
 *     site g8_amax (or null) -- what atst_layernorm_bwd's fp8 form does in the unfused step.  g (bf16) may be null.                                     */
int atst_gemm_nt_resid_ln_fp8(const uint8_t* A8, const uint8_t* B8, int M, int K, const float* dq, const float* a_scale, const float* bias, const float* resid,
                              const float* row_scale, int rows_per_seq, float* x_out, const float* ln_gamma, const float* ln_beta, uint16_t* ln_out,
                              uint8_t* ln_out8, const float* out8_scale, float* out8_amax, uint32_t* out8_sat, float* ln_mean, float* ln_rstd, void* stream);
int atst_gemm_nt_lnbwd_q8(const void* dY, const void* Wt, int operands_fp8, int M, int K, const float* dq, const float* dy_scale, const float* x, const float* mean,
                          const float* rstd, const float* gamma, const float* dres, float* dx, uint16_t* g, uint8_t* g8, const float* g8_scale, float* g8_amax,
                          const float* row_scale, int rows_per_seq, float* dgamma, float* dbeta, float* dbias_up, void* stream);
/* The same GEMM on OCP e4m3 operands (A8 [M,K], B8 [N,K] bytes; N % 384 == 0, K % 64 == 0) with v_mfma_scale_f32_32x32x64_f8f6f4:
 * C = epilogue(dq_mul * (*dq) * A8 B8^T); epilogues BF16 / F32 / BIAS_GELU / RESID.  north_star "fp8 MFMA QKV/MLP GEMMs".      */
int atst_gemm_nt_fp8(const uint8_t* A8, const uint8_t* B8, int M, int N, int K, int lda, int ldb, int epi, void* C, int ldc,
                     void* C2, const float* bias, const float* resid, const float* row_scale, int rows_per_seq,
                     const float* dq, float dq_mul, void* stream);
/* y = e4m3(clamp(scale * x, +-448)) for a bf16 tensor of n elements (n % 8 == 0)                                           */
int atst_quant_fp8_bf16(const uint16_t* x, size_t n, float scale, uint8_t* y, void* stream);
/* per-tensor-scaled e4m3 shadows of n tensors of a flat fp32 buffer: table int32 [n][2] = {element offset, numel};
 * dq[t] = amax_t / 448; amax = device scratch [n]                                                                             */
/* fp8 dgrad support: y = e4m3(clamp(x * *scale)) (y may be NULL: record only) and max |x| posted into the amax SITE (see ATST_AMAX_SLOTS:
 * ATST_AMAX_SITE_STRIDE floats) ; update_scales works on REDUCED values, one float per site: scale[i] = 448 / (margin amax[i]) then
 * amax[i] = 0 ; e4m3 copy of the transposed bf16 weight shadows with the forward copies' per-tensor factors dq.                      */
int atst_quant_fp8_dyn_bf16(const uint16_t* x, size_t n, const float* scale, uint8_t* y, float* amax, void* stream);
int atst_fp8_update_scales(float* amax, float* scale, int n, float margin, void* stream);
int atst_quant_bf16_table_fp8(const uint16_t* p16, const int32_t* table, int n, const float* dq, uint8_t* p8, void* stream);
int atst_quant_weights_fp8(const float* p32, const int32_t* table, int n, uint8_t* p8, float* dq, float* amax, void* stream);
/* dW[N,K] += dY[M,N]^T X[M,K]  (fp32 accumulate); autograd of the same nn.Linear calls.                              */
int atst_gemm_tn_bf16(const uint16_t* dY, const uint16_t* X, int M, int N, int K, int ldy, int ldx, float* dW, int ldw,
                      int m_per_split, void* stream);
/* The same on OCP e4m3 copies of both operands (m-major bytes, as the fp8 forward / dgrad write them), v_mfma_scale_f32_32x32x64_f8f6f4 on
 * transposed LDS reads (ds_read_b64_tr_b8): dW[N,K] += dY8[M,N]^T X8[M,K] / (*scale_y * *scale_x).  N, K multiples of 128, M a multiple of 64.       */
int atst_gemm_tn_fp8(const uint8_t* dY8, const uint8_t* X8, int M, int N, int K, int ldy, int ldx, float* dW, int ldw, const float* scale_y,
                     const float* scale_x, void* stream);
/* Up to 4 independent weight gradients in one launch (the four nn.Linear of a Block, modules/transformer.py:124-150):
 * each dW_i[N_i,K_i] += dY_i^T X_i.  Same result as n calls of atst_gemm_tn_bf16; fewer M-splits, fewer atomics.        */
typedef struct { const uint16_t* dY; const uint16_t* X; float* dW; int M, N, K, ldy, ldx, ldw; } atst_wgrad_t;
int atst_gemm_tn_group_bf16(const atst_wgrad_t* items, int n, void* stream);
/* ... and of their e4m3 form: up to 4 problems that share M (the four Linears of one Block) in one launch.  What a launch pays besides its MFMAs is
 * the fp32 atomics that combine the M-splits; alone every problem needs ~200 / tiles splits to occupy the chip, together the four need 2.       */
typedef struct { const uint8_t* dY8; const uint8_t* X8; float* dW; int N, K, ldy, ldx, ldw; const float* scale_y; const float* scale_x; } atst_wgrad8_t;
int atst_gemm_tn_group_fp8(const atst_wgrad8_t* items, int n, int M, void* stream);
/* nn.LayerNorm(eps=1e-6): audiossl/modules/transformer.py:128,132 ; audio_transformer.py:113                         */
int atst_layernorm_fwd(const float* x, const float* gamma, const float* beta, uint16_t* y, float* mean, float* rstd,
                       int M, int C, void* stream);
/* same LayerNorm with an fp32 output and no saved statistics: the inference API's norm(x_i) of the block taps
 * (audio_transformer.py:235-255, 277-286 ; atstframe/audio_transformer.py:259-281)                                    */
int atst_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, int M, int C, void* stream);
int atst_layernorm_bwd(const uint16_t* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                       const float* dres, float* dx, uint16_t* g, const float* row_scale, int rows_per_seq,
                       float* dgamma, float* dbeta, float* dbias_up, int M, int C, void* stream);
/* Attention.forward + get_attention_mask: audiossl/modules/transformer.py:107-121,152-159                            */
int atst_attention_fwd(const uint16_t* qkv, const int* valid, uint16_t* o, float* lse, int S, int H, int NP, void* stream);
/* 1 when the attention kernels of tile size NP write the e4m3 copy themselves (forward: atst_attention_fwd_fp8's o8 ; backward: atst_attention_bwd_fp8's dqkv8):
 * NP = 256 and, since round 6, NP = 32 (the 1 s local views).  A caller that plans `fp8_lean` = 2 for a pass asks this first (audiossl_amd/engine.py).        */
int atst_attention_fp8_ok(int NP, int H, int backward);
/* the same forward (NP == 256 and, since round 6, NP == 32 -- ask atst_attention_fp8_ok) that also writes the OCP e4m3 copy o8 [S*NP, C] = e4m3(bf16(o) * *scale), clamped to +-448 -- the operand of the
 * e4m3 proj GEMM -- next to o, or INSTEAD of it (o == NULL: inference passes).  amax_site (or NULL): max |bf16(o)| (ATST_AMAX_SITE_STRIDE floats,
 * atomicMax) ; sat (or NULL): number of clipped elements added.  ATST_EINVAL for NP = 64 / 128.                                                 */
int atst_attention_fwd_fp8(const uint16_t* qkv, const int* valid, uint16_t* o, uint8_t* o8, const float* scale, float* amax_site, uint32_t* sat,
                           float* lse, int S, int H, int NP, void* stream);
/* dscratch: optional fp32 [S,H,NP] scratch (rowsum(dO*O)); when given and NP == 256 the merged per-sequence kernel runs */
int atst_attention_bwd(const uint16_t* qkv, const int* valid, const uint16_t* o, const float* lse, const uint16_t* d_o,
                       uint16_t* dqkv, float* dscratch, int S, int H, int NP, void* stream);
/* the same backward (NP == 256 with dscratch, or -- round 6 -- NP == 32, the fused one-wave-per-pair kernel; dscratch unused there) writing dqkv as OCP e4m3 ONLY: dqkv8 [S*NP, 3*C] = e4m3(bf16(dqkv) * *scale), clamped to +-448;
 * max |bf16(dqkv)| is posted into the amax SITE (ATST_AMAX_SITE_STRIDE floats, atomicMax).  The operand of the e4m3 qkv dgrad / weight gradient of
 * the fp8 training step (atst_encoder_t.fp8_wgrad == 2); ATST_EINVAL for NP = 64 / 128.                                                        */
int atst_attention_bwd_fp8(const uint16_t* qkv, const int* valid, const uint16_t* o, const float* lse, const uint16_t* d_o,
                           uint8_t* dqkv8, const float* scale, float* amax_site, float* dscratch, int S, int H, int NP, void* stream);
/* PatchEmbed_v2 gather: audiossl/models/atst/audio_transformer.py:56-75 (bit-exact index map, bf16 values)          */
int atst_patchify_bf16(const float* mel, int S, int width, int NP, int use_cls, uint16_t* out, void* stream);
int atst_gather_rows_bf16(const uint16_t* src, const int* rows, int R, int C, float* dst, void* stream);
int atst_scatter_rows_bf16(const float* src, const int* rows, int R, int C, uint16_t* dst, void* stream);
int atst_colsum_bf16_f32(const uint16_t* x, int M, int N, int ld, float* out, void* stream);
int atst_cast_bf16(const float* x, size_t n, uint16_t* y, void* stream);
/* split-bf16 operand [R,3K]: [hi|lo|hi] (b_layout 0, activations) or [hi|hi|lo] (b_layout 1, weights): one bf16 GEMM over
 * 3K then yields x W^T to ~2^-16; used for the Linear in front of BatchNorm+ReLU (byol.py:13-16) */
int atst_split3_bf16(const float* x, int R, int K, int b_layout, uint16_t* y, void* stream);
int atst_transpose_bf16_2d(const uint16_t* src, int rows, int cols, uint16_t* dst, void* stream);
/* Every 2-D tensor of a flat bf16 parameter buffer transposed in one launch ([out,in] -> [in,out], the dgrad operand).
 * table (device, int32 [n][4]) = {element offset, rows, cols, first 64x64 tile index}; total_tiles = sum of tiles.     */
int atst_transpose_bf16_batch(const uint16_t* src_base, uint16_t* dst_base, const int32_t* table, int n, int total_tiles,
                              void* stream);

/* build_mlp's BatchNorm1d(train)+ReLU: audiossl/models/atst/byol.py:13-16                                            */
int atst_bn_stats_f32(const float* h, int R, int N, float* mean, float* m2, float* scratch /* [32 * N] row-block partials: fixed-order (run-to-run reproducible) reduction */, void* stream);
/* the rest of BatchNorm1d(train) on the (global) batch statistics, one launch: rstd = rsqrt(M2 / count + eps); running_mean / running_var
 * (unbiased: M2 / max(count - 1, 1)) momentum update in place (NULL = leave them); *num_batches_tracked += 1 (NULL = skip).  count_dev != NULL
 * overrides `count` with a device scalar (the cross-rank row count, never read back).                                                */
int atst_bn_finish_f32(const float* mean, const float* m2, float count, const float* count_dev, float momentum, float eps,
                       float* running_mean, float* running_var, int64_t* num_batches_tracked, float* rstd, int n, void* stream);
int atst_bn_apply_relu_bf16(const float* h, const float* mean, const float* rstd, const float* gamma, const float* beta,
                            int R, int N, uint16_t* y, void* stream);
/* same, output as split-bf16 operand [R, 3N] = [hi | lo | hi] for the following Linear */
int atst_bn_apply_relu_split3_bf16(const float* h, const float* mean, const float* rstd, const float* gamma,
                                   const float* beta, int R, int N, uint16_t* y, void* stream);
int atst_bn_relu_bwd_sums(const float* dy, const float* h, const float* mean, const float* rstd, const float* gamma,
                          const float* beta, int R, int N, float* sum_dy, float* sum_dy_xhat,
                          float* scratch /* [2 * 32 * N] row-block partials, fixed-order reduction */, void* stream);
int atst_bn_bwd_dx_bf16(const float* dy, const float* h, const float* mean, const float* rstd, const float* gamma,
                        const float* beta, const float* sum_dy, const float* sum_dy_xhat, float inv_count, int R, int N,
                        uint16_t* dh, void* stream);
/* the same with an fp32 dh (parity mode: the head backward then runs on split-bf16 operands made from it) */
int atst_bn_bwd_dx_f32(const float* dy, const float* h, const float* mean, const float* rstd, const float* gamma,
                       const float* beta, const float* sum_dy, const float* sum_dy_xhat, float inv_count, int R, int N,
                       float* dh, void* stream);
/* ByolLoss.forward + its backward: audiossl/models/atst/byol.py:24-78.  acc[0] = sum of pair cosines,
 * loss = 2 - 2*acc/((2*ncrops-2)*B); stats [4,256] = student/teacher column sums and square sums of normalised rows.
 * ncrops == -1: the asymmetric ATST-Frame loss (methods/atstframe/byol.py:83-84, model.py:73-76): student [B,256] (view 1)
 * against teacher [B,256] (view 0), loss = 2 - 2*acc/B.                                                                   */
int atst_byol_loss_f32(const float* student, const float* teacher, int B, int ncrops, int D, float* acc, float* dstudent,
                       float* stats, void* stream);

/* HF-AdamW + EMA teacher + bf16 shadow refresh over flat buffers: audiossl/methods/atst/model.py:44-51,
 * audiossl/models/atst/atst.py:29-34.  chunk_flags: one byte per 256 elements (bit0 weight-decay, bit1 update, bit2 EMA). */
int atst_adamw_ema_step(float* p, const float* g, float* m, float* v, float* t, uint16_t* p_bf16, uint16_t* t_bf16,
                        const uint8_t* chunk_flags, size_t n, size_t n_teacher, double lr, double wd, double beta1,
                        double beta2, double eps, double step_size, double ema_m, double grad_scale, void* stream);

/* ---- whole-encoder engine: AST.forward / FrameAST.forward and their autograd ---------------------------------------
 * replaces audiossl/models/atst/audio_transformer.py:188-221 and methods/atstframe/audio_transformer.py:183-207
 * (12 x Block.forward, audiossl/modules/transformer.py:136-150).  Offsets index the flat parameter buffers. */
typedef struct {
  int64_t ln1_w, ln1_b, qkv_w, proj_w, proj_b, ln2_w, ln2_b, fc1_w, fc1_b, fc2_w, fc2_b;
} atst_layer_off_t;

typedef struct {
  int64_t mask_embed, cls_token, pos_embed, patch_w, patch_b, norm_w, norm_b;
  atst_layer_off_t layer[ATST_MAX_DEPTH];
} atst_enc_off_t;

typedef struct {
  /* geometry */
  int S, NP, n_tok, width, C, H, depth, use_cls, train;
  /* parameters: fp32 masters, bf16 shadows ([out,in]) and bf16 transposed shadows ([in,out], backward only) */
  const float* p32; const uint16_t* p16; const uint16_t* p16t;
  float* g32;                         /* flat fp32 gradient buffer (backward accumulates into it) */
  atst_enc_off_t off;
  /* per-call inputs */
  const float* mel;                   /* [S,1,64,width] */
  const int* valid;                   /* [S] valid key tokens (patch_length + use_cls) */
  const uint8_t* rowflag;             /* [S*NP] 1 = substitute mask token (ATST-Frame student) or NULL */
  const float* dp_scale;              /* [depth,2,S] DropPath keep/keep_prob factors or NULL */
  void* ws; size_t ws_bytes;          /* activation workspace, atst_encoder_ws_bytes() */
  /* optional block taps (inference API, get_intermediate_layers): when tap != NULL the fp32 residual stream after
   * block i is copied to tap[(i - tap_first) * S*NP*C] for every i >= tap_first; works with train = 0 workspaces */
  float* tap; int tap_first;
  /* fp8 forward (ATST-base recipe, BASELINE.json configs[4]): when fp8 != 0 the four Linear layers of every block run on OCP
   * e4m3 operands (MX-scaled MFMA, unit block scales): p8 = e4m3 weight shadow at the SAME element offsets as p32
   * (atst_quant_weights_fp8), w_dq = [depth][4] per-tensor factors (qkv, proj, fc1, fc2) that undo the weight scales.
   * Activations are re-quantised per GEMM with fixed scales; saved tensors and the whole backward stay bf16.              */
  const uint8_t* p8; const float* w_dq; int fp8;
  /* patch geometry: one patch row of patch_h mel bands x patch_w frames (the reference's --patch_h / --patch_w with
   * patch_h = n_mels, audiossl/methods/atstframe/train.py:15,50-51); 0 = the shipped 64 x 4.  mel is [S,1,patch_h,width],
   * n_tok = width / patch_w, the patch-embedding weight is [C, patch_h * patch_w] (a multiple of 256, <= 1024).          */
  int patch_h, patch_w;
  /* fp8 dgrad (C = 768, fp8 != 0): p8t = e4m3 copy of the TRANSPOSED weight shadows (same offsets / per-tensor factors as p8);
   * g8_scale = [depth][4] device floats, g8_amax = [depth][4] amax SITES (ATST_AMAX_SITE_STRIDE floats each), one per gradient operand
   * (g into fc2, du into fc1, g2 into proj, dqkv into qkv):
   * the operand is quantised with g8_scale (delayed scaling: derived from the previous step's amax by atst_fp8_update_scales) and this
   * step's max |x| is recorded in g8_amax.  fp8_bwd: 0 = bf16 backward, 1 = bf16 backward + amax recording (first step), 2 = fp8 dgrad.
   * Weight gradients use the bf16 operands unless fp8_wgrad is set (below).                                                  */
  const uint8_t* p8t; const float* g8_scale; float* g8_amax; int fp8_bwd;
  /* rows between consecutive sequences in every [tokens, *] tensor of the pass (tokens, activations, gradients, atst_encoder_out):
   * 0 = NP.  n_tok + use_cls <= row_stride < NP packs the sequences (NP < 256 only): 1 s views are 26 tokens in tiles of 32, and the
   * GEMM / LayerNorm / weight-gradient kernels then run over S * row_stride rows.  The workspace is sized for NP either way.      */
  int row_stride;
  /* fp8 forward: device counter (or NULL) that every activation-quantising kernel of the pass adds its number of CLIPPED elements to
   * (values beyond +-448 / scale of their site; scales: f8_act_scale below, or the constants 8 / 8 / 8 / 4 when that is NULL); the
   * caller clears and reads it (AtstEngine.fp8_saturation()).  A non-zero count means a site's scale lagged behind the run.          */
  uint32_t* f8_sat;
  /* fp8 forward, running (delayed) activation scales: f8_act_scale [depth][4] device floats, f8_act_amax [depth][4] amax SITES
   * (ATST_AMAX_SITE_STRIDE floats each); site k of block i = 0: LayerNorm-1 output (qkv GEMM),
   * 1: attention output (proj), 2: LayerNorm-2 output (fc1), 3: GELU output (fc2).  f8_act_scale (or NULL = the constants 8, 8, 8, 4) is what
   * the producing kernels quantise with and the consuming GEMMs divide by; f8_act_amax (or NULL) receives max |x| of every site of this pass
   * (atomicMax) -- the caller turns it into the next step's scales (atst_fp8_update_scales; AtstEngine: 448 / (2 * max over 16 steps)).    */
  const float* f8_act_scale; float* f8_act_amax;
  /* fp8 weight gradients (ABI 120): fp8_wgrad != 0 with fp8_bwd == 2 runs the fc1 / fc2 / proj weight gradients on the e4m3 gradient operands of
   * the dgrad GEMMs and on the e4m3 activation copies the forward KEPT per layer (a training workspace with fp8 carves them).  f8_act_scale_bwd
   * [depth][4]: the activation scales the forward of THIS step quantised with (a snapshot: f8_act_scale itself is advanced between forward and backward).
   * fp8_wgrad == 2 (NP = 256 passes): the qkv Linear as well -- the attention backward writes dqkv as e4m3 ONLY (gradient site 3), the qkv weight
   * gradient and the qkv dgrad read that copy: all 12 GEMMs of a block on e4m3 operands.  fp8_wgrad == 3: as 1, and site 3's amax is recorded (the
   * step that gives the site its first scale; fp8_wgrad == 2 while fp8_bwd == 1 does the same).                                              */
  int fp8_wgrad; const float* f8_act_scale_bwd;
  /* fp8 training forward, bf16 copies without a reader: with e4m3 weight gradients the backward reads the e4m3 activation copies, so the forward need
   * not write the bf16 ones.  fp8_lean >= 1: LayerNorm-2 output and GELU output are written as e4m3 only (fc1 / fc2 weight gradients must be e4m3:
   * the backward of this pass needs fp8_bwd == 2, fp8_wgrad != 0, M % 64 == 0); >= 2: LayerNorm-1 output too (the qkv weight gradient must be e4m3:
   * fp8_wgrad == 2, NP == 256).  A backward that cannot honour it returns ATST_EINVAL.  Inference passes (train == 0) never write those copies.       */
  int fp8_lean;
} atst_encoder_t;

size_t atst_encoder_ws_bytes(int S, int NP, int C, int H, int depth, int train, int fp8 /* = atst_encoder_t.fp8: also carve the e4m3 operand copies */);
size_t atst_encoder_ws_bytes_geo(int S, int NP, int C, int H, int depth, int train, int fp8, int patch_h, int patch_w);   /* same, for a non-default patch geometry */
/* forward: leaves LN(final) of every token as bf16 [S*row_stride, C] at atst_encoder_out(); */
int atst_encoder_fwd(const atst_encoder_t* e, void* stream);
const uint16_t* atst_encoder_out(const atst_encoder_t* e);
/* backward: d(LN(final)) bf16 [S*NP, C] must have been written to atst_encoder_dout() (zero for unused rows) */
uint16_t* atst_encoder_dout(const atst_encoder_t* e);
int atst_encoder_bwd(const atst_encoder_t* e, void* stream);
/* The same backward in two calls: part 0 = final LayerNorm + blocks [split, depth), part 1 = blocks [0, split) + token
 * stage (the caller may start the gradient all-reduce of the upper blocks in between).                                */
int atst_encoder_bwd_part(const atst_encoder_t* e, int part, int split, void* stream);
/* General form: backward over blocks [lo, hi) (descending).  hi == depth also runs the final LayerNorm backward, lo == 0
 * also the token stage.  Consecutive calls hi..lo must tile [0, depth) from the top; between two calls the parameter
 * gradients of the blocks already walked are final, so the caller can reduce them across ranks bucket by bucket.      */
int atst_encoder_bwd_range(const atst_encoder_t* e, int lo, int hi, void* stream);
/* ---- high-precision twin (parity mode; csrc/engine_hp.hip) ------------------------------------------------------------------
 * The same encoder forward / backward with fp32 activations and gradients: every Linear runs on the production MFMA GEMMs with
 * split-bf16 operands ([hi|lo|hi] x [hi|hi|lo], ~2^-17), attention / LayerNorm / GELU in plain fp32 kernels.  Same
 * atst_encoder_t (train is implied, fp8 must be 0, p16 / p16t are not read), its own workspace size; output / upstream-gradient
 * rows are fp32 [S*NP, C].  ~30x slower than the bf16 path: for pinning gradients to the reference at <= 2e-3 on small shapes. */
size_t atst_encoder_hp_ws_bytes(int S, int NP, int C, int H, int depth, int patch_h, int patch_w);
int atst_encoder_hp_fwd(const atst_encoder_t* e, void* stream);
int atst_encoder_hp_bwd(const atst_encoder_t* e, void* stream);
const float* atst_encoder_hp_out(const atst_encoder_t* e);
float* atst_encoder_hp_dout(const atst_encoder_t* e);
const float* atst_encoder_hp_block_out(const atst_encoder_t* e, int i);
/* block-level activation taps for tests: fp32 residual stream after block i (i in [0,depth)), train=1 only */
const float* atst_encoder_block_out(const atst_encoder_t* e, int i);
const float* atst_encoder_tokens(const atst_encoder_t* e);

/* ---- batched spectrogram augmentations, random draws supplied by the caller --------------------- */
/* RandomResizeCrop.forward, audiossl/transforms/byol_a.py:33-49: in/out [B,H,W] fp32; params [B][4] = (i, j, h, w) of
 * get_params (byol_a.py:24-31); CH x CW = int(H * virtual_crop_scale[0]) x int(W * virtual_crop_scale[1]).            */
int atst_rrc_bicubic_f32(const float* in, float* out, const int32_t* params, int B, int H, int W, int CH, int CW, void* stream);
/* log_mixup_exp(x, z, 1 - a), byol_a.py:61-83 as called by Mixup.forward :104: out = log((1-a) e^x + a e^z + eps) on
 * the window of min(W, Wz) frames starting at xstart[b] in x and zstart[b] in z = bank[zidx[b]] (one of the two is 0);
 * frames of x outside the window become log(e^x + eps).  x/out [B,H,W], bank [n,H,Wz], all fp32.                       */
int atst_log_mixup_exp_f32(const float* x, const float* bank, const int32_t* zidx, const int32_t* zstart, const int32_t* xstart,
                           const float* alpha, float* out, int B, int H, int W, int Wz, void* stream);

/* ---- in-library kernel timing (HIP events on the launch stream); used by bench.py for the roofline object ---------- */
int atst_profile_enable(int on);      /* 0 = off ; n >= 1: HIP events around every n-th launch of each kernel kind     */
int atst_profile_kinds(void);
const char* atst_profile_name(int kind);
int atst_profile_collect(double* ms, double* work, double* bytes, long long* launches);   /* arrays of atst_profile_kinds(); work = FLOPs for MFMA kinds, bytes = algorithmic HBM bytes */

#ifdef __cplusplus
}
#endif
#endif /* ATST_HIP_H */
