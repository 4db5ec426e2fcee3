// extern "C" surface of libatst_hip.so (declared in include/atst_hip.h): thin, allocation-free wrappers that turn raw
// pointers + sizes + a stream into kernel launches.
#include "common.h"
#include "kernels.h"
#include "../../include/atst_hip.h"

#define ST(s) reinterpret_cast<hipStream_t>(s)
#define BF(p) reinterpret_cast<bf16*>(p)
#define CBF(p) reinterpret_cast<const bf16*>(p)

extern "C" {

int atst_version(void) { return ATST_ABI_VERSION; }
extern int g_tn_group_splits;
int atst_tune_gemm_variant(int v) { if (v >= 1500 && v < 1600) g_tn_group_splits = v - 1500; else if (v >= 1400 && v < 1500) g_tn8_splits = v - 1400; else if (v >= 400 && v < 1000) atst_attn_set_variant(v - 400); else atst_gemm_nt_set_variant(v); return 0; }

int atst_mel_frontend_f32(const float* wave, int n_clips, int n_samples, int wave_ld, int n_mels, int win_length, const float* window,
                          const float* fb_weights, const int* fb_start, const int* fb_len, int fb_maxlen,
                          float* out, uint32_t* clipmax, void* stream) {
  return atst_mel_frontend(wave, n_clips, n_samples, wave_ld, n_mels, win_length, window, fb_weights, fb_start, fb_len, fb_maxlen, out,
                           clipmax, ST(stream));
}

int atst_gemm_nt_bf16(const uint16_t* A, const uint16_t* B, int M, int N, int K, int lda, int ldb, int epi,
                      void* C, int ldc, void* C2, const float* bias, const float* resid, const float* row_scale,
                      int rows_per_seq, const uint16_t* U, const float* table, const uint8_t* rowflag, const float* alt,
                      float* colsum, void* stream) {
  GemmArgs a{};
  a.A = CBF(A); a.B = CBF(B); a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.epi = epi; a.C = C; a.ldc = ldc;
  a.C2 = C2; a.bias = bias; a.resid = resid; a.row_scale = row_scale; a.rows_per_seq = rows_per_seq > 0 ? rows_per_seq : 1;
  a.U = CBF(U); a.table = table; a.rowflag = rowflag; a.alt = alt; a.colsum = colsum;
  if ((epi == EPI_BIAS_GELU && (!bias || !C2)) || (epi == EPI_RESID && (!bias || !resid)) || (epi == EPI_DGELU && !U) ||
      (epi == EPI_PATCH && (!table || !bias)))
    return ATST_EINVAL;
  return atst_gemm_nt(a, ST(stream));
}

int atst_gemm_nt_resid_ln_bf16(const uint16_t* A, const uint16_t* B, int M, int K, const float* bias, const float* resid, const float* row_scale,
                               int rows_per_seq, float* x_out, const float* ln_gamma, const float* ln_beta, uint16_t* ln_out, float* ln_mean,
                               float* ln_rstd, void* stream) {
  if (!A || !B || !bias || !resid || !x_out || !ln_gamma || !ln_beta || !ln_out || !ln_mean || !ln_rstd) return ATST_EINVAL;
  GemmArgs a{};
  a.A = CBF(A); a.B = CBF(B); a.M = M; a.N = 384; a.K = K; a.lda = K; a.ldb = K; a.epi = EPI_RESID; a.C = x_out; a.ldc = 384;
  a.bias = bias; a.resid = resid; a.row_scale = row_scale; a.rows_per_seq = rows_per_seq > 0 ? rows_per_seq : 1;
  a.ln_gamma = ln_gamma; a.ln_beta = ln_beta; a.ln_out = BF(ln_out); a.ln_mean = ln_mean; a.ln_rstd = ln_rstd;
  return atst_gemm_nt(a, ST(stream));
}

int atst_gemm_nt_lnbwd_bf16(const uint16_t* dY, const uint16_t* Wt, int M, int K, const float* x, const float* mean, const float* rstd,
                            const float* gamma, const float* dres, float* dx, uint16_t* g, const float* row_scale, int rows_per_seq,
                            float* dgamma, float* dbeta, float* dbias_up, void* stream) {
  if (!dY || !Wt || !x || !mean || !rstd || !gamma || !dx || !dgamma || !dbeta) return ATST_EINVAL;
  GemmArgs a{};
  a.A = CBF(dY); a.B = CBF(Wt); a.M = M; a.N = 384; a.K = K; a.lda = K; a.ldb = K; a.epi = EPI_LNBWD; a.C = dx; a.ldc = 384;
  a.resid = dres; a.row_scale = row_scale; a.rows_per_seq = rows_per_seq > 0 ? rows_per_seq : 1; a.ln_gamma = gamma;
  a.ln_mean = const_cast<float*>(mean); a.ln_rstd = const_cast<float*>(rstd); a.lnb_x = x; a.lnb_g = BF(g);
  a.lnb_dgamma = dgamma; a.lnb_dbeta = dbeta; a.lnb_dbias_up = dbias_up;
  return atst_gemm_nt(a, ST(stream));
}

int atst_gemm_nt_fp8(const uint8_t* A8, const uint8_t* B8, int M, int N, int K, int lda, int ldb, int epi, void* C, int ldc,
                     void* C2, const float* bias, const float* resid, const float* row_scale, int rows_per_seq,
                     const float* dq, float dq_mul, void* stream) {
  GemmArgs a{};
  a.A = CBF(A8); a.B = CBF(B8); a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.epi = epi; a.C = C; a.ldc = ldc; a.C2 = C2;
  a.bias = bias; a.resid = resid; a.row_scale = row_scale; a.rows_per_seq = rows_per_seq > 0 ? rows_per_seq : 1;
  a.fp8 = 1; a.dq = dq; a.dq_mul = dq_mul;
  if ((epi == EPI_BIAS_GELU && (!bias || !C2)) || (epi == EPI_RESID && (!bias || !resid)) || epi == EPI_DGELU || epi == EPI_PATCH) return ATST_EINVAL;
  return atst_gemm_nt(a, ST(stream));
}
// round 6 (d = 384, all-e4m3 step): the two row-wise epilogues on e4m3 operands, with the e4m3 copy of the bf16 row they produce
int atst_gemm_nt_resid_ln_fp8(const uint8_t* A8, const uint8_t* B8, int M, int K, const float* dq, const float* a_scale, const float* bias, const float* resid,
                              const float* row_scale, int rows_per_seq, float* x_out, const float* ln_gamma, const float* ln_beta, uint16_t* ln_out,
                              uint8_t* ln_out8, const float* out8_scale, float* out8_amax, uint32_t* out8_sat, float* ln_mean, float* ln_rstd, void* stream) {
  if (!A8 || !B8 || !bias || !resid || !x_out || !ln_gamma || !ln_beta || (!ln_out && !ln_out8) || !ln_mean || !ln_rstd || (ln_out8 && !out8_scale)) return ATST_EINVAL;
  GemmArgs a{};
  a.A = CBF(A8); a.B = CBF(B8); a.M = M; a.N = 384; a.K = K; a.lda = K; a.ldb = K; a.epi = EPI_RESID; a.C = x_out; a.ldc = 384;
  a.bias = bias; a.resid = resid; a.row_scale = row_scale; a.rows_per_seq = rows_per_seq > 0 ? rows_per_seq : 1;
  a.fp8 = 1; a.dq = dq; a.dq_mul = 1.0f; a.dq_div = a_scale;
  a.ln_gamma = ln_gamma; a.ln_beta = ln_beta; a.ln_out = BF(ln_out); a.ln_mean = ln_mean; a.ln_rstd = ln_rstd;
  a.q8 = ln_out8; a.q8_scale_ptr = out8_scale; a.q8_amax = ln_out8 ? out8_amax : nullptr; a.q8_sat = out8_sat;
  return atst_gemm_nt(a, ST(stream));
}
int atst_gemm_nt_lnbwd_q8(const void* dY, const void* Wt, int operands_fp8, int M, int K, const float* dq, const float* dy_scale, const float* x, const float* mean,
                          const float* rstd, const float* gamma, const float* dres, float* dx, uint16_t* g, uint8_t* g8, const float* g8_scale, float* g8_amax,
                          const float* row_scale, int rows_per_seq, float* dgamma, float* dbeta, float* dbias_up, void* stream) {
  if (!dY || !Wt || !x || !mean || !rstd || !gamma || !dx || !dgamma || !dbeta || (g8 && !g8_scale) || (operands_fp8 && !dq)) return ATST_EINVAL;
  GemmArgs a{};
  a.A = reinterpret_cast<const bf16*>(dY); a.B = reinterpret_cast<const bf16*>(Wt); a.M = M; a.N = 384; a.K = K; a.lda = K; a.ldb = K; a.epi = EPI_LNBWD;
  a.C = dx; a.ldc = 384;
  if (operands_fp8) { a.fp8 = 1; a.dq = dq; a.dq_mul = 1.0f; a.dq_div = dy_scale; }
  a.resid = dres; a.row_scale = row_scale; a.rows_per_seq = rows_per_seq > 0 ? rows_per_seq : 1; a.ln_gamma = gamma;
  a.ln_mean = const_cast<float*>(mean); a.ln_rstd = const_cast<float*>(rstd); a.lnb_x = x; a.lnb_g = BF(g);
  a.lnb_dgamma = dgamma; a.lnb_dbeta = dbeta; a.lnb_dbias_up = dbias_up;
  a.q8 = g8; a.q8_scale_ptr = g8_scale; a.q8_amax = g8_amax;
  return atst_gemm_nt(a, ST(stream));
}
int atst_quant_fp8_bf16(const uint16_t* x, size_t n, float scale, uint8_t* y, void* stream) { return atst_quant_fp8(CBF(x), n, scale, y, ST(stream)); }
int atst_quant_fp8_dyn_bf16(const uint16_t* x, size_t n, const float* scale, uint8_t* y, float* amax, void* stream) {
  return atst_quant_fp8_dyn(CBF(x), n, scale, y, amax, ST(stream));
}
int atst_fp8_update_scales(float* amax, float* scale, int n, float margin, void* stream) { return ::atst_fp8_update_scales(amax, scale, n, margin, ST(stream)); }
int atst_quant_bf16_table_fp8(const uint16_t* p16, const int32_t* table, int n, const float* dq, uint8_t* p8, void* stream) {
  return ::atst_quant_bf16_table_fp8(CBF(p16), table, n, dq, p8, ST(stream));
}
int atst_quant_weights_fp8(const float* p32, const int32_t* table, int n, uint8_t* p8, float* dq, float* amax, void* stream) {
  return ::atst_quant_weights_fp8(p32, table, n, p8, dq, amax, ST(stream));
}

int atst_gemm_tn_bf16(const uint16_t* dY, const uint16_t* X, int M, int N, int K, int ldy, int ldx, float* dW, int ldw,
                      int m_per_split, void* stream) {
  WgradArgs a{};
  a.dY = CBF(dY); a.X = CBF(X); a.M = M; a.N = N; a.K = K; a.ldy = ldy; a.ldx = ldx; a.dW = dW; a.ldw = ldw;
  a.m_per_split = m_per_split;
  return atst_gemm_tn(a, ST(stream));
}

int atst_gemm_tn_fp8(const uint8_t* dY8, const uint8_t* X8, int M, int N, int K, int ldy, int ldx, float* dW, int ldw, const float* scale_y,
                     const float* scale_x, void* stream) {
  return atst_gemm_tn8(dY8, X8, M, N, K, ldy, ldx, dW, ldw, scale_y, scale_x, ST(stream));
}

int atst_gemm_tn_group_bf16(const atst_wgrad_t* items, int n, void* stream) {
  if (!items || n < 1 || n > ATST_WGRAD_GROUP_MAX) return ATST_EINVAL;
  WgradArgs a[ATST_WGRAD_GROUP_MAX] = {};
  for (int i = 0; i < n; ++i) {
    a[i].dY = CBF(items[i].dY); a[i].X = CBF(items[i].X); a[i].dW = items[i].dW;
    a[i].M = items[i].M; a[i].N = items[i].N; a[i].K = items[i].K;
    a[i].ldy = items[i].ldy; a[i].ldx = items[i].ldx; a[i].ldw = items[i].ldw; a[i].m_per_split = 0;
  }
  return atst_gemm_tn_group(a, n, ST(stream));
}

int atst_gemm_tn_group_fp8(const atst_wgrad8_t* items, int n, int M, void* stream) {
  if (!items || n < 1 || n > 4) return ATST_EINVAL;
  Wgrad8Item a[4] = {};
  for (int i = 0; i < n; ++i)
    a[i] = Wgrad8Item{items[i].dY8, items[i].X8, items[i].N, items[i].K, items[i].ldy, items[i].ldx, items[i].dW, items[i].ldw, items[i].scale_y, items[i].scale_x};
  return atst_gemm_tn8_group(a, n, M, ST(stream));
}

int atst_layernorm_fwd(const float* x, const float* gamma, const float* beta, uint16_t* y, float* mean, float* rstd,
                       int M, int C, void* stream) {
  return atst_ln_fwd(x, gamma, beta, BF(y), mean, rstd, M, C, ST(stream));
}

int atst_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, int M, int C, void* stream) {
  return atst_ln_fwd_f32(x, gamma, beta, y, M, C, ST(stream));
}

int atst_layernorm_bwd(const uint16_t* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                       const float* dres, float* dx, uint16_t* g, const float* row_scale, int rows_per_seq,
                       float* dgamma, float* dbeta, float* dbias_up, int M, int C, void* stream) {
  LnBwdArgs a{};
  a.dy = CBF(dy); a.x = x; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.dres = dres; a.dx = dx; a.g = BF(g);
  a.row_scale = row_scale; a.rows_per_seq = rows_per_seq > 0 ? rows_per_seq : 1; a.dgamma = dgamma; a.dbeta = dbeta;
  a.dbias_up = dbias_up; a.M = M; a.C = C;
  return atst_ln_bwd(a, ST(stream));
}

int atst_attention_fp8_ok(int NP, int H, int backward) { return backward ? (atst_attn_bwd_q8_ok(NP) ? 1 : 0) : (atst_attn_fwd_q8_ok(NP, H) ? 1 : 0); }
int atst_attention_fwd(const uint16_t* qkv, const int* valid, uint16_t* o, float* lse, int S, int H, int NP, void* stream) {
  AttnArgs a{};
  a.qkv = CBF(qkv); a.valid = valid; a.o = BF(o); a.lse = lse; a.S = S; a.H = H; a.NP = NP;
  return atst_attn_fwd(a, ST(stream));
}

int atst_attention_bwd(const uint16_t* qkv, const int* valid, const uint16_t* o, const float* lse, const uint16_t* d_o,
                       uint16_t* dqkv, float* dscratch, int S, int H, int NP, void* stream) {
  AttnArgs a{};
  a.qkv = CBF(qkv); a.valid = valid; a.o = BF(const_cast<uint16_t*>(o)); a.lse = const_cast<float*>(lse);
  a.d_o = CBF(d_o); a.dqkv = BF(dqkv); a.dscratch = dscratch; a.S = S; a.H = H; a.NP = NP;
  return atst_attn_bwd(a, ST(stream));
}

int atst_attention_fwd_fp8(const uint16_t* qkv, const int* valid, uint16_t* o, uint8_t* o8, const float* scale, float* amax_site, uint32_t* sat,
                           float* lse, int S, int H, int NP, void* stream) {
  if (!o8 || !scale) return ATST_EINVAL;
  AttnArgs a{};
  a.qkv = CBF(qkv); a.valid = valid; a.o = BF(o); a.lse = lse; a.S = S; a.H = H; a.NP = NP;
  a.o8 = o8; a.o8_scale = scale; a.o8_amax = amax_site; a.o8_sat = sat;
  return atst_attn_fwd(a, ST(stream));
}

int atst_attention_bwd_fp8(const uint16_t* qkv, const int* valid, const uint16_t* o, const float* lse, const uint16_t* d_o,
                           uint8_t* dqkv8, const float* scale, float* amax_site, float* dscratch, int S, int H, int NP, void* stream) {
  if (!dqkv8 || !scale || !amax_site) return ATST_EINVAL;
  AttnArgs a{};
  a.qkv = CBF(qkv); a.valid = valid; a.o = BF(const_cast<uint16_t*>(o)); a.lse = const_cast<float*>(lse);
  a.d_o = CBF(d_o); a.dqkv8 = dqkv8; a.q8_scale = scale; a.q8_amax = amax_site; a.dscratch = dscratch; a.S = S; a.H = H; a.NP = NP;
  return atst_attn_bwd(a, ST(stream));
}

int atst_patchify_bf16(const float* mel, int S, int width, int NP, int use_cls, uint16_t* out, void* stream) {
  return atst_patchify(mel, S, width, NP, use_cls, BF(out), ST(stream));
}
int atst_gather_rows_bf16(const uint16_t* src, const int* rows, int R, int C, float* dst, void* stream) {
  return atst_gather_rows(CBF(src), rows, R, C, dst, ST(stream));
}
int atst_scatter_rows_bf16(const float* src, const int* rows, int R, int C, uint16_t* dst, void* stream) {
  return atst_scatter_rows(src, rows, R, C, BF(dst), ST(stream));
}
int atst_colsum_bf16_f32(const uint16_t* x, int M, int N, int ld, float* out, void* stream) {
  return atst_colsum_bf16(CBF(x), M, N, ld, out, ST(stream));
}
int atst_cast_bf16(const float* x, size_t n, uint16_t* y, void* stream) { return atst_cast_f32_bf16(x, n, BF(y), ST(stream)); }
int atst_split3_bf16(const float* x, int R, int K, int b_layout, uint16_t* y, void* stream) {
  return atst_split3(x, R, K, b_layout, BF(y), ST(stream));
}
int atst_transpose_bf16_2d(const uint16_t* src, int rows, int cols, uint16_t* dst, void* stream) {
  return atst_transpose_bf16(CBF(src), rows, cols, BF(dst), ST(stream));
}

int atst_transpose_bf16_batch(const uint16_t* src_base, uint16_t* dst_base, const int32_t* table, int n, int total_tiles, void* stream) {
  if (!src_base || !dst_base || !table) return ATST_EINVAL;
  return atst_transpose_bf16_batch(CBF(src_base), BF(dst_base), table, n, total_tiles, ST(stream));
}

int atst_rrc_bicubic_f32(const float* in, float* out, const int32_t* params, int B, int H, int W, int CH, int CW, void* stream) {
  if (!in || !out || !params) return ATST_EINVAL;
  return atst_rrc_bicubic(in, out, params, B, H, W, CH, CW, ST(stream));
}
int atst_log_mixup_exp_f32(const float* x, const float* bank, const int32_t* zidx, const int32_t* zstart, const int32_t* xstart,
                           const float* alpha, float* out, int B, int H, int W, int Wz, void* stream) {
  if (!x || !bank || !zidx || !zstart || !xstart || !alpha || !out) return ATST_EINVAL;
  return atst_log_mixup_exp(x, bank, zidx, zstart, xstart, alpha, out, B, H, W, Wz, ST(stream));
}

int atst_bn_stats_f32(const float* h, int R, int N, float* mean, float* m2, float* scratch, void* stream) {
  return atst_bn_stats(h, R, N, mean, m2, scratch, ST(stream));
}
int atst_bn_finish_f32(const float* mean, const float* m2, float count, const float* count_dev, float momentum, float eps,
                       float* running_mean, float* running_var, int64_t* num_batches_tracked, float* rstd, int n, void* stream) {
  return atst_bn_finish(mean, m2, count, count_dev, momentum, eps, running_mean, running_var, reinterpret_cast<long long*>(num_batches_tracked), rstd, n, ST(stream));
}
int atst_bn_apply_relu_bf16(const float* h, const float* mean, const float* rstd, const float* gamma, const float* beta,
                            int R, int N, uint16_t* y, void* stream) {
  return atst_bn_apply_relu(h, mean, rstd, gamma, beta, R, N, BF(y), ST(stream));
}
int atst_bn_apply_relu_split3_bf16(const float* h, const float* mean, const float* rstd, const float* gamma,
                                   const float* beta, int R, int N, uint16_t* y, void* stream) {
  return atst_bn_apply_relu_split3(h, mean, rstd, gamma, beta, R, N, BF(y), ST(stream));
}
int atst_bn_relu_bwd_sums(const float* dy, const float* h, const float* mean, const float* rstd, const float* gamma,
                          const float* beta, int R, int N, float* sum_dy, float* sum_dy_xhat, float* scratch, void* stream) {
  return atst_bn_relu_bwd(dy, h, mean, rstd, gamma, beta, R, N, sum_dy, sum_dy_xhat, scratch, ST(stream));
}
int atst_bn_bwd_dx_bf16(const float* dy, const float* h, const float* mean, const float* rstd, const float* gamma,
                        const float* beta, const float* sum_dy, const float* sum_dy_xhat, float inv_count, int R, int N,
                        uint16_t* dh, void* stream) {
  return atst_bn_bwd_dx(dy, h, mean, rstd, gamma, beta, sum_dy, sum_dy_xhat, inv_count, R, N, BF(dh), ST(stream));
}
int atst_bn_bwd_dx_f32(const float* dy, const float* h, const float* mean, const float* rstd, const float* gamma,
                       const float* beta, const float* sum_dy, const float* sum_dy_xhat, float inv_count, int R, int N,
                       float* dh, void* stream) {
  return atst_bn_bwd_dx_fp32(dy, h, mean, rstd, gamma, beta, sum_dy, sum_dy_xhat, inv_count, R, N, dh, ST(stream));
}
int atst_byol_loss_f32(const float* student, const float* teacher, int B, int ncrops, int D, float* acc, float* dstudent,
                       float* stats, void* stream) {
  return atst_byol_loss(student, teacher, B, ncrops, D, acc, dstudent, stats, ST(stream));
}

int atst_adamw_ema_step(float* p, const float* g, float* m, float* v, float* t, uint16_t* p_bf16, uint16_t* t_bf16,
                        const uint8_t* chunk_flags, size_t n, size_t n_teacher, double lr, double wd, double beta1,
                        double beta2, double eps, double step_size, double ema_m, double grad_scale, void* stream) {
  OptimArgs a{};
  a.p = p; a.g = g; a.m = m; a.v = v; a.t = t; a.p_bf16 = BF(p_bf16); a.t_bf16 = BF(t_bf16); a.chunk_flags = chunk_flags;
  a.n = n; a.n_teacher = n_teacher;
  // scalars arrive as doubles and are rounded once here, exactly as torch rounds python floats for fp32 tensors
  a.lr_wd = (float)(lr * wd); a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.om_beta1 = (float)(1.0 - beta1);
  a.om_beta2 = (float)(1.0 - beta2); a.eps = (float)eps; a.step_size = (float)step_size; a.ema_m = (float)ema_m;
  a.om_ema = (float)(1.0 - ema_m); a.grad_scale = (float)grad_scale;
  return atst_adamw_ema(a, ST(stream));
}

}  // extern "C"
