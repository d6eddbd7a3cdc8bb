// Internal launcher prototypes (one per kernel family).  Not part of the C ABI.
#pragma once
#include "common.h"
#include "../../include/vag_nmt.h"

// ---------------- elem.hip ----------------
// out[(t*B+b), :] = W[idx[b*isb + t*ist], :] * dropout
// mask_out (optional, indexed like idx): 1 where the index is not the padding index 0
int vag_embed_gather_launch(const int64_t* idx, int64_t ist, int64_t isb, int64_t T, int64_t B, const float* W,
                            int64_t E, float* out, const uint64_t* rng, int sid, float p, hipStream_t s,
                            float* mask_out = nullptr);
// gW[idx, :] += g[(t*B+b), :] * dropout, skipping idx == 0 (padding_idx)
int vag_embed_scatter_launch(const int64_t* idx, int64_t ist, int64_t isb, int64_t T, int64_t B, const float* g,
                             int64_t E, float* gW, const uint64_t* rng, int sid, float p, hipStream_t s,
                             const unsigned* poison = nullptr);   // poison: see embed_scatter_kernel
// set by vag_train_step around its last phase: the encoder's embedding scatter then carries the persistent kernels' give-up word
// into the gradient buffer (the per-operator entry points never do: their gradients go to the caller's own optimiser)
extern thread_local bool g_step_poison_inject;

struct GruBwdSide {
    const float* dh_carry;   // (M,H) gradient arriving from the later time step, or NULL
    const float* dh_add;     // optional extra gradient source, element (m,j) at dh_add[m*ld_add + j]
    int64_t drop_idx0;       // dropout index of dh_add element (m,j) = m*ld_add + drop_idx0 + j
    const float* save;       // [4][M][H] r,z,n,hn from the forward step
    const float* hprev;      // (M,H) ld = ldh
    float* dgi;              // (M,3H) ld = ldgi
    float* dgh;              // (M,3H) ld = ldgh
    float* dh_prev;          // (M,H): z * dh (the W_hh^T dgh part is added by the following GEMM)
    int t;
};
struct GruBwdArgs {
    GruBwdSide s[2];
    int64_t ld_add, ldh, ldgi, ldgh;
    int M, H;
    const int* lengths;
    const uint64_t* rng; int sid; float p;   // dropout applied to dh_add (encoder context dropout)
};
int vag_gru_bwd_elem_launch(const GruBwdArgs& a, int nz, hipStream_t s);

// dx = dy * dropout(i) * (1 - t*t)   (in place allowed).  With dropout, y is the post-dropout value t*mul;
// t is recovered from it (dropped elements have zero gradient).
int vag_tanh_bwd_launch(const float* y, const float* dy, float* dx, int64_t n, const uint64_t* rng, int sid, float p,
                        hipStream_t s, int64_t idx0 = 0);      // idx0: dropout index of element 0 (a row chunk of a larger array)
// x[i] *= dropout(idx0 + i)
int vag_dropout_apply_launch(float* x, int64_t n, int64_t idx0, const uint64_t* rng, int sid, float p, hipStream_t s);
int vag_tanh_dropout_launch(float* x, int64_t n, int64_t idx0, const uint64_t* rng, int sid, float p, hipStream_t s);
int vag_dropout_mask_launch(const uint64_t* rng, int sid, int64_t n, float p, float* out, hipStream_t s);
// y (+)= a*x
int vag_axpy_launch(float a, const float* x, float* y, int64_t n, int accumulate, hipStream_t s);
// mask[b,t] = src[b,t] != 0
int vag_src_mask_launch(const int64_t* src, int64_t n, float* mask, hipStream_t s);
// xmix[b,c] = split*ctx[b,c] + (1-split) * sum_t enc[b,t,c] / sum_t mask[b,t]   (ctx NULL -> split treated as 0)
int vag_meanpool_mix_launch(const float* enc, const float* mask, const float* ctx, float split, int64_t B, int64_t Ts,
                            int64_t C, float* xmix, hipStream_t s);
// d_enc[b,t,c] (+)= coef * dx[b,c] / sum_t mask[b,t]
int vag_meanpool_bwd_launch(const float* mask, const float* dx, float coef, int64_t B, int64_t Ts, int64_t C,
                            float* d_enc, int accumulate, hipStream_t s);
int vag_rng_advance_launch(uint64_t* rng, hipStream_t s);
// out[i*ldo + j] = in[i*ldi + j] for a (rows x cols) block
int vag_copy2d_launch(const float* in, int64_t ldi, float* out, int64_t ldo, int64_t rows, int64_t cols, hipStream_t s);

int vag_gather_rows_i64_launch(const int64_t* in, int64_t ld, const int64_t* idx, int64_t rows, int64_t w, int64_t* out,
                               hipStream_t s);

// several small fills (kind 0) / copies (1) / transposes (2) of (rows x cols) fp32 matrices in one launch
struct VagJob { const float* src; float* dst; int64_t rows, cols, ld_src, ld_dst; int kind; };
constexpr int VAG_MAXJOBS = 12;
struct VagJobs { VagJob j[VAG_MAXJOBS]; int start[VAG_MAXJOBS + 1]; int n; };
int vag_jobs_launch(const VagJob* jobs, int n, hipStream_t s);

int vag_copy4_launch(const void* const* src, void* const* dst, const int64_t* bytes, int n, hipStream_t s);

// ---------------- attn.hip ----------------
#ifndef VAG_POST_SC
#define VAG_POST_SC 8          // source positions per attn_post_bwd block
#endif
static inline int64_t VAG_POST_CHUNKS(int64_t Ts) { return (Ts + VAG_POST_SC - 1) / VAG_POST_SC; }
// mode 0: scores[n,s] = sum_c v[c] tanh(pe[b,s,c] + q[n,c]); mode 1: scores[n,s] = sum_c q[n,c] * pe[b,s,c].
// b = n / rps.  mask (Bsrc,Ts) float or NULL: masked positions get -inf.
// q row stride ldq (>= C).
int vag_attn_scores_launch(int mode, const float* pe, const float* q, int64_t ldq, const float* v, const float* mask,
                           int64_t N, int64_t rps, int64_t Ts, int64_t C, float* scores, hipStream_t s);
int vag_attn_scores_ex_launch(int mode, const float* pe, const float* q, int64_t ldq, const float* v, const float* mask,
                              int64_t N, int64_t rps, int64_t rows_mod, int64_t Ts, int64_t C, const float* addend,
                              float* scores, hipStream_t s);
int vag_attn_ctx_gru_launch(const float* scores, const float* encwp, int64_t N, int64_t rps, int64_t Ts, int64_t H,
                            const float* b_ih, const float* hp, int64_t ldhp, const float* hprev, float* alpha, float* hout,
                            float* save, hipStream_t s, bool x16 = false,       // x16: encwp is fp16
                            const float* x2 = nullptr, int64_t W2 = 0, float* out2 = nullptr);     // out2 (N,W2) = alpha . x2 (B,Ts,W2)
int vag_attn_wsum_launch(int over_src, const float* a, const float* x, int64_t B, int64_t Ts, int64_t T, int64_t W, float* out,
                         hipStream_t s);
// softmax=1: alpha[n,:] = softmax(scores[n,:]) (written to alpha), ctx[n,c] = sum_s alpha[n,s] enc[b,s,c]
// softmax=0: weights = scores as given (alpha not written)
int vag_attn_ctx_launch(int softmax, const float* scores, const float* enc, int64_t N, int64_t rps, int64_t Ts,
                        int64_t C, float* alpha, float* ctx, hipStream_t s);
// dscore[n,s] = alpha[n,s] * (dalpha[n,s] - sum_s' alpha dalpha)
int vag_softmax_bwd_launch(const float* alpha, const float* dalpha, int64_t N, int64_t Ts, float* dscore, hipStream_t s);
// dq[n,c] = sum_s dscore[n,s] * v[c] * (1 - tanh^2(pe[n,s,c] + q[n,c]))      (training: rps = 1)
// alpha/dalpha given: dscore is first computed (softmax backward) and stored; NULL: dscore is an input.
int vag_attn_dq_launch(const float* pe, const float* q, int64_t ldq, const float* v, const float* alpha,
                       const float* dalpha, float* dscore, int64_t N, int64_t Ts, int64_t C, float* dq, int64_t lddq,
                       hipStream_t s, bool x16 = false);                       // x16: pe is fp16
// After the time loop: d_pe[b,s,c] = v[c] sum_t ds[t,b,s] (1-th^2);  dvp[(z*B+b),c] = sum_{t, s in chunk z} ds*th
// (VAG_POST_CHUNKS(Ts) = ceil(Ts/8) chunk rows per batch row; column-sum all rows for dv);
// d_enc[b,s,c] (+)= sum_t alpha[t,b,s] dc[t,b,c]   (skipped when dc == NULL)
int vag_attn_post_bwd_launch(const float* pe, const float* q_all, int64_t ldq, const float* v, const float* ds_all,
                             const float* alpha_all, const float* dc_all, int64_t B, int64_t Ts, int64_t Tt,
                             int64_t C, float* d_pe, float* dvp, float* d_enc, int accumulate_enc, hipStream_t s,
                             bool x16 = false);                                // x16: pe is fp16
// out[b,t,c] (+)= a1[b,t]*x1[b,c] + a2[b,t]*x2[b,c]
int vag_outer2_launch(const float* a1, const float* x1, const float* a2, const float* x2, int64_t B, int64_t Ts,
                      int64_t C, float* out, int accumulate, hipStream_t s);

// ---------------- head.hip ----------------
// per row: lse, nll = -w[tgt]*(x[tgt]-lse) (tgt NULL: skipped), argmax (may be NULL), logp_out (may be NULL)
// row r = t*B + b;  tgt index = tgt[b*Tt + t]
int vag_lse_nll_launch(const float* logits, int64_t ldl, int64_t rows, int64_t V, const int64_t* tgt, int64_t B,
                       int64_t Tt, const float* vw, float* lse, float* nll, int64_t* argmax, int64_t argmax_stride,
                       float* logp_out, int64_t ldlp, hipStream_t s);
int vag_inv_cnt_launch(const int64_t* tgt, int64_t B, int64_t Tt, float* inv_cnt, hipStream_t s);
int vag_loss_mt_launch(const float* nll, const float* inv_cnt, int64_t B, int64_t Tt, float* loss, hipStream_t s);
// the same, writing losses[1] = loss_mt and the mixed total losses[0] = w_mt*loss_mt + w_vse*losses[2] (V11.py:166)
void vag_set_loss_ring(int r);
int vag_loss_mt_mix_launch(const float* nll, const float* inv_cnt, int64_t B, int64_t Tt, float* losses, float w_mt,
                           float w_vse, int has_vse, hipStream_t s);
// in place: logits[r,j] = d_loss * inv_cnt[b]/B * w[tgt] * (softmax_j - [j==tgt]); pad columns [V,ldl) = 0
int vag_ce_bwd_launch(float* logits, int64_t ldl, int64_t rows, int64_t V, const int64_t* tgt, int64_t B, int64_t Tt,
                      const float* vw, const float* lse, const float* inv_cnt, const float* d_loss, hipStream_t s);
int vag_ce_bwd_colsum_launch(float* logits, int64_t ldl, int64_t rows, int64_t V, const int64_t* tgt, int64_t B, int64_t Tt,
                             const float* vw, const float* lse, const float* inv_cnt, const float* d_loss, float* g_bias,
                             hipStream_t s, void* out16 = nullptr);

int vag_logsoftmax_bwd_launch(const float* logp, int64_t ldlp, float* d, int64_t ldd, int64_t rows, int64_t V,
                              hipStream_t s);

// ---------------- vse.hip ----------------
int vag_l2norm_fwd_launch(const float* y, int64_t B, int64_t S, float* nrm, float* out, hipStream_t s);
// dy = l2norm backward of d_out, then (act) * (1 - y^2); written to dy (may alias d_out)
int vag_l2norm_bwd_launch(const float* y, const float* nrm, const float* out, const float* d_out, int64_t B, int64_t S,
                          int act, float* dy, hipStream_t s);
int vag_rank_loss_launch(const float* scores, int64_t B, float margin, int kind, float* G, float* loss, hipStream_t s,
                         const float* g_scale = nullptr);
int vag_retrieval_rank_launch(const float* scores, int64_t N, int* ranks, hipStream_t s);
// x[i] *= *scalar
int vag_scale_by_dev_launch(float* x, int64_t n, const float* scalar, hipStream_t s);

// ---------------- optim.hip ----------------
int vag_clip_adam_shard_launch(float* p, float* g, float* m, float* v, int64_t n, int nseg, const int64_t* seg_off,
                               const float* seg_lr, const float* seg_wd, float clip, float grad_scale, float beta1, float beta2,
                               float eps, int zero_grad, int32_t* step, float* norm_out, void* scratch, const float* lr_dev,
                               int64_t lo, int64_t hi, int phase, double* sumsq, hipStream_t s);
int vag_clip_adam_launch(float* p, float* g, float* m, float* v, int64_t n, int nseg, const int64_t* seg_off,
                         const float* seg_lr, const float* seg_wd, float clip, float grad_scale, float beta1,
                         float beta2, float eps, int zero_grad, int32_t* step, float* norm_out, void* scratch, const float* lr_dev,
                         hipStream_t s);

// ---------------- persist.hip: whole recurrences in one launch ----------------
bool vag_enc_persistent_ok(int64_t B, int64_t Ts, int64_t H);
int64_t vag_enc_persistent_sync_words(int64_t B, int64_t Ts);
int vag_enc_fwd_persistent_launch(const float* xp, const float* w_fw, const float* w_bw, const float* b_fw, const float* b_bw,
                                  const int* lengths, float* hst, float* gates, float* enc, unsigned* sync, const uint64_t* rng,
                                  float p_ctx, int64_t B, int64_t Ts, int64_t H, hipStream_t s);
int vag_enc_bwd_persistent_launch(const float* whhT, const float* d_enc, const float* gates, const float* hst, const int* lengths,
                                  const uint64_t* rng, float p_ctx, float* d_xp, float* dgh, unsigned* sync, int64_t B, int64_t Ts,
                                  int64_t H, hipStream_t s);
bool vag_enc_wide16_ok(int64_t B, int64_t Ts, int64_t H);
int vag_enc_fwd_wide16_launch(const float* xp, const vag_half* w16_fw, const vag_half* w16_bw, const float* b_fw, const float* b_bw,
                              const int* lengths, float* hst, float* gates, float* enc, vag_half* hx, unsigned* sync, const uint64_t* rng,
                              float p_ctx, int64_t B, int64_t Ts, int64_t H, hipStream_t s);
int vag_enc_bwd_wide16_launch(const vag_half* wt16, const float* d_enc, const float* gates, const float* hst, const int* lengths,
                              const uint64_t* rng, float p_ctx, float* d_xp, float* dgh, vag_half* gx, unsigned* sync, int64_t B,
                              int64_t Ts, int64_t H, hipStream_t s);
int vag_gemm_group_plan_host(int n, const int64_t* M, const int64_t* N, const int64_t* K, const int* accumulate, int* split,
                             int* order);
void vag_gemm_set_scratch(float* slab, int64_t floats, unsigned* tickets, int64_t ntickets);      // gemm.hip: scratch of the slab form of split-K for the calling thread (NULL: none)
void vag_gemm_group_leaf_stream(hipStream_t s, hipEvent_t ev);      // gemm.hip: side stream of the TN (weight-gradient) layout of the group flushes that follow (NULL: none)
bool vag_gemm_group_leaf_used();                                    // ... whether a flush went there since it was set
int vag_persistent_timeouts_read(void);
unsigned* vag_persist_guard(void);           // {void flag, give-up count} pair of the launches the calling thread enqueues (persist.hip)
void vag_persist_guard_set(unsigned* g);     // the caller's own pair (NULL: the process-wide pair)
unsigned* vag_persist_guard_peek(void);      // what vag_persist_guard_set last set on this thread (NULL: none)
int vag_persistent_time_read(int kind, double* ms_total, int* launches);
bool vag_dec_bwd_persistent_ok(int64_t B, int64_t Ts, int64_t Tt, int64_t H);
int vag_dec_bwd_persistent_launch(const float* pe, const float* encwp, const float* v, const float* wcatT, const float* whh1T,
                                  const float* h0, const float* h2_all, const float* h1, const float* g1, const float* g2,
                                  const float* qhp, const float* alpha, const float* d_h2_all, const float* dah, float* dgi2,
                                  float* dqgh, float* ds, float* dgi1, float* dgh1, float* d_h0, float* dal, unsigned* sync,
                                  int64_t B, int64_t Ts, int64_t Tt, int64_t H, hipStream_t s);
float* vag_cgru_bwd_scratch_de(float* scratch, int64_t B, int64_t Ts, int64_t Tt, int64_t E, int64_t H);
float* vag_cgru_bwd_scratch_du(float* scratch, int64_t B, int64_t Ts, int64_t Tt, int64_t E, int64_t H);
int vag_attn_dot_row_launch(bool bwd, const float* x, const float* q, int64_t ldq, const float* mask, const float* alpha, int64_t B,
                            int64_t Ts, int64_t C, float* wout, float* sum, hipStream_t s);      // attn.hip: one launch per dot attention
void vag_skinny_nn_out2(float* out2, int64_t ld, float scale, int accumulate);      // gemm.hip: a second destination for the next vag_skinny_nn_launch
int vag_rank_bwd_launch(const float* G, const float* im, const float* sv, const float* d_loss, int64_t B, int64_t S, float* d_im,
                        float* d_s, hipStream_t s);       // vse.hip: d_im = G s and d_s = G^T im (x *d_loss) in one launch, B <= 128
int vag_attn_wsum_pair_launch(const float* a, const float* y, int64_t Wy, float* out_src, const float* x, int64_t Wx, float* out_time,
                              int64_t B, int64_t Ts, int64_t T, hipStream_t s);        // attn.hip: sum over t of a y and sum over s of a x, one grid
void vag_loss_defer_begin();                // head.hip: hold the loss reduction back for the next ce_bwd_colsum launch ...
int vag_loss_defer_flush();                 // ... or launch it now if none came
void vag_attn_row_mix_request(float* xmix, float split);     // attn.hip: the next forward row launch also leaves the initial state's input ...
void vag_attn_row_mix_cancel();                                  // ... (a request nobody took must not outlive the call that made it)
bool vag_attn_row_mix_done(const float* xmix);               // ... if it could (asked once: resets)
void vag_persist_dh0_tanh_request(bool on);          // persist.hip: the next decoder backward launch applies (1 - h0^2) to d_h0 ...
bool vag_persist_dh0_tanh_done(const float* d_h0);   // ... whether it did (asked once: resets)
bool vag_attn_row_gru_ok(int64_t N, int64_t Ts, int64_t H, int64_t W2);       // attn.hip: a decoding step's attention + gru_2 + W2 c in one launch
int vag_attn_row_gru_launch(const float* pe, const float* q, int64_t ldq, const float* v, const float* mask, const float* keys,
                            const float* x2, int64_t N, int64_t rps, int64_t Ts, int64_t H, int64_t W2, const float* b_ih,
                            const float* hp, int64_t ldhp, const float* hprev, float* alpha, float* hout, float* out2, hipStream_t s);
void vag_rmw_defer_begin(float* out);      // attn.hip: hold back accumulating outer2 / meanpool_bwd launches into `out` ...
int vag_rmw_defer_flush(hipStream_t s);    // ... and do them in one pass
void vag_rmw_defer_abort();
bool vag_rmw_defer_meanpool(const float* mask, const float* dx, float coef, int64_t B, int64_t Ts, int64_t C, float* out, int accumulate);
void vag_step_set_gathered(bool v);        // api.hip: the step's prologue embedded the decoder's input tokens (e_all)
void vag_step_set_zeroed(bool v);          // api.hip: the step's prologue zeroed tmid and the encoder's dx
void vag_step_zero_ranges(float* ws_enc, float* ws_dec, int64_t B, int64_t Ts, int64_t Tt, int64_t Es, int64_t Et, int64_t H,
                          unsigned** p, int64_t* n);                 // api.hip
void vag_persist_set_prezeroed(bool v);      // calling thread: the launches below skip zeroing their counters / exchange buffers
bool vag_dec_persistent_ok(int64_t B, int64_t Ts, int64_t Tt, int64_t H);
int64_t vag_dec_persistent_sync_words(int64_t B, int64_t Tt);
int vag_dec_fwd_persistent_launch(const float* pe, const float* mask, const float* h0, const float* xp1, const float* W1,
                                  const float* b1, const float* wcat, const float* bcat, const float* v, const float* encwp,
                                  const float* b_ih2, float* h1, float* g1, float* qhp, float* alpha, float* h2_all, float* g2,
                                  float* psc, unsigned* sync, int64_t B, int64_t Ts, int64_t Tt, int64_t H, hipStream_t s);
// free-running form (the kernel feeds its own arg-max back): see persist.hip
bool vag_dec_free_persistent_ok(int64_t B, int64_t Ts, int64_t Tt, int64_t E, int64_t H, int64_t V);
int64_t vag_dec_free_tables_floats(int64_t B, int64_t Ts, int64_t Tt, int64_t E, int64_t H, int64_t V);
int vag_dec_free_persistent_launch(const float* pe, const float* mask, const float* h0, const float* W1, const float* b1,
                                   const float* wcat, const float* bcat, const float* v, const float* encwp, const float* b_ih2,
                                   float* h1, float* g1, float* qhp, float* alpha, float* h2_all, float* g2, float* psc,
                                   unsigned* sync, const float* tables, const float* hw1, const float* hb1, const float* hb2,
                                   const float* hb3, const float* out_w, const float* out_b, float* tmid, float* logits, int64_t ldl,
                                   int64_t* tok, const uint64_t* rng, float p_out, int64_t B, int64_t Ts, int64_t Tt, int64_t E,
                                   int64_t H, int64_t V, hipStream_t s);

// ---------------- beam.hip ----------------
int64_t vag_beam_scratch_bytes_impl(int64_t B, int64_t k, int64_t V);
int vag_beam_step_launch(float* logp, int64_t ldl, float* nll, int64_t* beam, int64_t di, int32_t* di_state,
                         int64_t max_len, const float* h_in, float* h_out, int64_t* tok_out, int64_t B, int64_t k,
                         int64_t V, int64_t H, int32_t* n_alive, void* scratch, hipStream_t s, const float* parts = nullptr,
                         int64_t nparts = 0);
int vag_beam_finish_launch(const float* nll, const int64_t* beam, int64_t max_len, int64_t steps, int64_t B, int64_t k,
                           int64_t* out, float* best, hipStream_t s);

// ---------------- api.hip internals shared with step.hip ----------------
void vag_set_derived_override(const float* d);
void vag_set_store16(bool on);
const float* vag_get_derived_override();
bool vag_get_store16();
void vag_set_head_chunk(int64_t rows);
void vag_set_head_fuse(const vag_head_g* g, const float* d_loss, float* dt);
// the fused step's ranking loss: G pre-multiplied by a device scalar in the forward (g_scale), no scaling pass in the backward (d_loss NULL)
int vag_rank_loss_fwd_impl(const float* im, const float* sv, int64_t B, int64_t S, float margin, int kind, float* scores,
                           float* G, float* loss, const float* g_scale, hipStream_t s);
int vag_rank_loss_bwd_impl(const float* im, const float* sv, const float* G, const float* d_loss, int64_t B, int64_t S,
                           float* d_im, float* d_s, hipStream_t s);
int vag_dec_init_bwd_impl(const float* mask, const float* xmix, const float* h0, float split, const float* W, float* d_h0,
                          int64_t B, int64_t Ts, int64_t C, int64_t H, float* d_enc, int accumulate_enc, float* d_ctx,
                          int accumulate_ctx, float* g_W, float* g_b, float* scratch, hipStream_t s);
int vag_imagine_attn_ctx_bwd_impl(const float* im_emb, const float* enc, const float* mask, const float* ctx2ctx,
                                  const float* emb2ctx, const float* mlp_w, int method, int64_t B, int64_t Ts, int64_t C,
                                  int64_t S, const float* alpha, const float* d_ctx, float* ws, float* d_enc,
                                  int accumulate_enc, float* d_im_emb, int accumulate_im, float* g_ctx2ctx,
                                  float* g_emb2ctx, float* g_mlp_w, hipStream_t s);
int vag_head_ce_seq_fwd_impl(const float* h2_all, const float* c_all, const float* e_all, vag_head_w w, const int64_t* tgt,
                             const float* vocab_weight, int64_t B, int64_t Tt, int64_t E, int64_t H, int64_t V, float p_out,
                             const uint64_t* rng, int logits_ready, float* tmid, float* logits, int64_t ldl, float* lse,
                             float* nll, float* inv_cnt, int inv_cnt_ready, float* loss_mt, float* losses, float w_mt,
                             float w_vse, int has_vse, hipStream_t s);

// decoder parameter gradients from the rows of time steps [t0, t1) (first: this chunk initialises the folded-product
// gradient instead of adding to it), and what remains once every chunk is in
int vag_cgru_bwd_weights_chunk(const float* h0, const int64_t* tok, vag_dec_w w, int64_t B, int64_t Ts, int64_t Tt, int64_t E,
                               int64_t H, const float* h2_all, const float* c_all, const float* e_all, const float* d_e_all,
                               float* ws, vag_dec_g g, float* scratch, int64_t t0, int64_t t1, bool first, hipStream_t s);
int vag_cgru_bwd_weights_scatter(const int64_t* tok, int64_t B, int64_t Ts, int64_t Tt, int64_t E, int64_t H, vag_dec_g g,
                                 float* scratch, int64_t t0, int64_t t1, hipStream_t s);
int vag_cgru_bwd_weights_finish(vag_dec_w w, int64_t B, int64_t Ts, int64_t Tt, int64_t E, int64_t H, vag_dec_g g,
                                float* scratch, bool with_attn_v, hipStream_t s);
