/* vag_nmt.h -- C ABI of libvagnmt.so: the VAG-NMT per-step training / decode hot path on MI355X (gfx950).
 *
 * This is the drop-in boundary.  The reference (Eurus-Holmes/VAG-NMT) is pure Python on torch; the functions
 * below replace the torch-op sequences of its hot path, and each comment cites the reference code it stands
 * in for (paths relative to the reference checkout).  The Python host side in
 * vag-nmt_amd/machine_translation_vision/ mirrors the reference's module API and calls these entry points
 * through ctypes.
 *
 * Conventions
 *   - All tensors are dense row-major fp32 on the current HIP device unless stated; token ids are int64
 *     (torch.LongTensor, as in the reference); lengths are int32 on the device.
 *   - Source-side sequences are batch-major inside the library: enc/pe are (B,Ts,C), C = 2H.
 *     Decoder-side per-step tensors are time-major: (Tt,B,*).
 *   - Ownership: the caller owns every buffer, including workspaces ("ws", sizes from *_ws_floats()).  The
 *     library never allocates or frees device memory and keeps no pointer after a call returns.
 *   - All work is enqueued on `stream`; nothing synchronises with the host, so every call can be captured
 *     into a HIP graph.  Calls are re-entrant across streams.
 *   - Gradient outputs named g_* are ACCUMULATED (+=) into the caller's buffers (zero them per step);
 *     outputs named d_* are written.
 *   - Return value: 0 ok; <0 argument/shape error (-22 = EINVAL); >0 a hipError_t.
 *   - Dropout: `rng` points to two device uint64 {seed, step}; masks are a pure function of
 *     (seed, step, stream-id, element index) so backward recomputes them.  rng == NULL or p == 0: no dropout
 *     (eval mode).  vag_dropout_mask() materialises a mask for tests.
 */
#ifndef VAG_NMT_H
#define VAG_NMT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vag_stream_t; /* hipStream_t */

/* GRU parameter bundle, torch gate order (r,z,n): w_ih (3H,in), w_hh (3H,H), b_ih (3H), b_hh (3H). */
typedef struct { const float *w_ih, *w_hh, *b_ih, *b_hh; } vag_gru_w;
typedef struct { float *w_ih, *w_hh, *b_ih, *b_hh; } vag_gru_g;

/* Decoder recurrent parameters (layers/NMT_Decoder.py:78-86): embedding (V,E); gru_1 (in=E); attn_h (C,H);
 * attn.v (C); context2hid (H,C); gru_2 (in=H). */
typedef struct {
    const float* emb;
    vag_gru_w gru1;
    const float *attn_h, *attn_v, *c2h;
    vag_gru_w gru2;
} vag_dec_w;
typedef struct {
    float* emb;
    vag_gru_g gru1;
    float *attn_h, *attn_v, *c2h;
    vag_gru_g gru2;
} vag_dec_g;

/* Output-head parameters (layers/NMT_Decoder.py:89-106): W1 (E,H), W2 (E,C), W3 (E,E), out (V,E)+(V).
 * With tied embeddings `out_w` is the decoder embedding matrix. */
typedef struct { const float *w1, *b1, *w2, *b2, *w3, *b3, *out_w, *out_b; } vag_head_w;
typedef struct { float *w1, *b1, *w2, *b2, *w3, *b3, *out_w, *out_b; } vag_head_g;

int vag_version(void);

/* ---- generic dense products (torch.nn.Linear / torch.mm call sites on the path) ---------------------- */
/* C[M,N] = act(alpha * op(A) op(B) + beta*C + bias[n]).  A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn];
 * one stride of each operand must be 1.  act: 0 none, 1 tanh.  Arithmetic: fp32-grade.  Products with M,N > 64 run on
 * the bf16 matrix pipes with every fp32 operand split exactly into three bf16 parts and the six partial products with
 * i+j <= 4 accumulated in fp32 ("bf16x6": relative error of a term ~2^-24, the class of a reassociated fp32 sum;
 * bounded against the exact-f32 kernel in tests/test_gpu_kernels.py); smaller ones and VAG_GEMM_F32MFMA=1 use the
 * f32-input MFMA (exact f32 fma chains). */
int vag_gemm_f32(int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t sam, int64_t sak,
                 const float* B, int64_t sbk, int64_t sbn, float beta, float* C, int64_t ldc,
                 const float* bias, int act, vag_stream_t stream);
/* y[M,N] = act(x[M,K] W[N,K]^T + bias)  -- nn.Linear forward; picks the small-M kernel for M <= 128. */
int vag_linear_fwd(int64_t M, int64_t N, int64_t K, const float* x, const float* W, const float* bias, int act,
                   float* y, vag_stream_t stream);
/* nn.Linear backward.  If act==1, dy is first multiplied in place by (1 - y^2).  d_x (may be NULL) is
 * written or, if accumulate_dx, added to; g_W, g_b (may be NULL) are accumulated. */
int vag_linear_bwd(int64_t M, int64_t N, int64_t K, const float* x, const float* W, const float* y, float* dy,
                   int act, float* d_x, int accumulate_dx, float* g_W, float* g_b, vag_stream_t stream);

/* ---- embedding (nn.Embedding(padding_idx=0), layers/Encoder.py:22,50; layers/NMT_Decoder.py:78,118) --- */
int vag_embed_fwd(const int64_t* idx, int64_t n, const float* W, int64_t E, float* out, vag_stream_t stream);
int vag_embed_bwd(const int64_t* idx, int64_t n, const float* d_out, int64_t E, float* g_W, vag_stream_t stream);

/* ---- a3: bi-GRU encoder, layers/Encoder.py:36-66 ------------------------------------------------------ */
/* src (B,Ts) int64 padded with 0; lengths int32[B] on device (descending).  Writes enc (B,Ts,2H)
 * (forward direction in [:H], reverse in [H:], zeros past each row's length) and mask (B,Ts) = (src != 0).
 * p_emb / p_ctx: dropout on the embedded input / on the output (Encoder.py:51-52,:63-64).
 * ws: vag_bigru_ws_floats(B,Ts,E,H) floats, must stay untouched until the matching backward. */
int64_t vag_bigru_ws_floats(int64_t B, int64_t Ts, int64_t E, int64_t H);
int vag_bigru_seq_fwd(const int64_t* src, const int32_t* lengths, const float* emb, vag_gru_w fwd, vag_gru_w bwd,
                      float p_emb, float p_ctx, const uint64_t* rng, int64_t B, int64_t Ts, int64_t E, int64_t H,
                      float* enc, float* mask, float* ws, vag_stream_t stream);
/* d_enc (B,Ts,2H) is consumed (overwritten).  Accumulates g_emb (Vs,E; pad row untouched) and both GRUs. */
int vag_bigru_seq_bwd(const int64_t* src, const int32_t* lengths, vag_gru_w fwd, vag_gru_w bwd, float p_emb,
                      float p_ctx, const uint64_t* rng, int64_t B, int64_t Ts, int64_t E, int64_t H, float* d_enc,
                      float* ws, float* g_emb, vag_gru_g g_fwd, vag_gru_g g_bwd, vag_stream_t stream);

/* ---- one GRU cell step (torch nn.GRU on a length-1 sequence, layers/NMT_Decoder.py:121) ------------------ */
/* gi (M,3H) = W_ih x + b_ih (already projected); computes W_hh h_prev + b_hh, the gates and the blend in one
 * launch (the kernel every recurrent step of the encoder and decoder runs).  save: NULL or [4][M][H] (r,z,n,hn). */
int vag_gru_cell_fwd(const float* gi, const float* h_prev, const float* w_hh, const float* b_hh, int64_t M,
                     int64_t H, float* h_out, float* save, vag_stream_t stream);

/* One backward step of the same recurrence (what autograd replays per time step for nn.GRU, layers/Encoder.py:58,
 * layers/NMT_Decoder.py:121,129), fused the way the sequence operators run it:
 *   dh  = dgh_next W_hh + carry + d_out        dgh_next (M,3H): gradient of the LATER step's hidden projection,
 *                                              w_hh_t (H,3H) = W_hh^T, carry / d_out (M,H) may be NULL
 *   dgi (M,3H), dgh (M,3H) = cell backward of THIS step (save [4][M][H] from vag_gru_cell_fwd, h_prev (M,H))
 *   carry_out (M,H) = z * dh                   (the part of dh that flows to the previous step directly) */
int vag_gru_cell_bwd(const float* dgh_next, const float* w_hh_t, const float* carry, const float* d_out,
                     const float* save, const float* h_prev, int64_t M, int64_t H, float* dgi, float* dgh,
                     float* carry_out, vag_stream_t stream);

/* ---- a4 (hoisted part): attention keys pe = enc W_e^T, layers/NMT_Decoder.py:47 ----------------------- */
/* The reference recomputes attn_e(encoder_outputs) at every decoder step; it does not depend on the step,
 * so it is computed once per batch.  rows = B*Ts. */
int vag_attn_keys_proj(const float* enc, const float* attn_e, int64_t rows, int64_t C, float* pe,
                       vag_stream_t stream);
/* a4 stand-alone (BahdanauAttn.forward, layers/NMT_Decoder.py:27-51), inference: q (N,C) = attn_h(hidden);
 * alpha[n,:] = softmax_s(v . tanh(pe[n/rows_per_src,s] + q[n])) with masked positions at -inf; ctx = alpha . enc. */
int vag_bahdanau_attn_fwd(const float* pe, const float* q, const float* v, const float* mask, const float* enc,
                          int64_t N, int64_t rows_per_src, int64_t Ts, int64_t C, float* scores, float* alpha,
                          float* ctx, vag_stream_t stream);
/* d_enc (+)= d_pe attn_e ; g_attn_e += d_pe^T enc */
int vag_attn_keys_proj_bwd(const float* enc, const float* attn_e, const float* d_pe, int64_t rows, int64_t C,
                           float* d_enc, int accumulate_enc, float* g_attn_e, vag_stream_t stream);

/* ---- a5: cGRU decoder with Bahdanau attention, layers/NMT_Decoder.py:109-131 -------------------------- */
/* Whole target sequence.  tok (Tt+1,B) int64: row 0 = SOS, row t+1 = input of step t+1.  Teacher forcing
 * (models/...V11.py:138-146): the caller fills every row.  Free running (V11.py:148-160, free_run=1): rows
 * 1.. are written here with the argmax of each step's output distribution, for which the head parameters,
 * `tmid` (Tt,B,E) and `logits` (Tt*B, ldl) are also produced step by step (p_out = head dropout).
 * Outputs for the head: h2_all (Tt,B,H), c_all (Tt,B,C), e_all (Tt,B,E).
 * ws: vag_cgru_ws_floats(B,Ts,Tt,E,H) floats, kept for the backward. */
int64_t vag_cgru_ws_floats(int64_t B, int64_t Ts, int64_t Tt, int64_t E, int64_t H);
/* float offset of a saved per-step tensor inside ws (parity tests): 0 alpha (Tt,B,Ts), 1 h1 (Tt,B,H), 2 [q | W_hh2 h1 + b] */
int64_t vag_cgru_ws_offset(int64_t B, int64_t Ts, int64_t Tt, int64_t E, int64_t H, int which);
int vag_cgru_attn_decode_seq_fwd(const float* enc, const float* pe, const float* mask, const float* h0,
                                 int64_t* tok, vag_dec_w w, int64_t B, int64_t Ts, int64_t Tt, int64_t E,
                                 int64_t H, int64_t V, float* h2_all, float* c_all, float* e_all, float* ws,
                                 int free_run, const vag_head_w* head, float p_out, const uint64_t* rng,
                                 float* tmid, float* logits, int64_t ldl, vag_stream_t stream);
/* The free-running form as ONE launch (round 4; persist.hip: the recurrence kernel also forms the head's hidden layer, the
 * logits of its vocabulary tiles and the arg-max, and feeds the token back: two more hand-offs per step instead of nine
 * launches).  Same outputs and saved tensors as vag_cgru_attn_decode_seq_fwd(free_run = 1), so the backward entry points
 * are unchanged.  vag_cgru_free_supported: H = 512, E = 256, B <= 64, keys fit the LDS, every workgroup resident.
 * tables: vag_cgru_free_tables_floats floats of scratch (the input projection of every vocabulary entry, emb W3^T, enc W2^T,
 * arg-max candidates), filled here.  logits may be NULL (greedy decoding, V11.py:207-226: only tok is read); c_all / e_all
 * may be NULL then too. */
int vag_cgru_free_supported(int64_t B, int64_t Ts, int64_t Tt, int64_t E, int64_t H, int64_t V);
int64_t vag_cgru_free_tables_floats(int64_t B, int64_t Ts, int64_t Tt, int64_t E, int64_t H, int64_t V);
int vag_cgru_attn_decode_free_fwd(const float* enc, const float* pe, const float* mask, const float* h0, int64_t* tok,
                                  vag_dec_w w, int64_t B, int64_t Ts, int64_t Tt, int64_t E, int64_t H, int64_t V,
                                  float* h2_all, float* c_all, float* e_all, float* ws, const vag_head_w* head,
                                  float p_out, const uint64_t* rng, float* tmid, float* logits, int64_t ldl,
                                  float* tables, vag_stream_t stream);
/* Backward through time.  Inputs: gradients w.r.t. the three outputs (d_h2_all, d_c_all are consumed;
 * d_e_all may be NULL).  Writes d_enc_out (B,Ts,C) (accumulate_enc: adds), d_pe (B,Ts,C), d_h0 (B,H);
 * accumulates the parameter gradients in g (g.emb: pad row untouched). */
int vag_cgru_attn_decode_seq_bwd(const float* enc, const float* pe, const float* mask, const float* h0,
                                 const int64_t* tok, vag_dec_w w, int64_t B, int64_t Ts, int64_t Tt, int64_t E,
                                 int64_t H, int64_t V, const float* h2_all, const float* c_all, const float* e_all,
                                 float* d_h2_all, float* d_c_all, const float* d_e_all, float* ws, float* d_enc_out,
                                 int accumulate_enc, float* d_pe, float* d_h0, vag_dec_g g, float* scratch,
                                 vag_stream_t stream);
int64_t vag_cgru_bwd_scratch_floats(int64_t B, int64_t Ts, int64_t Tt, int64_t E, int64_t H);
/* The same backward in two phases, for callers that overlap them on two streams: _loop is the recurrence (writes
 * d_enc_out, d_pe, d_h0 and leaves the per-step tensors in `scratch`); _weights turns those into the parameter
 * gradients (large products nothing downstream waits for).  ws/scratch must stay untouched in between. */
int vag_cgru_attn_decode_seq_bwd_loop(const float* enc, const float* pe, const float* mask, const float* h0,
                                      const int64_t* tok, vag_dec_w w, int64_t B, int64_t Ts, int64_t Tt, int64_t E,
                                      int64_t H, int64_t V, const float* h2_all, const float* c_all,
                                      const float* e_all, float* d_h2_all, float* d_c_all, const float* d_e_all,
                                      float* ws, float* d_enc_out, int accumulate_enc, float* d_pe, float* d_h0,
                                      float* scratch, vag_stream_t stream);
int vag_cgru_attn_decode_seq_bwd_weights(const float* h0, const int64_t* tok, vag_dec_w w, int64_t B, int64_t Ts,
                                         int64_t Tt, int64_t E, int64_t H, const float* h2_all, const float* c_all,
                                         const float* e_all, const float* d_e_all, float* ws, vag_dec_g g,
                                         float* scratch, vag_stream_t stream);

/* One inference step for N hypotheses (greedy / beam search, models/...V11.py:207-226,259-313).  Hypothesis n
 * attends over source sentence n / rows_per_src (the reference tiles encoder_outputs by beam_size, :253).
 * tok int64[N]; h_in (N,H) -> h_out (N,H), c (N,C), e (N,E).  scratch: vag_cgru_step_scratch_floats(). */
/* `prep`: vag_cgru_prep_floats(H) floats filled by vag_cgru_prepare() once per decode call (derived weights:
 * [attn_h ; gru_2.w_hh] stacked so both products of h1 are one launch, and gru_2.w_ih . context2hid folded: decoding
 * steps and the free-running launch chain read it; the teacher-forced sequence operators apply context2hid and gru_2.w_ih to the
 * keys one after the other instead). */
int64_t vag_cgru_prep_floats(int64_t H);
int vag_cgru_prepare(vag_dec_w w, int64_t H, float* prep, vag_stream_t stream);
int64_t vag_cgru_step_scratch_floats(int64_t N, int64_t Ts, int64_t E, int64_t H);
int vag_cgru_attn_decode_step(const float* enc, const float* pe, const float* mask, int64_t rows_per_src,
                              const int64_t* tok, const float* h_in, vag_dec_w w, const float* prep, int64_t N,
                              int64_t Ts, int64_t E, int64_t H, float* h_out, float* c, float* e, float* alpha,
                              float* scratch, vag_stream_t stream);

/* ---- a5 (head) + a2 loss: layers/NMT_Decoder.py:137-143, models/...V11.py:140,164 --------------------- */
/* t = tanh(W1 h2 + W2 c + W3 e + b1+b2+b3); dropout p_out; logits = t out_w^T + out_b; log_softmax;
 * nll[t,b] = -weight[tgt[b,t]] * logp[tgt[b,t]]  (nn.NLLLoss(weight, reduce=False));
 * loss_mt = mean_b( sum_t nll[t,b] / #nonpad(tgt[b,:]) ).
 * rows = Tt*B time-major; tgt (B,Tt) int64; logits (rows, ldl) with ldl >= V, ldl % 4 == 0 (kept for bwd).
 * logits_ready=1: tmid/logits were already produced (free-running decode), only the loss is computed.
 * Outputs: lse (rows), nll (rows), loss_mt (1), inv_cnt (B). */
int vag_head_ce_seq_fwd(const float* h2_all, const float* c_all, const float* e_all, vag_head_w w,
                        const int64_t* tgt, const float* vocab_weight, int64_t B, int64_t Tt, int64_t E, int64_t H,
                        int64_t V, float p_out, const uint64_t* rng, int logits_ready, float* tmid, float* logits,
                        int64_t ldl, float* lse, float* nll, float* inv_cnt, float* loss_mt, vag_stream_t stream);
/* d_loss: device scalar (gradient of loss_mt).  logits is overwritten with d(logits).  Writes d_h2_all,
 * d_c_all, d_e_all; accumulates g.  scratch: rows*E floats. */
int vag_head_ce_seq_bwd(const float* h2_all, const float* c_all, const float* e_all, vag_head_w w,
                        const int64_t* tgt, const float* vocab_weight, int64_t B, int64_t Tt, int64_t E, int64_t H,
                        int64_t V, float p_out, const uint64_t* rng, const float* tmid, float* logits, int64_t ldl,
                        const float* lse, const float* inv_cnt, const float* d_loss, float* d_h2_all,
                        float* d_c_all, float* d_e_all, vag_head_g g, float* scratch, vag_stream_t stream);
/* vag_head_ce_seq_bwd in two phases (see vag_cgru_attn_decode_seq_bwd_loop): _data writes d_h2_all/d_c_all/d_e_all and
 * leaves d(logits) in `logits` and d(pre-activation) in `scratch` (R*E floats); vag_head_bwd_weights accumulates g. */
int vag_head_ce_seq_bwd_data(vag_head_w w, const int64_t* tgt, const float* vocab_weight, int64_t B, int64_t Tt,
                             int64_t E, int64_t H, int64_t V, float p_out, const uint64_t* rng, const float* tmid,
                             float* logits, int64_t ldl, const float* lse, const float* inv_cnt, const float* d_loss,
                             float* d_h2_all, float* d_c_all, float* d_e_all, float* scratch, vag_stream_t stream);
int vag_head_bwd_weights(const float* h2_all, const float* c_all, const float* e_all, int64_t R, int64_t E, int64_t H,
                         int64_t V, const float* tmid, const float* dlogits, int64_t ldl, const float* dt, vag_head_g g,
                         vag_stream_t stream);
/* Same head producing the log-probabilities themselves (R rows) with a backward from d_logp -- the form the
 * per-step layer API (NMT_Decoder.forward -> logp, layers/NMT_Decoder.py:143) and arbitrary criteria need.
 * tmid (R,E) saved; d_logp (R,ldl) is consumed.  scratch: R*E floats. */
int vag_head_logp_seq_fwd(const float* h2, const float* c, const float* e, vag_head_w w, int64_t R, int64_t E, int64_t H,
                          int64_t V, float p_out, const uint64_t* rng, float* tmid, float* logp, int64_t ldl,
                          vag_stream_t stream);
int vag_head_logp_seq_bwd(const float* h2, const float* c, const float* e, vag_head_w w, int64_t R, int64_t E, int64_t H,
                          int64_t V, float p_out, const uint64_t* rng, const float* tmid, const float* logp,
                          float* d_logp, int64_t ldl, float* d_h2, float* d_c, float* d_e, vag_head_g g, float* scratch,
                          vag_stream_t stream);
/* Single step, inference: logp (N,V) = log_softmax(out(tanh(...))) and argmax (int64[N], may be NULL).
 * scratch: N*E floats. */
int vag_head_logp_step(const float* h2, const float* c, const float* e, vag_head_w w, int64_t N, int64_t E,
                       int64_t H, int64_t V, float* logp, int64_t ldl, int64_t* argmax, float* scratch,
                       vag_stream_t stream);

/* ---- a7/a8: shared-space projections, layers/VSE_Imagine_Enc.py:123-132,138-145; utils/utils.py:6-10 - */
/* out = l2norm(act(x W^T + b)); x (B,K), W (S,K).  y (B,S) = activation output and nrm (B) are saved. */
int vag_img_proj_l2_fwd(const float* x, const float* W, const float* b, int64_t B, int64_t K, int64_t S, int act,
                        float* y, float* nrm, float* out, vag_stream_t stream);
/* d_out is consumed.  d_x may be NULL; g_W, g_b accumulated. */
int vag_img_proj_l2_bwd(const float* x, const float* W, const float* y, const float* nrm, const float* out,
                        float* d_out, int64_t B, int64_t K, int64_t S, int act, float* d_x, float* g_W, float* g_b,
                        vag_stream_t stream);

/* a8 alone: out = x / max(||x||, 1e-12) per row (utils/utils.py:6-10). */
int vag_l2norm_fwd(const float* x, int64_t B, int64_t S, float* nrm, float* out, vag_stream_t stream);
int vag_l2norm_bwd(const float* x, const float* nrm, const float* out, const float* d_out, int64_t B, int64_t S,
                   float* dx, vag_stream_t stream);

/* ---- a6: image-conditioned attention + attended context, VSE_Imagine_Enc.py:29-79,135-137 ------------- */
/* method 0 = 'dot' (score_dot), 1 = 'mlp' (score_mlp; mlp_w (C)).  im_emb (B,S), enc (B,Ts,C), mask (B,Ts).
 * Outputs alpha (B,Ts), ctx (B,C).  ws: vag_imagine_ws_floats() floats kept for backward.
 * 'dot' uses e[b,t] = enc[b,t] . (W_cc^T (W_ec im_emb[b])): identical in exact arithmetic to the reference's
 * bmm(emb2ctx(im), ctx2ctx(enc)^T) and avoids the (B*Ts,C,C) product. */
int64_t vag_imagine_ws_floats(int64_t B, int64_t Ts, int64_t C, int64_t S, int method);
int vag_imagine_attn_ctx_fwd(const float* im_emb, const float* enc, const float* mask, const float* ctx2ctx,
                             const float* emb2ctx, const float* mlp_w, int method, int64_t B, int64_t Ts,
                             int64_t C, int64_t S, float* alpha, float* ctx, float* ws, vag_stream_t stream);
/* d_ctx (B,C) in.  d_enc (B,Ts,C) is written or added to (accumulate_enc); d_im_emb (B,S) written. */
int vag_imagine_attn_ctx_bwd(const float* im_emb, const float* enc, const float* mask, const float* ctx2ctx,
                             const float* emb2ctx, const float* mlp_w, int method, int64_t B, int64_t Ts,
                             int64_t C, int64_t S, const float* alpha, const float* d_ctx, float* ws,
                             float* d_enc, int accumulate_enc, float* d_im_emb, float* g_ctx2ctx,
                             float* g_emb2ctx, float* g_mlp_w, vag_stream_t stream);

/* ---- a9: max-margin ranking losses, losses/PairwiseRankingLoss.py:9-24, ImageRetrievalRankingLoss.py -- */
/* kind 0 = pairwise (both directions), 1 = image retrieval (cost_s only).  im, s (B,S).
 * loss (1).  G (B,B) = d loss / d scores, kept for backward.  scores (B,B) scratch. */
int vag_rank_loss_fwd(const float* im, const float* s, int64_t B, int64_t S, float margin, int kind, float* scores,
                      float* G, float* loss, vag_stream_t stream);
int vag_rank_loss_bwd(const float* im, const float* s, const float* G, const float* d_loss, int64_t B, int64_t S,
                      float* d_im, float* d_s, vag_stream_t stream);

/* ---- next: batch assembly from a device-resident corpus, preprocessing.py:308-384 ---------- */
/* out[i, 0:w] = in[idx[i], 0:w] for int64 token matrices (in: (N, ld) padded with 0).  Image-feature rows use
 * vag_embed_fwd (a row gather of an fp32 matrix). */
int vag_gather_rows_i64(const int64_t* in, int64_t ld, const int64_t* idx, int64_t rows, int64_t w, int64_t* out,
                        vag_stream_t stream);

/* ---- next: retrieval evaluation, utils/im_retrieval_eval.py:4-57 --------------------------- */
/* The reference loops over N queries with one torch.mm + torch.sort each; here: one (N,S)x(S,N) product into
 * `scores` (N,N scratch) and one rank kernel.  ranks[i] = 0-based position of key i in the descending sort of
 * scores[i,:].  t2i: queries = caption embeddings, keys = image embeddings; i2t: the other way round. */
int vag_retrieval_ranks(const float* queries, const float* keys, int64_t N, int64_t S, float* scores, int32_t* ranks,
                        vag_stream_t stream);

/* ---- a2: decoder initial state, models/...V11.py:118 / NMT_Seq2Seq_Beam_V2.py:85 ---------------------- */
/* x = split*ctx + (1-split)*sum_t enc/sum_t mask (ctx NULL: text-only, x = mean);  h0 = tanh(W x + b).
 * xmix (B,C) is saved. */
int vag_dec_init_fwd(const float* enc, const float* mask, const float* ctx, float split, const float* W,
                     const float* b, int64_t B, int64_t Ts, int64_t C, int64_t H, float* xmix, float* h0,
                     vag_stream_t stream);
/* d_h0 consumed.  d_enc written or added; d_ctx (may be NULL) written.  scratch: B*C floats. */
int vag_dec_init_bwd(const float* mask, const float* xmix, const float* h0, float split, const float* W,
                     float* d_h0, int64_t B, int64_t Ts, int64_t C, int64_t H, float* d_enc, int accumulate_enc,
                     float* d_ctx, float* g_W, float* g_b, float* scratch, vag_stream_t stream);

/* ---- a10: beam-search step, models/...V11.py:262-313 -------------------------------------------------- */
/* One expansion for B sentences x k beams over V words (step di >= 0; step 0 expands one hypothesis per sentence).
 * logp (B*k_in, ldl) is read through the reference's penalties (repeat-token suppression :279-280, finished
 * hypotheses may only emit EOS at cost 0 :291-294, inf = -1e5).  nll (B,k) running scores in/out.
 * beam (2*max_len,B,k) int64 history: row di receives the chosen words (:306) and row max_len+di the index of the
 * hypothesis each one extends (:303); the reference's per-step permutation of all earlier rows (:309) is replaced
 * by these back-pointers, resolved once in vag_beam_finish.  h_in (B*k_in,H) -> h_out (B*k,H) re-ordered by
 * back-pointer (:273,:313); n_alive (1) int32 = number of new hypotheses whose word is not EOS (host-sync-free
 * early-exit test).  scratch: vag_beam_scratch_bytes. */
int64_t vag_beam_scratch_bytes(int64_t B, int64_t k, int64_t V, int64_t max_len);
int vag_beam_step(float* logp, int64_t ldl, float* nll, int64_t* beam, int64_t di, int64_t max_len,
                  const float* h_in, float* h_out, int64_t B, int64_t k, int64_t V, int64_t H, int32_t* n_alive,
                  void* scratch, vag_stream_t stream);
/* The same expansion with the step index held in device memory, so that one captured HIP graph serves every step:
 * di_state int32[2] = {di (>= 1 on entry, incremented by the call), 0 (internal arrival counter)}.  A call with
 * di >= max_len does nothing.  tok_out (B*k) int64 (may be NULL) also receives row di, the next step's input words. */
int vag_beam_step_dev(float* logp, int64_t ldl, float* nll, int64_t* beam, int32_t* di_state, int64_t max_len,
                      const float* h_in, float* h_out, int64_t* tok_out, int64_t B, int64_t k, int64_t V, int64_t H,
                      int32_t* n_alive, void* scratch, vag_stream_t stream);
/* Final selection (:315-324) after `steps` calls of vag_beam_step (steps < max_len after an early stop): follow the
 * back-pointers, force EOS in the last row, length-normalise, pick the best hypothesis.
 * out (B,max_len) int64 (0 past the written rows), best_score (B). */
/* The decoding step in its hoisted form (round 4): the keys as gru_2 sees them and the head's share of them are projected once per
 * decode call (vag_cgru_decode_keys: keys = [(W_ih2 W_c2h) enc (B,Ts,3H) | enc W2^T (B,Ts,E)], vag_cgru_decode_keys_floats floats;
 * prep from vag_cgru_prepare, w2 = head W2 (E,C)); a step is then four launches and returns cw (N,E) = W2 c instead of the
 * context c, which vag_head_logp_step_h / vag_head_logits_step_h take in its place.  N <= 256 hypotheses, rows_per_src divides N.
 * scratch: vag_cgru_step_scratch_floats. */
int64_t vag_cgru_decode_keys_floats(int64_t B, int64_t Ts, int64_t E, int64_t H);
int vag_cgru_decode_keys(const float* enc, const float* prep, const float* w2, int64_t B, int64_t Ts, int64_t E, int64_t H,
                         float* keys, vag_stream_t stream);
/* tables (optional, NULL: none): [emb W_ih1^T + b_ih1 (V,3H) | emb W3^T (V,E)] from vag_cgru_decode_tables, once per call
 * (w3 = head W3): a step then has no embedding / input-projection launch, `e` is not produced (may be NULL) and the head reads
 * its share of the embedded token from the table line `tok` picks. */
int64_t vag_cgru_decode_tables_floats(int64_t V, int64_t E, int64_t H);
int vag_cgru_decode_tables(vag_dec_w w, const float* w3, int64_t V, int64_t E, int64_t H, float* tables, vag_stream_t stream);
int vag_cgru_attn_decode_step_h(const float* pe, const float* mask, const float* keys, const float* tables, int64_t V,
                                int64_t rows_per_src, const int64_t* tok, const float* h_in, vag_dec_w w, const float* prep,
                                int64_t N, int64_t Ts, int64_t E, int64_t H, float* h_out, float* cw, float* e, float* alpha,
                                float* scratch, vag_stream_t stream);
int vag_head_logp_step_h(const float* h2, const float* cw, const float* e, const float* tables, const int64_t* tok, vag_head_w w,
                         int64_t N, int64_t E, int64_t H, int64_t V, float* logp, int64_t ldl, int64_t* argmax, float* scratch,
                         vag_stream_t stream);
int vag_head_logits_step_h(const float* h2, const float* cw, const float* e, const float* tables, const int64_t* tok, vag_head_w w,
                           int64_t N, int64_t E, int64_t H, int64_t V, float* logits, int64_t ldl, float* parts, float* scratch,
                           vag_stream_t stream);
/* Beam step on RAW logits (round 4): the vocabulary product of vag_head_logits_step leaves, per row, vag_head_logits_parts_count
 * (max, sum exp) pairs -- the pieces of the row's log-sum-exp -- in `parts` (count, N, 2); the expansion kernel normalises the
 * candidates it reads with them, so no pass over the (B k, V) logits is needed between product and selection (V11.py:276,297).
 * count = 0: the shape is not taken (use vag_head_logp_step + vag_beam_step_dev).  scratch of vag_head_logits_step: 2 N E floats. */
int64_t vag_head_logits_parts_count(vag_head_w w, int64_t N, int64_t E, int64_t V);
int vag_head_logits_step(const float* h2, const float* c, const float* e, vag_head_w w, int64_t N, int64_t E, int64_t H,
                         int64_t V, float* logits, int64_t ldl, float* parts, float* scratch, vag_stream_t stream);
int vag_beam_step_logits_dev(float* logits, int64_t ldl, const float* parts, int64_t nparts, float* nll, int64_t* beam,
                             int32_t* di_state, int64_t max_len, const float* h_in, float* h_out, int64_t* tok_out, int64_t B,
                             int64_t k, int64_t V, int64_t H, int32_t* n_alive, void* scratch, vag_stream_t stream);
int vag_beam_finish(const float* nll, const int64_t* beam, int64_t max_len, int64_t steps, int64_t B, int64_t k,
                    int64_t* out, float* best_score, vag_stream_t stream);

/* ---- a13: optimiser step, train.py:46-49 + nmt_multimodal_beam_DE.py:303-332 -------------------------- */
/* Global-norm clip (clip_grad_norm_, eps 1e-6) fused with Adam over one flat fp32 buffer of n elements split
 * into nseg contiguous segments [seg_off[i], seg_off[i+1]) with their own lr / L2 weight decay (the reference's
 * param groups).  grad_scale multiplies every gradient first (1/world_size after a sum all-reduce).
 * seg_off (nseg+1), seg_lr, seg_wd are HOST arrays (read while enqueuing).  step: device int32 counter,
 * incremented here.  norm_out (1): total gradient norm before clipping.  scratch: VAG_ADAM_SCRATCH_BYTES, 8-byte aligned. */
#define VAG_ADAM_SCRATCH_BYTES 2048
/* A void gradient is never applied (train.py:44-49 holds for every step that IS applied): when the gradient norm is not
 * finite, or a persistent recurrence kernel launched under THIS scratch's guard pair (VAG_ADAM_SCRATCH_GUARD_OFFSET) gave up a wait
 * since the previous call (on a replica
 * of a data-parallel run the give-up reaches every rank as a non-finite entry of the all-reduced gradient), this call leaves
 * p, m, v and *step unchanged, still zeroes g (zero_grad), writes NaN to norm_out and adds one to the uint32 at byte
 * VAG_ADAM_SCRATCH_SKIPPED_OFFSET of the scratch (a host reads it from there whenever it likes; TrainStep.skipped_steps). */
#define VAG_ADAM_SCRATCH_SKIPPED_OFFSET 28
/* The driver's guard pair {void flag, give-up count} (two uint32) lives at this byte offset of the scratch: pass its address as
 * vag_step_cfg.guard (or vag_set_operator_guard) and this call skips exactly the steps whose OWN recurrence launches gave up a
 * wait; the count is the driver's to read and reset (TrainStep.check). */
#define VAG_ADAM_SCRATCH_GUARD_OFFSET 32
/* zero_grad != 0: g is left zeroed (the next step's backward accumulates into it; no separate fill pass).
 * scratch must be zero before the FIRST call; every call leaves it ready for the next one.  Three launches.
 * lr_dev: NULL, or one DEVICE float that multiplies every seg_lr when the kernels run: with seg_lr = the groups' relative
 * rates and *lr_dev = the current learning rate, a captured graph of this call serves every learning rate. */
int vag_clip_adam_flat(float* p, float* g, float* m, float* v, int64_t n, int nseg, const int64_t* seg_off,
                       const float* seg_lr, const float* seg_wd, float clip, float grad_scale, float beta1,
                       float beta2, float eps, int zero_grad, int32_t* step, float* norm_out, void* scratch,
                       const float* lr_dev, vag_stream_t stream);

/* The same optimiser step for ONE contiguous shard [lo, hi) of the flat buffer (lo a multiple of 4): the data-parallel option of
 * SURVEY.md 8e / section 5 for train.py:46-49 -- gradient by reduce-scatter, sharded sum of squares / clip / Adam, parameters back by
 * all-gather (vagnmt_hip.trainer.TrainStep(zero1=True)).  Two calls per step with one all-reduce of ONE double in between:
 *   phase 0: sumsq[0] (device) = sum of squares of g[lo, hi)            -- the caller sums it over the ranks
 *   phase 1: clip coefficient from that global sum (norm_out: the global norm), step counter, Adam on the segments cut to [lo, hi);
 *            zero_grad: all of g[0, n) is zeroed.  Skips (non-finite norm, guard flag) as vag_clip_adam_flat; the step counter
 *            advances on every rank alike because every rank sees the same global sum. */
int vag_clip_adam_shard(float* p, float* g, float* m, float* v, int64_t n, int nseg, const int64_t* seg_off,
                        const float* seg_lr, const float* seg_wd, float clip, float grad_scale, float beta1,
                        float beta2, float eps, int zero_grad, int32_t* step, float* norm_out, void* scratch,
                        const float* lr_dev, int64_t lo, int64_t hi, int phase, double* sumsq, vag_stream_t stream);

/* ---- a2 + a13: the whole training step (train.py:36-51 around models/...V11.py:82-168 and
 * NMT_Seq2Seq_Beam_V2.py:58-113) as ONE call: every operator above in the order autograd would run them, on one
 * caller-owned workspace.  Gradients are ACCUMULATED into g (keep it zeroed between steps: vag_clip_adam_flat with
 * zero_grad does).  The per-operator entry points stay the public API for the reference's module-level calls
 * (model.forward + loss.backward()); this is what the step driver (vagnmt_hip/trainer.py) replays from a HIP graph. */
typedef struct {
    const float* enc_emb;                 /* encoder.embedding.weight (Vs,Es) */
    vag_gru_w enc_fw, enc_bw;             /* encoder.gru.*_l0 / *_l0_reverse */
    /* vse_imagine.* -- all NULL for the text-only model (NMT_Seq2Seq_Beam_V2) */
    const float *im_w, *im_b, *txt_w, *txt_b, *ctx2ctx, *emb2ctx, *mlp_w;
    const float *ini_w, *ini_b;           /* decoderini */
    const float* attn_e;                  /* decoder.attn.attn_e.weight (C,C) */
    vag_dec_w dec;
    vag_head_w head;
} vag_model_w;
typedef struct {
    float* enc_emb;
    vag_gru_g enc_fw, enc_bw;
    float *im_w, *im_b, *txt_w, *txt_b, *ctx2ctx, *emb2ctx, *mlp_w;
    float *ini_w, *ini_b;
    float* attn_e;
    vag_dec_g dec;
    vag_head_g head;
} vag_model_g;
typedef struct {
    int64_t B, Ts, Tt, Es, Et, H, S, I, V, ldl;   /* Es/Et: source/target embedding size; ldl = ceil4(V) */
    int32_t multimodal;                   /* 1: V11 (image branch + ranking loss), 0: text-only V2 */
    int32_t attn_method;                  /* imagine attention: 0 'dot', 1 'mlp' */
    int32_t activation_vse;               /* tanh on the shared-space projections */
    int32_t rank_kind;                    /* 0 pairwise, 1 image retrieval, -1 no criterion_vse (loss_vse = 0) */
    int32_t free_run;                     /* 0 teacher forcing, 1 feed back the argmax (V11.py:148-160) */
    int32_t storage;                      /* 0: everything fp32.  1 (BASELINE configs[4], "fp16"): what the recurrences stream at
                                           * every time step -- their weights and the attention keys / projected keys -- is
                                           * kept as fp16 in HBM; accumulation, master weights, recurrent state, saved gates and
                                           * all gradients stay fp32.  Teacher-forced steps only; needs `derived` built with
                                           * with_fp16 and H % 8 == 0 */
    float margin, loss_w, init_split, p_emb, p_ctx, p_out;
    int32_t loss_ring;                    /* R > 0: `losses` holds 4 + 4 R floats; the forward phase also stores {loss, loss_mt, loss_vse}
                                           * of its n-th execution (n kept in the bit pattern of losses[3]) at losses[4 + 4 (n % R)]:
                                           * a driver that replays captured graphs hands out results that stay valid for R steps
                                           * without a copy launch per step.  0: losses is 4 floats */
    void* guard;                          /* NULL, or two caller-owned DEVICE uint32 {void flag, give-up count} (zero before the first
                                           * call): the persistent recurrence kernels of THIS call report a give-up there and nowhere
                                           * else, so several drivers on one device cannot void each other's steps.  The step driver
                                           * passes (char*)adam_scratch + VAG_ADAM_SCRATCH_GUARD_OFFSET, the words vag_clip_adam_flat
                                           * reads.  NULL: the calling thread's operator guard (vag_set_operator_guard), else the
                                           * process-wide pair no optimiser reads */
} vag_step_cfg;
/* phases: bit 0 forward (losses[0..2] = loss, loss_mt, loss_vse), bit 1 backward down to the encoder states (final for
 * every gradient except the encoder's), bit 2 the encoder's backward.  A data-parallel driver all-reduces the first
 * gradient bucket while phase 4 runs.  Phase 2 may also be called as its two halves: 16 = output head + decoder (final for
 * the head's, the decoder's and attn_e's gradients), 32 = visual grounding + initial state (final for vse_imagine.* and
 * decoderini.*), for a driver that cuts the gradient into three buckets (phases 1|16, then 32, then 4).  rng: {seed, step} (step is advanced by the forward phase) or NULL (no dropout).
 * derived: vag_derived_floats(H) floats kept current with vag_derive_weights() after every optimiser step, or NULL
 * (derived weights are then rebuilt inside the call).  ws: vag_step_ws_floats(cfg) floats, kept between phases. */
int64_t vag_step_ws_floats(const vag_step_cfg* cfg);
int64_t vag_step_ws_offset(const vag_step_cfg* cfg, int which);
int vag_train_step(const vag_step_cfg* cfg, const vag_model_w* w, const vag_model_g* g, const int64_t* src,
                   const int32_t* lengths, const int64_t* tgt, const float* im, const float* vocab_weight, uint64_t* rng,
                   const float* derived, float* ws, float* losses, int phases, vag_stream_t stream);
/* The per-operator entry points (vag_bigru_seq_*, vag_attn_keys_proj, vag_cgru_attn_decode_seq_*) normally rebuild the
 * derived weights inside their workspaces and use fp32 storage.  This sets, for the CALLING THREAD until changed, the
 * driver-owned derived buffer they should read instead and the storage mode (vag_step_cfg.storage); vag_train_step does the
 * same for the duration of its call.  derived = NULL, storage = 0 restores the defaults. */
int vag_set_operator_context(const float* derived, int storage);
/* Process-wide debug / tuning options by name, for the parity tests and tuning scripts (none is needed in normal use):
 * "gemm_f32mfma" (1: every product on the exact f32-input MFMA kernels), "gemm_nogroup", "gemm_force_tile" +
 * "gemm_force_splitk", "gemm_debug", "head_chunk" (rows per chunk of the output head; -1 automatic, 0 never),
 * "head_fuse", "head_bf16_grads", "head_bf16_dlogits" (2-byte mode, chunked head: d(logits) as bf16), "s16_one_plane"
 * (2-byte mode of vag_train_step: one-plane products), "persistent" / "persistent_dec_bwd" (0: launch chains instead of the
 * one-launch recurrence kernels), "persist_timing" (vag_recurrence_time), "dec_stamps" / "dec_bwd_stamps" (device address
 * for phase timestamps of the decoder kernels).  Returns VAG_EINVAL for an unknown name. */
int vag_set_option(const char* name, int64_t value);
/* Up to four contiguous device byte ranges copied by one launch (src[i] -> dst[i], bytes[i]; host arrays): a batch's
 * src / lengths / tgt / image rows into the step driver's static input buffers. */
int vag_copy4(const void* const* src, void* const* dst, const int64_t* bytes, int n, vag_stream_t stream);
/* Weights derived from the parameters alone ([attn_h; gru_2.w_hh] stacked and transposed, gru_1.w_hh^T, both encoder w_hh^T;
 * until round 4 also gru_2.w_ih . context2hid, which the training step no longer reads): per optimiser step, not per training step.  with_fp16: also the fp16 copies of the
 * recurrent matrices the 2-byte storage mode reads (vag_step_cfg.storage = 1; H % 8 == 0). */
int64_t vag_derived_floats(int64_t H);
int vag_derive_weights(vag_dec_w w, const float* enc_whh_fw, const float* enc_whh_bw, int64_t H, int with_fp16,
                       float* derived, vag_stream_t stream);

/* ---- persistent recurrences (round 3, persist.hip) -------------------------------------------------------- */
/* The teacher-forced decoder recurrence (every step's gru_1 cell, attention query / scores / softmax, projected context and
 * gru_2 cell: layers/NMT_Decoder.py:121-129 x models/...V11.py:138-146) as ONE launch: recurrent weights in registers as
 * bf16x3 planes, the attention keys of a row tile in LDS, four workgroup-to-workgroup exchanges per step (write-through
 * stores + counters + sc1 loads).  vag_cgru_attn_decode_seq_fwd uses it whenever vag_recurrence_supported(1, ...) says so
 * (H = 512, B <= 64, keys fit the LDS); this entry point runs it alone (bench.py times it; parity: tests/test_gpu_round3.py).
 * kind: 0 = the bi-GRU encoder kernels (H in {256, 512, 1024}, 2 * ceil(B/16) * H/16 workgroups <= CUs), 1 = the decoder.
 * xp1 (Tt,B,3H) = W_ih1 e_t + b_ih1; wcat (C+3H,H) = [attn_h; gru_2.w_hh], bcat (C+3H) = [0; gru_2.b_hh]; encwp (B,Ts,3H) =
 * (gru_2.w_ih context2hid) enc; outputs as vag_cgru_attn_decode_seq_fwd saves them: h1 (Tt,B,H), g1 / g2 (Tt,4,B,H),
 * qhp (Tt,B,C+3H), alpha (Tt,B,Ts), h2_all (Tt,B,H); psc (Tt,4,B,Ts) floats (the score accumulators, four copies) and sync (vag_recurrence_sync_words 32-bit
 * words) are scratch. */
int vag_recurrence_supported(int kind, int64_t B, int64_t Ts, int64_t Tt, int64_t H);
/* The persistent kernels wait on each other inside one launch, which needs every workgroup resident at once (one per CU;
 * vag_recurrence_supported checks the CU count).  Their waits are bounded: on a device where that does not hold they give
 * up after ~1 s instead of hanging, the launch's results are then void -- and never applied: the give-up sets the launch's guard
 * flag (vag_step_cfg.guard, VAG_ADAM_SCRATCH_GUARD_OFFSET), which vag_clip_adam_flat reads on the device (the optimiser step is
 * skipped, see VAG_ADAM_SCRATCH_SKIPPED_OFFSET) and which
 * the last launch of vag_train_step's backward turns into a non-finite gradient entry, so that every replica of a
 * data-parallel run skips the same step after the all-reduce.  This returns how many waits gave up PROCESS-WIDE since the last
 * call (0 in a healthy run; every driver's and every unguarded launch's) and resets the count; it synchronises the device --
 * call it at checkpoints, not per step.
 * vag_set_option("persist_spin_limit", n): polls before a wait gives up (0 = default 2^19; tests force a give-up with 1). */
int vag_persistent_timeouts(void);
/* Operators called one by one (the module API under torch.autograd, decoding) have no vag_step_cfg: this sets the guard pair
 * (see vag_step_cfg.guard) of every persistent launch the CALLING THREAD enqueues until changed; NULL = the process-wide pair. */
int vag_set_operator_guard(void* guard);
/* Measurement: with vag_set_option("persist_timing", 1) every EAGER launch (not inside a stream capture) of a recurrence
 * kernel is bracketed by HIP events on its stream.  This returns the accumulated kernel time and launch count of one kind
 * (0 encoder forward, 1 decoder forward, 2 encoder backward, 3 decoder backward) since the last call and resets them; it
 * waits for the last bracketed launch.  bench.py derives the per-family roofline rows from it. */
int vag_recurrence_time(int kind, double* ms_total, int* launches);
/* Host-only (no device is touched): the plan the library makes for one grouped launch of n (<= 12) large products C_i (M_i x
 * N_i) over K_i -- products an operator issues between its group brackets go out as ONE grid per operand layout.  The chip
 * runs 512 blocks of 128 x 128 at a time and hands them out in index order; the plan is split[i] = number of K slices of
 * product i (accumulate[i] != 0: the slices add into C; 0: C is overwritten, slicing costs a fill launch) and order[] = the
 * products sorted by slice length, longest first, chosen by simulating that schedule.  Exposed for tests and tuning. */
int vag_gemm_group_plan(int n, const int64_t* M, const int64_t* N, const int64_t* K, const int* accumulate, int* split, int* order);
int64_t vag_recurrence_sync_words(int kind, int64_t B, int64_t T);
int vag_cgru_recurrence_fwd(const float* pe, const float* mask, const float* h0, const float* xp1, vag_dec_w w, const float* wcat,
                            const float* bcat, const float* encwp, int64_t B, int64_t Ts, int64_t Tt, int64_t H, float* h1,
                            float* g1, float* qhp, float* alpha, float* h2_all, float* g2, float* psc, void* sync,
                            vag_stream_t stream);

/* ---- data-parallel gradient exchange over RCCL (SURVEY 8b "C1", 8e) -------------------------------------------------------
 * One process per GPU, one communicator per process.  Rank 0 draws an id (128 opaque bytes) and ships it to the other
 * ranks by any channel (the Python host uses the torch.distributed store); every rank then calls vag_comm_init with the
 * same id -- a collective that binds the communicator to the CURRENT device.  vag_comm_allreduce sums `n` floats in place
 * across the ranks (ncclAllReduce, ncclSum -- the caller scales by 1/N; vag_clip_adam_flat's grad_scale does) on `stream`;
 * like every call here it only enqueues, so it can be captured with the step's kernels.  The reference has no
 * distributed backend (nmt_multimodal_beam_DE.py:277-282 leaves nn.DataParallel commented out); this is the exchange
 * SURVEY 8e adds.  librccl is loaded on the first vag_comm_* call (-38 = ENOSYS if it cannot be); RCCL's own errors come
 * back as 10000 + ncclResult_t. */
#define VAG_COMM_ID_BYTES 128
typedef struct vag_comm_s* vag_comm_t;
int vag_comm_unique_id(void* id);
int vag_comm_init(vag_comm_t* comm, int nranks, int rank, const void* id);
int vag_comm_allreduce(vag_comm_t comm, float* buf, int64_t n, vag_stream_t stream);
int vag_comm_size(vag_comm_t comm);
int vag_comm_destroy(vag_comm_t comm);

/* ---- dropout helpers ---------------------------------------------------------------------------------- */
/* which: 1 encoder-embedding (Ts,B,E), 2 encoder-context (B,Ts,2H), 3 decoder-output (Tt,B,E). */
int vag_dropout_mask(const uint64_t* rng, int which, int64_t n, float p, float* out, vag_stream_t stream);
int vag_rng_advance(uint64_t* rng, vag_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VAG_NMT_H */
