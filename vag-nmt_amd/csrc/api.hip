// C ABI of libvagnmt.so (include/vag_nmt.h): host-side orchestration of the kernel launches for every
// operator of the VAG-NMT hot path.  No allocation, no host synchronisation: every function only enqueues
// work on the caller's stream, so whole training steps can be captured into a HIP graph.
#include "../../include/vag_nmt.h"
#include "kernels.h"
#include <cstring>

#define S_(x) reinterpret_cast<hipStream_t>(x)

// Fills and copies are kernels (not hipMemset/hipMemcpy nodes) so a captured step is a pure chain of kernel nodes.
static inline int zero_async(void* p, size_t bytes, hipStream_t s) {
    if (bytes == 0) return VAG_OK;
    return vag_axpy_launch(0.f, reinterpret_cast<const float*>(p), reinterpret_cast<float*>(p), (int64_t)(bytes / 4), 2, s);
}
static inline int copy_async(void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (bytes == 0) return VAG_OK;
    return vag_axpy_launch(1.f, reinterpret_cast<const float*>(src), reinterpret_cast<float*>(dst), (int64_t)(bytes / 4), 0, s);
}
// y = act(x W^T + b): small-M kernel for a single time step, tiled kernel otherwise.
static int linear_fwd(int64_t M, int64_t N, int64_t K, const float* x, int64_t ldx, const float* W, const float* bias,
                      int act, float* y, int64_t ldy, hipStream_t s) {
    if (M <= 256) return vag_skinny_launch(M, N, K, x, ldx, W, K, bias, nullptr, 0, y, ldy, act, s);
    return vag_gemm_launch(M, N, K, 1.f, x, ldx, 1, W, 1, K, 0.f, y, ldy, bias, act, s);
}
// C (+)= A^T B with A (R,M) lda, B (R,N) ldb: weight gradients  g_W[m,n] += sum_r dY[r,m] X[r,n]
// g_bias (optional): += sum_r A[r,:], the bias gradient that goes with this weight gradient -- taken from the A tiles inside the
// product (GemmArgs::rowsum), no second pass over dY
static int gemm_tn_acc(int64_t M, int64_t N, int64_t R, const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
                       int64_t ldc, hipStream_t s, float* g_bias = nullptr) {
    if (R == 0) return VAG_OK;
    return vag_gemm_launch(M, N, R, 1.f, A, 1, lda, B, ldb, 1, 1.f, C, ldc, nullptr, VAG_ACT_NONE, s, 0, g_bias);
}
// C = beta*C + A B with A (M,K) lda, B (K,N) ldb: data gradients  dX = dY W
static int gemm_nn(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb, float beta,
                   float* C, int64_t ldc, hipStream_t s) {
    if (M <= 128 && (beta == 0.f || beta == 1.f)) return vag_skinny_nn_launch(M, N, K, A, lda, B, ldb, beta, C, ldc, s);
    return vag_gemm_launch(M, N, K, 1.f, A, lda, 1, B, ldb, 1, beta, C, ldc, nullptr, VAG_ACT_NONE, s);
}

// Derived weights (functions of the parameters only: stacked / folded / transposed matrices the recurrences read).
// The stand-alone operators rebuild them per call inside their workspaces; a step driver that owns the optimiser
// refreshes them once per optimiser step (vag_derive_weights) and points the operators at that copy for the duration of a
// call through this thread-local (same pattern as the grouped-GEMM bracket: one host thread drives a stream).
static thread_local const float* g_derived = nullptr;
// vag_train_step's prologue launch has zeroed the head's tmid and the encoder's dx of this step (step.hip): the two operators
// that accumulate into them from grouped products skip their own fill launch
static thread_local bool g_step_zeroed = false;
void vag_step_set_zeroed(bool v) { g_step_zeroed = v; }
// ... and has already embedded the decoder's input tokens of every step (teacher-forced form) into e_all
static thread_local bool g_step_gathered = false;
void vag_step_set_gathered(bool v) { g_step_gathered = v; }
void vag_set_derived_override(const float* d) { g_derived = d; }
// 2-byte storage mode of the step driver (vag_step_cfg.storage = 1): the tensors the recurrences stream at every time step
// -- their weights (fp16 copies in the derived buffer) and the attention keys pe / projected keys encwp -- are fp16 in
// memory; every product still accumulates in fp32, master weights, recurrent state, saved gates and all gradients stay fp32.
static thread_local bool g_store16 = false;
thread_local bool g_step_poison_inject = false;
void vag_set_store16(bool on) { g_store16 = on; vag_gemm_set_planes(on ? 2 : 3); }
const float* vag_get_derived_override() { return g_derived; }
bool vag_get_store16() { return g_store16; }
static inline const float* as_f(const vag_half* p) { return reinterpret_cast<const float*>(p); }
// Row chunk of the output head (0 = whole sequence at once).  With a chunk set the (Tt*B, V) logits are never formed as a
// whole: forward computes them chunk by chunk for the log-sum-exp / NLL, backward RECOMPUTES each chunk, turns it into
// d(logits) in place and consumes it with the two products that need it -- the chunk (sized to stay inside the 256 MB
// Infinity Cache) is the only logits storage that is touched.  Set by the step driver for large Tt*B*V (configs[4]).
static thread_local int64_t g_head_chunk = 0;
void vag_set_head_chunk(int64_t rows) { g_head_chunk = rows > 0 ? rows : 0; }
// With a chunk set AND a backward that is known to follow in the same call (the step driver with phases 1|2), the forward
// finishes each chunk completely: a row's log-sum-exp needs only that row, and d(loss)/d(loss_mt) = w_mt and 1/count are
// known before the step starts, so d(logits) of the chunk, its share of d(tmid), of g(out.weight) and of g(out.bias) are
// produced while the chunk is still on the die -- nothing is recomputed and the backward starts at d(tmid).
struct HeadFuse { const vag_head_g* g; const float* d_loss; float* dt; bool done; };
static thread_local HeadFuse g_head_fuse = {nullptr, nullptr, nullptr, false};
void vag_set_head_fuse(const vag_head_g* g, const float* d_loss, float* dt) { g_head_fuse = HeadFuse{g, d_loss, dt, false}; }

// The two vocabulary-sized gradient products of the head, d(tmid) = d(logits) out.weight and g(out.weight) += d(logits)^T
// tmid.  2-byte storage mode: plain bf16 operands, one MFMA product instead of three (fp32 accumulation) -- gradients of
// a softmax over V classes are sums of thousands of small terms whose 2^-9 rounding errors average out; the forward
// logits keep the two-plane product.  VAG_HEAD_BF16_GRADS=0 keeps two planes here too.
static bool head_grads_one_plane() {
    return vag_opt().head_bf16_grads != 0 && g_store16;
}
// dl16 (optional): the same d(logits) chunk as its producer stored it in bf16 (vag_ce_bwd_colsum_launch's out16), row stride ldl
static int head_dt_gemm(int64_t R, int64_t E, int64_t V, const float* dlogits, int64_t ldl, const float* out_w, float* dt,
                        hipStream_t s, const void* dl16 = nullptr) {
    if (dl16) return vag_gemm_launch_planes(1, R, E, V, 1.f, reinterpret_cast<const float*>(dl16), ldl, 1, out_w, E, 1, 0.f, dt, E, s, 1);
    if (head_grads_one_plane() && R > 128) return vag_gemm_launch_planes(1, R, E, V, 1.f, dlogits, ldl, 1, out_w, E, 1, 0.f, dt, E, s);
    return gemm_nn(R, E, V, dlogits, ldl, out_w, E, 0.f, dt, E, s);
}
static int head_outw_gemm(int64_t V, int64_t E, int64_t R, const float* dlogits, int64_t ldl, const float* tmid, float* g_out_w,
                          hipStream_t s, const void* dl16 = nullptr) {
    if (R == 0) return VAG_OK;
    if (dl16) return vag_gemm_launch_planes(1, V, E, R, 1.f, reinterpret_cast<const float*>(dl16), 1, ldl, tmid, E, 1, 1.f, g_out_w, E, s, 1);
    if (head_grads_one_plane() && R > 128) return vag_gemm_launch_planes(1, V, E, R, 1.f, dlogits, 1, ldl, tmid, E, 1, 1.f, g_out_w, E, s);
    return gemm_tn_acc(V, E, R, dlogits, ldl, tmid, E, g_out_w, E, s);
}

// 2-byte storage mode, chunked head: the bf16 copy of a chunk's d(logits) lives behind the chunk inside the (whole-sequence
// sized) logits buffer; NULL when the mode is off, the chunk is small, or there is no room behind it.
static void* head_dl16_slot(float* logits, int64_t ldl, int64_t R, int64_t CH, int64_t E) {
    if (!head_grads_one_plane() || CH <= 128 || vag_opt().head_bf16_dlogits == 0) return nullptr;
    // a bf16 A operand exists only for the 128 x 128 one-plane kernel (gemm.hip): narrow embeddings (the cost model then picks
    // 64-wide tiles), the f32-MFMA switch and a forced tile keep the fp32 in-place d(logits)
    if (E <= 64 || vag_opt().gemm_f32mfma != 0 || vag_opt().gemm_force_tile != 0) return nullptr;
    if (CH * 3 > R * 2) return nullptr;                      // CH * ldl floats of chunk + CH * ldl / 2 floats of bf16 must fit R * ldl
    return logits + CH * ldl;
}

VagOptions& vag_opt() {
    static VagOptions o;
    return o;
}

extern "C" {

int vag_version(void) { return 310; }      // 310: vag_clip_adam_shard, the slab scratch inside vag_step_ws_floats (round 6)

// Debug / tuning options by name (common.h: VagOptions); process-wide, takes effect for calls enqueued afterwards.
int vag_set_option(const char* name, int64_t value) {
    VAG_CHECK_ARG(name != nullptr);
    VagOptions& o = vag_opt();
    const struct { const char* n; int* p; } ints[] = {
        {"gemm_f32mfma", &o.gemm_f32mfma}, {"gemm_slabs", &o.gemm_slabs}, {"gemm_nogroup", &o.gemm_nogroup}, {"gemm_big", &o.gemm_big}, {"gemm_force_tile", &o.gemm_force_tile},
        {"gemm_force_splitk", &o.gemm_force_splitk}, {"head_fuse", &o.head_fuse},
        {"head_bf16_grads", &o.head_bf16_grads}, {"persistent", &o.persistent}, {"persistent_dec_bwd", &o.persistent_dec_bwd}, {"persistent_enc_bwd", &o.persistent_enc_bwd}, {"free_persistent", &o.free_persistent}, {"attn_dot_reg", &o.attn_dot_reg},
        {"persist_timing", &o.persist_timing}, {"s16_one_plane", &o.s16_one_plane}, {"leaf_queue", &o.leaf_queue}, {"attn_row", &o.attn_row}, {"dec_xcd_map", &o.dec_xcd_map}, {"loss_ride", &o.loss_ride}, {"step_fork", &o.step_fork},
        {"head_bf16_dlogits", &o.head_bf16_dlogits}};
    for (const auto& e : ints)
        if (strcmp(name, e.n) == 0) { *e.p = (int)value; return VAG_OK; }
    if (strcmp(name, "head_chunk") == 0) { o.head_chunk = value; return VAG_OK; }
    if (strcmp(name, "persist_spin_limit") == 0) { o.persist_spin_limit = value; return VAG_OK; }
#ifdef VAG_LAB          // lab build only (make LAB=1): phase timestamps of the persistent decoder kernels, per-product choice printing
    if (strcmp(name, "dec_stamps") == 0) { o.dec_stamps = value; return VAG_OK; }
    if (strcmp(name, "dec_bwd_stamps") == 0) { o.dec_bwd_stamps = value; return VAG_OK; }
    if (strcmp(name, "gemm_debug") == 0) { o.gemm_debug = (int)value; return VAG_OK; }
    if (strcmp(name, "gemm_planes") == 0) { vag_gemm_set_planes((int)value); return VAG_OK; }      // calling thread: 3, 2, 1, 11
#endif
    return VAG_EINVAL;
}

// Operator-level access to what vag_train_step sets up for itself: the derived-weights buffer and the storage mode the
// per-operator entry points use ON THE CALLING THREAD until changed (derived NULL / storage 0 = the defaults).
int vag_set_operator_context(const float* derived, int storage) {
    VAG_CHECK_ARG(storage == 0 || (storage == 1 && derived));
    g_derived = derived;
    vag_set_store16(storage == 1);
    return VAG_OK;
}

int vag_gemm_f32(int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t sam, int64_t sak, const float* B,
                 int64_t sbk, int64_t sbn, float beta, float* C, int64_t ldc, const float* bias, int act,
                 vag_stream_t stream) {
    return vag_gemm_launch(M, N, K, alpha, A, sam, sak, B, sbk, sbn, beta, C, ldc, bias, act, S_(stream));
}

int vag_linear_fwd(int64_t M, int64_t N, int64_t K, const float* x, const float* W, const float* bias, int act, float* y,
                   vag_stream_t stream) {
    VAG_CHECK_ARG(x && W && y && M >= 0 && N > 0 && K > 0);
    return linear_fwd(M, N, K, x, K, W, bias, act, y, N, S_(stream));
}

int vag_linear_bwd(int64_t M, int64_t N, int64_t K, const float* x, const float* W, const float* y, float* dy, int act,
                   float* d_x, int accumulate_dx, float* g_W, float* g_b, vag_stream_t stream) {
    hipStream_t s = S_(stream);
    VAG_CHECK_ARG(x && W && dy && M >= 0 && N > 0 && K > 0 && (!act || y));
    if (M == 0) return VAG_OK;
    if (act) VAG_TRY(vag_tanh_bwd_launch(y, dy, dy, M * N, nullptr, 0, 0.f, s));
    if (d_x) VAG_TRY(gemm_nn(M, K, N, dy, N, W, K, accumulate_dx ? 1.f : 0.f, d_x, K, s));
    if (g_W) VAG_TRY(gemm_tn_acc(N, K, M, dy, N, x, K, g_W, K, s));
    if (g_b) VAG_TRY(vag_colsum_launch(dy, M, N, N, g_b, s));
    return VAG_OK;
}

int vag_embed_fwd(const int64_t* idx, int64_t n, const float* W, int64_t E, float* out, vag_stream_t stream) {
    return vag_embed_gather_launch(idx, 1, 0, n, 1, W, E, out, nullptr, 0, 0.f, S_(stream));
}
int vag_embed_bwd(const int64_t* idx, int64_t n, const float* d_out, int64_t E, float* g_W, vag_stream_t stream) {
    return vag_embed_scatter_launch(idx, 1, 0, n, 1, d_out, E, g_W, nullptr, 0, 0.f, S_(stream));
}

// Layout of the derived-weights buffer: [wcat | bcat | wp] (= CgruPrep), then the transposes the backward recurrences
// read: wcatT (H, C+3H), whh1T (H, 3H), encT (2 x (H, 3H): forward / reverse encoder W_hh^T).
struct DerivedW {
    float *prep, *wcatT, *whh1T, *encT;
    // fp16 copies for the 2-byte storage mode (element counts in halves; each region starts 256-byte aligned):
    // wcat16 (Q,H), wcatT16 (H,Q), whh1_16 (3H,H), whh1T16 (H,3H), enc16 (2 x (3H,H)), encT16 (2 x (H,3H))
    vag_half *wcat16, *wcatT16, *whh1_16, *whh1T16, *enc16, *encT16;
    int64_t total;
};
static int64_t cgru_prep_total(int64_t H);
static DerivedW derived_layout(float* p, int64_t H) {
    DerivedW w;
    const int64_t C = 2 * H, Q = C + 3 * H;
    int64_t o = 0;
    auto take = [&](int64_t n) { float* q = p ? p + o : nullptr; o += (n + 63) & ~63ll; return q; };
    auto take16 = [&](int64_t nh) { return reinterpret_cast<vag_half*>(take((nh + 1) / 2)); };
    w.prep = take(cgru_prep_total(H)); w.wcatT = take(Q * H); w.whh1T = take(3 * H * H); w.encT = take(2 * 3 * H * H);
    w.wcat16 = take16(Q * H); w.wcatT16 = take16(Q * H); w.whh1_16 = take16(3 * H * H); w.whh1T16 = take16(3 * H * H);
    w.enc16 = take16(2 * 3 * H * H); w.encT16 = take16(2 * 3 * H * H);
    w.total = o;
    return w;
}

// =====================================================================================================
// bi-GRU encoder
// =====================================================================================================
struct BiGruWs {
    float *x, *xp, *hst, *gates, *dgh, *carry, *dx, *whhT, *gx;
    unsigned *sync, *sync_b;             // counters of the persistent recurrence kernels (persist.hip): forward, backward
    int64_t total;
};
static BiGruWs bigru_ws(float* ws, int64_t B, int64_t Ts, int64_t E, int64_t H) {
    BiGruWs w;
    int64_t o = 0;
    auto take = [&](int64_t n) { float* p = ws ? ws + o : nullptr; o += (n + 63) & ~63ll; return p; };
    w.x = take(Ts * B * E);              // embedded (+dropout) input, time-major (Ts,B,E)
    w.xp = take(Ts * B * 6 * H);         // input projections [fwd 3H | rev 3H]; reused as d_xp in backward
    w.hst = take(2 * (Ts + 1) * B * H);  // [dir][step][B][H] hidden states in processing order, step 0 = zeros
    w.gates = take(2 * Ts * 4 * B * H);  // [dir][step][r,z,n,hn][B][H]
    w.dgh = take(2 * Ts * B * 3 * H);    // backward: [dir][step][B][3H]
    w.carry = take(4 * B * H);           // backward: [dir][2][B][H]
    w.dx = take(Ts * B * E);             // backward: d(embedded input)
    w.whhT = take(2 * 3 * H * H);        // backward: W_hh^T per direction (H,3H)
    w.gx = take(3 * Ts * B * H);         // backward, 2-byte mode one-launch kernel: [dir][step][B][3H] fp16 copy of dgh for the exchange
    w.sync = reinterpret_cast<unsigned*>(take(vag_enc_persistent_sync_words(B, Ts)));
    w.sync_b = reinterpret_cast<unsigned*>(take(vag_enc_persistent_sync_words(B, Ts)));      // (adjacent: one range to zero)
    w.total = o;
    return w;
}
int64_t vag_bigru_ws_floats(int64_t B, int64_t Ts, int64_t E, int64_t H) { return bigru_ws(nullptr, B, Ts, E, H).total; }

int vag_bigru_seq_fwd(const int64_t* src, const int32_t* lengths, const float* emb, vag_gru_w fw, vag_gru_w bw, float p_emb,
                      float p_ctx, const uint64_t* rng, int64_t B, int64_t Ts, int64_t E, int64_t H, float* enc,
                      float* mask, float* ws, vag_stream_t stream) {
    hipStream_t s = S_(stream);
    VAG_CHECK_ARG(src && lengths && emb && enc && mask && ws && B > 0 && Ts > 0 && E > 0 && H > 0);
    VAG_CHECK_ARG(E % 4 == 0 && H % 4 == 0 && aligned16(ws) && aligned16(enc));
    VAG_CHECK_ARG(fw.w_ih && fw.w_hh && fw.b_ih && fw.b_hh && bw.w_ih && bw.w_hh && bw.b_ih && bw.b_hh);
    BiGruWs w = bigru_ws(ws, B, Ts, E, H);
    const int64_t R = Ts * B;
    VAG_TRY(vag_embed_gather_launch(src, 1, Ts, Ts, B, emb, E, w.x, rng, VAG_DROP_ENC_EMB, p_emb, s, mask));    // (+ the source mask)
    VagGemmGroup grp0;              // both directions' input projections: one grouped launch
    VAG_TRY(vag_gemm_launch(R, 3 * H, E, 1.f, w.x, E, 1, fw.w_ih, 1, E, 0.f, w.xp, 6 * H, fw.b_ih, 0, s));
    VAG_TRY(vag_gemm_launch(R, 3 * H, E, 1.f, w.x, E, 1, bw.w_ih, 1, E, 0.f, w.xp + 3 * H, 6 * H, bw.b_ih, 0, s));
    VAG_TRY(grp0.end(s));
    const int64_t BH = B * H;
    const bool s16 = g_store16 && g_derived;
    VAG_CHECK_ARG(!g_store16 || (g_derived && H % 8 == 0));
    if (!s16 && vag_opt().persistent && vag_enc_persistent_ok(B, Ts, H)) {
        // the whole recurrence, both directions, in ONE launch (persist.hip): W_hh stays in registers for all Ts steps
        // (the context dropout is applied as the kernel writes enc: no separate pass)
        return vag_enc_fwd_persistent_launch(w.xp, fw.w_hh, bw.w_hh, fw.b_hh, bw.b_hh, lengths, w.hst, w.gates, enc, w.sync, rng,
                                             p_ctx, B, Ts, H, s);
    }
    const vag_half* w16 = s16 ? derived_layout(const_cast<float*>(g_derived), H).enc16 : nullptr;
    if (s16 && vag_opt().persistent && vag_enc_wide16_ok(B, Ts, H) && B >= 64) {
        // 2-byte mode, wide batches: one launch, the fp16 weight slice of a workgroup in registers, four row tiles through
        // it per step; the fp16 copy of the states that the workgroups exchange lives in the backward's (still unused) dgh
        // (the context dropout is applied as the kernel writes enc, as in the fp32 kernel: no separate pass)
        return vag_enc_fwd_wide16_launch(w.xp, w16, w16 + 3 * H * H, fw.b_hh, bw.b_hh, lengths, w.hst, w.gates, enc,
                                         reinterpret_cast<vag_half*>(w.dgh), w.sync, rng, p_ctx, B, Ts, H, s);
    }
    {
        const VagJob zj[2] = {{nullptr, w.hst, B, H, H, H, 0}, {nullptr, w.hst + (Ts + 1) * BH, B, H, H, H, 0}};
        VAG_TRY(vag_jobs_launch(zj, 2, s));          // initial states of both directions (the one-launch kernels write them themselves)
    }
    GruStepArgs a = {};
    a.lda = H; a.ldw = H; a.ldother = 6 * H; a.ldh = H; a.ld2 = Ts * 2 * H;
    a.M = (int)B; a.K = (int)H; a.H = (int)H; a.lengths = lengths; a.comp_hidden = 1;
    for (int64_t k = 0; k < Ts; ++k) {
        for (int d = 0; d < 2; ++d) {
            const int64_t t = d == 0 ? k : Ts - 1 - k;
            const vag_gru_w& g = d == 0 ? fw : bw;
            float* hs = w.hst + d * (Ts + 1) * BH;
            GruSide& sd = a.s[d];
            sd.A = hs + k * BH; sd.W = s16 ? as_f(w16 + d * 3 * H * H) : g.w_hh; sd.bias = g.b_hh;
            sd.other = w.xp + t * B * 6 * H + d * 3 * H;
            sd.hprev = hs + k * BH; sd.hout = hs + (k + 1) * BH;
            sd.out2 = enc + t * 2 * H + d * H;
            sd.save = w.gates + (d * Ts + k) * 4 * BH;
            sd.t = (int)t;
        }
        VAG_TRY(vag_gru_step_launch(a, 2, s, s16));
    }
    VAG_TRY(vag_dropout_apply_launch(enc, B * Ts * 2 * H, 0, rng, VAG_DROP_ENC_CTX, p_ctx, s));
    return VAG_OK;
}

int vag_bigru_seq_bwd(const int64_t* src, const int32_t* lengths, vag_gru_w fw, vag_gru_w bw, float p_emb, float p_ctx,
                      const uint64_t* rng, int64_t B, int64_t Ts, int64_t E, int64_t H, float* d_enc, float* ws,
                      float* g_emb, vag_gru_g g_fw, vag_gru_g g_bw, vag_stream_t stream) {
    hipStream_t s = S_(stream);
    VAG_CHECK_ARG(src && lengths && d_enc && ws && g_emb && B > 0 && Ts > 0 && E % 4 == 0 && H % 4 == 0);
    BiGruWs w = bigru_ws(ws, B, Ts, E, H);
    const int64_t R = Ts * B, BH = B * H;
    float* d_xp = w.xp;      // forward input projections are no longer needed (gates are saved)
    const float* whhT = w.whhT;
    const bool s16 = g_store16 && g_derived;
    VAG_CHECK_ARG(!g_store16 || (g_derived && H % 8 == 0));
    if (s16) {
        whhT = as_f(derived_layout(const_cast<float*>(g_derived), H).encT16);
    } else if (g_derived) {
        whhT = derived_layout(const_cast<float*>(g_derived), H).encT;
    } else {
        for (int d = 0; d < 2; ++d) {
            const vag_gru_w& g = d == 0 ? fw : bw;
            VAG_TRY(vag_transpose_launch(g.w_hh, 3 * H, H, w.whhT + d * 3 * H * H, s));
        }
    }
    const bool pers_on = vag_opt().persistent && vag_opt().persistent_enc_bwd;
    const bool wide16 = s16 && pers_on && vag_enc_wide16_ok(B, Ts, H) && B >= 64;
    const bool persist = wide16 || (!s16 && pers_on && (H == 512 || H == 256) && vag_enc_persistent_ok(B, Ts, H));
    if (wide16) {
        // 2-byte mode, wide batches: the twin of the forward's one-launch kernel (fp16 W_hh^T slice in registers, gate gradients
        // exchanged as fp16 x 2^12 -- what the chain's fp16-pipe product rounds them to)
        VAG_TRY(vag_enc_bwd_wide16_launch(reinterpret_cast<const vag_half*>(whhT), d_enc, w.gates, w.hst, lengths, rng, p_ctx, d_xp,
                                          w.dgh, reinterpret_cast<vag_half*>(w.gx), w.sync_b, B, Ts, H, s));
    } else if (persist) {
        // the whole backward recurrence, both directions, in ONE launch (persist.hip)
        VAG_TRY(vag_enc_bwd_persistent_launch(whhT, d_enc, w.gates, w.hst, lengths, rng, p_ctx, d_xp, w.dgh, w.sync_b, B, Ts, H, s));
    }
    // last processed step: plain elementwise cell backward (no gradient arrives from a later step)
    GruBwdArgs a = {};
    a.ld_add = Ts * 2 * H; a.ldh = H; a.ldgi = 6 * H; a.ldgh = 3 * H;
    a.M = (int)B; a.H = (int)H; a.lengths = lengths; a.rng = rng; a.sid = VAG_DROP_ENC_CTX; a.p = p_ctx;
    int cur = 0;
    {
        const int64_t k = Ts - 1;
        for (int d = 0; d < 2; ++d) {
            const int64_t t = d == 0 ? k : Ts - 1 - k;
            GruBwdSide& sd = a.s[d];
            sd.dh_carry = nullptr;
            sd.dh_add = d_enc + t * 2 * H + d * H;
            sd.drop_idx0 = t * 2 * H + d * H;
            sd.save = w.gates + (d * Ts + k) * 4 * BH;
            sd.hprev = w.hst + (d * (Ts + 1) + k) * BH;
            sd.dgi = d_xp + t * B * 6 * H + d * 3 * H;
            sd.dgh = w.dgh + (d * Ts + k) * B * 3 * H;
            sd.dh_prev = w.carry + (d * 2 + cur) * BH;
            sd.t = (int)t;
        }
        if (!persist) VAG_TRY(vag_gru_bwd_elem_launch(a, 2, s));
    }
    // every other step: dh = dgh[k] W_hh + z*dh[k]  fused with the cell backward of step k-1 (both directions / launch)
    GruBwdStepArgs f = {};
    f.lda = 3 * H; f.ldw = 3 * H; f.ld_add = Ts * 2 * H; f.ldh = H; f.ldgi = 6 * H; f.ldgh = 3 * H;
    f.M = (int)B; f.K = (int)(3 * H); f.H = (int)H; f.lengths = lengths; f.rng = rng; f.sid = VAG_DROP_ENC_CTX; f.p = p_ctx;
    f.has_cell = 1;
    for (int64_t k = Ts - 1; k >= 1 && !persist; --k) {
        for (int d = 0; d < 2; ++d) {
            const int64_t k1 = k - 1;
            const int64_t t1 = d == 0 ? k1 : Ts - 1 - k1;
            GruBwdStepSide& sd = f.s[d];
            sd.A = w.dgh + (d * Ts + k) * B * 3 * H;
            sd.WT = s16 ? as_f(reinterpret_cast<const vag_half*>(whhT) + d * 3 * H * H) : whhT + d * 3 * H * H;
            sd.addend = w.carry + (d * 2 + cur) * BH;
            sd.dh_add = d_enc + t1 * 2 * H + d * H;
            sd.drop_idx0 = t1 * 2 * H + d * H;
            sd.save = w.gates + (d * Ts + k1) * 4 * BH;
            sd.hprev = w.hst + (d * (Ts + 1) + k1) * BH;
            sd.dgi = d_xp + t1 * B * 6 * H + d * 3 * H;
            sd.dgh = w.dgh + (d * Ts + k1) * B * 3 * H;
            sd.dh_direct = w.carry + (d * 2 + (cur ^ 1)) * BH;
            sd.dh_out = nullptr;
            sd.t = (int)t1;
        }
        VAG_TRY(vag_gru_bwd_step_launch(f, 2, s, s16));
        cur ^= 1;
    }
    VagGemmGroup grp1;      // the four weight gradients (both directions) go out as one grouped launch
    for (int d = 0; d < 2; ++d) {
        const vag_gru_g& gg = d == 0 ? g_fw : g_bw;
        const float* dgh = w.dgh + d * Ts * B * 3 * H;
        const float* hs = w.hst + d * (Ts + 1) * BH;
        VAG_TRY(gemm_tn_acc(3 * H, H, R, dgh, 3 * H, hs, H, gg.w_hh, H, s, gg.b_hh));                    // + bias gradient
        VAG_TRY(gemm_tn_acc(3 * H, E, R, d_xp + d * 3 * H, 6 * H, w.x, E, gg.w_ih, E, s, gg.b_ih));
    }
    VAG_TRY(grp1.end(s));
    // d(embedded inputs) = sum over the directions of dgi W_ih: both products add into a zeroed buffer, one grouped launch
    if (!g_step_zeroed) VAG_TRY(zero_async(w.dx, R * E * sizeof(float), s));
    VagGemmGroup grp2;
    for (int d = 0; d < 2; ++d)
        VAG_TRY(gemm_nn(R, E, 3 * H, d_xp + d * 3 * H, 6 * H, (d == 0 ? fw : bw).w_ih, E, 1.f, w.dx, E, s));
    VAG_TRY(grp2.end(s));
    VAG_TRY(vag_embed_scatter_launch(src, 1, Ts, Ts, B, w.dx, E, g_emb, rng, VAG_DROP_ENC_EMB, p_emb, s,
                                     g_step_poison_inject ? vag_persist_guard() : nullptr));
    return VAG_OK;
}

// =====================================================================================================
// single GRU cell step (torch nn.GRU on a length-1 sequence: layers/NMT_Decoder.py:121,129)
// =====================================================================================================
int vag_gru_cell_fwd(const float* gi, const float* h_prev, const float* w_hh, const float* b_hh, int64_t M, int64_t H,
                     float* h_out, float* save, vag_stream_t stream) {
    VAG_CHECK_ARG(gi && h_prev && w_hh && b_hh && h_out && M > 0 && H > 0 && H % 4 == 0);
    GruStepArgs a = {};
    a.lda = H; a.ldw = H; a.ldother = 3 * H; a.ldh = H; a.ld2 = 0;
    a.M = (int)M; a.K = (int)H; a.H = (int)H; a.lengths = nullptr; a.comp_hidden = 1;
    a.s[0].A = h_prev; a.s[0].W = w_hh; a.s[0].bias = b_hh; a.s[0].other = gi; a.s[0].hprev = h_prev;
    a.s[0].hout = h_out; a.s[0].out2 = nullptr; a.s[0].save = save; a.s[0].t = 0;
    return vag_gru_step_launch(a, 1, S_(stream));
}

// One backward step of a GRU recurrence (what torch autograd replays per time step for nn.GRU): the hidden-state
// gradient arriving through the later step's recurrent projection, plus the carried and the direct contributions, then
// the cell backward.  save = [4][M][H] (r, z, n, W_hn h + b_hn) as written by vag_gru_cell_fwd.
int vag_gru_cell_bwd(const float* dgh_next, const float* w_hh_t, const float* carry, const float* d_out, const float* save,
                     const float* h_prev, int64_t M, int64_t H, float* dgi, float* dgh, float* carry_out,
                     vag_stream_t stream) {
    VAG_CHECK_ARG(dgh_next && w_hh_t && save && h_prev && dgi && dgh && carry_out && M > 0 && H > 0 && H % 4 == 0);
    GruBwdStepArgs f = {};
    f.lda = 3 * H; f.ldw = 3 * H; f.ld_add = H; f.ldh = H; f.ldgi = 3 * H; f.ldgh = 3 * H;
    f.M = (int)M; f.K = (int)(3 * H); f.H = (int)H; f.lengths = nullptr; f.rng = nullptr; f.sid = 0; f.p = 0.f;
    f.has_cell = 1;
    GruBwdStepSide& sd = f.s[0];
    sd.A = dgh_next; sd.WT = w_hh_t; sd.addend = carry; sd.dh_add = d_out; sd.drop_idx0 = 0; sd.save = save;
    sd.hprev = h_prev; sd.dgi = dgi; sd.dgh = dgh; sd.dh_direct = carry_out; sd.dh_out = nullptr; sd.t = 0;
    return vag_gru_bwd_step_launch(f, 1, S_(stream));
}

// =====================================================================================================
// attention keys
// =====================================================================================================
int vag_attn_keys_proj(const float* enc, const float* attn_e, int64_t rows, int64_t C, float* pe, vag_stream_t stream) {
    VAG_CHECK_ARG(enc && attn_e && pe && rows > 0 && C > 0);
    if (g_store16)      // 2-byte storage mode: the keys are written as fp16
        return vag_gemm_launch(rows, C, C, 1.f, enc, C, 1, attn_e, 1, C, 0.f, pe, C, nullptr, 0, S_(stream), 1);
    return linear_fwd(rows, C, C, enc, C, attn_e, nullptr, 0, pe, C, S_(stream));
}
// a4 stand-alone (inference): alpha = softmax_s(v . tanh(pe_s + q)), ctx = sum_s alpha_s enc_s
int vag_bahdanau_attn_fwd(const float* pe, const float* q, const float* v, const float* mask, const float* enc, int64_t N,
                          int64_t rows_per_src, int64_t Ts, int64_t C, float* scores, float* alpha, float* ctx,
                          vag_stream_t stream) {
    hipStream_t s = S_(stream);
    VAG_CHECK_ARG(pe && q && v && enc && scores && alpha && ctx);
    VAG_TRY(vag_attn_scores_launch(0, pe, q, C, v, mask, N, rows_per_src, Ts, C, scores, s));
    return vag_attn_ctx_launch(1, scores, enc, N, rows_per_src, Ts, C, alpha, ctx, s);
}
int vag_attn_keys_proj_bwd(const float* enc, const float* attn_e, const float* d_pe, int64_t rows, int64_t C, float* d_enc,
                           int accumulate_enc, float* g_attn_e, vag_stream_t stream) {
    hipStream_t s = S_(stream);
    VAG_CHECK_ARG(enc && attn_e && d_pe && rows > 0 && C > 0);
    if (d_enc) VAG_TRY(gemm_nn(rows, C, C, d_pe, C, attn_e, C, accumulate_enc ? 1.f : 0.f, d_enc, C, s));
    if (g_attn_e) VAG_TRY(gemm_tn_acc(C, C, rows, d_pe, C, enc, C, g_attn_e, C, s));
    return VAG_OK;
}

// =====================================================================================================
// cGRU decoder
// =====================================================================================================
// Per-call derived weights (vag_cgru_prepare): Wcat = [attn_h ; gru_2.w_hh] (C+3H,H) so that q = attn_h(h1) and the
// hidden projection of gru_2 come out of ONE product; bcat = [0 ; gru_2.b_hh]; Wp = gru_2.w_ih . context2hid (3H,C)
// so that gru_2's input projection is taken straight from the context (context2hid folded in; exact in real
// arithmetic, fp32 reassociation only).
struct CgruPrep {
    float *wcat, *bcat, *wp;
    int64_t total;
};
static CgruPrep cgru_prep(float* p, int64_t H) {
    CgruPrep w;
    const int64_t C = 2 * H, Q = C + 3 * H;
    int64_t o = 0;
    auto take = [&](int64_t n) { float* q = p ? p + o : nullptr; o += (n + 63) & ~63ll; return q; };
    w.wcat = take(Q * H); w.bcat = take(Q); w.wp = take(3 * H * C);
    w.total = o;
    return w;
}
int64_t vag_cgru_prep_floats(int64_t H) { return cgru_prep(nullptr, H).total; }
static int64_t cgru_prep_total(int64_t H) { return cgru_prep(nullptr, H).total; }
int64_t vag_derived_floats(int64_t H) { return derived_layout(nullptr, H).total; }

static bool dec_w_ok(const vag_dec_w& w) {
    return w.emb && w.gru1.w_ih && w.gru1.w_hh && w.gru1.b_ih && w.gru1.b_hh && w.attn_h && w.attn_v && w.c2h &&
           w.gru2.w_ih && w.gru2.w_hh && w.gru2.b_ih && w.gru2.b_hh;
}

int vag_cgru_prepare(vag_dec_w w, int64_t H, float* prep, vag_stream_t stream) {
    hipStream_t s = S_(stream);
    VAG_CHECK_ARG(dec_w_ok(w) && prep && H > 0 && H % 4 == 0 && aligned16(prep));
    const int64_t C = 2 * H;
    CgruPrep p = cgru_prep(prep, H);
    VAG_TRY(copy_async(p.wcat, w.attn_h, C * H * sizeof(float), s));
    VAG_TRY(copy_async(p.wcat + C * H, w.gru2.w_hh, 3 * H * H * sizeof(float), s));
    VAG_TRY(zero_async(p.bcat, C * sizeof(float), s));
    VAG_TRY(copy_async(p.bcat + C, w.gru2.b_hh, 3 * H * sizeof(float), s));
    return gemm_nn(3 * H, C, H, w.gru2.w_ih, H, w.c2h, C, 0.f, p.wp, C, s);
}

// Everything the recurrences read that is a function of the parameters alone, refreshed once per optimiser step by a
// step driver (the stand-alone operators rebuild their share per call): [attn_h; W_hh2] stacked and transposed, the
// folded W_ih2 W_c2h, W_hh1^T, both encoder W_hh^T.  derived: vag_derived_floats(H) floats.
int vag_derive_weights(vag_dec_w w, const float* enc_whh_fw, const float* enc_whh_bw, int64_t H, int with_fp16,
                       float* derived, vag_stream_t stream) {
    hipStream_t s = S_(stream);
    VAG_CHECK_ARG(dec_w_ok(w) && enc_whh_fw && enc_whh_bw && derived && H > 0 && H % 4 == 0 && aligned16(derived));
    const int64_t C = 2 * H, Q = C + 3 * H;
    DerivedW d = derived_layout(derived, H);
    CgruPrep p = cgru_prep(d.prep, H);
    const VagJob jobs[9] = {
        {w.attn_h, p.wcat, C, H, H, H, 1},                                 // wcat = [attn_h ; W_hh2]
        {w.gru2.w_hh, p.wcat + C * H, 3 * H, H, H, H, 1},
        {nullptr, p.bcat, 1, C, C, C, 0},                                  // bcat = [0 ; b_hh2]
        {w.gru2.b_hh, p.bcat + C, 1, 3 * H, 3 * H, 3 * H, 1},
        {w.attn_h, d.wcatT, C, H, H, Q, 2},                                // wcatT (H, C+3H) = [attn_h^T | W_hh2^T]
        {w.gru2.w_hh, d.wcatT + C, 3 * H, H, H, Q, 2},
        {w.gru1.w_hh, d.whh1T, 3 * H, H, H, 3 * H, 2},
        {enc_whh_fw, d.encT, 3 * H, H, H, 3 * H, 2},
        {enc_whh_bw, d.encT + 3 * H * H, 3 * H, H, H, 3 * H, 2},
    };
    VAG_TRY(vag_jobs_launch(jobs, 9, s));
    if (with_fp16) {        // fp16 copies of what the recurrences re-read every time step (2-byte storage mode)
        VAG_CHECK_ARG(H % 8 == 0);
        auto h = [&](vag_half* q) { return reinterpret_cast<float*>(q); };
            const VagJob j16[10] = {
            {w.attn_h, h(d.wcat16), C, H, H, H, 3},
            {w.gru2.w_hh, h(d.wcat16 + C * H), 3 * H, H, H, H, 3},
            {w.attn_h, h(d.wcatT16), C, H, H, Q, 4},
            {w.gru2.w_hh, h(d.wcatT16 + C), 3 * H, H, H, Q, 4},
            {w.gru1.w_hh, h(d.whh1_16), 3 * H, H, H, H, 3},
            {w.gru1.w_hh, h(d.whh1T16), 3 * H, H, H, 3 * H, 4},
            {enc_whh_fw, h(d.enc16), 3 * H, H, H, H, 3},
            {enc_whh_bw, h(d.enc16 + 3 * H * H), 3 * H, H, H, H, 3},
            {enc_whh_fw, h(d.encT16), 3 * H, H, H, 3 * H, 4},
            {enc_whh_bw, h(d.encT16 + 3 * H * H), 3 * H, H, H, 3 * H, 4},
        };
        VAG_TRY(vag_jobs_launch(j16, 10, s));
    }
    // (rounds 2-4 also formed the folded product Wp = W_ih2 W_c2h here, 29 us per optimiser step: the training step takes
    // context2hid and W_ih2 one after the other now (cgru: uk), and the one path that still wants Wp -- the free-running launch
    // chain -- forms it for itself)
    (void)p;
    return VAG_OK;
}

struct CgruWs {
    float *xp1, *h1, *g1, *g2, *qhp, *scores, *alpha, *tmp, *prep, *encwp;
    float* uk;                  // (B,Ts,H) enc W_c2h^T: context2hid applied to the keys once per batch (round 5: the projected keys are
                                // encwp = uk W_ih2^T -- two products through the H-wide intermediate instead of one through W_ih2 W_c2h)
    float* psc;                 // persistent decoder (persist.hip): the steps' scores as exchanged between workgroups (Tt,4,B,Ts)
    unsigned* sync;             // ... and its counters
    unsigned* sync_b;           // the backward kernel's counters ...
    float* dal;                 // ... and its d alpha accumulator (Tt,4,B,Ts)   [h1 .. dal: one range to zero]
    int64_t total;
};
static CgruWs cgru_ws(float* ws, int64_t B, int64_t Ts, int64_t Tt, int64_t E, int64_t H) {
    CgruWs w;
    const int64_t C = 2 * H;
    int64_t o = 0;
    auto take = [&](int64_t n) { float* p = ws ? ws + o : nullptr; o += (n + 63) & ~63ll; return p; };
    w.xp1 = take(Tt * B * 3 * H);
    w.g1 = take(Tt * 4 * B * H);
    w.g2 = take(Tt * 4 * B * H);
    w.qhp = take(Tt * B * (C + 3 * H));     // [q | W_hh2 h1 + b_hh2] per step
    w.scores = take(B * Ts);
    w.alpha = take(Tt * B * Ts);
    w.tmp = take(B * E);
    w.prep = take(cgru_prep(nullptr, H).total);
    w.encwp = take(B * Ts * 3 * H);         // (W_ih2 W_c2h) enc[b,s,:]: the keys as gru_2 sees them, once per batch
    w.uk = take(B * Ts * H);
    w.h1 = take(Tt * B * H);                // (exchanged between workgroups with marked words: starts from zero, see vag_step_zero_ranges)
    w.psc = take(Tt * B * Ts * 4);          // four copies (persist.hip: ACC_SHARDS)
    w.sync = reinterpret_cast<unsigned*>(take(vag_dec_persistent_sync_words(B, Tt)));
    w.sync_b = reinterpret_cast<unsigned*>(take(vag_dec_persistent_sync_words(B, Tt)));
    w.dal = take(Tt * B * Ts * 4);
    w.total = o;
    return w;
}
}  // extern "C"
// What the recurrence kernels of one training step expect to find zeroed, as two word ranges (vag_train_step's prologue launch
// zeroes them and tells the launch functions so: persist.hip, g_prezeroed)
void vag_step_zero_ranges(float* ws_enc, float* ws_dec, int64_t B, int64_t Ts, int64_t Tt, int64_t Es, int64_t Et, int64_t H,
                          unsigned** p, int64_t* n) {
    BiGruWs e = bigru_ws(ws_enc, B, Ts, Es, H);
    p[0] = e.sync;
    n[0] = (reinterpret_cast<unsigned*>(ws_enc) + e.total) - e.sync;
    CgruWs d = cgru_ws(ws_dec, B, Ts, Tt, Et, H);
    p[2] = reinterpret_cast<unsigned*>(e.dx);          // the encoder's d(embedded inputs): two grouped products add into it
    n[2] = Ts * B * Es;
    p[1] = reinterpret_cast<unsigned*>(d.h1);          // h1 | psc | sync | sync_b | dal: one range
    n[1] = (reinterpret_cast<unsigned*>(ws_dec) + d.total) - p[1];
}
extern "C" {
int64_t vag_cgru_ws_floats(int64_t B, int64_t Ts, int64_t Tt, int64_t E, int64_t H) {
    return cgru_ws(nullptr, B, Ts, Tt, E, H).total;
}
// float offset of a saved per-step tensor inside the workspace (parity tests read them): 0 = alpha (Tt,B,Ts),
// 1 = h1 (Tt,B,H), 2 = [q | W_hh2 h1 + b] (Tt,B,C+3H)
int64_t vag_cgru_ws_offset(int64_t B, int64_t Ts, int64_t Tt, int64_t E, int64_t H, int which) {
    static float base[1];
    CgruWs k = cgru_ws(base, B, Ts, Tt, E, H);
    const float* p = which == 0 ? k.alpha : which == 1 ? k.h1 : which == 2 ? k.qhp : nullptr;
    return p ? (int64_t)(p - base) : -1;
}

struct StepBufs {   // per-step buffers of one decoder step (training: slices of the sequence arrays)
    const float* xp1;   // (N,3H) input projection of gru_1 (bias included)
    const float* hprev; // (N,H)
    float *h1, *g1, *g2, *qhp, *scores, *alpha, *c, *h2;
};

// One cGRU step for N rows (layers/NMT_Decoder.py:121-129): 5 launches.
static int cgru_step(const float* enc, const float* pe, const float* mask, int64_t rps, const vag_dec_w& w,
                     const CgruPrep& p, int64_t N, int64_t Ts, int64_t H, const StepBufs& b, hipStream_t s) {
    const int64_t C = 2 * H, Q = C + 3 * H;
    GruStepArgs a = {};
    a.lda = H; a.ldw = H; a.ldother = 3 * H; a.ldh = H; a.ld2 = 0;
    a.M = (int)N; a.K = (int)H; a.H = (int)H; a.lengths = nullptr; a.comp_hidden = 1;
    a.s[0].A = b.hprev; a.s[0].W = w.gru1.w_hh; a.s[0].bias = w.gru1.b_hh; a.s[0].other = b.xp1;
    a.s[0].hprev = b.hprev; a.s[0].hout = b.h1; a.s[0].out2 = nullptr; a.s[0].save = b.g1; a.s[0].t = 0;
    VAG_TRY(vag_gru_step_launch(a, 1, s));                                                         // gru_1        :121
    VAG_TRY(vag_skinny_launch(N, Q, H, b.h1, H, p.wcat, H, p.bcat, nullptr, 0, b.qhp, Q, 0, s));    // attn_h(h1) :47 | W_hh2 h1
    VAG_TRY(vag_attn_scores_launch(0, pe, b.qhp, Q, w.attn_v, mask, N, rps, Ts, C, b.scores, s));   // :47-51, :41-43
    VAG_TRY(vag_attn_ctx_launch(1, b.scores, enc, N, rps, Ts, C, b.alpha, b.c, s));                 // :44, :126
    a.lda = C; a.ldw = C; a.ldother = Q; a.K = (int)C; a.comp_hidden = 0;
    a.s[0].A = b.c; a.s[0].W = p.wp; a.s[0].bias = w.gru2.b_ih; a.s[0].other = b.qhp + C;
    a.s[0].hprev = b.h1; a.s[0].hout = b.h2; a.s[0].save = b.g2;
    VAG_TRY(vag_gru_step_launch(a, 1, s));                                                         // context2hid + gru_2 :127-129
    return VAG_OK;
}

// tanh(W1 h2 + W2 c + W3 e + biases) -> dropout -> logits   for N rows of one step (NMT_Decoder.py:137-143)
// tmid for a step whose context share cw (N,E) = alpha . (enc W2^T) is at hand instead of the context (hoisted decoding step)
// embw3 / tok: W3 e as a table line (vag_cgru_decode_tables) instead of a product over the embedded token e
static int head_pre_step_h(const float* h2, const float* cw, const float* e, const float* embw3, const int64_t* tok,
                           const vag_head_w& w, int64_t N, int64_t E, int64_t H, float* tmid, hipStream_t s) {
    const float* A3[3] = {h2, nullptr, embw3 ? nullptr : e};
    const float* W3[3] = {w.w1, nullptr, embw3 ? nullptr : w.w3};
    const float* B3[3] = {w.b1, w.b2, w.b3};
    const int64_t ld3[3] = {H, 0, E}, K3[3] = {H, 0, embw3 ? 0 : E};
    return vag_skinny3_launch(N, E, A3, ld3, W3, ld3, K3, B3, tmid, E, VAG_ACT_TANH, nullptr, VAG_DROP_DEC_OUT, 0.f, 0, s, cw, E,
                              embw3, tok, E);
}
static int head_step(const float* h2, const float* c, const float* e, const vag_head_w& w, int64_t N, int64_t E, int64_t H,
                     int64_t V, float p_out, const uint64_t* rng, int64_t drop_idx0, float* tmp, float* tmid,
                     float* logits, int64_t ldl, hipStream_t s) {
    const int64_t C = 2 * H;
    if (N <= 256 && aligned16(h2) && aligned16(c) && aligned16(e) && aligned16(w.w1) && aligned16(w.w2) && aligned16(w.w3)) {
        // one launch: the three products, the biases, tanh and the dropout multiplier (round 2; was four launches)
        const float* A3[3] = {h2, c, e};
        const float* W3[3] = {w.w1, w.w2, w.w3};
        const float* B3[3] = {w.b1, w.b2, w.b3};
        const int64_t ld3[3] = {H, C, E}, K3[3] = {H, C, E};
        VAG_TRY(vag_skinny3_launch(N, E, A3, ld3, W3, ld3, K3, B3, tmid, E, VAG_ACT_TANH, rng, VAG_DROP_DEC_OUT, p_out, drop_idx0, s));
    } else {
        VAG_TRY(vag_skinny_launch(N, E, H, h2, H, w.w1, H, w.b1, nullptr, 0, tmp, E, 0, s));
        VAG_TRY(vag_skinny_launch(N, E, C, c, C, w.w2, C, w.b2, tmp, E, tmp, E, 0, s));
        VAG_TRY(vag_skinny_launch(N, E, E, e, E, w.w3, E, w.b3, tmp, E, tmid, E, VAG_ACT_TANH, s));
        VAG_TRY(vag_dropout_apply_launch(tmid, N * E, drop_idx0, rng, VAG_DROP_DEC_OUT, p_out, s));
    }
    VAG_TRY(linear_fwd(N, V, E, tmid, E, w.out_w, w.out_b, 0, logits, ldl, s));
    return VAG_OK;
}

int vag_cgru_attn_decode_seq_fwd(const float* enc, const float* pe, const float* mask, const float* h0, int64_t* tok,
                                 vag_dec_w w, int64_t B, int64_t Ts, int64_t Tt, int64_t E, int64_t H, int64_t V,
                                 float* h2_all, float* c_all, float* e_all, float* ws, int free_run,
                                 const vag_head_w* head, float p_out, const uint64_t* rng, float* tmid, float* logits,
                                 int64_t ldl, vag_stream_t stream) {
    hipStream_t s = S_(stream);
    VAG_CHECK_ARG(enc && pe && mask && h0 && tok && h2_all && c_all && e_all && ws && dec_w_ok(w));
    VAG_CHECK_ARG(B > 0 && Ts > 0 && Tt > 0 && E % 4 == 0 && H % 4 == 0 && E > 0 && H > 0 && aligned16(ws));
    VAG_CHECK_ARG(!free_run || (head && tmid && logits && ldl >= V && ldl % 4 == 0));
    const int64_t C = 2 * H, Q = C + 3 * H, BH = B * H;
    CgruWs k = cgru_ws(ws, B, Ts, Tt, E, H);
    CgruPrep p = cgru_prep(k.prep, H);
    if (g_derived) p = cgru_prep(derived_layout(const_cast<float*>(g_derived), H).prep, H);
    else VAG_TRY(vag_cgru_prepare(w, H, k.prep, stream));
    // Teacher forcing runs the step with the context projection hoisted (4 launches, see attn_ctx_gru_kernel): the cell
    // only needs sum_s alpha_s (W_ih2 W_c2h enc_s), the contexts themselves are formed for all steps after the loop.
    // Free running needs each context at once for the head, so it keeps the 5-launch step; backward is common to both
    // and always works on the projected keys.
    const bool hoist = !free_run;
    // 2-byte storage mode: teacher-forced (hoisted) path only, weights from the driver's derived buffer
    const bool s16 = g_store16;
    VAG_CHECK_ARG(!s16 || (hoist && g_derived && H % 8 == 0 && Ts * B < (1ll << 28)));
    DerivedW dw16 = {};
    if (s16) dw16 = derived_layout(const_cast<float*>(g_derived), H);
    {
        VagGemmGroup grp;
        if (!free_run) {
            // every input token is known -> embed and project all steps at once
            if (!g_step_gathered) VAG_TRY(vag_embed_gather_launch(tok, B, 1, Tt, B, w.emb, E, e_all, nullptr, 0, 0.f, s));
            VAG_TRY(vag_gemm_launch(Tt * B, 3 * H, E, 1.f, e_all, E, 1, w.gru1.w_ih, 1, E, 0.f, k.xp1, 3 * H, w.gru1.b_ih, 0, s));
        }
        // the keys as gru_2 sees them (NMT_Decoder.py:127-129 hoisted): context2hid on the keys (joins the group), then W_ih2 on
        // the H-wide result -- 6.7 instead of 9.6 GFLOP at configs[1] against the one product through W_ih2 W_c2h, and the
        // per-optimiser-step product that forms W_ih2 W_c2h is gone (exact in real arithmetic; the reference's own order)
        VAG_TRY(vag_gemm_launch(B * Ts, H, C, 1.f, enc, C, 1, w.c2h, 1, C, 0.f, k.uk, H, nullptr, 0, s));
        VAG_TRY(grp.end(s));
    }
    {
        VagGemmGroup grp2;      // (launched at once: the recurrence below reads it)
        VAG_TRY(vag_gemm_launch(B * Ts, 3 * H, H, 1.f, k.uk, H, 1, w.gru2.w_ih, 1, H, 0.f, k.encwp, 3 * H, nullptr, 0, s, s16 ? 1 : 0));
        VAG_TRY(grp2.end(s));
    }
    auto hoisted_step = [&](int64_t t) -> int {
        const float* hprev = t == 0 ? h0 : h2_all + (t - 1) * BH;
        float* h1 = k.h1 + t * BH;
        float* qhp = k.qhp + t * B * Q;
        GruStepArgs a = {};
        a.lda = H; a.ldw = H; a.ldother = 3 * H; a.ldh = H; a.ld2 = 0;
        a.M = (int)B; a.K = (int)H; a.H = (int)H; a.lengths = nullptr; a.comp_hidden = 1;
        a.s[0].A = hprev; a.s[0].W = s16 ? as_f(dw16.whh1_16) : w.gru1.w_hh; a.s[0].bias = w.gru1.b_hh;
        a.s[0].other = k.xp1 + t * B * 3 * H;
        a.s[0].hprev = hprev; a.s[0].hout = h1; a.s[0].out2 = nullptr; a.s[0].save = k.g1 + t * 4 * BH; a.s[0].t = 0;
        VAG_TRY(vag_gru_step_launch(a, 1, s, s16));                                                         // gru_1 :121
        if (s16) {
            VAG_TRY(vag_skinny_launch(B, C, H, h1, H, as_f(dw16.wcat16), H, nullptr, nullptr, 0, qhp, Q, 0, s, true));  // :47
            VAG_TRY(vag_attn_dot_side_launch(0, pe, qhp, Q, w.attn_v, mask, nullptr, B, Ts, C, k.scores, B, 3 * H, H, h1, H,
                                             as_f(dw16.wcat16 + C * H), H, p.bcat + C, nullptr, qhp + C, Q, s, true));
            VAG_TRY(vag_attn_ctx_gru_launch(k.scores, k.encwp, B, 1, Ts, H, w.gru2.b_ih, qhp + C, Q, h1, k.alpha + t * B * Ts,
                                            h2_all + t * BH, k.g2 + t * 4 * BH, s, true));
            return VAG_OK;
        }
        if (Ts * B < (1ll << 28)) {
            // q = attn_h h1, then the scores with W_hh2 h1 + b_hh2 (not needed before the cell) in the same grid
            VAG_TRY(vag_skinny_launch(B, C, H, h1, H, p.wcat, H, nullptr, nullptr, 0, qhp, Q, 0, s));                  // :47
            VAG_TRY(vag_attn_dot_side_launch(0, pe, qhp, Q, w.attn_v, mask, nullptr, B, Ts, C, k.scores, B, 3 * H, H, h1, H,
                                             p.wcat + C * H, H, p.bcat + C, nullptr, qhp + C, Q, s));                   // :47-51, :41-43
        } else {
            VAG_TRY(vag_skinny_launch(B, Q, H, h1, H, p.wcat, H, p.bcat, nullptr, 0, qhp, Q, 0, s));         // attn_h(h1) | W_hh2 h1
            VAG_TRY(vag_attn_scores_launch(0, pe, qhp, Q, w.attn_v, mask, B, 1, Ts, C, k.scores, s));       // :47-51, :41-43
        }
        VAG_TRY(vag_attn_ctx_gru_launch(k.scores, k.encwp, B, 1, Ts, H, w.gru2.b_ih, qhp + C, Q, h1, k.alpha + t * B * Ts,
                                        h2_all + t * BH, k.g2 + t * 4 * BH, s));                            // :44, :126-129
        return VAG_OK;
    };
    if (hoist && !s16 && vag_opt().persistent && vag_dec_persistent_ok(B, Ts, Tt, H)) {
        // all Tt steps in ONE launch (persist.hip): weights in registers, keys in LDS, four exchanges per step
        VAG_TRY(vag_dec_fwd_persistent_launch(pe, mask, h0, k.xp1, w.gru1.w_hh, w.gru1.b_hh, p.wcat, p.bcat, w.attn_v, k.encwp,
                                              w.gru2.b_ih, k.h1, k.g1, k.qhp, k.alpha, h2_all, k.g2, k.psc, k.sync, B, Ts, Tt, H,
                                              s));
        return vag_attn_wsum_launch(1, k.alpha, enc, B, Ts, Tt, C, c_all, s);                              // all contexts :126
    }
    for (int64_t t = 0; hoist && t < Tt; ++t) VAG_TRY(hoisted_step(t));
    if (hoist) return vag_attn_wsum_launch(1, k.alpha, enc, B, Ts, Tt, C, c_all, s);                       // all contexts :126
    if (g_derived) {
        // the free-running launch chain takes gru_2's input projection from the context through the folded product
        // Wp = W_ih2 W_c2h (cgru_step); a driver's derived buffer does not carry it any more: formed here, in this call's workspace
        CgruPrep pk = cgru_prep(k.prep, H);
        VagGemmGroup now;
        VAG_TRY(gemm_nn(3 * H, C, H, w.gru2.w_ih, H, w.c2h, C, 0.f, pk.wp, C, s));
        VAG_TRY(now.end(s));
        p.wp = pk.wp;
    }
    for (int64_t t = 0; t < Tt; ++t) {
        if (free_run) {
            if (B <= 256 && aligned16(w.emb) && aligned16(w.gru1.w_ih) && aligned16(e_all)) {
                // embedding lookup + input projection of gru_1 in one launch
                VAG_TRY(vag_skinny_gather_launch(B, 3 * H, E, w.emb, E, tok + t * B, w.gru1.w_ih, E, w.gru1.b_ih,
                                                 k.xp1 + t * B * 3 * H, 3 * H, e_all + t * B * E, E, s));
            } else {
                VAG_TRY(vag_embed_gather_launch(tok + t * B, B, 1, 1, B, w.emb, E, e_all + t * B * E, nullptr, 0, 0.f, s));
                VAG_TRY(vag_skinny_launch(B, 3 * H, E, e_all + t * B * E, E, w.gru1.w_ih, E, w.gru1.b_ih, nullptr, 0,
                                          k.xp1 + t * B * 3 * H, 3 * H, 0, s));
            }
        }
        StepBufs b;
        b.xp1 = k.xp1 + t * B * 3 * H;
        b.hprev = t == 0 ? h0 : h2_all + (t - 1) * BH;
        b.h1 = k.h1 + t * BH; b.g1 = k.g1 + t * 4 * BH; b.g2 = k.g2 + t * 4 * BH;
        b.qhp = k.qhp + t * B * Q; b.scores = k.scores; b.alpha = k.alpha + t * B * Ts;
        b.c = c_all + t * B * C; b.h2 = h2_all + t * BH;
        VAG_TRY(cgru_step(enc, pe, mask, 1, w, p, B, Ts, H, b, s));
        if (free_run) {
            VAG_TRY(head_step(b.h2, b.c, e_all + t * B * E, *head, B, E, H, V, p_out, rng, t * B * E, k.tmp,
                              tmid + t * B * E, logits + t * B * ldl, ldl, s));
            // next input = argmax of this step's distribution (V11.py:157), written to tok row t+1
            VAG_TRY(vag_lse_nll_launch(logits + t * B * ldl, ldl, B, V, nullptr, 0, 0, nullptr, nullptr, nullptr,
                                       tok + (t + 1) * B, 1, nullptr, 0, s));
        }
    }
    return VAG_OK;
}

int vag_cgru_free_supported(int64_t B, int64_t Ts, int64_t Tt, int64_t E, int64_t H, int64_t V) {
    return vag_opt().persistent && vag_opt().free_persistent && !g_store16 && vag_dec_free_persistent_ok(B, Ts, Tt, E, H, V) ? 1 : 0;
}
int64_t vag_cgru_free_tables_floats(int64_t B, int64_t Ts, int64_t Tt, int64_t E, int64_t H, int64_t V) {
    return vag_dec_free_tables_floats(B, Ts, Tt, E, H, V);
}
int vag_cgru_attn_decode_free_fwd(const float* enc, const float* pe, const float* mask, const float* h0, int64_t* tok,
                                  vag_dec_w w, int64_t B, int64_t Ts, int64_t Tt, int64_t E, int64_t H, int64_t V,
                                  float* h2_all, float* c_all, float* e_all, float* ws, const vag_head_w* head, float p_out,
                                  const uint64_t* rng, float* tmid, float* logits, int64_t ldl, float* tables,
                                  vag_stream_t stream) {
    hipStream_t s = S_(stream);
    VAG_CHECK_ARG(enc && pe && mask && h0 && tok && h2_all && ws && head && tmid && tables && dec_w_ok(w));
    VAG_CHECK_ARG(head->w1 && head->b1 && head->w2 && head->b2 && head->w3 && head->b3 && head->out_w && head->out_b);
    VAG_CHECK_ARG(vag_cgru_free_supported(B, Ts, Tt, E, H, V) && aligned16(ws) && aligned16(tables));
    VAG_CHECK_ARG(!logits || (ldl >= V && ldl % 4 == 0));
    const int64_t C = 2 * H;
    CgruWs k = cgru_ws(ws, B, Ts, Tt, E, H);
    CgruPrep p = cgru_prep(k.prep, H);
    if (g_derived) p = cgru_prep(derived_layout(const_cast<float*>(g_derived), H).prep, H);
    else VAG_TRY(vag_cgru_prepare(w, H, k.prep, stream));
    auto r64 = [](int64_t n) { return (n + 63) & ~63ll; };
    float* embp = tables;
    float* embw3 = embp + r64(V * 3 * H);
    float* encw2 = embw3 + r64(V * E);
    {
        // what a step needs of a token or of a key, for every token and every key, as four products in one grouped launch:
        // the input projection of gru_1 (:118-121), W3 e (:137), the keys as gru_2 sees them (:127-129), W2 enc (:137)
        VagGemmGroup grp;
        VAG_TRY(vag_gemm_launch(V, 3 * H, E, 1.f, w.emb, E, 1, w.gru1.w_ih, 1, E, 0.f, embp, 3 * H, w.gru1.b_ih, 0, s));
        VAG_TRY(vag_gemm_launch(V, E, E, 1.f, w.emb, E, 1, head->w3, 1, E, 0.f, embw3, E, nullptr, 0, s));
        VAG_TRY(vag_gemm_launch(B * Ts, H, C, 1.f, enc, C, 1, w.c2h, 1, C, 0.f, k.uk, H, nullptr, 0, s));     // (see vag_cgru_attn_decode_seq_fwd)
        VAG_TRY(vag_gemm_launch(B * Ts, E, C, 1.f, enc, C, 1, head->w2, 1, C, 0.f, encw2, E, nullptr, 0, s));
        VAG_TRY(grp.end(s));
    }
    {
        VagGemmGroup grp2;
        VAG_TRY(vag_gemm_launch(B * Ts, 3 * H, H, 1.f, k.uk, H, 1, w.gru2.w_ih, 1, H, 0.f, k.encwp, 3 * H, nullptr, 0, s));
        VAG_TRY(grp2.end(s));
    }
    VAG_TRY(vag_dec_free_persistent_launch(pe, mask, h0, w.gru1.w_hh, w.gru1.b_hh, p.wcat, p.bcat, w.attn_v, k.encwp, w.gru2.b_ih,
                                           k.h1, k.g1, k.qhp, k.alpha, h2_all, k.g2, k.psc, k.sync, tables, head->w1, head->b1,
                                           head->b2, head->b3, head->out_w, head->out_b, tmid, logits, ldl, tok, rng, p_out, B, Ts,
                                           Tt, E, H, V, s));
    if (c_all) VAG_TRY(vag_attn_wsum_launch(1, k.alpha, enc, B, Ts, Tt, C, c_all, s));                       // all contexts :126
    if (e_all) VAG_TRY(vag_embed_gather_launch(tok, B, 1, Tt, B, w.emb, E, e_all, nullptr, 0, 0.f, s));     // the inputs it chose
    return VAG_OK;
}

// set by vag_cgru_attn_decode_seq_bwd_loop when it has formed u_all beside d_uk, consumed by the weight-gradient function of the same
// backward (same scratch, same host thread)
static thread_local const float* g_u_all_ready = nullptr;
struct CgruBwdScratch {
    float *wcatT, *wpT, *whh1T, *dgi2, *dqgh, *dalpha, *ds, *dgi1, *dgh1, *dh1d, *carry, *de, *dvp, *dwp, *dah, *dencwp, *pbuf;
    float *u_all, *du_all, *duk;          // (R,H) context2hid(c_t); (R,H) its gradient dgi2 W_ih2; (B,Ts,H) gradient of uk
    int64_t total;
};
static CgruBwdScratch cgru_bwd_scratch(float* p, int64_t B, int64_t Ts, int64_t Tt, int64_t E, int64_t H) {
    CgruBwdScratch w;
    const int64_t C = 2 * H, Q = C + 3 * H, R = Tt * B;
    int64_t o = 0;
    auto take = [&](int64_t n) { float* q = p ? p + o : nullptr; o += (n + 63) & ~63ll; return q; };
    w.wcatT = take(Q * H); w.wpT = take(3 * H * C); w.whh1T = take(3 * H * H);
    w.dgi2 = take(R * 3 * H); w.dqgh = take(R * Q);
    w.dalpha = take(B * Ts); w.ds = take(R * Ts);
    w.dgi1 = take(R * 3 * H); w.dgh1 = take(R * 3 * H);
    w.dh1d = take(B * H); w.carry = take(B * H); w.de = take(R * E); w.dvp = take(VAG_POST_CHUNKS(Ts) * B * C);
    w.dwp = take(3 * H * C);
    w.dah = take(R * Ts);                   // d alpha through the head's use of the context, all steps
    w.dencwp = take(B * Ts * 3 * H);        // gradient of the projected keys
    w.pbuf = take(B * H);                   // dgh2 W_hh2 + carry, computed beside the attention backward
    w.u_all = take(R * H); w.du_all = take(R * H); w.duk = take(B * Ts * H);
    w.total = o;
    return w;
}
// where the decoder's backward keeps d(embedded inputs) (R,E) inside its scratch: a step driver lets the head's backward write its
// share straight there (vag_cgru_bwd_weights_chunk then adds gru_1's share without a copy)
}  // extern "C"
float* vag_cgru_bwd_scratch_de(float* scratch, int64_t B, int64_t Ts, int64_t Tt, int64_t E, int64_t H) {
    return cgru_bwd_scratch(scratch, B, Ts, Tt, E, H).de;
}
float* vag_cgru_bwd_scratch_du(float* scratch, int64_t B, int64_t Ts, int64_t Tt, int64_t E, int64_t H) {     // (Tt*B, H): dgi2 W_ih2
    return cgru_bwd_scratch(scratch, B, Ts, Tt, E, H).du_all;
}
extern "C" {
int64_t vag_cgru_bwd_scratch_floats(int64_t B, int64_t Ts, int64_t Tt, int64_t E, int64_t H) {
    return cgru_bwd_scratch(nullptr, B, Ts, Tt, E, H).total;
}

int vag_cgru_attn_decode_seq_bwd_loop(const float* enc, const float* pe, const float* mask, const float* h0, const int64_t* tok,
                                 vag_dec_w w, int64_t B, int64_t Ts, int64_t Tt, int64_t E, int64_t H, int64_t V,
                                 const float* h2_all, const float* c_all, const float* e_all, float* d_h2_all,
                                 float* d_c_all, const float* d_e_all, float* ws, float* d_enc_out, int accumulate_enc,
                                 float* d_pe, float* d_h0, float* scratch, vag_stream_t stream) {
    hipStream_t s = S_(stream);
    VAG_CHECK_ARG(enc && pe && mask && h0 && tok && h2_all && c_all && e_all && d_h2_all && d_c_all && ws && d_enc_out &&
                  d_pe && d_h0 && scratch && dec_w_ok(w));
    VAG_CHECK_ARG(B > 0 && Ts > 0 && Tt > 0 && E % 4 == 0 && H % 4 == 0 && aligned16(ws) && aligned16(scratch));
    (void)V; (void)mask; (void)d_e_all;
    const int64_t C = 2 * H, Q = C + 3 * H, BH = B * H;
    CgruWs k = cgru_ws(ws, B, Ts, Tt, E, H);
    CgruPrep p = cgru_prep(k.prep, H);          // Wcat / Wp from the forward call are still in the workspace
    CgruBwdScratch z = cgru_bwd_scratch(scratch, B, Ts, Tt, E, H);
    const bool s16 = g_store16;
    VAG_CHECK_ARG(!s16 || (g_derived && H % 8 == 0));
    if (g_derived) {
        DerivedW dw = derived_layout(const_cast<float*>(g_derived), H);
        p = cgru_prep(dw.prep, H);
        z.wcatT = dw.wcatT; z.whh1T = dw.whh1T;
        if (s16) {      // the same two transposes as fp16 (element strides are unchanged)
            z.wcatT = const_cast<float*>(as_f(dw.wcatT16)); z.whh1T = const_cast<float*>(as_f(dw.whh1T16));
        }
    } else {
        // per-step products are written as x W^T, so transpose the (derived) weights once per call
        VAG_TRY(vag_transpose_launch(p.wcat, Q, H, z.wcatT, s));            // (H, C+3H) = [attn_h^T | W_hh2^T]
        VAG_TRY(vag_transpose_launch(w.gru1.w_hh, 3 * H, H, z.whh1T, s));   // (H, 3H)
    }
    const bool persist = !s16 && vag_opt().persistent && vag_opt().persistent_dec_bwd && vag_dec_bwd_persistent_ok(B, Ts, Tt, H);
    // gru_2 cell backward of the last step: nothing arrives from a later step
    if (!persist) {
        GruBwdArgs a = {};
        a.ld_add = H; a.ldh = H; a.ldgi = 3 * H; a.ldgh = Q; a.M = (int)B; a.H = (int)H;
        const int64_t t = Tt - 1;
        GruBwdSide& sd = a.s[0];
        sd.dh_carry = nullptr; sd.dh_add = d_h2_all + t * BH; sd.drop_idx0 = 0;
        sd.save = k.g2 + t * 4 * BH; sd.hprev = k.h1 + t * BH;
        sd.dgi = z.dgi2 + t * B * 3 * H; sd.dgh = z.dqgh + t * B * Q + C; sd.dh_prev = z.dh1d; sd.t = 0;
        VAG_TRY(vag_gru_bwd_elem_launch(a, 1, s));
    }
    GruBwdStepArgs f = {};
    f.ld_add = H; f.ldh = H; f.M = (int)B; f.H = (int)H; f.lengths = nullptr; f.rng = nullptr; f.sid = 0; f.p = 0.f;
    // The context reaches the loss through the head (d_c_all) and through gru_2's input projection.  The first part of
    // d alpha does not depend on the recurrence: all steps at once, before the loop.  The second is taken on the
    // projected keys, d alpha[b,s] += encwp[b,s,:] . dgi2[b,:], so no per-step product dc = dgi2 W is needed.
    // per sentence b: dah[:, b, :] (Tt,Ts) = d_c_all[:, b, :] (Tt,C) enc[b]^T (C,Ts): B small products in one launch
    if (C % 4 == 0 && aligned16(enc) && aligned16(d_c_all) && B < 65536)
        VAG_TRY(vag_skinny_batched_launch(B, Tt, Ts, C, d_c_all, B * C, C, enc, C, Ts * C, z.dah, B * Ts, Ts, s));
    else
        VAG_TRY(vag_attn_scores_ex_launch(1, enc, d_c_all, C, nullptr, nullptr, Tt * B, 1, B, Ts, C, nullptr, z.dah, s));
    if (persist) {
        // the whole backward recurrence in ONE launch (persist.hip); k.dal holds d alpha
        VAG_TRY(vag_dec_bwd_persistent_launch(pe, k.encwp, w.attn_v, z.wcatT, z.whh1T, h0, h2_all, k.h1, k.g1, k.g2, k.qhp, k.alpha,
                                              d_h2_all, z.dah, z.dgi2, z.dqgh, z.ds, z.dgi1, z.dgh1, d_h0, k.dal, k.sync_b, B, Ts, Tt, H,
                                              s));
    }
    for (int64_t t = Tt - 1; t >= 0 && !persist; --t) {
        float* dgi2 = z.dgi2 + t * B * 3 * H;
        float* dqgh = z.dqgh + t * B * Q;
        // attention backward: d alpha ; softmax backward ; dq = sum_s ds v (1 - tanh^2)
        // ... in one grid with the hidden-side half of dh1 (dgh2 W_hh2 + z2*dh2), whose operand is already known
        const float* wside = s16 ? as_f(reinterpret_cast<const vag_half*>(z.wcatT) + C) : z.wcatT + C;
        VAG_TRY(vag_attn_dot_side_launch(1, k.encwp, dgi2, 3 * H, nullptr, nullptr, z.dah + t * B * Ts, B, Ts, 3 * H, z.dalpha,
                                         B, H, 3 * H, dqgh + C, Q, wside, Q, nullptr, z.dh1d, z.pbuf, H, s, s16));
        VAG_TRY(vag_attn_dq_launch(pe, k.qhp + t * B * Q, Q, w.attn_v, k.alpha + t * B * Ts, z.dalpha, z.ds + t * B * Ts, B,
                                   Ts, C, dqgh, Q, s, s16));
        // dh1 = dq attn_h + (dgh2 W_hh2 + z2*dh2)   -> gru_1 cell backward of this step
        f.lda = Q; f.ldw = Q; f.K = (int)C; f.ldgi = 3 * H; f.ldgh = 3 * H; f.has_cell = 1;
        GruBwdStepSide& sd = f.s[0];
        sd.A = dqgh; sd.WT = z.wcatT; sd.addend = z.pbuf; sd.dh_add = nullptr; sd.drop_idx0 = 0;
        sd.save = k.g1 + t * 4 * BH; sd.hprev = (t == 0) ? h0 : h2_all + (t - 1) * BH;
        sd.dgi = z.dgi1 + t * B * 3 * H; sd.dgh = z.dgh1 + t * B * 3 * H; sd.dh_direct = z.carry; sd.dh_out = nullptr; sd.t = 0;
        VAG_TRY(vag_gru_bwd_step_launch(f, 1, s, s16));
        // d h2[t-1] = dgh1 W_hh1 + z1*dh1 (+ head's d_h2[t-1])   -> gru_2 cell backward of step t-1 (or d_h0 at t = 0)
        f.lda = 3 * H; f.ldw = 3 * H; f.K = (int)(3 * H);
        sd.A = z.dgh1 + t * B * 3 * H; sd.WT = z.whh1T; sd.addend = z.carry;
        if (t > 0) {
            const int64_t t1 = t - 1;
            f.has_cell = 1; f.ldgi = 3 * H; f.ldgh = Q;
            sd.dh_add = d_h2_all + t1 * BH;
            sd.save = k.g2 + t1 * 4 * BH; sd.hprev = k.h1 + t1 * BH;
            sd.dgi = z.dgi2 + t1 * B * 3 * H; sd.dgh = z.dqgh + t1 * B * Q + C; sd.dh_direct = z.dh1d; sd.dh_out = nullptr;
        } else {
            f.has_cell = 0;
            sd.dh_add = nullptr; sd.save = nullptr; sd.hprev = nullptr; sd.dgi = nullptr; sd.dgh = nullptr;
            sd.dh_direct = nullptr; sd.dh_out = d_h0;
        }
        VAG_TRY(vag_gru_bwd_step_launch(f, 1, s, s16));
    }
    // after the loop: everything that does not sit on the recurrence's critical path, as large products
    VAG_TRY(vag_attn_post_bwd_launch(pe, k.qhp, Q, w.attn_v, z.ds, k.alpha, d_c_all, B, Ts, Tt, C, d_pe, z.dvp, d_enc_out,
                                     accumulate_enc, s, s16));
    // gru_2's share of d_enc, through the projected keys encwp = (enc W_c2h^T) W_ih2^T: du_t = dgi2_t W_ih2 for all steps (H wide),
    // d_uk[b,s] = sum_t alpha[t,b,s] du_t[b], d_enc += d_uk W_c2h.  (Rounds 2-4: d_enc += (sum_t alpha_t dgi2_t) (W_ih2 W_c2h), 8 GFLOP.)
    {
        VagGemmGroup now;       // du is read by the weighted sum right below: launched at once (with whatever was queued before)
        VAG_TRY(gemm_nn(Tt * B, H, 3 * H, z.dgi2, 3 * H, w.gru2.w_ih, H, 0.f, z.du_all, H, s));
        VAG_TRY(now.end(s));
    }
    // (in the same launch: u_t = sum_s alpha[t,b,s] uk[b,s] for all steps, which the weight-gradient function needs -- forward data only)
    VAG_TRY(vag_attn_wsum_pair_launch(k.alpha, z.du_all, H, z.duk, k.uk, H, z.u_all, B, Ts, Tt, s));
    g_u_all_ready = z.u_all;
    return gemm_nn(B * Ts, C, H, z.duk, H, w.c2h, C, 1.f, d_enc_out, C, s);
}

// Parameter gradients of the decoder from the per-step tensors the loop left in `scratch` (large products; nothing
// downstream waits for them, so they may run on a side stream beside the encoder's backward recurrence).
int vag_cgru_attn_decode_seq_bwd_weights(const float* h0, const int64_t* tok, vag_dec_w w, int64_t B, int64_t Ts, int64_t Tt,
                                         int64_t E, int64_t H, const float* h2_all, const float* c_all, const float* e_all,
                                         const float* d_e_all, float* ws, vag_dec_g g, float* scratch, vag_stream_t stream) {
    hipStream_t s = S_(stream);
    VAG_CHECK_ARG(h0 && tok && h2_all && c_all && e_all && ws && scratch && dec_w_ok(w));
    VAG_CHECK_ARG(g.emb && g.gru1.w_ih && g.gru1.w_hh && g.gru1.b_ih && g.gru1.b_hh && g.attn_h && g.attn_v && g.c2h &&
                  g.gru2.w_ih && g.gru2.w_hh && g.gru2.b_ih && g.gru2.b_hh);
    const int64_t C = 2 * H;
    CgruBwdScratch z = cgru_bwd_scratch(scratch, B, Ts, Tt, E, H);
    VagGemmGroup grp3;              // the independent K = Tt*B weight gradients go out as one grouped launch,
    VAG_TRY(vag_colsum_launch(z.dvp, VAG_POST_CHUNKS(Ts) * B, C, C, g.attn_v, s));      // the bias sums as another
    VAG_TRY(vag_cgru_bwd_weights_chunk(h0, tok, w, B, Ts, Tt, E, H, h2_all, c_all, e_all, d_e_all, ws, g, scratch, 0, Tt, true,
                                       s));
    VAG_TRY(grp3.end(s));      // z.dwp and z.de are complete from here on
    VAG_TRY(vag_cgru_bwd_weights_scatter(tok, B, Ts, Tt, E, H, g, scratch, 0, Tt, s));
    return vag_cgru_bwd_weights_finish(w, B, Ts, Tt, E, H, g, scratch, false, s);
}
}  // extern "C"

// Rows of the time steps [t0, t1).  The products are queued when the caller holds a group bracket; the embedding-gradient
// tail (d(embedded inputs) -> scatter) reads nothing the queued products write.
int vag_cgru_bwd_weights_chunk(const float* h0, const int64_t* tok, vag_dec_w w, int64_t B, int64_t Ts, int64_t Tt, int64_t E,
                               int64_t H, const float* h2_all, const float* c_all, const float* e_all, const float* d_e_all,
                               float* ws, vag_dec_g g, float* scratch, int64_t t0, int64_t t1, bool first, hipStream_t s) {
    VAG_CHECK_ARG(0 <= t0 && t0 < t1 && t1 <= Tt);
    const int64_t C = 2 * H, Q = C + 3 * H, r0 = t0 * B, n = (t1 - t0) * B;
    CgruWs k = cgru_ws(ws, B, Ts, Tt, E, H);
    CgruBwdScratch z = cgru_bwd_scratch(scratch, B, Ts, Tt, E, H);
    const float* dqgh = z.dqgh + r0 * Q;
    const float* dgh2 = dqgh + C;        // (n,3H) row stride Q
    const float* dgi2 = z.dgi2 + r0 * 3 * H;
    const float* dgi1 = z.dgi1 + r0 * 3 * H;
    const float* dgh1 = z.dgh1 + r0 * 3 * H;
    const float* h1 = k.h1 + r0 * H;
    VAG_TRY(gemm_tn_acc(3 * H, H, n, dgh2, Q, h1, H, g.gru2.w_hh, H, s, g.gru2.b_hh));       // (each with its bias gradient)
    VAG_TRY(gemm_tn_acc(C, H, n, dqgh, Q, h1, H, g.attn_h, H, s));
    // gru_2's input side, gi2_t = W_ih2 u_t with u_t = W_c2h c_t = sum_s alpha[t,b,s] uk[b,s] (formed here for all steps by one
    // weighted sum): d W_ih2 += dgi2^T u (with the bias gradient), d W_c2h += du^T c  (du = dgi2 W_ih2: the loop function left it)
    if (first) {
        if (g_u_all_ready != z.u_all) VAG_TRY(vag_attn_wsum_launch(1, k.alpha, k.uk, B, Ts, Tt, H, z.u_all, s));
        g_u_all_ready = nullptr;        // (the loop function of the same backward left it: see there)
    }
    VAG_TRY(gemm_tn_acc(3 * H, H, n, dgi2, 3 * H, z.u_all + r0 * H, H, g.gru2.w_ih, H, s, g.gru2.b_ih));
    VAG_TRY(gemm_tn_acc(H, C, n, z.du_all + r0 * H, H, c_all + r0 * C, C, g.c2h, C, s));
    if (h0 + B * H == h2_all) {
        // caller keeps [h0, h2_all] in one buffer: the previous states of all steps are one (R,H) operand
        VAG_TRY(gemm_tn_acc(3 * H, H, n, dgh1, 3 * H, h0 + r0 * H, H, g.gru1.w_hh, H, s, g.gru1.b_hh));
    } else {
        if (t0 == 0) VAG_TRY(gemm_tn_acc(3 * H, H, B, dgh1, 3 * H, h0, H, g.gru1.w_hh, H, s, g.gru1.b_hh));
        const int64_t skip = t0 == 0 ? B : 0;
        if (n > skip)
            VAG_TRY(gemm_tn_acc(3 * H, H, n - skip, dgh1 + skip * 3 * H, 3 * H, h2_all + (r0 + skip - B) * H, H, g.gru1.w_hh, H,
                                s, g.gru1.b_hh));      // (the two row ranges add their own shares of the bias gradient)
    }
    VAG_TRY(gemm_tn_acc(3 * H, E, n, dgi1, 3 * H, e_all + r0 * E, E, g.gru1.w_ih, E, s, g.gru1.b_ih));
    // d(embedded inputs) = dgi1 W_ih1 (+ the head's W3 path), scattered into the embedding gradient
    // (z.de: the sum; a caller that hands over d_e_all with the scratch's own address -- the step driver, whose head backward
    // writes there -- saves the copy)
    float* de = z.de + r0 * E;
    if (d_e_all && d_e_all != z.de) VAG_TRY(copy_async(de, d_e_all + r0 * E, n * E * sizeof(float), s));
    VAG_TRY(gemm_nn(n, E, 3 * H, dgi1, 3 * H, w.gru1.w_ih, E, d_e_all ? 1.f : 0.f, de, E, s));
    return VAG_OK;
}
// After every chunk's products have been LAUNCHED (group brackets closed): the embedding scatter of [t0, t1) ...
int vag_cgru_bwd_weights_scatter(const int64_t* tok, int64_t B, int64_t Ts, int64_t Tt, int64_t E, int64_t H, vag_dec_g g,
                                 float* scratch, int64_t t0, int64_t t1, hipStream_t s) {
    CgruBwdScratch z = cgru_bwd_scratch(scratch, B, Ts, Tt, E, H);
    return vag_embed_scatter_launch(tok + t0 * B, B, 1, t1 - t0, B, z.de + t0 * B * E, E, g.emb, nullptr, 0, 0.f, s);
}
// ... and, once all chunks are in: the attention vector's gradient and the chain rule through the folded product.
int vag_cgru_bwd_weights_finish(vag_dec_w w, int64_t B, int64_t Ts, int64_t Tt, int64_t E, int64_t H, vag_dec_g g,
                                float* scratch, bool with_attn_v, hipStream_t s) {
    const int64_t C = 2 * H;
    CgruBwdScratch z = cgru_bwd_scratch(scratch, B, Ts, Tt, E, H);
    (void)w;
    if (with_attn_v) VAG_TRY(vag_colsum_launch(z.dvp, VAG_POST_CHUNKS(Ts) * B, C, C, g.attn_v, s));
    return VAG_OK;              // (rounds 2-4: the chain rule through the folded product W_ih2 W_c2h, two more products)
}
extern "C" {

int vag_cgru_attn_decode_seq_bwd(const float* enc, const float* pe, const float* mask, const float* h0, const int64_t* tok,
                                 vag_dec_w w, int64_t B, int64_t Ts, int64_t Tt, int64_t E, int64_t H, int64_t V,
                                 const float* h2_all, const float* c_all, const float* e_all, float* d_h2_all,
                                 float* d_c_all, const float* d_e_all, float* ws, float* d_enc_out, int accumulate_enc,
                                 float* d_pe, float* d_h0, vag_dec_g g, float* scratch, vag_stream_t stream) {
    VAG_TRY(vag_cgru_attn_decode_seq_bwd_loop(enc, pe, mask, h0, tok, w, B, Ts, Tt, E, H, V, h2_all, c_all, e_all, d_h2_all,
                                              d_c_all, d_e_all, ws, d_enc_out, accumulate_enc, d_pe, d_h0, scratch, stream));
    return vag_cgru_attn_decode_seq_bwd_weights(h0, tok, w, B, Ts, Tt, E, H, h2_all, c_all, e_all, d_e_all, ws, g, scratch,
                                                stream);
}

int64_t vag_cgru_step_scratch_floats(int64_t N, int64_t Ts, int64_t E, int64_t H) {
    (void)E;
    return N * (3 * H + H + 5 * H + Ts) + 64;
}
int vag_cgru_attn_decode_step(const float* enc, const float* pe, const float* mask, int64_t rows_per_src, const int64_t* tok,
                              const float* h_in, vag_dec_w w, const float* prep, int64_t N, int64_t Ts, int64_t E, int64_t H,
                              float* h_out, float* c, float* e, float* alpha, float* scratch, vag_stream_t stream) {
    hipStream_t s = S_(stream);
    VAG_CHECK_ARG(enc && pe && mask && tok && h_in && h_out && c && e && alpha && scratch && prep && dec_w_ok(w));
    VAG_CHECK_ARG(N > 0 && Ts > 0 && E % 4 == 0 && H % 4 == 0 && rows_per_src >= 1 && aligned16(scratch) && aligned16(prep));
    const int64_t C = 2 * H, Q = C + 3 * H;
    CgruPrep p = cgru_prep(const_cast<float*>(prep), H);
    float* q = scratch;
    float* xp1 = q; q += N * 3 * H;
    float* h1 = q; q += N * H;
    float* qhp = q; q += N * Q;
    float* scores = q;
    if (N <= 256 && aligned16(w.emb) && aligned16(w.gru1.w_ih) && aligned16(e)) {                          // :118, one launch
        VAG_TRY(vag_skinny_gather_launch(N, 3 * H, E, w.emb, E, tok, w.gru1.w_ih, E, w.gru1.b_ih, xp1, 3 * H, e, E, s));
    } else {
        VAG_TRY(vag_embed_gather_launch(tok, 1, 0, N, 1, w.emb, E, e, nullptr, 0, 0.f, s));
        VAG_TRY(linear_fwd(N, 3 * H, E, e, E, w.gru1.w_ih, w.gru1.b_ih, 0, xp1, 3 * H, s));
    }
    StepBufs b;
    b.xp1 = xp1; b.hprev = h_in; b.h1 = h1; b.g1 = nullptr; b.g2 = nullptr; b.qhp = qhp; b.scores = scores;
    b.alpha = alpha; b.c = c; b.h2 = h_out;
    return cgru_step(enc, pe, mask, rows_per_src, w, p, N, Ts, H, b, s);
}

// The decoding step in its HOISTED form (round 4): what the training chain does since round 2 -- the keys as gru_2 sees them are
// projected once per call (encwp = (W_ih2 W_c2h) enc), so the second cell has no product left and rides as the epilogue of the
// softmax + weighted-sum kernel -- plus the head's share of the context as a second weighted sum in the same launch
// (cw = alpha . (enc W2^T)): the context itself is never formed.  Four launches instead of five, and the head's product over
// C = 2H columns is gone.  vag_cgru_decode_keys: once per decode call; `keys` = [encwp (B,Ts,3H) | encw2 (B,Ts,E)].
int64_t vag_cgru_decode_keys_floats(int64_t B, int64_t Ts, int64_t E, int64_t H) {
    return ((B * Ts * 3 * H + 63) & ~63ll) + ((B * Ts * E + 63) & ~63ll);
}
int vag_cgru_decode_keys(const float* enc, const float* prep, const float* w2, int64_t B, int64_t Ts, int64_t E, int64_t H,
                         float* keys, vag_stream_t stream) {
    hipStream_t s = S_(stream);
    VAG_CHECK_ARG(enc && prep && w2 && keys && B > 0 && Ts > 0 && E % 4 == 0 && H % 4 == 0 && aligned16(keys) && aligned16(prep));
    const int64_t C = 2 * H;
    CgruPrep p = cgru_prep(const_cast<float*>(prep), H);
    float* encw2 = keys + ((B * Ts * 3 * H + 63) & ~63ll);
    VagGemmGroup grp;
    VAG_TRY(vag_gemm_launch(B * Ts, 3 * H, C, 1.f, enc, C, 1, p.wp, 1, C, 0.f, keys, 3 * H, nullptr, 0, s));
    VAG_TRY(vag_gemm_launch(B * Ts, E, C, 1.f, enc, C, 1, w2, 1, C, 0.f, encw2, E, nullptr, 0, s));
    return grp.end(s);
}
// ... and what a step needs of a TOKEN, for every vocabulary entry, once per call: tables = [emb W_ih1^T + b_ih1 (V,3H) | emb W3^T (V,E)]
// (the free-running recurrence kernel's tables, persist.hip).  With them a step has no embedding / input-projection launch: gru_1 reads
// its input projection from the line its row's token picks, the head its share of the embedded token likewise.
int64_t vag_cgru_decode_tables_floats(int64_t V, int64_t E, int64_t H) { return ((V * 3 * H + 63) & ~63ll) + ((V * E + 63) & ~63ll); }
int vag_cgru_decode_tables(vag_dec_w w, const float* w3, int64_t V, int64_t E, int64_t H, float* tables, vag_stream_t stream) {
    hipStream_t s = S_(stream);
    VAG_CHECK_ARG(dec_w_ok(w) && w3 && tables && V > 0 && E % 4 == 0 && H % 4 == 0 && aligned16(tables));
    VagGemmGroup grp;
    VAG_TRY(vag_gemm_launch(V, 3 * H, E, 1.f, w.emb, E, 1, w.gru1.w_ih, 1, E, 0.f, tables, 3 * H, w.gru1.b_ih, 0, s));
    VAG_TRY(vag_gemm_launch(V, E, E, 1.f, w.emb, E, 1, w3, 1, E, 0.f, tables + ((V * 3 * H + 63) & ~63ll), E, nullptr, 0, s));
    return grp.end(s);
}
int vag_cgru_attn_decode_step_h(const float* pe, const float* mask, const float* keys, const float* tables, int64_t V,
                                int64_t rows_per_src, const int64_t* tok, const float* h_in, vag_dec_w w, const float* prep,
                                int64_t N, int64_t Ts, int64_t E, int64_t H, float* h_out, float* cw, float* e, float* alpha,
                                float* scratch, vag_stream_t stream) {
    hipStream_t s = S_(stream);
    VAG_CHECK_ARG(pe && mask && keys && tok && h_in && h_out && cw && (e || tables) && alpha && scratch && prep && dec_w_ok(w));
    VAG_CHECK_ARG(!tables || (V > 0 && aligned16(tables)));
    VAG_CHECK_ARG(N > 0 && N <= 256 && Ts > 0 && E % 4 == 0 && H % 4 == 0 && rows_per_src >= 1 && N % rows_per_src == 0 &&
                  aligned16(scratch) && aligned16(prep) && aligned16(keys) && aligned16(w.emb) && aligned16(w.gru1.w_ih) && aligned16(e));
    const int64_t C = 2 * H, Q = C + 3 * H, Bs = N / rows_per_src;
    CgruPrep p = cgru_prep(const_cast<float*>(prep), H);
    const float* encw2 = keys + ((Bs * Ts * 3 * H + 63) & ~63ll);
    float* q = scratch;
    float* xp1 = q; q += N * 3 * H;
    float* h1 = q; q += N * H;
    float* qhp = q; q += N * Q;
    float* scores = q;
    if (!tables) VAG_TRY(vag_skinny_gather_launch(N, 3 * H, E, w.emb, E, tok, w.gru1.w_ih, E, w.gru1.b_ih, xp1, 3 * H, e, E, s));      // :118
    GruStepArgs a = {};
    a.lda = H; a.ldw = H; a.ldother = 3 * H; a.ldh = H; a.ld2 = 0;
    a.M = (int)N; a.K = (int)H; a.H = (int)H; a.lengths = nullptr; a.comp_hidden = 1;
    a.s[0].A = h_in; a.s[0].W = w.gru1.w_hh; a.s[0].bias = w.gru1.b_hh; a.s[0].other = tables ? tables : xp1;
    a.s[0].other_idx = tables ? tok : nullptr;
    a.s[0].hprev = h_in; a.s[0].hout = h1; a.s[0].out2 = nullptr; a.s[0].save = nullptr; a.s[0].t = 0;
    VAG_TRY(vag_gru_step_launch(a, 1, s));                                                                               // gru_1 :121
    // (the training chain lets W_hh2 h1 ride in the score kernel's grid; at 192 rows the one (q | hp2) product + the plain score
    // kernel measured 3.5 us less than q + scores-with-rider: the rider is a long K loop of few workgroups)
    VAG_TRY(vag_skinny_launch(N, Q, H, h1, H, p.wcat, H, p.bcat, nullptr, 0, qhp, Q, 0, s));                  // attn_h(h1) :47 | W_hh2 h1 + b
    if (vag_attn_row_gru_ok(N, Ts, H, E) && aligned16(pe) && aligned16(w.attn_v) && aligned16(w.gru2.b_ih) && aligned16(h_out) &&
        aligned16(cw))                                                                                        // :47-51, :41-44, :126-129
        return vag_attn_row_gru_launch(pe, qhp, Q, w.attn_v, mask, keys, encw2, N, rows_per_src, Ts, H, E, w.gru2.b_ih, qhp + C, Q, h1,
                                       alpha, h_out, cw, s);
    VAG_TRY(vag_attn_scores_launch(0, pe, qhp, Q, w.attn_v, mask, N, rows_per_src, Ts, C, scores, s));        // :47-51, :41-43
    return vag_attn_ctx_gru_launch(scores, keys, N, rows_per_src, Ts, H, w.gru2.b_ih, qhp + C, Q, h1, alpha, h_out, nullptr, s,
                                   false, encw2, E, cw);                                                                 // :44, :126-129
}

// =====================================================================================================
// output head + cross entropy
// =====================================================================================================
// tmid (R,E) = dropout(tanh(W1 h2 + b1 + W2 c + b2 + W3 e + b3)) for all R = Tt*B rows (NMT_Decoder.py:137-141): the three
// products accumulate into a zeroed tmid from ONE grouped launch (each alone is 40 tiles), then one elementwise pass.
static int head_pre_seq(const float* h2, const float* c, const float* e, const vag_head_w& w, int64_t R, int64_t E, int64_t H,
                        float p_out, const uint64_t* rng, float* tmid, hipStream_t s) {
    const int64_t C = 2 * H;
    if (!g_step_zeroed) VAG_TRY(zero_async(tmid, R * E * sizeof(float), s));
    VagGemmGroup grp4;
    VAG_TRY(vag_gemm_launch(R, E, H, 1.f, h2, H, 1, w.w1, 1, H, 1.f, tmid, E, w.b1, 0, s));
    VAG_TRY(vag_gemm_launch(R, E, C, 1.f, c, C, 1, w.w2, 1, C, 1.f, tmid, E, w.b2, 0, s));
    VAG_TRY(vag_gemm_launch(R, E, E, 1.f, e, E, 1, w.w3, 1, E, 1.f, tmid, E, w.b3, 0, s));
    VAG_TRY(grp4.end(s));
    return vag_tanh_dropout_launch(tmid, R * E, 0, rng, VAG_DROP_DEC_OUT, p_out, s);
}

int vag_head_ce_seq_fwd(const float* h2_all, const float* c_all, const float* e_all, vag_head_w w, const int64_t* tgt,
                        const float* vocab_weight, int64_t B, int64_t Tt, int64_t E, int64_t H, int64_t V, float p_out,
                        const uint64_t* rng, int logits_ready, float* tmid, float* logits, int64_t ldl, float* lse,
                        float* nll, float* inv_cnt, float* loss_mt, vag_stream_t stream) {
    VAG_CHECK_ARG(loss_mt != nullptr);
    return vag_head_ce_seq_fwd_impl(h2_all, c_all, e_all, w, tgt, vocab_weight, B, Tt, E, H, V, p_out, rng, logits_ready, tmid,
                                    logits, ldl, lse, nll, inv_cnt, 0, loss_mt, nullptr, 0.f, 0.f, 0, S_(stream));
}
}  // extern "C"
// inv_cnt_ready: the caller has filled inv_cnt already (step prologue).  losses != NULL: losses[1] = loss_mt and the
// weighted total losses[0] are written instead of loss_mt (one launch for both, V11.py:164-166).
int vag_head_ce_seq_fwd_impl(const float* h2_all, const float* c_all, const float* e_all, vag_head_w w, const int64_t* tgt,
                             const float* vocab_weight, int64_t B, int64_t Tt, int64_t E, int64_t H, int64_t V, float p_out,
                             const uint64_t* rng, int logits_ready, float* tmid, float* logits, int64_t ldl, float* lse,
                             float* nll, float* inv_cnt, int inv_cnt_ready, float* loss_mt, float* losses, float w_mt,
                             float w_vse, int has_vse, hipStream_t s) {
    VAG_CHECK_ARG(h2_all && c_all && e_all && tgt && vocab_weight && tmid && logits && lse && nll && inv_cnt &&
                  (loss_mt || losses));
    VAG_CHECK_ARG(w.w1 && w.b1 && w.w2 && w.b2 && w.w3 && w.b3 && w.out_w && w.out_b);
    VAG_CHECK_ARG(B > 0 && Tt > 0 && E % 4 == 0 && H % 4 == 0 && V > 0 && ldl >= V && ldl % 4 == 0);
    const int64_t R = Tt * B;
    if (!inv_cnt_ready) VAG_TRY(vag_inv_cnt_launch(tgt, B, Tt, inv_cnt, s));
    const int64_t CH = (!logits_ready && g_head_chunk > 0 && g_head_chunk < R) ? (g_head_chunk + B - 1) / B * B : 0;
    if (CH > 0) {
        VAG_TRY(head_pre_seq(h2_all, c_all, e_all, w, R, E, H, p_out, rng, tmid, s));
        for (int64_t r0 = 0; r0 < R; r0 += CH) {        // chunks start on a time-step boundary: row r0 = step r0 / B
            const int64_t rows = R - r0 < CH ? R - r0 : CH;
            VAG_TRY(vag_gemm_launch(rows, V, E, 1.f, tmid + r0 * E, E, 1, w.out_w, 1, E, 0.f, logits, ldl, w.out_b, 0, s));
            VAG_TRY(vag_lse_nll_launch(logits, ldl, rows, V, tgt + r0 / B, B, Tt, vocab_weight, lse + r0, nll + r0, nullptr, 0,
                                       nullptr, 0, s));
            if (g_head_fuse.g) {
                const vag_head_g& g = *g_head_fuse.g;
                void* dl16 = rows > 128 ? head_dl16_slot(logits, ldl, R, CH, E) : nullptr;
                VAG_TRY(vag_ce_bwd_colsum_launch(logits, ldl, rows, V, tgt + r0 / B, B, Tt, vocab_weight, lse + r0, inv_cnt,
                                                 g_head_fuse.d_loss, g.out_b, s, dl16));
                VAG_TRY(head_dt_gemm(rows, E, V, logits, ldl, w.out_w, g_head_fuse.dt + r0 * E, s, dl16));
                VAG_TRY(head_outw_gemm(V, E, rows, logits, ldl, tmid + r0 * E, g.out_w, s, dl16));
            }
        }
        if (g_head_fuse.g) g_head_fuse.done = true;
    } else {
        if (!logits_ready) {
            VAG_TRY(head_pre_seq(h2_all, c_all, e_all, w, R, E, H, p_out, rng, tmid, s));
            VAG_TRY(vag_gemm_launch(R, V, E, 1.f, tmid, E, 1, w.out_w, 1, E, 0.f, logits, ldl, w.out_b, 0, s));
        }
        VAG_TRY(vag_lse_nll_launch(logits, ldl, R, V, tgt, B, Tt, vocab_weight, lse, nll, nullptr, 0, nullptr, 0, s));
    }
    if (losses) return vag_loss_mt_mix_launch(nll, inv_cnt, B, Tt, losses, w_mt, w_vse, has_vse, s);
    return vag_loss_mt_launch(nll, inv_cnt, B, Tt, loss_mt, s);
}
extern "C" {

// d(logits) -> d(tmid) -> through dropout+tanh -> input gradients.  dt (R,E) is left holding d(pre-activation).
static int head_bwd_data(const vag_head_w& w, int64_t R, int64_t E, int64_t H, int64_t V, float p_out, const uint64_t* rng,
                         const float* tmid, const float* dlogits, int64_t ldl, float* d_h2_all, float* d_c_all,
                         float* d_e_all, float* dt, hipStream_t s, bool dt_ready = false) {
    const int64_t C = 2 * H;
    if (!dt_ready) VAG_TRY(head_dt_gemm(R, E, V, dlogits, ldl, w.out_w, dt, s));
    VAG_TRY(vag_tanh_bwd_launch(tmid, dt, dt, R * E, rng, VAG_DROP_DEC_OUT, p_out, s));   // tmid holds tanh(.)*mul
    VagGemmGroup grp5;              // three independent products of d(pre-activation): one grouped launch
    VAG_TRY(gemm_nn(R, H, E, dt, E, w.w1, H, 0.f, d_h2_all, H, s));
    VAG_TRY(gemm_nn(R, C, E, dt, E, w.w2, C, 0.f, d_c_all, C, s));
    VAG_TRY(gemm_nn(R, E, E, dt, E, w.w3, E, 0.f, d_e_all, E, s));
    return grp5.end(s);
}
// Parameter gradients of the head from d(logits) and dt = d(pre-activation): nothing downstream waits for these,
// so they may run on a side stream beside the decoder's backward recurrence.
static int head_bwd_weights(const float* h2_all, const float* c_all, const float* e_all, int64_t R, int64_t E, int64_t H,
                            int64_t V, const float* tmid, const float* dlogits, int64_t ldl, const float* dt,
                            const vag_head_g& g, hipStream_t s, bool out_b_done = false, bool out_w_done = false) {
    const int64_t C = 2 * H;
    VagGemmGroup grp6;
    if (!out_w_done) VAG_TRY(head_outw_gemm(V, E, R, dlogits, ldl, tmid, g.out_w, s));
    if (!out_b_done) VAG_TRY(vag_colsum_launch(dlogits, R, V, ldl, g.out_b, s));
    // b1, b2, b3 enter the same sum (NMT_Decoder.py:137): each of the three products leaves sum_r dt[r,:] in one of them
    VAG_TRY(gemm_tn_acc(E, H, R, dt, E, h2_all, H, g.w1, H, s, g.b1));
    VAG_TRY(gemm_tn_acc(E, C, R, dt, E, c_all, C, g.w2, C, s, g.b2));
    VAG_TRY(gemm_tn_acc(E, E, R, dt, E, e_all, E, g.w3, E, s, g.b3));
    return grp6.end(s);
}

static bool head_g_ok(const vag_head_g& g) { return g.w1 && g.b1 && g.w2 && g.b2 && g.w3 && g.b3 && g.out_w && g.out_b; }

int vag_head_ce_seq_bwd_data(vag_head_w w, const int64_t* tgt, const float* vocab_weight, int64_t B, int64_t Tt, int64_t E,
                             int64_t H, int64_t V, float p_out, const uint64_t* rng, const float* tmid, float* logits,
                             int64_t ldl, const float* lse, const float* inv_cnt, const float* d_loss, float* d_h2_all,
                             float* d_c_all, float* d_e_all, float* scratch, vag_stream_t stream) {
    hipStream_t s = S_(stream);
    VAG_CHECK_ARG(tgt && vocab_weight && tmid && logits && lse && inv_cnt && d_loss && scratch && d_h2_all && d_c_all && d_e_all);
    VAG_CHECK_ARG(w.w1 && w.w2 && w.w3 && w.out_w);
    VAG_CHECK_ARG(B > 0 && Tt > 0 && E % 4 == 0 && H % 4 == 0 && V > 0 && ldl >= V && ldl % 4 == 0);
    const int64_t R = Tt * B;
    VAG_TRY(vag_ce_bwd_launch(logits, ldl, R, V, tgt, B, Tt, vocab_weight, lse, inv_cnt, d_loss, s));
    return head_bwd_data(w, R, E, H, V, p_out, rng, tmid, logits, ldl, d_h2_all, d_c_all, d_e_all, scratch, s);
}

int vag_head_bwd_weights(const float* h2_all, const float* c_all, const float* e_all, int64_t R, int64_t E, int64_t H,
                         int64_t V, const float* tmid, const float* dlogits, int64_t ldl, const float* dt, vag_head_g g,
                         vag_stream_t stream) {
    VAG_CHECK_ARG(h2_all && c_all && e_all && tmid && dlogits && dt && head_g_ok(g) && R > 0 && ldl >= V);
    return head_bwd_weights(h2_all, c_all, e_all, R, E, H, V, tmid, dlogits, ldl, dt, g, S_(stream));
}

int vag_head_ce_seq_bwd(const float* h2_all, const float* c_all, const float* e_all, vag_head_w w, const int64_t* tgt,
                        const float* vocab_weight, int64_t B, int64_t Tt, int64_t E, int64_t H, int64_t V, float p_out,
                        const uint64_t* rng, const float* tmid, float* logits, int64_t ldl, const float* lse,
                        const float* inv_cnt, const float* d_loss, float* d_h2_all, float* d_c_all, float* d_e_all,
                        vag_head_g g, float* scratch, vag_stream_t stream) {
    hipStream_t s = S_(stream);
    VAG_CHECK_ARG(h2_all && c_all && e_all && head_g_ok(g));
    VAG_CHECK_ARG(tgt && vocab_weight && tmid && logits && lse && inv_cnt && d_loss && scratch && d_h2_all && d_c_all && d_e_all);
    VAG_CHECK_ARG(w.w1 && w.w2 && w.w3 && w.out_w);
    VAG_CHECK_ARG(B > 0 && Tt > 0 && E % 4 == 0 && H % 4 == 0 && V > 0 && ldl >= V && ldl % 4 == 0);
    const int64_t R = Tt * B;
    const int64_t CH = (g_head_chunk > 0 && g_head_chunk < R) ? (g_head_chunk + B - 1) / B * B : 0;
    if (CH > 0) {
        // chunked head (see g_head_chunk): per chunk, logits again -> d(logits) in place (+ bias gradient) -> its share of
        // d(tmid) and of the out.weight gradient; then everything that no longer needs the logits, as in the unchunked path
        VAG_CHECK_ARG(w.out_b != nullptr);
        const bool fused = g_head_fuse.done && g_head_fuse.dt == scratch;       // the forward of this call did it all
        for (int64_t r0 = 0; r0 < R && !fused; r0 += CH) {
            const int64_t rows = R - r0 < CH ? R - r0 : CH;
            VAG_TRY(vag_gemm_launch(rows, V, E, 1.f, tmid + r0 * E, E, 1, w.out_w, 1, E, 0.f, logits, ldl, w.out_b, 0, s));
            void* dl16 = rows > 128 ? head_dl16_slot(logits, ldl, R, CH, E) : nullptr;
            VAG_TRY(vag_ce_bwd_colsum_launch(logits, ldl, rows, V, tgt + r0 / B, B, Tt, vocab_weight, lse + r0, inv_cnt, d_loss,
                                             g.out_b, s, dl16));
            VAG_TRY(head_dt_gemm(rows, E, V, logits, ldl, w.out_w, scratch + r0 * E, s, dl16));
            VAG_TRY(head_outw_gemm(V, E, rows, logits, ldl, tmid + r0 * E, g.out_w, s, dl16));
        }
        VAG_TRY(head_bwd_data(w, R, E, H, V, p_out, rng, tmid, logits, ldl, d_h2_all, d_c_all, d_e_all, scratch, s, true));
        return head_bwd_weights(h2_all, c_all, e_all, R, E, H, V, tmid, logits, ldl, scratch, g, s, true, true);
    }
    // d(logits) and the output-bias gradient in one pass over the logits
    VAG_TRY(vag_ce_bwd_colsum_launch(logits, ldl, R, V, tgt, B, Tt, vocab_weight, lse, inv_cnt, d_loss, g.out_b, s));
    VAG_TRY(head_bwd_data(w, R, E, H, V, p_out, rng, tmid, logits, ldl, d_h2_all, d_c_all, d_e_all, scratch, s));
    return head_bwd_weights(h2_all, c_all, e_all, R, E, H, V, tmid, logits, ldl, scratch, g, s, true);
}

}  // extern "C"

extern "C" {

int vag_head_logp_seq_fwd(const float* h2, const float* c, const float* e, vag_head_w w, int64_t R, int64_t E, int64_t H,
                          int64_t V, float p_out, const uint64_t* rng, float* tmid, float* logp, int64_t ldl,
                          vag_stream_t stream) {
    hipStream_t s = S_(stream);
    VAG_CHECK_ARG(h2 && c && e && tmid && logp && R > 0 && E % 4 == 0 && H % 4 == 0 && V > 0 && ldl >= V && ldl % 4 == 0);
    VAG_CHECK_ARG(w.w1 && w.b1 && w.w2 && w.b2 && w.w3 && w.b3 && w.out_w && w.out_b);
    VAG_TRY(head_pre_seq(h2, c, e, w, R, E, H, p_out, rng, tmid, s));
    VAG_TRY(linear_fwd(R, V, E, tmid, E, w.out_w, w.out_b, 0, logp, ldl, s));
    return vag_lse_nll_launch(logp, ldl, R, V, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, 0, logp, ldl, s);
}

int vag_head_logp_seq_bwd(const float* h2, const float* c, const float* e, vag_head_w w, int64_t R, int64_t E, int64_t H,
                          int64_t V, float p_out, const uint64_t* rng, const float* tmid, const float* logp, float* d_logp,
                          int64_t ldl, float* d_h2, float* d_c, float* d_e, vag_head_g g, float* scratch,
                          vag_stream_t stream) {
    hipStream_t s = S_(stream);
    VAG_CHECK_ARG(h2 && c && e && tmid && logp && d_logp && d_h2 && d_c && d_e && scratch && R > 0 && ldl >= V);
    VAG_CHECK_ARG(g.w1 && g.b1 && g.w2 && g.b2 && g.w3 && g.b3 && g.out_w && g.out_b);
    VAG_TRY(vag_logsoftmax_bwd_launch(logp, ldl, d_logp, ldl, R, V, s));
    VAG_TRY(head_bwd_data(w, R, E, H, V, p_out, rng, tmid, d_logp, ldl, d_h2, d_c, d_e, scratch, s));
    return head_bwd_weights(h2, c, e, R, E, H, V, tmid, d_logp, ldl, scratch, g, s);
}

int vag_head_logp_step(const float* h2, const float* c, const float* e, vag_head_w w, int64_t N, int64_t E, int64_t H,
                       int64_t V, float* logp, int64_t ldl, int64_t* argmax, float* scratch, vag_stream_t stream) {
    hipStream_t s = S_(stream);
    VAG_CHECK_ARG(h2 && c && e && logp && scratch && N > 0 && E % 4 == 0 && H % 4 == 0 && V > 0 && ldl >= V);
    VAG_CHECK_ARG(w.w1 && w.b1 && w.w2 && w.b2 && w.w3 && w.b3 && w.out_w && w.out_b);
    float* tmp = scratch;            // (N,E)
    float* tmid = scratch + N * E;   // (N,E)
    VAG_TRY(head_step(h2, c, e, w, N, E, H, V, 0.f, nullptr, 0, tmp, tmid, logp, ldl, s));
    VAG_TRY(vag_lse_nll_launch(logp, ldl, N, V, nullptr, 0, 0, nullptr, nullptr, nullptr, argmax, 1, logp, ldl, s));
    return VAG_OK;
}

// ... with the context share of a hoisted decoding step (vag_cgru_attn_decode_step_h) in place of the context
int vag_head_logp_step_h(const float* h2, const float* cw, const float* e, const float* tables, const int64_t* tok, vag_head_w w,
                         int64_t N, int64_t E, int64_t H, int64_t V, float* logp, int64_t ldl, int64_t* argmax, float* scratch,
                         vag_stream_t stream) {
    hipStream_t s = S_(stream);
    VAG_CHECK_ARG(h2 && cw && (tables ? tok != nullptr : e != nullptr) && logp && scratch && N > 0 && N <= 256 && E % 4 == 0 &&
                  H % 4 == 0 && V > 0 && ldl >= V);
    VAG_CHECK_ARG(w.w1 && w.b1 && w.b2 && w.w3 && w.b3 && w.out_w && w.out_b && aligned16(h2) && (tables || aligned16(e)) &&
                  aligned16(w.w1) && aligned16(w.w3));
    float* tmid = scratch + N * E;   // (N,E)
    VAG_TRY(head_pre_step_h(h2, cw, e, tables ? tables + ((V * 3 * H + 63) & ~63ll) : nullptr, tok, w, N, E, H, tmid, s));
    VAG_TRY(linear_fwd(N, V, E, tmid, E, w.out_w, w.out_b, 0, logp, ldl, s));
    return vag_lse_nll_launch(logp, ldl, N, V, nullptr, 0, 0, nullptr, nullptr, nullptr, argmax, 1, logp, ldl, s);
}
int vag_head_logits_step_h(const float* h2, const float* cw, const float* e, const float* tables, const int64_t* tok, vag_head_w w,
                           int64_t N, int64_t E, int64_t H, int64_t V, float* logits, int64_t ldl, float* parts, float* scratch,
                           vag_stream_t stream) {
    hipStream_t s = S_(stream);
    VAG_CHECK_ARG(h2 && cw && (tables ? tok != nullptr : e != nullptr) && logits && parts && scratch && N > 0 && N <= 256 &&
                  E % 4 == 0 && H % 4 == 0 && V > 0 && ldl >= V);
    VAG_CHECK_ARG(w.w1 && w.b1 && w.b2 && w.w3 && w.b3 && w.out_w && w.out_b && aligned16(scratch) && aligned16(h2) &&
                  (tables || aligned16(e)) && aligned16(w.w1) && aligned16(w.w3));
    float* tmid = scratch + N * E;   // (N,E)
    VAG_TRY(head_pre_step_h(h2, cw, e, tables ? tables + ((V * 3 * H + 63) & ~63ll) : nullptr, tok, w, N, E, H, tmid, s));
    return vag_logits_parts_launch(N, V, E, tmid, E, w.out_w, E, w.out_b, logits, ldl, parts, s);
}
// The same step for beam search without the normalising pass: raw logits plus, per row, the pieces of its log-sum-exp written by
// the vocabulary product's epilogue (gemm.hip, TallArgs::parts); vag_beam_step_logits_dev normalises on the fly.  The count is 0
// for shapes the tall-skinny kernel does not take (N = B k <= 96 rows, V < 4096, E % 256 != 0): use vag_head_logp_step there.
int64_t vag_head_logits_parts_count(vag_head_w w, int64_t N, int64_t E, int64_t V) {
    if (!w.out_w || N <= 0 || E <= 0 || V <= 0) return 0;
    alignas(16) static const float dummy[4] = {0.f, 0.f, 0.f, 0.f};
    return vag_logits_parts_count(N, V, E, dummy, E, w.out_w, E);
}
int vag_head_logits_step(const float* h2, const float* c, const float* e, vag_head_w w, int64_t N, int64_t E, int64_t H,
                         int64_t V, float* logits, int64_t ldl, float* parts, float* scratch, vag_stream_t stream) {
    hipStream_t s = S_(stream);
    VAG_CHECK_ARG(h2 && c && e && logits && parts && scratch && N > 0 && E % 4 == 0 && H % 4 == 0 && V > 0 && ldl >= V);
    VAG_CHECK_ARG(w.w1 && w.b1 && w.w2 && w.b2 && w.w3 && w.b3 && w.out_w && w.out_b && aligned16(scratch));
    VAG_CHECK_ARG(N <= 256 && aligned16(h2) && aligned16(c) && aligned16(e) && aligned16(w.w1) && aligned16(w.w2) && aligned16(w.w3));
    float* tmid = scratch + N * E;   // (N,E)
    const float* A3[3] = {h2, c, e};
    const float* W3[3] = {w.w1, w.w2, w.w3};
    const float* B3[3] = {w.b1, w.b2, w.b3};
    const int64_t C = 2 * H, ld3[3] = {H, C, E}, K3[3] = {H, C, E};
    VAG_TRY(vag_skinny3_launch(N, E, A3, ld3, W3, ld3, K3, B3, tmid, E, VAG_ACT_TANH, nullptr, VAG_DROP_DEC_OUT, 0.f, 0, s));
    return vag_logits_parts_launch(N, V, E, tmid, E, w.out_w, E, w.out_b, logits, ldl, parts, s);
}

// =====================================================================================================
// shared-space projections
// =====================================================================================================
int vag_img_proj_l2_fwd(const float* x, const float* W, const float* b, int64_t B, int64_t K, int64_t S, int act, float* y,
                        float* nrm, float* out, vag_stream_t stream) {
    hipStream_t s = S_(stream);
    VAG_CHECK_ARG(x && W && b && y && nrm && out && B > 0 && K > 0 && S > 0);
    VAG_TRY(linear_fwd(B, S, K, x, K, W, b, act ? VAG_ACT_TANH : 0, y, S, s));
    return vag_l2norm_fwd_launch(y, B, S, nrm, out, s);
}
int vag_img_proj_l2_bwd(const float* x, const float* W, const float* y, const float* nrm, const float* out, float* d_out,
                        int64_t B, int64_t K, int64_t S, int act, float* d_x, float* g_W, float* g_b, vag_stream_t stream) {
    hipStream_t s = S_(stream);
    VAG_CHECK_ARG(x && W && y && nrm && out && d_out && B > 0 && K > 0 && S > 0);
    VAG_TRY(vag_l2norm_bwd_launch(y, nrm, out, d_out, B, S, act, d_out, s));
    if (d_x) VAG_TRY(gemm_nn(B, K, S, d_out, S, W, K, 0.f, d_x, K, s));
    if (g_W) VAG_TRY(gemm_tn_acc(S, K, B, d_out, S, x, K, g_W, K, s));
    if (g_b) VAG_TRY(vag_colsum_launch(d_out, B, S, S, g_b, s));
    return VAG_OK;
}

int vag_l2norm_fwd(const float* x, int64_t B, int64_t S, float* nrm, float* out, vag_stream_t stream) {
    return vag_l2norm_fwd_launch(x, B, S, nrm, out, S_(stream));
}
int vag_l2norm_bwd(const float* x, const float* nrm, const float* out, const float* d_out, int64_t B, int64_t S, float* dx,
                   vag_stream_t stream) {
    return vag_l2norm_bwd_launch(x, nrm, out, d_out, B, S, 0, dx, S_(stream));
}

// =====================================================================================================
// image-conditioned attention
// =====================================================================================================
struct ImgWs {
    float *u, *w, *scores, *dalpha, *de, *dw, *du, *pre, *dpre, *dvp;
    int64_t total;
};
static ImgWs imagine_ws(float* p, int64_t B, int64_t Ts, int64_t C, int method) {
    ImgWs w;
    int64_t o = 0;
    auto take = [&](int64_t n) { float* q = p ? p + o : nullptr; o += (n + 63) & ~63ll; return q; };
    w.u = take(B * C); w.w = take(B * C); w.scores = take(B * Ts); w.dalpha = take(B * Ts); w.de = take(B * Ts);
    w.dw = take(B * C); w.du = take(B * C); w.dvp = take(VAG_POST_CHUNKS(Ts) * B * C);
    w.pre = method == 1 ? take(B * Ts * C) : nullptr;
    w.dpre = method == 1 ? take(B * Ts * C) : nullptr;
    w.total = o;
    return w;
}
int64_t vag_imagine_ws_floats(int64_t B, int64_t Ts, int64_t C, int64_t S, int method) {
    (void)S;
    return imagine_ws(nullptr, B, Ts, C, method).total;
}

int vag_imagine_attn_ctx_fwd(const float* im_emb, const float* enc, const float* mask, const float* ctx2ctx,
                             const float* emb2ctx, const float* mlp_w, int method, int64_t B, int64_t Ts, int64_t C,
                             int64_t S, float* alpha, float* ctx, float* ws, vag_stream_t stream) {
    hipStream_t s = S_(stream);
    VAG_CHECK_ARG(im_emb && enc && mask && ctx2ctx && emb2ctx && alpha && ctx && ws && (method == 0 || (method == 1 && mlp_w)));
    VAG_CHECK_ARG(B > 0 && Ts > 0 && C % 4 == 0 && S % 4 == 0 && aligned16(ws));
    ImgWs w = imagine_ws(ws, B, Ts, C, method);
    VAG_TRY(linear_fwd(B, C, S, im_emb, S, emb2ctx, nullptr, 0, w.u, C, s));                     // emb2ctx(image_vec) :58/:76
    if (method == 0) {
        // e[b,t] = (W_cc enc[b,t]) . u[b] = enc[b,t] . (W_cc^T u[b])                                 :57-64
        VAG_TRY(gemm_nn(B, C, C, w.u, C, ctx2ctx, C, 0.f, w.w, C, s));
        if (Ts <= 4096 && vag_opt().attn_row != 0)          // scores, softmax :46 and bmm :137 in one launch
            return vag_attn_dot_row_launch(false, enc, w.w, C, mask, nullptr, B, Ts, C, alpha, ctx, s);
        VAG_TRY(vag_attn_scores_launch(1, enc, w.w, C, nullptr, mask, B, 1, Ts, C, w.scores, s));
    } else {
        VAG_TRY(linear_fwd(B * Ts, C, C, enc, C, ctx2ctx, nullptr, 0, w.pre, C, s));              // ctx2ctx(decoder_hidden) :75
        VAG_TRY(vag_attn_scores_launch(0, w.pre, w.u, C, mlp_w, mask, B, 1, Ts, C, w.scores, s));   // mlp(tanh(ctx_+im_)) :78
    }
    return vag_attn_ctx_launch(1, w.scores, enc, B, 1, Ts, C, alpha, ctx, s);                      // softmax :46, bmm :137
}

int vag_imagine_attn_ctx_bwd(const float* im_emb, const float* enc, const float* mask, const float* ctx2ctx,
                             const float* emb2ctx, const float* mlp_w, int method, int64_t B, int64_t Ts, int64_t C,
                             int64_t S, const float* alpha, const float* d_ctx, float* ws, float* d_enc, int accumulate_enc,
                             float* d_im_emb, float* g_ctx2ctx, float* g_emb2ctx, float* g_mlp_w, vag_stream_t stream) {
    return vag_imagine_attn_ctx_bwd_impl(im_emb, enc, mask, ctx2ctx, emb2ctx, mlp_w, method, B, Ts, C, S, alpha, d_ctx, ws,
                                         d_enc, accumulate_enc, d_im_emb, 0, g_ctx2ctx, g_emb2ctx, g_mlp_w, S_(stream));
}
}  // extern "C"
int vag_imagine_attn_ctx_bwd_impl(const float* im_emb, const float* enc, const float* mask, const float* ctx2ctx,
                                  const float* emb2ctx, const float* mlp_w, int method, int64_t B, int64_t Ts, int64_t C,
                                  int64_t S, const float* alpha, const float* d_ctx, float* ws, float* d_enc,
                                  int accumulate_enc, float* d_im_emb, int accumulate_im, float* g_ctx2ctx,
                                  float* g_emb2ctx, float* g_mlp_w, hipStream_t s) {
    VAG_CHECK_ARG(im_emb && enc && ctx2ctx && emb2ctx && alpha && d_ctx && ws && d_enc && d_im_emb && g_ctx2ctx && g_emb2ctx);
    VAG_CHECK_ARG((method == 0) || (method == 1 && mlp_w && g_mlp_w));
    VAG_CHECK_ARG(B > 0 && Ts > 0 && C % 4 == 0 && S % 4 == 0);
    (void)mask;
    ImgWs w = imagine_ws(ws, B, Ts, C, method);
    const bool row = Ts <= 4096 && vag_opt().attn_row != 0;
    if (row) {      // d alpha = d_ctx . enc, the softmax backward and (dot method) dw[b] = sum_t de enc in one launch
        VAG_TRY(vag_attn_dot_row_launch(true, enc, d_ctx, C, nullptr, alpha, B, Ts, C, w.de, method == 0 ? w.dw : nullptr, s));
    } else {
        VAG_TRY(vag_attn_scores_launch(1, enc, d_ctx, C, nullptr, nullptr, B, 1, Ts, C, w.dalpha, s));   // d alpha = d_ctx . enc
        VAG_TRY(vag_softmax_bwd_launch(alpha, w.dalpha, B, Ts, w.de, s));
    }
    if (method == 0) {
        VAG_TRY(vag_outer2_launch(alpha, d_ctx, w.de, w.w, B, Ts, C, d_enc, accumulate_enc, s));
        if (!row) VAG_TRY(vag_attn_ctx_launch(0, w.de, enc, B, 1, Ts, C, nullptr, w.dw, s));      // dw[b] = sum_t de enc
        VAG_TRY(linear_fwd(B, C, C, w.dw, C, ctx2ctx, nullptr, 0, w.du, C, s));                   // du = dw W_cc^T
        VAG_TRY(gemm_tn_acc(C, C, B, w.u, C, w.dw, C, g_ctx2ctx, C, s));                          // g_cc[i,j] += u[b,i] dw[b,j]
    } else {
        VAG_TRY(vag_outer2_launch(alpha, d_ctx, nullptr, nullptr, B, Ts, C, d_enc, accumulate_enc, s));
        VAG_TRY(vag_attn_post_bwd_launch(w.pre, w.u, C, mlp_w, w.de, alpha, nullptr, B, Ts, 1, C, w.dpre, w.dvp, nullptr, 0, s));
        VAG_TRY(vag_colsum_launch(w.dvp, VAG_POST_CHUNKS(Ts) * B, C, C, g_mlp_w, s));
        VAG_TRY(vag_attn_dq_launch(w.pre, w.u, C, mlp_w, nullptr, nullptr, w.de, B, Ts, C, w.du, C, s));
        VAG_TRY(gemm_nn(B * Ts, C, C, w.dpre, C, ctx2ctx, C, 1.f, d_enc, C, s));
        VAG_TRY(gemm_tn_acc(C, C, B * Ts, w.dpre, C, enc, C, g_ctx2ctx, C, s));
    }
    VAG_TRY(gemm_nn(B, S, C, w.du, C, emb2ctx, S, accumulate_im ? 1.f : 0.f, d_im_emb, S, s));
    VAG_TRY(gemm_tn_acc(C, S, B, w.du, C, im_emb, S, g_emb2ctx, S, s));
    return VAG_OK;
}
extern "C" {

// =====================================================================================================
// ranking loss
// =====================================================================================================
int vag_rank_loss_fwd(const float* im, const float* sv, int64_t B, int64_t S, float margin, int kind, float* scores,
                      float* G, float* loss, vag_stream_t stream) {
    return vag_rank_loss_fwd_impl(im, sv, B, S, margin, kind, scores, G, loss, nullptr, S_(stream));
}
int vag_rank_loss_bwd(const float* im, const float* sv, const float* G, const float* d_loss, int64_t B, int64_t S,
                      float* d_im, float* d_s, vag_stream_t stream) {
    VAG_CHECK_ARG(d_loss != nullptr);
    return vag_rank_loss_bwd_impl(im, sv, G, d_loss, B, S, d_im, d_s, S_(stream));
}
}  // extern "C"
int vag_rank_loss_fwd_impl(const float* im, const float* sv, int64_t B, int64_t S, float margin, int kind, float* scores,
                           float* G, float* loss, const float* g_scale, hipStream_t s) {
    VAG_CHECK_ARG(im && sv && scores && G && loss && B > 0 && S > 0);
    VAG_TRY(linear_fwd(B, B, S, im, S, sv, nullptr, 0, scores, B, s));                             // im s^T  (:12)
    return vag_rank_loss_launch(scores, B, margin, kind, G, loss, s, g_scale);
}
int vag_rank_loss_bwd_impl(const float* im, const float* sv, const float* G, const float* d_loss, int64_t B, int64_t S,
                           float* d_im, float* d_s, hipStream_t s) {
    VAG_CHECK_ARG(im && sv && G && d_im && d_s && B > 0 && S > 0);
    if (B <= 128 && B % 16 == 0 && S % 4 == 0 && aligned16(G) && aligned16(im) && aligned16(sv)) return vag_rank_bwd_launch(G, im, sv, d_loss, B, S, d_im, d_s, s);               // both products (and the scale) in one launch
    VAG_TRY(gemm_nn(B, S, B, G, B, sv, S, 0.f, d_im, S, s));                                       // d_im = G s
    VAG_TRY(vag_gemm_launch(B, S, B, 1.f, G, 1, B, im, S, 1, 0.f, d_s, S, nullptr, 0, s));        // d_s  = G^T im
    if (!d_loss) return VAG_OK;                                                                    // (G came pre-multiplied)
    VAG_TRY(vag_scale_by_dev_launch(d_im, B * S, d_loss, s));
    return vag_scale_by_dev_launch(d_s, B * S, d_loss, s);
}
extern "C" {

// batch assembly from a device-resident corpus (preprocessing.py:308-384)
int vag_gather_rows_i64(const int64_t* in, int64_t ld, const int64_t* idx, int64_t rows, int64_t w, int64_t* out,
                        vag_stream_t stream) {
    return vag_gather_rows_i64_launch(in, ld, idx, rows, w, out, S_(stream));
}

// =====================================================================================================
// retrieval evaluation: utils/im_retrieval_eval.py:4-57
// =====================================================================================================
int vag_retrieval_ranks(const float* queries, const float* keys, int64_t N, int64_t S, float* scores, int32_t* ranks,
                        vag_stream_t stream) {
    hipStream_t s = S_(stream);
    VAG_CHECK_ARG(queries && keys && scores && ranks && N > 0 && S > 0);
    VAG_TRY(vag_gemm_launch(N, N, S, 1.f, queries, S, 1, keys, 1, S, 0.f, scores, N, nullptr, 0, s));   // one (N,S)x(S,N) product
    return vag_retrieval_rank_launch(scores, N, ranks, s);
}

// =====================================================================================================
// decoder initial state
// =====================================================================================================
int vag_dec_init_fwd(const float* enc, const float* mask, const float* ctx, float split, const float* W, const float* b,
                     int64_t B, int64_t Ts, int64_t C, int64_t H, float* xmix, float* h0, vag_stream_t stream) {
    hipStream_t s = S_(stream);
    VAG_CHECK_ARG(enc && mask && W && b && xmix && h0 && B > 0 && Ts > 0 && C % 4 == 0 && H > 0);
    // (a step driver's visual-attention launch may already have left xmix: vag_attn_row_mix_request)
    if (!(ctx && vag_attn_row_mix_done(xmix))) VAG_TRY(vag_meanpool_mix_launch(enc, mask, ctx, split, B, Ts, C, xmix, s));
    return linear_fwd(B, H, C, xmix, C, W, b, VAG_ACT_TANH, h0, H, s);
}
int vag_dec_init_bwd(const float* mask, const float* xmix, const float* h0, float split, const float* W, float* d_h0,
                     int64_t B, int64_t Ts, int64_t C, int64_t H, float* d_enc, int accumulate_enc, float* d_ctx, float* g_W,
                     float* g_b, float* scratch, vag_stream_t stream) {
    return vag_dec_init_bwd_impl(mask, xmix, h0, split, W, d_h0, B, Ts, C, H, d_enc, accumulate_enc, d_ctx, 0, g_W, g_b,
                                 scratch, S_(stream));
}
}  // extern "C"
int vag_dec_init_bwd_impl(const float* mask, const float* xmix, const float* h0, float split, const float* W, float* d_h0,
                          int64_t B, int64_t Ts, int64_t C, int64_t H, float* d_enc, int accumulate_enc, float* d_ctx,
                          int accumulate_ctx, float* g_W, float* g_b, float* scratch, hipStream_t s) {
    VAG_CHECK_ARG(mask && xmix && h0 && W && d_h0 && d_enc && g_W && g_b && scratch && B > 0 && Ts > 0 && C % 4 == 0);
    float* dx = scratch;    // (B,C)
    // (a step driver's persistent decoder backward may already have applied the tanh's derivative: vag_persist_dh0_tanh_request)
    if (!vag_persist_dh0_tanh_done(d_h0)) VAG_TRY(vag_tanh_bwd_launch(h0, d_h0, d_h0, B * H, nullptr, 0, 0.f, s));
    VAG_TRY(gemm_tn_acc(H, C, B, d_h0, H, xmix, C, g_W, C, s));
    VAG_TRY(vag_colsum_launch(d_h0, B, H, H, g_b, s));
    const bool ride = d_ctx && B <= 128;                       // d_ctx (+)= split * dx leaves with the product that forms dx
    if (ride) vag_skinny_nn_out2(d_ctx, C, split, accumulate_ctx ? 1 : 0);
    VAG_TRY(gemm_nn(B, C, H, d_h0, H, W, C, 0.f, dx, C, s));
    const float s_eff = d_ctx ? split : 0.f;
    VAG_TRY(vag_meanpool_bwd_launch(mask, dx, 1.f - s_eff, B, Ts, C, d_enc, accumulate_enc, s));
    if (d_ctx && !ride) VAG_TRY(vag_axpy_launch(split, dx, d_ctx, B * C, accumulate_ctx ? 1 : 0, s));
    return VAG_OK;
}
extern "C" {

// =====================================================================================================
// beam search, optimiser, dropout helpers
// =====================================================================================================
int64_t vag_beam_scratch_bytes(int64_t B, int64_t k, int64_t V, int64_t max_len) {
    (void)max_len;
    return vag_beam_scratch_bytes_impl(B, k, V);
}
int vag_beam_step(float* logp, int64_t ldl, float* nll, int64_t* beam, int64_t di, int64_t max_len, const float* h_in,
                  float* h_out, int64_t B, int64_t k, int64_t V, int64_t H, int32_t* n_alive, void* scratch,
                  vag_stream_t stream) {
    return vag_beam_step_launch(logp, ldl, nll, beam, di, nullptr, max_len, h_in, h_out, nullptr, B, k, V, H, n_alive, scratch,
                                S_(stream));
}
int vag_beam_step_dev(float* logp, int64_t ldl, float* nll, int64_t* beam, int32_t* di_state, int64_t max_len,
                      const float* h_in, float* h_out, int64_t* tok_out, int64_t B, int64_t k, int64_t V, int64_t H,
                      int32_t* n_alive, void* scratch, vag_stream_t stream) {
    VAG_CHECK_ARG(di_state != nullptr);
    return vag_beam_step_launch(logp, ldl, nll, beam, 0, di_state, max_len, h_in, h_out, tok_out, B, k, V, H, n_alive, scratch,
                                S_(stream));
}
int vag_beam_step_logits_dev(float* logits, int64_t ldl, const float* parts, int64_t nparts, float* nll, int64_t* beam,
                             int32_t* di_state, int64_t max_len, const float* h_in, float* h_out, int64_t* tok_out, int64_t B,
                             int64_t k, int64_t V, int64_t H, int32_t* n_alive, void* scratch, vag_stream_t stream) {
    VAG_CHECK_ARG(di_state != nullptr && parts != nullptr && nparts > 0);
    return vag_beam_step_launch(logits, ldl, nll, beam, 0, di_state, max_len, h_in, h_out, tok_out, B, k, V, H, n_alive, scratch,
                                S_(stream), parts, nparts);
}
int vag_beam_finish(const float* nll, const int64_t* beam, int64_t max_len, int64_t steps, int64_t B, int64_t k, int64_t* out,
                    float* best_score, vag_stream_t stream) {
    return vag_beam_finish_launch(nll, beam, max_len, steps, B, k, out, best_score, S_(stream));
}

int vag_clip_adam_flat(float* p, float* g, float* m, float* v, int64_t n, int nseg, const int64_t* seg_off,
                       const float* seg_lr, const float* seg_wd, float clip, float grad_scale, float beta1, float beta2,
                       float eps, int zero_grad, int32_t* step, float* norm_out, void* scratch, const float* lr_dev,
                       vag_stream_t stream) {
    return vag_clip_adam_launch(p, g, m, v, n, nseg, seg_off, seg_lr, seg_wd, clip, grad_scale, beta1, beta2, eps, zero_grad,
                                step, norm_out, scratch, lr_dev, S_(stream));
}

int vag_clip_adam_shard(float* p, float* g, float* m, float* v, int64_t n, int nseg, const int64_t* seg_off, const float* seg_lr,
                        const float* seg_wd, float clip, float grad_scale, float beta1, float beta2, float eps, int zero_grad,
                        int32_t* step, float* norm_out, void* scratch, const float* lr_dev, int64_t lo, int64_t hi, int phase,
                        double* sumsq, vag_stream_t stream) {
    return vag_clip_adam_shard_launch(p, g, m, v, n, nseg, seg_off, seg_lr, seg_wd, clip, grad_scale, beta1, beta2, eps, zero_grad,
                                      step, norm_out, scratch, lr_dev, lo, hi, phase, sumsq, S_(stream));
}

int vag_copy4(const void* const* src, void* const* dst, const int64_t* bytes, int n, vag_stream_t stream) {
    return vag_copy4_launch(src, dst, bytes, n, S_(stream));
}

int vag_recurrence_supported(int kind, int64_t B, int64_t Ts, int64_t Tt, int64_t H) {
    if (kind == 0) return vag_enc_persistent_ok(B, Ts, H) ? 1 : 0;
    if (kind == 1) return vag_dec_persistent_ok(B, Ts, Tt, H) ? 1 : 0;
    return 0;
}
int vag_persistent_timeouts(void) { return vag_persistent_timeouts_read(); }
int vag_set_operator_guard(void* guard) {
    VAG_CHECK_ARG((reinterpret_cast<uintptr_t>(guard) & 3) == 0);
    vag_persist_guard_set(reinterpret_cast<unsigned*>(guard));
    return VAG_OK;
}
int vag_gemm_group_plan(int n, const int64_t* M, const int64_t* N, const int64_t* K, const int* accumulate, int* split,
                        int* order) {
    return vag_gemm_group_plan_host(n, M, N, K, accumulate, split, order);
}
int vag_recurrence_time(int kind, double* ms_total, int* launches) { return vag_persistent_time_read(kind, ms_total, launches); }
int64_t vag_recurrence_sync_words(int kind, int64_t B, int64_t T) {
    return kind == 0 ? vag_enc_persistent_sync_words(B, T) : vag_dec_persistent_sync_words(B, T);
}
int vag_cgru_recurrence_fwd(const float* pe, const float* mask, const float* h0, const float* xp1, vag_dec_w w, const float* wcat,
                            const float* bcat, const float* encwp, int64_t B, int64_t Ts, int64_t Tt, int64_t H, float* h1,
                            float* g1, float* qhp, float* alpha, float* h2_all, float* g2, float* psc, void* sync,
                            vag_stream_t stream) {
    VAG_CHECK_ARG(dec_w_ok(w) && vag_dec_persistent_ok(B, Ts, Tt, H));
    return vag_dec_fwd_persistent_launch(pe, mask, h0, xp1, w.gru1.w_hh, w.gru1.b_hh, wcat, bcat, w.attn_v, encwp, w.gru2.b_ih, h1,
                                         g1, qhp, alpha, h2_all, g2, psc, reinterpret_cast<unsigned*>(sync), B, Ts, Tt, H,
                                         S_(stream));
}

int vag_dropout_mask(const uint64_t* rng, int which, int64_t n, float p, float* out, vag_stream_t stream) {
    VAG_CHECK_ARG(which >= 1 && which <= 3);
    return vag_dropout_mask_launch(rng, which, n, p, out, S_(stream));
}
int vag_rng_advance(uint64_t* rng, vag_stream_t stream) { return vag_rng_advance_launch(rng, S_(stream)); }

}  // extern "C"
