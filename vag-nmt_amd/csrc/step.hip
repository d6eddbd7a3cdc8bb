// Fused training step (train.py:36-51 around models/...V11.py:82-168 / NMT_Seq2Seq_Beam_V2.py:58-113): zero-grad is the
// optimiser's job (vag_clip_adam_flat leaves the gradient buffer zeroed), so one call here enqueues the whole forward and
// backward of one mini-batch on the caller's stream -- every operator of include/vag_nmt.h in the order autograd would run
// them, on ONE caller-owned workspace, with the gradient w.r.t. the encoder states accumulated in place by its four
// consumers (decoder, attention keys, decoder initial state, image-conditioned attention) instead of summed by the host
// framework.  No allocation, no host synchronisation: a step is capturable into a HIP graph, and the `phases` mask lets a
// data-parallel driver cut it where a gradient bucket becomes final (after the decoder side: everything but the encoder's
// parameters; after the encoder's backward recurrence: the rest).
#include "../../include/vag_nmt.h"
#include <vector>
#include "kernels.h"

#define S_(x) reinterpret_cast<hipStream_t>(x)

namespace {

struct StepWs {
    float *enc, *mask, *ws_enc, *pe;
    float *y_im, *nrm_im, *im_emb, *ws_img, *alpha_v, *ctx, *y_txt, *nrm_txt, *txt_emb, *rscores, *G;
    float *xmix, *hseq, *c_all, *e_all, *ws_dec, *tmid, *logits, *lse, *nll, *inv_cnt, *consts;
    int64_t* tok;
    float *d_enc, *d_pe, *d_h2, *d_c, *d_e, *d_h0, *d_ctx, *d_im, *d_txt, *scr_dec, *scr_head, *scr_ini;
    float* free_tab;            // tables of the one-launch free-running decoder (shapes it can take only)
    float* gemm_slab;           // scratch of the slab form of split-K (gemm.hip: GemmArgs::slab), fp32 storage only
    unsigned* gemm_ticket;      // ... and its per-tile arrival counters (zeroed by the step's prologue launch)
    int64_t gemm_slab_floats, gemm_tickets;
    int64_t total;
};

StepWs step_ws(float* p, const vag_step_cfg& c) {
    StepWs w;
    const int64_t B = c.B, Ts = c.Ts, Tt = c.Tt, H = c.H, C = 2 * c.H, S = c.S, R = c.Tt * c.B;
    int64_t o = 0;
    auto take = [&](int64_t n) { float* q = p ? p + o : nullptr; o += (n + 63) & ~63ll; return q; };
    w.enc = take(B * Ts * C); w.mask = take(B * Ts); w.ws_enc = take(vag_bigru_ws_floats(B, Ts, c.Es, H));
    w.pe = take(B * Ts * C);
    const bool mm = c.multimodal != 0;
    w.y_im = take(mm ? B * S : 0); w.nrm_im = take(mm ? B : 0); w.im_emb = take(mm ? B * S : 0);
    w.ws_img = take(mm ? vag_imagine_ws_floats(B, Ts, C, S, c.attn_method) : 0);
    w.alpha_v = take(mm ? B * Ts : 0); w.ctx = take(mm ? B * C : 0);
    w.y_txt = take(mm ? B * S : 0); w.nrm_txt = take(mm ? B : 0); w.txt_emb = take(mm ? B * S : 0);
    w.rscores = take(mm ? B * B : 0); w.G = take(mm ? B * B : 0);
    w.xmix = take(B * C); w.hseq = take((Tt + 1) * B * H); w.c_all = take(R * C); w.e_all = take(R * c.Et);
    w.ws_dec = take(vag_cgru_ws_floats(B, Ts, Tt, c.Et, H));
    w.tmid = take(R * c.Et); w.logits = take(R * c.ldl); w.lse = take(R); w.nll = take(R); w.inv_cnt = take(B);
    w.consts = take(8);
    w.tok = reinterpret_cast<int64_t*>(take(2 * (Tt + 1) * B));
    w.d_enc = take(B * Ts * C); w.d_pe = take(B * Ts * C); w.d_h2 = take(R * H); w.d_c = take(R * C); w.d_e = take(R * c.Et);
    w.d_h0 = take(B * H); w.d_ctx = take(mm ? B * C : 0); w.d_im = take(mm ? B * S : 0); w.d_txt = take(mm ? B * S : 0);
    w.scr_dec = take(vag_cgru_bwd_scratch_floats(B, Ts, Tt, c.Et, H)); w.scr_head = take(R * c.Et); w.scr_ini = take(B * C);
    // (by shape, not by device: a workspace size must not depend on where it is asked for)
    w.free_tab = take(H == 512 && c.Et == 256 && B <= 64 ? vag_dec_free_tables_floats(B, Ts, Tt, c.Et, H, c.V) : 0);
    // split-K slabs: the largest flush of a step parks (k-slices x its output elements) floats -- at configs[1] 17 M (the d_enc /
    // d_e group), 14 M (the decoder's weight gradients), 7.8 M (d tmid in 12 slices): twelve times the widest activation covers them;
    // a product that does not fit keeps its atomics (gemm_take_slabs).  The 2-byte mode's one-plane kernels do not use slabs.
    const int64_t widest = std::max(std::max(B * Ts * C, R * 3 * H), std::max(3 * H * C, R * c.Et));
    w.gemm_slab_floats = c.storage == 0 ? std::min<int64_t>(12 * widest, 192ll << 20) : 0;      // (at most 768 MB: configs[4] sizes in fp32)
    w.gemm_tickets = c.storage == 0 ? 16384 : 0;
    w.gemm_slab = take(w.gemm_slab_floats);
    w.gemm_ticket = reinterpret_cast<unsigned*>(take(w.gemm_tickets));
    w.total = o;
    return w;
}

bool cfg_ok(const vag_step_cfg* c) {
    return c && c->B > 0 && c->Ts > 0 && c->Tt > 0 && c->Es > 0 && c->Et > 0 && c->H > 0 && c->V > 0 && c->Es % 4 == 0 &&
           c->Et % 4 == 0 && c->H % 4 == 0 && c->ldl >= c->V && c->ldl % 4 == 0 && c->loss_ring >= 0 && (reinterpret_cast<uintptr_t>(c->guard) & 3) == 0 &&
           (!c->multimodal || (c->S > 0 && c->S % 4 == 0 && c->I > 0 && (c->attn_method == 0 || c->attn_method == 1) &&
                               c->rank_kind >= -1 && c->rank_kind <= 1));
}

// rng step counter, the two loss-mix constants, the decoder's input token matrix (row 0 = SOS, row t+1 = target word t:
// V11.py:117,146), the per-sentence token counts -- one launch for the step's scalar bookkeeping.
// Also zeroes the counters and exchange buffers of the step's four recurrence kernels (two word ranges, api.hip:
// vag_step_zero_ranges): one launch instead of four.
// further ranges the prologue zeroes, in 16-byte units: the decoder's hidden states h2 (exchanged between workgroups with marked
// words, persist.hip: tag1), the head's tmid and the encoder's d(embedded inputs) (grouped products accumulate into them)
// ... and two outputs of sliced overwriting products of the backward pass (d(tmid), dgi2 W_ih2), which then skip their fill launches
struct ZeroRanges { uint4* p[6]; int64_t n[6]; };
// the decoder's embedded input tokens of every step (teacher-forced: V11.py:117,146 -> NMT_Decoder.py:118), from the target matrix itself
struct GatherTask { const float* emb; float* out; int E4; };
__global__ __launch_bounds__(256) void step_prologue_kernel(uint64_t* rng, const int64_t* __restrict__ tgt, int B, int Tt,
                                                            int64_t* __restrict__ tok, float* __restrict__ consts,
                                                            float* __restrict__ inv_cnt, float w_mt, float w_vse,
                                                            unsigned* __restrict__ z0, int64_t n0, unsigned* __restrict__ z1,
                                                            int64_t n1, ZeroRanges zr, GatherTask ga) {
    const int64_t gid = blockIdx.x * 256ll + threadIdx.x;
    for (int64_t i = gid; i < n0; i += (int64_t)gridDim.x * 256) z0[i] = 0u;
    if (ga.out) {
        const int64_t rows = (int64_t)Tt * B, tot = rows * ga.E4;
        for (int64_t i = gid; i < tot; i += (int64_t)gridDim.x * 256) {
            const int64_t row = i / ga.E4;
            const int e4 = (int)(i - row * ga.E4);
            const int t = (int)(row / B), b = (int)(row - (int64_t)t * B);
            const int64_t tk = t == 0 ? 2 : tgt[(int64_t)b * Tt + (t - 1)];
            reinterpret_cast<float4*>(ga.out)[i] = reinterpret_cast<const float4*>(ga.emb)[tk * ga.E4 + e4];
        }
    }
#pragma unroll
    for (int r = 0; r < 6; ++r)                                                                           // (16-byte units)
        for (int64_t i = gid; i < zr.n[r]; i += (int64_t)gridDim.x * 256) zr.p[r][i] = make_uint4(0u, 0u, 0u, 0u);
    {   // the large range: 16 bytes per thread (the range starts 256-byte aligned; its tail word by word)
        const int64_t n4 = n1 >> 2;
        uint4* z4 = reinterpret_cast<uint4*>(z1);
        for (int64_t i = gid; i < n4; i += (int64_t)gridDim.x * 256) z4[i] = make_uint4(0u, 0u, 0u, 0u);
        for (int64_t i = (n4 << 2) + gid; i < n1; i += (int64_t)gridDim.x * 256) z1[i] = 0u;
    }
    if (gid < B) {                          // inv_cnt[b] = 1 / #(tgt[b,:] != 0)   (V11.py:164)
        int c = 0;
        for (int t = 0; t < Tt; ++t) c += tgt[gid * Tt + t] != 0;
        inv_cnt[gid] = 1.f / (float)c;
    }
    if (gid == 0) {
        if (rng) rng[1] += 1;
        consts[0] = w_mt; consts[1] = w_vse; consts[2] = 0.f; consts[3] = 0.f;
    }
    const int64_t total = (int64_t)(Tt + 1) * B;
    for (int64_t i = gid; i < total; i += (int64_t)gridDim.x * 256) {
        const int t = (int)(i / B), b = (int)(i - (int64_t)t * B);
        tok[i] = t == 0 ? 2 : tgt[(int64_t)b * Tt + (t - 1)];
    }
}

struct LossRingScope {
    explicit LossRingScope(int r) { vag_set_loss_ring(r); }
    ~LossRingScope() { vag_set_loss_ring(0); }
};
struct PrezeroScope {       // the recurrence kernels of this call find their counters zeroed by the step's prologue launch,
    PrezeroScope() { vag_persist_set_prezeroed(true); vag_step_set_zeroed(true); }       // the head and the encoder's backward
    ~PrezeroScope() {                                                                    // their accumulation buffers
        vag_persist_set_prezeroed(false); vag_step_set_zeroed(false); vag_step_set_gathered(false);
        vag_gemm_prezeroed_set(0, nullptr); vag_gemm_prezeroed_set(1, nullptr);
        (void)vag_loss_defer_flush();       // (an error return between the head's forward and backward: the loss is still written)
        // requests of this call that nobody consumed (an error return in between) must not outlive it
        (void)vag_attn_row_mix_done(nullptr); (void)vag_persist_dh0_tanh_done(nullptr); vag_attn_row_mix_cancel();
    }
};
struct DerivedScope {       // points the operators at the driver's derived weights, storage mode and head chunk for one call
    const float* prev_d;    // ... and puts back what the caller had set with vag_set_operator_context ("until changed")
    bool prev16;
    DerivedScope(const float* d, bool store16, int64_t chunk) : prev_d(vag_get_derived_override()), prev16(vag_get_store16()) {
        vag_set_derived_override(d); vag_set_store16(store16); vag_set_head_chunk(chunk);
    }
    ~DerivedScope() {
        vag_set_derived_override(prev_d); vag_set_store16(prev16); vag_set_head_chunk(0); vag_set_head_fuse(nullptr, nullptr, nullptr);
    }
};

}  // namespace

extern "C" {

int64_t vag_step_ws_floats(const vag_step_cfg* cfg) {
    if (!cfg_ok(cfg)) return VAG_EINVAL;
    return step_ws(nullptr, *cfg).total;
}

// float offset of a workspace tensor, for tests and diagnostics: 0 enc (B,Ts,C), 1 alpha_vse (B,Ts), 2 hseq (Tt+1,B,H),
// 3 logits (Tt*B,ldl), 4 decoder workspace (see vag_cgru_ws_offset), 5 d_enc (B,Ts,C), 6 im_emb, 7 txt_emb, 8 tok (int64)
int64_t vag_step_ws_offset(const vag_step_cfg* cfg, int which) {
    if (!cfg_ok(cfg)) return VAG_EINVAL;
    static float base[1];
    StepWs w = step_ws(base, *cfg);
    const float* p = nullptr;
    switch (which) {
        case 0: p = w.enc; break;
        case 1: p = w.alpha_v; break;
        case 2: p = w.hseq; break;
        case 3: p = w.logits; break;
        case 4: p = w.ws_dec; break;
        case 5: p = w.d_enc; break;
        case 6: p = w.im_emb; break;
        case 7: p = w.txt_emb; break;
        case 8: p = reinterpret_cast<const float*>(w.tok); break;
        default: return VAG_EINVAL;
    }
    return (int64_t)(p - base);
}

}  // extern "C"

// Side branches of a step (round 5).  What a step launches is one chain on the caller's stream, except for work that nothing on
// the chain waits for soon: the image projection of the forward pass (needs only the batch: it runs while the encoder does) and
// the held-back weight-gradient leaves of the backward pass (needed by the optimiser only: they run beside the encoder's backward
// recurrence).  Such work goes to a library-owned side stream between fork() and join(): an event recorded on the caller's
// stream, waited for by the side stream, and back.  Under stream capture both become graph edges (the side stream joins the
// capture through the event), eagerly they are two cheap runtime calls each.  One side stream and four events per host thread
// and device, created on first use, never destroyed (a handful per process).
namespace {
struct ForkState { hipStream_t side = nullptr; hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; int dev = -1; };
thread_local ForkState g_fork[8];
ForkState* fork_state() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    ForkState* f = nullptr;
    for (auto& x : g_fork) if (x.dev == dev) { f = &x; break; }
    if (!f) for (auto& x : g_fork) if (x.dev < 0) { f = &x; break; }
    if (!f) return nullptr;
    if (!f->side) {
        // lowest priority: what runs here is throughput work nothing waits for soon; the chain on the caller's stream goes first
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); lo = 0; }
        if (hipStreamCreateWithPriority(&f->side, hipStreamNonBlocking, lo) != hipSuccess) { (void)hipGetLastError(); f->side = nullptr; return nullptr; }
        for (auto& e : f->ev)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        f->dev = dev;
    }
    return f;
}
// a branch: fork(main, slot) returns the side stream (or `main` itself when forks are off / unavailable: the work then simply
// stays on the chain); join(main, slot) makes `main` wait for everything the side stream was given since
struct StepBranch {
    ForkState* f = nullptr;
    bool open = false;
    hipStream_t fork(hipStream_t main, int slot, int bit) {
        if ((vag_opt().step_fork & bit) == 0) return main;
        f = fork_state();
        if (!f) return main;
        if (hipEventRecord(f->ev[slot], main) != hipSuccess || hipStreamWaitEvent(f->side, f->ev[slot], 0) != hipSuccess) {
            (void)hipGetLastError();
            return main;
        }
        open = true;
        return f->side;
    }
    int join(hipStream_t main, int slot) {
        if (!open) return VAG_OK;
        open = false;
        if (hipEventRecord(f->ev[slot], f->side) != hipSuccess || hipStreamWaitEvent(main, f->ev[slot], 0) != hipSuccess)
            return (int)hipGetLastError();
        return VAG_OK;
    }
};
}  // namespace

extern "C" {

int vag_train_step(const vag_step_cfg* cfg, const vag_model_w* wp, const vag_model_g* gp, const int64_t* src,
                   const int32_t* lengths, const int64_t* tgt, const float* im, const float* vocab_weight, uint64_t* rng,
                   const float* derived, float* ws, float* losses, int phases, vag_stream_t stream) {
    hipStream_t s = S_(stream);
    VAG_CHECK_ARG(cfg_ok(cfg) && wp && gp && src && lengths && tgt && vocab_weight && ws && losses && aligned16(ws));
    const vag_step_cfg& c = *cfg;
    const vag_model_w& w = *wp;
    const vag_model_g& g = *gp;
    const bool mm = c.multimodal != 0;
    VAG_CHECK_ARG(!mm || (im && w.im_w && w.im_b && w.txt_w && w.txt_b && w.ctx2ctx && w.emb2ctx &&
                          (c.attn_method == 0 || w.mlp_w)));
    VAG_CHECK_ARG(w.enc_emb && w.ini_w && w.ini_b && w.attn_e && (phases & 55) != 0 && (phases & ~55) == 0);
    const int64_t B = c.B, Ts = c.Ts, Tt = c.Tt, H = c.H, C = 2 * H, S = c.S, V = c.V, Et = c.Et;
    StepWs k = step_ws(ws, c);
    VAG_CHECK_ARG(c.storage == 0 || (c.storage == 1 && derived && !c.free_run && H % 8 == 0));
    // output head in row chunks once the logits would exceed 1 GiB (configs[4]: 3.3 GB): the (Tt*B, V) logits are then never
    // formed -- one chunk buffer (<= 640 MiB) is reused.  Measured at configs[4] (fp16 storage, ms per step): whole 46.80;
    // chunks finished in the forward 4096 rows 46.84, 2048 47.7, 1024 48.2, 512 (96 MB, Infinity-Cache sized) 55.7 -- the
    // head is bound by its three products, not by the logits' traffic, and short chunks make those inefficient
    // (d(tmid) = 512 x 512 x K=40000 is 16 tiles); so chunks are as long as the buffer bound allows.
    int64_t chunk = 0;
    if (!c.free_run && (double)c.Tt * (double)c.B * (double)c.ldl * 4.0 > 1073741824.0) {
        chunk = (int64_t)(671088640.0 / ((double)c.ldl * 4.0)) / c.B * c.B;
        if (chunk < c.B) chunk = c.B;
    }
    if (vag_opt().head_chunk >= 0)      // vag_set_option("head_chunk", rows): rows per chunk (0 = never chunk); tests
        chunk = c.free_run ? 0 : vag_opt().head_chunk;
    DerivedScope scope(derived, c.storage == 1, chunk);
    LossRingScope lring(c.loss_ring);
    PrezeroScope prezero;           // (a backward-only call relies on the forward call of the same step having run first)
    // the grouped / single bf16x6 products of this call park their split-K slices in the workspace's slab region instead of adding
    // them into their outputs with one atomic per element and slice (gemm.hip: GemmArgs::slab)
    struct SlabScope {
        SlabScope(float* p, int64_t n, unsigned* t, int64_t nt) { vag_gemm_set_scratch(p, n, t, nt); }
        ~SlabScope() { vag_gemm_set_scratch(nullptr, 0, nullptr, 0); }
    } slab_scope(k.gemm_slab, k.gemm_slab_floats, k.gemm_ticket, k.gemm_tickets);
    // this call's persistent recurrence launches report a give-up to the caller's guard pair (vag_step_cfg.guard), not process-wide
    struct GuardScope {
        unsigned* prev; bool on;
        explicit GuardScope(void* g) : prev(nullptr), on(g != nullptr) {
            if (on) { prev = vag_persist_guard_peek(); vag_persist_guard_set(reinterpret_cast<unsigned*>(g)); }
        }
        ~GuardScope() { if (on) vag_persist_guard_set(prev); }
    } guard_scope(c.guard);
    // forward and backward in one call: the chunked head finishes each chunk (d(logits) and its products) in the forward;
    // a backward called on its own (phases = 2 after an earlier phases = 1) recomputes the chunks instead
    if (chunk > 0 && (phases & 1) && (phases & (2 | 16)) && vag_opt().head_fuse != 0)
        vag_set_head_fuse(&g.head, k.consts + 0, k.scr_head);
    const bool has_vse = mm && c.rank_kind >= 0;
    const float w_mt = mm ? c.loss_w : 1.f, w_vse = mm ? 1.f - c.loss_w : 0.f;
    float* h0 = k.hseq;                    // [h0, h2_0 .. h2_{Tt-1}] in one buffer: the W_hh1 gradient is one product
    float* h2_all = k.hseq + B * H;
    const uint64_t* crng = rng;
    float* d_e = vag_cgru_bwd_scratch_de(k.scr_dec, B, Ts, Tt, Et, H);       // d(embedded target inputs): head's share + gru_1's, in place
    // 2-byte storage mode: what feeds the large products carries no more than fp16 there (fp16-stored keys and weights,
    // activations the next fp16 product rounds anyway), so they run on one plane: fp16 operands forward (11 significand bits),
    // bf16 operands for every gradient product (fp16 would flush small gradients; their rounding errors average over the long sums)
    const bool one_plane = c.storage == 1 && !c.free_run && vag_opt().s16_one_plane != 0;
    StepBranch br_im, br_leaf, br_dw;
    if (phases & 1) {
        if (one_plane) vag_gemm_set_planes(11);
        if (mm) {
            // V11.py:114, VSE_Imagine_Enc.py:123-132: the image projection needs the batch only -- beside the encoder
            hipStream_t si = br_im.fork(s, 0, 1);
            VAG_TRY(vag_img_proj_l2_fwd(im, w.im_w, w.im_b, B, c.I, S, c.activation_vse, k.y_im, k.nrm_im, k.im_emb, si));
        }
        {
            unsigned* zp[3];
            int64_t zn[3];
            vag_step_zero_ranges(k.ws_enc, k.ws_dec, B, Ts, Tt, c.Es, Et, H, zp, zn);
            int64_t nb = cdiv64((Tt + 1) * B, 256);
            if (nb < cdiv64(zn[1], 1024)) nb = cdiv64(zn[1], 1024);
            if (nb > 1024) nb = 1024;
            ZeroRanges zr;
            zr.p[0] = reinterpret_cast<uint4*>(h2_all); zr.n[0] = Tt * B * H / 4;
            zr.p[1] = reinterpret_cast<uint4*>(k.tmid); zr.n[1] = Tt * B * Et / 4;
            zr.p[2] = reinterpret_cast<uint4*>(zp[2]); zr.n[2] = zn[2] / 4;
            zr.p[3] = reinterpret_cast<uint4*>(k.scr_head); zr.n[3] = chunk > 0 ? 0 : Tt * B * Et / 4;
            zr.p[4] = reinterpret_cast<uint4*>(vag_cgru_bwd_scratch_du(k.scr_dec, B, Ts, Tt, Et, H)); zr.n[4] = Tt * B * H / 4;
            zr.p[5] = reinterpret_cast<uint4*>(k.gemm_ticket); zr.n[5] = k.gemm_tickets / 4;      // split-K tickets (gemm.hip; the last block of a
                                                                                                   // tile resets its own: this is the belt to those braces)
            GatherTask ga = {nullptr, nullptr, 0};
            if (!c.free_run) { ga.emb = w.dec.emb; ga.out = k.e_all; ga.E4 = (int)(Et / 4); }
            hipLaunchKernelGGL(step_prologue_kernel, dim3((unsigned)nb), dim3(256), 0, s, rng, tgt, (int)B, (int)Tt, k.tok,
                               k.consts, k.inv_cnt, w_mt, w_vse, zp[0], zn[0], zp[1], zn[1], zr, ga);
            VAG_LAUNCH_CHECK();
            if (ga.out) vag_step_set_gathered(true);
        }
        VAG_TRY(vag_bigru_seq_fwd(src, lengths, w.enc_emb, w.enc_fw, w.enc_bw, c.p_emb, c.p_ctx, crng, B, Ts, c.Es, H, k.enc,
                                  k.mask, k.ws_enc, stream));                                           // V11.py:111
        if (mm) {                                                                                       // V11.py:114
            VAG_TRY(br_im.join(s, 1));
            vag_attn_row_mix_request(k.xmix, c.init_split);     // (consumed by the dot method's one-launch attention, else dropped)
            const int rc_att = vag_imagine_attn_ctx_fwd(k.im_emb, k.enc, k.mask, w.ctx2ctx, w.emb2ctx, w.mlp_w, c.attn_method, B, Ts, C, S,
                                                        k.alpha_v, k.ctx, k.ws_img, stream);
            vag_attn_row_mix_cancel();
            VAG_TRY(rc_att);
            VAG_TRY(vag_img_proj_l2_fwd(k.ctx, w.txt_w, w.txt_b, B, C, S, c.activation_vse, k.y_txt, k.nrm_txt, k.txt_emb,
                                        stream));
            if (has_vse)
                VAG_TRY(vag_rank_loss_fwd_impl(k.im_emb, k.txt_emb, B, S, c.margin, c.rank_kind, k.rscores, k.G, losses + 2,
                                               k.consts + 1, s));      // (G times the loss weight: no scaling pass in the backward)
        }
        VAG_TRY(vag_dec_init_fwd(k.enc, k.mask, mm ? k.ctx : nullptr, mm ? c.init_split : 0.f, w.ini_w, w.ini_b, B, Ts, C, H,
                                 k.xmix, h0, stream));                                                  // V11.py:118
        {
            // the key projection joins the decoder's per-batch products (projected keys, input projection of every step) in
            // one grouped launch: the bracket is flushed by the decoder operator's own bracket before its time loop starts.
            // (Only with the driver's derived weights: otherwise the operator first builds W_ih2 W_c2h, which the queue
            // would hold back.)
            VagGemmGroup outer(derived != nullptr);
            VAG_TRY(vag_attn_keys_proj(k.enc, w.attn_e, B * Ts, C, k.pe, stream));                      // NMT_Decoder.py:47
            if (c.free_run && vag_cgru_free_supported(B, Ts, Tt, Et, H, V))
                VAG_TRY(vag_cgru_attn_decode_free_fwd(k.enc, k.pe, k.mask, h0, k.tok, w.dec, B, Ts, Tt, Et, H, V, h2_all, k.c_all,
                                                      k.e_all, k.ws_dec, &w.head, c.p_out, crng, k.tmid, k.logits, c.ldl,
                                                      k.free_tab, stream));                             // V11.py:148-160, one launch
            else
            VAG_TRY(vag_cgru_attn_decode_seq_fwd(k.enc, k.pe, k.mask, h0, k.tok, w.dec, B, Ts, Tt, Et, H, V, h2_all, k.c_all,
                                                 k.e_all, k.ws_dec, c.free_run, &w.head, c.p_out, crng, k.tmid, k.logits,
                                                 c.ldl, stream));                                       // V11.py:138-160
            VAG_TRY(outer.end(s));
        }
        // with the head's backward in the same call the loss reduction rides in that backward's first launch
        if ((phases & (2 | 16)) && chunk == 0 && vag_opt().loss_ride != 0) vag_loss_defer_begin();
        VAG_TRY(vag_head_ce_seq_fwd_impl(h2_all, k.c_all, k.e_all, w.head, tgt, vocab_weight, B, Tt, Et, H, V, c.p_out, crng,
                                         c.free_run ? 1 : 0, k.tmid, k.logits, c.ldl, k.lse, k.nll, k.inv_cnt, 1, nullptr,
                                         losses, w_mt, w_vse, has_vse ? 1 : 0, s));                     // V11.py:140,164-166
    }
    if (one_plane && (phases & 54)) vag_gemm_set_planes(1);
    // phase 2 = its two halves 16 (head + decoder: final for the head's, the decoder's and attn_e's gradients) and 32 (visual grounding
    // + initial state: final for vse_imagine.* and decoderini.*): a data-parallel driver with three buckets calls them one by one
    if (phases & (2 | 16)) {
        // zeroed by this step's prologue launch (a backward-only call: by the forward call of the same step)
        if (chunk == 0) vag_gemm_prezeroed_set(0, k.scr_head);
        vag_gemm_prezeroed_set(1, vag_cgru_bwd_scratch_du(k.scr_dec, B, Ts, Tt, Et, H));
        VAG_TRY(vag_head_ce_seq_bwd(h2_all, k.c_all, k.e_all, w.head, tgt, vocab_weight, B, Tt, Et, H, V, c.p_out, crng, k.tmid,
                                    k.logits, c.ldl, k.lse, k.inv_cnt, k.consts + 0, k.d_h2, k.d_c, d_e, g.head, k.scr_head,
                                    stream));
        VAG_TRY(vag_loss_defer_flush());
        {
            // after the backward recurrence: the products that add into d_enc (projected keys, attention keys) and the weight
            // gradients of the decoder and of attn_e are queued by layout and go out as two grouped launches
            VagGemmGroup outer(true);
            // step_fork bit 2 (value 4; round 6): the decoder's weight-gradient products (the TN layout of this
            // bracket's flushes: eight K = Tt*B products that only the optimiser reads) leave on the low-priority side stream and run
            // beside the VSE / initial-state backward and the encoder's backward recurrence.  Only when that backward follows in this
            // call: a data-parallel driver's phase call ends here and all-reduces these gradients next.
            struct DwSide {
                ForkState* f = nullptr;
                explicit DwSide(bool on) { if (on && (f = fork_state())) vag_gemm_group_leaf_stream(f->side, f->ev[4]); }
                ~DwSide() { if (f) vag_gemm_group_leaf_stream(nullptr, nullptr); }
            } dw_side((phases & 4) && (phases & (2 | 32)) && (vag_opt().step_fork & 4));
            // the initial state's backward follows in this call: d_h0 leaves the recurrence kernel as the gradient of tanh's argument
            struct Dh0 { Dh0(bool on) { vag_persist_dh0_tanh_request(on); } ~Dh0() { vag_persist_dh0_tanh_request(false); } } dh0((phases & 2) != 0);
            VAG_TRY(vag_cgru_attn_decode_seq_bwd_loop(k.enc, k.pe, k.mask, h0, k.tok, w.dec, B, Ts, Tt, Et, H, V, h2_all, k.c_all,
                                                      k.e_all, k.d_h2, k.d_c, d_e, k.ws_dec, k.d_enc, 0, k.d_pe, k.d_h0,
                                                      k.scr_dec, stream));
            VAG_TRY(vag_attn_keys_proj_bwd(k.enc, w.attn_e, k.d_pe, B * Ts, C, k.d_enc, 1, g.attn_e, stream));
            VAG_TRY(vag_cgru_attn_decode_seq_bwd_weights(h0, k.tok, w.dec, B, Ts, Tt, Et, H, h2_all, k.c_all, k.e_all, d_e,
                                                         k.ws_dec, g.dec, k.scr_dec, stream));
            VAG_TRY(outer.end(s));
            if (dw_side.f && vag_gemm_group_leaf_used()) { br_dw.f = dw_side.f; br_dw.open = true; }      // joined at the end of the call
        }
    }
    if (phases & (2 | 32)) {
        // the weight-gradient products of the VSE branch and of the initial state (rank-B updates nothing later in the step reads)
        // and their bias sums are held back and leave as ONE launch, on a side branch that joins before the optimiser
        struct LeafScope { bool on = true; LeafScope() { vag_leaf_begin(); } ~LeafScope() { if (on) vag_leaf_abort(); } } leaf;
        // ... and the two accumulations into d_enc of this block (initial state: mean pool; visual attention: outer products) are one pass
        struct RmwScope { RmwScope(float* p) { vag_rmw_defer_begin(p); } ~RmwScope() { vag_rmw_defer_abort(); } } rmw(k.d_enc);
        if (mm) {
            if (has_vse) {
                VAG_TRY(vag_rank_loss_bwd_impl(k.im_emb, k.txt_emb, k.G, nullptr, B, S, k.d_im, k.d_txt, s));
            } else {
                VAG_TRY(vag_axpy_launch(0.f, k.d_im, k.d_im, B * S, 2, s));
                VAG_TRY(vag_axpy_launch(0.f, k.d_txt, k.d_txt, B * S, 2, s));
            }
            VAG_TRY(vag_img_proj_l2_bwd(k.ctx, w.txt_w, k.y_txt, k.nrm_txt, k.txt_emb, k.d_txt, B, C, S, c.activation_vse,
                                        k.d_ctx, g.txt_w, g.txt_b, stream));
        }
        VAG_TRY(vag_dec_init_bwd_impl(k.mask, k.xmix, h0, mm ? c.init_split : 0.f, w.ini_w, k.d_h0, B, Ts, C, H, k.d_enc, 1,
                                      mm ? k.d_ctx : nullptr, 1, g.ini_w, g.ini_b, k.scr_ini, s));
        if (mm) {
            VAG_TRY(vag_imagine_attn_ctx_bwd_impl(k.im_emb, k.enc, k.mask, w.ctx2ctx, w.emb2ctx, w.mlp_w, c.attn_method, B, Ts,
                                                  C, S, k.alpha_v, k.d_ctx, k.ws_img, k.d_enc, 1, k.d_im, 1, g.ctx2ctx,
                                                  g.emb2ctx, g.mlp_w, s));
            VAG_TRY(vag_img_proj_l2_bwd(im, w.im_w, k.y_im, k.nrm_im, k.im_emb, k.d_im, B, c.I, S, c.activation_vse, nullptr,
                                        g.im_w, g.im_b, stream));
        }
        VAG_TRY(vag_rmw_defer_flush(s));
        leaf.on = false;
        {
            // with the encoder's backward in the same call the leaves run beside its recurrence; a data-parallel driver's phase
            // call ends here (these gradients belong to the bucket it all-reduces next): joined at once
            hipStream_t sl = (phases & 4) ? br_leaf.fork(s, 2, 2) : s;
            VAG_TRY(vag_leaf_flush(sl));
        }
    }
    if (phases & 4) {
        // the last launch of the backward pass turns a give-up of any persistent recurrence of this step into a non-finite
        // gradient entry (persist.hip, optim.hip: the optimiser then skips the step, on every replica)
        struct Inject { Inject() { g_step_poison_inject = true; } ~Inject() { g_step_poison_inject = false; } } inject;
        VAG_TRY(vag_bigru_seq_bwd(src, lengths, w.enc_fw, w.enc_bw, c.p_emb, c.p_ctx, crng, B, Ts, c.Es, H, k.d_enc, k.ws_enc,
                                  g.enc_emb, g.enc_fw, g.enc_bw, stream));
    }
    VAG_TRY(br_leaf.join(s, 3));
    VAG_TRY(br_dw.join(s, 5));
    return VAG_OK;
}

}  // extern "C"
