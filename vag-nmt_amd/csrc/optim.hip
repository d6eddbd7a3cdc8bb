// Fused global-norm clip + Adam over one flat fp32 parameter buffer (train.py:46-49).
// Three launches: sum of squares (per-block partials into 128 slots), scalar prep (norm, clip coefficient, bias
// corrections in double, step counter; re-zeroes the slots) and the streaming update of every param group in one grid
// (7 x 4 B per parameter), which also leaves the gradient buffer zeroed for the next step.
#include "kernels.h"

constexpr int SUMSQ_SLOTS = 128;
struct AdamScalars {          // lives in the caller's scratch (VAG_ADAM_SCRATCH_BYTES); zero before the first call
    double sumsq;
    float coef;               // grad_scale * min(1, clip / (norm + 1e-6))
    float bc1;                // 1 - beta1^t
    float bc2_sqrt;           // sqrt(1 - beta2^t)
    float lr_base;            // multiplies every segment's learning rate: *lr_dev, or 1
    unsigned skip;            // 1: this step's gradient is void (non-finite norm, or a persistent recurrence gave up a wait):
                              // adam_kernel leaves parameters and moments alone and only zeroes the gradient
    unsigned skipped;         // number of skipped steps so far (VAG_ADAM_SCRATCH_SKIPPED_OFFSET: hosts read it from the scratch)
    unsigned guard[2];        // VAG_ADAM_SCRATCH_GUARD_OFFSET: {void flag, give-up count} of THIS driver's persistent recurrence launches
                              // (persist.hip: note_timeout; the driver passes the address as vag_step_cfg.guard)
    // the blocks' partial sums land in 128 slots (2048 double atomics on ONE address serialise in L2: ~25 of the
    // pass's 31 us); the last block to arrive adds the slots up in a fixed order
    double part[SUMSQ_SLOTS];
};
static_assert(sizeof(AdamScalars) <= 2048, "vag_clip_adam_flat scratch contract");
static_assert(offsetof(AdamScalars, skipped) == VAG_ADAM_SCRATCH_SKIPPED_OFFSET, "vag_nmt.h: VAG_ADAM_SCRATCH_SKIPPED_OFFSET");
static_assert(offsetof(AdamScalars, guard) == VAG_ADAM_SCRATCH_GUARD_OFFSET, "vag_nmt.h: VAG_ADAM_SCRATCH_GUARD_OFFSET");

// Pass 1: sum of squares of the gradient into 128 slots (plain relaxed atomics, no fences: an in-kernel "last block does the
// scalar work" variant needs a device-scope fence per block, and 2048 of them cost 80 us -- measured -- against 2 us for
// the tiny kernel below).
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, AdamScalars* sc) {
    __shared__ double sh[4];
    double acc = 0.0;
    const int64_t n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    // four 16-byte loads in flight per thread (the loop is latency-bound otherwise: 33 -> ~12 us for 64 MB)
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = blockIdx.x * 256ll + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const float4 a = g4[i], b = g4[i + stride], c = g4[i + 2 * stride], d = g4[i + 3 * stride];
        const float fa = a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
        const float fb = b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
        const float fc = c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w;
        const float fd = d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
        acc += ((double)fa + (double)fb) + ((double)fc + (double)fd);
    }
    for (; i < n4; i += stride) {
        const float4 v = g4[i];
        acc += (double)(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float v = g[(n4 << 2) + threadIdx.x];
        acc += (double)v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&sc->part[blockIdx.x % SUMSQ_SLOTS], sh[0] + sh[1] + sh[2] + sh[3]);
}

// Scalar work between the passes (norm, clip coefficient, bias corrections in double, step counter); leaves the slots zeroed
// for the next call, so no separate zeroing launch is needed.
__global__ void adam_prep_kernel(AdamScalars* sc, float clip, float grad_scale, float beta1, float beta2, int32_t* step,
                                 float* norm_out, const float* lr_dev) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double ss = 0.0;
    for (int k = 0; k < SUMSQ_SLOTS; ++k) {                        // fixed order
        ss += sc->part[k];
        sc->part[k] = 0.0;
    }
    sc->sumsq = ss;
    // A void gradient is never applied: the step is skipped when a persistent recurrence kernel OF THIS DRIVER gave up a wait since
    // its last optimiser step (persist.hip: results of that launch are void; the flag lives in this scratch, so another model's
    // or a decoder's give-up on the same device does not reach here) or when the gradient norm is not finite (which is also how a
    // give-up on ANOTHER data-parallel replica arrives here: elem.hip, embed_scatter_kernel).  Skipped: no parameter, moment or
    // step-counter change; the gradient buffer is still zeroed; the reported norm is NaN; `skipped` counts.
    bool bad = !(ss == ss) || ss > 1.0e300 || ss * (double)grad_scale * (double)grad_scale > 3.0e38;
    if (__hip_atomic_load(&sc->guard[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        bad = true;
        __hip_atomic_store(&sc->guard[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    sc->skip = bad ? 1u : 0u;
    if (bad) {
        sc->skipped += 1u;
        sc->coef = 0.f; sc->bc1 = 1.f; sc->bc2_sqrt = 1.f; sc->lr_base = 0.f;
        if (norm_out) norm_out[0] = __builtin_nanf("");
        return;
    }
    const double norm = sqrt(ss) * (double)grad_scale;             // norm of the scaled (averaged) gradient
    double c = (double)clip / (norm + 1e-6);                        // torch.nn.utils.clip_grad_norm_
    if (c > 1.0) c = 1.0;
    if (clip <= 0.f) c = 1.0;
    const int t = step[0] + 1;
    step[0] = t;
    sc->coef = (float)(c * (double)grad_scale);
    sc->bc1 = (float)(1.0 - pow((double)beta1, (double)t));
    sc->bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)t));
    sc->lr_base = lr_dev ? lr_dev[0] : 1.f;
    if (norm_out) norm_out[0] = (float)norm;
}

// Sharded form (vag_clip_adam_shard): the slots' sum into a caller-visible double (which the caller all-reduces over the ranks),
// slots re-zeroed.
__global__ void sumsq_collect_kernel(AdamScalars* sc, double* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double ss = 0.0;
    for (int k = 0; k < SUMSQ_SLOTS; ++k) { ss += sc->part[k]; sc->part[k] = 0.0; }
    out[0] = ss;
}
// ... and, after that all-reduce, the total back into slot 0, where adam_prep_kernel looks for it
__global__ void sumsq_restore_kernel(AdamScalars* sc, const double* in) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    sc->part[0] = in[0];
}

constexpr int ADAM_MAX_SEG = 16;
struct AdamSegs {
    int64_t off[ADAM_MAX_SEG], cnt[ADAM_MAX_SEG];
    float lr[ADAM_MAX_SEG], wd[ADAM_MAX_SEG];
};
// Pass 2: the streaming update, every segment (param group) in one grid (blockIdx.y = segment).  zero_grad: the gradient
// buffer is left zeroed for the next step's accumulation (saves the separate 4 B/param fill pass).
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, AdamSegs sg, float beta1, float beta2, float eps,
                                                   int zero_grad, const AdamScalars* __restrict__ sc) {
    const int seg = blockIdx.y;
    const int64_t off = sg.off[seg], cnt = sg.cnt[seg];
    const float lr = sg.lr[seg], wd = sg.wd[seg];
    const float coef = sc->coef;
    const float step_size = lr * sc->lr_base / sc->bc1;
    const float inv_bc2s = 1.f / sc->bc2_sqrt;
    if (sc->skip) {                                                 // void gradient (adam_prep_kernel): only throw it away
        if (zero_grad)
            for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * 256) g[off + i] = 0.f;
        return;
    }
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * 256) {
        const int64_t k = off + i;
        const float pk = p[k];
        float gk = g[k] * coef;
        if (wd != 0.f) gk += wd * pk;                               // L2 weight decay (torch.optim.Adam)
        const float mk = beta1 * m[k] + (1.f - beta1) * gk;
        const float vk = beta2 * v[k] + (1.f - beta2) * gk * gk;
        m[k] = mk;
        v[k] = vk;
        if (zero_grad) g[k] = 0.f;
        const float denom = sqrtf(vk) * inv_bc2s + eps;
        p[k] = pk - step_size * (mk / denom);
    }
}

int vag_clip_adam_launch(float* p, float* g, float* m, float* v, int64_t n, int nseg, const int64_t* seg_off,
                         const float* seg_lr, const float* seg_wd, float clip, float grad_scale, float beta1,
                         float beta2, float eps, int zero_grad, int32_t* step, float* norm_out, void* scratch,
                         const float* lr_dev, hipStream_t s) {
    VAG_CHECK_ARG(p && g && m && v && n > 0 && nseg >= 1 && nseg <= ADAM_MAX_SEG && seg_off && seg_lr && seg_wd && step &&
                  scratch);
    VAG_CHECK_ARG(aligned16(g) && seg_off[0] == 0 && seg_off[nseg] == n);
    AdamScalars* sc = reinterpret_cast<AdamScalars*>(scratch);
    int64_t blocks = cdiv64(n / 4 + 1, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, s, g, n, sc);
    VAG_LAUNCH_CHECK();
    hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(64), 0, s, sc, clip, grad_scale, beta1, beta2, step, norm_out, lr_dev);
    VAG_LAUNCH_CHECK();
    AdamSegs sg;
    int64_t maxcnt = 0;
    for (int i = 0; i < nseg; ++i) {
        sg.off[i] = seg_off[i]; sg.cnt[i] = seg_off[i + 1] - seg_off[i];
        sg.lr[i] = seg_lr[i]; sg.wd[i] = seg_wd[i];
        VAG_CHECK_ARG(sg.cnt[i] >= 0);
        if (sg.cnt[i] > maxcnt) maxcnt = sg.cnt[i];
    }
    int64_t b = cdiv64(maxcnt, 256);
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)b, (unsigned)nseg), dim3(256), 0, s, p, g, m, v, sg, beta1, beta2, eps,
                       zero_grad, sc);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// ZeRO-1 style sharded optimiser step (SURVEY 8e option; train.py:46-49 semantics unchanged): every rank holds the whole flat
// parameter and gradient buffers but updates only ONE contiguous shard [lo, hi) of them -- the gradient arrives by reduce-scatter
// (each rank receives the sum of its shard), the updated shards leave by all-gather; Adam's moments exist for the shard only as far
// as traffic goes (7 x 4 B per parameter / world per step instead of per rank).
//   phase 0: sumsq[0] = sum of squares of g[lo, hi)                 -> the caller all-reduces sumsq[0] (sum) over the ranks
//   phase 1: clip coefficient from sumsq[0] (the GLOBAL sum now), step counter, Adam on the segments cut to [lo, hi);
//            zero_grad: the WHOLE gradient buffer g[0, n) is left zeroed (the next step accumulates into all of it)
int vag_clip_adam_shard_launch(float* p, float* g, float* m, float* v, int64_t n, int nseg, const int64_t* seg_off,
                               const float* seg_lr, const float* seg_wd, float clip, float grad_scale, float beta1, float beta2,
                               float eps, int zero_grad, int32_t* step, float* norm_out, void* scratch, const float* lr_dev,
                               int64_t lo, int64_t hi, int phase, double* sumsq, hipStream_t s) {
    VAG_CHECK_ARG(p && g && m && v && n > 0 && nseg >= 1 && nseg <= ADAM_MAX_SEG && seg_off && seg_lr && seg_wd && step && scratch && sumsq);
    VAG_CHECK_ARG(aligned16(g) && seg_off[0] == 0 && seg_off[nseg] == n && 0 <= lo && lo <= hi && hi <= n && lo % 4 == 0);
    VAG_CHECK_ARG(phase == 0 || phase == 1);
    AdamScalars* sc = reinterpret_cast<AdamScalars*>(scratch);
    if (phase == 0) {
        if (hi > lo) {
            int64_t blocks = cdiv64((hi - lo) / 4 + 1, 256);
            if (blocks > 2048) blocks = 2048;
            hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, s, g + lo, hi - lo, sc);
            VAG_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(sumsq_collect_kernel, dim3(1), dim3(64), 0, s, sc, sumsq);
        VAG_LAUNCH_CHECK();
        return VAG_OK;
    }
    hipLaunchKernelGGL(sumsq_restore_kernel, dim3(1), dim3(64), 0, s, sc, sumsq);
    VAG_LAUNCH_CHECK();
    hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(64), 0, s, sc, clip, grad_scale, beta1, beta2, step, norm_out, lr_dev);
    VAG_LAUNCH_CHECK();
    AdamSegs sg;
    int ns = 0;
    int64_t maxcnt = 0;
    for (int i = 0; i < nseg; ++i) {                       // the segments cut to the shard
        const int64_t a = seg_off[i] > lo ? seg_off[i] : lo, b = seg_off[i + 1] < hi ? seg_off[i + 1] : hi;
        VAG_CHECK_ARG(seg_off[i + 1] >= seg_off[i]);
        if (b <= a) continue;
        sg.off[ns] = a; sg.cnt[ns] = b - a; sg.lr[ns] = seg_lr[i]; sg.wd[ns] = seg_wd[i];
        if (b - a > maxcnt) maxcnt = b - a;
        ++ns;
    }
    if (ns > 0) {
        int64_t b = cdiv64(maxcnt, 256);
        if (b > 4096) b = 4096;
        hipLaunchKernelGGL(adam_kernel, dim3((unsigned)b, (unsigned)ns), dim3(256), 0, s, p, g, m, v, sg, beta1, beta2, eps, zero_grad, sc);
        VAG_LAUNCH_CHECK();
    }
    if (zero_grad) {                                        // what lies outside the shard (other ranks' sums, partly reduced data)
        if (lo > 0) VAG_TRY(vag_axpy_launch(0.f, g, g, lo, 2, s));
        if (hi < n) VAG_TRY(vag_axpy_launch(0.f, g + hi, g + hi, n - hi, 2, s));
    }
    return VAG_OK;
}
