// Fused global-norm clip + Adam over one flat fp32 parameter buffer (train.py:46-49).
// Three launches: sum of squares (per-block partials -> one double atomic), scalar prep (norm, clip coefficient,
// bias corrections in double, step counter), and the streaming update (7 x 4 B per parameter).
#include "kernels.h"

constexpr int SUMSQ_SLOTS = 128;
struct AdamScalars {          // lives in the caller's scratch (VAG_ADAM_SCRATCH_BYTES)
    double sumsq;
    float coef;               // grad_scale * min(1, clip / (norm + 1e-6))
    float bc1;                // 1 - beta1^t
    float bc2_sqrt;           // sqrt(1 - beta2^t)
    float pad;
    // the blocks' partial sums land in 128 slots (2048 double atomics on ONE address serialise in L2: ~25 of the
    // pass's 31 us); adam_prep_kernel adds the slots up
    double part[SUMSQ_SLOTS];
};
static_assert(sizeof(AdamScalars) <= 2048, "vag_clip_adam_flat scratch contract");

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

__global__ void adam_zero_kernel(AdamScalars* sc) {
    if (threadIdx.x == 0) { sc->sumsq = 0.0; sc->coef = 0.f; sc->bc1 = 1.f; sc->bc2_sqrt = 1.f; sc->pad = 0.f; }
    for (int i = threadIdx.x; i < SUMSQ_SLOTS; i += blockDim.x) sc->part[i] = 0.0;
}

__global__ void adam_prep_kernel(AdamScalars* sc, float clip, float grad_scale, float beta1, float beta2,
                                 int32_t* step, float* norm_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double ss = 0.0;
    for (int i = 0; i < SUMSQ_SLOTS; ++i) ss += sc->part[i];       // fixed order
    sc->sumsq = ss;
    const double norm = sqrt(ss) * (double)grad_scale;             // norm of the scaled (averaged) gradient
    double c = (double)clip / (norm + 1e-6);                        // torch.nn.utils.clip_grad_norm_
    if (c > 1.0) c = 1.0;
    if (clip <= 0.f) c = 1.0;
    const int t = step[0] + 1;
    step[0] = t;
    sc->coef = (float)(c * (double)grad_scale);
    sc->bc1 = (float)(1.0 - pow((double)beta1, (double)t));
    sc->bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)t));
    if (norm_out) norm_out[0] = (float)norm;
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t off,
                                                   int64_t cnt, float lr, float wd, float beta1, float beta2, float eps,
                                                   const AdamScalars* __restrict__ sc) {
    const float coef = sc->coef;
    const float step_size = lr / sc->bc1;
    const float inv_bc2s = 1.f / sc->bc2_sqrt;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * 256) {
        const int64_t k = off + i;
        const float pk = p[k];
        float gk = g[k] * coef;
        if (wd != 0.f) gk += wd * pk;                               // L2 weight decay (torch.optim.Adam)
        const float mk = beta1 * m[k] + (1.f - beta1) * gk;
        const float vk = beta2 * v[k] + (1.f - beta2) * gk * gk;
        m[k] = mk;
        v[k] = vk;
        const float denom = sqrtf(vk) * inv_bc2s + eps;
        p[k] = pk - step_size * (mk / denom);
    }
}

int vag_clip_adam_launch(float* p, const float* g, float* m, float* v, int64_t n, int nseg, const int64_t* seg_off,
                         const float* seg_lr, const float* seg_wd, float clip, float grad_scale, float beta1,
                         float beta2, float eps, int32_t* step, float* norm_out, void* scratch, hipStream_t s) {
    VAG_CHECK_ARG(p && g && m && v && n > 0 && nseg >= 1 && nseg <= 16 && seg_off && seg_lr && seg_wd && step && scratch);
    VAG_CHECK_ARG(aligned16(g) && seg_off[0] == 0 && seg_off[nseg] == n);
    AdamScalars* sc = reinterpret_cast<AdamScalars*>(scratch);
    hipLaunchKernelGGL(adam_zero_kernel, dim3(1), dim3(64), 0, s, sc);
    VAG_LAUNCH_CHECK();
    int64_t blocks = cdiv64(n / 4 + 1, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, s, g, n, sc);
    VAG_LAUNCH_CHECK();
    hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(64), 0, s, sc, clip, grad_scale, beta1, beta2, step, norm_out);
    VAG_LAUNCH_CHECK();
    for (int i = 0; i < nseg; ++i) {
        const int64_t off = seg_off[i], cnt = seg_off[i + 1] - off;
        VAG_CHECK_ARG(cnt >= 0);
        if (cnt == 0) continue;
        int64_t b = cdiv64(cnt, 256);
        if (b > 4096) b = 4096;
        hipLaunchKernelGGL(adam_kernel, dim3((unsigned)b), dim3(256), 0, s, p, g, m, v, off, cnt, seg_lr[i], seg_wd[i],
                           beta1, beta2, eps, sc);
        VAG_LAUNCH_CHECK();
    }
    return VAG_OK;
}
