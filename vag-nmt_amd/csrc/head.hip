// Output-head loss kernels: row-wise log-sum-exp + weighted NLL (+ argmax for free-running / greedy decode),
// the softmax-minus-onehot backward (in place over the logits buffer) and the per-sentence normalisation.
#include "kernels.h"

__device__ __forceinline__ float block_reduce_max(float v, float* sh) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    float r = sh[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = fmaxf(r, sh[i]);
    __syncthreads();
    return r;
}
__device__ __forceinline__ float block_reduce_sum(float v, float* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
    __syncthreads();
    return r;
}

// One 256-thread block per row.  NV > 0: V <= 256*NV and the row is held in registers (one read of the logits, all of
// a thread's loads in flight together); NV == 0: any V, the row is re-read from L2 by the later passes.
template <int NV>
__global__ __launch_bounds__(256) void lse_nll_kernel(const float* __restrict__ logits, int64_t ldl, int V,
                                                      const int64_t* __restrict__ tgt, int B, int Tt,
                                                      const float* __restrict__ vw, float* __restrict__ lse,
                                                      float* __restrict__ nll, int64_t* __restrict__ argmax,
                                                      int64_t argmax_stride, float* __restrict__ logp_out,
                                                      int64_t ldlp) {
    __shared__ float sh[4];
    __shared__ int shi[4];
    const int64_t row = blockIdx.x;
    const float* x = logits + row * ldl;
    float c[NV > 0 ? NV : 1];
    if (NV > 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int j = threadIdx.x + 256 * i;
            c[i] = j < V ? x[j] : -INFINITY;
        }
    }
    float mx = -INFINITY;
    int mi = 0x7fffffff;
    float sum = 0.f;
    float bm;
    if (NV > 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (c[i] > mx) { mx = c[i]; mi = threadIdx.x + 256 * i; }      // strict >: first occurrence within a thread
        bm = block_reduce_max(mx, sh);
#pragma unroll
        for (int i = 0; i < NV; ++i) sum += __expf(c[i] - bm);             // exp(-inf) = 0 past V
    } else {
        // ONE pass over the row (round 3; before: a max pass and a sum pass of 4-byte loads, the second out of L2: 1.7 TB/s
        // at V = 40000): 16-byte loads, four in flight per thread, a running (max, sum of exp) pair rescaled once per group
        // of 16 values; the threads' pairs are merged after the block-wide max.  Row bases and ldl are 16-byte aligned.
        const float4* x4 = reinterpret_cast<const float4*>(x);
        const int n4 = (V + 3) >> 2;
        for (int i0 = threadIdx.x; i0 < n4; i0 += 1024) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i4 = i0 + 256 * u;
                v[u] = i4 < n4 ? x4[i4] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            }
            float e[16];
            float gm = mx;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = 4 * (i0 + 256 * u);
                e[4 * u + 0] = j + 0 < V ? v[u].x : -INFINITY;
                e[4 * u + 1] = j + 1 < V ? v[u].y : -INFINITY;
                e[4 * u + 2] = j + 2 < V ? v[u].z : -INFINITY;
                e[4 * u + 3] = j + 3 < V ? v[u].w : -INFINITY;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (e[4 * u + q] > gm) { gm = e[4 * u + q]; mi = j + q; }          // ascending j within a thread: first occurrence
            }
            if (gm > -INFINITY) {
                float part = 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) part += __expf(e[q] - gm);                 // exp(-inf) = 0 past V
                sum = (mx > -INFINITY ? sum * __expf(mx - gm) : 0.f) + part;
                mx = gm;
            }
        }
        bm = block_reduce_max(mx, sh);
        sum = mx > -INFINITY ? sum * __expf(mx - bm) : 0.f;
    }
    sum = block_reduce_sum(sum, sh);
    const float l = bm + __logf(sum);
    if (argmax) {
        // smallest index attaining the maximum (torch.topk/argmax tie-break is unspecified; this is deterministic)
        int cand = (mx == bm) ? mi : 0x7fffffff;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
        if ((threadIdx.x & 63) == 0) shi[threadIdx.x >> 6] = cand;
        __syncthreads();
        if (threadIdx.x == 0) argmax[row * argmax_stride] = (int64_t)min(min(shi[0], shi[1]), min(shi[2], shi[3]));
    }
    if (threadIdx.x == 0) {
        if (lse) lse[row] = l;
        if (tgt) {
            const int t = (int)(row / B), b = (int)(row - (int64_t)t * B);
            const int64_t tg = tgt[(int64_t)b * Tt + t];
            nll[row] = -vw[tg] * (x[tg] - l);
        }
    }
    if (logp_out) {
        float* o = logp_out + row * ldlp;
        if (NV > 0) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int j = threadIdx.x + 256 * i;
                if (j < V) o[j] = c[i] - l;
            }
        } else {
            for (int j = threadIdx.x; j < V; j += 256) o[j] = x[j] - l;
        }
    }
}

int vag_lse_nll_launch(const float* logits, int64_t ldl, int64_t rows, int64_t V, const int64_t* tgt, int64_t B,
                       int64_t Tt, const float* vw, float* lse, float* nll, int64_t* argmax, int64_t argmax_stride,
                       float* logp_out, int64_t ldlp, hipStream_t s) {
    VAG_CHECK_ARG(logits && rows >= 0 && V > 0 && ldl >= V);
    VAG_CHECK_ARG(!tgt || (vw && nll && B > 0 && Tt > 0));
    if (rows == 0) return VAG_OK;
#define VAG_LSE_GO(NV)                                                                                              \
    hipLaunchKernelGGL(lse_nll_kernel<NV>, dim3((unsigned)rows), dim3(256), 0, s, logits, ldl, (int)V, tgt, (int)B,  \
                       (int)Tt, vw, lse, nll, argmax, argmax_stride, logp_out, ldlp)
    if (V <= 256 * 8) VAG_LSE_GO(8);
    else if (V <= 256 * 40) VAG_LSE_GO(40);
    else VAG_LSE_GO(0);
#undef VAG_LSE_GO
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// inv_cnt[b] = 1 / #(tgt[b,:] != 0)       (models/...V11.py:164: loss_mt / tgt_mask.sum(-1))
__global__ __launch_bounds__(256) void inv_cnt_kernel(const int64_t* __restrict__ tgt, int B, int Tt,
                                                      float* __restrict__ inv_cnt) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    int c = 0;
    for (int t = 0; t < Tt; ++t) c += tgt[(int64_t)b * Tt + t] != 0;
    inv_cnt[b] = 1.f / (float)c;
}
int vag_inv_cnt_launch(const int64_t* tgt, int64_t B, int64_t Tt, float* inv_cnt, hipStream_t s) {
    VAG_CHECK_ARG(tgt && inv_cnt && B > 0 && Tt > 0);
    hipLaunchKernelGGL(inv_cnt_kernel, dim3((unsigned)cdiv64(B, 256)), dim3(256), 0, s, tgt, (int)B, (int)Tt, inv_cnt);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// loss = (1/B) sum_b inv_cnt[b] sum_t nll[t,b]    (single block; fixed summation order -> deterministic).
// Thread (g, b) = (tid / 64, tid % 64) walks t = g, g+4, ... of sentences b, b+64, ...; partial sums meet in LDS.
// losses != NULL: also the weighted total of V11.py:166 (losses = {loss, loss_mt, loss_vse}).
__device__ __forceinline__ void loss_mt_body(float (&sh)[4], const float* __restrict__ nll, const float* __restrict__ inv_cnt,
                                             int B, int Tt, float* __restrict__ loss, float* __restrict__ losses,
                                             float w_mt, float w_vse, int has_vse, int ring) {
    const int g = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float acc = 0.f;
    for (int b = lane; b < B; b += 64) {
        float L = 0.f;
        for (int t = g; t < Tt; t += 4) L += nll[(int64_t)t * B + b];
        acc += L * inv_cnt[b];
    }
    acc = block_reduce_sum(acc, sh);
    if (threadIdx.x == 0) {
        const float mt = acc / (float)B;
        if (loss) loss[0] = mt;
        if (losses) {
            losses[1] = mt;
            if (!has_vse) losses[2] = 0.f;
            const float tot = w_mt * mt + (has_vse ? w_vse * losses[2] : 0.f);
            losses[0] = tot;
            if (ring > 0) {                 // vag_step_cfg.loss_ring: the n-th execution's results stay readable for `ring` steps
                const unsigned n = __float_as_uint(losses[3]);
                float* r = losses + 4 + 4 * (n % (unsigned)ring);
                r[0] = tot; r[1] = mt; r[2] = losses[2];
                losses[3] = __uint_as_float(n + 1u);
            }
        }
    }
}
__global__ __launch_bounds__(256) void loss_mt_kernel(const float* __restrict__ nll, const float* __restrict__ inv_cnt,
                                                      int B, int Tt, float* __restrict__ loss, float* __restrict__ losses,
                                                      float w_mt, float w_vse, int has_vse, int ring) {
    __shared__ float sh[4];
    loss_mt_body(sh, nll, inv_cnt, B, Tt, loss, losses, w_mt, w_vse, has_vse, ring);
}
// The loss reduction as a passenger of the launch that follows it in a training step (ce_bwd_colsum_kernel, which needs none of its
// results: d(loss) is a constant of the step): between vag_loss_defer_begin() and vag_loss_defer_flush() a vag_loss_mt_mix_launch is
// held back and handed to the next vag_ce_bwd_colsum_launch, whose block (0,0) does it first; the flush launches it on its own if
// no such launch came.  Calling thread.
struct LossTask {
    const float* nll = nullptr; const float* inv_cnt = nullptr; float* losses = nullptr;
    int B = 0, Tt = 0, has_vse = 0, ring = 0;
    float w_mt = 0.f, w_vse = 0.f;
};
static thread_local LossTask g_loss_task;
static thread_local bool g_loss_defer = false;
static thread_local hipStream_t g_loss_stream = nullptr;
void vag_loss_defer_begin() { g_loss_defer = true; g_loss_task = LossTask(); }
int vag_loss_defer_flush() {
    g_loss_defer = false;
    const LossTask t = g_loss_task;
    g_loss_task = LossTask();
    if (!t.nll) return VAG_OK;
    hipLaunchKernelGGL(loss_mt_kernel, dim3(1), dim3(256), 0, g_loss_stream, t.nll, t.inv_cnt, t.B, t.Tt, (float*)nullptr, t.losses,
                       t.w_mt, t.w_vse, t.has_vse, t.ring);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}
static thread_local int g_loss_ring = 0;        // vag_train_step sets it for its call (vag_step_cfg.loss_ring)
void vag_set_loss_ring(int r) { g_loss_ring = r; }
int vag_loss_mt_launch(const float* nll, const float* inv_cnt, int64_t B, int64_t Tt, float* loss, hipStream_t s) {
    VAG_CHECK_ARG(nll && inv_cnt && loss && B > 0 && Tt > 0);
    hipLaunchKernelGGL(loss_mt_kernel, dim3(1), dim3(256), 0, s, nll, inv_cnt, (int)B, (int)Tt, loss, (float*)nullptr, 0.f,
                       0.f, 0, 0);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}
int vag_loss_mt_mix_launch(const float* nll, const float* inv_cnt, int64_t B, int64_t Tt, float* losses, float w_mt,
                           float w_vse, int has_vse, hipStream_t s) {
    VAG_CHECK_ARG(nll && inv_cnt && losses && B > 0 && Tt > 0);
    if (g_loss_defer) {
        LossTask& t = g_loss_task;
        t.nll = nll; t.inv_cnt = inv_cnt; t.losses = losses; t.B = (int)B; t.Tt = (int)Tt; t.has_vse = has_vse; t.ring = g_loss_ring;
        t.w_mt = w_mt; t.w_vse = w_vse;
        g_loss_stream = s;
        return VAG_OK;
    }
    hipLaunchKernelGGL(loss_mt_kernel, dim3(1), dim3(256), 0, s, nll, inv_cnt, (int)B, (int)Tt, (float*)nullptr, losses, w_mt,
                       w_vse, has_vse, g_loss_ring);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// grid (ceil(ldl/1024), rows): d logits in place.
__global__ __launch_bounds__(256) void ce_bwd_kernel(float* __restrict__ logits, int64_t ldl, int V,
                                                     const int64_t* __restrict__ tgt, int B, int Tt,
                                                     const float* __restrict__ vw, const float* __restrict__ lse,
                                                     const float* __restrict__ inv_cnt, const float* __restrict__ d_loss) {
    const int64_t row = blockIdx.y;
    const int t = (int)(row / B), b = (int)(row - (int64_t)t * B);
    const int64_t tg = tgt[(int64_t)b * Tt + t];
    const float coef = d_loss[0] * inv_cnt[b] / (float)B * vw[tg];
    const float l = lse[row];
    float* x = logits + row * ldl;
    const int j0 = (blockIdx.x * 256 + threadIdx.x) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = j0 + i;
        if (j < ldl) {
            float g = 0.f;
            if (j < V) g = coef * (__expf(x[j] - l) - (j == tg ? 1.f : 0.f));
            x[j] = g;
        }
    }
}
int vag_ce_bwd_launch(float* logits, int64_t ldl, int64_t rows, int64_t V, const int64_t* tgt, int64_t B, int64_t Tt,
                      const float* vw, const float* lse, const float* inv_cnt, const float* d_loss, hipStream_t s) {
    VAG_CHECK_ARG(logits && tgt && vw && lse && inv_cnt && d_loss && rows == B * Tt && V > 0 && ldl >= V);
    if (rows == 0) return VAG_OK;
    dim3 grid((unsigned)cdiv64(ldl, 1024), (unsigned)rows);
    hipLaunchKernelGGL(ce_bwd_kernel, grid, dim3(256), 0, s, logits, ldl, (int)V, tgt, (int)B, (int)Tt, vw, lse, inv_cnt,
                       d_loss);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// The same transformation fused with the column sums of its result (the gradient of the output bias): a thread owns
// one vocabulary column over a strip of rows, so d(logits) is not read a second time (96 MB at cfg2).
// grid (ceil(ldl/256), ceil(rows/rows_per)).
__global__ __launch_bounds__(256) void ce_bwd_colsum_kernel(float* __restrict__ logits, int64_t ldl, int V,
                                                            const int64_t* __restrict__ tgt, int B, int Tt,
                                                            const float* __restrict__ vw, const float* __restrict__ lse,
                                                            const float* __restrict__ inv_cnt,
                                                            const float* __restrict__ d_loss, int rows, int rows_per,
                                                            float* __restrict__ g_bias, unsigned short* __restrict__ out16,
                                                            LossTask lt) {
    __shared__ float lsh[4];
    if (lt.nll && blockIdx.x == 0 && blockIdx.y == 0)           // a held-back loss reduction (see LossTask)
        loss_mt_body(lsh, lt.nll, lt.inv_cnt, lt.B, lt.Tt, nullptr, lt.losses, lt.w_mt, lt.w_vse, lt.has_vse, lt.ring);
    // out16 != NULL (2-byte storage mode, chunked head): d(logits) is written as bf16 into out16 (row stride ldl elements) and
    // the fp32 logits are left alone -- its two consumers round it to one bf16 plane anyway, and read half the bytes this way.
    // Round 4: a thread owns FOUR consecutive columns (16-byte loads, 16-byte fp32 / 8-byte bf16 stores; ldl % 4 == 0): with one
    // column per thread the bf16 rows went out as 2-byte stores (342 us per 655 MB chunk at configs[4] = 2.9 TB/s, 52 us at configs[1]).
    const int j = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int r0 = blockIdx.y * rows_per, r1 = j < ldl ? min(rows, r0 + rows_per) : r0;      // columns past ldl: no rows, zero sums
    const float dl = d_loss[0] / (float)B;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = r0; r < r1; r += 4) {
        float4 x[4];
        float coef[4], l[4];
        int tg[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = min(r + u, r1 - 1);                    // uniform over the block
            const int t = row / B, b = row - t * B;
            const int64_t g = tgt[(int64_t)b * Tt + t];
            tg[u] = (int)g;
            coef[u] = dl * inv_cnt[b] * vw[g];
            l[u] = lse[row];
            x[u] = *reinterpret_cast<const float4*>(logits + (int64_t)row * ldl + j);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (r + u < r1) {
                const float xv[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
                float g[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    g[q] = (j + q < V) ? coef[u] * (__expf(xv[q] - l[u]) - (j + q == tg[u] ? 1.f : 0.f)) : 0.f;
                    acc[q] += g[q];
                }
                if (out16) {
                    typedef __bf16 hbf2 __attribute__((ext_vector_type(2)));
                    typedef float hf2 __attribute__((ext_vector_type(2)));
                    const hf2 lo = {g[0], g[1]}, hi = {g[2], g[3]};
                    const hbf2 hl = __builtin_convertvector(lo, hbf2), hh = __builtin_convertvector(hi, hbf2);     // v_cvt_pk_bf16_f32
                    *reinterpret_cast<uint2*>(out16 + (int64_t)(r + u) * ldl + j) =
                        make_uint2(__builtin_bit_cast(unsigned, hl), __builtin_bit_cast(unsigned, hh));
                } else {
                    *reinterpret_cast<float4*>(logits + (int64_t)(r + u) * ldl + j) = make_float4(g[0], g[1], g[2], g[3]);
                }
            }
        }
    }
    // the strip's column sums: through LDS so that every atomic wave-instruction adds 64 CONSECUTIVE columns (256 contiguous bytes:
    // the shape float atomics run at full rate for); straight from the registers a lane would own every fourth dword
    __shared__ float cs[1024];
#pragma unroll
    for (int q = 0; q < 4; ++q) cs[4 * threadIdx.x + q] = acc[q];
    __syncthreads();
    const int jb = blockIdx.x * 1024;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = jb + q * 256 + threadIdx.x;
        if (c < V) atomicAdd(g_bias + c, cs[q * 256 + threadIdx.x]);
    }
}
int vag_ce_bwd_colsum_launch(float* logits, int64_t ldl, int64_t rows, int64_t V, const int64_t* tgt, int64_t B, int64_t Tt,
                             const float* vw, const float* lse, const float* inv_cnt, const float* d_loss, float* g_bias,
                             hipStream_t s, void* out16) {
    VAG_CHECK_ARG(logits && tgt && vw && lse && inv_cnt && d_loss && g_bias && rows % B == 0 && rows <= B * Tt && V > 0 &&
                  ldl >= V);      // rows < B*Tt: a chunk of whole time steps (tgt / lse already offset by the caller)
    if (rows == 0) return VAG_OK;
    VAG_CHECK_ARG(ldl % 4 == 0 && aligned16(logits) && (!out16 || (reinterpret_cast<uintptr_t>(out16) & 7) == 0));
    const int64_t nbx = cdiv64(ldl, 1024);
    int64_t splits = cdiv64(1024, nbx);                  // ~1024 blocks: every strip costs ldl atomics
    if (splits > cdiv64(rows, 8)) splits = cdiv64(rows, 8);
    const int rows_per = (int)cdiv64(rows, splits);
    dim3 grid((unsigned)nbx, (unsigned)cdiv64(rows, rows_per));
    LossTask lt;
    if (g_loss_task.nll && g_loss_stream == s) { lt = g_loss_task; g_loss_task = LossTask(); }
    hipLaunchKernelGGL(ce_bwd_colsum_kernel, grid, dim3(256), 0, s, logits, ldl, (int)V, tgt, (int)B, (int)Tt, vw, lse,
                       inv_cnt, d_loss, (int)rows, rows_per, g_bias, reinterpret_cast<unsigned short*>(out16), lt);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// d logits from d log-softmax, in place over d_logp: dl_j = d_j - exp(logp_j) * sum_j d_j.  One block per row.
__global__ __launch_bounds__(256) void logsoftmax_bwd_kernel(const float* __restrict__ logp, int64_t ldlp,
                                                             float* __restrict__ d, int64_t ldd, int V) {
    __shared__ float sh[4];
    const int64_t row = blockIdx.x;
    const float* lp = logp + row * ldlp;
    float* dr = d + row * ldd;
    float sum = 0.f;
    for (int j = threadIdx.x; j < V; j += 256) sum += dr[j];
    sum = block_reduce_sum(sum, sh);
    for (int j = threadIdx.x; j < V; j += 256) dr[j] = dr[j] - __expf(lp[j]) * sum;
    for (int j = V + threadIdx.x; j < ldd; j += 256) dr[j] = 0.f;
}
int vag_logsoftmax_bwd_launch(const float* logp, int64_t ldlp, float* d, int64_t ldd, int64_t rows, int64_t V,
                              hipStream_t s) {
    VAG_CHECK_ARG(logp && d && rows >= 0 && V > 0 && ldlp >= V && ldd >= V);
    if (rows == 0) return VAG_OK;
    hipLaunchKernelGGL(logsoftmax_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, s, logp, ldlp, d, ldd, (int)V);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}
