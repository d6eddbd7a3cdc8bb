// Attention kernels (Bahdanau MLP attention of the decoder, image-conditioned attention of the VSE module).
// HBM/L2-bound streaming over the (B,Ts,C) key (pe) and value (enc) tensors: one wave per (row, position)
// with 16-byte loads for the score pass, wavefront shuffle reductions, LDS softmax.
#include "kernels.h"

// ------------------------------------------------------------------ scores
template <int MODE>
__global__ __launch_bounds__(256) void attn_scores_kernel(const float* __restrict__ pe, const float* __restrict__ q,
                                                          const float* __restrict__ v, const float* __restrict__ mask,
                                                          int64_t total, int rps, int Ts, int C, int64_t ldq,
                                                          float* __restrict__ scores) {
    // grid (ceil(Ts/4), N): no 64-bit divisions on the way to the first load (they cost more than the row's arithmetic)
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= Ts) return;
    const int64_t n = blockIdx.y;
    const int64_t pair = n * Ts + s;
    const int64_t b = rps == 1 ? n : (int64_t)((int)blockIdx.y / rps);
    const float* pr = pe + (b * Ts + s) * C;
    const float* qr = q + n * ldq;
    float acc = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 pv = *reinterpret_cast<const float4*>(pr + c);
        const float4 qv = *reinterpret_cast<const float4*>(qr + c);
        if (MODE == 0) {
            const float4 vv = *reinterpret_cast<const float4*>(v + c);
            acc += vv.x * vag_tanh(pv.x + qv.x);
            acc += vv.y * vag_tanh(pv.y + qv.y);
            acc += vv.z * vag_tanh(pv.z + qv.z);
            acc += vv.w * vag_tanh(pv.w + qv.w);
        } else {
            acc += pv.x * qv.x + pv.y * qv.y + pv.z * qv.z + pv.w * qv.w;
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) {
        if (mask && mask[b * Ts + s] == 0.f) acc = -INFINITY;
        scores[pair] = acc;
    }
}

int vag_attn_scores_launch(int mode, const float* pe, const float* q, int64_t ldq, const float* v, const float* mask,
                           int64_t N, int64_t rps, int64_t Ts, int64_t C, float* scores, hipStream_t s) {
    VAG_CHECK_ARG(pe && q && scores && N > 0 && Ts > 0 && C > 0 && C % 4 == 0 && rps >= 1 && ldq % 4 == 0 && ldq >= C);
    VAG_CHECK_ARG(mode == 1 || v);
    const int64_t total = N * Ts;
    VAG_CHECK_ARG(N < 65536);
    dim3 grid((unsigned)cdiv64(Ts, 4), (unsigned)N);
    if (mode == 0)
        hipLaunchKernelGGL(attn_scores_kernel<0>, grid, dim3(256), 0, s, pe, q, v, mask, total, (int)rps, (int)Ts, (int)C, ldq, scores);
    else
        hipLaunchKernelGGL(attn_scores_kernel<1>, grid, dim3(256), 0, s, pe, q, v, mask, total, (int)rps, (int)Ts, (int)C, ldq, scores);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// ------------------------------------------------------------------ softmax + context
// grid (ceil(C/256), N), CTX_WAVES waves per workgroup: lane owns one float4 of c, wave w walks the source positions
// s = w, w + CTX_WAVES, ... with 5 value rows in flight, partial sums meet in LDS.  (One wave per workgroup streamed its
// 41 KB in ~3.5 us; tools/stream_probe.hip: the more waves of a CU have loads in flight, the closer to HBM rate.)
// The Ts-element softmax is recomputed by every wave (it is tiny).
constexpr int CTX_WAVES = 4;
__global__ __launch_bounds__(64 * CTX_WAVES) void attn_ctx_kernel(int softmax, const float* __restrict__ scores,
                                                      const float* __restrict__ enc, int rps, int Ts, int C,
                                                      float* __restrict__ alpha, float* __restrict__ ctx) {
    extern __shared__ __attribute__((aligned(16))) float w[];   // Ts weights, then CTX_WAVES x 64 float4 partials
    float4* part = reinterpret_cast<float4*>(w + ((Ts + 3) & ~3));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n = blockIdx.y;
    const int64_t b = rps == 1 ? n : (int64_t)((int)blockIdx.y / rps);
    const float* sc = scores + n * Ts;
    if (softmax) {
        float mx = -INFINITY;
        for (int s = lane; s < Ts; s += 64) mx = fmaxf(mx, sc[s]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int s = lane; s < Ts; s += 64) sum += __expf(sc[s] - mx);
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        for (int s = threadIdx.x; s < Ts; s += 64 * CTX_WAVES) {
            const float a = __expf(sc[s] - mx) * inv;
            w[s] = a;
            if (blockIdx.x == 0 && alpha) alpha[n * Ts + s] = a;
        }
    } else {
        for (int s = threadIdx.x; s < Ts; s += 64 * CTX_WAVES) w[s] = sc[s];
    }
    __syncthreads();
    const int c = (blockIdx.x * 64 + lane) * 4;
    const bool cok = c < C;
    float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cok) {
        const float* e = enc + b * Ts * C + c;
        constexpr int U = 5;
        for (int s0 = wave; s0 < Ts; s0 += U * CTX_WAVES) {
            float4 ev[U];
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const int s = min(s0 + i * CTX_WAVES, Ts - 1);
                ev[i] = *reinterpret_cast<const float4*>(e + (int64_t)s * C);
            }
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const int s = s0 + i * CTX_WAVES;
                const float a = (s < Ts) ? w[s] : 0.f;
                acc0.x += a * ev[i].x; acc0.y += a * ev[i].y; acc0.z += a * ev[i].z; acc0.w += a * ev[i].w;
            }
        }
    }
    part[wave * 64 + lane] = acc0;
    __syncthreads();
    if (wave == 0 && cok) {
#pragma unroll
        for (int k = 1; k < CTX_WAVES; ++k) {
            const float4 o = part[k * 64 + lane];
            acc0.x += o.x; acc0.y += o.y; acc0.z += o.z; acc0.w += o.w;
        }
        *reinterpret_cast<float4*>(ctx + n * C + c) = acc0;
    }
}

int vag_attn_ctx_launch(int softmax, const float* scores, const float* enc, int64_t N, int64_t rps, int64_t Ts,
                        int64_t C, float* alpha, float* ctx, hipStream_t s) {
    VAG_CHECK_ARG(scores && enc && ctx && N > 0 && Ts > 0 && C > 0 && C % 4 == 0 && rps >= 1);
    dim3 grid((unsigned)cdiv64(C, 256), (unsigned)N);
    const size_t lds = (size_t)((Ts + 3) & ~3) * sizeof(float) + (size_t)CTX_WAVES * 64 * 16;
    hipLaunchKernelGGL(attn_ctx_kernel, grid, dim3(64 * CTX_WAVES), lds, s, softmax, scores, enc, (int)rps,
                       (int)Ts, (int)C, alpha, ctx);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// (A one-launch scores + softmax + context kernel -- one 1024-thread workgroup per query row, 3.8 us for its 328 KB by
// tools/stream_probe.hip -- measured the same 9.9 us as the two launches: the Ts*C tanh evaluations of a row are bound
// by the transcendental rate of the 64 CUs that then hold them, 3.2 us, instead of 0.8 us spread over the chip.)

// ------------------------------------------------------------------ softmax backward (one wave per row)
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ alpha, const float* __restrict__ dalpha,
                                                          int64_t N, int Ts, float* __restrict__ dscore) {
    const int lane = threadIdx.x & 63;
    const int64_t n = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (n >= N) return;
    float dot = 0.f;
    for (int s = lane; s < Ts; s += 64) dot += alpha[n * Ts + s] * dalpha[n * Ts + s];
    dot = wave_sum(dot);
    for (int s = lane; s < Ts; s += 64) dscore[n * Ts + s] = alpha[n * Ts + s] * (dalpha[n * Ts + s] - dot);
}
int vag_softmax_bwd_launch(const float* alpha, const float* dalpha, int64_t N, int64_t Ts, float* dscore, hipStream_t s) {
    VAG_CHECK_ARG(alpha && dalpha && dscore && N > 0 && Ts > 0);
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)cdiv64(N, 4)), dim3(256), 0, s, alpha, dalpha, N, (int)Ts, dscore);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// ------------------------------------------------------------------ dq (inside the backward time loop)
// grid (ceil(C/256), N), 4 waves per workgroup: lane owns a float4 of c, wave w walks the source positions s = w, w+4, ...
// (the Ts*C tanh evaluations of a row are transcendental-rate bound: one wave per (row, column block) left three of a
// CU's four SIMDs idle), partial sums meet in LDS.  With alpha/dalpha given, the softmax backward
// ds = alpha * (dalpha - sum alpha dalpha) is done in the prologue (the first column block stores it for the post-loop
// kernel).
constexpr int DQ_WAVES = 4;
__global__ __launch_bounds__(64 * DQ_WAVES) void attn_dq_kernel(const float* __restrict__ pe, const float* __restrict__ q,
                                                     int64_t ldq, const float* __restrict__ v,
                                                     const float* __restrict__ alpha, const float* __restrict__ dalpha,
                                                     float* __restrict__ dscore, int Ts, int C, float* __restrict__ dq,
                                                     int64_t lddq) {
    extern __shared__ __attribute__((aligned(16))) float w[];      // Ts weights, then DQ_WAVES x 64 float4 partials
    float4* part = reinterpret_cast<float4*>(w + ((Ts + 3) & ~3));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n = blockIdx.y;
    if (alpha) {
        float dot = 0.f;
        for (int s = lane; s < Ts; s += 64) dot += alpha[n * Ts + s] * dalpha[n * Ts + s];
        dot = wave_sum(dot);
        for (int s = threadIdx.x; s < Ts; s += 64 * DQ_WAVES) {
            const float d = alpha[n * Ts + s] * (dalpha[n * Ts + s] - dot);
            w[s] = d;
            if (blockIdx.x == 0) dscore[n * Ts + s] = d;
        }
    } else {
        for (int s = threadIdx.x; s < Ts; s += 64 * DQ_WAVES) w[s] = dscore[n * Ts + s];
    }
    __syncthreads();
    const int c = (blockIdx.x * 64 + lane) * 4;
    const bool cok = c < C;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cok) {
        const float4 qv = *reinterpret_cast<const float4*>(q + n * ldq + c);
        const float* p = pe + n * Ts * C + c;
        // 4 key rows in flight per lane
        for (int s0 = wave; s0 < Ts; s0 += 4 * DQ_WAVES) {
            float4 pv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int s = min(s0 + i * DQ_WAVES, Ts - 1);
                pv[i] = *reinterpret_cast<const float4*>(p + (int64_t)s * C);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int s = s0 + i * DQ_WAVES;
                const float d = (s < Ts) ? w[s] : 0.f;             // masked positions carry d = 0
                float th;
                th = vag_tanh(pv[i].x + qv.x); acc.x += d * (1.f - th * th);
                th = vag_tanh(pv[i].y + qv.y); acc.y += d * (1.f - th * th);
                th = vag_tanh(pv[i].z + qv.z); acc.z += d * (1.f - th * th);
                th = vag_tanh(pv[i].w + qv.w); acc.w += d * (1.f - th * th);
            }
        }
    }
    part[wave * 64 + lane] = acc;
    __syncthreads();
    if (wave == 0 && cok) {
#pragma unroll
        for (int k = 1; k < DQ_WAVES; ++k) {
            const float4 o = part[k * 64 + lane];
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
        }
        const float4 vv = *reinterpret_cast<const float4*>(v + c);
        acc.x *= vv.x; acc.y *= vv.y; acc.z *= vv.z; acc.w *= vv.w;
        *reinterpret_cast<float4*>(dq + n * lddq + c) = acc;
    }
}
int vag_attn_dq_launch(const float* pe, const float* q, int64_t ldq, const float* v, const float* alpha,
                       const float* dalpha, float* dscore, int64_t N, int64_t Ts, int64_t C, float* dq, int64_t lddq,
                       hipStream_t s) {
    VAG_CHECK_ARG(pe && q && v && dscore && dq && N > 0 && Ts > 0 && C > 0 && C % 4 == 0 && ldq % 4 == 0 && lddq % 4 == 0);
    VAG_CHECK_ARG((alpha == nullptr) == (dalpha == nullptr));
    dim3 grid((unsigned)cdiv64(C, 256), (unsigned)N);
    const size_t lds = (size_t)((Ts + 3) & ~3) * sizeof(float) + (size_t)DQ_WAVES * 64 * 16;
    hipLaunchKernelGGL(attn_dq_kernel, grid, dim3(64 * DQ_WAVES), lds, s, pe, q, ldq, v, alpha, dalpha, dscore,
                       (int)Ts, (int)C, dq, lddq);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// ------------------------------------------------------------------ after the time loop: d_pe, dv partials, d_enc
// grid (ceil(C/256), B); thread owns one c and walks source positions in chunks of SC, all Tt steps per chunk.
constexpr int SC = VAG_POST_SC;
__global__ __launch_bounds__(256) void attn_post_bwd_kernel(const float* __restrict__ pe, const float* __restrict__ q_all,
                                                            const float* __restrict__ v, const float* __restrict__ ds_all,
                                                            const float* __restrict__ alpha_all,
                                                            const float* __restrict__ dc_all, int B, int Ts, int Tt, int C,
                                                            int64_t ldq, float* __restrict__ d_pe, float* __restrict__ dvp,
                                                            float* __restrict__ d_enc, int acc_enc) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float vc = v[c];
    float dv = 0.f;
    {
        const int s0 = blockIdx.z * SC;          // one chunk of SC source positions per block (grid.z chunks)
        float pv[SC], ape[SC], aen[SC];
#pragma unroll
        for (int i = 0; i < SC; ++i) {
            const int s = min(s0 + i, Ts - 1);
            pv[i] = pe[((int64_t)b * Ts + s) * C + c];
            ape[i] = 0.f; aen[i] = 0.f;
        }
        for (int t = 0; t < Tt; ++t) {
            const float qv = q_all[((int64_t)t * B + b) * ldq + c];
            const float dcv = dc_all ? dc_all[((int64_t)t * B + b) * C + c] : 0.f;
            const float* dsr = ds_all + ((int64_t)t * B + b) * Ts;
            const float* alr = alpha_all + ((int64_t)t * B + b) * Ts;
#pragma unroll
            for (int i = 0; i < SC; ++i) {
                const int s = min(s0 + i, Ts - 1);
                const float d = dsr[s];
                const float th = vag_tanh(pv[i] + qv);
                ape[i] += d * (1.f - th * th);
                if (s0 + i < Ts) dv += d * th;
                aen[i] += alr[s] * dcv;
            }
        }
#pragma unroll
        for (int i = 0; i < SC; ++i) {
            const int s = s0 + i;
            if (s < Ts) {
                const int64_t o = ((int64_t)b * Ts + s) * C + c;
                d_pe[o] = vc * ape[i];
                if (d_enc && dc_all) d_enc[o] = acc_enc ? d_enc[o] + aen[i] : aen[i];
            }
        }
    }
    if (dvp) dvp[((int64_t)blockIdx.z * B + b) * C + c] = dv;      // partial per chunk: rows z*B + b
}
int vag_attn_post_bwd_launch(const float* pe, const float* q_all, int64_t ldq, const float* v, const float* ds_all,
                             const float* alpha_all, const float* dc_all, int64_t B, int64_t Ts, int64_t Tt,
                             int64_t C, float* d_pe, float* dvp, float* d_enc, int accumulate_enc, hipStream_t s) {
    VAG_CHECK_ARG(pe && q_all && v && ds_all && alpha_all && d_pe && B > 0 && Ts > 0 && Tt > 0 && C > 0);
    dim3 grid((unsigned)cdiv64(C, 256), (unsigned)B, (unsigned)cdiv64(Ts, SC));
    hipLaunchKernelGGL(attn_post_bwd_kernel, grid, dim3(256), 0, s, pe, q_all, v, ds_all, alpha_all, dc_all, (int)B,
                       (int)Ts, (int)Tt, (int)C, ldq, d_pe, dvp, d_enc, accumulate_enc);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// ------------------------------------------------------------------ out[b,t,c] (+)= a1[b,t] x1[b,c] + a2[b,t] x2[b,c]
__global__ __launch_bounds__(256) void outer2_kernel(const float* __restrict__ a1, const float* __restrict__ x1,
                                                     const float* __restrict__ a2, const float* __restrict__ x2,
                                                     int Ts, int C, float* __restrict__ out, int acc) {
    const int b = blockIdx.y;
    const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (c >= C) return;
    const float4 u = *reinterpret_cast<const float4*>(x1 + (int64_t)b * C + c);
    const float4 w = a2 ? *reinterpret_cast<const float4*>(x2 + (int64_t)b * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < Ts; ++t) {
        const float p = a1[(int64_t)b * Ts + t];
        const float r = a2 ? a2[(int64_t)b * Ts + t] : 0.f;
        float* o = out + ((int64_t)b * Ts + t) * C + c;
        float4 v = make_float4(p * u.x + r * w.x, p * u.y + r * w.y, p * u.z + r * w.z, p * u.w + r * w.w);
        if (acc) {
            const float4 old = *reinterpret_cast<const float4*>(o);
            v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
        }
        *reinterpret_cast<float4*>(o) = v;
    }
}
int vag_outer2_launch(const float* a1, const float* x1, const float* a2, const float* x2, int64_t B, int64_t Ts,
                      int64_t C, float* out, int accumulate, hipStream_t s) {
    VAG_CHECK_ARG(a1 && x1 && out && B > 0 && Ts > 0 && C > 0 && C % 4 == 0);
    dim3 grid((unsigned)cdiv64(C, 1024), (unsigned)B);
    hipLaunchKernelGGL(outer2_kernel, grid, dim3(256), 0, s, a1, x1, a2, x2, (int)Ts, (int)C, out, accumulate);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}
