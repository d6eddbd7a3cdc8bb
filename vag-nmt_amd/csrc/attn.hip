// Attention kernels (Bahdanau MLP attention of the decoder, image-conditioned attention of the VSE module).
// HBM/L2-bound streaming over the (B,Ts,C) key (pe) and value (enc) tensors: one wave per (row, position)
// with 16-byte loads for the score pass, wavefront shuffle reductions, LDS softmax.
#include "kernels.h"

// ------------------------------------------------------------------ scores
template <int MODE>
__global__ __launch_bounds__(256) void attn_scores_kernel(const float* __restrict__ pe, const float* __restrict__ q,
                                                          const float* __restrict__ v, const float* __restrict__ mask,
                                                          int64_t total, int rps, int rmod, int Ts, int C, int64_t ldq,
                                                          const float* __restrict__ addend, float* __restrict__ scores) {
    // grid (ceil(Ts/4), N): no 64-bit divisions on the way to the first load (they cost more than the row's arithmetic)
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= Ts) return;
    const int64_t n = blockIdx.y;
    const int64_t pair = n * Ts + s;
    // source row of query row n: n / rps (beam search: rps hypotheses per sentence) or n % rmod (time-major (t,b) rows)
    const int64_t b = rmod > 0 ? (int64_t)((int)blockIdx.y % rmod) : (rps == 1 ? n : (int64_t)((int)blockIdx.y / rps));
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
        if (addend) acc += addend[pair];
        if (mask && mask[b * Ts + s] == 0.f) acc = -INFINITY;
        scores[pair] = acc;
    }
}

// rows_mod > 0: query row n reads source row n % rows_mod (rows ordered (t, b)); else n / rps.  addend (N,Ts) may be NULL.
int vag_attn_scores_ex_launch(int mode, const float* pe, const float* q, int64_t ldq, const float* v, const float* mask,
                              int64_t N, int64_t rps, int64_t rows_mod, int64_t Ts, int64_t C, const float* addend,
                              float* scores, hipStream_t s) {
    VAG_CHECK_ARG(pe && q && scores && N > 0 && Ts > 0 && C > 0 && C % 4 == 0 && rps >= 1 && ldq % 4 == 0 && ldq >= C);
    VAG_CHECK_ARG(mode == 1 || v);
    const int64_t total = N * Ts;
    VAG_CHECK_ARG(N < 65536 && rows_mod >= 0);
    dim3 grid((unsigned)cdiv64(Ts, 4), (unsigned)N);
    if (mode == 0)
        hipLaunchKernelGGL(attn_scores_kernel<0>, grid, dim3(256), 0, s, pe, q, v, mask, total, (int)rps, (int)rows_mod,
                           (int)Ts, (int)C, ldq, addend, scores);
    else
        hipLaunchKernelGGL(attn_scores_kernel<1>, grid, dim3(256), 0, s, pe, q, v, mask, total, (int)rps, (int)rows_mod,
                           (int)Ts, (int)C, ldq, addend, scores);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}
int vag_attn_scores_launch(int mode, const float* pe, const float* q, int64_t ldq, const float* v, const float* mask,
                           int64_t N, int64_t rps, int64_t Ts, int64_t C, float* scores, hipStream_t s) {
    return vag_attn_scores_ex_launch(mode, pe, q, ldq, v, mask, N, rps, 0, Ts, C, nullptr, scores, s);
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

// ------------------------------------------------------------------ softmax + projected context + GRU cell
// The decoder's second cell takes W_ih2 W_c2h c as its input projection, c = sum_s alpha_s enc_s.  With
// encwp[b,s,:] = (W_ih2 W_c2h) enc[b,s,:] computed once per batch, that projection is sum_s alpha_s encwp[b,s,:]: the
// weighted sum this kernel already does, over 3H instead of C columns -- and the cell has no product left, so it is
// this kernel's epilogue (the step loses a launch and the 6 MB weight re-read of the folded matrix).
// grid (ceil(H/256), N), CG_WAVES waves: lane owns 4 hidden units (three float4 gate columns per position), wave w walks
// the positions s = w, w + CG_WAVES, ...; partial sums meet in LDS, wave 0 finishes the cell.
// hp (N, ldhp) = W_hh2 h1 + b_hh2 (three gate blocks of H), save = [4][N][H] (r, z, n, hp_n) or NULL.
constexpr int CG_WAVES = 8;
template <bool XH>
__global__ __launch_bounds__(64 * CG_WAVES) void attn_ctx_gru_kernel(const float* __restrict__ scores,
                                                                     const float* __restrict__ encwp, int rps, int Ts, int H,
                                                                     const float* __restrict__ b_ih,
                                                                     const float* __restrict__ hp, int64_t ldhp,
                                                                     const float* __restrict__ hprev, int64_t N,
                                                                     float* __restrict__ alpha, float* __restrict__ hout,
                                                                     float* __restrict__ save, const float* __restrict__ x2,
                                                                     int W2, float* __restrict__ out2) {
    // x2 (B, Ts, W2) / out2 (N, W2), optional: a second weighted sum with the same attention weights, out2[n] = sum_s alpha_s x2[b,s,:]
    // (decoding: x2 = enc W2^T, the head's share of the context -- the context itself is then never formed), done by the blocks
    // blockIdx.x >= ceil(H / 256).
    extern __shared__ __attribute__((aligned(16))) float w[];   // Ts weights, CG_WAVES x 3 x 64 float4 partials
    float4* part = reinterpret_cast<float4*>(w + ((Ts + 3) & ~3));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n = blockIdx.y;
    const int64_t b = rps == 1 ? n : (int64_t)((int)blockIdx.y / rps);
    const float* sc = scores + n * Ts;
    const int hblocks = (H + 255) >> 8;
    if ((int)blockIdx.x >= hblocks) {
        const int c = (((int)blockIdx.x - hblocks) * 64 + lane) * 4;
        float mx = -INFINITY;
        for (int s = lane; s < Ts; s += 64) mx = fmaxf(mx, sc[s]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int s = lane; s < Ts; s += 64) sum += __expf(sc[s] - mx);
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        float4 a2 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < W2)
            for (int s = wave; s < Ts; s += CG_WAVES) {
                const float al = __expf(sc[s] - mx) * inv;
                const float4 v = *reinterpret_cast<const float4*>(x2 + (b * Ts + s) * (int64_t)W2 + c);
                a2.x += al * v.x; a2.y += al * v.y; a2.z += al * v.z; a2.w += al * v.w;
            }
        part[wave * 64 + lane] = a2;
        __syncthreads();
        if (wave != 0 || c >= W2) return;
        for (int q = 1; q < CG_WAVES; ++q) {
            const float4 o = part[q * 64 + lane];
            a2.x += o.x; a2.y += o.y; a2.z += o.z; a2.w += o.w;
        }
        *reinterpret_cast<float4*>(out2 + n * W2 + c) = a2;
        return;
    }
    const int u = (blockIdx.x * 64 + lane) * 4;
    const bool uok = u < H;
    // Everything that does not depend on the softmax is requested first -- the first U value rows of this wave and, for
    // wave 0, the cell's other operands -- so that the weights cost no extra round of memory latency.
    constexpr int U = 5;
    const int64_t e0 = b * Ts * 3 * H + (uok ? u : 0);       // element offset into encwp (fp32 or fp16)
    float4 ev[U][3];
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int s = min(wave + i * CG_WAVES, Ts - 1);
#pragma unroll
        for (int g = 0; g < 3; ++g) ev[i][g] = ld4_any<XH>(encwp, e0 + (int64_t)s * 3 * H + g * H);
    }
    float4 bb[3], hg[3], h1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (wave == 0 && uok) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            bb[g] = *reinterpret_cast<const float4*>(b_ih + g * H + u);
            hg[g] = *reinterpret_cast<const float4*>(hp + n * ldhp + g * H + u);
        }
        h1 = *reinterpret_cast<const float4*>(hprev + n * H + u);
    }
    {
        float mx = -INFINITY;
        for (int s = lane; s < Ts; s += 64) mx = fmaxf(mx, sc[s]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int s = lane; s < Ts; s += 64) sum += __expf(sc[s] - mx);
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        for (int s = threadIdx.x; s < Ts; s += 64 * CG_WAVES) {
            const float a = __expf(sc[s] - mx) * inv;
            w[s] = a;
            if (blockIdx.x == 0 && alpha) alpha[n * Ts + s] = a;
        }
    }
    __syncthreads();
    float4 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (uok) {
        for (int s0 = wave; s0 < Ts; s0 += U * CG_WAVES) {
            if (s0 != wave) {       // later rounds (Ts > U * CG_WAVES)
#pragma unroll
                for (int i = 0; i < U; ++i) {
                    const int s = min(s0 + i * CG_WAVES, Ts - 1);
#pragma unroll
                    for (int g = 0; g < 3; ++g) ev[i][g] = ld4_any<XH>(encwp, e0 + (int64_t)s * 3 * H + g * H);
                }
            }
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const int s = s0 + i * CG_WAVES;
                const float a = (s < Ts) ? w[s] : 0.f;
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    acc[g].x += a * ev[i][g].x; acc[g].y += a * ev[i][g].y; acc[g].z += a * ev[i][g].z; acc[g].w += a * ev[i][g].w;
                }
            }
        }
    }
#pragma unroll
    for (int g = 0; g < 3; ++g) part[(wave * 3 + g) * 64 + lane] = acc[g];
    __syncthreads();
    if (wave != 0 || !uok) return;
    float gi[3][4];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        float4 t = acc[g];
#pragma unroll
        for (int k = 1; k < CG_WAVES; ++k) {
            const float4 o = part[(k * 3 + g) * 64 + lane];
            t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
        }
        gi[g][0] = t.x + bb[g].x; gi[g][1] = t.y + bb[g].y; gi[g][2] = t.z + bb[g].z; gi[g][3] = t.w + bb[g].w;
    }
    const float4 hr = hg[0], hz = hg[1], hn = hg[2];
    const float ghr[4] = {hr.x, hr.y, hr.z, hr.w}, ghz[4] = {hz.x, hz.y, hz.z, hz.w}, ghn[4] = {hn.x, hn.y, hn.z, hn.w};
    const float hpv[4] = {h1.x, h1.y, h1.z, h1.w};
    float rr[4], zz[4], nn[4], ho[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        rr[i] = vag_sigmoid(gi[0][i] + ghr[i]);
        zz[i] = vag_sigmoid(gi[1][i] + ghz[i]);
        nn[i] = vag_tanh(gi[2][i] + rr[i] * ghn[i]);
        ho[i] = (1.f - zz[i]) * nn[i] + zz[i] * hpv[i];
    }
    const int64_t o = n * H + u;
    *reinterpret_cast<float4*>(hout + o) = make_float4(ho[0], ho[1], ho[2], ho[3]);
    if (save) {
        const int64_t NH = N * H;
        *reinterpret_cast<float4*>(save + o) = make_float4(rr[0], rr[1], rr[2], rr[3]);
        *reinterpret_cast<float4*>(save + NH + o) = make_float4(zz[0], zz[1], zz[2], zz[3]);
        *reinterpret_cast<float4*>(save + 2 * NH + o) = make_float4(nn[0], nn[1], nn[2], nn[3]);
        *reinterpret_cast<float4*>(save + 3 * NH + o) = hn;
    }
}
int vag_attn_ctx_gru_launch(const float* scores, const float* encwp, int64_t N, int64_t rps, int64_t Ts, int64_t H,
                            const float* b_ih, const float* hp, int64_t ldhp, const float* hprev, float* alpha, float* hout,
                            float* save, hipStream_t s, bool x16, const float* x2, int64_t W2, float* out2) {
    VAG_CHECK_ARG(!x2 || (out2 && W2 > 0 && W2 % 4 == 0 && aligned16(x2) && aligned16(out2)));
    VAG_CHECK_ARG(scores && encwp && b_ih && hp && hprev && hout && N > 0 && N < 65536 && Ts > 0 && H > 0 && H % 4 == 0);
    VAG_CHECK_ARG(ldhp % 4 == 0 && rps >= 1 && aligned16(encwp) && aligned16(hp) && aligned16(hprev) && aligned16(hout) &&
                  aligned16(b_ih) && (!save || aligned16(save)));
    dim3 grid((unsigned)(cdiv64(H, 256) + (x2 ? cdiv64(W2, 256) : 0)), (unsigned)N);
    const size_t lds = (size_t)((Ts + 3) & ~3) * sizeof(float) + (size_t)CG_WAVES * 3 * 64 * 16;
    if (x16)
        hipLaunchKernelGGL(attn_ctx_gru_kernel<true>, grid, dim3(64 * CG_WAVES), lds, s, scores, encwp, (int)rps, (int)Ts,
                           (int)H, b_ih, hp, ldhp, hprev, N, alpha, hout, save, x2, (int)W2, out2);
    else
        hipLaunchKernelGGL(attn_ctx_gru_kernel<false>, grid, dim3(64 * CG_WAVES), lds, s, scores, encwp, (int)rps, (int)Ts,
                           (int)H, b_ih, hp, ldhp, hprev, N, alpha, hout, save, x2, (int)W2, out2);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// ------------------------------------------------------------------ a decoding step's attention and second cell in ONE launch
// (beam search / step-wise decoding at N = B k hypothesis rows: NMT_Decoder.py:47-51, :41-44, :126-129 and the head's W2 c).
// attn_scores_kernel + attn_ctx_gru_kernel were 5.9 + 9.8 us per step at 192 rows, two launches of mostly launch and ramp.  Here one
// 512-thread workgroup per hypothesis row n (source sentence b = n / rps): wave w takes the positions s = w, w + 8, ... in BOTH
// passes -- (1) e[s] = v . tanh(pe[b,s,:] + q[n,:]) with its four 16-byte loads per lane and position all in flight, (2) softmax in
// every wave, (3) the weighted sums over the projected keys (3H gate columns) and over enc W2^T (E columns): NG + NW float4 per lane
// and position, the first two positions requested before pass 1 starts -- the eight waves' partial sums meet in LDS, then the
// cell (threads 0 .. H/4 - 1) and the head's share (the next NW x 64 threads).  C = 2H = 1024, 3H = 256 NG, E = 256 NW.
template <int NG, int NW>
__global__ __launch_bounds__(512) void attn_row_gru_kernel(const float* __restrict__ pe, const float* __restrict__ q, int64_t ldq,
                                                           const float* __restrict__ v, const float* __restrict__ mask,
                                                           const float* __restrict__ keys, const float* __restrict__ x2, int rps,
                                                           int Ts, const float* __restrict__ b_ih, const float* __restrict__ hp,
                                                           int64_t ldhp, const float* __restrict__ hprev,
                                                           float* __restrict__ alpha, float* __restrict__ hout,
                                                           float* __restrict__ out2) {
    constexpr int C = 1024, H3 = 256 * NG, H = H3 / 3, W2 = 256 * NW, NA = NG + NW, PRE = 2;
    extern __shared__ __attribute__((aligned(16))) float rsm[];      // Ts_pad scores -> weights, then part[8][NA][64] float4
    const int Tp = (Ts + 3) & ~3;
    float4* part = reinterpret_cast<float4*>(rsm + Tp);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n = blockIdx.x, b = rps == 1 ? n : (int64_t)((int)blockIdx.x / rps);
    const float* kb = keys + b * Ts * (int64_t)H3 + 4 * lane;
    const float* xb = x2 + b * Ts * (int64_t)W2 + 4 * lane;
    auto load_keys = [&](int s, float4 (&k)[NA]) {
#pragma unroll
        for (int j = 0; j < NG; ++j) k[j] = *reinterpret_cast<const float4*>(kb + (int64_t)s * H3 + 256 * j);
#pragma unroll
        for (int j = 0; j < NW; ++j) k[NG + j] = *reinterpret_cast<const float4*>(xb + (int64_t)s * W2 + 256 * j);
    };
    float4 kpre[PRE][NA];
#pragma unroll
    for (int p = 0; p < PRE; ++p) load_keys(min(wave + 8 * p, Ts - 1), kpre[p]);
    // the cell's other operands (threads 0 .. H/4 - 1): requested now, used after pass 3
    float4 bb[3], hg[3], h1 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool cellt = threadIdx.x < H / 4;
    if (cellt) {
        const int u = 4 * threadIdx.x;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            bb[g] = *reinterpret_cast<const float4*>(b_ih + g * H + u);
            hg[g] = *reinterpret_cast<const float4*>(hp + n * ldhp + g * H + u);
        }
        h1 = *reinterpret_cast<const float4*>(hprev + n * H + u);
    }
    // ---- pass 1: scores
    {
        float4 qv[4], vv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            qv[j] = *reinterpret_cast<const float4*>(q + n * ldq + 4 * lane + 256 * j);
            vv[j] = *reinterpret_cast<const float4*>(v + 4 * lane + 256 * j);
        }
        const float* pb = pe + b * Ts * (int64_t)C + 4 * lane;
        for (int s0 = wave; s0 < Ts; s0 += 32) {
            float4 x[4][4];
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int j = 0; j < 4; ++j) x[p][j] = *reinterpret_cast<const float4*>(pb + (int64_t)min(s0 + 8 * p, Ts - 1) * C + 256 * j);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int sp = s0 + 8 * p;
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc += vv[j].x * vag_tanh(x[p][j].x + qv[j].x) + vv[j].y * vag_tanh(x[p][j].y + qv[j].y);
                    acc += vv[j].z * vag_tanh(x[p][j].z + qv[j].z) + vv[j].w * vag_tanh(x[p][j].w + qv[j].w);
                }
                acc = wave_sum(acc);
                if (lane == 0 && sp < Ts) rsm[sp] = (mask && mask[b * Ts + sp] == 0.f) ? -INFINITY : acc;
            }
        }
    }
    __syncthreads();
    // ---- pass 2: softmax (every wave for itself), weights to LDS
    {
        float mx = -INFINITY;
        for (int s = lane; s < Ts; s += 64) mx = fmaxf(mx, rsm[s]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int s = lane; s < Ts; s += 64) sum += __expf(rsm[s] - mx);
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        __syncthreads();
        for (int s = threadIdx.x; s < Ts; s += 512) {
            const float a = __expf(rsm[s] - mx) * inv;
            rsm[s] = a;
            if (alpha) alpha[n * Ts + s] = a;
        }
    }
    __syncthreads();
    // ---- pass 3: weighted sums of the wave's positions
    float4 acc[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto add = [&](int s, const float4 (&k)[NA]) {
        const float a = s < Ts ? rsm[s] : 0.f;
#pragma unroll
        for (int j = 0; j < NA; ++j) { acc[j].x += a * k[j].x; acc[j].y += a * k[j].y; acc[j].z += a * k[j].z; acc[j].w += a * k[j].w; }
    };
#pragma unroll
    for (int p = 0; p < PRE; ++p) add(wave + 8 * p, kpre[p]);
    for (int s0 = wave + 8 * PRE; s0 < Ts; s0 += 8 * PRE) {
        float4 k[PRE][NA];
#pragma unroll
        for (int p = 0; p < PRE; ++p) load_keys(min(s0 + 8 * p, Ts - 1), k[p]);
#pragma unroll
        for (int p = 0; p < PRE; ++p) add(s0 + 8 * p, k[p]);
    }
#pragma unroll
    for (int j = 0; j < NA; ++j) part[(wave * NA + j) * 64 + lane] = acc[j];
    __syncthreads();
    // column quad c4 of [gi2 (3H) | W2 c (E)] lives at part[w][c4 / 64][c4 % 64]
    auto colsum = [&](int c4) {
        float4 t = part[(c4 >> 6) * 64 + (c4 & 63)];
#pragma unroll
        for (int w = 1; w < 8; ++w) {
            const float4 o = part[(w * NA + (c4 >> 6)) * 64 + (c4 & 63)];
            t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
        }
        return t;
    };
    if (cellt) {
        const int u4 = threadIdx.x;
        float gi[3][4];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const float4 t = colsum(g * (H / 4) + u4);
            gi[g][0] = t.x + bb[g].x; gi[g][1] = t.y + bb[g].y; gi[g][2] = t.z + bb[g].z; gi[g][3] = t.w + bb[g].w;
        }
        const float ghr[4] = {hg[0].x, hg[0].y, hg[0].z, hg[0].w}, ghz[4] = {hg[1].x, hg[1].y, hg[1].z, hg[1].w};
        const float ghn[4] = {hg[2].x, hg[2].y, hg[2].z, hg[2].w}, hpv[4] = {h1.x, h1.y, h1.z, h1.w};
        float ho[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float rr = vag_sigmoid(gi[0][i] + ghr[i]);
            const float zz = vag_sigmoid(gi[1][i] + ghz[i]);
            const float nn = vag_tanh(gi[2][i] + rr * ghn[i]);
            ho[i] = (1.f - zz) * nn + zz * hpv[i];
        }
        *reinterpret_cast<float4*>(hout + n * H + 4 * u4) = make_float4(ho[0], ho[1], ho[2], ho[3]);
    } else if ((int)threadIdx.x < H / 4 + W2 / 4) {
        const int e4 = threadIdx.x - H / 4;
        *reinterpret_cast<float4*>(out2 + n * W2 + 4 * e4) = colsum(64 * NG + e4);
    }
}
// H = 512, E = 256 (configs[1]'s decoder): NG = 6, NW = 1, and source lengths up to 32: one workgroup streams the whole row's keys
// (11 KB per position) through ONE CU, ~50 GB/s -- measured at 192 rows, us per beam step, fused | two launches (tools/exp_row_gru.py):
// Ts 12: 87.3 | 89.3, 16: 86.9 | 89.3, 24: 88.0 | 90.6, 32: 89.4 | 91.4, 40: 94.6 | 93.7.  Other shapes: the two launches.
bool vag_attn_row_gru_ok(int64_t N, int64_t Ts, int64_t H, int64_t W2) {
    return vag_opt().attn_row != 0 && H == 512 && W2 == 256 && N > 0 && N < (1ll << 30) && Ts > 0 && Ts <= 32;
}
int vag_attn_row_gru_launch(const float* pe, const float* q, int64_t ldq, const float* v, const float* mask, const float* keys,
                            const float* x2, int64_t N, int64_t rps, int64_t Ts, int64_t H, int64_t W2, const float* b_ih,
                            const float* hp, int64_t ldhp, const float* hprev, float* alpha, float* hout, float* out2, hipStream_t s) {
    VAG_CHECK_ARG(vag_attn_row_gru_ok(N, Ts, H, W2) && pe && q && v && keys && x2 && b_ih && hp && hprev && hout && out2 && rps >= 1 &&
                  ldq % 4 == 0 && ldhp % 4 == 0 && aligned16(pe) && aligned16(q) && aligned16(v) && aligned16(keys) && aligned16(x2) &&
                  aligned16(b_ih) && aligned16(hp) && aligned16(hprev) && aligned16(hout) && aligned16(out2));
    const size_t lds = (size_t)((Ts + 3) & ~3) * sizeof(float) + (size_t)8 * 7 * 64 * 16;
    hipLaunchKernelGGL((attn_row_gru_kernel<6, 1>), dim3((unsigned)N), dim3(512), lds, s, pe, q, ldq, v, mask, keys, x2, (int)rps, (int)Ts,
                       b_ih, hp, ldhp, hprev, alpha, hout, out2);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// ------------------------------------------------------------------ batched over time: out[t,b,:] = sum_s a[t,b,s] x[b,s,:]
// (the contexts of all steps after the loop) and out[b,s,:] = sum_t a[t,b,s] y[t,b,:] (gradient of the projected keys).
// grid (ceil(W/256), B, ceil(T/8)) / (ceil(W/256), B, ceil(Ts/8)); thread owns one float4 column, 8 outputs in registers.
// Round 3: 256-thread blocks, wave w takes every 4th position of the summed dimension and the four partial sums meet in LDS (one
// wave per block walked all Ts / T positions on its own: ~5 waves per CU, a serial chain of 40 loads: 17-20 us each).
__device__ __forceinline__ void attn_wsum_time_body(float4 (&part)[3][8][64], int bz, const float* __restrict__ a,
                                                    const float* __restrict__ x, int B, int Ts, int T, int W, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y, t0 = bz * 8;
    const int c = (blockIdx.x * 64 + lane) * 4;
    const bool cok = c < W;
    float4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cok) {
        const float* xr = x + (int64_t)b * Ts * W + c;
        for (int s = wave; s < Ts; s += 4) {
            const float4 v = *reinterpret_cast<const float4*>(xr + (int64_t)s * W);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int t = min(t0 + i, T - 1);
                const float al = a[((int64_t)t * B + b) * Ts + s];
                acc[i].x += al * v.x; acc[i].y += al * v.y; acc[i].z += al * v.z; acc[i].w += al * v.w;
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) part[wave - 1][i][lane] = acc[i];
    }
    __syncthreads();
    if (wave == 0 && cok) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float4 r = acc[i];
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                const float4 o = part[w][i][lane];
                r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
            }
            if (t0 + i < T) *reinterpret_cast<float4*>(out + ((int64_t)(t0 + i) * B + b) * W + c) = r;
        }
    }
}
__global__ __launch_bounds__(256) void attn_wsum_time_kernel(const float* __restrict__ a, const float* __restrict__ x, int B,
                                                             int Ts, int T, int W, float* __restrict__ out) {
    __shared__ float4 part[3][8][64];
    attn_wsum_time_body(part, blockIdx.z, a, x, B, Ts, T, W, out);
}
__device__ __forceinline__ void attn_wsum_src_body(float4 (&part)[3][8][64], int bz, const float* __restrict__ a,
                                                   const float* __restrict__ y, int B, int Ts, int T, int W, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y, s0 = bz * 8;
    const int c = (blockIdx.x * 64 + lane) * 4;
    const bool cok = c < W;
    float4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cok) {
        for (int t = wave; t < T; t += 4) {
            const float4 v = *reinterpret_cast<const float4*>(y + ((int64_t)t * B + b) * W + c);
            const float* ar = a + ((int64_t)t * B + b) * Ts;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float al = ar[min(s0 + i, Ts - 1)];
                acc[i].x += al * v.x; acc[i].y += al * v.y; acc[i].z += al * v.z; acc[i].w += al * v.w;
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) part[wave - 1][i][lane] = acc[i];
    }
    __syncthreads();
    if (wave == 0 && cok) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float4 r = acc[i];
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                const float4 o = part[w][i][lane];
                r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
            }
            if (s0 + i < Ts) *reinterpret_cast<float4*>(out + ((int64_t)b * Ts + s0 + i) * W + c) = r;
        }
    }
}
__global__ __launch_bounds__(256) void attn_wsum_src_kernel(const float* __restrict__ a, const float* __restrict__ y, int B,
                                                            int Ts, int T, int W, float* __restrict__ out) {
    __shared__ float4 part[3][8][64];
    attn_wsum_src_body(part, blockIdx.z, a, y, B, Ts, T, W, out);
}
// Both sums over one weight tensor a (T,B,Ts) in one grid: blocks z < ceil(Ts/8): out_src (B,Ts,Wy) = sum_t a y;  the rest:
// out_time (T,B,Wx) = sum_s a x  (the decoder's backward needs one of each and they share nothing but a)
__global__ __launch_bounds__(256) void attn_wsum_pair_kernel(const float* __restrict__ a, const float* __restrict__ y, int Wy,
                                                             float* __restrict__ out_src, const float* __restrict__ x, int Wx,
                                                             float* __restrict__ out_time, int B, int Ts, int T, int nz_src) {
    __shared__ float4 part[3][8][64];
    if ((int)blockIdx.z < nz_src) attn_wsum_src_body(part, blockIdx.z, a, y, B, Ts, T, Wy, out_src);
    else attn_wsum_time_body(part, blockIdx.z - nz_src, a, x, B, Ts, T, Wx, out_time);
}
int vag_attn_wsum_pair_launch(const float* a, const float* y, int64_t Wy, float* out_src, const float* x, int64_t Wx, float* out_time,
                              int64_t B, int64_t Ts, int64_t T, hipStream_t s) {
    VAG_CHECK_ARG(a && y && x && out_src && out_time && B > 0 && Ts > 0 && T > 0 && Wy > 0 && Wx > 0 && Wy % 4 == 0 && Wx % 4 == 0 &&
                  aligned16(x) && aligned16(y) && aligned16(out_src) && aligned16(out_time));
    const int64_t W = Wy > Wx ? Wy : Wx;
    const int nz_src = (int)cdiv64(Ts, 8);
    hipLaunchKernelGGL(attn_wsum_pair_kernel, dim3((unsigned)cdiv64(W, 256), (unsigned)B, (unsigned)(nz_src + cdiv64(T, 8))), dim3(256), 0, s,
                       a, y, (int)Wy, out_src, x, (int)Wx, out_time, (int)B, (int)Ts, (int)T, nz_src);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}
int vag_attn_wsum_launch(int over_src, const float* a, const float* x, int64_t B, int64_t Ts, int64_t T, int64_t W, float* out,
                         hipStream_t s) {
    VAG_CHECK_ARG(a && x && out && B > 0 && Ts > 0 && T > 0 && W > 0 && W % 4 == 0 && aligned16(x) && aligned16(out));
    if (over_src) {     // out (T,B,W) = sum_s a x
        hipLaunchKernelGGL(attn_wsum_time_kernel, dim3((unsigned)cdiv64(W, 256), (unsigned)B, (unsigned)cdiv64(T, 8)), dim3(256), 0,
                           s, a, x, (int)B, (int)Ts, (int)T, (int)W, out);
    } else {            // out (B,Ts,W) = sum_t a y
        hipLaunchKernelGGL(attn_wsum_src_kernel, dim3((unsigned)cdiv64(W, 256), (unsigned)B, (unsigned)cdiv64(Ts, 8)), dim3(256), 0,
                           s, a, x, (int)B, (int)Ts, (int)T, (int)W, out);
    }
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

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

// ------------------------------------------------------------------ one row of a dot-product attention in ONE launch
// (visual grounding, VSE_Imagine_Enc.py:57-64,46,137 and its backward: once per batch, B rows -- the three launches it replaces
// were ~5 us each whatever they did).  One 1024-thread workgroup per row b:
//   forward  (BWD = false): e[t] = x[b,t,:] . q[b,:], -inf where mask == 0;  wt = softmax(e) -> wout;  sum[b,:] = sum_t wt[t] x[b,t,:]
//   backward (BWD = true):  d[t] = x[b,t,:] . q[b,:];  wt[t] = alpha[t] (d[t] - sum alpha d) -> wout;  sum[b,:] likewise (or none: sum NULL)
// Phase 1: wave w takes positions w, w + 16, ...; phase 3: thread = (column quad, position group), groups meet in LDS.
constexpr int ROW_THREADS = 1024;
template <bool BWD>
__global__ __launch_bounds__(ROW_THREADS) void attn_dot_row_kernel(const float* __restrict__ x, const float* __restrict__ q, int64_t ldq,
                                                                    const float* __restrict__ mask, const float* __restrict__ alpha,
                                                                    int Ts, int C, float* __restrict__ wout, float* __restrict__ sum) {
    extern __shared__ __attribute__((aligned(16))) float rw[];       // Ts_pad weights, then the position groups' partial sums
    const int Tp = (Ts + 3) & ~3;
    float4* part = reinterpret_cast<float4*>(rw + Tp);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t b = blockIdx.x;
    const float* xb = x + b * Ts * (int64_t)C;
    const float* qr = q + b * ldq;
    for (int t = wave; t < Ts; t += ROW_THREADS / 64) {
        const float* xr = xb + (int64_t)t * C;
        float acc = 0.f;
        for (int c = lane * 4; c < C; c += 256) {
            const float4 a = *reinterpret_cast<const float4*>(xr + c), v = *reinterpret_cast<const float4*>(qr + c);
            acc += a.x * v.x + a.y * v.y + a.z * v.z + a.w * v.w;
        }
        acc = wave_sum(acc);
        if (lane == 0) rw[t] = (!BWD && mask && mask[b * Ts + t] == 0.f) ? -INFINITY : acc;
    }
    __syncthreads();
    {   // the Ts-element reduction, by every wave for itself (it is tiny)
        if (!BWD) {
            float mx = -INFINITY;
            for (int t = lane; t < Ts; t += 64) mx = fmaxf(mx, rw[t]);
            mx = wave_max(mx);
            float sm = 0.f;
            for (int t = lane; t < Ts; t += 64) sm += __expf(rw[t] - mx);
            sm = wave_sum(sm);
            const float inv = 1.f / sm;
            __syncthreads();
            for (int t = threadIdx.x; t < Ts; t += ROW_THREADS) {
                const float a = __expf(rw[t] - mx) * inv;
                rw[t] = a;
                wout[b * Ts + t] = a;
            }
        } else {
            float dot = 0.f;
            for (int t = lane; t < Ts; t += 64) dot += alpha[b * Ts + t] * rw[t];
            dot = wave_sum(dot);
            __syncthreads();
            for (int t = threadIdx.x; t < Ts; t += ROW_THREADS) {
                const float a = alpha[b * Ts + t] * (rw[t] - dot);
                rw[t] = a;
                wout[b * Ts + t] = a;
            }
        }
    }
    __syncthreads();
    if (!sum) return;
    const int nc4 = C >> 2;
    const int G = nc4 >= ROW_THREADS ? 1 : ROW_THREADS / nc4;          // position groups
    if (G == 1) {
        for (int c4 = threadIdx.x; c4 < nc4; c4 += ROW_THREADS) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int t = 0; t < Ts; ++t) {
                const float a = rw[t];
                const float4 e = *reinterpret_cast<const float4*>(xb + (int64_t)t * C + 4 * c4);
                acc.x += a * e.x; acc.y += a * e.y; acc.z += a * e.z; acc.w += a * e.w;
            }
            *reinterpret_cast<float4*>(sum + b * C + 4 * c4) = acc;
        }
        return;
    }
    const int grp = threadIdx.x / nc4, c4 = threadIdx.x - grp * nc4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (grp < G) {
        constexpr int U = 5;
        const float* e = xb + 4 * c4;
        for (int t0 = grp; t0 < Ts; t0 += U * G) {
            float4 ev[U];
#pragma unroll
            for (int i = 0; i < U; ++i) ev[i] = *reinterpret_cast<const float4*>(e + (int64_t)min(t0 + i * G, Ts - 1) * C);
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const int t = t0 + i * G;
                const float a = t < Ts ? rw[t] : 0.f;
                acc.x += a * ev[i].x; acc.y += a * ev[i].y; acc.z += a * ev[i].z; acc.w += a * ev[i].w;
            }
        }
        part[grp * nc4 + c4] = acc;
    }
    __syncthreads();
    if (grp == 0) {
        for (int g = 1; g < G; ++g) {
            const float4 o = part[g * nc4 + c4];
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
        }
        *reinterpret_cast<float4*>(sum + b * C + 4 * c4) = acc;
    }
}
// The same with the row of x held in REGISTERS between the phases (C = 256 NC, Ts <= 16 NP, NC * NP <= 16 float4 per thread): x is
// read from memory once (the kernel above reads it twice, ~50 GB/s per CU: 11.7 us for 2 x 160 KB at configs[1] -- no faster than the
// launches it replaced).  Thread (wave w, lane l) holds x[b, w + 16 p, 4 l + 256 j .. + 3].  The 16 waves' partial sums meet in a
// four-round tree through LDS (fixed order).  MIX (forward only): also xmix[b,:] = split * sum[b,:] + (1 - split) * mean_t x[b,t,:]
// (V11.py:118: the initial state's input; mean over the sentence's own length, cnt = #(mask[b,:])) -- the mean-pool launch saved.
template <bool BWD, int NC, int NP, bool MIX>
__global__ __launch_bounds__(ROW_THREADS) void attn_dot_row_reg_kernel(const float* __restrict__ x, const float* __restrict__ q, int64_t ldq,
                                                                        const float* __restrict__ mask, const float* __restrict__ alpha,
                                                                        int Ts, float* __restrict__ wout, float* __restrict__ sum,
                                                                        float* __restrict__ xmix, float split) {
    constexpr int C = 256 * NC, NACC = NC * (MIX ? 2 : 1);
    extern __shared__ __attribute__((aligned(16))) float rw[];       // 64 weights (+ cnt at [64]), then part[8][NACC][64] float4
    float4* part = reinterpret_cast<float4*>(rw + 80);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t b = blockIdx.x;
    const float* xb = x + b * Ts * (int64_t)C + 4 * lane;
    float4 xr[NP][NC], qv[NC];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int t = wave + 16 * p;
            xr[p][j] = t < Ts ? *reinterpret_cast<const float4*>(xb + (int64_t)t * C + 256 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
    for (int j = 0; j < NC; ++j) qv[j] = *reinterpret_cast<const float4*>(q + b * ldq + 4 * lane + 256 * j);
    if (MIX && threadIdx.x < 64) {
        float cnt = 0.f;
        for (int t = lane; t < Ts; t += 64) cnt += mask[b * Ts + t];
        cnt = wave_sum(cnt);
        if (lane == 0) rw[64] = cnt;
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < NC; ++j) acc += xr[p][j].x * qv[j].x + xr[p][j].y * qv[j].y + xr[p][j].z * qv[j].z + xr[p][j].w * qv[j].w;
        acc = wave_sum(acc);
        const int t = wave + 16 * p;
        if (lane == 0 && t < Ts) rw[t] = (!BWD && mask && mask[b * Ts + t] == 0.f) ? -INFINITY : acc;
    }
    __syncthreads();
    float wt[NP];
    {
        const float e = lane < Ts ? rw[lane] : (BWD ? 0.f : -INFINITY);          // Ts <= 64: one value per lane, every wave for itself
        float a;
        if (!BWD) {
            const float mx = wave_max(e);
            const float ex = lane < Ts ? __expf(e - mx) : 0.f;
            a = ex / wave_sum(ex);
        } else {
            const float al = lane < Ts ? alpha[b * Ts + lane] : 0.f;
            const float dot = wave_sum(al * e);
            a = al * (e - dot);
        }
        if (wave == 0 && lane < Ts) wout[b * Ts + lane] = a;
#pragma unroll
        for (int p = 0; p < NP; ++p) wt[p] = __shfl(a, min(wave + 16 * p, 63), 64);      // (positions past Ts hold zero rows)
    }
    if (!sum) return;
    float4 acc[NACC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f), m = v;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            v.x += wt[p] * xr[p][j].x; v.y += wt[p] * xr[p][j].y; v.z += wt[p] * xr[p][j].z; v.w += wt[p] * xr[p][j].w;
            if (MIX) { m.x += xr[p][j].x; m.y += xr[p][j].y; m.z += xr[p][j].z; m.w += xr[p][j].w; }
        }
        acc[j] = v;
        if (MIX) acc[NC + j] = m;
    }
#pragma unroll
    for (int half = 8; half >= 1; half >>= 1) {
        if (wave >= half && wave < 2 * half) {
#pragma unroll
            for (int j = 0; j < NACC; ++j) part[((wave - half) * NACC + j) * 64 + lane] = acc[j];
        }
        __syncthreads();
        if (wave < half) {
#pragma unroll
            for (int j = 0; j < NACC; ++j) {
                const float4 o = part[(wave * NACC + j) * 64 + lane];
                acc[j].x += o.x; acc[j].y += o.y; acc[j].z += o.z; acc[j].w += o.w;
            }
        }
        if (half > 1) __syncthreads();
    }
    if (wave == 0) {
        const float cnt = MIX ? rw[64] : 1.f;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            *reinterpret_cast<float4*>(sum + b * C + 4 * lane + 256 * j) = acc[j];
            if (MIX) {
                const float4 m = acc[NC + j];
                const float k = (1.f - split) / cnt;
                *reinterpret_cast<float4*>(xmix + b * C + 4 * lane + 256 * j) =
                    make_float4(split * acc[j].x + k * m.x, split * acc[j].y + k * m.y, split * acc[j].z + k * m.z, split * acc[j].w + k * m.w);
            }
        }
    }
}
// A rider for the next forward row launch of the calling thread: it also leaves xmix (B,C) = split * context + (1 - split) * mean-pool
// (what vag_dec_init_fwd would compute from that context with a launch of its own).  vag_attn_row_mix_done(xmix) tells whether it did.
struct RowMix { float* xmix = nullptr; float split = 0.f; };
static thread_local RowMix g_row_mix;
static thread_local const float* g_row_mix_done = nullptr;
void vag_attn_row_mix_request(float* xmix, float split) { g_row_mix = RowMix{xmix, split}; g_row_mix_done = nullptr; }
void vag_attn_row_mix_cancel() { g_row_mix = RowMix(); }
bool vag_attn_row_mix_done(const float* xmix) { const bool d = xmix && g_row_mix_done == xmix; g_row_mix_done = nullptr; return d; }

template <bool BWD, int NC, int NP, bool MIX>
static int row_reg_go(const float* x, const float* q, int64_t ldq, const float* mask, const float* alpha, int64_t B, int64_t Ts,
                      float* wout, float* sum, float* xmix, float split, hipStream_t s) {
    constexpr size_t lds = 80 * sizeof(float) + (size_t)8 * NC * (MIX ? 2 : 1) * 64 * 16;
    static bool attr = false;
    if (!attr && lds > 65536) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_dot_row_reg_kernel<BWD, NC, NP, MIX>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VAG_EINVAL;
        attr = true;
    }
    hipLaunchKernelGGL((attn_dot_row_reg_kernel<BWD, NC, NP, MIX>), dim3((unsigned)B), dim3(ROW_THREADS), lds, s, x, q, ldq, mask, alpha,
                       (int)Ts, wout, sum, xmix, split);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}
// forward: wout = alpha (B,Ts), sum = context (B,C).  backward: alpha given, wout = d(scores) (B,Ts), sum = sum_t wout x or NULL.
int vag_attn_dot_row_launch(bool bwd, const float* x, const float* q, int64_t ldq, const float* mask, const float* alpha, int64_t B,
                            int64_t Ts, int64_t C, float* wout, float* sum, hipStream_t s) {
    VAG_CHECK_ARG(x && q && wout && B > 0 && Ts > 0 && Ts <= 4096 && C > 0 && C % 4 == 0 && ldq % 4 == 0 && ldq >= C && (!bwd || alpha));
    const RowMix mix = g_row_mix;
    g_row_mix = RowMix();
    const int np = (int)cdiv64(Ts, 16);
    if (aligned16(x) && aligned16(q) && (!sum || aligned16(sum)) && np <= 4 && (C == 1024 || C == 512)) {
        const bool mx = !bwd && mix.xmix && mask && sum && aligned16(mix.xmix);
#define VAG_ROW(NC_, NP_)                                                                                                            \
        { if (bwd) return row_reg_go<true, NC_, NP_, false>(x, q, ldq, mask, alpha, B, Ts, wout, sum, nullptr, 0.f, s);               \
          if (!mx) return row_reg_go<false, NC_, NP_, false>(x, q, ldq, mask, alpha, B, Ts, wout, sum, nullptr, 0.f, s);              \
          VAG_TRY((row_reg_go<false, NC_, NP_, true>(x, q, ldq, mask, alpha, B, Ts, wout, sum, mix.xmix, mix.split, s)));             \
          g_row_mix_done = mix.xmix; return VAG_OK; }
        if (C == 1024) { if (np == 1) VAG_ROW(4, 1) else if (np == 2) VAG_ROW(4, 2) else if (np == 3) VAG_ROW(4, 3) else VAG_ROW(4, 4) }
        else { if (np == 1) VAG_ROW(2, 1) else if (np == 2) VAG_ROW(2, 2) else if (np == 3) VAG_ROW(2, 3) else VAG_ROW(2, 4) }
#undef VAG_ROW
    }
    const size_t lds = (size_t)((Ts + 3) & ~3) * sizeof(float) + (size_t)ROW_THREADS * 16;
    if (bwd) hipLaunchKernelGGL(attn_dot_row_kernel<true>, dim3((unsigned)B), dim3(ROW_THREADS), lds, s, x, q, ldq, mask, alpha, (int)Ts, (int)C, wout, sum);
    else hipLaunchKernelGGL(attn_dot_row_kernel<false>, dim3((unsigned)B), dim3(ROW_THREADS), lds, s, x, q, ldq, mask, alpha, (int)Ts, (int)C, wout, sum);
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
template <bool XH>
__global__ __launch_bounds__(64 * DQ_WAVES) void attn_dq_kernel(const float* __restrict__ pe, const float* __restrict__ q,
                                                     int64_t ldq, const float* __restrict__ v,
                                                     const float* __restrict__ alpha, const float* __restrict__ dalpha,
                                                     float* __restrict__ dscore, int Ts, int C, float* __restrict__ dq,
                                                     int64_t lddq) {
    extern __shared__ __attribute__((aligned(16))) float w[];      // Ts weights, then DQ_WAVES x 64 float4 partials
    float4* part = reinterpret_cast<float4*>(w + ((Ts + 3) & ~3));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n = blockIdx.y;
    // the first round of key rows does not depend on the prologue: request it first
    constexpr int DQ_U = 10;
    const int c = (blockIdx.x * 64 + lane) * 4;
    const bool cok = c < C;
    const int64_t p0 = n * Ts * C + (cok ? c : 0);              // element offset into pe (fp32 or fp16)
    float4 pv[DQ_U];
#pragma unroll
    for (int i = 0; i < DQ_U; ++i) pv[i] = ld4_any<XH>(pe, p0 + (int64_t)min(wave + i * DQ_WAVES, Ts - 1) * C);
    const float4 qv = cok ? *reinterpret_cast<const float4*>(q + n * ldq + c) : make_float4(0.f, 0.f, 0.f, 0.f);
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
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cok) {
        // DQ_U key rows in flight per lane: a wave's whole share of a 40-position source in one round of requests
        for (int s0 = wave; s0 < Ts; s0 += DQ_U * DQ_WAVES) {
            if (s0 != wave) {       // later rounds (Ts > DQ_U * DQ_WAVES)
#pragma unroll
                for (int i = 0; i < DQ_U; ++i) {
                    const int s = min(s0 + i * DQ_WAVES, Ts - 1);
                    pv[i] = ld4_any<XH>(pe, p0 + (int64_t)s * C);
                }
            }
#pragma unroll
            for (int i = 0; i < DQ_U; ++i) {
                const int s = s0 + i * DQ_WAVES;
                if (s < Ts) {                                      // wave-uniform: no tanh work for slots past the end
                    const float d = w[s];                          // masked positions carry d = 0
                    float th;
                    th = vag_tanh(pv[i].x + qv.x); acc.x += d * (1.f - th * th);
                    th = vag_tanh(pv[i].y + qv.y); acc.y += d * (1.f - th * th);
                    th = vag_tanh(pv[i].z + qv.z); acc.z += d * (1.f - th * th);
                    th = vag_tanh(pv[i].w + qv.w); acc.w += d * (1.f - th * th);
                }
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
                       hipStream_t s, bool x16) {
    VAG_CHECK_ARG(pe && q && v && dscore && dq && N > 0 && Ts > 0 && C > 0 && C % 4 == 0 && ldq % 4 == 0 && lddq % 4 == 0);
    VAG_CHECK_ARG((alpha == nullptr) == (dalpha == nullptr));
    dim3 grid((unsigned)cdiv64(C, 256), (unsigned)N);
    const size_t lds = (size_t)((Ts + 3) & ~3) * sizeof(float) + (size_t)DQ_WAVES * 64 * 16;
    if (x16)
        hipLaunchKernelGGL(attn_dq_kernel<true>, grid, dim3(64 * DQ_WAVES), lds, s, pe, q, ldq, v, alpha, dalpha, dscore,
                           (int)Ts, (int)C, dq, lddq);
    else
        hipLaunchKernelGGL(attn_dq_kernel<false>, grid, dim3(64 * DQ_WAVES), lds, s, pe, q, ldq, v, alpha, dalpha, dscore,
                           (int)Ts, (int)C, dq, lddq);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// ------------------------------------------------------------------ after the time loop: d_pe, dv partials, d_enc
// grid (ceil(C/256), B); thread owns one c and walks source positions in chunks of SC, all Tt steps per chunk.
constexpr int SC = VAG_POST_SC;
constexpr int POST_TMAX = 96;           // steps whose (ds, alpha) rows are staged in LDS per pass
template <bool XH>
__global__ __launch_bounds__(256) void attn_post_bwd_kernel(const float* __restrict__ pe, const float* __restrict__ q_all,
                                                            const float* __restrict__ v, const float* __restrict__ ds_all,
                                                            const float* __restrict__ alpha_all,
                                                            const float* __restrict__ dc_all, int B, int Ts, int Tt, int C,
                                                            int64_t ldq, float* __restrict__ d_pe, float* __restrict__ dvp,
                                                            float* __restrict__ d_enc, int acc_enc) {
    // the block's (ds, alpha) values -- Tt x SC of each, the same for every thread -- are staged in LDS once instead of being
    // fetched step by step inside the loop; the per-step query / context-gradient loads run four steps ahead
    __shared__ float sd[POST_TMAX][SC], sa[POST_TMAX][SC];
    const int b = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    const bool cok = c < C;
    const int s0 = blockIdx.z * SC;          // one chunk of SC source positions per block (grid.z chunks)
    const float vc = cok ? v[c] : 0.f;
    float dv = 0.f;
    float pv[SC], ape[SC], aen[SC];
#pragma unroll
    for (int i = 0; i < SC; ++i) {
        const int s = min(s0 + i, Ts - 1);
        pv[i] = cok ? ld1_any<XH>(pe, ((int64_t)b * Ts + s) * C + c) : 0.f;
        ape[i] = 0.f; aen[i] = 0.f;
    }
    for (int t0 = 0; t0 < Tt; t0 += POST_TMAX) {
        const int nt = min(POST_TMAX, Tt - t0);
        __syncthreads();
        for (int i = threadIdx.x; i < nt * SC; i += 256) {
            const int t = i / SC, k = i - t * SC;
            const int s = min(s0 + k, Ts - 1);
            const int64_t o = ((int64_t)(t0 + t) * B + b) * Ts + s;
            sd[t][k] = (s0 + k < Ts) ? ds_all[o] : 0.f;           // positions past the end contribute nothing
            sa[t][k] = alpha_all[o];
        }
        __syncthreads();
        if (!cok) continue;
        constexpr int U = 4;
        for (int t = 0; t < nt; t += U) {
            float qv[U], dcv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int tt = min(t + u, nt - 1) + t0;
                qv[u] = q_all[((int64_t)tt * B + b) * ldq + c];
                dcv[u] = dc_all ? dc_all[((int64_t)tt * B + b) * C + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (t + u < nt) {
#pragma unroll
                    for (int i = 0; i < SC; ++i) {
                        const float d = sd[t + u][i];
                        const float th = vag_tanh(pv[i] + qv[u]);
                        ape[i] += d * (1.f - th * th);
                        dv += d * th;
                        aen[i] += sa[t + u][i] * dcv[u];
                    }
                }
            }
        }
    }
    if (!cok) return;
#pragma unroll
    for (int i = 0; i < SC; ++i) {
        const int s = s0 + i;
        if (s < Ts) {
            const int64_t o = ((int64_t)b * Ts + s) * C + c;
            d_pe[o] = vc * ape[i];
            if (d_enc && dc_all) d_enc[o] = acc_enc ? d_enc[o] + aen[i] : aen[i];
        }
    }
    if (dvp) dvp[((int64_t)blockIdx.z * B + b) * C + c] = dv;      // partial per chunk: rows z*B + b
}
int vag_attn_post_bwd_launch(const float* pe, const float* q_all, int64_t ldq, const float* v, const float* ds_all,
                             const float* alpha_all, const float* dc_all, int64_t B, int64_t Ts, int64_t Tt,
                             int64_t C, float* d_pe, float* dvp, float* d_enc, int accumulate_enc, hipStream_t s, bool x16) {
    VAG_CHECK_ARG(pe && q_all && v && ds_all && alpha_all && d_pe && B > 0 && Ts > 0 && Tt > 0 && C > 0);
    dim3 grid((unsigned)cdiv64(C, 256), (unsigned)B, (unsigned)cdiv64(Ts, SC));
    if (x16)
        hipLaunchKernelGGL(attn_post_bwd_kernel<true>, grid, dim3(256), 0, s, pe, q_all, v, ds_all, alpha_all, dc_all, (int)B,
                           (int)Ts, (int)Tt, (int)C, ldq, d_pe, dvp, d_enc, accumulate_enc);
    else
        hipLaunchKernelGGL(attn_post_bwd_kernel<false>, grid, dim3(256), 0, s, pe, q_all, v, ds_all, alpha_all, dc_all, (int)B,
                           (int)Ts, (int)Tt, (int)C, ldq, d_pe, dvp, d_enc, accumulate_enc);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// ------------------------------------------------------------------ out[b,t,c] (+)= a1[b,t] x1[b,c] + a2[b,t] x2[b,c]
__global__ __launch_bounds__(256) void outer2_kernel(const float* __restrict__ a1, const float* __restrict__ x1,
                                                     const float* __restrict__ a2, const float* __restrict__ x2,
                                                     int Ts, int C, float* __restrict__ out, int acc) {
    // blockIdx.z = a chunk of 4 positions (round 3: one thread walked all Ts positions of its column, a chain of Ts dependent
    // read-modify-writes on 64 workgroups: 24.8 us for 21 MB)
    const int b = blockIdx.y;
    const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (c >= C) return;
    const float4 u = *reinterpret_cast<const float4*>(x1 + (int64_t)b * C + c);
    const float4 w = a2 ? *reinterpret_cast<const float4*>(x2 + (int64_t)b * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int t_end = min(Ts, (int)(blockIdx.z + 1) * 4);
    for (int t = blockIdx.z * 4; t < t_end; ++t) {
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
// Held-back accumulations into one (B,Ts,C) tensor (vag_train_step: d_enc of the visual-grounding / initial-state backward).  Between
// vag_rmw_defer_begin(out) and vag_rmw_defer_flush() an accumulating vag_outer2_launch / vag_meanpool_bwd_launch into `out` is only
// recorded; the flush makes ONE pass over the tensor for both (they were two read-modify-write passes of 21 MB each).  Calling thread.
struct RmwDefer {
    float* out = nullptr;
    const float *a1 = nullptr, *x1 = nullptr, *a2 = nullptr, *x2 = nullptr;      // outer2's operands
    const float *mask = nullptr, *dx = nullptr; float coef = 0.f;                // meanpool_bwd's
    int64_t B = 0, Ts = 0, C = 0;
};
static thread_local RmwDefer g_rmw;
void vag_rmw_defer_begin(float* out) { g_rmw = RmwDefer(); g_rmw.out = out; }
void vag_rmw_defer_abort() { g_rmw = RmwDefer(); }
static bool rmw_shape(int64_t B, int64_t Ts, int64_t C) {
    if (g_rmw.B == 0) { g_rmw.B = B; g_rmw.Ts = Ts; g_rmw.C = C; return true; }
    return g_rmw.B == B && g_rmw.Ts == Ts && g_rmw.C == C;
}
bool vag_rmw_defer_meanpool(const float* mask, const float* dx, float coef, int64_t B, int64_t Ts, int64_t C, float* out, int accumulate) {
    if (!g_rmw.out || out != g_rmw.out || !accumulate || g_rmw.dx || C % 4 != 0 || !rmw_shape(B, Ts, C)) return false;
    g_rmw.mask = mask; g_rmw.dx = dx; g_rmw.coef = coef;
    return true;
}
// out[b,t,c] += a1[b,t] x1[b,c] + a2[b,t] x2[b,c] + coef dx[b,c] / #(mask[b,:])   (any of the three terms may be absent)
__global__ __launch_bounds__(256) void outer3_kernel(RmwDefer d) {
    const int b = blockIdx.y, Ts = (int)d.Ts, C = (int)d.C;
    const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (c >= C) return;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 u = d.a1 ? *reinterpret_cast<const float4*>(d.x1 + (int64_t)b * C + c) : z;
    const float4 w = d.a2 ? *reinterpret_cast<const float4*>(d.x2 + (int64_t)b * C + c) : z;
    float4 m = z;
    if (d.dx) {
        float cnt = 0.f;
        for (int t = 0; t < Ts; ++t) cnt += d.mask[(int64_t)b * Ts + t];
        const float4 x = *reinterpret_cast<const float4*>(d.dx + (int64_t)b * C + c);
        m = make_float4(d.coef * x.x / cnt, d.coef * x.y / cnt, d.coef * x.z / cnt, d.coef * x.w / cnt);
    }
    const int t0 = blockIdx.z * 4, t_end = min(Ts, t0 + 4);
    float4 old[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (t0 + i < t_end) old[i] = *reinterpret_cast<const float4*>(d.out + ((int64_t)b * Ts + t0 + i) * C + c);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (t0 + i < t_end) {
            const int t = t0 + i;
            const float p = d.a1 ? d.a1[(int64_t)b * Ts + t] : 0.f;
            const float r = d.a2 ? d.a2[(int64_t)b * Ts + t] : 0.f;
            float4 v = old[i];
            // (the order the two separate passes added in: the mean-pool term first, then the outer products)
            v.x += m.x; v.y += m.y; v.z += m.z; v.w += m.w;
            v.x += p * u.x + r * w.x; v.y += p * u.y + r * w.y; v.z += p * u.z + r * w.z; v.w += p * u.w + r * w.w;
            *reinterpret_cast<float4*>(d.out + ((int64_t)b * Ts + t) * C + c) = v;
        }
}
int vag_rmw_defer_flush(hipStream_t s) {
    RmwDefer d = g_rmw;
    g_rmw = RmwDefer();
    if (!d.out || (!d.a1 && !d.dx)) return VAG_OK;
    dim3 grid((unsigned)cdiv64(d.C, 1024), (unsigned)d.B, (unsigned)cdiv64(d.Ts, 4));
    hipLaunchKernelGGL(outer3_kernel, grid, dim3(256), 0, s, d);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}
int vag_outer2_launch(const float* a1, const float* x1, const float* a2, const float* x2, int64_t B, int64_t Ts,
                      int64_t C, float* out, int accumulate, hipStream_t s) {
    VAG_CHECK_ARG(a1 && x1 && out && B > 0 && Ts > 0 && C > 0 && C % 4 == 0);
    if (g_rmw.out && out == g_rmw.out && accumulate && !g_rmw.a1 && rmw_shape(B, Ts, C)) {
        g_rmw.a1 = a1; g_rmw.x1 = x1; g_rmw.a2 = a2; g_rmw.x2 = a2 ? x2 : nullptr;
        return VAG_OK;
    }
    dim3 grid((unsigned)cdiv64(C, 1024), (unsigned)B, (unsigned)cdiv64(Ts, 4));
    hipLaunchKernelGGL(outer2_kernel, grid, dim3(256), 0, s, a1, x1, a2, x2, (int)Ts, (int)C, out, accumulate);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}
