// Visual-semantic-embedding kernels: row L2 normalisation (fwd/bwd, fused with the tanh backward) and the
// max-margin ranking loss over the BxB similarity matrix (loss value and d loss / d scores in one pass).
#include "kernels.h"

// one 256-thread block per row (rows are a mini-batch: few of them, so a wave per row leaves the chip idle and pays a
// dependent load round per 64 elements)
__device__ __forceinline__ float vse_block_sum(float v, float* sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return r;
}
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ y, int64_t B, int S,
                                                         float* __restrict__ nrm, float* __restrict__ out) {
    __shared__ float sh[4];
    const int64_t b = blockIdx.x;
    const float* r = y + b * S;
    float ss = 0.f;
    for (int j = threadIdx.x; j < S; j += 256) ss += r[j] * r[j];
    ss = vse_block_sum(ss, sh);
    const float n = fmaxf(sqrtf(ss), 1e-12f);       // utils/utils.py:10  clamp(min=eps)
    if (threadIdx.x == 0) nrm[b] = n;
    const float inv = 1.f / n;
    for (int j = threadIdx.x; j < S; j += 256) out[b * S + j] = r[j] * inv;
}
int vag_l2norm_fwd_launch(const float* y, int64_t B, int64_t S, float* nrm, float* out, hipStream_t s) {
    VAG_CHECK_ARG(y && nrm && out && B > 0 && S > 0);
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((unsigned)B), dim3(256), 0, s, y, B, (int)S, nrm, out);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// out = y / n, n = max(|y|, eps).  If |y| > eps: dy = (d_out - out (out . d_out)) / n, else dy = d_out / n.
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ y, const float* __restrict__ nrm,
                                                         const float* __restrict__ out, const float* __restrict__ d_out,
                                                         int64_t B, int S, int act, float* __restrict__ dy) {
    __shared__ float sh[4];
    const int64_t b = blockIdx.x;
    const float n = nrm[b];
    float dot = 0.f;
    for (int j = threadIdx.x; j < S; j += 256) dot += out[b * S + j] * d_out[b * S + j];
    dot = vse_block_sum(dot, sh);
    if (!(n > 1e-12f)) dot = 0.f;
    const float inv = 1.f / n;
    for (int j = threadIdx.x; j < S; j += 256) {
        float g = (d_out[b * S + j] - out[b * S + j] * dot) * inv;
        if (act) {
            const float yy = y[b * S + j];
            g *= (1.f - yy * yy);
        }
        dy[b * S + j] = g;
    }
}
int vag_l2norm_bwd_launch(const float* y, const float* nrm, const float* out, const float* d_out, int64_t B, int64_t S,
                          int act, float* dy, hipStream_t s) {
    VAG_CHECK_ARG(y && nrm && out && d_out && dy && B > 0 && S > 0);
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((unsigned)B), dim3(256), 0, s, y, nrm, out, d_out, B, (int)S, act, dy);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// loss = sum_{i != j} max(0, m - d_j + S_ij)  [+ sum_{i != j} max(0, m - d_i + S_ij) when kind == 0]
// G = d loss / d S.  Single 1024-thread block (B is a mini-batch size); column/row hinge counts go through LDS.
// g_scale (optional, one device float): G is stored multiplied by it -- the fused step hands in the loss weight, its backward then
// needs no scaling pass over d(im) / d(s).
__global__ __launch_bounds__(1024) void rank_loss_kernel(const float* __restrict__ S, int B, float margin, int kind,
                                                         float* __restrict__ G, float* __restrict__ loss,
                                                         const float* __restrict__ g_scale) {
    const float gs = g_scale ? g_scale[0] : 1.f;
    extern __shared__ float sh[];        // [B] diag-grad accumulators, then 16 floats of reduction space
    float* dacc = sh;
    float* red = sh + B;
    for (int i = threadIdx.x; i < B; i += 1024) dacc[i] = 0.f;
    __syncthreads();
    float part = 0.f;
    const int64_t total = (int64_t)B * B;
    for (int64_t e = threadIdx.x; e < total; e += 1024) {
        const int i = (int)(e / B), j = (int)(e - (int64_t)i * B);
        if (i == j) continue;
        const float sij = S[e];
        float g = 0.f;
        const float cs = margin - S[(int64_t)j * B + j] + sij;      // PairwiseRankingLoss.py:16
        if (cs > 0.f) { part += cs; g += 1.f; atomicAdd(&dacc[j], -1.f); }
        if (kind == 0) {
            const float ci = margin - S[(int64_t)i * B + i] + sij;  // PairwiseRankingLoss.py:18
            if (ci > 0.f) { part += ci; g += 1.f; atomicAdd(&dacc[i], -1.f); }
        }
        G[e] = g * gs;
    }
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += red[w];
        loss[0] = t;
    }
    for (int i = threadIdx.x; i < B; i += 1024) G[(int64_t)i * B + i] = dacc[i] * gs;
}
int vag_rank_loss_launch(const float* scores, int64_t B, float margin, int kind, float* G, float* loss, hipStream_t s,
                         const float* g_scale) {
    VAG_CHECK_ARG(scores && G && loss && B > 0 && B <= 8192 && (kind == 0 || kind == 1));
    hipLaunchKernelGGL(rank_loss_kernel, dim3(1), dim3(1024), (size_t)(B + 16) * sizeof(float), s, scores, (int)B, margin,
                       kind, G, loss, g_scale);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// Both gradients of the similarity matrix in one launch: d_im = G s (PairwiseRankingLoss.py:12: scores = im s^T), d_s = G^T im.
// grid (ceil(S/64), ceil(B/16), 2); block = 64 columns x 4 row groups of 4 rows; the 16 x B slice of G (or G^T) and the B x 64 tile
// of the other operand are staged in LDS with every load of the block in flight at once (the k loop then runs out of LDS).
__global__ __launch_bounds__(256) void rank_bwd_kernel(const float* __restrict__ G, const float* __restrict__ im,
                                                       const float* __restrict__ sv, const float* __restrict__ d_loss, int B, int S,
                                                       float* __restrict__ d_im, float* __restrict__ d_s) {
    extern __shared__ __attribute__((aligned(16))) float gsh[];      // [16][B] slice of G / G^T, then [B][64] tile of x
    float* xs = gsh + 16 * B;
    const bool tr = blockIdx.z == 1;            // d_s: rows of G^T
    const float* x = tr ? im : sv;
    float* out = tr ? d_s : d_im;
    const int i0 = blockIdx.y * 16, c0 = blockIdx.x * 64;
    // every load of the block is issued before the first LDS store (B % 16 == 0, S % 4 == 0: 16-byte loads; one at a time -- a
    // load, its store, the next load -- the staging alone took ~10 us out of cold memory)
    {
        const int B4 = B >> 2;
        constexpr int U = 4;
        for (int e0 = threadIdx.x; e0 < 4 * B; e0 += 256 * U) {          // G slice: 16 rows x B (or B rows x 16 of G for G^T)
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = e0 + 256 * u;
                if (e < 4 * B) v[u] = tr ? *reinterpret_cast<const float4*>(G + (int64_t)(e >> 2) * B + i0 + 4 * (e & 3))
                                         : *reinterpret_cast<const float4*>(G + (int64_t)(i0 + e / B4) * B + 4 * (e % B4));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = e0 + 256 * u;
                if (e >= 4 * B) continue;
                if (tr) {
                    const int j = e >> 2, r = 4 * (e & 3);
                    gsh[(r + 0) * B + j] = v[u].x; gsh[(r + 1) * B + j] = v[u].y; gsh[(r + 2) * B + j] = v[u].z; gsh[(r + 3) * B + j] = v[u].w;
                } else {
                    *reinterpret_cast<float4*>(gsh + (e / B4) * B + 4 * (e % B4)) = v[u];
                }
            }
        }
        for (int e0 = threadIdx.x; e0 < 16 * B; e0 += 256 * U) {         // x tile: B rows x 64 columns
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = e0 + 256 * u, j = e >> 4, c = c0 + 4 * (e & 15);
                v[u] = (e < 16 * B && c < S) ? *reinterpret_cast<const float4*>(x + (int64_t)j * S + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = e0 + 256 * u;
                if (e < 16 * B) *reinterpret_cast<float4*>(xs + 4 * e) = v[u];
            }
        }
    }
    __syncthreads();
    const int cl = threadIdx.x & 63, c = c0 + cl, rg = threadIdx.x >> 6;
    if (c >= S) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* g = gsh + 4 * rg * B;
    for (int j = 0; j < B; ++j) {
        const float xv = xs[j * 64 + cl];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += g[r * B + j] * xv;
    }
    const float sc = d_loss ? d_loss[0] : 1.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + 4 * rg + r;
        if (i < B) out[(int64_t)i * S + c] = acc[r] * sc;
    }
}
int vag_rank_bwd_launch(const float* G, const float* im, const float* sv, const float* d_loss, int64_t B, int64_t S, float* d_im,
                        float* d_s, hipStream_t s) {
    VAG_CHECK_ARG(G && im && sv && d_im && d_s && B > 0 && B <= 128 && B % 16 == 0 && S > 0 && S % 4 == 0 && aligned16(G) && aligned16(im) &&
                  aligned16(sv));
    dim3 grid((unsigned)cdiv64(S, 64), (unsigned)cdiv64(B, 16), 2);
    hipLaunchKernelGGL(rank_bwd_kernel, grid, dim3(256), (size_t)(80 * B) * sizeof(float), s, G, im, sv, d_loss, (int)B, (int)S, d_im, d_s);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

__global__ __launch_bounds__(256) void scale_by_dev_kernel(float* __restrict__ x, int64_t n, const float* __restrict__ sc) {
    const float a = sc[0];
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] *= a;
}
int vag_scale_by_dev_launch(float* x, int64_t n, const float* scalar, hipStream_t s) {
    if (n == 0) return VAG_OK;
    int64_t b = cdiv64(n, 256);
    if (b > 4096) b = 4096;
    hipLaunchKernelGGL(scale_by_dev_kernel, dim3((unsigned)b), dim3(256), 0, s, x, n, scalar);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// Retrieval ranks (utils/im_retrieval_eval.py:15-24): rank[i] = position of key i in the descending sort of scores[i,:]
// = #{j : S_ij > S_ii} (+ equal scores at smaller j, the order a stable descending sort gives).  One wave per query.
__global__ __launch_bounds__(256) void retrieval_rank_kernel(const float* __restrict__ S, int N, int* __restrict__ ranks) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    const float* row = S + (int64_t)i * N;
    const float d = row[i];
    int cnt = 0;
    for (int j = lane; j < N; j += 64) {
        const float v = row[j];
        cnt += (v > d) || (v == d && j < i);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (lane == 0) ranks[i] = cnt;
}
int vag_retrieval_rank_launch(const float* scores, int64_t N, int* ranks, hipStream_t s) {
    VAG_CHECK_ARG(scores && ranks && N > 0 && N < (1ll << 30));
    hipLaunchKernelGGL(retrieval_rank_kernel, dim3((unsigned)cdiv64(N, 4)), dim3(256), 0, s, scores, (int)N, ranks);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}
