// Batched beam-search expansion (models/NMT_AttentionImagine_Seq2Seq_Beam_V11.py:259-324) without host syncs.
//   stage 1: grid (chunks, B): each block selects the k best of an 8192-candidate slice of the k_in*V
//            continuations, applying the reference's penalties on the fly (repeat-token suppression,
//            finished hypotheses may only emit EOS at cost 0).
//   stage 2: one block per sentence merges the chunk winners, updates running scores, token history
//            (back-pointer permutation) and re-orders the decoder hidden state for the next step.
// Selection uses the total order (score desc, flat index asc), so results are deterministic; the reference's
// topk(sorted=False) leaves the order of equal-score candidates unspecified.
#include "kernels.h"

constexpr int EPT = 32;                  // candidates per thread in stage 1
constexpr int CHUNK = 256 * EPT;
constexpr float NEG_PEN = -1e5f;         // the reference's "inf" (V11.py:257)
constexpr int64_t EOS = 3;

struct Cand { float v; int idx; };

__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

// block-wide argmax under the (value desc, index asc) order; result valid in all threads
__device__ __forceinline__ Cand block_best(Cand c, Cand* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(c.v, o, 64);
        const int oi = __shfl_xor(c.idx, o, 64);
        if (better(ov, oi, c.v, c.idx)) { c.v = ov; c.idx = oi; }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = c;
    __syncthreads();
    Cand r = sh[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i)
        if (better(sh[i].v, sh[i].idx, r.v, r.idx)) r = sh[i];
    __syncthreads();
    return r;
}

// Selection in both stages: every thread caches the best of the candidates it owns; a round is one block-wide argmax
// of the cached bests, and only the winner's owner rescans its (register- or LDS-resident) candidates.
__global__ __launch_bounds__(256) void beam_stage1_kernel(const float* __restrict__ logp, int64_t ldl,
                                                          const float* __restrict__ nll, const int64_t* __restrict__ prev_tok,
                                                          int k_in, int k, int V, int penal, float* __restrict__ cval,
                                                          int* __restrict__ cidx, int32_t* __restrict__ n_alive) {
    __shared__ Cand sh[4];
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *n_alive = 0;   // stage 2 (next launch) counts into it
    const int b = blockIdx.y, chunk = blockIdx.x, chunks = gridDim.x;
    const int total = k_in * V;
    const int f0 = chunk * CHUNK + threadIdx.x;
    const float rV = 1.f / (float)V;
    float val[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int f = f0 + e * 256;                               // flat index j*V + w
        float v = -INFINITY;
        if (f < total) {
            int j = (int)((float)f * rV);                         // f < 2^24: exact up to one unit
            if (j * V > f) --j;
            else if ((j + 1) * V <= f) ++j;
            const int w = f - j * V;
            const int64_t n = (int64_t)b * k_in + j;
            float lp = logp[n * ldl + w];
            if (penal) {
                const int64_t pt = prev_tok[n];
                if (pt == EOS) lp = (w == EOS) ? 0.f : NEG_PEN;   // V11.py:291-294
                else if (w == pt) lp = NEG_PEN;                   // V11.py:279-280
            }
            v = (nll ? nll[n] : 0.f) + lp;                        // V11.py:297
        }
        val[e] = v;
    }
    unsigned taken = 0;
    auto scan = [&]() {
        Cand c = {-INFINITY, 0x7fffffff};
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int f = f0 + e * 256;
            if (!((taken >> e) & 1u) && f < total && better(val[e], f, c.v, c.idx)) { c.v = val[e]; c.idx = f; }
        }
        return c;
    };
    Cand mine = scan();
    for (int r = 0; r < k; ++r) {
        const Cand c = block_best(mine, sh);
        if (threadIdx.x == 0) {
            const int64_t o = ((int64_t)b * chunks + chunk) * k + r;
            cval[o] = c.v; cidx[o] = c.idx;
        }
        if (c.idx != 0x7fffffff && mine.idx == c.idx) {
            taken |= 1u << ((c.idx - f0) >> 8);
            mine = scan();
        }
    }
}

constexpr int S2_LDS = 2048;             // candidates kept in LDS by stage 2 (more: selection works on the scratch copy)

__global__ __launch_bounds__(256) void beam_stage2_kernel(float* __restrict__ cval, int* __restrict__ cidx,
                                                          int chunks, int k_in, int k, int V, int H,
                                                          float* __restrict__ nll, int64_t* __restrict__ beam, int di,
                                                          int B, const float* __restrict__ h_in, float* __restrict__ h_out,
                                                          int32_t* __restrict__ n_alive) {
    __shared__ Cand sh[4];
    __shared__ int sel_idx[64];
    __shared__ float sel_val[64];
    __shared__ float lv[S2_LDS];
    __shared__ int li[S2_LDS];
    const int b = blockIdx.x;
    const int ncand = chunks * k;
    float* pv = cval + (int64_t)b * ncand;
    int* pi = cidx + (int64_t)b * ncand;
    if (ncand <= S2_LDS) {
        for (int e = threadIdx.x; e < ncand; e += 256) { lv[e] = pv[e]; li[e] = pi[e]; }
        pv = lv; pi = li;
        __syncthreads();
    }
    int mine_e = -1;
    auto scan = [&]() {
        Cand c = {-INFINITY, 0x7fffffff};
        mine_e = -1;
        for (int e = threadIdx.x; e < ncand; e += 256) {
            const int f = pi[e];
            if (f != 0x7fffffff && better(pv[e], f, c.v, c.idx)) { c.v = pv[e]; c.idx = f; mine_e = e; }
        }
        return c;
    };
    Cand mine = scan();
    for (int r = 0; r < k; ++r) {
        const Cand c = block_best(mine, sh);
        if (threadIdx.x == 0) { sel_idx[r] = c.idx; sel_val[r] = c.v; }
        if (c.idx != 0x7fffffff && mine.idx == c.idx) {
            pi[mine_e] = 0x7fffffff;                               // taken (only its owner reads this slot again)
            mine = scan();
        }
    }
    __syncthreads();
    // history permutation (V11.py:309): every thread owns time steps t, reads the k old tokens, writes the new ones
    for (int t = threadIdx.x; t < di; t += 256) {
        int64_t* row = beam + ((int64_t)t * B + b) * k;
        int64_t old[64];
        for (int j = 0; j < k_in; ++j) old[j] = row[j];
        for (int j = 0; j < k; ++j) row[j] = old[sel_idx[j] / V];
    }
    if (threadIdx.x < k) {
        const int j = threadIdx.x;
        const int f = sel_idx[j];
        const int64_t w = f % V;
        beam[((int64_t)di * B + b) * k + j] = w;                    // V11.py:306
        nll[(int64_t)b * k + j] = sel_val[j];
        if (w != EOS) atomicAdd(n_alive, 1);
    }
    // hidden-state re-tiling for the next step (V11.py:273,:313)
    if ((H & 3) == 0) {
        const int H4 = H >> 2;
        for (int e = threadIdx.x; e < k * H4; e += 256) {
            const int j = e / H4, c = e - j * H4;
            const int src = sel_idx[j] / V;
            reinterpret_cast<float4*>(h_out + ((int64_t)b * k + j) * H)[c] =
                reinterpret_cast<const float4*>(h_in + ((int64_t)b * k_in + src) * H)[c];
        }
    } else {
        for (int e = threadIdx.x; e < k * H; e += 256) {
            const int j = e / H, c = e - j * H;
            const int src = sel_idx[j] / V;
            h_out[((int64_t)b * k + j) * H + c] = h_in[((int64_t)b * k_in + src) * H + c];
        }
    }
}

int64_t vag_beam_scratch_bytes_impl(int64_t B, int64_t k, int64_t V) {
    const int64_t chunks = cdiv64(k * V, CHUNK);
    return B * chunks * k * 8 + 64;
}

int vag_beam_step_launch(float* logp, int64_t ldl, float* nll, int64_t* beam, int64_t di, int64_t max_len,
                         const float* h_in, float* h_out, int64_t B, int64_t k, int64_t V, int64_t H,
                         int32_t* n_alive, void* scratch, hipStream_t s) {
    VAG_CHECK_ARG(logp && nll && beam && h_in && h_out && n_alive && scratch);
    VAG_CHECK_ARG(B > 0 && k > 0 && k <= 64 && V > 0 && H > 0 && di >= 0 && di < max_len && ldl >= V);
    const int k_in = di == 0 ? 1 : (int)k;
    const int64_t total = (int64_t)k_in * V;
    VAG_CHECK_ARG(total < (1ll << 31) && total >= k);
    const int chunks = (int)cdiv64(total, CHUNK);
    float* cval = reinterpret_cast<float*>(scratch);
    int* cidx = reinterpret_cast<int*>(cval + B * cdiv64(k * V, CHUNK) * k);
    const int64_t* prev = di > 0 ? beam + (di - 1) * B * k : nullptr;
    hipLaunchKernelGGL(beam_stage1_kernel, dim3((unsigned)chunks, (unsigned)B), dim3(256), 0, s, logp, ldl,
                       di > 0 ? nll : (const float*)nullptr, prev, k_in, (int)k, (int)V, di > 0 ? 1 : 0, cval, cidx, n_alive);
    VAG_LAUNCH_CHECK();
    hipLaunchKernelGGL(beam_stage2_kernel, dim3((unsigned)B), dim3(256), 0, s, cval, cidx, chunks, k_in, (int)k, (int)V,
                       (int)H, nll, beam, (int)di, (int)B, h_in, h_out, n_alive);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// V11.py:315-324: force EOS in the last row, normalise by the number of tokens > 3, pick the best hypothesis.
__global__ __launch_bounds__(64) void beam_finish_kernel(const float* __restrict__ nll, int64_t* __restrict__ beam,
                                                         int max_len, int B, int k, int64_t* __restrict__ out,
                                                         float* __restrict__ best) {
    const int b = blockIdx.x, j = threadIdx.x;
    float sc = -INFINITY;
    if (j < k) {
        beam[((int64_t)(max_len - 1) * B + b) * k + j] = EOS;    // EOS (=3) never counts towards the length
        int len = 0;
        for (int t = 0; t < max_len - 1; ++t) len += beam[((int64_t)t * B + b) * k + j] > 3;
        if (len < 1) len = 1;
        sc = nll[(int64_t)b * k + j] / (float)len;
    }
    float bv = sc;
    int bi = j < k ? j : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    for (int t = j; t < max_len; t += 64)
        out[(int64_t)b * max_len + t] = (t == max_len - 1) ? EOS : beam[((int64_t)t * B + b) * k + bi];
    if (j == 0 && best) best[b] = bv;
}

int vag_beam_finish_launch(const float* nll, int64_t* beam, int64_t max_len, int64_t B, int64_t k, int64_t* out,
                           float* best, hipStream_t s) {
    VAG_CHECK_ARG(nll && beam && out && max_len > 0 && B > 0 && k > 0 && k <= 64);
    hipLaunchKernelGGL(beam_finish_kernel, dim3((unsigned)B), dim3(64), 0, s, nll, beam, (int)max_len, (int)B, (int)k, out, best);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}
