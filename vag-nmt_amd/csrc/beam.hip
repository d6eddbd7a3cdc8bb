// Batched beam-search expansion (models/NMT_AttentionImagine_Seq2Seq_Beam_V11.py:259-324) without host syncs.
//   stage 1: grid (chunks, B): each block selects the k best of a 2048-candidate slice of the k_in*V
//            continuations, applying the reference's penalties on the fly (repeat-token suppression,
//            finished hypotheses may only emit EOS at cost 0).
//   stage 2: one block per sentence merges the chunk winners, updates running scores, appends (token, parent) to the
//            history and re-orders the decoder hidden state for the next step.  The history is kept as back-pointers
//            (rows [max_len, 2 max_len) of the beam buffer) and resolved once by the finish kernel, instead of
//            permuting all earlier rows at every step as the reference does (V11.py:309) -- same hypotheses.
// Selection uses the total order (score desc, flat index asc), so results are deterministic; the reference's
// topk(sorted=False) leaves the order of equal-score candidates unspecified.
#include "kernels.h"

constexpr int EPT = 8;                   // candidates per thread in stage 1 (a rescan after each pick walks these)
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

// ---- selection by radix search (round 2; 56-bit unique keys: round 5) ----
// The k best of the candidates one WAVE holds in registers (E per lane), under the total order (value desc, flat index asc),
// without any cross-lane data movement.  A candidate's KEY is its order-preserving value bits followed by its inverted flat
// index (indices are below 2^24: vag_beam_step_launch checks k V < 2^24): keys are unique, a larger key is a better candidate,
// and ties on the value need no handling of their own.  That matters: every continuation of a FINISHED hypothesis except EOS
// carries the same value (its score - 1e5, V11.py:291-294), so once hypotheses have ended whole 2048-candidate slices tie.  With
// 32-bit value keys such slices fell through to an exact search plus a tie loop in every wave: the expansion took 25-32 us
// instead of 12, a beam step of a trained model 120 us instead of 94 (profiles/r05_exp_beam.txt).
__device__ __forceinline__ unsigned fkey(float v) {       // order-preserving float -> uint (larger float, larger key)
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
// A key in two words: hi = value bits (0 = hole; fkey(-inf) = 0x007fffff > 0, so a real candidate beats a hole), lo = inverted flat
// index.  The searches below run over the 32 value bits and go on into the 24 index bits only when the k-th value is tied.
struct Key2 { unsigned hi, lo; };
__device__ __forceinline__ Key2 ckey(float v, int idx) {
    Key2 q;
    q.hi = idx == 0x7fffffff ? 0u : fkey(v);
    q.lo = 0xffffffu - ((unsigned)idx & 0xffffffu);
    return q;
}
__device__ __forceinline__ bool key_ge(Key2 a, Key2 t) { return a.hi > t.hi || (a.hi == t.hi && a.lo >= t.lo); }
// the k-th largest of one key per lane (at least k lanes hold a valid key), found bit by bit from ballots
__device__ __forceinline__ Key2 wave_kth_largest(Key2 x, int k) {
    unsigned prefix = 0;
    int need = k;
    for (int b = 31; b >= 0; --b) {
        const unsigned test = prefix | (1u << b);
        const int c = __popcll(__ballot((x.hi >> b) == (test >> b)));
        if (c >= need) prefix = test;
        else need -= c;
    }
    Key2 t = {prefix, 0u};
    if (__popcll(__ballot(x.hi == prefix)) == need) return t;          // no tie on the k-th value: every key of that value counts
    for (int b = 23; b >= 0; --b) {                                     // tie: the `need` smallest indices among the tied lanes
        const unsigned test = t.lo | (1u << b);
        const int c = __popcll(__ballot(x.hi == prefix && (x.lo >> b) == (test >> b)));
        if (c >= need) t.lo = test;
        else need -= c;
    }
    return t;
}
// exact search over all E keys per lane (rare: the bound of wave_topk let more than 64 candidates through)
template <int E>
__device__ __forceinline__ void wave_select(const Key2 (&key)[E], int k, bool (&sel)[E]) {
    int nvalid = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) { nvalid += __popcll(__ballot(key[e].hi != 0u)); sel[e] = false; }
    const int kk = min(k, nvalid);
    if (kk == 0) return;
    unsigned prefix = 0;
    int need = kk;
    for (int b = 31; b >= 0; --b) {
        const unsigned test = prefix | (1u << b);
        int c = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) c += __popcll(__ballot((key[e].hi >> b) == (test >> b)));
        if (c >= need) prefix = test;
        else need -= c;
    }
    Key2 t = {prefix, 0u};
    int neq = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) neq += __popcll(__ballot(key[e].hi == prefix));
    if (neq != need) {
        for (int b = 23; b >= 0; --b) {
            const unsigned test = t.lo | (1u << b);
            int c = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) c += __popcll(__ballot(key[e].hi == prefix && (key[e].lo >> b) == (test >> b)));
            if (c >= need) t.lo = test;
            else need -= c;
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) sel[e] = key[e].hi != 0u && key_ge(key[e], t);      // (keys are unique: exactly kk)
}
// Writes the selected candidates of a wave densely to (ov, oi)[0 .. count): returns count (uniform over the wave).
template <int E>
__device__ __forceinline__ int wave_compact(const float (&val)[E], const int (&idx)[E], const bool (&sel)[E], float* ov, int* oi) {
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    int n = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const unsigned long long m = __ballot(sel[e]);
        if (sel[e]) {
            const int pos = n + __popcll(m & lt);
            ov[pos] = val[e]; oi[pos] = idx[e];
        }
        n += __popcll(m);
    }
    return n;
}

// The k best of a wave's candidates, RANKED (ov/oi[0..n): best first), n = min(k, #valid) returned.  Most candidates are
// discarded by a bound that costs 32 ballots whatever E is: the k-th largest of the 64 lane-local maxima is a lower
// bound of the k-th largest overall (those k lane maxima are k distinct candidates), so only candidates >= it can be
// winners -- typically k to 2k survive.  Survivors (<= 64: one per lane) are ranked by counting who beats them; with more
// survivors (heavy ties) the exact radix search over all E takes over.  sv/si: 64 entries of LDS scratch of this wave.
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
template <int E>
__device__ __forceinline__ int wave_topk(const float (&val)[E], const int (&idx)[E], int k, float* sv, int* si, float* ov, int* oi) {
    const int lane = threadIdx.x & 63;
    Key2 key[E], kb = {0u, 0u};
#pragma unroll
    for (int e = 0; e < E; ++e) {
        key[e] = ckey(val[e], idx[e]);
        if (key[e].hi != 0u && (kb.hi == 0u || !key_ge(kb, key[e]))) kb = key[e];
    }
    Key2 t0 = {1u, 0u};                                 // every valid key has hi > 1
    if (__popcll(__ballot(kb.hi != 0u)) >= k) t0 = wave_kth_largest(kb, k);     // k-th largest of the lane maxima
    bool sel[E];
    int n = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        sel[e] = key[e].hi != 0u && key_ge(key[e], t0);
        n += __popcll(__ballot(sel[e]));
    }
    if (n > 64) {                                       // more than one per lane got through: exact search over everything
        wave_select<E>(key, k, sel);
    }
    n = wave_compact<E>(val, idx, sel, sv, si);
    wave_lds_fence();
    if (lane < n) {
        const float mv = sv[lane];
        const int mi = si[lane];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += better(sv[j], si[j], mv, mi) ? 1 : 0;
        if (rank < k) { ov[rank] = mv; oi[rank] = mi; }
    }
    return min(n, k);
}

// Selection in both stages (fallback path): every thread caches the best of the candidates it owns; a round is one block-wide argmax
// of the cached bests, and only the winner's owner rescans its (register- or LDS-resident) candidates.
// The step index comes from the host (di_host) or, for launches replayed from a HIP graph, from device memory
// (di_state[0], advanced by stage 2; such launches are always steps >= 1, i.e. k_in == k).
__global__ __launch_bounds__(256) void beam_stage1_kernel(const float* __restrict__ logp, int64_t ldl,
                                                          const float* __restrict__ nll_in, const int64_t* __restrict__ beam,
                                                          const int32_t* di_state, int di_host, int max_len, int B,
                                                          int k_in, int k, int V, float* __restrict__ cval,
                                                          int* __restrict__ cidx, int32_t* __restrict__ n_alive,
                                                          const float* __restrict__ parts, int nparts) {
    const int di = di_state ? __atomic_load_n(di_state, __ATOMIC_RELAXED) : di_host;
    if (di >= max_len || (di_state && di < 1)) return;                          // replayed past the end: nothing to do
    // parts != NULL: `logp` holds raw logits and parts (nparts, rows, 2) the (max, sum exp) pieces of every row's log-sum-exp
    // (the vocabulary product's epilogue wrote them: gemm.hip, TallArgs::parts).  A chunk of 2048 candidates touches at most
    // ceil(2048 / V) + 1 rows; waves 0..3 combine the pieces of the first four of them (V >= 683 whenever pieces exist).
    __shared__ float lse_s[4];
    const int jfirst = (int)(((int64_t)blockIdx.x * CHUNK) / V);
    if (parts) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const int j = min(jfirst + wave, k_in - 1);
        const int64_t rows = (int64_t)B * k_in;                    // pieces are laid out [piece][row]
        const float* pr = parts + ((int64_t)blockIdx.y * k_in + j) * 2;
        float m = -INFINITY;
        for (int x = lane; x < nparts; x += 64) m = fmaxf(m, pr[2 * x * rows]);
        m = wave_max(m);
        float sm = 0.f;
        for (int x = lane; x < nparts; x += 64) sm += pr[2 * x * rows + 1] * __expf(pr[2 * x * rows] - m);
        sm = wave_sum(sm);
        if (lane == 0) lse_s[wave] = m + __logf(sm);
        __syncthreads();
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *n_alive = 0;   // stage 2 (next launch) counts into it
    const int penal = di > 0;
    const float* nll = di > 0 ? nll_in : nullptr;
    const int64_t* prev_tok = di > 0 ? beam + (int64_t)(di - 1) * B * k : beam;
    const int b = blockIdx.y, chunk = blockIdx.x, chunks = gridDim.x;
    const int total = k_in * V;
    const int f0 = chunk * CHUNK + threadIdx.x;
    const float rV = 1.f / (float)V;
    // branch-free so that all 3*EPT loads of a thread are in flight together (indices clamped, result selected)
    float val[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int f = f0 + e * 256;                               // flat index j*V + w
        const int fc = min(f, total - 1);
        int j = (int)((float)fc * rV);                            // fc < 2^24: exact up to one unit
        if (j * V > fc) --j;
        else if ((j + 1) * V <= fc) ++j;
        const int w = fc - j * V;
        const int64_t n = (int64_t)b * k_in + j;
        float lp = logp[n * ldl + w];
        if (parts) lp -= lse_s[min(j - jfirst, 3)];
        const int64_t pt = penal ? prev_tok[n] : (int64_t)-1;
        const float base = nll ? nll[n] : 0.f;
        if (pt == EOS) lp = (w == EOS) ? 0.f : NEG_PEN;           // V11.py:291-294
        else if (w == pt) lp = NEG_PEN;                           // V11.py:279-280
        val[e] = f < total ? base + lp : -INFINITY;               // V11.py:297
    }
    // each wave ranks the k best of its 512 candidates; wave 0 then ranks the k best of those 4k
    __shared__ float wv[4 * 64], sv[4 * 64];
    __shared__ int wi[4 * 64], si[4 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int idx[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) idx[e] = (f0 + e * 256 < total) ? f0 + e * 256 : 0x7fffffff;
    wv[wave * 64 + lane] = -INFINITY; wi[wave * 64 + lane] = 0x7fffffff;
    wave_lds_fence();
    wave_topk<EPT>(val, idx, k, sv + wave * 64, si + wave * 64, wv + wave * 64, wi + wave * 64);
    __syncthreads();
    if (wave != 0) return;
    float v2[4];
    int i2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v2[e] = wv[e * 64 + lane]; i2[e] = wi[e * 64 + lane]; }
    const int64_t o = ((int64_t)b * chunks + chunk) * k;
    const int n = wave_topk<4>(v2, i2, k, sv, si, cval + o, cidx + o);
    for (int r = n + lane; r < k; r += 64) { cval[o + r] = -INFINITY; cidx[o + r] = 0x7fffffff; }
}

constexpr int S2_LDS = 4096;             // candidates kept in LDS by stage 2 (more: selection works on the scratch copy)

__global__ __launch_bounds__(256) void beam_stage2_kernel(float* __restrict__ cval, int* __restrict__ cidx,
                                                          int chunks, int k_in, int k, int V, int H,
                                                          float* __restrict__ nll, int64_t* __restrict__ beam,
                                                          int32_t* di_state, int di_host, int max_len, int B,
                                                          const float* __restrict__ h_in, float* __restrict__ h_out,
                                                          int64_t* __restrict__ tok_out, int32_t* __restrict__ n_alive) {
    __shared__ Cand sh[4];
    __shared__ int sel_idx[64];
    const int di = di_state ? __atomic_load_n(di_state, __ATOMIC_RELAXED) : di_host;
    if (di >= max_len || (di_state && di < 1)) return;
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
    constexpr int E2 = 16;                     // fast path: up to 1024 chunk winners, held by ONE wave (16 per lane)
    if (ncand <= 64 * E2) {
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x;
            float v2[E2];
            int i2[E2];
#pragma unroll
            for (int e = 0; e < E2; ++e) {
                const int c = e * 64 + lane;
                v2[e] = c < ncand ? pv[c] : -INFINITY;
                i2[e] = c < ncand ? pi[c] : 0x7fffffff;
            }
            __shared__ float tv[64];
            __shared__ int ti[64];
            const int n = wave_topk<E2>(v2, i2, k, tv, ti, sel_val, sel_idx);       // ranked: slot j = j-th best
            for (int r = n + lane; r < k; r += 64) { sel_idx[r] = 0x7fffffff; sel_val[r] = -INFINITY; }
        }
    } else {
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
    }
    __syncthreads();
    if (threadIdx.x < k) {
        const int j = threadIdx.x;
        const int f = sel_idx[j];
        const int64_t w = f % V;
        beam[((int64_t)di * B + b) * k + j] = w;                                // V11.py:306
        beam[((int64_t)(max_len + di) * B + b) * k + j] = f / V;                // parent hypothesis (V11.py:303,309)
        if (tok_out) tok_out[(int64_t)b * k + j] = w;                           // next step's input words
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
    if (di_state && threadIdx.x == 0) {
        // every block has read di_state[0] before it arrives here; the last one to arrive advances the step
        __threadfence();
        if (atomicAdd(&di_state[1], 1) == B - 1) {
            di_state[1] = 0;
            __atomic_store_n(di_state, di + 1, __ATOMIC_RELAXED);
        }
    }
}

int64_t vag_beam_scratch_bytes_impl(int64_t B, int64_t k, int64_t V) {
    const int64_t chunks = cdiv64(k * V, CHUNK);
    return B * chunks * k * 8 + 64;
}

int vag_beam_step_launch(float* logp, int64_t ldl, float* nll, int64_t* beam, int64_t di, int32_t* di_state,
                         int64_t max_len, const float* h_in, float* h_out, int64_t* tok_out, int64_t B, int64_t k,
                         int64_t V, int64_t H, int32_t* n_alive, void* scratch, hipStream_t s, const float* parts, int64_t nparts) {
    VAG_CHECK_ARG(logp && nll && beam && h_in && h_out && n_alive && scratch);
    VAG_CHECK_ARG(!parts || (nparts > 0 && V >= CHUNK));         // (a chunk then spans at most two rows)
    VAG_CHECK_ARG(B > 0 && k > 0 && k <= 64 && V > 0 && H > 0 && ldl >= V && max_len > 0);
    VAG_CHECK_ARG(di_state || (di >= 0 && di < max_len));
    const int k_in = (!di_state && di == 0) ? 1 : (int)k;
    const int64_t total = (int64_t)k_in * V;
    VAG_CHECK_ARG(total < (1ll << 24) && total >= k);          // stage 1 splits flat indices with a float reciprocal
    const int chunks = (int)cdiv64(total, CHUNK);
    float* cval = reinterpret_cast<float*>(scratch);
    int* cidx = reinterpret_cast<int*>(cval + B * cdiv64(k * V, CHUNK) * k);
    hipLaunchKernelGGL(beam_stage1_kernel, dim3((unsigned)chunks, (unsigned)B), dim3(256), 0, s, logp, ldl, nll, beam,
                       di_state, (int)di, (int)max_len, (int)B, k_in, (int)k, (int)V, cval, cidx, n_alive, parts, (int)nparts);
    VAG_LAUNCH_CHECK();
    hipLaunchKernelGGL(beam_stage2_kernel, dim3((unsigned)B), dim3(256), 0, s, cval, cidx, chunks, k_in, (int)k, (int)V,
                       (int)H, nll, beam, di_state, (int)di, (int)max_len, (int)B, h_in, h_out, tok_out, n_alive);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// V11.py:315-324: force EOS in the last row, normalise by the number of tokens > 3, pick the best hypothesis.
// `steps` rows of history were written (fewer than max_len after an early stop; the rest reads as padding 0).
// Thread j walks the back-pointers of final hypothesis j (steps dependent 8-byte loads, once per decode).
constexpr int FIN_LDS = 4096;            // (word, parent) pairs of one sentence's history kept in LDS: steps * k <= 4096
__global__ __launch_bounds__(64) void beam_finish_kernel(const float* __restrict__ nll, const int64_t* __restrict__ beam,
                                                         int max_len, int steps, int B, int k, int64_t* __restrict__ out,
                                                         float* __restrict__ best) {
    const int b = blockIdx.x, j = threadIdx.x;
    const int64_t* par = beam + (int64_t)max_len * B * k;
    // the sentence's history into LDS first (coalesced rows of k words / k parents per step): the walks below are chains of
    // `steps` dependent reads -- from global memory 80 steps took ~52 us per call (a memory round trip each), from LDS ~5
    __shared__ int hw[FIN_LDS], hp[FIN_LDS];
    const bool lds = steps * k <= FIN_LDS;
    if (lds) {
        for (int e = j; e < steps * k; e += 64) {
            const int t = e / k, p = e - t * k;
            const int64_t o = ((int64_t)t * B + b) * k + p;
            hw[e] = (int)beam[o];
            hp[e] = (int)par[o];
        }
        __syncthreads();
    }
    float sc = -INFINITY;
    if (j < k) {
        int len = 0, p = j;
        for (int t = steps - 1; t >= 0; --t) {
            const int64_t o = ((int64_t)t * B + b) * k + p;
            const int w = lds ? hw[t * k + p] : (int)beam[o];
            if (t < max_len - 1) len += w > 3;             // row max_len-1 is forced to EOS (= 3), which never counts
            p = lds ? hp[t * k + p] : (int)par[o];
        }
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
    int64_t* row = out + (int64_t)b * max_len;
    for (int t = steps + j; t < max_len; t += 64) row[t] = 0;
    if (j == 0) {
        int p = bi;
        for (int t = steps - 1; t >= 0; --t) {
            const int64_t o = ((int64_t)t * B + b) * k + p;
            row[t] = lds ? (int64_t)hw[t * k + p] : beam[o];
            p = lds ? hp[t * k + p] : (int)par[o];
        }
        row[max_len - 1] = EOS;
        if (best) best[b] = bv;
    }
}

int vag_beam_finish_launch(const float* nll, const int64_t* beam, int64_t max_len, int64_t steps, int64_t B, int64_t k,
                           int64_t* out, float* best, hipStream_t s) {
    VAG_CHECK_ARG(nll && beam && out && max_len > 0 && steps > 0 && steps <= max_len && B > 0 && k > 0 && k <= 64);
    hipLaunchKernelGGL(beam_finish_kernel, dim3((unsigned)B), dim3(64), 0, s, nll, beam, (int)max_len, (int)steps, (int)B,
                       (int)k, out, best);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}
