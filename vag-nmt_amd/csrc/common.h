// Shared device helpers for the VAG-NMT gfx950 kernels.  CDNA4 only: 64-lane waves, MFMA f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define VAG_OK 0
#define VAG_EINVAL (-22)
#define VAG_ENOSYS (-38)      // librccl could not be loaded (vag_comm_*)

#define VAG_CHECK_ARG(cond)                     \
    do {                                        \
        if (!(cond)) return VAG_EINVAL;         \
    } while (0)

// After a launch: surface launch-configuration errors as a positive hipError_t.
#define VAG_LAUNCH_CHECK()                                  \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return (int)e__;             \
    } while (0)

#define VAG_TRY(expr)                  \
    do {                               \
        int r__ = (expr);              \
        if (r__ != 0) return r__;      \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// Wave-wide all-reduce without the LDS pipe (round 4): four row_ror DPP steps inside the rows of 16 lanes, then gfx950's
// v_permlane16_swap and v_permlane32_swap (odd rows / the upper half of one copy <-> even rows / the lower half of the other) with one
// combine each: ~12 vector instructions.  Six __shfl_xor steps are six DEPENDENT ds_bpermute of ~64 cycles each (~400 cycles), which
// the persistent decoder kernels pay twice per time step on their critical path (softmax: max, then sum).
template <bool MAX>
__device__ __forceinline__ float wave_allreduce(float v) {
#define VAG_DPP_ROR(N) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + (N), 0xf, 0xf, false))
    { const float o = VAG_DPP_ROR(8); v = MAX ? fmaxf(v, o) : v + o; }
    { const float o = VAG_DPP_ROR(4); v = MAX ? fmaxf(v, o) : v + o; }
    { const float o = VAG_DPP_ROR(2); v = MAX ? fmaxf(v, o) : v + o; }
    { const float o = VAG_DPP_ROR(1); v = MAX ? fmaxf(v, o) : v + o; }
#undef VAG_DPP_ROR
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    v = MAX ? fmaxf(a, b) : a + b;
    a = v; b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return MAX ? fmaxf(a, b) : a + b;
}
// sum over the four lanes of a quad (DPP quad_perm [1,0,3,2] then [2,3,0,1]: two VALU instructions, no LDS pipe)
__device__ __forceinline__ float quad_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) { return wave_allreduce<false>(v); }
__device__ __forceinline__ float wave_max(float v) { return wave_allreduce<true>(v); }
// tanh / sigmoid on the v_exp_f32 path; absolute error ~1e-7, saturates cleanly at +-1 / 0,1.
// v_rcp_f32 is accurate to 1 ulp; one v_exp_f32 + one v_rcp_f32 per activation instead of an IEEE division.
__device__ __forceinline__ float vag_tanh(float x) {
    float e = __expf(2.0f * x);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}
__device__ __forceinline__ float vag_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// Counter-based dropout: keep-mask and multiplier are a pure function of (seed, stream, index),
// so the backward pass recomputes them instead of storing masks.  splitmix64 finaliser.
__device__ __host__ __forceinline__ uint64_t vag_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// rng[0] = seed, rng[1] = step counter (bumped once per training step on the device).
__device__ __forceinline__ float vag_drop_mul(const uint64_t* rng, uint32_t stream_id, uint64_t idx, float p) {
    if (rng == nullptr || p <= 0.0f) return 1.0f;
    uint64_t key = vag_mix64(rng[0] ^ (rng[1] * 0xD1342543DE82EF95ull) ^ ((uint64_t)stream_id << 56));
    uint64_t r = vag_mix64(key + idx);
    float u = (float)(r >> 40) * (1.0f / 16777216.0f);   // 24 random bits -> [0,1)
    return (u >= p) ? 1.0f / (1.0f - p) : 0.0f;
}

// ---- 2-byte storage mode: tensors the recurrences stream at every time step (their weights, the attention keys) may be
// kept as fp16 in memory; arithmetic and accumulation stay fp32.  Pointers keep the type `const float*` in the argument
// structs and are re-cast where the template flag says fp16.
typedef _Float16 vag_half;
__device__ __forceinline__ float h16_lo(unsigned p) { return (float)__builtin_bit_cast(vag_half, (unsigned short)(p & 0xffffu)); }
__device__ __forceinline__ float h16_hi(unsigned p) { return (float)__builtin_bit_cast(vag_half, (unsigned short)(p >> 16)); }
// 4 consecutive elements (index `elem` .. `elem`+3, elem % 4 == 0) of a tensor stored as fp32 or fp16
template <bool XH> __device__ __forceinline__ float4 ld4_any(const float* base, int64_t elem) {
    if (XH) {
        const uint2 p = *reinterpret_cast<const uint2*>(reinterpret_cast<const vag_half*>(base) + elem);
        return make_float4(h16_lo(p.x), h16_hi(p.x), h16_lo(p.y), h16_hi(p.y));
    }
    return *reinterpret_cast<const float4*>(base + elem);
}
template <bool XH> __device__ __forceinline__ float ld1_any(const float* base, int64_t elem) {
    if (XH) return (float)reinterpret_cast<const vag_half*>(base)[elem];
    return base[elem];
}

// Process-wide debug / tuning options (api.hip: vag_set_option).  Defaults are what ships; the parity tests flip
// gemm_f32mfma (exact-f32 reference products), head_chunk / head_fuse (chunked output head at small sizes).
struct VagOptions {
    int gemm_f32mfma = 0;        // 1: every product on the f32-input MFMA kernels (no bf16 split)
    int gemm_big = 1;            // 0: the one-plane products of the 2-byte storage mode stay on 128 x 128 tiles (round-3 kernels)
    int gemm_slabs = 1;          // 0: split-K slices add into C with atomics even where a slab scratch is at hand (rounds 1-5)
    int gemm_nogroup = 0;        // 1: products inside a group bracket are launched one by one
    int gemm_force_tile = 0;     // 64 / 128 together with gemm_force_splitk >= 1: override the tile / split-K choice
    int gemm_force_splitk = 0;
    int gemm_debug = 0;          // 1: print the choice per product
    int64_t head_chunk = -1;     // rows per chunk of the output head (-1: automatic, 0: never chunk)
    int head_fuse = 1;           // 0: the chunked head recomputes its chunks in backward instead of finishing them in forward
    int head_bf16_grads = 1;     // 2-byte storage mode: one bf16 plane in the head's two vocabulary-sized gradient products
    int persistent = 1;          // 0: the recurrences always run as chains of per-step launches (persist.hip off)
    int persistent_dec_bwd = 1;  // 0: only the decoder's backward recurrence stays a launch chain
    int persistent_enc_bwd = 1;  // 0: only the encoder's backward recurrence stays a launch chain (data parallelism: the one persistent
                                 // kernel that shares its window with a collective's kernels -- bench.py --dp-encoder-chain)
    int attn_dot_reg = 1;        // 0: the per-step score / d-alpha reductions of the launch chains keep the round-2 kernel (q re-read per position)
    int free_persistent = 1;     // 0: free-running decoder steps (and greedy decoding) stay chains of per-step launches
    int s16_one_plane = 1;       // 2-byte storage mode of the step driver: forward products on ONE fp16 plane, gradient products on
                                 // ONE bf16 plane (0: two bf16 planes everywhere, as the operators on their own use)
    int head_bf16_dlogits = 1;   // 2-byte storage mode, chunked head: d(logits) of a chunk is written and read as bf16 (0: fp32 in place)
    int leaf_queue = 1;          // 0: the small weight-gradient products of the VSE / initial-state backward go out one by one (round 4)
    int attn_row = 1;            // 0: the visual-grounding attention as separate scores / softmax / context launches (rounds 1-4)
    int loss_ride = 1;           // 0: the loss reduction of a training step is its own launch (rounds 1-4)
    int dec_xcd_map = 33;        // slice placement of the persistent decoder kernels (persist.hip: dec_slice_map), forward | backward << 4;
                                 // 0: block order (rounds 3-5), 1 / 2 / 3: eight / two / four consecutive slices per XCD
    int step_fork = 0;           // side-stream branches inside vag_train_step, bit 0: the image projection beside the encoder, bit 1: the
                                 // held-back weight-gradient leaves beside the encoder's backward.  Off: measured SLOWER
    int64_t persist_spin_limit = 0;   // > 0: polls before a persistent kernel's wait gives up (default 2^19); tests force a give-up with 1
    int persist_timing = 0;      // 1: HIP events around the recurrence kernels of eager launches (vag_recurrence_time)
    int64_t dec_bwd_stamps = 0;  // the same for the persistent decoder backward
    int64_t dec_stamps = 0;      // device address of Tt x 8 uint64 for the persistent decoder's phase timestamps (0: none)
};
VagOptions& vag_opt();

enum { VAG_ACT_NONE = 0, VAG_ACT_TANH = 1 };
enum { VAG_DROP_ENC_EMB = 1, VAG_DROP_ENC_CTX = 2, VAG_DROP_DEC_OUT = 3 };

// ---- internal kernel-launching helpers shared across translation units (gemm.hip) ----
// C[M,N] = act(alpha * op(A) op(B) + beta * C + bias[n]);  A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn].
int vag_gemm_launch(int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t sam, int64_t sak,
                    const float* B, int64_t sbk, int64_t sbn, float beta, float* C, int64_t ldc,
                    const float* bias, int act, hipStream_t stream, int c_half = 0,    // c_half: C stored as fp16 (beta = 0)
                    float* rowsum = nullptr,      // rowsum[m] += sum_k A(m,k) (A outer-contiguous): bias gradient of g_W += dY^T X
                    int a_bf16 = 0);              // A stored as bf16 (one-plane bf16 kernel, ungrouped): the 2-byte mode's d(logits)
// out[m,n] = act(sum_k A[m,k] W[n,k] + bias[n] + addend[m,n]);  M small (decode/recurrent steps).
void vag_gemm_set_planes(int planes);
void vag_gemm_prezeroed_set(int slot, const float* p);      // gemm.hip: an output the caller has zeroed (a sliced overwrite skips its fill), used once
int vag_gemm_launch_planes(int planes, int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t sam, int64_t sak,
                           const float* B, int64_t sbk, int64_t sbn, float beta, float* C, int64_t ldc, hipStream_t stream,
                           int a_bf16 = 0);      // planes 3: bf16x6 (default), 2: bf16x3 (2-byte storage mode), calling thread
int64_t vag_logits_parts_count(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* W, int64_t ldw);
int vag_logits_parts_launch(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* W, int64_t ldw,
                            const float* bias, float* out, int64_t ldo, float* parts, hipStream_t stream);
int vag_skinny_launch(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* W, int64_t ldw,
                      const float* bias, const float* addend, int64_t ldadd, float* out, int64_t ldo, int act,
                      hipStream_t stream, bool w16 = false);      // w16: W is stored as fp16 (2-byte storage mode)
// out[n] += sum_m X[m*ld + n]
int vag_attn_dot_side_launch(int mode, const float* x, const float* q, int64_t ldq, const float* v, const float* mask,
                             const float* addend, int64_t N, int64_t Ts, int64_t W, float* out, int64_t M, int64_t Np,
                             int64_t K, const float* A, int64_t lda, const float* Wt, int64_t ldw, const float* pbias,
                             const float* padd, float* P, int64_t ldp, hipStream_t stream, bool s16 = false);   // s16: x and Wt fp16
int vag_skinny_gather_launch(int64_t M, int64_t N, int64_t K, const float* table, int64_t ldt, const int64_t* idx, const float* W,
                             int64_t ldw, const float* bias, float* out, int64_t ldo, float* gathered, int64_t ldg,
                             hipStream_t stream);
int vag_skinny3_launch(int64_t M, int64_t N, const float* const* A, const int64_t* lda, const float* const* W, const int64_t* ldw,
                       const int64_t* K, const float* const* bias, float* out, int64_t ldo, int act, const uint64_t* rng, int sid,
                       float p, int64_t drop_idx0, hipStream_t stream, const float* addend = nullptr, int64_t ldadd = 0,
                       const float* addend2 = nullptr, const int64_t* idx2 = nullptr, int64_t ldadd2 = 0);
int vag_skinny_batched_launch(int64_t nb, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, int64_t bsA,
                              const float* W, int64_t ldw, int64_t bsW, float* out, int64_t ldo, int64_t bsO,
                              hipStream_t stream);
int vag_skinny_nn_launch(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                         float beta, float* C, int64_t ldc, hipStream_t stream);
// queue the qualifying products issued between begin and end, one queue per operand layout, each launched as one grouped
// grid at end; nothing queued may be read or overwritten by work enqueued before vag_gemm_group_end.  Brackets nest: an
// inner end flushes everything queued so far
void vag_gemm_group_begin();
int vag_gemm_group_end(hipStream_t stream);
void vag_gemm_group_abort();
// scope guard: an early error return inside a bracket must not leave the queue open
struct VagGemmGroup {
    bool open = true;
    explicit VagGemmGroup(bool enable = true) : open(enable) { if (enable) vag_gemm_group_begin(); }
    int end(hipStream_t s) { if (!open) return VAG_OK; open = false; return vag_gemm_group_end(s); }
    ~VagGemmGroup() { if (open) vag_gemm_group_abort(); }
    VagGemmGroup(const VagGemmGroup&) = delete;
    VagGemmGroup& operator=(const VagGemmGroup&) = delete;
};
void vag_leaf_begin();                 // hold back small rank-B weight-gradient products (+ their column sums): gemm.hip
int vag_leaf_flush(hipStream_t stream);  // ... and launch them as one grid
void vag_leaf_abort();
int vag_colsum_launch(const float* X, int64_t M, int64_t N, int64_t ld, float* out, hipStream_t stream);
int vag_colsum3_launch(const float* X, int64_t M, int64_t N, int64_t ld, float* out, float* out2, float* out3,
                       hipStream_t stream);
// out[N,M] = in[M,N]^T
int vag_transpose_launch(const float* in, int64_t M, int64_t N, float* out, hipStream_t stream);

// ---- fused GRU cell step (gemm.hip) ----
struct GruSide {
    const float* A;      // (M,K) operand of the projection computed here (h_prev when comp_hidden, else x)
    const float* W;      // (3H,K) weight of that projection, gate order r,z,n
    const float* bias;   // (3H) bias of that projection (may be NULL)
    const float* other;  // (M,3H) the other projection, bias already added
    const int64_t* other_idx;   // NULL, or (M) row indices into `other` (decoding: the input projection of every vocabulary entry
                                // is a table, a row's token picks its line)
    const float* hprev;  // (M,H) previous hidden state
    float* hout;         // (M,H) new hidden state (rows past their length keep hprev)
    float* out2;         // optional second copy with row stride ld2; zero for rows past their length
    float* save;         // optional [4][M][H]: r, z, n, (W_hn h + b_hn) for the backward pass
    int t;               // time index compared against lengths[]
};
struct GruStepArgs {
    GruSide s[2];        // s[1] only used when two independent cells run in one launch (encoder directions)
    int64_t lda, ldw, ldother, ldh, ld2;
    int M, K, H;
    const int* lengths;  // device int32[M] or NULL
    int comp_hidden;     // 1: computed projection is the hidden one (W_hh h), 0: the input one (W_ih x)
};
int vag_gru_step_launch(const GruStepArgs& a, int nz, hipStream_t stream, bool w16 = false);      // w16: s[].W is fp16

// ---- fused "gradient w.r.t. a hidden state, then the GRU cell backward it feeds" (gemm.hip) ----
// dh[m,j] = sum_k A[m,k] WT[j,k] + addend[m,j]   (WT = transposed weights, one row per hidden unit)
// then, if has_cell, the elementwise backward of the GRU cell whose OUTPUT gradient dh is:
//   dh += dropout(dh_add[m*ld_add + j]);  (dgi, dgh, dh_direct) = cell_bwd(dh, saved gates, hprev)
// Rows past their length (encoder): dgi = dgh = 0 and dh_direct = dh (state was carried through).
struct GruBwdStepSide {
    const float* A;          // (M,K) row stride lda
    const float* WT;         // (H,K) row stride ldw
    const float* addend;     // (M,H) contiguous, may be NULL
    const float* dh_add;     // optional upstream gradient of the cell's output
    int64_t drop_idx0;       // dropout index of dh_add element (m,j) = m*ld_add + drop_idx0 + j
    const float* save;       // [4][M][H] gates of the target cell
    const float* hprev;      // (M,H) row stride ldh: the target cell's previous state
    float* dgi;              // (M,3H) row stride ldgi
    float* dgh;              // (M,3H) row stride ldgh
    float* dh_direct;        // (M,H): z * dh
    float* dh_out;           // (M,H): written when !has_cell
    int t;
};
struct GruBwdStepArgs {
    GruBwdStepSide s[2];
    int64_t lda, ldw, ld_add, ldh, ldgi, ldgh;
    int M, K, H;
    const int* lengths;
    const uint64_t* rng; int sid; float p;
    int has_cell;
};
int vag_gru_bwd_step_launch(const GruBwdStepArgs& a, int nz, hipStream_t stream, bool w16 = false);   // w16: s[].WT is fp16
