// Persistent recurrence kernels: a whole GRU recurrence -- every time step of both encoder directions -- in ONE
// launch, with the recurrent weights resident on chip for all steps and the hidden state exchanged between workgroups
// through global memory (layers/Encoder.py:55-60: nn.GRU over the packed sequence).
//
// Why: as a chain of launches a step costs ~7.4 us for ~3.8 MB of algorithmic bytes -- a dependent-kernel boundary
// (1.45 us) plus one round of per-CU ingest in which every workgroup pulls its 48 weight rows (96 KB) out of L2 / the
// Infinity Cache again, although they never change.  Here:
//   * grid = 2 directions x ceil(B/16) row tiles x H/16 unit slices (256 workgroups at B = 64, H = 512: one per CU, all
//     resident: 84 KB of LDS are declared so that two never share a CU);
//   * a workgroup owns 16 hidden units x 3 gates x 16 batch rows for ALL steps.  Its slice of W_hh (48 rows x H) is split
//     once, exactly, into three bf16 planes held in REGISTERS in MFMA operand layout (72 VGPRs at H = 512); a step's
//     product is 36 v_mfma_f32_16x16x32_bf16 per wave (six bf16 products per fp32 product: fp32-grade, as gemm.hip) on
//     the freshly split hidden-state rows, K split over the 8 waves, one LDS reduction;
//   * the only per-step global traffic of a workgroup is its 16 x H input rows of h (32 KB, L2 / Infinity Cache), the
//     16 x 48 input projections, and its 16 x 16 outputs;
//   * hand-off (MI355X_MICROARCH.md, inter-workgroup visibility, table row 1): h is stored write-through (sc1, 16 bytes per
//     lane), the storing wave drains (s_waitcnt vmcnt(0)), ONE lane adds to the counter of (direction, row tile, step);
//     a consumer polls that counter with relaxed agent-scope loads from one lane (bounded: a give-up sets an error word
//     and lets the grid drain), the workgroup's barrier follows, and EVERY load of h is an sc1 load to registers.  No
//     fences.  A step waits only for the 32 workgroups of its own direction and row tile.
#include "kernels.h"
#include "gemm_shared.h"
#include <algorithm>
#include <atomic>
#include <mutex>

namespace {

typedef unsigned __attribute__((address_space(1))) gu32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr unsigned SPIN_LIMIT = 1u << 19;       // polls before giving up (~a second; a healthy wait is a few hundred polls);
                                                // vag_set_option("persist_spin_limit", n) overrides it (tests force a give-up)
// Waits that gave up since the last vag_persistent_timeouts() (results of such a launch are void): the kernels assume that
// every workgroup of the grid is resident at once -- one per CU, checked on the host against the CU count -- and a spin is
// bounded so that a device on which that does not hold (CU masks, a partitioned GPU, another process's or a collective's
// kernels squatting on CUs) drains instead of hanging.  A give-up also sets the launch's GUARD: two caller-owned words
// {void flag, give-up count} (vag_step_cfg.guard / vag_set_operator_guard; the step driver keeps them in its optimiser scratch,
// VAG_ADAM_SCRATCH_GUARD_OFFSET).  adam_prep_kernel (optim.hip) reads the flag of ITS driver and skips the update of a step whose
// recurrences gave up a wait, and the encoder's embedding scatter -- the last launch of a backward pass -- turns it into a
// non-finite gradient entry so that under data parallelism EVERY replica sees it after the all-reduce and skips the same step
// (train.py:44-49 semantics are kept for every step that is applied; a void gradient is never applied).  Two drivers on one
// device (two models, a trainer and a decoder) therefore cannot void each other's steps.  Launches without a guard of their own
// (operators called one by one, decoding) share the process-wide pair below; g_persist_timeouts counts every give-up of the
// process for vag_persistent_timeouts().
__device__ unsigned g_persist_timeouts;
__device__ unsigned g_persist_guard[2];
__device__ __forceinline__ void note_timeout(unsigned* err, unsigned* guard) {
    __hip_atomic_store((gu32*)err, 1u, RLX_AGENT);
    __hip_atomic_store((gu32*)guard, 1u, RLX_AGENT);
    atomicAdd(guard + 1, 1u);
    atomicAdd(&g_persist_timeouts, 1u);
}

struct EncPArgs {
    const float* xp;            // (Ts, B, 6H): input projections [fwd r z n | rev r z n], biases included
    const float* W[2];          // (3H, H) recurrent weights per direction
    const float* bias[2];       // (3H) b_hh
    const int* lengths;         // (B)
    float* hst;                 // [2][Ts+1][B][H]; slot 0 (zeros) is written here
    float* gates;               // [2][Ts][4][B][H]
    float* enc;                 // (B, Ts, 2H)
    unsigned* cnt;              // [2][RT][Ts], zero on entry
    unsigned* err;              // 1 word, set when a wait gave up
    unsigned* guard;            // {void flag, give-up count} of the driver this launch belongs to
    unsigned spin;              // polls before a wait gives up
    const uint64_t* rng;        // context dropout (Encoder.py:63-64) applied to enc as it is written (NULL / p_ctx = 0: none)
    float p_ctx;
    int B, Ts, H, RT, CS;       // RT: row tiles of the whole batch (the counters are indexed by the global row tile)
    int rt0, RTP;               // this launch: row tiles [rt0, rt0 + RTP) (a batch wider than the chip goes in passes of row tiles)
};

// N x 32 floats of one row read back from another workgroup's sc1 stores: 2 N sc1 loads of 16 bytes (k-step s: floats
// [32 s, 32 s + 8) of p) and their wait in ONE asm statement (cdna_hip_programming.md 5.7 item 1, form (i)).  Inline asm
// because __builtin_amdgcn_raw_buffer_load_b128 / _b64 come out as ONE buffer_load_dword whose value is used for every
// element with this toolchain (ROCm 7.2 hipcc, gfx950; checked in the .s), and __hip_atomic_load stops at 8 bytes.
template <int N> __device__ __forceinline__ void ld_rows_sc1(const float* p, float4 (&a)[N], float4 (&b)[N]);
template <> __device__ __forceinline__ void ld_rows_sc1<1>(const float* p, float4 (&a)[1], float4 (&b)[1]) {
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a[0]), "=&v"(b[0]) : "v"(p) : "memory");
}
template <> __device__ __forceinline__ void ld_rows_sc1<2>(const float* p, float4 (&a)[2], float4 (&b)[2]) {
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %4, off offset:128 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:144 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a[0]), "=&v"(b[0]), "=&v"(a[1]), "=&v"(b[1]) : "v"(p) : "memory");
}
template <> __device__ __forceinline__ void ld_rows_sc1<3>(const float* p, float4 (&a)[3], float4 (&b)[3]) {
    asm volatile("global_load_dwordx4 %0, %6, off sc1\n\tglobal_load_dwordx4 %1, %6, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %6, off offset:128 sc1\n\tglobal_load_dwordx4 %3, %6, off offset:144 sc1\n\t"
                 "global_load_dwordx4 %4, %6, off offset:256 sc1\n\tglobal_load_dwordx4 %5, %6, off offset:272 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a[0]), "=&v"(b[0]), "=&v"(a[1]), "=&v"(b[1]), "=&v"(a[2]), "=&v"(b[2]) : "v"(p) : "memory");
}
template <> __device__ __forceinline__ void ld_rows_sc1<4>(const float* p, float4 (&a)[4], float4 (&b)[4]) {
    asm volatile("global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %8, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %8, off offset:128 sc1\n\tglobal_load_dwordx4 %3, %8, off offset:144 sc1\n\t"
                 "global_load_dwordx4 %4, %8, off offset:256 sc1\n\tglobal_load_dwordx4 %5, %8, off offset:272 sc1\n\t"
                 "global_load_dwordx4 %6, %8, off offset:384 sc1\n\tglobal_load_dwordx4 %7, %8, off offset:400 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a[0]), "=&v"(b[0]), "=&v"(a[1]), "=&v"(b[1]), "=&v"(a[2]), "=&v"(b[2]), "=&v"(a[3]), "=&v"(b[3])
                 : "v"(p) : "memory");
}
template <> __device__ __forceinline__ void ld_rows_sc1<6>(const float* p, float4 (&a)[6], float4 (&b)[6]) {
    asm volatile("global_load_dwordx4 %0, %12, off sc1\n\tglobal_load_dwordx4 %1, %12, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %12, off offset:128 sc1\n\tglobal_load_dwordx4 %3, %12, off offset:144 sc1\n\t"
                 "global_load_dwordx4 %4, %12, off offset:256 sc1\n\tglobal_load_dwordx4 %5, %12, off offset:272 sc1\n\t"
                 "global_load_dwordx4 %6, %12, off offset:384 sc1\n\tglobal_load_dwordx4 %7, %12, off offset:400 sc1\n\t"
                 "global_load_dwordx4 %8, %12, off offset:512 sc1\n\tglobal_load_dwordx4 %9, %12, off offset:528 sc1\n\t"
                 "global_load_dwordx4 %10, %12, off offset:640 sc1\n\tglobal_load_dwordx4 %11, %12, off offset:656 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a[0]), "=&v"(b[0]), "=&v"(a[1]), "=&v"(b[1]), "=&v"(a[2]), "=&v"(b[2]), "=&v"(a[3]), "=&v"(b[3]),
                   "=&v"(a[4]), "=&v"(b[4]), "=&v"(a[5]), "=&v"(b[5])
                 : "v"(p) : "memory");
}
// rows handed off with marks (see tag1): loaded until every word of this wave's share carries its mark
template <int N>
__device__ __forceinline__ void ld_rows_tagged(const float* p, float4 (&a)[N], float4 (&b)[N], unsigned spin, bool& dead,
                                               unsigned* err, unsigned* guard);
__device__ __forceinline__ void st_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float4 v) {
    const u32x4 u = {__builtin_bit_cast(unsigned, v.x), __builtin_bit_cast(unsigned, v.y), __builtin_bit_cast(unsigned, v.z),
                     __builtin_bit_cast(unsigned, v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(u, r, byte_off, 0, 16);
}
__device__ __forceinline__ void st_sc1_f4(float* p, float4 v) {
    const f32x4 x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(x) : "memory");
}
// Tagged hand-off (round 5).  A handed-off fp32 word carries its own "written" mark: its lowest significand bit is forced to 1
// by the producer (at most one ulp: 6e-8 relative, below what the bf16x3 split keeps of it), and the buffer is all zeros when
// the launch starts (the step's prologue launch / the launch function zero it).  The consumer still learns from the counter
// WHEN to look, but the producer no longer drains its stores before it signals (s_waitcnt vmcnt(0) behind a write-through
// store is a round trip to the memory side, ~0.8 us on the critical path of every hand-off): a word whose mark is not there
// yet is simply loaded again.  The only assumptions are R2's (cdna_hip_programming.md, Guideline 16): an aligned 4-byte word
// is never torn, and an sc1 load eventually observes an sc1 store.
#ifndef VAG_TAGGED
#define VAG_TAGGED 1
#endif
__device__ __forceinline__ float tag1(float x) { return VAG_TAGGED ? __uint_as_float(__float_as_uint(x) | 1u) : x; }
__device__ __forceinline__ float4 tag4(float4 v) { return make_float4(tag1(v.x), tag1(v.y), tag1(v.z), tag1(v.w)); }
__device__ __forceinline__ unsigned tagbits(float4 v) {
    return __float_as_uint(v.x) & __float_as_uint(v.y) & __float_as_uint(v.z) & __float_as_uint(v.w);
}
template <int N>
__device__ __forceinline__ void ld_rows_tagged(const float* p, float4 (&a)[N], float4 (&b)[N], unsigned spin, bool& dead,
                                               unsigned* err, unsigned* guard) {
    for (unsigned tries = 0;; ++tries) {
        ld_rows_sc1<N>(p, a, b);
        if (!VAG_TAGGED) return;
        unsigned m = 1u;
#pragma unroll
        for (int s = 0; s < N; ++s) m &= tagbits(a[s]) & tagbits(b[s]);
        if (__all((m & 1u) != 0u) || dead) return;
        if (tries > spin) { if ((threadIdx.x & 63) == 0) note_timeout(err, guard); dead = true; return; }
    }
}
__device__ __forceinline__ void st_sc1_f1(float* p, float v) {
    asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}
// Row loads in a quad-contiguous lane mapping.  The MFMA wants lane (row fr = lane & 15, k-group fg = lane >> 4), but a wave
// request in which the four lanes of every quad touch four different rows costs the address unit one line request per lane
// (64 per instruction).  So lane l LOADS row ld_row(l), k-group l & 3 -- the quad reads four 16-byte pieces of ONE 128-byte
// line, 16 line requests per instruction -- and every loaded dword is then moved to the MFMA's lane through ds_bpermute (no
// LDS storage): the MFMA lane (fr, fg) takes what lane 16 (fr & 3) + 4 (fr >> 2) + fg loaded.  Same trick as the launch-chain
// kernels' skinny_xpose (gemm.hip).
__device__ __forceinline__ int ld_row(int lane) { return 4 * ((lane >> 2) & 3) + (lane >> 4); }
__device__ __forceinline__ int ld_src4(int lane) { return 4 * (16 * (lane & 3) + 4 * ((lane & 15) >> 2) + (lane >> 4)); }
__device__ __forceinline__ float4 perm4(float4 v, int src4) {
    float4 o;
    o.x = __int_as_float(__builtin_amdgcn_ds_bpermute(src4, __float_as_int(v.x)));
    o.y = __int_as_float(__builtin_amdgcn_ds_bpermute(src4, __float_as_int(v.y)));
    o.z = __int_as_float(__builtin_amdgcn_ds_bpermute(src4, __float_as_int(v.z)));
    o.w = __int_as_float(__builtin_amdgcn_ds_bpermute(src4, __float_as_int(v.w)));
    return o;
}
// 8 consecutive floats -> three bf16x8 planes
__device__ __forceinline__ void split8(const float4 a, const float4 b, bf16x8 (&p)[3]) {
    unsigned q[3][4];
    split3(a.x, a.y, q[0][0], q[1][0], q[2][0]);
    split3(a.z, a.w, q[0][1], q[1][1], q[2][1]);
    split3(b.x, b.y, q[0][2], q[1][2], q[2][2]);
    split3(b.z, b.w, q[0][3], q[1][3], q[2][3]);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const u32x4 t = {q[i][0], q[i][1], q[i][2], q[i][3]};
        p[i] = __builtin_bit_cast(bf16x8, t);
    }
}
// six products a_i b_j, i + j <= 4 (0-based planes: i + j <= 2), smallest first
__device__ __forceinline__ f32x4 mma6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
    return acc;
}

// KS: k-steps of 32 per wave (H = 256 * KS)
template <int KS>
__global__ __launch_bounds__(512, 1) void enc_fwd_persistent_kernel(EncPArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[21504];      // 84 KB: [0, 6144) reduction; the rest keeps the CU to ourselves
    const int wg = blockIdx.x;
    const int cs = wg % a.CS, rt = a.rt0 + (wg / a.CS) % a.RTP, d = wg / (a.CS * a.RTP);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int H = a.H, B = a.B, Ts = a.Ts;
    const int m0 = rt * 16, u0 = cs * 16;
    const int fr = lane & 15, fg = lane >> 4;            // fragment row (unit / batch row) and k-group of this lane
    const int kbase = wave * (H >> 3);
    const int64_t BH = (int64_t)B * H;

    // ---- this workgroup's slice of W_hh as bf16 planes in registers: wf[s][gate][plane], A operand of D[unit][batch row]
    bf16x8 wf[KS][3][3];
    {
        const float* W = a.W[d];
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float* p = W + (int64_t)(j * H + u0 + fr) * H + kbase + 32 * s + 8 * fg;
                split8(*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4), wf[s][j]);
            }
    }
    // ---- epilogue threads (wave 0): lane = (batch row lane >> 2, unit quad lane & 3): units eu .. eu + 3 of row em.  A quad
    // then loads / stores four consecutive 16-byte pieces of ONE row (the accumulator layout -- row lane & 15, quad lane >> 4 --
    // would put four rows into every quad: four line requests per quad in each of the ~10 loads and stores of a step); the
    // reduction reads the other lane's LDS slot instead
    const int erow = lane >> 2, equad = lane & 3, eslot = erow + 16 * equad;
    const int em = m0 + erow, eu = u0 + 4 * equad;
    const bool eok = wave == 0 && em < B;
    float4 bb[3] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
    int len = 0;
    if (eok) {
#pragma unroll
        for (int j = 0; j < 3; ++j) bb[j] = *reinterpret_cast<const float4*>(a.bias[d] + j * H + eu);
        len = a.lengths[em];
    }
    float4 hp = make_float4(0.f, 0.f, 0.f, 0.f);           // this thread's previous state (step 0: zeros)

    float* hs = a.hst + (int64_t)d * (Ts + 1) * BH;
    if (eok) *reinterpret_cast<float4*>(hs + (int64_t)em * H + eu) = hp;        // slot 0 (zeros): the backward pass reads it as h_prev
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(hs, 0, (unsigned)((int64_t)(Ts + 1) * BH * 4), 0x00020000);
    gu32* cnt = (gu32*)(a.cnt + ((int64_t)d * a.RT + rt) * Ts);
    const int lrow = min(m0 + ld_row(lane), B - 1);        // the batch row this lane LOADS (quad-contiguous mapping, see ld_row)
    const int src4 = ld_src4(lane);
    float4* red = reinterpret_cast<float4*>(lds);          // [wave][gate][lane]
    bool dead = false;                                      // a wait gave up: stop waiting (results are void, the grid drains)

    for (int k = 0; k < Ts; ++k) {
        const int t = d == 0 ? k : Ts - 1 - k;
        // the other projection of this step (independent of the recurrence): requested before the wait
        float4 xo[3];
        if (eok) {
            const float* xp = a.xp + ((int64_t)t * B + em) * 6 * H + d * 3 * H + eu;
#pragma unroll
            for (int j = 0; j < 3; ++j) xo[j] = *reinterpret_cast<const float4*>(xp + j * H);
        }
        if (k > 0) {
            if (threadIdx.x == 0 && !dead) {
                unsigned spins = 0;
                while (__hip_atomic_load(cnt + (k - 1), RLX_AGENT) < (unsigned)a.CS) {
                    if (++spins > a.spin) { note_timeout(a.err, a.guard); dead = true; break; }
                }
            }
            __syncthreads();
        }
        // ---- h_k rows of this row tile (all H columns; this wave: its K share), sc1 loads, split, six-product MFMAs
        f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        float4 h0[KS], h1[KS];
        if (k > 0) {
            ld_rows_sc1<KS>(hs + ((int64_t)k * B + lrow) * H + kbase + 8 * (lane & 3), h0, h1);
        } else {                                            // the initial state is zero: nothing to read
#pragma unroll
            for (int s = 0; s < KS; ++s) h0[s] = h1[s] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bf16x8 hf[3];
            split8(perm4(h0[s], src4), perm4(h1[s], src4), hf);
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[j] = mma6(wf[s][j], hf, acc[j]);
        }
        // D[unit 4 fg + i][batch row fr] in acc[gate][i]: one 16-byte LDS store per gate, summed over the waves by wave 0
#pragma unroll
        for (int j = 0; j < 3; ++j) red[(wave * 3 + j) * 64 + lane] = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
        __syncthreads();
        if (wave == 0) {
            float4 c[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float4 sum = red[j * 64 + eslot];
#pragma unroll
                for (int w = 1; w < 8; ++w) {
                    const float4 o = red[(w * 3 + j) * 64 + eslot];
                    sum.x += o.x; sum.y += o.y; sum.z += o.z; sum.w += o.w;
                }
                c[j] = make_float4(sum.x + bb[j].x, sum.y + bb[j].y, sum.z + bb[j].z, sum.w + bb[j].w);
            }
            if (eok) {
                const bool active = t < len;
                const float cr[4] = {c[0].x, c[0].y, c[0].z, c[0].w}, cz[4] = {c[1].x, c[1].y, c[1].z, c[1].w};
                const float cn[4] = {c[2].x, c[2].y, c[2].z, c[2].w};
                const float xr[4] = {xo[0].x, xo[0].y, xo[0].z, xo[0].w}, xz[4] = {xo[1].x, xo[1].y, xo[1].z, xo[1].w};
                const float xn[4] = {xo[2].x, xo[2].y, xo[2].z, xo[2].w};
                const float hpv[4] = {hp.x, hp.y, hp.z, hp.w};
                float rr[4], zz[4], nn[4], ho[4], o2[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    rr[i] = vag_sigmoid(cr[i] + xr[i]);
                    zz[i] = vag_sigmoid(cz[i] + xz[i]);
                    nn[i] = vag_tanh(xn[i] + rr[i] * cn[i]);
                    const float hn = (1.f - zz[i]) * nn[i] + zz[i] * hpv[i];
                    ho[i] = active ? hn : hpv[i];
                    o2[i] = active ? hn : 0.f;
                }
                hp = make_float4(ho[0], ho[1], ho[2], ho[3]);
                const int64_t o = (int64_t)em * H + eu;
                st_sc1(rs, (unsigned)((((int64_t)(k + 1) * B) * H + o) * 4), hp);                  // h_{k+1}: read by other workgroups
                // publish before the saves below (nobody in this launch reads those): this wave is the only one that stored
                // h; drain, then ONE lane signals for the workgroup
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(cnt + k, 1u, RLX_AGENT);      // (lane 0 is row m0: always a valid row)
                float* sv = a.gates + ((int64_t)(d * Ts + k) * 4) * BH + o;
                *reinterpret_cast<float4*>(sv) = make_float4(rr[0], rr[1], rr[2], rr[3]);
                *reinterpret_cast<float4*>(sv + BH) = make_float4(zz[0], zz[1], zz[2], zz[3]);
                *reinterpret_cast<float4*>(sv + 2 * BH) = make_float4(nn[0], nn[1], nn[2], nn[3]);
                *reinterpret_cast<float4*>(sv + 3 * BH) = c[2];
                const int64_t eo = ((int64_t)em * Ts + t) * 2 * H + d * H + eu;
                if (a.rng && a.p_ctx > 0.f) {               // the counter-based mask of this element (the backward kernel recomputes it)
#pragma unroll
                    for (int i = 0; i < 4; ++i) o2[i] *= vag_drop_mul(a.rng, VAG_DROP_ENC_CTX, (uint64_t)eo + i, a.p_ctx);
                }
                *reinterpret_cast<float4*>(a.enc + eo) = make_float4(o2[0], o2[1], o2[2], o2[3]);
            }
        }
        // (the barrier behind the next step's wait separates wave 0's LDS reads of this step from the next step's writes)
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Encoder forward recurrence, 2-byte storage mode, WIDE batches (configs[4]: B = 256, H = 1024): both directions in ONE
// launch.  The fp32 kernel above keeps a 48 x H slice of W_hh as three bf16 planes per workgroup and scales by adding row
// tiles to the GRID (2 x B/16 x H/16 workgroups), which stops at B = 64 (H = 512) on 256 CUs.  Here the stored-fp16
// recurrent weights (derived buffer, vag_derive_weights(with_fp16 = 1)) make a 96 x H slice fit the registers as ONE fp16
// plane (96 VGPRs at H = 1024), and a workgroup takes FOUR row tiles (64 batch rows) through that slice per step:
//   grid = 2 directions x ceil(B/64) row groups x H/32 unit slices (256 workgroups at B = 256, H = 1024);
//   wave (kq, ch) of 8: K quarter kq (H/4 columns of h), unit half ch (16 units x 3 gates) -> per step and row tile
//   3 x H/128 v_mfma_f32_16x16x32_f16; partial sums of the four K quarters meet in LDS (96 KB, which also keeps the CU to
//   one workgroup), and wave (r, ch) runs the cell for row tile r: every wave has a product share AND an epilogue share.
// The hidden state is published for the other workgroups' products as fp16 (8 bytes per lane, write-through), which is
// what an fp16-operand product would round it to anyway; the carried state (registers), the saved states and gates and
// the encoder output stay fp32.  Per step a workgroup ingests 64 x H fp16 = 128 KB instead of the launch chain's 48 x H
// weights + rows per 16-row tile; hand-off as above (sc1 stores, drain, one counter add per wave, bounded poll, sc1 loads).
struct EncWArgs {
    const float* xp;            // (Ts, B, 6H): input projections [fwd r z n | rev r z n], biases included
    const vag_half* W16[2];     // (3H, H) recurrent weights per direction, fp16
    const float* bias[2];       // (3H) b_hh
    const int* lengths;         // (B)
    float* hst;                 // [2][Ts+1][B][H]; slot 0 (zeros) is written here
    float* gates;               // [2][Ts][4][B][H]
    float* enc;                 // (B, Ts, 2H)
    vag_half* hx;               // [2][Ts+1][B][H] fp16 copy of the states for the exchange; slot 0 is written here (zeros)
    unsigned* cnt;              // [2][RG][Ts], zero on entry
    unsigned* err;
    unsigned* guard;            // {void flag, give-up count} of the driver this launch belongs to
    unsigned spin;
    int B, Ts, H, RG, CS;
    const uint64_t* rng;        // context dropout (Encoder.py:63-64) applied to enc as it is written, as the fp32 kernel does (round 6:
    float p_ctx;                // a separate pass over the 2 x B x Ts x H outputs before); NULL / 0: none
};
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
// N sc1 loads of 16 bytes at a 64-byte stride (k-steps of 32 halves) + their wait in one statement
template <int N> __device__ __forceinline__ void ld16_sc1(const vag_half* p, u32x4 (&v)[N]);
template <> __device__ __forceinline__ void ld16_sc1<4>(const vag_half* p, u32x4 (&v)[4]) {
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:64 sc1\n\t"
                 "global_load_dwordx4 %2, %4, off offset:128 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:192 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(p) : "memory");
}
template <> __device__ __forceinline__ void ld16_sc1<8>(const vag_half* p, u32x4 (&v)[8]) {
    asm volatile("global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %8, off offset:64 sc1\n\t"
                 "global_load_dwordx4 %2, %8, off offset:128 sc1\n\tglobal_load_dwordx4 %3, %8, off offset:192 sc1\n\t"
                 "global_load_dwordx4 %4, %8, off offset:256 sc1\n\tglobal_load_dwordx4 %5, %8, off offset:320 sc1\n\t"
                 "global_load_dwordx4 %6, %8, off offset:384 sc1\n\tglobal_load_dwordx4 %7, %8, off offset:448 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(p) : "memory");
}
__device__ __forceinline__ void st_sc1_h4(vag_half* p, float a, float b, float c, float d) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 v = {pack_f16(a, b), pack_f16(c, d)};
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}

// KST: k-steps of 32 per wave = H / 128
template <int KST>
__global__ __launch_bounds__(512, 1) void enc_fwd_wide16_kernel(EncWArgs a) {
    extern __shared__ __attribute__((aligned(16))) float wide_lds[];      // [4 row tiles][4 K quarters][2 unit halves][3 gates][64] float4
    const int wg = blockIdx.x;
    const int cs = wg % a.CS, rg = (wg / a.CS) % a.RG, d = wg / (a.CS * a.RG);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kq = wave & 3, ch = wave >> 2;             // product role: K quarter, unit half
    const int er = wave & 3, ech = wave >> 2;            // epilogue role: row tile, unit half
    const int H = a.H, B = a.B, Ts = a.Ts;
    const int fr = lane & 15, fg = lane >> 4;
    const int u0 = cs * 32, m0 = rg * 64;
    const int kbase = kq * (H >> 2);
    const int64_t BH = (int64_t)B * H;

    // this wave's share of the weight slice: wf[s][gate] = W16[gate*H + u0 + 16 ch + fr][kbase + 32 s + 8 fg .. +7]
    h16x8 wf[KST][3];
    {
        const vag_half* W = a.W16[d];
#pragma unroll
        for (int s = 0; s < KST; ++s)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const uint4 q = *reinterpret_cast<const uint4*>(W + (int64_t)(j * H + u0 + 16 * ch + fr) * H + kbase + 32 * s + 8 * fg);
                const u32x4 t = {q.x, q.y, q.z, q.w};
                wf[s][j] = __builtin_bit_cast(h16x8, t);
            }
    }
    // epilogue identity: batch row em, units eu .. eu + 3
    const int em = m0 + 16 * er + fr, eu = u0 + 16 * ech + 4 * fg;
    const bool eok = em < B;
    float4 bb[3] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
    int len = 0;
    if (eok) {
#pragma unroll
        for (int j = 0; j < 3; ++j) bb[j] = *reinterpret_cast<const float4*>(a.bias[d] + j * H + eu);
        len = a.lengths[em];
    }
    float4 hp = make_float4(0.f, 0.f, 0.f, 0.f);

    float* hs = a.hst + (int64_t)d * (Ts + 1) * BH;
    if (eok) *reinterpret_cast<float4*>(hs + (int64_t)em * H + eu) = hp;        // slot 0 (zeros): the backward pass reads it as h_prev
    vag_half* hx = a.hx + (int64_t)d * (Ts + 1) * BH;
    gu32* cnt = (gu32*)(a.cnt + ((int64_t)d * a.RG + rg) * Ts);
    const unsigned target = (unsigned)a.CS * 8u;           // every wave of every workgroup of the row group signs a step
    float4* red = reinterpret_cast<float4*>(wide_lds);
    const int wsrc4 = ld_src4(lane);
    bool dead = false;

    for (int k = 0; k < Ts; ++k) {
        const int t = d == 0 ? k : Ts - 1 - k;
        float4 xo[3];
        if (eok) {
            const float* xp = a.xp + ((int64_t)t * B + em) * 6 * H + d * 3 * H + eu;
#pragma unroll
            for (int j = 0; j < 3; ++j) xo[j] = *reinterpret_cast<const float4*>(xp + j * H);
        }
        if (k > 0) {
            if (threadIdx.x == 0 && !dead) {
                unsigned spins = 0;
                while (__hip_atomic_load(cnt + (k - 1), RLX_AGENT) < target) {
                    if (++spins > a.spin) { note_timeout(a.err, a.guard); dead = true; break; }
                }
            }
            __syncthreads();
            // ---- products: the four row tiles of the group against this wave's weight share
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = min(m0 + 16 * r + ld_row(lane), B - 1);          // quad-contiguous load mapping (see ld_row)
                u32x4 hq[KST];
                ld16_sc1<KST>(hx + ((int64_t)k * B + row) * H + kbase + 8 * (lane & 3), hq);
#pragma unroll
                for (int s = 0; s < KST; ++s) {
                    hq[s][0] = (unsigned)__builtin_amdgcn_ds_bpermute(wsrc4, (int)hq[s][0]);
                    hq[s][1] = (unsigned)__builtin_amdgcn_ds_bpermute(wsrc4, (int)hq[s][1]);
                    hq[s][2] = (unsigned)__builtin_amdgcn_ds_bpermute(wsrc4, (int)hq[s][2]);
                    hq[s][3] = (unsigned)__builtin_amdgcn_ds_bpermute(wsrc4, (int)hq[s][3]);
                }
                f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int s = 0; s < KST; ++s) {
                    const h16x8 hf = __builtin_bit_cast(h16x8, hq[s]);
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[s][j], hf, acc[j], 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    red[((((r * 4 + kq) * 2 + ch) * 3) + j) * 64 + lane] = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
            }
            __syncthreads();
        }
        // ---- cell of row tile er, unit half ech (step 0: h = 0, the product is zero)
        float4 c[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k > 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 o = red[((((er * 4 + q) * 2 + ech) * 3) + j) * 64 + lane];
                    sum.x += o.x; sum.y += o.y; sum.z += o.z; sum.w += o.w;
                }
            }
            c[j] = make_float4(sum.x + bb[j].x, sum.y + bb[j].y, sum.z + bb[j].z, sum.w + bb[j].w);
        }
        if (eok) {
            const bool active = t < len;
            const float cr[4] = {c[0].x, c[0].y, c[0].z, c[0].w}, cz[4] = {c[1].x, c[1].y, c[1].z, c[1].w};
            const float cn[4] = {c[2].x, c[2].y, c[2].z, c[2].w};
            const float xr[4] = {xo[0].x, xo[0].y, xo[0].z, xo[0].w}, xz[4] = {xo[1].x, xo[1].y, xo[1].z, xo[1].w};
            const float xn[4] = {xo[2].x, xo[2].y, xo[2].z, xo[2].w};
            const float hpv[4] = {hp.x, hp.y, hp.z, hp.w};
            float rr[4], zz[4], nn[4], ho[4], o2[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                rr[i] = vag_sigmoid(cr[i] + xr[i]);
                zz[i] = vag_sigmoid(cz[i] + xz[i]);
                nn[i] = vag_tanh(xn[i] + rr[i] * cn[i]);
                const float hn = (1.f - zz[i]) * nn[i] + zz[i] * hpv[i];
                ho[i] = active ? hn : hpv[i];
                o2[i] = active ? hn : 0.f;
            }
            hp = make_float4(ho[0], ho[1], ho[2], ho[3]);
            const int64_t o = (int64_t)em * H + eu;
            st_sc1_h4(hx + (int64_t)(k + 1) * BH + o, ho[0], ho[1], ho[2], ho[3]);       // read by the other workgroups
            // publish before the saves below (nobody in this launch reads those): drain, one lane signs for the wave
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(cnt + k, 1u, RLX_AGENT);      // (lane 0 = first row of the tile: valid whenever any is)
            *reinterpret_cast<float4*>(hs + (int64_t)(k + 1) * BH + o) = hp;                 // fp32 state: backward's h_prev
            float* sv = a.gates + ((int64_t)(d * Ts + k) * 4) * BH + o;
            *reinterpret_cast<float4*>(sv) = make_float4(rr[0], rr[1], rr[2], rr[3]);
            *reinterpret_cast<float4*>(sv + BH) = make_float4(zz[0], zz[1], zz[2], zz[3]);
            *reinterpret_cast<float4*>(sv + 2 * BH) = make_float4(nn[0], nn[1], nn[2], nn[3]);
            *reinterpret_cast<float4*>(sv + 3 * BH) = c[2];
            const int64_t eo = ((int64_t)em * Ts + t) * 2 * H + d * H + eu;
            if (a.rng && a.p_ctx > 0.f) {               // the counter-based mask of this element (the backward kernel recomputes it)
#pragma unroll
                for (int i = 0; i < 4; ++i) o2[i] *= vag_drop_mul(a.rng, VAG_DROP_ENC_CTX, (uint64_t)eo + i, a.p_ctx);
            }
            *reinterpret_cast<float4*>(a.enc + eo) = make_float4(o2[0], o2[1], o2[2], o2[3]);
        } else if (lane == 0) {
            __hip_atomic_fetch_add(cnt + k, 1u, RLX_AGENT);                     // a tile wholly past the batch edge still signs
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Encoder backward recurrence (what autograd replays for nn.GRU, layers/Encoder.py:58), both directions, in ONE launch.
// Same decomposition and hand-off as the forward kernel: a workgroup owns 16 hidden units x 16 batch rows; its 16 rows of
// W_hh^T (K = 3H) stay in registers as bf16x3 planes (72 VGPRs at H = 512); per step it reads the row tile's 16 x 3H gate
// gradients of the later step (sc1), forms dh = dgh[k+1] W_hh + z*dh[k+1] for its units (36 MFMAs per wave), runs the cell
// backward of step k in the epilogue and publishes its 16 x 48 gate gradients.  The carried z*dh of its own units never
// leaves the registers.
struct EncBArgs {
    const float* WT[2];         // (H, 3H) W_hh^T per direction
    const float* d_enc;         // (B, Ts, 2H) gradient of the encoder states (before the context dropout)
    const float* gates;         // [2][Ts][4][B][H]
    const float* hst;           // [2][Ts+1][B][H]
    const int* lengths;
    const uint64_t* rng; float p_ctx;
    float* d_xp;                // (Ts, B, 6H): dgi of both directions
    float* dgh;                 // [2][Ts][B][3H]
    unsigned* cnt;              // [2][RT][Ts]
    unsigned* err;
    unsigned* guard;            // {void flag, give-up count} of the driver this launch belongs to
    unsigned spin;
    int B, Ts, H, RT, CS;
    int rt0, RTP;               // as EncPArgs
};
template <int KS>               // k-steps of 32 per wave: 3H / 8 / 32 (H = 512: 6)
__global__ __launch_bounds__(512, 1) void enc_bwd_persistent_kernel(EncBArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[21504];      // 84 KB: [0, 2048) reduction; the rest keeps the CU to ourselves
    const int wg = blockIdx.x;
    const int cs = wg % a.CS, rt = a.rt0 + (wg / a.CS) % a.RTP, d = wg / (a.CS * a.RTP);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int H = a.H, B = a.B, Ts = a.Ts, K = 3 * H;
    const int m0 = rt * 16, u0 = cs * 16;
    const int fr = lane & 15, fg = lane >> 4;
    const int kbase = wave * (K >> 3);
    const int64_t BH = (int64_t)B * H;
    bf16x8 wf[KS][3];
    {
        const float* WT = a.WT[d] + (int64_t)(u0 + fr) * K + kbase + 8 * fg;
#pragma unroll
        for (int s = 0; s < KS; ++s) split8(*reinterpret_cast<const float4*>(WT + 32 * s), *reinterpret_cast<const float4*>(WT + 32 * s + 4), wf[s]);
    }
    // epilogue lane = (batch row lane >> 2, unit quad lane & 3): a quad touches ONE row (see the forward kernel)
    const int erow = lane >> 2, equad = lane & 3, eslot = erow + 16 * equad;
    const int em = m0 + erow, eu = u0 + 4 * equad;
    const bool eok = wave == 0 && em < B;
    const int len = eok ? a.lengths[em] : 0;
    float4 carry = make_float4(0.f, 0.f, 0.f, 0.f);           // z * dh of the later step, own units
    float* dghd = a.dgh + (int64_t)d * Ts * B * K;
    gu32* cnt = (gu32*)(a.cnt + ((int64_t)d * a.RT + rt) * Ts);
    const int lrow = min(m0 + ld_row(lane), B - 1);         // the row this lane LOADS (quad-contiguous mapping)
    const int src4 = ld_src4(lane);
    float4* red = reinterpret_cast<float4*>(lds);
    bool dead = false;
    const int64_t ld_add = (int64_t)Ts * 2 * H;

    for (int k = Ts - 1; k >= 0; --k) {
        const int t = d == 0 ? k : Ts - 1 - k;
        // epilogue operands of step k (independent of the recurrence): requested before the wait
        float4 sv[4], hp, e4;
        if (eok) {
            const int64_t o = (int64_t)em * H + eu;
            const float* g = a.gates + ((int64_t)(d * Ts + k) * 4) * BH + o;
#pragma unroll
            for (int q = 0; q < 4; ++q) sv[q] = *reinterpret_cast<const float4*>(g + q * BH);
            hp = *reinterpret_cast<const float4*>(a.hst + ((int64_t)d * (Ts + 1) + k) * BH + o);
            e4 = *reinterpret_cast<const float4*>(a.d_enc + (int64_t)em * ld_add + (int64_t)t * 2 * H + d * H + eu);
        }
        float4 dh = carry;
        if (k < Ts - 1) {
            if (threadIdx.x == 0 && !dead) {
                unsigned spins = 0;
                while (__hip_atomic_load(cnt + (k + 1), RLX_AGENT) < (unsigned)a.CS) {
                    if (++spins > a.spin) { note_timeout(a.err, a.guard); dead = true; break; }
                }
            }
            __syncthreads();
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            float4 ga[KS], gb[KS];
            ld_rows_sc1<KS>(dghd + ((int64_t)(k + 1) * B + lrow) * K + kbase + 8 * (lane & 3), ga, gb);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                bf16x8 hf[3];
                split8(perm4(ga[s], src4), perm4(gb[s], src4), hf);
                acc = mma6(wf[s], hf, acc);
            }
            red[wave * 64 + lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    const float4 o = red[w * 64 + eslot];
                    dh.x += o.x; dh.y += o.y; dh.z += o.z; dh.w += o.w;
                }
            }
        }
        if (eok) {
            const bool active = t < len;
            float dhv[4] = {dh.x, dh.y, dh.z, dh.w};
            float gi[3][4], gh[3][4], cy[4];
            if (!active) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { gi[0][q] = gi[1][q] = gi[2][q] = gh[0][q] = gh[1][q] = gh[2][q] = 0.f; cy[q] = dhv[q]; }
            } else {
                const float ev[4] = {e4.x, e4.y, e4.z, e4.w};
                const float r_[4] = {sv[0].x, sv[0].y, sv[0].z, sv[0].w}, z_[4] = {sv[1].x, sv[1].y, sv[1].z, sv[1].w};
                const float n_[4] = {sv[2].x, sv[2].y, sv[2].z, sv[2].w}, hn[4] = {sv[3].x, sv[3].y, sv[3].z, sv[3].w};
                const float hpv[4] = {hp.x, hp.y, hp.z, hp.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float e = ev[q];
                    if (a.rng && a.p_ctx > 0.f)
                        e *= vag_drop_mul(a.rng, VAG_DROP_ENC_CTX, (uint64_t)em * ld_add + (uint64_t)t * 2 * H + d * H + eu + q, a.p_ctx);
                    const float x = dhv[q] + e;
                    const float dn_pre = x * (1.f - z_[q]) * (1.f - n_[q] * n_[q]);
                    const float dz_pre = x * (hpv[q] - n_[q]) * z_[q] * (1.f - z_[q]);
                    const float dr_pre = dn_pre * hn[q] * r_[q] * (1.f - r_[q]);
                    gi[0][q] = dr_pre; gi[1][q] = dz_pre; gi[2][q] = dn_pre;
                    gh[0][q] = dr_pre; gh[1][q] = dz_pre; gh[2][q] = dn_pre * r_[q];
                    cy[q] = x * z_[q];
                }
            }
            carry = make_float4(cy[0], cy[1], cy[2], cy[3]);
            float* gho = dghd + ((int64_t)k * B + em) * K + eu;
#pragma unroll
            for (int g = 0; g < 3; ++g) st_sc1_f4(gho + g * H, make_float4(gh[g][0], gh[g][1], gh[g][2], gh[g][3]));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(cnt + k, 1u, RLX_AGENT);      // (lane 0 is row m0: always a valid row)
            float* gio = a.d_xp + ((int64_t)t * B + em) * 6 * H + d * 3 * H + eu;
#pragma unroll
            for (int g = 0; g < 3; ++g) *reinterpret_cast<float4*>(gio + g * H) = make_float4(gi[g][0], gi[g][1], gi[g][2], gi[g][3]);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Encoder backward recurrence, 2-byte storage mode, wide batches (configs[4]): the twin of enc_fwd_wide16_kernel.  Same grid
// (2 directions x ceil(B/64) row groups x H/32 unit slices), same wave roles: wave (kq, ch) holds rows [16 ch, 16 ch + 16) of
// the workgroup's 32 rows of the stored-fp16 W_hh^T over K quarter kq of K = 3H (96 VGPRs at H = 1024) and takes the four
// row tiles of the group through it; wave (r, ch) runs the cell backward of row tile r.  The gate gradients are published for
// the other workgroups' products as fp16 scaled by 2^12 -- exactly what the launch chain's fp16-pipe product rounds them to
// (skinny_mma_h16, a_scale) -- beside the fp32 copy the weight-gradient products read afterwards.
struct EncWBArgs {
    const vag_half* WT16[2];    // (H, 3H) W_hh^T per direction, fp16
    const float* d_enc;         // (B, Ts, 2H) gradient of the encoder states (before the context dropout)
    const float* gates;         // [2][Ts][4][B][H]
    const float* hst;           // [2][Ts+1][B][H]
    const int* lengths;
    const uint64_t* rng; float p_ctx;
    float* d_xp;                // (Ts, B, 6H)
    float* dgh;                 // [2][Ts][B][3H] fp32
    vag_half* gx;               // [2][Ts][B][3H] fp16 x 2^12: the exchanged copy
    unsigned* cnt;              // [2][RG][Ts], zero on entry
    unsigned* err;
    unsigned* guard;            // {void flag, give-up count} of the driver this launch belongs to
    unsigned spin;
    int B, Ts, H, RG, CS;
};
template <> __device__ __forceinline__ void ld16_sc1<6>(const vag_half* p, u32x4 (&v)[6]) {
    asm volatile("global_load_dwordx4 %0, %6, off sc1\n\tglobal_load_dwordx4 %1, %6, off offset:64 sc1\n\t"
                 "global_load_dwordx4 %2, %6, off offset:128 sc1\n\tglobal_load_dwordx4 %3, %6, off offset:192 sc1\n\t"
                 "global_load_dwordx4 %4, %6, off offset:256 sc1\n\tglobal_load_dwordx4 %5, %6, off offset:320 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]) : "v"(p) : "memory");
}
template <> __device__ __forceinline__ void ld16_sc1<12>(const vag_half* p, u32x4 (&v)[12]) {
    asm volatile("global_load_dwordx4 %0, %12, off sc1\n\tglobal_load_dwordx4 %1, %12, off offset:64 sc1\n\t"
                 "global_load_dwordx4 %2, %12, off offset:128 sc1\n\tglobal_load_dwordx4 %3, %12, off offset:192 sc1\n\t"
                 "global_load_dwordx4 %4, %12, off offset:256 sc1\n\tglobal_load_dwordx4 %5, %12, off offset:320 sc1\n\t"
                 "global_load_dwordx4 %6, %12, off offset:384 sc1\n\tglobal_load_dwordx4 %7, %12, off offset:448 sc1\n\t"
                 "global_load_dwordx4 %8, %12, off offset:512 sc1\n\tglobal_load_dwordx4 %9, %12, off offset:576 sc1\n\t"
                 "global_load_dwordx4 %10, %12, off offset:640 sc1\n\tglobal_load_dwordx4 %11, %12, off offset:704 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]),
                   "=&v"(v[8]), "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11])
                 : "v"(p) : "memory");
}

// HK: half of a wave's k-steps of 32 (K quarter = 3H / 4 = 64 HK): 12 at H = 1024, 6 at H = 512
template <int HK>
__global__ __launch_bounds__(512, 1) void enc_bwd_wide16_kernel(EncWBArgs a) {
    extern __shared__ __attribute__((aligned(16))) float wide_lds[];      // [4 row tiles][4 K quarters][2 unit halves][64] float4
    constexpr float GSCALE = 4096.f;
    const int wg = blockIdx.x;
    const int cs = wg % a.CS, rg = (wg / a.CS) % a.RG, d = wg / (a.CS * a.RG);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kq = wave & 3, ch = wave >> 2;             // product role
    const int er = wave & 3, ech = wave >> 2;            // epilogue role
    const int H = a.H, B = a.B, Ts = a.Ts, K = 3 * H;
    const int fr = lane & 15, fg = lane >> 4;
    const int u0 = cs * 32, m0 = rg * 64;
    const int kbase = kq * (K >> 2);
    const int64_t BH = (int64_t)B * H;
    const int wsrc4 = ld_src4(lane);

    h16x8 wf[2 * HK];
    {
        const vag_half* WT = a.WT16[d] + (int64_t)(u0 + 16 * ch + fr) * K + kbase + 8 * fg;
#pragma unroll
        for (int s = 0; s < 2 * HK; ++s) {
            const uint4 q = *reinterpret_cast<const uint4*>(WT + 32 * s);
            const u32x4 t = {q.x, q.y, q.z, q.w};
            wf[s] = __builtin_bit_cast(h16x8, t);
        }
    }
    const int em = m0 + 16 * er + fr, eu = u0 + 16 * ech + 4 * fg;
    const bool eok = em < B;
    const int len = eok ? a.lengths[em] : 0;
    float4 carry = make_float4(0.f, 0.f, 0.f, 0.f);           // z * dh of the later step, own units
    float* dghd = a.dgh + (int64_t)d * Ts * B * K;
    vag_half* gxd = a.gx + (int64_t)d * Ts * B * K;
    gu32* cnt = (gu32*)(a.cnt + ((int64_t)d * a.RG + rg) * Ts);
    const unsigned target = (unsigned)a.CS * 8u;
    float4* red = reinterpret_cast<float4*>(wide_lds);
    bool dead = false;
    const int64_t ld_add = (int64_t)Ts * 2 * H;

    for (int k = Ts - 1; k >= 0; --k) {
        const int t = d == 0 ? k : Ts - 1 - k;
        float4 sv[4], hp, e4;
        if (eok) {
            const int64_t o = (int64_t)em * H + eu;
            const float* g = a.gates + ((int64_t)(d * Ts + k) * 4) * BH + o;
#pragma unroll
            for (int q = 0; q < 4; ++q) sv[q] = *reinterpret_cast<const float4*>(g + q * BH);
            hp = *reinterpret_cast<const float4*>(a.hst + ((int64_t)d * (Ts + 1) + k) * BH + o);
            e4 = *reinterpret_cast<const float4*>(a.d_enc + (int64_t)em * ld_add + (int64_t)t * 2 * H + d * H + eu);
        }
        float4 dh = carry;
        if (k < Ts - 1) {
            if (threadIdx.x == 0 && !dead) {
                unsigned spins = 0;
                while (__hip_atomic_load(cnt + (k + 1), RLX_AGENT) < target) {
                    if (++spins > a.spin) { note_timeout(a.err, a.guard); dead = true; break; }
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = min(m0 + 16 * r + ld_row(lane), B - 1);          // quad-contiguous load mapping (see ld_row)
                const vag_half* gp = gxd + ((int64_t)(k + 1) * B + row) * K + kbase + 8 * (lane & 3);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int hf_ = 0; hf_ < 2; ++hf_) {
                    u32x4 gq[HK];
                    ld16_sc1<HK>(gp + hf_ * HK * 32, gq);
#pragma unroll
                    for (int s = 0; s < HK; ++s) {
                        u32x4 q = gq[s];
                        q[0] = (unsigned)__builtin_amdgcn_ds_bpermute(wsrc4, (int)q[0]);
                        q[1] = (unsigned)__builtin_amdgcn_ds_bpermute(wsrc4, (int)q[1]);
                        q[2] = (unsigned)__builtin_amdgcn_ds_bpermute(wsrc4, (int)q[2]);
                        q[3] = (unsigned)__builtin_amdgcn_ds_bpermute(wsrc4, (int)q[3]);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[hf_ * HK + s], __builtin_bit_cast(h16x8, q), acc, 0, 0, 0);
                    }
                }
                red[(((r * 4 + kq) * 2 + ch)) * 64 + lane] =
                    make_float4(acc[0] * (1.f / GSCALE), acc[1] * (1.f / GSCALE), acc[2] * (1.f / GSCALE), acc[3] * (1.f / GSCALE));
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 o = red[(((er * 4 + q) * 2 + ech)) * 64 + lane];
                dh.x += o.x; dh.y += o.y; dh.z += o.z; dh.w += o.w;
            }
        }
        if (eok) {
            const bool active = t < len;
            float dhv[4] = {dh.x, dh.y, dh.z, dh.w};
            float gi[3][4], gh[3][4], cy[4];
            if (!active) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { gi[0][q] = gi[1][q] = gi[2][q] = gh[0][q] = gh[1][q] = gh[2][q] = 0.f; cy[q] = dhv[q]; }
            } else {
                const float ev[4] = {e4.x, e4.y, e4.z, e4.w};
                const float r_[4] = {sv[0].x, sv[0].y, sv[0].z, sv[0].w}, z_[4] = {sv[1].x, sv[1].y, sv[1].z, sv[1].w};
                const float n_[4] = {sv[2].x, sv[2].y, sv[2].z, sv[2].w}, hn[4] = {sv[3].x, sv[3].y, sv[3].z, sv[3].w};
                const float hpv[4] = {hp.x, hp.y, hp.z, hp.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float e = ev[q];
                    if (a.rng && a.p_ctx > 0.f)
                        e *= vag_drop_mul(a.rng, VAG_DROP_ENC_CTX, (uint64_t)em * ld_add + (uint64_t)t * 2 * H + d * H + eu + q, a.p_ctx);
                    const float x = dhv[q] + e;
                    const float dn_pre = x * (1.f - z_[q]) * (1.f - n_[q] * n_[q]);
                    const float dz_pre = x * (hpv[q] - n_[q]) * z_[q] * (1.f - z_[q]);
                    const float dr_pre = dn_pre * hn[q] * r_[q] * (1.f - r_[q]);
                    gi[0][q] = dr_pre; gi[1][q] = dz_pre; gi[2][q] = dn_pre;
                    gh[0][q] = dr_pre; gh[1][q] = dz_pre; gh[2][q] = dn_pre * r_[q];
                    cy[q] = x * z_[q];
                }
            }
            carry = make_float4(cy[0], cy[1], cy[2], cy[3]);
            const int64_t go = ((int64_t)k * B + em) * K + eu;
#pragma unroll
            for (int g = 0; g < 3; ++g)
                st_sc1_h4(gxd + go + g * H, gh[g][0] * GSCALE, gh[g][1] * GSCALE, gh[g][2] * GSCALE, gh[g][3] * GSCALE);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(cnt + k, 1u, RLX_AGENT);
#pragma unroll
            for (int g = 0; g < 3; ++g) *reinterpret_cast<float4*>(dghd + go + g * H) = make_float4(gh[g][0], gh[g][1], gh[g][2], gh[g][3]);
            float* gio = a.d_xp + ((int64_t)t * B + em) * 6 * H + d * 3 * H + eu;
#pragma unroll
            for (int g = 0; g < 3; ++g) *reinterpret_cast<float4*>(gio + g * H) = make_float4(gi[g][0], gi[g][1], gi[g][2], gi[g][3]);
        } else if (lane == 0) {
            __hip_atomic_fetch_add(cnt + k, 1u, RLX_AGENT);                     // a tile wholly past the batch edge still signs
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Decoder forward recurrence, teacher forced (layers/NMT_Decoder.py:109-129 x the loop of models/...V11.py:138-146), in ONE
// launch.  As a chain of launches a step is 4 kernels and ~24 us; every one of them re-reads, per workgroup, weights that
// never change and keys that never change.  Here everything that is constant over the steps lives on chip:
//   * grid = ceil(B/16) row tiles x 64 workgroups (256 at B = 64), one per CU.  Workgroup i of a row tile owns hidden units
//     [8i, 8i+8) of both cells and attention-query columns [16i, 16i+16) for the tile's 16 batch rows;
//   * registers: its rows of W_hh1 (24), attn_h (16) and W_hh2 (24) as bf16x3 planes in MFMA operand layout;
//   * LDS: its 16 query columns of the attention keys pe and its 24 gate columns of the projected keys encwp, for all
//     16 x Ts (row, position) pairs of the tile -- the 21 MB per step that the launch chain streams from
//     L2 / the Infinity Cache at every step are read from memory ONCE per sequence;
//   * a step = 3 phases with an exchange between them (sc1 stores / fp32 atomics + counter + sc1 loads, as in the encoder
//     kernel above):   h2[t-1] -> (gru_1 cell) -> h1 -> (q = attn_h h1, hp2 = W_hh2 h1 + b, own columns' share of every score)
//                      -> scores -> (softmax, projected context of own columns, gru_2 cell) -> h2[t].
// Everything the backward pass needs (h1, both cells' gates, [q | hp2], alpha, h2) is saved in the launch chain's own
// layout, so the backward operators are unchanged.
struct DecPArgs {
    const float *pe, *mask, *h0, *xp1, *W1, *b1, *wcat, *bcat, *v, *encwp, *b_ih2;
    float *h1, *g1, *qhp, *alpha, *h2_all, *g2, *psc;
    unsigned* cnt;              // [5 phases][RT][Tt], zero on entry (2 and 4: free-running form only)
    unsigned* err;
    unsigned* guard;            // {void flag, give-up count} of the driver this launch belongs to
    unsigned spin;
    unsigned long long* dbg;    // NULL, or [Tt][8] 100 MHz timestamps of workgroup 0's phase boundaries (tools/exp_dec_phases.py)
    int B, Ts, Tt, H, RT;       // RT: row tiles of the whole batch (the counters are indexed by the global row tile)
    int rt0;                    // first row tile of THIS launch (batches wider than the chip: passes of row tiles)
    // ---- free-running form (FREE = true): the kernel feeds its own arg-max back (V11.py:148-160 / :207-226), xp1 is not read
    const float *embp;          // (V, 3H)  emb W_ih1^T + b_ih1: the input projection of gru_1 for every possible token
    const float *embw3;         // (V, E)   emb W3^T: the head's share of the embedded input (NMT_Decoder.py:137)
    const float *encw2;         // (B, Ts, E) enc W2^T: the head's share of a context is alpha . encw2
    const float *hw1, *hb1, *hb2, *hb3, *out_w, *out_b;     // head: W1 (E, H), the three biases, out (V, E), (V)
    float *tmid;                // (Tt, B, E) tanh(.) * dropout, exchanged between the workgroups and kept for the backward pass
    float *logits;              // (Tt * B, ldl) or NULL (decoding: nobody reads them)
    unsigned long long* cand;   // (Tt, B, 64) per-workgroup arg-max candidates
    int64_t* tok;               // (Tt + 1, B): row 0 given; row t + 1 = arg-max of step t
    const uint64_t* rng;        // output dropout (:140-141), NULL / p_out = 0: none
    float p_out;
    int V, ldl;
    int xcd_map;                // 1: slices dealt so that the blocks of one XCD hold consecutive ones (see the kernel)
};

// A counter of the decoder kernel is 4 shards, 64 bytes apart (64 arrivals on ONE word serialise at the memory side, ~12 ns
// each: MI355X_MICROARCH.md, fanin): workgroup i arrives on shard i & 3; lanes 0-3 of wave 0 poll one shard each with relaxed
// agent-scope loads until every shard holds its 16 arrivals, then the workgroup's barrier.
constexpr int SHARDS = 4, SHARD_STRIDE = 16, CNT_WORDS = SHARDS * SHARD_STRIDE;
__device__ __forceinline__ void arrive(gu32* c, int i) { __hip_atomic_fetch_add(c + (i & (SHARDS - 1)) * SHARD_STRIDE, 1u, RLX_AGENT); }
__device__ __forceinline__ void wait_count(gu32* c, unsigned want_per_shard, unsigned* err, unsigned* guard, unsigned spin, bool& dead) {
    if (threadIdx.x < 64 && !dead) {
        const int lane = threadIdx.x;
        unsigned spins = 0;
        for (;;) {
            const bool ok = lane >= SHARDS || __hip_atomic_load(c + lane * SHARD_STRIDE, RLX_AGENT) >= want_per_shard;
            if (__all(ok)) break;
            if (++spins > spin) { if (lane == 0) note_timeout(err, guard); dead = true; break; }
        }
    }
    __syncthreads();
}

// two accumulator words (see ACC_SHARDS) as the sums of their four copies, `stride` floats apart: eight 4-byte sc1 loads in flight
__device__ __forceinline__ void ld_acc_shards(const float* p0, const float* p1, int64_t stride, float& v0, float& v1) {
    float a0, a1, a2, a3, b0, b1, b2, b3;
    const float *p02 = p0 + 2 * stride, *p12 = p1 + 2 * stride;
    const float *p01 = p0 + stride, *p03 = p02 + stride, *p11 = p1 + stride, *p13 = p12 + stride;
    asm volatile("global_load_dword %0, %8, off sc1\n\tglobal_load_dword %1, %9, off sc1\n\t"
                 "global_load_dword %2, %10, off sc1\n\tglobal_load_dword %3, %11, off sc1\n\t"
                 "global_load_dword %4, %12, off sc1\n\tglobal_load_dword %5, %13, off sc1\n\t"
                 "global_load_dword %6, %14, off sc1\n\tglobal_load_dword %7, %15, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)
                 : "v"(p0), "v"(p01), "v"(p02), "v"(p03), "v"(p1), "v"(p11), "v"(p12), "v"(p13) : "memory");
    v0 = (a0 + a1) + (a2 + a3);
    v1 = (b0 + b1) + (b2 + b3);
}

// ld_acc_shards and the six k-steps of a wave's share of a handed-off row (ld_rows_sc1<6>) in ONE statement: the decoder backward
// requests the dgh2 rows of its hidden-side product together with the d-alpha words of phase B, one round of memory latency for both
// (the rows then wait in registers through phase B instead of costing the hidden-side product a round trip of its own).
// 12 + 8 outputs + 9 addresses: 29 operands, the asm statement's limit is 30.
__device__ __forceinline__ void ld_acc_shards_and_rows6(const float* p0, const float* p1, int64_t stride, float& v0, float& v1,
                                                        const float* rows, float4 (&a)[6], float4 (&b)[6]) {
    float a0, a1, a2, a3, b0, b1, b2, b3;
    const float *p02 = p0 + 2 * stride, *p12 = p1 + 2 * stride;
    const float *p01 = p0 + stride, *p03 = p02 + stride, *p11 = p1 + stride, *p13 = p12 + stride;
    asm volatile("global_load_dword %0, %20, off sc1\n\tglobal_load_dword %1, %21, off sc1\n\t"
                 "global_load_dword %2, %22, off sc1\n\tglobal_load_dword %3, %23, off sc1\n\t"
                 "global_load_dword %4, %24, off sc1\n\tglobal_load_dword %5, %25, off sc1\n\t"
                 "global_load_dword %6, %26, off sc1\n\tglobal_load_dword %7, %27, off sc1\n\t"
                 "global_load_dwordx4 %8, %28, off sc1\n\tglobal_load_dwordx4 %9, %28, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %10, %28, off offset:128 sc1\n\tglobal_load_dwordx4 %11, %28, off offset:144 sc1\n\t"
                 "global_load_dwordx4 %12, %28, off offset:256 sc1\n\tglobal_load_dwordx4 %13, %28, off offset:272 sc1\n\t"
                 "global_load_dwordx4 %14, %28, off offset:384 sc1\n\tglobal_load_dwordx4 %15, %28, off offset:400 sc1\n\t"
                 "global_load_dwordx4 %16, %28, off offset:512 sc1\n\tglobal_load_dwordx4 %17, %28, off offset:528 sc1\n\t"
                 "global_load_dwordx4 %18, %28, off offset:640 sc1\n\tglobal_load_dwordx4 %19, %28, off offset:656 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3),
                   "=&v"(a[0]), "=&v"(b[0]), "=&v"(a[1]), "=&v"(b[1]), "=&v"(a[2]), "=&v"(b[2]), "=&v"(a[3]), "=&v"(b[3]),
                   "=&v"(a[4]), "=&v"(b[4]), "=&v"(a[5]), "=&v"(b[5])
                 : "v"(p0), "v"(p01), "v"(p02), "v"(p03), "v"(p1), "v"(p11), "v"(p12), "v"(p13), "v"(rows) : "memory");
    v0 = (a0 + a1) + (a2 + a3);
    v1 = (b0 + b1) + (b2 + b3);
}

// ... and three k-steps (H = 256)
__device__ __forceinline__ void ld_acc_shards_and_rows3(const float* p0, const float* p1, int64_t stride, float& v0, float& v1,
                                                        const float* rows, float4 (&a)[3], float4 (&b)[3]) {
    float a0, a1, a2, a3, b0, b1, b2, b3;
    const float *p02 = p0 + 2 * stride, *p12 = p1 + 2 * stride;
    const float *p01 = p0 + stride, *p03 = p02 + stride, *p11 = p1 + stride, *p13 = p12 + stride;
    asm volatile("global_load_dword %0, %14, off sc1\n\tglobal_load_dword %1, %15, off sc1\n\t"
                 "global_load_dword %2, %16, off sc1\n\tglobal_load_dword %3, %17, off sc1\n\t"
                 "global_load_dword %4, %18, off sc1\n\tglobal_load_dword %5, %19, off sc1\n\t"
                 "global_load_dword %6, %20, off sc1\n\tglobal_load_dword %7, %21, off sc1\n\t"
                 "global_load_dwordx4 %8, %22, off sc1\n\tglobal_load_dwordx4 %9, %22, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %10, %22, off offset:128 sc1\n\tglobal_load_dwordx4 %11, %22, off offset:144 sc1\n\t"
                 "global_load_dwordx4 %12, %22, off offset:256 sc1\n\tglobal_load_dwordx4 %13, %22, off offset:272 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3),
                   "=&v"(a[0]), "=&v"(b[0]), "=&v"(a[1]), "=&v"(b[1]), "=&v"(a[2]), "=&v"(b[2])
                 : "v"(p0), "v"(p01), "v"(p02), "v"(p03), "v"(p1), "v"(p11), "v"(p12), "v"(p13), "v"(rows) : "memory");
    v0 = (a0 + a1) + (a2 + a3);
    v1 = (b0 + b1) + (b2 + b3);
}
__device__ __forceinline__ void ld_acc_shards_and_rows(const float* p0, const float* p1, int64_t stride, float& v0, float& v1,
                                                       const float* rows, float4 (&a)[6], float4 (&b)[6]) {
    ld_acc_shards_and_rows6(p0, p1, stride, v0, v1, rows, a, b);
}
__device__ __forceinline__ void ld_acc_shards_and_rows(const float* p0, const float* p1, int64_t stride, float& v0, float& v1,
                                                       const float* rows, float4 (&a)[3], float4 (&b)[3]) {
    ld_acc_shards_and_rows3(p0, p1, stride, v0, v1, rows, a, b);
}

constexpr int DEC_WGS = 64;          // workgroups per row tile (H = 512)
// Workgroup -> (slice i, row tile rt).  Blocks b and b + 8 share an XCD (round-robin placement: observed, speed only), and consecutive
// slices share memory lines of the key images the kernels copy into LDS (two slices per 128-byte line of pe, four per line of the
// projected keys' gate blocks) and of the pieces they exchange.  In block order every line crossed the fabric once per slice.
// mode 1: the eight blocks of a row tile that share an XCD take eight CONSECUTIVE slices -- the second to fourth reader of a line
// finds it in that XCD's L2 (forward kernel, keys into LDS: 11.4 -> 6.6 us per launch, the launch 484.7 -> 469.2 us; optimiser step
// -10..13 us on three boxes).  The backward kernel is 0..5 us per launch SLOWER with it (its 32- and 64-byte pieces are WRITTEN by
// neighbouring slices under a drained hand-off) and 5-9 us faster with PAIRS of consecutive slices per XCD (mode 2: 832.7 / 833.7 /
// 834.8 -> 829.7 / 828.0 / 829.3 us, one box, alternating runs); four per XCD (mode 3) and pairs for the forward kernel are within
// noise of those.  One row tile per XCD pair (its exchanges into two L2s instead of eight) measured the same as mode 1: not kept.
// The encoder's kernels: no gain (forward) / +3 us (backward) with mode 1: they keep block order.
// WGS: workgroups per row tile (64 at H = 512, 32 at H = 256: a row tile then has four blocks per XCD, so at most four consecutive
// slices can share one).  rt0: the first row tile of this launch (a batch wider than the chip is taken in passes of row tiles).
template <int WGS>
__device__ __forceinline__ void dec_slice_map(int mode, int rt0, int& i, int& rt) {
    const int bx = blockIdx.x % WGS;
    rt = rt0 + blockIdx.x / WGS;
    int g = mode == 1 ? 8 : (mode == 2 ? 2 : (mode == 3 ? 4 : 1));      // consecutive slices per XCD
    if (g > WGS / 8) g = WGS / 8;
    i = (bx & 7) * g + ((bx >> 3) % g) + 8 * g * (bx / (8 * g));
}
// The 64 workgroups of a row tile add their shares of a step's scores (forward) / d alpha (backward) with fp32 atomics.  Atomics
// execute at the memory side and adds to ONE address serialise there (~12 ns each: MI355X_MICROARCH.md, global float atomics /
// fanin): 64 adders per word kept every wave's atomics outstanding for ~1.1 us.  So the accumulators exist in ACC_SHARDS copies,
// workgroup i adds into copy i % ACC_SHARDS (16 adders per word) and the readers sum the copies (four loads in flight instead of one).
constexpr int ACC_SHARDS = 4;
constexpr int DEC_U = 8;             // hidden units per workgroup (H = 512)

constexpr int DEC_E = 4 * DEC_WGS;   // free-running form: embedding width, 4 columns of the head's hidden layer per workgroup
__device__ __forceinline__ unsigned long long cand_key(float v, int idx) {       // larger value first, then smaller index
    const unsigned u = __builtin_bit_cast(unsigned, v);
    return ((unsigned long long)(u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u)) << 32) | (unsigned long long)(0xffffffffu - (unsigned)idx);
}

// WGS workgroups per row tile, each owning DEC_U = 8 hidden units: H = 8 WGS (512 or 256).  The attention width is C = 2H, so a
// workgroup's share of the query is C / WGS = 16 columns at either size; only the K share of a wave (H / 8: two k-steps or one) and
// the fan-in of the exchanges change.  The free-running form exists at H = 512 only (its head slices assume E = 4 WGS = 256).
template <bool FREE, int WGS = DEC_WGS>
__global__ __launch_bounds__(512, 1) void dec_fwd_persistent_kernel(DecPArgs a) {
    extern __shared__ __attribute__((aligned(16))) float dlds[];
    static_assert(WGS == 64 || (WGS == 32 && !FREE), "H = 512 (both forms) or H = 256 (teacher-forced form)");
    constexpr int H = WGS * DEC_U, C = 2 * H, Q = C + 3 * H, KS = H / 256;     // K share of a wave: H / 8 = 32 KS
    constexpr int E = DEC_E;
    int i, rt;
    dec_slice_map<WGS>(a.xcd_map, a.rt0, i, rt);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int B = a.B, Ts = a.Ts, Tt = a.Tt;
    const int m0 = rt * 16, u0 = i * DEC_U;
    const int fr = lane & 15, fg = lane >> 4;
    const int kbase = wave * (H >> 3);
    const int64_t BH = (int64_t)B * H;
    // LDS carve-up (floats)
    float4* red = reinterpret_cast<float4*>(dlds);                  // [8 waves][3 tiles][64 lanes]
    float* pe_s = dlds + 6144;                                       // [16 Ts pairs][16 own query columns] of the keys
    float* ew_s = pe_s + 16 * Ts * 16;                               // [16 rows][Ts][24]
    float* sc_s = ew_s + 16 * Ts * 24;                               // [16][Ts] scores -> alpha
    float* gi_s = sc_s + 16 * Ts;                                    // [16][24] projected context of own columns
    float* hp_s = gi_s + 16 * 24;                                    // [16][24] hidden-side projection of gru_2 (own units)
    float* bs_s = hp_s + 16 * 24;                                    // [3 kinds][3 gates][8]: b_hh1, b_hh2, b_ih2 of the own units
    float* hs_s = bs_s + 80;                                         // [2][16][8]: own units of h1 (this step) and h2 (previous step)
    float* v_s = hs_s + 256;                                         // [C] attention vector
    float* mk_s = v_s + C;                                           // [16][Ts] source mask of the tile's pairs
    float* q_s = mk_s + 16 * Ts;                                     // [16][16] this step's query, own columns
    float* cw_s = q_s + 256;                                         // FREE: [16][4] alpha . encw2 of the own head columns
    float* hb_s = cw_s + 64;                                         // FREE: [4] b1 + b2 + b3 of the own head columns (+ 12 pad)
    int* tok_s = reinterpret_cast<int*>(hb_s + 16);                  // FREE: [16] the tile's input tokens of the current step
    float* ew2_s = reinterpret_cast<float*>(tok_s + 16);             // FREE: [16][Ts][4] own head columns of encw2
    unsigned long long* cb_s = reinterpret_cast<unsigned long long*>(q_s);            // FREE: [128] arg-max partials (q_s is idle then)
    bf16x8* tm_s = reinterpret_cast<bf16x8*>(red);                                   // FREE: [3 planes][8 k-steps][64 lanes] the head's hidden layer, while `red` is idle

#ifdef VAG_LAB          // prologue stamps (teacher-forced form): entry | weights in registers | keys in LDS, in row Tt of the stamp array
#define VAG_PSTAMP(k) do { if (!FREE && a.dbg && blockIdx.x == 0 && threadIdx.x == 0) a.dbg[Tt * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define VAG_PSTAMP(k) do { } while (0)
#endif
    VAG_PSTAMP(0);
#ifdef VAG_LAB          // ... and per row tile: entry and exit of its first workgroup (rows Tt + 1, Tt + 2), entry of its LAST one (row Tt + 3)
    if (!FREE && a.dbg && threadIdx.x == 0 && rt < 8) {
        if (i == 0) a.dbg[(Tt + 1) * 8 + rt] = __builtin_amdgcn_s_memrealtime();
        if (i == WGS - 1) a.dbg[(Tt + 3) * 8 + rt] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    // ---- weights as bf16 planes in registers (A operands: rows = output columns).  Row of a tile held by lane fr:
    //   cell tiles: tile 0 = [r | z] of the 8 own units (fr < 8: gate r, unit fr; else gate z, unit fr - 8), tile 1 = [n | n again]
    bf16x8 w1[KS][2][3], w2[KS][3][3];
    {
        const int ur = fr & 7, g01 = fr >> 3;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int ko = kbase + 32 * s + 8 * fg;
            const float* p0 = a.W1 + (int64_t)(g01 * H + u0 + ur) * H + ko;
            const float* p1 = a.W1 + (int64_t)(2 * H + u0 + ur) * H + ko;
            // FREE: rows 8..11 of tile 1 (a second copy of n otherwise) are the own four rows of the head's W1: W1 h2[t-1]
            // comes out of the product that gru_1 of step t needs anyway
            if (FREE && fr >= 8 && fr < 12) p1 = a.hw1 + (int64_t)(4 * i + fr - 8) * H + ko;
            split8(*reinterpret_cast<const float4*>(p0), *reinterpret_cast<const float4*>(p0 + 4), w1[s][0]);
            split8(*reinterpret_cast<const float4*>(p1), *reinterpret_cast<const float4*>(p1 + 4), w1[s][1]);
            const float* q0 = a.wcat + (int64_t)(16 * i + fr) * H + ko;                         // attn_h rows = query columns
            const float* q1 = a.wcat + (int64_t)(C + g01 * H + u0 + ur) * H + ko;               // W_hh2 [r | z]
            const float* q2 = a.wcat + (int64_t)(C + 2 * H + u0 + ur) * H + ko;                 // W_hh2 [n | n]
            split8(*reinterpret_cast<const float4*>(q0), *reinterpret_cast<const float4*>(q0 + 4), w2[s][0]);
            split8(*reinterpret_cast<const float4*>(q1), *reinterpret_cast<const float4*>(q1 + 4), w2[s][1]);
            split8(*reinterpret_cast<const float4*>(q2), *reinterpret_cast<const float4*>(q2 + 4), w2[s][2]);
        }
    }
    VAG_PSTAMP(1);
    // ---- own 16 query columns of the keys of all 16 x Ts pairs, own 24 gate columns of the projected keys -> LDS (read from
    // memory once per sequence: 11.4-11.6 us of the launch, tools/exp_dec_phases.py; batching the loads of a thread -- eight in flight
    // before the first LDS store -- changed nothing, so it is the 64-byte pieces out of 26 MB, not load latency, that set it)
    for (int x = threadIdx.x; x < 16 * Ts * 4; x += 512) {           // (pair, column quad) -> one float4
        const int P = x >> 2, c4 = x & 3;
        const int r = P / Ts, sp = P - r * Ts, b = min(m0 + r, B - 1);
        reinterpret_cast<float4*>(pe_s)[x] = *reinterpret_cast<const float4*>(a.pe + ((int64_t)b * Ts + sp) * C + 16 * i + 4 * c4);
    }
    for (int x = threadIdx.x; x < 16 * Ts * 6; x += 512) {           // (row, position, gate, half) -> one float4
        const int hf = x & 1, g = (x >> 1) % 3, rs_ = x / 6;
        const int r = rs_ / Ts, sp = rs_ - r * Ts, b = min(m0 + r, B - 1);
        reinterpret_cast<float4*>(ew_s)[rs_ * 6 + g * 2 + hf] =
            *reinterpret_cast<const float4*>(a.encwp + ((int64_t)b * Ts + sp) * 3 * H + g * H + u0 + 4 * hf);
    }
    // ---- epilogue threads: e = threadIdx.x < 32: batch row fr, unit quad hq: units u0 + 4 hq .. + 3
    const int hq = (threadIdx.x >> 4) & 1;
    const int em = m0 + fr, eu = u0 + 4 * hq;
    const bool ep = threadIdx.x < 32, eok = ep && em < B;
    if (threadIdx.x < 72) {
        const int kind = threadIdx.x / 24, g = (threadIdx.x / 8) % 3, u = threadIdx.x & 7;
        const float* src = kind == 0 ? a.b1 : (kind == 1 ? a.bcat + C : a.b_ih2);
        bs_s[threadIdx.x] = src[g * H + u0 + u];
    }
    for (int x = threadIdx.x; x < C / 4; x += 512) reinterpret_cast<float4*>(v_s)[x] = reinterpret_cast<const float4*>(a.v)[x];
    for (int x = threadIdx.x; x < 16 * Ts; x += 512) {
        const int r = x / Ts, sp = x - r * Ts;
        mk_s[x] = a.mask[(int64_t)min(m0 + r, B - 1) * Ts + sp];
    }
    if (ep) *reinterpret_cast<float4*>(hs_s + 128 + fr * 8 + 4 * hq) = eok ? *reinterpret_cast<const float4*>(a.h0 + (int64_t)em * H + eu)
                                                                             : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 e3 = make_float4(0.f, 0.f, 0.f, 0.f);          // FREE, threads 0..15: embw3 row of the current input token, own columns
    if (FREE) {
        for (int x = threadIdx.x; x < 16 * Ts; x += 512) {           // (row, position) -> one float4
            const int r = x / Ts, sp = x - r * Ts, b = min(m0 + r, B - 1);
            reinterpret_cast<float4*>(ew2_s)[x] = *reinterpret_cast<const float4*>(a.encw2 + ((int64_t)b * Ts + sp) * E + 4 * i);
        }
        if (threadIdx.x < 4) hb_s[threadIdx.x] = a.hb1[4 * i + threadIdx.x] + a.hb2[4 * i + threadIdx.x] + a.hb3[4 * i + threadIdx.x];
        if (threadIdx.x < 16) {
            const int tk = min(max((int)a.tok[min(m0 + (int)threadIdx.x, B - 1)], 0), a.V - 1);     // (a table row, whatever the caller wrote)
            tok_s[threadIdx.x] = tk;
            e3 = *reinterpret_cast<const float4*>(a.embw3 + (int64_t)tk * E + 4 * i);
        }
    }
    gu32* c1 = (gu32*)(a.cnt + ((int64_t)0 * a.RT + rt) * Tt * CNT_WORDS);       // h2[t] published (waited on by step t + 1)
    gu32* c2 = (gu32*)(a.cnt + ((int64_t)1 * a.RT + rt) * Tt * CNT_WORDS);       // h1
    gu32* c4 = (gu32*)(a.cnt + ((int64_t)3 * a.RT + rt) * Tt * CNT_WORDS);       // scores
    gu32* c5 = (gu32*)(a.cnt + ((int64_t)2 * a.RT + rt) * Tt * CNT_WORDS);       // FREE: the head's hidden layer of step t
    gu32* c6 = (gu32*)(a.cnt + ((int64_t)4 * a.RT + rt) * Tt * CNT_WORDS);       // FREE: arg-max candidates of step t
    constexpr unsigned PER_SHARD = WGS / SHARDS;
    bool dead = false;
    __syncthreads();
    VAG_PSTAMP(2);
#ifdef VAG_LAB          // phase timestamps: lab builds only (make LAB=1 -> libvagnmt_lab.so); the product kernels carry no hook
    const bool stamp = a.dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
#define VAG_STAMP(k) do { if (stamp) a.dbg[t * (FREE ? 16 : 8) + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define VAG_STAMP(k) do { } while (0)
#endif

    for (int t = 0; t < Tt + (FREE ? 1 : 0); ++t) {       // FREE: one more pass for the head and arg-max of the last step
        // FREE: the thread's coordinates again, opaque to the compiler: every address of a pass is then recomputed (a few integer
        // instructions) instead of being hoisted out of the time loop -- hoisted, they and the weight planes they displace end
        // up in scratch memory (measured: 110 spilled registers, 3-5 us per step).  Teacher-forced form: the same values as outside.
        int tx = threadIdx.x;
        if (FREE) asm volatile("" : "+v"(tx));
        const int lane = tx & 63, wave = FREE ? __builtin_amdgcn_readfirstlane(tx >> 6) : tx >> 6, fr = lane & 15, fg = lane >> 4;
        const int kbase = wave * (H >> 3), hq = (tx >> 4) & 1, em = m0 + fr, eu = u0 + 4 * hq;
        const bool ep = tx < 32, eok = ep && em < B;
        const int arow = min(m0 + fr, B - 1);
        const int lrow = min(m0 + ld_row(lane), B - 1);    // the batch row this lane LOADS (quad-contiguous mapping, see ld_row)
        const int src4 = ld_src4(lane);
        VAG_STAMP(0);
        // ================= phase 1: gru_1 cell (NMT_Decoder.py:121) =================
        float4 ha[KS], hb[KS];
        if (t == 0) {
            const float* hp = a.h0 + (int64_t)arow * H + kbase + 8 * fg;          // written before this launch: plain loads
#pragma unroll
            for (int s = 0; s < KS; ++s) { ha[s] = *reinterpret_cast<const float4*>(hp + 32 * s); hb[s] = *reinterpret_cast<const float4*>(hp + 32 * s + 4); }
        } else {
            wait_count(c1 + (t - 1) * CNT_WORDS, PER_SHARD, a.err, a.guard, a.spin, dead);
            VAG_STAMP(1);
            ld_rows_tagged<KS>(a.h2_all + ((int64_t)(t - 1) * B + lrow) * H + kbase + 8 * (lane & 3), ha, hb, a.spin, dead, a.err, a.guard);
#pragma unroll
            for (int s = 0; s < KS; ++s) { ha[s] = perm4(ha[s], src4); hb[s] = perm4(hb[s], src4); }
        }
        {
            f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                bf16x8 hf[3];
                split8(ha[s], hb[s], hf);
                acc[0] = mma6(w1[s][0], hf, acc[0]);
                acc[1] = mma6(w1[s][1], hf, acc[1]);
            }
            red[(wave * 3 + 0) * 64 + lane] = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
            red[(wave * 3 + 1) * 64 + lane] = make_float4(acc[1][0], acc[1][1], acc[1][2], acc[1][3]);
        }
        float4 xo[3];                                   // the input projection of this step: arrives under the barrier + reduction
        if (!FREE && eok) {
            const float* xp = a.xp1 + ((int64_t)t * B + em) * 3 * H + eu;
#pragma unroll
            for (int g = 0; g < 3; ++g) xo[g] = *reinterpret_cast<const float4*>(xp + g * H);
        }
        __syncthreads();
        // D[tile row][batch row]: lane (fr, fg) of a tile holds rows 4 fg + i.  r of units 4 hq + i: tile 0, fg = hq;
        // z: tile 0, fg = 2 + hq; n: tile 1, fg = hq; FREE: the head's W1 h2[t-1], own four columns: tile 1, fg = 2
        float4 cr = make_float4(0, 0, 0, 0), cz = cr, cn = cr, hw = cr;
        if (ep) {
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const float4 x0 = red[(w * 3 + 0) * 64 + hq * 16 + fr], x1 = red[(w * 3 + 0) * 64 + (2 + hq) * 16 + fr];
                const float4 x2 = red[(w * 3 + 1) * 64 + hq * 16 + fr];
                cr.x += x0.x; cr.y += x0.y; cr.z += x0.z; cr.w += x0.w;
                cz.x += x1.x; cz.y += x1.y; cz.z += x1.z; cz.w += x1.w;
                cn.x += x2.x; cn.y += x2.y; cn.z += x2.z; cn.w += x2.w;
                if (FREE) {
                    const float4 x3 = red[(w * 3 + 1) * 64 + 2 * 16 + fr];
                    hw.x += x3.x; hw.y += x3.y; hw.z += x3.z; hw.w += x3.w;
                }
            }
        }
        if (FREE) {
            VAG_STAMP(8);
            if (t >= 1) {
                // ---- head of step t - 1 (NMT_Decoder.py:137-141): own four columns of tanh(W1 h2 + W2 c + W3 e + b), dropout
                const int ts = t - 1;
                const int V = a.V, NVT = (V + 15) >> 4, NJ = i < NVT ? (NVT - i + 63) >> 6 : 0;      // own tiles: i, i + 64, ...
                // Tiles to waves, in passes of four k-steps: tile i + 64 (wave + 8 n) whole for n < NJ / 8 (two passes each); of the
                // NJ % 8 tiles left over, each goes to a PAIR of waves by halves of K when there are at most three (one pass each; the
                // odd wave's partial tile crosses through LDS), else whole to one wave.  (At V = 9391: 9-10 tiles, 3 passes per wave
                // at most instead of 4.)
                const int nfull = NJ >> 3, rext = NJ & 7;
                const bool splitx = rext >= 1 && rext <= 3;
                const int xj = splitx ? (wave < 2 * rext ? 8 * nfull + (wave >> 1) : -1) : (wave < rext ? 8 * nfull + wave : -1);
                const int npass = 2 * nfull + (xj < 0 ? 0 : (splitx ? 1 : 2));
                auto pass_ptr = [&](int g, int u) -> const float* {               // weight rows of pass g, k-step u of it (this lane's 32 bytes)
                    const int j = g < 2 * nfull ? wave + 8 * (g >> 1) : xj;
                    const int k0 = g < 2 * nfull ? 4 * (g & 1) : (splitx ? 4 * (wave & 1) : 4 * (g - 2 * nfull));
                    return a.out_w + (int64_t)min(16 * (i + 64 * j) + fr, V - 1) * E + 32 * (k0 + u) + 8 * fg;
                };
                float4 wr[4][2];                                                   // out.weight rows, four k-steps in flight
                if (npass > 0) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const float* wp = pass_ptr(0, u); wr[u][0] = *reinterpret_cast<const float4*>(wp); wr[u][1] = *reinterpret_cast<const float4*>(wp + 4); }
                }
                if (tx < 16) {
                    const float4 cw = *reinterpret_cast<const float4*>(cw_s + 4 * fr);
                    const float4 bb = *reinterpret_cast<const float4*>(hb_s);
                    const uint64_t di = ((uint64_t)ts * B + (uint64_t)min(m0 + fr, B - 1)) * E + 4 * i;
                    float4 tv;
                    tv.x = vag_tanh(hw.x + cw.x + e3.x + bb.x) * vag_drop_mul(a.rng, VAG_DROP_DEC_OUT, di + 0, a.p_out);
                    tv.y = vag_tanh(hw.y + cw.y + e3.y + bb.y) * vag_drop_mul(a.rng, VAG_DROP_DEC_OUT, di + 1, a.p_out);
                    tv.z = vag_tanh(hw.z + cw.z + e3.z + bb.z) * vag_drop_mul(a.rng, VAG_DROP_DEC_OUT, di + 2, a.p_out);
                    tv.w = vag_tanh(hw.w + cw.w + e3.w + bb.w) * vag_drop_mul(a.rng, VAG_DROP_DEC_OUT, di + 3, a.p_out);
                    if (m0 + fr < B) st_sc1_f4(a.tmid + ((int64_t)ts * B + m0 + fr) * E + 4 * i, tv);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if (tx == 0) arrive(c5 + ts * CNT_WORDS, i);
                wait_count(c5 + ts * CNT_WORDS, PER_SHARD, a.err, a.guard, a.spin, dead);        // (its barrier also frees `red`)
                VAG_STAMP(9);
                // ---- logits of step t - 1 (:143) for the own vocabulary tiles (16 words each, tile i + 64 j) and their arg-max.
                // The step's hidden layer (16 rows x E) is split once into bf16 planes and parked in `red` in MFMA operand
                // layout (each wave contributes the k-step it loaded); a wave then owns whole tiles (j = wave, wave + 8, ...)
                // over all of K = E: no reduction between waves.  The weight rows stream from L2 (the slices of the 8
                // workgroups of an XCD stay there: 1.2 MB), four k-steps ahead.
                {
                    float4 ta[1], tb[1];
                    bf16x8 tf[3];
                    ld_rows_sc1<1>(a.tmid + ((int64_t)ts * B + lrow) * E + 32 * wave + 8 * (lane & 3), ta, tb);
                    split8(perm4(ta[0], src4), perm4(tb[0], src4), tf);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) tm_s[(pl * 8 + wave) * 64 + lane] = tf[pl];
                }
                __syncthreads();
                unsigned long long best = 0ull;                                   // batch row fr
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                float* part_s = gi_s;                                              // [3][64 lanes][4]: partial tiles of the odd waves (gi_s, hp_s are idle)
                auto finish_tile = [&](int j, const f32x4& t4) {                  // words v0 .. v0 + 3 of batch row fr
                    const int v0 = 16 * (i + 64 * j) + 4 * fg;
                    float lv[4] = {t4[0], t4[1], t4[2], t4[3]};
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (v0 + q < V) {
                            lv[q] += a.out_b[v0 + q];
                            const unsigned long long key = cand_key(lv[q], v0 + q);
                            best = key > best ? key : best;
                        }
                    if (a.logits && m0 + fr < B) {
                        float* lp = a.logits + ((int64_t)ts * B + m0 + fr) * a.ldl + v0;
                        if (v0 + 3 < V) *reinterpret_cast<float4*>(lp) = make_float4(lv[0], lv[1], lv[2], lv[3]);
                        else
                            for (int q = 0; q < 4; ++q) if (v0 + q < V) lp[q] = lv[q];
                    }
                };
                for (int g = 0; g < npass; ++g) {                                 // four k-steps per pass
                    const int k0 = g < 2 * nfull ? 4 * (g & 1) : (splitx ? 4 * (wave & 1) : 4 * (g - 2 * nfull));
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float4 ca = wr[u][0], cb = wr[u][1];
                        if (g + 1 < npass) {
                            const float* wp = pass_ptr(g + 1, u);
                            wr[u][0] = *reinterpret_cast<const float4*>(wp); wr[u][1] = *reinterpret_cast<const float4*>(wp + 4);
                        }
                        bf16x8 aw[3], bw[3];
                        split8(ca, cb, aw);
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) bw[pl] = tm_s[(pl * 8 + k0 + u) * 64 + lane];
                        acc = mma6(aw, bw, acc);
                    }
                    if (g < 2 * nfull ? (g & 1) : (!splitx && g == npass - 1)) {   // a whole tile is complete
                        finish_tile(g < 2 * nfull ? wave + 8 * (g >> 1) : xj, acc);
                        acc = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
                if (splitx) {                                                      // (uniform over the workgroup)
                    if (xj >= 0 && (wave & 1)) *reinterpret_cast<f32x4*>(part_s + ((wave >> 1) * 64 + lane) * 4) = acc;
                    __syncthreads();
                    if (xj >= 0 && !(wave & 1)) finish_tile(xj, acc + *reinterpret_cast<const f32x4*>(part_s + ((wave >> 1) * 64 + lane) * 4));
                }
                VAG_STAMP(10);
                {   // the four row groups of the wave hold the same batch rows: combine them, lanes 0..15 publish the wave's best
                    unsigned lo = (unsigned)best, hi = (unsigned)(best >> 32);
#pragma unroll
                    for (int o = 16; o <= 32; o <<= 1) {
                        const unsigned olo = __shfl_xor(lo, o, 64), ohi = __shfl_xor(hi, o, 64);
                        const unsigned long long ob = ((unsigned long long)ohi << 32) | olo;
                        if (ob > best) { best = ob; lo = olo; hi = ohi; }
                    }
                    if (lane < 16) cb_s[wave * 16 + lane] = best;
                }
                __syncthreads();
                if (tx < 16) {                                            // batch row fr: the 8 waves' bests
                    unsigned long long b = 0ull;
#pragma unroll
                    for (int x = 0; x < 8; ++x) { const unsigned long long c = cb_s[16 * x + fr]; b = c > b ? c : b; }
                    if (m0 + fr < B)
                        __hip_atomic_store(a.cand + ((int64_t)ts * B + m0 + fr) * DEC_WGS + i, b, RLX_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if (tx == 0) arrive(c6 + ts * CNT_WORDS, i);
                wait_count(c6 + ts * CNT_WORDS, PER_SHARD, a.err, a.guard, a.spin, dead);
                VAG_STAMP(11);
                // every workgroup of the tile reduces the 64 candidates of its 16 rows itself: 32 lanes per row, two each
                if (tx < 128) {
                    const int r = tx >> 3, c0 = tx & 7;
                    const unsigned long long* cp = a.cand + ((int64_t)ts * B + min(m0 + r, B - 1)) * DEC_WGS + c0;
                    unsigned long long k[8];
#pragma unroll
                    for (int x = 0; x < 8; ++x) k[x] = __hip_atomic_load(cp + 8 * x, RLX_AGENT);
                    unsigned long long b = k[0];
#pragma unroll
                    for (int x = 1; x < 8; ++x) b = k[x] > b ? k[x] : b;
                    cb_s[tx] = b;
                }
                __syncthreads();
                if (tx < 16) {
                    unsigned long long b = 0ull;
#pragma unroll
                    for (int x = 0; x < 8; ++x) { const unsigned long long c = cb_s[fr * 8 + x]; b = c > b ? c : b; }
                    const int tk = min(max((int)(0xffffffffu - (unsigned)(b & 0xffffffffull)), 0), V - 1);
                    tok_s[fr] = tk;
                    if (i == 0 && m0 + fr < B) a.tok[(int64_t)t * B + m0 + fr] = tk;             // V11.py:157
                    e3 = *reinterpret_cast<const float4*>(a.embw3 + (int64_t)tk * E + 4 * i);
                }
                __syncthreads();
            }
            VAG_STAMP(12);
            if (t == Tt) break;

            if (eok) {
                const float* xp = a.embp + (int64_t)tok_s[fr] * 3 * H + eu;      // (bias included)
#pragma unroll
                for (int g = 0; g < 3; ++g) xo[g] = *reinterpret_cast<const float4*>(xp + g * H);
            }
        }
        if (ep) {
            if (eok) {
                const float4 b0 = *reinterpret_cast<const float4*>(bs_s + 0 * 8 + 4 * hq), b1_ = *reinterpret_cast<const float4*>(bs_s + 1 * 8 + 4 * hq);
                const float4 b2_ = *reinterpret_cast<const float4*>(bs_s + 2 * 8 + 4 * hq);
                const float r_[4] = {cr.x + b0.x + xo[0].x, cr.y + b0.y + xo[0].y, cr.z + b0.z + xo[0].z, cr.w + b0.w + xo[0].w};
                const float z_[4] = {cz.x + b1_.x + xo[1].x, cz.y + b1_.y + xo[1].y, cz.z + b1_.z + xo[1].z, cz.w + b1_.w + xo[1].w};
                const float gh[4] = {cn.x + b2_.x, cn.y + b2_.y, cn.z + b2_.z, cn.w + b2_.w};
                const float xn[4] = {xo[2].x, xo[2].y, xo[2].z, xo[2].w};
                const float4 hprev = *reinterpret_cast<const float4*>(hs_s + 128 + fr * 8 + 4 * hq);
                const float hpv[4] = {hprev.x, hprev.y, hprev.z, hprev.w};
                float rr[4], zz[4], nn[4], ho[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    rr[q] = vag_sigmoid(r_[q]);
                    zz[q] = vag_sigmoid(z_[q]);
                    nn[q] = vag_tanh(xn[q] + rr[q] * gh[q]);
                    ho[q] = (1.f - zz[q]) * nn[q] + zz[q] * hpv[q];
                }
                const float4 h1v = tag4(make_float4(ho[0], ho[1], ho[2], ho[3]));
                *reinterpret_cast<float4*>(hs_s + fr * 8 + 4 * hq) = h1v;
                const int64_t o = (int64_t)em * H + eu;
                st_sc1_f4(a.h1 + (int64_t)t * BH + o, h1v);
                if (!VAG_TAGGED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (VAG_TAGGED && tx == 0) arrive(c2 + t * CNT_WORDS, i);            // marked words: no drain before the signal
                float* sv = a.g1 + (int64_t)t * 4 * BH + o;
                *reinterpret_cast<float4*>(sv) = make_float4(rr[0], rr[1], rr[2], rr[3]);
                *reinterpret_cast<float4*>(sv + BH) = make_float4(zz[0], zz[1], zz[2], zz[3]);
                *reinterpret_cast<float4*>(sv + 2 * BH) = make_float4(nn[0], nn[1], nn[2], nn[3]);
                *reinterpret_cast<float4*>(sv + 3 * BH) = make_float4(gh[0], gh[1], gh[2], gh[3]);
            }
            // (threads 0..31 are half of wave 0: the drain above covered every storing lane of the wave)
            if (tx == 0 && (!VAG_TAGGED || !eok)) arrive(c2 + t * CNT_WORDS, i);
        }
        // ================= phase 2: q = attn_h h1 (:47), hp2 = W_hh2 h1 + b_hh2 (hidden side of gru_2, :129) =================
        VAG_STAMP(2);
        wait_count(c2 + t * CNT_WORDS, PER_SHARD, a.err, a.guard, a.spin, dead);
        VAG_STAMP(3);
        ld_rows_tagged<KS>(a.h1 + ((int64_t)t * B + lrow) * H + kbase + 8 * (lane & 3), ha, hb, a.spin, dead, a.err, a.guard);
#pragma unroll
        for (int s = 0; s < KS; ++s) { ha[s] = perm4(ha[s], src4); hb[s] = perm4(hb[s], src4); }
        bf16x8 hf2[KS][3];
        {
            // the query first: its score shares (fp32 atomics) are in flight while the hidden-side products follow
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                split8(ha[s], hb[s], hf2[s]);
                acc = mma6(w2[s][0], hf2[s], acc);
            }
            red[(wave * 3 + 0) * 64 + lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
        __syncthreads();
        if (wave == 0) {
            // q: lane (fr, fg) -> batch row fr, columns 16 i + 4 fg .. + 3
            float4 qv = make_float4(0, 0, 0, 0);
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const float4 x0 = red[(w * 3 + 0) * 64 + lane];
                qv.x += x0.x; qv.y += x0.y; qv.z += x0.z; qv.w += x0.w;
            }
            if (m0 + fr < B) *reinterpret_cast<float4*>(a.qhp + ((int64_t)t * B + m0 + fr) * Q + 16 * i + 4 * fg) = qv;     // saved for backward
            *reinterpret_cast<float4*>(q_s + fr * 16 + 4 * fg) = qv;
        }
        __syncthreads();
        // ---- scores (:47-51): score[b,s] = sum_c v_c tanh(pe[b,s,c] + q[b,c]) is a sum over the query columns, so the owner
        // of 16 columns adds ITS share for all 16 x Ts pairs of the tile (keys of those columns: LDS) into the step's score
        // array with fp32 atomics -- no exchange of q, one hop less per step.  (The order in which the 64 shares of a score
        // arrive is not fixed: scores are reproducible to fp32 rounding, like the split-K products of gemm.hip.)
        {
            const int cq = tx & 3;
            const float4 vq = *reinterpret_cast<const float4*>(v_s + 16 * i + 4 * cq);
            for (int P = tx >> 2; P < 16 * Ts; P += 128) {
                const int r = P / Ts;
                const float4 pv = reinterpret_cast<const float4*>(pe_s)[P * 4 + cq];
                const float4 qq = *reinterpret_cast<const float4*>(q_s + r * 16 + 4 * cq);
                float acc = vq.x * vag_tanh(pv.x + qq.x) + vq.y * vag_tanh(pv.y + qq.y) + vq.z * vag_tanh(pv.z + qq.z) +
                            vq.w * vag_tanh(pv.w + qq.w);
                acc = quad_sum(acc);
                if (cq == 0 && m0 + r < B)
                    atomicAdd(a.psc + (((int64_t)t * ACC_SHARDS + (i & (ACC_SHARDS - 1))) * B + m0 + r) * Ts + (P - r * Ts), acc);
            }
        }
        {
            // hidden side of gru_2 (needed in phase 4 only): under the atomics' flight
            f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                acc[0] = mma6(w2[s][1], hf2[s], acc[0]);
                acc[1] = mma6(w2[s][2], hf2[s], acc[1]);
            }
            red[(wave * 3 + 1) * 64 + lane] = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
            red[(wave * 3 + 2) * 64 + lane] = make_float4(acc[1][0], acc[1][1], acc[1][2], acc[1][3]);
        }
        VAG_STAMP(4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every wave's atomics have been performed ...
        __syncthreads();                                       // ... before the one lane that signals for all of them
        if (tx == 0) arrive(c4 + t * CNT_WORDS, i);
        VAG_STAMP(5);
        if (ep) {
            float4 hp2[3] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const float4 x0 = red[(w * 3 + 1) * 64 + hq * 16 + fr], x1 = red[(w * 3 + 1) * 64 + (2 + hq) * 16 + fr];
                const float4 x2 = red[(w * 3 + 2) * 64 + hq * 16 + fr];
                hp2[0].x += x0.x; hp2[0].y += x0.y; hp2[0].z += x0.z; hp2[0].w += x0.w;
                hp2[1].x += x1.x; hp2[1].y += x1.y; hp2[1].z += x1.z; hp2[1].w += x1.w;
                hp2[2].x += x2.x; hp2[2].y += x2.y; hp2[2].z += x2.z; hp2[2].w += x2.w;
            }
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const float4 bb = *reinterpret_cast<const float4*>(bs_s + 24 + g * 8 + 4 * hq);
                hp2[g].x += bb.x; hp2[g].y += bb.y; hp2[g].z += bb.z; hp2[g].w += bb.w;
                *reinterpret_cast<float4*>(hp_s + fr * 24 + g * 8 + 4 * hq) = hp2[g];      // read back by the same thread in phase 4
                if (eok) *reinterpret_cast<float4*>(a.qhp + ((int64_t)t * B + em) * Q + C + g * H + eu) = hp2[g];     // saved for backward
            }
        }
        // ================= phase 4: softmax (:44), projected context of own columns, gru_2 cell (:126-129) =================
        VAG_STAMP(6);
        wait_count(c4 + t * CNT_WORDS, PER_SHARD, a.err, a.guard, a.spin, dead);
        VAG_STAMP(7);
        for (int x0 = tx; x0 < 16 * Ts; x0 += 1024) {                                // 4-byte sc1 loads, two in flight
            const int x1 = x0 + 512;
            const int r0 = x0 / Ts, r1 = min(x1, 16 * Ts - 1) / Ts;
            const float* p0 = a.psc + ((int64_t)t * ACC_SHARDS * B + min(m0 + r0, B - 1)) * Ts + (x0 - r0 * Ts);
            const float* p1 = a.psc + ((int64_t)t * ACC_SHARDS * B + min(m0 + r1, B - 1)) * Ts + (min(x1, 16 * Ts - 1) - r1 * Ts);
            float v0, v1;
            ld_acc_shards(p0, p1, (int64_t)B * Ts, v0, v1);
            sc_s[x0] = mk_s[x0] == 0.f ? -INFINITY : v0;                                      // mask :41-43
            if (x1 < 16 * Ts) sc_s[x1] = mk_s[x1] == 0.f ? -INFINITY : v1;
        }
        __syncthreads();
        for (int r = wave; r < 16; r += 8) {
            float mx = -INFINITY;
            for (int sp = lane; sp < Ts; sp += 64) mx = fmaxf(mx, sc_s[r * Ts + sp]);
            mx = wave_max(mx);
            float sum = 0.f;
            for (int sp = lane; sp < Ts; sp += 64) sum += __expf(sc_s[r * Ts + sp] - mx);
            sum = wave_sum(sum);
            const float inv = 1.f / sum;
            for (int sp = lane; sp < Ts; sp += 64) {
                const float al = __expf(sc_s[r * Ts + sp] - mx) * inv;
                sc_s[r * Ts + sp] = al;
                // saved for the backward pass; every workgroup of the tile holds all weights: each writes 1/64 of them
                if (((r * Ts + sp) & (WGS - 1)) == i && m0 + r < B) a.alpha[((int64_t)t * B + m0 + r) * Ts + sp] = al;
            }
        }
        __syncthreads();
        if (tx < 16 * 24) {
            const int r = tx / 24, col = tx - r * 24;
            const float* al = sc_s + r * Ts;
            const float* ew = ew_s + (int64_t)r * Ts * 24 + col;
            float s0 = 0.f, s1 = 0.f;
            int sp = 0;
            for (; sp + 1 < Ts; sp += 2) { s0 += al[sp] * ew[sp * 24]; s1 += al[sp + 1] * ew[(sp + 1) * 24]; }
            if (sp < Ts) s0 += al[sp] * ew[sp * 24];
            gi_s[tx] = s0 + s1;
        } else if (FREE && tx < 16 * 24 + 64) {              // the head's share of the context (W2 c = alpha . encw2), own columns
            const int x = tx - 16 * 24, r = x >> 2, col = x & 3;
            const float* al = sc_s + r * Ts;
            const float* ew = ew2_s + (int64_t)r * Ts * 4 + col;
            float s0 = 0.f, s1 = 0.f;
            int sp = 0;
            for (; sp + 1 < Ts; sp += 2) { s0 += al[sp] * ew[sp * 4]; s1 += al[sp + 1] * ew[(sp + 1) * 4]; }
            if (sp < Ts) s0 += al[sp] * ew[sp * 4];
            cw_s[x] = s0 + s1;
        }
        __syncthreads();
        if (eok) {
            float gi[3][4];
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const float4 x = *reinterpret_cast<const float4*>(gi_s + fr * 24 + g * 8 + 4 * hq);
                gi[g][0] = x.x; gi[g][1] = x.y; gi[g][2] = x.z; gi[g][3] = x.w;
            }
            float bi[3][4], hh[3][4];
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const float4 x = *reinterpret_cast<const float4*>(bs_s + 48 + g * 8 + 4 * hq);
                const float4 y = *reinterpret_cast<const float4*>(hp_s + fr * 24 + g * 8 + 4 * hq);
                bi[g][0] = x.x; bi[g][1] = x.y; bi[g][2] = x.z; bi[g][3] = x.w;
                hh[g][0] = y.x; hh[g][1] = y.y; hh[g][2] = y.z; hh[g][3] = y.w;
            }
            const float* hr = hh[0];
            const float* hz = hh[1];
            const float* hn = hh[2];
            const float4 h1v = *reinterpret_cast<const float4*>(hs_s + fr * 8 + 4 * hq);
            const float h1a[4] = {h1v.x, h1v.y, h1v.z, h1v.w};
            float rr[4], zz[4], nn[4], ho[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                rr[q] = vag_sigmoid(gi[0][q] + bi[0][q] + hr[q]);
                zz[q] = vag_sigmoid(gi[1][q] + bi[1][q] + hz[q]);
                nn[q] = vag_tanh(gi[2][q] + bi[2][q] + rr[q] * hn[q]);
                ho[q] = (1.f - zz[q]) * nn[q] + zz[q] * h1a[q];
            }
            const float4 h2v = tag4(make_float4(ho[0], ho[1], ho[2], ho[3]));
            *reinterpret_cast<float4*>(hs_s + 128 + fr * 8 + 4 * hq) = h2v;
            const int64_t o = (int64_t)em * H + eu;
            st_sc1_f4(a.h2_all + (int64_t)t * BH + o, h2v);
            if (!VAG_TAGGED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (VAG_TAGGED && tx == 0) arrive(c1 + t * CNT_WORDS, i);
            float* sv = a.g2 + (int64_t)t * 4 * BH + o;
            *reinterpret_cast<float4*>(sv) = make_float4(rr[0], rr[1], rr[2], rr[3]);
            *reinterpret_cast<float4*>(sv + BH) = make_float4(zz[0], zz[1], zz[2], zz[3]);
            *reinterpret_cast<float4*>(sv + 2 * BH) = make_float4(nn[0], nn[1], nn[2], nn[3]);
            *reinterpret_cast<float4*>(sv + 3 * BH) = make_float4(hn[0], hn[1], hn[2], hn[3]);
        }
        if (tx == 0 && (!VAG_TAGGED || !eok)) arrive(c1 + t * CNT_WORDS, i);
    }
#ifdef VAG_LAB
    if (!FREE && a.dbg && threadIdx.x == 0 && i == 0 && rt < 8) a.dbg[(Tt + 2) * 8 + rt] = __builtin_amdgcn_s_memrealtime();
#endif
#undef VAG_STAMP
}

// ------------------------------------------------------------------------------------------------------------------
// Decoder backward recurrence (teacher forced), all Tt steps in ONE launch.  Same decomposition as the forward kernel:
// 64 workgroups per 16-row tile, workgroup i owns hidden units [8i, 8i+8) of both cells, query columns [16i, 16i+16) and the
// 24 gate columns of its units.  Per step t (descending), three hand-offs:
//   A  (own units' gru_2 cell backward of step t is at hand: dgi2, dgh2, z2*dh2)  the own gate columns' share of every
//      d alpha[b,s] = encwp[b,s,:] . dgi2[b,:] from the LDS-resident projected keys, added with fp32 atomics; dgh2 published
//   B  complete d alpha (+ the head's part, dah) -> softmax backward ds -> dq for the own 16 query columns from the
//      LDS-resident keys, published;  beside that hand-off: the hidden side dgh2 W_hh2 + z2*dh2 for the own units
//      (W_hh2^T rows streamed from L2: the third weight slice does not fit the registers)
//   C  dh1 = dq attn_h + hidden side -> gru_1 cell backward -> dgi1 saved, dgh1 published, z1*dh1 kept
//   D  dh2[t-1] = dgh1 W_hh1 + z1*dh1 + head's d_h2[t-1] -> gru_2 cell backward of step t-1 -> (A) of the next step, local
// Outputs in the launch chain's layout (dgi2, [dq | dgh2], ds, dgi1, dgh1, d_h0): the post-loop operators are unchanged.
struct DecBArgs {
    const float *pe, *encwp, *v, *wcatT, *whh1T;            // wcatT (H, C+3H) = [attn_h^T | W_hh2^T], whh1T (H, 3H)
    const float *h0, *h2_all, *h1, *g1, *g2, *qhp, *alpha;  // saved by the forward pass
    const float *d_h2_all, *dah;                            // head's gradient of h2 (Tt,B,H); d alpha through the head (Tt,B,Ts)
    float *dgi2, *dqgh, *ds, *dgi1, *dgh1, *d_h0;
    float* dal;                 // (Tt,B,Ts) accumulated with atomics: zero on entry
    unsigned* cnt;              // [3 phases][RT][Tt] x CNT_WORDS, zero on entry
    unsigned* err;
    unsigned* guard;            // {void flag, give-up count} of the driver this launch belongs to
    unsigned spin;
    unsigned long long* dbg;    // NULL, or [Tt][8] timestamps of workgroup 0 (tools/exp_dec_bwd_phases.py)
    int B, Ts, Tt, H, RT;       // RT: row tiles of the whole batch
    int rt0;                    // first row tile of this launch (see DecPArgs::rt0)
    int xcd_map;                // as DecPArgs::xcd_map
    int h0_tanh;                // 1: d_h0 leaves as the gradient of the PRE-activation of h0 = tanh(.) (V11.py:118): d_h0 * (1 - h0^2)
};

// GRU cell backward for 4 units (see gru_bwd_elem_kernel): dh = total gradient of the cell's output
__device__ __forceinline__ void cell_bwd4(const float (&dh)[4], const float4 (&sv)[4], const float4 hp, float (&gi)[3][4],
                                          float (&gh)[3][4], float (&direct)[4]) {
    const float r_[4] = {sv[0].x, sv[0].y, sv[0].z, sv[0].w}, z_[4] = {sv[1].x, sv[1].y, sv[1].z, sv[1].w};
    const float n_[4] = {sv[2].x, sv[2].y, sv[2].z, sv[2].w}, hn[4] = {sv[3].x, sv[3].y, sv[3].z, sv[3].w};
    const float hpv[4] = {hp.x, hp.y, hp.z, hp.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float dn_pre = dh[q] * (1.f - z_[q]) * (1.f - n_[q] * n_[q]);
        const float dz_pre = dh[q] * (hpv[q] - n_[q]) * z_[q] * (1.f - z_[q]);
        const float dr_pre = dn_pre * hn[q] * r_[q] * (1.f - r_[q]);
        gi[0][q] = dr_pre; gi[1][q] = dz_pre; gi[2][q] = dn_pre;
        gh[0][q] = dr_pre; gh[1][q] = dz_pre; gh[2][q] = dn_pre * r_[q];
        direct[q] = dh[q] * z_[q];
    }
}

template <int WGS>               // workgroups per row tile: H = 8 WGS (64 -> 512, 32 -> 256), as the forward kernel
__global__ __launch_bounds__(512, 1) void dec_bwd_persistent_kernel(DecBArgs a) {
    extern __shared__ __attribute__((aligned(16))) float dlds[];
    constexpr int H = WGS * DEC_U, C = 2 * H, Q = C + 3 * H;
    constexpr unsigned PER_SHARD = WGS / SHARDS;
    int i, rt;
    dec_slice_map<WGS>(a.xcd_map, a.rt0, i, rt);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int B = a.B, Ts = a.Ts, Tt = a.Tt;
#ifdef VAG_LAB          // prologue / exit stamps of block 0 in rows Tt .. of the stamp array (tools/exp_dec_bwd_phases.py)
#define VAG_BPSTAMP(k) do { if (a.dbg && blockIdx.x == 0 && threadIdx.x == 0) a.dbg[Tt * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define VAG_BPSTAMP(k) do { } while (0)
#endif
    VAG_BPSTAMP(0);
    const int m0 = rt * 16, u0 = i * DEC_U;
    const int fr = lane & 15, fg = lane >> 4;
    const int64_t BH = (int64_t)B * H;
    const int NP = 16 * Ts;                                          // (row, position) pairs of the tile
    float4* red = reinterpret_cast<float4*>(dlds);                   // [8 waves][64 lanes]
    float* pe_s = dlds + 2048;                                       // [NP][16 own query columns]
    float* ew_t = pe_s + NP * 16;                                    // [24 own gate columns][NP]  (column-major: pair-parallel reads)
    float* da_s = ew_t + 24 * NP;                                    // [NP] d alpha -> ds
    float* al_s = da_s + NP;                                         // [NP] alpha of this step
    float* gi_s = al_s + NP;                                         // [16][24] dgi2 of the own columns
    float* d1_s = gi_s + 384;                                        // [16][8] z2 * dh2: the direct path into dh1
    float* pb_s = d1_s + 128;                                        // [16][8] dgh2 W_hh2 + z2 * dh2
    float* c1_s = pb_s + 128;                                        // [16][8] z1 * dh1: the direct path into dh2[t-1]
    float* dq_s = c1_s + 128;                                        // [16][16] dq of the own columns (staging for 16-byte stores)
    float* wb_s = dq_s + 256;                                        // [8 own units][C + 4] attn_h^T rows, fp32 (padded: the 8 rows of a
                                                                     // fragment read fall on different banks); split on the fly
    constexpr int WBLD = C + 4;

    // ---- register-resident weights (A operands, 16-row tile = the 8 own units twice): attn_h^T (K = C) and W_hh1^T (K = 3H)
    constexpr int KB = C / 8 / 32, KC = 3 * H / 8 / 32;             // k-steps per wave: 4 and 6
    bf16x8 wc[KC][3];
    {
        const float* pc = a.whh1T + (int64_t)(u0 + (fr & 7)) * (3 * H) + wave * (3 * H >> 3) + 8 * fg;
#pragma unroll
        for (int s = 0; s < KC; ++s) split8(*reinterpret_cast<const float4*>(pc + 32 * s), *reinterpret_cast<const float4*>(pc + 32 * s + 4), wc[s]);
    }
    for (int x = threadIdx.x; x < 8 * (C / 4); x += 512) {           // attn_h^T rows of the own units -> LDS
        const int u = x / (C / 4), c4 = x - u * (C / 4);
        *reinterpret_cast<float4*>(wb_s + u * WBLD + 4 * c4) = *reinterpret_cast<const float4*>(a.wcatT + (int64_t)(u0 + u) * Q + 4 * c4);
    }
    const float* wb_row = wb_s + (fr & 7) * WBLD + wave * (C >> 3) + 8 * fg;
    // W_hh2^T, streamed per step (L2 hits): loaded in the quad-contiguous lane mapping too and permuted into place
    const float* wa_row = a.wcatT + (int64_t)(u0 + (ld_row(lane) & 7)) * Q + C + wave * (3 * H >> 3) + 8 * (lane & 3);
    // ---- keys -> LDS
    for (int x = threadIdx.x; x < NP * 4; x += 512) {
        const int P = x >> 2, c4 = x & 3;
        const int r = P / Ts, sp = P - r * Ts, b = min(m0 + r, B - 1);
        reinterpret_cast<float4*>(pe_s)[x] = *reinterpret_cast<const float4*>(a.pe + ((int64_t)b * Ts + sp) * C + 16 * i + 4 * c4);
    }
    for (int x = threadIdx.x; x < NP * 6; x += 512) {                // (pair, gate, half): four columns each
        const int hf = x & 1, g = (x >> 1) % 3, P = x / 6;
        const int r = P / Ts, sp = P - r * Ts, b = min(m0 + r, B - 1);
        const float4 e = *reinterpret_cast<const float4*>(a.encwp + ((int64_t)b * Ts + sp) * 3 * H + g * H + u0 + 4 * hf);
        const int col = g * 8 + 4 * hf;
        ew_t[(col + 0) * NP + P] = e.x; ew_t[(col + 1) * NP + P] = e.y; ew_t[(col + 2) * NP + P] = e.z; ew_t[(col + 3) * NP + P] = e.w;
    }
    const int hq = (threadIdx.x >> 4) & 1;
    const int em = m0 + fr, eu = u0 + 4 * hq;
    const bool ep = threadIdx.x < 32, eok = ep && em < B;
    const int lrow = min(m0 + ld_row(lane), B - 1);        // the batch row this lane LOADS (quad-contiguous mapping, see ld_row)
    const int src4 = ld_src4(lane);
    gu32* cA = (gu32*)(a.cnt + ((int64_t)0 * a.RT + rt) * Tt * CNT_WORDS);
    gu32* cB = (gu32*)(a.cnt + ((int64_t)1 * a.RT + rt) * Tt * CNT_WORDS);
    gu32* cC = (gu32*)(a.cnt + ((int64_t)2 * a.RT + rt) * Tt * CNT_WORDS);
    bool dead = false;
    const float vq = a.v[16 * i + (threadIdx.x & 15)];               // attention vector, this thread's query column

    // gru_2 cell backward of step t for the own units (epilogue threads), given the total gradient dh2 of its output;
    // leaves dgi2 in LDS + memory, dgh2 published (sc1), z2 * dh2 in LDS
    auto cell2_bwd = [&](int t, const float (&dh2)[4], const float4 (&sv)[4], const float4 hp) {
        float gi[3][4], gh[3][4], dr[4];
        cell_bwd4(dh2, sv, hp, gi, gh, dr);
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            st_sc1_f4(a.dqgh + ((int64_t)t * B + em) * Q + C + g * H + eu, make_float4(gh[g][0], gh[g][1], gh[g][2], gh[g][3]));
            const float4 x = make_float4(gi[g][0], gi[g][1], gi[g][2], gi[g][3]);
            *reinterpret_cast<float4*>(gi_s + fr * 24 + g * 8 + 4 * hq) = x;
            *reinterpret_cast<float4*>(a.dgi2 + ((int64_t)t * B + em) * 3 * H + g * H + eu) = x;
        }
        *reinterpret_cast<float4*>(d1_s + fr * 8 + 4 * hq) = make_float4(dr[0], dr[1], dr[2], dr[3]);
    };
    // sum of the 8 waves' partial tiles for this epilogue thread's 4 units (tile rows 4 hq + i, batch row fr)
    auto red4 = [&](float (&x)[4]) {
        float4 sum = red[hq * 16 + fr];
#pragma unroll
        for (int w = 1; w < 8; ++w) {
            const float4 o = red[w * 64 + hq * 16 + fr];
            sum.x += o.x; sum.y += o.y; sum.z += o.z; sum.w += o.w;
        }
        x[0] = sum.x; x[1] = sum.y; x[2] = sum.z; x[3] = sum.w;
    };

    VAG_BPSTAMP(1);
    // ---- step Tt-1: nothing arrives from a later step
    if (ep) {
        if (eok) {
            const float4 d = *reinterpret_cast<const float4*>(a.d_h2_all + (int64_t)(Tt - 1) * BH + (int64_t)em * H + eu);
            const float dh2[4] = {d.x, d.y, d.z, d.w};
            const int64_t o = (int64_t)em * H + eu;
            float4 sv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) sv[q] = *reinterpret_cast<const float4*>(a.g2 + ((int64_t)(Tt - 1) * 4 + q) * BH + o);
            cell2_bwd(Tt - 1, dh2, sv, *reinterpret_cast<const float4*>(a.h1 + (int64_t)(Tt - 1) * BH + o));
        } else {
#pragma unroll
            for (int g = 0; g < 3; ++g) *reinterpret_cast<float4*>(gi_s + fr * 24 + g * 8 + 4 * hq) = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(d1_s + fr * 8 + 4 * hq) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();

#ifdef VAG_LAB          // phase timestamps: lab builds only (make LAB=1 -> libvagnmt_lab.so); the product kernels carry no hook
    const bool stamp = a.dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
#define VAG_STAMP(k) do { if (stamp) a.dbg[t * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define VAG_STAMP(k) do { } while (0)
#endif
    for (int t = Tt - 1; t >= 0; --t) {
        VAG_STAMP(0);
        // ================= A: d alpha shares of the own gate columns (atomics); dgh2[t] was published by cell2_bwd =================
        for (int P = threadIdx.x; P < NP; P += 512) {
            const int r = P / Ts;
            const float* g = gi_s + r * 24;
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 24; ++k) acc += ew_t[k * NP + P] * g[k];
            if (m0 + r < B)
                atomicAdd(a.dal + (((int64_t)t * ACC_SHARDS + (i & (ACC_SHARDS - 1))) * B + m0 + r) * Ts + (P - r * Ts), acc);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every wave's atomics and the epilogue's sc1 stores are done ...
        __syncthreads();
        if (threadIdx.x == 0) arrive(cA + t * CNT_WORDS, i);
        VAG_STAMP(1);
        // operands of phase B that do not depend on the hand-off
        float4 hga[KC], hgb[KC];                                // dgh2 rows of the hidden-side product, requested in phase B (its weight rows
                                                                // too: 70 instead of 18 spilled registers -- they stay where they are used)
        static_assert(KC == 6 || KC == 3, "ld_acc_shards_and_rows<KC>");
        float al0 = 0.f, al1 = 0.f, dh0 = 0.f, dh1_ = 0.f;
        const int x0 = threadIdx.x, x1 = threadIdx.x + 512;
        {
            const int r0 = min(x0, NP - 1) / Ts, r1 = min(x1, NP - 1) / Ts;
            const int64_t o0 = ((int64_t)t * B + min(m0 + r0, B - 1)) * Ts + (min(x0, NP - 1) - r0 * Ts);
            const int64_t o1 = ((int64_t)t * B + min(m0 + r1, B - 1)) * Ts + (min(x1, NP - 1) - r1 * Ts);
            al0 = a.alpha[o0]; dh0 = a.dah[o0];
            if (x1 < NP) { al1 = a.alpha[o1]; dh1_ = a.dah[o1]; }
            const float qv = a.qhp[((int64_t)t * B + min(m0 + (int)((threadIdx.x & 255) >> 4), B - 1)) * Q + 16 * i + (threadIdx.x & 15)];
            // ================= B: complete d alpha -> ds -> dq of the own query columns =================
            wait_count(cA + t * CNT_WORDS, PER_SHARD, a.err, a.guard, a.spin, dead);
            VAG_STAMP(2);
            float v0, v1;
            const float* p0 = a.dal + o0 + (int64_t)t * (ACC_SHARDS - 1) * B * Ts;      // (o0 = (t B + row) Ts + s: shard 0 of step t)
            const float* p1 = a.dal + o1 + (int64_t)t * (ACC_SHARDS - 1) * B * Ts;
            // (with them: this wave's share of the dgh2 rows for the hidden-side product below -- complete since cA as well)
            ld_acc_shards_and_rows(p0, p1, (int64_t)B * Ts, v0, v1,
                                   a.dqgh + ((int64_t)t * B + lrow) * Q + C + wave * (3 * H >> 3) + 8 * (lane & 3), hga, hgb);
            if (x0 < NP) { da_s[x0] = v0 + dh0; al_s[x0] = al0; }
            if (x1 < NP) { da_s[x1] = v1 + dh1_; al_s[x1] = al1; }
            __syncthreads();
            for (int r = wave; r < 16; r += 8) {               // softmax backward, one wave per row
                float dot = 0.f;
                for (int sp = lane; sp < Ts; sp += 64) dot += al_s[r * Ts + sp] * da_s[r * Ts + sp];
                dot = wave_sum(dot);
                for (int sp = lane; sp < Ts; sp += 64) {
                    const float d = al_s[r * Ts + sp] * (da_s[r * Ts + sp] - dot);
                    da_s[r * Ts + sp] = d;
                    if (((r * Ts + sp) & (WGS - 1)) == i && m0 + r < B) a.ds[((int64_t)t * B + m0 + r) * Ts + sp] = d;
                }
            }
            __syncthreads();
            {   // thread (half, row r, column c): the positions of its parity; the halves meet in the (idle) reduction space
                const int tt = threadIdx.x & 255, half = threadIdx.x >> 8;
                const int r = tt >> 4, c = tt & 15;
                const float* pr = pe_s + (int64_t)r * Ts * 16 + c;
                const float* dsr = da_s + r * Ts;
                float acc0 = 0.f, acc1 = 0.f;
                int sp = half;
                for (; sp + 2 < Ts; sp += 4) {
                    const float t0 = vag_tanh(pr[sp * 16] + qv), t1 = vag_tanh(pr[(sp + 2) * 16] + qv);
                    acc0 += dsr[sp] * (1.f - t0 * t0);
                    acc1 += dsr[sp + 2] * (1.f - t1 * t1);
                }
                if (sp < Ts) { const float t0 = vag_tanh(pr[sp * 16] + qv); acc0 += dsr[sp] * (1.f - t0 * t0); }
                float* half_s = reinterpret_cast<float*>(red);
                if (half) half_s[tt] = acc0 + acc1;
                __syncthreads();
                if (!half) dq_s[tt] = ((acc0 + acc1) + half_s[tt]) * vq;
            }
            __syncthreads();
            if (threadIdx.x < 64) {                             // 16 rows x 4 column quads: one 16-byte sc1 store each
                const int r = threadIdx.x >> 2, c4 = threadIdx.x & 3;
                if (m0 + r < B) st_sc1_f4(a.dqgh + ((int64_t)t * B + m0 + r) * Q + 16 * i + 4 * c4, *reinterpret_cast<const float4*>(dq_s + r * 16 + 4 * c4));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (threadIdx.x == 0) arrive(cB + t * CNT_WORDS, i);
            }
        }
        VAG_STAMP(3);
        // ---- beside that hand-off: hidden side of dh1 for the own units, dgh2[t] W_hh2 + z2 * dh2 (dgh2 rows: complete since cA)
        {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KC; ++s) {
                bf16x8 wf[3], hf[3];
                split8(perm4(*reinterpret_cast<const float4*>(wa_row + 32 * s), src4),
                       perm4(*reinterpret_cast<const float4*>(wa_row + 32 * s + 4), src4), wf);
                split8(perm4(hga[s], src4), perm4(hgb[s], src4), hf);
                acc = mma6(wf, hf, acc);
            }
            red[wave * 64 + lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
            __syncthreads();
            if (ep) {
                float x[4];
                red4(x);
                const float4 d1 = *reinterpret_cast<const float4*>(d1_s + fr * 8 + 4 * hq);
                *reinterpret_cast<float4*>(pb_s + fr * 8 + 4 * hq) = make_float4(x[0] + d1.x, x[1] + d1.y, x[2] + d1.z, x[3] + d1.w);
            }
        }
        // ================= C: dh1 = dq attn_h + hidden side -> gru_1 cell backward =================
        float4 s1[4], hp1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (eok) {
            const int64_t o = (int64_t)em * H + eu;
#pragma unroll
            for (int q = 0; q < 4; ++q) s1[q] = *reinterpret_cast<const float4*>(a.g1 + ((int64_t)t * 4 + q) * BH + o);
            hp1 = *reinterpret_cast<const float4*>((t > 0 ? a.h2_all + (int64_t)(t - 1) * BH : a.h0) + o);
        }
        VAG_STAMP(4);
        wait_count(cB + t * CNT_WORDS, PER_SHARD, a.err, a.guard, a.spin, dead);
        VAG_STAMP(5);
        {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            float4 ga[KB], gb[KB];
            ld_rows_sc1<KB>(a.dqgh + ((int64_t)t * B + lrow) * Q + wave * (C >> 3) + 8 * (lane & 3), ga, gb);
#pragma unroll
            for (int s = 0; s < KB; ++s) { ga[s] = perm4(ga[s], src4); gb[s] = perm4(gb[s], src4); }
#pragma unroll
            for (int s = 0; s < KB; ++s) {
                bf16x8 wf[3], hf[3];
                split8(*reinterpret_cast<const float4*>(wb_row + 32 * s), *reinterpret_cast<const float4*>(wb_row + 32 * s + 4), wf);
                split8(ga[s], gb[s], hf);
                acc = mma6(wf, hf, acc);
            }
            red[wave * 64 + lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
        __syncthreads();
        if (ep) {
            float x[4];
            red4(x);
            if (eok) {
                const float4 pbv = *reinterpret_cast<const float4*>(pb_s + fr * 8 + 4 * hq);
                const float dh1[4] = {x[0] + pbv.x, x[1] + pbv.y, x[2] + pbv.z, x[3] + pbv.w};
                float gi[3][4], gh[3][4], dr[4];
                cell_bwd4(dh1, s1, hp1, gi, gh, dr);
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    st_sc1_f4(a.dgh1 + ((int64_t)t * B + em) * 3 * H + g * H + eu, make_float4(gh[g][0], gh[g][1], gh[g][2], gh[g][3]));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (threadIdx.x == 0) arrive(cC + t * CNT_WORDS, i);
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    *reinterpret_cast<float4*>(a.dgi1 + ((int64_t)t * B + em) * 3 * H + g * H + eu) = make_float4(gi[g][0], gi[g][1], gi[g][2], gi[g][3]);
                *reinterpret_cast<float4*>(c1_s + fr * 8 + 4 * hq) = make_float4(dr[0], dr[1], dr[2], dr[3]);
            } else if (threadIdx.x == 0) {
                arrive(cC + t * CNT_WORDS, i);
            }
        }
        // ================= D: dh2[t-1] = dgh1 W_hh1 + z1 * dh1 (+ the head's part) -> gru_2 cell backward of step t-1 =================
        float4 dadd = make_float4(0.f, 0.f, 0.f, 0.f), s2[4], hp2 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (eok && t > 0) {           // operands of the gru_2 cell backward of step t-1: requested before the wait
            const int64_t o = (int64_t)em * H + eu;
            dadd = *reinterpret_cast<const float4*>(a.d_h2_all + (int64_t)(t - 1) * BH + o);
#pragma unroll
            for (int q = 0; q < 4; ++q) s2[q] = *reinterpret_cast<const float4*>(a.g2 + ((int64_t)(t - 1) * 4 + q) * BH + o);
            hp2 = *reinterpret_cast<const float4*>(a.h1 + (int64_t)(t - 1) * BH + o);
        }
        VAG_STAMP(6);
        wait_count(cC + t * CNT_WORDS, PER_SHARD, a.err, a.guard, a.spin, dead);
        VAG_STAMP(7);
        {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            float4 ga[KC], gb[KC];
            ld_rows_sc1<KC>(a.dgh1 + ((int64_t)t * B + lrow) * 3 * H + wave * (3 * H >> 3) + 8 * (lane & 3), ga, gb);
#pragma unroll
            for (int s = 0; s < KC; ++s) { ga[s] = perm4(ga[s], src4); gb[s] = perm4(gb[s], src4); }
#pragma unroll
            for (int s = 0; s < KC; ++s) {
                bf16x8 hf[3];
                split8(ga[s], gb[s], hf);
                acc = mma6(wc[s], hf, acc);
            }
            red[wave * 64 + lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
        __syncthreads();
        if (ep) {
            float x[4];
            red4(x);
            if (eok) {
                const float4 c1 = *reinterpret_cast<const float4*>(c1_s + fr * 8 + 4 * hq);
                const float dh2[4] = {x[0] + c1.x + dadd.x, x[1] + c1.y + dadd.y, x[2] + c1.z + dadd.z, x[3] + c1.w + dadd.w};
                if (t > 0) cell2_bwd(t - 1, dh2, s2, hp2);
                else {
                    float4 o = make_float4(dh2[0], dh2[1], dh2[2], dh2[3]);
                    if (a.h0_tanh) {
                        const float4 h = *reinterpret_cast<const float4*>(a.h0 + (int64_t)em * H + eu);
                        o.x *= 1.f - h.x * h.x; o.y *= 1.f - h.y * h.y; o.z *= 1.f - h.z * h.z; o.w *= 1.f - h.w * h.w;
                    }
                    *reinterpret_cast<float4*>(a.d_h0 + (int64_t)em * H + eu) = o;
                }
            }
        }
        __syncthreads();
    }
    VAG_BPSTAMP(2);
#undef VAG_STAMP
}

__global__ __launch_bounds__(256) void zero_u32_kernel(unsigned* p, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0u;
}
// two regions in one launch (the decoder kernels' counters and their atomically accumulated scores)
__global__ __launch_bounds__(256) void zero2_u32_kernel(unsigned* p, int n, unsigned* q, int m) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0u;
    else if (i - n < m) q[i - n] = 0u;
}

}  // namespace

// Eligibility: hidden size a multiple of 256 up to 1024 (register budget of the weight planes), all workgroups resident at
// once (one per CU).
// Optional HIP-event timing of the four recurrence kernels (option "persist_timing"; eager launches only -- events cannot be
// read back from inside a stream capture).  kind: 0 encoder forward, 1 decoder forward, 2 encoder backward, 3 decoder backward.
// vag_train_step zeroes every counter / exchange buffer of a step's recurrence kernels in its prologue launch (one launch
// instead of four) and says so for the duration of its call: the launch functions below then skip their own zeroing.
static thread_local bool g_prezeroed = false;
void vag_persist_set_prezeroed(bool v) { g_prezeroed = v; }

struct PersistTimer { hipEvent_t e0 = nullptr, e1 = nullptr; bool pending = false; double ms = 0.0; int n = 0; };
static PersistTimer g_ptimer[4];
static std::mutex g_ptimer_mu;          // measurement state, touched only with "persist_timing" on: one lock for the four timers
static void ptimer_collect(PersistTimer& t) {
    if (!t.pending) return;
    float ms = 0.f;
    if (hipEventSynchronize(t.e1) == hipSuccess && hipEventElapsedTime(&ms, t.e0, t.e1) == hipSuccess) { t.ms += ms; ++t.n; }
    t.pending = false;
}
static bool ptimer_begin(int kind, hipStream_t s) {
    if (!vag_opt().persist_timing) return false;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (st != hipStreamCaptureStatusNone) return false;
    std::lock_guard<std::mutex> lk(g_ptimer_mu);
    PersistTimer& t = g_ptimer[kind];
    ptimer_collect(t);
    if (!t.e0 && (hipEventCreate(&t.e0) != hipSuccess || hipEventCreate(&t.e1) != hipSuccess)) return false;
    return hipEventRecord(t.e0, s) == hipSuccess;
}
static void ptimer_end(int kind, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_ptimer_mu);
    if (hipEventRecord(g_ptimer[kind].e1, s) == hipSuccess) g_ptimer[kind].pending = true;
}
int vag_persistent_time_read(int kind, double* ms_total, int* launches) {
    VAG_CHECK_ARG(kind >= 0 && kind < 4 && ms_total && launches);
    std::lock_guard<std::mutex> lk(g_ptimer_mu);
    PersistTimer& t = g_ptimer[kind];
    ptimer_collect(t);
    *ms_total = t.ms; *launches = t.n;
    t.ms = 0.0; t.n = 0;
    return VAG_OK;
}

// CU count and LDS capacity of the CURRENT device, cached per device id (a process may drive several devices from several
// threads: atomics, no locks; a racing first query writes the same values twice).
struct PersistDev { std::atomic<int> cus{-1}; std::atomic<int> lds{-1}; };
static PersistDev g_pdev[64];
static PersistDev* persist_dev() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    PersistDev& d = g_pdev[dev];
    if (d.cus.load(std::memory_order_acquire) < 0) {
        int n = 0, l = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        if (hipDeviceGetAttribute(&l, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) l = 0;
        d.lds.store(l, std::memory_order_release);
        d.cus.store(n, std::memory_order_release);
    }
    return &d;
}
static int persist_cu_count() {
    PersistDev* d = persist_dev();
    return d ? d->cus.load(std::memory_order_acquire) : 0;
}
// The kernels are written for gfx950's 160 KB of LDS per CU (84 KB static / up to 160 KB dynamic per workgroup): a device that
// offers less takes the launch chains.
static bool persist_lds_ok(int64_t bytes) {
    PersistDev* d = persist_dev();
    return d && (int64_t)d->lds.load(std::memory_order_acquire) >= bytes;
}
static unsigned spin_limit() {
    const int64_t v = vag_opt().persist_spin_limit;
    return v > 0 ? (unsigned)(v > 0x7fffffff ? 0x7fffffff : v) : SPIN_LIMIT;
}
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel and device, from whichever thread comes first
struct AttrOnce { std::atomic<unsigned long long> done{0}; };
static bool set_max_lds_once(AttrOnce& o, const void* fn) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    const unsigned long long bit = 1ull << dev;
    if (o.done.load(std::memory_order_acquire) & bit) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
    o.done.fetch_or(bit, std::memory_order_release);
    return true;
}
// Row tiles one launch of the encoder kernels holds (both directions, one workgroup per CU), and the pass policy of the decoder
// kernels (dec_passes_ok below): a wider batch goes in at most two passes, the second at least three quarters full.
static int enc_tiles_per_pass(int64_t H) {
    const int cus = persist_cu_count();
    return (H == 256 || H == 512 || H == 1024) ? (int)(cus / (2 * (H / 16))) : 0;
}
static bool passes_ok(int64_t rt, int tpp, int max_passes) {
    if (tpp <= 0 || rt > (int64_t)tpp * max_passes) return false;
    return rt <= tpp || 4 * (rt - tpp * ((rt - 1) / tpp)) >= 3 * tpp;
}
bool vag_enc_persistent_ok(int64_t B, int64_t Ts, int64_t H) {
    if (!(H == 256 || H == 512 || H == 1024) || B <= 0 || Ts <= 0) return false;
    return passes_ok(cdiv64(B, 16), enc_tiles_per_pass(H), 2) && persist_lds_ok(84 * 1024) && (int64_t)(Ts + 1) * B * H * 4 < (1ll << 31);
}
int64_t vag_enc_persistent_sync_words(int64_t B, int64_t Ts) { return 2 * cdiv64(B, 16) * Ts + 64; }

int vag_enc_fwd_persistent_launch(const float* xp, const float* w_fw, const float* w_bw, const float* b_fw, const float* b_bw,
                                  const int* lengths, float* hst, float* gates, float* enc, unsigned* sync, const uint64_t* rng,
                                  float p_ctx, int64_t B, int64_t Ts, int64_t H, hipStream_t s) {
    VAG_CHECK_ARG(xp && w_fw && w_bw && b_fw && b_bw && lengths && hst && gates && enc && sync && vag_enc_persistent_ok(B, Ts, H));
    VAG_CHECK_ARG(aligned16(xp) && aligned16(w_fw) && aligned16(w_bw) && aligned16(b_fw) && aligned16(b_bw) && aligned16(hst) &&
                  aligned16(gates) && aligned16(enc));
    EncPArgs a;
    a.xp = xp; a.W[0] = w_fw; a.W[1] = w_bw; a.bias[0] = b_fw; a.bias[1] = b_bw; a.lengths = lengths;
    a.hst = hst; a.gates = gates; a.enc = enc; a.rng = rng; a.p_ctx = p_ctx;
    a.B = (int)B; a.Ts = (int)Ts; a.H = (int)H; a.RT = (int)cdiv64(B, 16); a.CS = (int)(H / 16);
    const int nwords = (int)vag_enc_persistent_sync_words(B, Ts);
    a.cnt = sync; a.err = sync + (nwords - 64); a.spin = spin_limit(); a.guard = vag_persist_guard();
    if (!g_prezeroed) {
        hipLaunchKernelGGL(zero_u32_kernel, dim3((unsigned)cdiv64(nwords, 256)), dim3(256), 0, s, sync, nwords);
        VAG_LAUNCH_CHECK();
    }
    const bool timed = ptimer_begin(0, s);
    const int tpp = enc_tiles_per_pass(H);
    for (a.rt0 = 0; a.rt0 < a.RT; a.rt0 += tpp) {                    // passes of row tiles (one at B <= 64 for H = 512)
        a.RTP = std::min(tpp, a.RT - a.rt0);
        const dim3 grid((unsigned)(2 * a.RTP * a.CS));
        if (H == 256) hipLaunchKernelGGL(enc_fwd_persistent_kernel<1>, grid, dim3(512), 0, s, a);
        else if (H == 512) hipLaunchKernelGGL(enc_fwd_persistent_kernel<2>, grid, dim3(512), 0, s, a);
        else hipLaunchKernelGGL(enc_fwd_persistent_kernel<4>, grid, dim3(512), 0, s, a);
    }
    if (timed) ptimer_end(0, s);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// Eligibility of the persistent decoder: H = 512 or 256 (8 units per workgroup x H / 8 workgroups per row tile), keys of a row tile
// fit the LDS, 4-float alignment of the row strides.  One launch holds as many 16-row tiles as the chip has room for, one
// workgroup per CU (B <= 64 at H = 512, <= 128 at H = 256 on 256 CUs); a wider batch is taken in PASSES of row tiles, launch after
// launch -- batch rows never meet inside the recurrence, and every per-row array is indexed by the global row, so a pass is the
// same kernel with a row-tile offset.  A pass costs what a full launch costs however many tiles it holds, while the launch chains
// grow with the batch, so passes pay only while they are few and full (tools/exp_dec_passes.py, whole optimiser steps, Ts = Tt = 40:
// H = 512 B = 128 5.43 vs 5.99 ms, B = 96 5.01 vs 4.74, B = 192 8.20 vs 8.00, B = 256 10.3 vs 9.3; H = 256 B = 256 5.18 vs 5.33,
// B = 384 7.74 vs 7.57): at most DEC_MAX_PASSES = 2, the last one at least three quarters full.
constexpr int DEC_MAX_PASSES = 2;
static int64_t dec_persistent_lds_bytes(int64_t Ts, bool free_run = false) {
    return 4 * (6144 + 16 * Ts * 16 + 16 * Ts * 24 + 16 * Ts + 16 * 24 + 16 * 24 + 80 + 256 + 1024 + 16 * Ts + 256 +
                (free_run ? 64 + 16 + 16 + 16 * Ts * 4 : 0));
}
static int dec_wgs(int64_t H) { return (int)(H / DEC_U); }                   // workgroups per row tile
static int dec_tiles_per_pass(int64_t H) {                                   // row tiles one launch holds (0: none)
    const int cus = persist_cu_count();
    return (H == 256 || H == 512) ? cus / dec_wgs(H) : 0;
}
bool vag_dec_persistent_ok(int64_t B, int64_t Ts, int64_t Tt, int64_t H) {
    if ((H != 512 && H != 256) || B <= 0 || Ts <= 0 || Tt <= 0 || Ts > 512) return false;
    const int tpp = dec_tiles_per_pass(H);
    if (!passes_ok(cdiv64(B, 16), tpp, DEC_MAX_PASSES)) return false;                     // (a mostly empty last pass: the chains win)
    return dec_persistent_lds_bytes(Ts) <= 160 * 1024 && persist_lds_ok(160 * 1024);
}
int64_t vag_dec_persistent_sync_words(int64_t B, int64_t Tt) { return 5 * cdiv64(B, 16) * Tt * CNT_WORDS + 64; }

int vag_dec_fwd_persistent_launch(const float* pe, const float* mask, const float* h0, const float* xp1, const float* W1,
                                  const float* b1, const float* wcat, const float* bcat, const float* v, const float* encwp,
                                  const float* b_ih2, float* h1, float* g1, float* qhp, float* alpha, float* h2_all, float* g2,
                                  float* psc, unsigned* sync, int64_t B, int64_t Ts, int64_t Tt, int64_t H, hipStream_t s) {
    VAG_CHECK_ARG(pe && mask && h0 && xp1 && W1 && b1 && wcat && bcat && v && encwp && b_ih2 && h1 && g1 && qhp && alpha && h2_all &&
                  g2 && psc && sync && vag_dec_persistent_ok(B, Ts, Tt, H));
    VAG_CHECK_ARG(aligned16(pe) && aligned16(h0) && aligned16(xp1) && aligned16(W1) && aligned16(b1) && aligned16(wcat) &&
                  aligned16(bcat) && aligned16(v) && aligned16(encwp) && aligned16(b_ih2) && aligned16(h1) && aligned16(g1) &&
                  aligned16(qhp) && aligned16(h2_all) && aligned16(g2));
    DecPArgs a = {};
    a.pe = pe; a.mask = mask; a.h0 = h0; a.xp1 = xp1; a.W1 = W1; a.b1 = b1; a.wcat = wcat; a.bcat = bcat; a.v = v; a.encwp = encwp;
    a.b_ih2 = b_ih2; a.h1 = h1; a.g1 = g1; a.qhp = qhp; a.alpha = alpha; a.h2_all = h2_all; a.g2 = g2; a.psc = psc;
    a.B = (int)B; a.Ts = (int)Ts; a.Tt = (int)Tt; a.H = (int)H; a.RT = (int)cdiv64(B, 16);
    const int nwords = (int)vag_dec_persistent_sync_words(B, Tt);
    a.cnt = sync; a.err = sync + (nwords - 64); a.spin = spin_limit(); a.guard = vag_persist_guard(); a.xcd_map = vag_opt().dec_xcd_map & 15;
#ifdef VAG_LAB
    a.dbg = reinterpret_cast<unsigned long long*>(vag_opt().dec_stamps);
#else
    a.dbg = nullptr;
#endif
    const int nsc = (int)(Tt * B * Ts) * ACC_SHARDS;                 // the scores are accumulated with atomics: start from zero
    if (!g_prezeroed) {
        hipLaunchKernelGGL(zero2_u32_kernel, dim3((unsigned)cdiv64((int64_t)nwords + nsc, 256)), dim3(256), 0, s, sync, nwords,
                           reinterpret_cast<unsigned*>(psc), nsc);
        VAG_LAUNCH_CHECK();
    }
    if (VAG_TAGGED && !g_prezeroed) {        // the marked hand-off buffers start from zero (tag1; the step driver's prologue did it)
        const int nh = (int)(Tt * B * H);
        hipLaunchKernelGGL(zero2_u32_kernel, dim3((unsigned)cdiv64(2 * (int64_t)nh, 256)), dim3(256), 0, s,
                           reinterpret_cast<unsigned*>(h1), nh, reinterpret_cast<unsigned*>(h2_all), nh);
        VAG_LAUNCH_CHECK();
    }
    int64_t lds = dec_persistent_lds_bytes(Ts);
    if (lds < 84 * 1024) lds = 84 * 1024;                            // never two workgroups on one CU
    static AttrOnce once;
    if (!set_max_lds_once(once, reinterpret_cast<const void*>(dec_fwd_persistent_kernel<false>))) return VAG_EINVAL;
    static AttrOnce once256;
    if (H == 256 && !set_max_lds_once(once256, reinterpret_cast<const void*>(dec_fwd_persistent_kernel<false, 32>))) return VAG_EINVAL;
    const bool timed = ptimer_begin(1, s);
    const int tpp = dec_tiles_per_pass(H);
    for (a.rt0 = 0; a.rt0 < a.RT; a.rt0 += tpp) {                    // passes of row tiles (one at B <= 64 / 128)
        const unsigned tiles = (unsigned)std::min(tpp, a.RT - a.rt0);
        if (H == 512) hipLaunchKernelGGL((dec_fwd_persistent_kernel<false, 64>), dim3(tiles * 64), dim3(512), (size_t)lds, s, a);
        else hipLaunchKernelGGL((dec_fwd_persistent_kernel<false, 32>), dim3(tiles * 32), dim3(512), (size_t)lds, s, a);
    }
    if (timed) ptimer_end(1, s);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// Free-running form of the same kernel (V11.py:148-160 training without teacher forcing, :207-226 greedy decoding): the kernel
// computes the head's hidden layer, the logits of its own vocabulary tiles and the arg-max itself and feeds the token back,
// two more hand-offs per step.  E = 256 (four head columns per workgroup); the keys' share of the head is one more LDS slice.
bool vag_dec_free_persistent_ok(int64_t B, int64_t Ts, int64_t Tt, int64_t E, int64_t H, int64_t V) {
    if (H != 512 || E != DEC_E || B <= 0 || B > 64 || Ts <= 0 || Tt <= 0 || Ts > 512 || V < 16 || V >= (1ll << 30)) return false;
    const int cus = persist_cu_count();
    return cdiv64(B, 16) * DEC_WGS <= cus && dec_persistent_lds_bytes(Ts, true) <= 160 * 1024 && persist_lds_ok(160 * 1024);
}
// tables: [embp (V,3H) | embw3 (V,E) | encw2 (B,Ts,E) | cand (Tt,B,64) 8-byte words]
int64_t vag_dec_free_tables_floats(int64_t B, int64_t Ts, int64_t Tt, int64_t E, int64_t H, int64_t V) {
    auto r = [](int64_t n) { return (n + 63) & ~63ll; };
    return r(V * 3 * H) + r(V * E) + r(B * Ts * E) + r(2 * Tt * B * DEC_WGS);
}
int vag_dec_free_persistent_launch(const float* pe, const float* mask, const float* h0, const float* W1, const float* b1,
                                   const float* wcat, const float* bcat, const float* v, const float* encwp, const float* b_ih2,
                                   float* h1, float* g1, float* qhp, float* alpha, float* h2_all, float* g2, float* psc,
                                   unsigned* sync, const float* tables, const float* hw1, const float* hb1, const float* hb2,
                                   const float* hb3, const float* out_w, const float* out_b, float* tmid, float* logits, int64_t ldl,
                                   int64_t* tok, const uint64_t* rng, float p_out, int64_t B, int64_t Ts, int64_t Tt, int64_t E,
                                   int64_t H, int64_t V, hipStream_t s) {
    VAG_CHECK_ARG(pe && mask && h0 && W1 && b1 && wcat && bcat && v && encwp && b_ih2 && h1 && g1 && qhp && alpha && h2_all && g2 &&
                  psc && sync && tables && hw1 && hb1 && hb2 && hb3 && out_w && out_b && tmid && tok &&
                  vag_dec_free_persistent_ok(B, Ts, Tt, E, H, V));
    VAG_CHECK_ARG(aligned16(pe) && aligned16(h0) && aligned16(W1) && aligned16(b1) && aligned16(wcat) && aligned16(bcat) &&
                  aligned16(v) && aligned16(encwp) && aligned16(b_ih2) && aligned16(h1) && aligned16(g1) && aligned16(qhp) &&
                  aligned16(h2_all) && aligned16(g2) && aligned16(tables) && aligned16(hw1) && aligned16(out_w) && aligned16(tmid));
    VAG_CHECK_ARG(!logits || (aligned16(logits) && ldl >= V && ldl % 4 == 0));
    auto r = [](int64_t n) { return (n + 63) & ~63ll; };
    DecPArgs a = {};
    a.pe = pe; a.mask = mask; a.h0 = h0; a.xp1 = nullptr; a.W1 = W1; a.b1 = b1; a.wcat = wcat; a.bcat = bcat; a.v = v; a.encwp = encwp;
    a.b_ih2 = b_ih2; a.h1 = h1; a.g1 = g1; a.qhp = qhp; a.alpha = alpha; a.h2_all = h2_all; a.g2 = g2; a.psc = psc;
    a.B = (int)B; a.Ts = (int)Ts; a.Tt = (int)Tt; a.H = (int)H; a.RT = (int)cdiv64(B, 16);
    a.embp = tables; a.embw3 = a.embp + r(V * 3 * H); a.encw2 = a.embw3 + r(V * E);
    a.cand = reinterpret_cast<unsigned long long*>(const_cast<float*>(a.encw2 + r(B * Ts * E)));
    a.hw1 = hw1; a.hb1 = hb1; a.hb2 = hb2; a.hb3 = hb3; a.out_w = out_w; a.out_b = out_b; a.tmid = tmid; a.logits = logits;
    a.tok = tok; a.rng = rng; a.p_out = p_out; a.V = (int)V; a.ldl = (int)ldl;
    const int nwords = (int)vag_dec_persistent_sync_words(B, Tt);
    a.cnt = sync; a.err = sync + (nwords - 64); a.spin = spin_limit(); a.guard = vag_persist_guard(); a.xcd_map = vag_opt().dec_xcd_map & 15;
#ifdef VAG_LAB
    a.dbg = reinterpret_cast<unsigned long long*>(vag_opt().dec_stamps);       // (Tt + 1) x 16 words in this form
#else
    a.dbg = nullptr;
#endif
    const int nsc = (int)(Tt * B * Ts) * ACC_SHARDS;
    if (!g_prezeroed) {
        hipLaunchKernelGGL(zero2_u32_kernel, dim3((unsigned)cdiv64((int64_t)nwords + nsc, 256)), dim3(256), 0, s, sync, nwords,
                           reinterpret_cast<unsigned*>(psc), nsc);
        VAG_LAUNCH_CHECK();
    }
    if (VAG_TAGGED && !g_prezeroed) {
        const int nh = (int)(Tt * B * H);
        hipLaunchKernelGGL(zero2_u32_kernel, dim3((unsigned)cdiv64(2 * (int64_t)nh, 256)), dim3(256), 0, s,
                           reinterpret_cast<unsigned*>(h1), nh, reinterpret_cast<unsigned*>(h2_all), nh);
        VAG_LAUNCH_CHECK();
    }
    int64_t lds = dec_persistent_lds_bytes(Ts, true);
    if (lds < 84 * 1024) lds = 84 * 1024;
    static AttrOnce once;
    if (!set_max_lds_once(once, reinterpret_cast<const void*>(dec_fwd_persistent_kernel<true>))) return VAG_EINVAL;
    const bool timed = ptimer_begin(1, s);
    hipLaunchKernelGGL(dec_fwd_persistent_kernel<true>, dim3((unsigned)(a.RT * DEC_WGS)), dim3(512), (size_t)lds, s, a);
    if (timed) ptimer_end(1, s);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// Eligibility of the wide fp16 forward kernel: H a multiple of 128 with 4 or 8 k-steps per wave, every workgroup resident.
bool vag_enc_wide16_ok(int64_t B, int64_t Ts, int64_t H) {
    if (!(H == 512 || H == 1024) || B < 1 || Ts < 1) return false;
    const int64_t wgs = 2 * cdiv64(B, 64) * (H / 32);
    if (wgs > persist_cu_count() || !persist_lds_ok(160 * 1024)) return false;
    return 2 * cdiv64(B, 64) * Ts + 64 <= vag_enc_persistent_sync_words(B, Ts);
}
int vag_enc_fwd_wide16_launch(const float* xp, const vag_half* w16_fw, const vag_half* w16_bw, const float* b_fw, const float* b_bw,
                              const int* lengths, float* hst, float* gates, float* enc, vag_half* hx, unsigned* sync, const uint64_t* rng,
                              float p_ctx, int64_t B, int64_t Ts, int64_t H, hipStream_t s) {
    VAG_CHECK_ARG(xp && w16_fw && w16_bw && b_fw && b_bw && lengths && hst && gates && enc && hx && sync &&
                  vag_enc_wide16_ok(B, Ts, H));
    VAG_CHECK_ARG(aligned16(xp) && aligned16(w16_fw) && aligned16(w16_bw) && aligned16(b_fw) && aligned16(b_bw) && aligned16(hst) &&
                  aligned16(gates) && aligned16(enc) && aligned16(hx));
    EncWArgs a;
    a.xp = xp; a.W16[0] = w16_fw; a.W16[1] = w16_bw; a.bias[0] = b_fw; a.bias[1] = b_bw; a.lengths = lengths;
    a.hst = hst; a.gates = gates; a.enc = enc; a.hx = hx; a.rng = rng; a.p_ctx = p_ctx;
    a.B = (int)B; a.Ts = (int)Ts; a.H = (int)H; a.RG = (int)cdiv64(B, 64); a.CS = (int)(H / 32);
    const int nwords = (int)vag_enc_persistent_sync_words(B, Ts);
    a.cnt = sync; a.err = sync + (nwords - 64); a.spin = spin_limit(); a.guard = vag_persist_guard();
    if (!g_prezeroed) {
        hipLaunchKernelGGL(zero_u32_kernel, dim3((unsigned)cdiv64(nwords, 256)), dim3(256), 0, s, sync, nwords);
        VAG_LAUNCH_CHECK();
    }
    const size_t lds = 96 * 1024;
    static AttrOnce once4, once8;
    if (!set_max_lds_once(once4, reinterpret_cast<const void*>(enc_fwd_wide16_kernel<4>)) ||
        !set_max_lds_once(once8, reinterpret_cast<const void*>(enc_fwd_wide16_kernel<8>)))
        return VAG_EINVAL;
    const dim3 grid((unsigned)(2 * a.RG * a.CS));
    const bool timed = ptimer_begin(0, s);
    if (H == 512) hipLaunchKernelGGL(enc_fwd_wide16_kernel<4>, grid, dim3(512), lds, s, a);
    else hipLaunchKernelGGL(enc_fwd_wide16_kernel<8>, grid, dim3(512), lds, s, a);
    if (timed) ptimer_end(0, s);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

int vag_enc_bwd_wide16_launch(const vag_half* wt16, const float* d_enc, const float* gates, const float* hst, const int* lengths,
                              const uint64_t* rng, float p_ctx, float* d_xp, float* dgh, vag_half* gx, unsigned* sync, int64_t B,
                              int64_t Ts, int64_t H, hipStream_t s) {
    VAG_CHECK_ARG(wt16 && d_enc && gates && hst && lengths && d_xp && dgh && gx && sync && vag_enc_wide16_ok(B, Ts, H));
    VAG_CHECK_ARG(aligned16(wt16) && aligned16(d_enc) && aligned16(gates) && aligned16(hst) && aligned16(d_xp) && aligned16(dgh) &&
                  aligned16(gx));
    EncWBArgs a;
    a.WT16[0] = wt16; a.WT16[1] = wt16 + 3 * H * H; a.d_enc = d_enc; a.gates = gates; a.hst = hst; a.lengths = lengths;
    a.rng = rng; a.p_ctx = p_ctx; a.d_xp = d_xp; a.dgh = dgh; a.gx = gx;
    a.B = (int)B; a.Ts = (int)Ts; a.H = (int)H; a.RG = (int)cdiv64(B, 64); a.CS = (int)(H / 32);
    const int nwords = (int)vag_enc_persistent_sync_words(B, Ts);
    a.cnt = sync; a.err = sync + (nwords - 64); a.spin = spin_limit(); a.guard = vag_persist_guard();
    if (!g_prezeroed) {
        hipLaunchKernelGGL(zero_u32_kernel, dim3((unsigned)cdiv64(nwords, 256)), dim3(256), 0, s, sync, nwords);
        VAG_LAUNCH_CHECK();
    }
    const size_t lds = 84 * 1024;                      // 32 KB of reduction space; the rest keeps the CU to one workgroup
    static AttrOnce once6, once12;
    if (!set_max_lds_once(once6, reinterpret_cast<const void*>(enc_bwd_wide16_kernel<6>)) ||
        !set_max_lds_once(once12, reinterpret_cast<const void*>(enc_bwd_wide16_kernel<12>)))
        return VAG_EINVAL;
    const dim3 grid((unsigned)(2 * a.RG * a.CS));
    const bool timed = ptimer_begin(2, s);
    if (H == 512) hipLaunchKernelGGL(enc_bwd_wide16_kernel<6>, grid, dim3(512), lds, s, a);
    else hipLaunchKernelGGL(enc_bwd_wide16_kernel<12>, grid, dim3(512), lds, s, a);
    if (timed) ptimer_end(2, s);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

int vag_enc_bwd_persistent_launch(const float* whhT, const float* d_enc, const float* gates, const float* hst, const int* lengths,
                                  const uint64_t* rng, float p_ctx, float* d_xp, float* dgh, unsigned* sync, int64_t B, int64_t Ts,
                                  int64_t H, hipStream_t s) {
    VAG_CHECK_ARG(whhT && d_enc && gates && hst && lengths && d_xp && dgh && sync && (H == 512 || H == 256) && vag_enc_persistent_ok(B, Ts, H));
    VAG_CHECK_ARG(aligned16(whhT) && aligned16(d_enc) && aligned16(gates) && aligned16(hst) && aligned16(d_xp) && aligned16(dgh));
    EncBArgs a;
    a.WT[0] = whhT; a.WT[1] = whhT + 3 * H * H; a.d_enc = d_enc; a.gates = gates; a.hst = hst; a.lengths = lengths;
    a.rng = rng; a.p_ctx = p_ctx; a.d_xp = d_xp; a.dgh = dgh;
    a.B = (int)B; a.Ts = (int)Ts; a.H = (int)H; a.RT = (int)cdiv64(B, 16); a.CS = (int)(H / 16);
    const int nwords = (int)vag_enc_persistent_sync_words(B, Ts);
    a.cnt = sync; a.err = sync + (nwords - 64); a.spin = spin_limit(); a.guard = vag_persist_guard();
    if (!g_prezeroed) {
        hipLaunchKernelGGL(zero_u32_kernel, dim3((unsigned)cdiv64(nwords, 256)), dim3(256), 0, s, sync, nwords);
        VAG_LAUNCH_CHECK();
    }
    const bool timed = ptimer_begin(2, s);
    const int tpp = enc_tiles_per_pass(H);
    for (a.rt0 = 0; a.rt0 < a.RT; a.rt0 += tpp) {
        a.RTP = std::min(tpp, a.RT - a.rt0);
        const dim3 grid((unsigned)(2 * a.RTP * a.CS));
        if (H == 512) hipLaunchKernelGGL(enc_bwd_persistent_kernel<6>, grid, dim3(512), 0, s, a);
        else hipLaunchKernelGGL(enc_bwd_persistent_kernel<3>, grid, dim3(512), 0, s, a);
    }
    if (timed) ptimer_end(2, s);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// The guard pair of launches enqueued by the calling thread: the caller's own (vag_train_step: vag_step_cfg.guard; operators:
// vag_set_operator_guard) or the process-wide pair.
static thread_local unsigned* g_thread_guard = nullptr;
void vag_persist_guard_set(unsigned* g) { g_thread_guard = g; }
unsigned* vag_persist_guard_peek(void) { return g_thread_guard; }
unsigned* vag_persist_guard(void) {
    if (g_thread_guard) return g_thread_guard;
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_persist_guard)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return reinterpret_cast<unsigned*>(p);
}
// Number of waits that gave up since the last call (synchronises the device).  0 in a healthy run.
// Reading the count also takes the process-wide guard pair down: whoever polls the count has been told about the give-ups, and
// nothing else resets that pair (a driver's own pair is reset by its optimiser kernels / its check()).
int vag_persistent_timeouts_read(void) {
    unsigned v = 0, z = 0;
    const unsigned zz[2] = {0u, 0u};
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_persist_timeouts), sizeof(v)) != hipSuccess) return -1;
    if (v != 0 && (hipMemcpyToSymbol(HIP_SYMBOL(g_persist_timeouts), &z, sizeof(z)) != hipSuccess ||
                   hipMemcpyToSymbol(HIP_SYMBOL(g_persist_guard), zz, sizeof(zz)) != hipSuccess)) return -1;
    return (int)v;
}

static int64_t dec_bwd_persistent_lds_bytes(int64_t Ts, int64_t H = 512) {
    const int64_t NP = 16 * Ts;
    return 4 * (2048 + NP * 16 + 24 * NP + 2 * NP + 384 + 3 * 128 + 256 + 8 * (2 * H + 4) + 64);
}
bool vag_dec_bwd_persistent_ok(int64_t B, int64_t Ts, int64_t Tt, int64_t H) {
    return vag_dec_persistent_ok(B, Ts, Tt, H) && dec_bwd_persistent_lds_bytes(Ts, H) <= 160 * 1024;
}
// A step driver whose initial state is h0 = tanh(.) asks the next backward launch of the calling thread to apply the tanh's
// derivative to d_h0 on its way out (a launch saved); vag_persist_dh0_tanh_done(d_h0) tells the consumer whether it happened.
static thread_local bool g_dh0_tanh_req = false;
static thread_local const float* g_dh0_tanh_done = nullptr;
void vag_persist_dh0_tanh_request(bool on) { g_dh0_tanh_req = on; if (on) g_dh0_tanh_done = nullptr; }
bool vag_persist_dh0_tanh_done(const float* d_h0) { const bool d = d_h0 && g_dh0_tanh_done == d_h0; g_dh0_tanh_done = nullptr; return d; }
int vag_dec_bwd_persistent_launch(const float* pe, const float* encwp, const float* v, const float* wcatT, const float* whh1T,
                                  const float* h0, const float* h2_all, const float* h1, const float* g1, const float* g2,
                                  const float* qhp, const float* alpha, const float* d_h2_all, const float* dah, float* dgi2,
                                  float* dqgh, float* ds, float* dgi1, float* dgh1, float* d_h0, float* dal, unsigned* sync,
                                  int64_t B, int64_t Ts, int64_t Tt, int64_t H, hipStream_t s) {
    VAG_CHECK_ARG(pe && encwp && v && wcatT && whh1T && h0 && h2_all && h1 && g1 && g2 && qhp && alpha && d_h2_all && dah && dgi2 &&
                  dqgh && ds && dgi1 && dgh1 && d_h0 && dal && sync && vag_dec_bwd_persistent_ok(B, Ts, Tt, H));
    VAG_CHECK_ARG(aligned16(pe) && aligned16(encwp) && aligned16(wcatT) && aligned16(whh1T) && aligned16(h0) && aligned16(h2_all) &&
                  aligned16(h1) && aligned16(g1) && aligned16(g2) && aligned16(qhp) && aligned16(d_h2_all) && aligned16(dgi2) &&
                  aligned16(dqgh) && aligned16(dgi1) && aligned16(dgh1) && aligned16(d_h0));
    DecBArgs a;
    a.pe = pe; a.encwp = encwp; a.v = v; a.wcatT = wcatT; a.whh1T = whh1T; a.h0 = h0; a.h2_all = h2_all; a.h1 = h1; a.g1 = g1;
    a.g2 = g2; a.qhp = qhp; a.alpha = alpha; a.d_h2_all = d_h2_all; a.dah = dah; a.dgi2 = dgi2; a.dqgh = dqgh; a.ds = ds;
    a.dgi1 = dgi1; a.dgh1 = dgh1; a.d_h0 = d_h0; a.dal = dal;
    a.h0_tanh = g_dh0_tanh_req ? 1 : 0;
    if (g_dh0_tanh_req) g_dh0_tanh_done = d_h0;
    g_dh0_tanh_req = false;
#ifdef VAG_LAB
    a.dbg = reinterpret_cast<unsigned long long*>(vag_opt().dec_bwd_stamps);
#else
    a.dbg = nullptr;
#endif
    a.B = (int)B; a.Ts = (int)Ts; a.Tt = (int)Tt; a.H = (int)H; a.RT = (int)cdiv64(B, 16);
    const int nwords = (int)vag_dec_persistent_sync_words(B, Tt);
    a.cnt = sync; a.err = sync + (nwords - 64); a.spin = spin_limit(); a.guard = vag_persist_guard(); a.xcd_map = vag_opt().dec_xcd_map >> 4;
    const int nsc = (int)(Tt * B * Ts) * ACC_SHARDS;
    if (!g_prezeroed) {
        hipLaunchKernelGGL(zero2_u32_kernel, dim3((unsigned)cdiv64((int64_t)nwords + nsc, 256)), dim3(256), 0, s, sync, nwords,
                           reinterpret_cast<unsigned*>(dal), nsc);
        VAG_LAUNCH_CHECK();
    }
    int64_t lds = dec_bwd_persistent_lds_bytes(Ts, H);
    if (lds < 84 * 1024) lds = 84 * 1024;
    static AttrOnce once, once256;
    if (!set_max_lds_once(once, reinterpret_cast<const void*>(dec_bwd_persistent_kernel<64>))) return VAG_EINVAL;
    if (H == 256 && !set_max_lds_once(once256, reinterpret_cast<const void*>(dec_bwd_persistent_kernel<32>))) return VAG_EINVAL;
    const bool timed = ptimer_begin(3, s);
    const int tpp = dec_tiles_per_pass(H);
    for (a.rt0 = 0; a.rt0 < a.RT; a.rt0 += tpp) {                    // passes of row tiles, as the forward launch
        const unsigned tiles = (unsigned)std::min(tpp, a.RT - a.rt0);
        if (H == 512) hipLaunchKernelGGL(dec_bwd_persistent_kernel<64>, dim3(tiles * 64), dim3(512), (size_t)lds, s, a);
        else hipLaunchKernelGGL(dec_bwd_persistent_kernel<32>, dim3(tiles * 32), dim3(512), (size_t)lds, s, a);
    }
    if (timed) ptimer_end(3, s);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}
