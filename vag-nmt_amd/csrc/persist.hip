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

namespace {

typedef unsigned __attribute__((address_space(1))) gu32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr unsigned SPIN_LIMIT = 1u << 19;       // polls before giving up (~a second; a healthy wait is a few hundred polls)

struct EncPArgs {
    const float* xp;            // (Ts, B, 6H): input projections [fwd r z n | rev r z n], biases included
    const float* W[2];          // (3H, H) recurrent weights per direction
    const float* bias[2];       // (3H) b_hh
    const int* lengths;         // (B)
    float* hst;                 // [2][Ts+1][B][H], step 0 zeros (caller)
    float* gates;               // [2][Ts][4][B][H]
    float* enc;                 // (B, Ts, 2H)
    unsigned* cnt;              // [2][RT][Ts], zero on entry
    unsigned* err;              // 1 word, set when a wait gave up
    int B, Ts, H, RT, CS;
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
template <> __device__ __forceinline__ void ld_rows_sc1<4>(const float* p, float4 (&a)[4], float4 (&b)[4]) {
    asm volatile("global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %8, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %8, off offset:128 sc1\n\tglobal_load_dwordx4 %3, %8, off offset:144 sc1\n\t"
                 "global_load_dwordx4 %4, %8, off offset:256 sc1\n\tglobal_load_dwordx4 %5, %8, off offset:272 sc1\n\t"
                 "global_load_dwordx4 %6, %8, off offset:384 sc1\n\tglobal_load_dwordx4 %7, %8, off offset:400 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a[0]), "=&v"(b[0]), "=&v"(a[1]), "=&v"(b[1]), "=&v"(a[2]), "=&v"(b[2]), "=&v"(a[3]), "=&v"(b[3])
                 : "v"(p) : "memory");
}
__device__ __forceinline__ void st_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float4 v) {
    const u32x4 u = {__builtin_bit_cast(unsigned, v.x), __builtin_bit_cast(unsigned, v.y), __builtin_bit_cast(unsigned, v.z),
                     __builtin_bit_cast(unsigned, v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(u, r, byte_off, 0, 16);
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
    const int cs = wg % a.CS, rt = (wg / a.CS) % a.RT, d = wg / (a.CS * a.RT);
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
    // ---- epilogue threads (wave 0): lane = (batch row fr, unit quad fg): units u0 + 4 fg .. + 3 of row m0 + fr
    const int em = m0 + fr, eu = u0 + 4 * fg;
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
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(hs, 0, (unsigned)((int64_t)(Ts + 1) * BH * 4), 0x00020000);
    gu32* cnt = (gu32*)(a.cnt + ((int64_t)d * a.RT + rt) * Ts);
    const int arow = min(m0 + fr, B - 1);                  // batch row of this lane's B-operand fragment (clamped past the edge)
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
                    if (++spins > SPIN_LIMIT) { __hip_atomic_store((gu32*)a.err, 1u, RLX_AGENT); dead = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
        }
        // ---- h_k rows of this row tile (all H columns; this wave: its K share), sc1 loads, split, six-product MFMAs
        f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        float4 h0[KS], h1[KS];
        ld_rows_sc1<KS>(hs + ((int64_t)k * B + arow) * H + kbase + 8 * fg, h0, h1);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bf16x8 hf[3];
            split8(h0[s], h1[s], hf);
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
                float4 sum = red[j * 64 + lane];
#pragma unroll
                for (int w = 1; w < 8; ++w) {
                    const float4 o = red[(w * 3 + j) * 64 + lane];
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
                float* sv = a.gates + ((int64_t)(d * Ts + k) * 4) * BH + o;
                *reinterpret_cast<float4*>(sv) = make_float4(rr[0], rr[1], rr[2], rr[3]);
                *reinterpret_cast<float4*>(sv + BH) = make_float4(zz[0], zz[1], zz[2], zz[3]);
                *reinterpret_cast<float4*>(sv + 2 * BH) = make_float4(nn[0], nn[1], nn[2], nn[3]);
                *reinterpret_cast<float4*>(sv + 3 * BH) = c[2];
                *reinterpret_cast<float4*>(a.enc + ((int64_t)em * Ts + t) * 2 * H + d * H + eu) = make_float4(o2[0], o2[1], o2[2], o2[3]);
            }
            // publish: this wave is the only one that stored; drain, then ONE lane signals for the workgroup
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(cnt + k, 1u, RLX_AGENT);
        }
        // (the barrier behind the next step's wait separates wave 0's LDS reads of this step from the next step's writes)
    }
}

__global__ __launch_bounds__(256) void zero_u32_kernel(unsigned* p, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0u;
}

}  // namespace

// Eligibility: hidden size a multiple of 256 up to 1024 (register budget of the weight planes), all workgroups resident at
// once (one per CU).
bool vag_enc_persistent_ok(int64_t B, int64_t Ts, int64_t H) {
    static int cus = -1;
    if (cus < 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        cus = n;
    }
    if (!(H == 256 || H == 512 || H == 1024) || B <= 0 || Ts <= 0) return false;
    const int64_t wgs = 2 * cdiv64(B, 16) * (H / 16);
    return wgs <= cus && (int64_t)(Ts + 1) * B * H * 4 < (1ll << 31);
}
int64_t vag_enc_persistent_sync_words(int64_t B, int64_t Ts) { return 2 * cdiv64(B, 16) * Ts + 64; }

int vag_enc_fwd_persistent_launch(const float* xp, const float* w_fw, const float* w_bw, const float* b_fw, const float* b_bw,
                                  const int* lengths, float* hst, float* gates, float* enc, unsigned* sync, int64_t B, int64_t Ts,
                                  int64_t H, hipStream_t s) {
    VAG_CHECK_ARG(xp && w_fw && w_bw && b_fw && b_bw && lengths && hst && gates && enc && sync && vag_enc_persistent_ok(B, Ts, H));
    VAG_CHECK_ARG(aligned16(xp) && aligned16(w_fw) && aligned16(w_bw) && aligned16(b_fw) && aligned16(b_bw) && aligned16(hst) &&
                  aligned16(gates) && aligned16(enc));
    EncPArgs a;
    a.xp = xp; a.W[0] = w_fw; a.W[1] = w_bw; a.bias[0] = b_fw; a.bias[1] = b_bw; a.lengths = lengths;
    a.hst = hst; a.gates = gates; a.enc = enc;
    a.B = (int)B; a.Ts = (int)Ts; a.H = (int)H; a.RT = (int)cdiv64(B, 16); a.CS = (int)(H / 16);
    const int nwords = (int)vag_enc_persistent_sync_words(B, Ts);
    a.cnt = sync; a.err = sync + (nwords - 64);
    hipLaunchKernelGGL(zero_u32_kernel, dim3((unsigned)cdiv64(nwords, 256)), dim3(256), 0, s, sync, nwords);
    VAG_LAUNCH_CHECK();
    const dim3 grid((unsigned)(2 * a.RT * a.CS));
    if (H == 256) hipLaunchKernelGGL(enc_fwd_persistent_kernel<1>, grid, dim3(512), 0, s, a);
    else if (H == 512) hipLaunchKernelGGL(enc_fwd_persistent_kernel<2>, grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL(enc_fwd_persistent_kernel<4>, grid, dim3(512), 0, s, a);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}
