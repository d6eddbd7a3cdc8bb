// Software-pipelined bf16x6 product kernel (round 4).
//
// What bounds the single-stage kernel of gemm.hip (profiles/r04_exp_gemm_pp.txt): not the matrix pipe (46 % busy), the LDS or
// memory, but the way vector work and matrix work share a SIMD.  Its two blocks per CU run in lockstep -- every wave splits and
// stores, then every wave issues MFMAs -- and a kernel that forces the two waves of a SIMD into opposite phases gains nothing
// either, because an MFMA-only wave and a VALU-only wave slow each other down by ~50 %.  The documented way to hide vector work
// under MFMAs (MI355X_MICROARCH.md, cycle constants: an MFMA holds vector issue for 8 of its 32 cycles; <= 5 single-issue
// instructions per gap hide) is inside ONE wave's instruction stream.  So here every wave runs one uniform stream per k-tile:
//     24 MFMAs of k-tile t (fragments out of LDS stage t & 1)
//   + the split and LDS store of its share of k-tile t + 1 (from registers, into stage (t + 1) & 1)
//   + the global loads of k-tile t + 2 (into the registers k-tile t left free)
// interleaved instruction by instruction (sched_group_barrier: 1 MFMA, 4 VALU, 1 LDS access, ...), ONE barrier per k-tile.
// 128 x 128 x 32 block tile, 8 waves (64 x 32 each), two LDS stages of 60 KB (one block per CU, up to 256 VGPRs per wave).
// Same arithmetic as gemm_split_body: the six products enter every accumulator in the same order, results are bitwise equal.
#define VAG_SPLIT_PLAIN_SUB 1
#include "gemm_shared.h"
#include <atomic>

namespace {

constexpr int SW_STAGE = 2 * 3 * SP_PLANE;            // bf16 elements per stage (A planes, B planes): 60 KB
constexpr int SW_LDS_BYTES = 2 * SW_STAGE * 2;        // 120 KB

// scheduling-group masks (LLVM AMDGPU): 0x002 VALU, 0x008 MFMA, 0x020 VMEM read, 0x100 DS read, 0x200 DS write
#define SW_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

__device__ __forceinline__ void sw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct SwCtx {
    int foff_a, foff_b, oa, obn;      // fragment offsets (k-contiguous operands) / first outer index (outer-contiguous ones)
};

template <bool AKC, bool BKC>
__device__ __forceinline__ void sw_mfma(const __bf16* S, const SwCtx& c, f32x16 (&acc)[2]) {
    const __bf16* As = S;
    const __bf16* Bs = S + 3 * SP_PLANE;
    bf16x8 af[2][2][3], bf[2][3];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            bf[ks][p] = BKC ? sp_frag(Bs + c.foff_b + p * SP_PLANE + ks * 16) : sp_frag_tr(Bs + p * SP_PLANE, c.obn, ks);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                af[ks][i][p] = AKC ? sp_frag(As + c.foff_a + p * SP_PLANE + i * 32 * SP_LD + ks * 16)
                                   : sp_frag_tr(As + p * SP_PLANE, c.oa + 32 * i, ks);
        }
    // per accumulator and k-step: a0b2, a1b1, a2b0, a0b1, a1b0, a0b0 (smallest terms first), exactly as gemm.hip's sp_compute
    constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i][PA[q]], bf[ks][PB[q]], acc[i], 0, 0, 0);
}

template <bool AKC>
__device__ __forceinline__ void sw_rowsum(const SpRegs& ra, float4& rs) {
    rs.x += ra.v[0] + ra.v[4]; rs.y += ra.v[1] + ra.v[5]; rs.z += ra.v[2] + ra.v[6]; rs.w += ra.v[3] + ra.v[7];
}

// LDS element offset (inside plane 0 of an operand's image) of this thread's item i: the same images sp_store fills
template <bool KC>
__device__ __forceinline__ int sw_item_off(int i) {
    const int idx = threadIdx.x + i * 512;
    return KC ? sp_row(idx >> 3) * SP_LD + ((idx & 7) << 2) : sp_oc_off(idx >> 5, (idx & 31) << 2);
}
__device__ __forceinline__ float sw_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float sw_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
#define SW_FENCE() __builtin_amdgcn_sched_barrier(0)

// One k-tile: 24 MFMAs of k-tile t out of stage `st`; in the issue shadow of MFMA g (g = 0..23) one sixth of the split of one
// 4-float item of k-tile t + 1 (four items per thread: two of A, two of B; 22 vector instructions and three 8-byte LDS stores per
// item) and, in the first gaps, the k-step-1 fragment reads.  sched_barrier(0) after every gap freezes that order: the machine
// scheduler may arrange a gap's handful of instructions, nothing crosses a gap boundary.  LOAD: k-tile t + 2 exists (its four
// 16-byte loads go out first), STORE: k-tile t + 1 exists.
template <bool AKC, bool BKC, bool LOAD, bool STORE>
__device__ __forceinline__ void sw_iter(__bf16* smem, int st, const SwCtx& c, f32x16 (&acc)[2], const SpRegs& ra_st, const SpRegs& rb_st,
                                        SpRegs& ra_ld, SpRegs& rb_ld, SpFast<AKC>& fa, SpFast<BKC>& fb, bool do_rs, float4& rs,
                                        const int (&ioff)[4]) {
    if (LOAD) {
        sp_fast_load<AKC>(fa, ra_ld);
        sp_fast_load<BKC>(fb, rb_ld);
    }
    const __bf16* As = smem + st * SW_STAGE;
    const __bf16* Bs = As + 3 * SP_PLANE;
    __bf16* W = smem + (st ^ 1) * SW_STAGE;
    bf16x8 af[2][2][3], bf[2][3];
    auto frag = [&](int ks, int n) {            // n = 0..8: b0 a00 a10 | b1 a01 a11 | b2 a02 a12  (plane-major)
        const int p = n / 3, w = n % 3;
        if (w == 0) bf[ks][p] = BKC ? sp_frag(Bs + c.foff_b + p * SP_PLANE + ks * 16) : sp_frag_tr(Bs + p * SP_PLANE, c.obn, ks);
        else af[ks][w - 1][p] = AKC ? sp_frag(As + c.foff_a + p * SP_PLANE + (w - 1) * 32 * SP_LD + ks * 16)
                                    : sp_frag_tr(As + p * SP_PLANE, c.oa + 32 * (w - 1), ks);
    };
#pragma unroll
    for (int n = 0; n < 9; ++n) frag(0, n);
    if (STORE && !AKC && do_rs) sw_rowsum<AKC>(ra_st, rs);
    SW_FENCE();
    constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};
    unsigned p1a = 0, p1b = 0, p2a = 0, p2b = 0, p3a = 0, p3b = 0;      // the current item's packed planes: a = floats 0,1; b = floats 2,3
    float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;                       // its residuals
#pragma unroll
    for (int g = 0; g < 24; ++g) {
        const int ks = g / 12, q = (g % 12) / 2, i = g & 1;
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i][PA[q]], bf[ks][PB[q]], acc[i], 0, 0, 0);
        if (g < 9) frag(1, g);
        if (STORE) {
            const int it = g / 6, ph = g % 6;
            const float* x = it < 2 ? &ra_st.v[4 * it] : &rb_st.v[4 * (it - 2)];
            __bf16* d = (it < 2 ? W : W + 3 * SP_PLANE) + ioff[it];
            if (ph == 0) {
                p1a = pack_bf16(x[0], x[1]); p1b = pack_bf16(x[2], x[3]);
                r0 = x[0] - sw_lo(p1a); r1 = x[1] - sw_hi(p1a);
            } else if (ph == 1) {
                r2 = x[2] - sw_lo(p1b); r3 = x[3] - sw_hi(p1b);
                *reinterpret_cast<uint2*>(d) = make_uint2(p1a, p1b);
            } else if (ph == 2) {
                p2a = pack_bf16(r0, r1); p2b = pack_bf16(r2, r3);
                r0 = r0 - sw_lo(p2a); r1 = r1 - sw_hi(p2a);
            } else if (ph == 3) {
                r2 = r2 - sw_lo(p2b); r3 = r3 - sw_hi(p2b);
                *reinterpret_cast<uint2*>(d + SP_PLANE) = make_uint2(p2a, p2b);
            } else if (ph == 4) {
                p3a = pack_bf16(r0, r1); p3b = pack_bf16(r2, r3);
            } else {
                *reinterpret_cast<uint2*>(d + 2 * SP_PLANE) = make_uint2(p3a, p3b);
            }
        }
        SW_FENCE();
    }
    sw_barrier();
    SW_FENCE();
}

template <bool AKC, bool BKC>
__device__ __forceinline__ void sw_body(const GemmArgs& a, __bf16* smem, int bx, int by, int bz) {
    const int m0 = by * 128, n0 = bx * 128;
    const int kbeg = bz * a.kchunk;
    const int kend = min(a.K, kbeg + a.kchunk);
    const int nfull = (kend - kbeg) / SP_BK;
    const bool partial = (kend - kbeg) % SP_BK != 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);      // the second-dispatched half loses every issue arbitration otherwise
    SwCtx c;
    c.foff_a = (wm * 64 + (lane & 31)) * SP_LD + 8 * (lane >> 5);
    c.foff_b = (wn * 32 + (lane & 31)) * SP_LD + 8 * (lane >> 5);
    c.oa = wm * 64; c.obn = wn * 32;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const bool do_rs = !AKC && a.rowsum != nullptr && bx == 0;
    float4 rs = make_float4(0.f, 0.f, 0.f, 0.f);
    SpRegs ra0, rb0, ra1, rb1;          // slot (j & 1) holds k-tile j between its load and its store
    const int ioff[4] = {sw_item_off<AKC>(0), sw_item_off<AKC>(1), sw_item_off<BKC>(0), sw_item_off<BKC>(1)};
    SpFast<AKC> fa;
    SpFast<BKC> fb;
    sp_fast_init<AKC>(fa, a.A, a.sa_o, a.sa_k, m0, kbeg, a.M);
    sp_fast_init<BKC>(fb, a.B, a.sb_o, a.sb_k, n0, kbeg, a.N);
    if (nfull > 0) {
        sp_fast_load<AKC>(fa, ra0);
        sp_fast_load<BKC>(fb, rb0);
        if (!AKC && do_rs) sw_rowsum<AKC>(ra0, rs);
        sp_store<AKC, 3>(smem, ra0);
        sp_store<BKC, 3>(smem + 3 * SP_PLANE, rb0);
        if (nfull > 1) {
            sp_fast_load<AKC>(fa, ra1);
            sp_fast_load<BKC>(fb, rb1);
        }
        sw_barrier();
        // invariant at the top of iteration t: stage t & 1 holds k-tile t, register slot (t + 1) & 1 holds k-tile t + 1
        int t = 0;
        for (; t + 3 < nfull; t += 2) {
            sw_iter<AKC, BKC, true, true>(smem, 0, c, acc, ra1, rb1, ra0, rb0, fa, fb, do_rs, rs, ioff);      // t even: store slot 1, load slot 0
            sw_iter<AKC, BKC, true, true>(smem, 1, c, acc, ra0, rb0, ra1, rb1, fa, fb, do_rs, rs, ioff);      // t + 1
        }
        for (; t < nfull; ++t) {             // the last one to three k-tiles: the same stream without the load / the store
            const bool hn = t + 1 < nfull, hn2 = t + 2 < nfull;
            if (t & 1) {
                if (hn2) sw_iter<AKC, BKC, true, true>(smem, 1, c, acc, ra0, rb0, ra1, rb1, fa, fb, do_rs, rs, ioff);
                else if (hn) sw_iter<AKC, BKC, false, true>(smem, 1, c, acc, ra0, rb0, ra1, rb1, fa, fb, do_rs, rs, ioff);
                else sw_iter<AKC, BKC, false, false>(smem, 1, c, acc, ra0, rb0, ra1, rb1, fa, fb, do_rs, rs, ioff);
            } else {
                if (hn2) sw_iter<AKC, BKC, true, true>(smem, 0, c, acc, ra1, rb1, ra0, rb0, fa, fb, do_rs, rs, ioff);
                else if (hn) sw_iter<AKC, BKC, false, true>(smem, 0, c, acc, ra1, rb1, ra0, rb0, fa, fb, do_rs, rs, ioff);
                else sw_iter<AKC, BKC, false, false>(smem, 0, c, acc, ra1, rb1, ra0, rb0, fa, fb, do_rs, rs, ioff);
            }
        }
    }
    if (partial) {                       // K % 32 != 0: the last, zero-filled k-tile through the bounds-checked loader
        const int k0 = kbeg + nfull * SP_BK;
        sp_load<AKC, true>(a.A, a.sa_o, a.sa_k, m0, k0, a.M, kend, ra0);
        sp_load<BKC, true>(a.B, a.sb_o, a.sb_k, n0, k0, a.N, kend, rb0);
        if (!AKC && do_rs) sw_rowsum<AKC>(ra0, rs);
        sp_store<AKC, 3>(smem, ra0);                                  // every wave is past its last read of both stages
        sp_store<BKC, 3>(smem + 3 * SP_PLANE, rb0);
        sw_barrier();
        sw_mfma<AKC, BKC>(smem, c, acc);
    }
    if (!AKC && do_rs) {                 // bias gradient from the A tiles, as gemm_split_body
        __syncthreads();
        float4* rs_s = reinterpret_cast<float4*>(smem);
        rs_s[threadIdx.x] = rs;
        __syncthreads();
        if (threadIdx.x < 32) {
            float4 t4 = rs_s[threadIdx.x];
#pragma unroll
            for (int q = 1; q < 16; ++q) {
                const float4 o = rs_s[threadIdx.x + 32 * q];
                t4.x += o.x; t4.y += o.y; t4.z += o.z; t4.w += o.w;
            }
            const int m = m0 + 4 * threadIdx.x;
            if (m + 0 < a.M) atomicAdd(a.rowsum + m + 0, t4.x);
            if (m + 1 < a.M) atomicAdd(a.rowsum + m + 1, t4.y);
            if (m + 2 < a.M) atomicAdd(a.rowsum + m + 2, t4.z);
            if (m + 3 < a.M) atomicAdd(a.rowsum + m + 3, t4.w);
        }
    }
    const bool atomic = a.splitk > 1;
    const bool first = bz == 0;
    const int col = n0 + wn * 32 + (lane & 31);
    if (col >= a.N) return;
    const float bv = (a.bias && first) ? a.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row0 = m0 + wm * 64 + i * 32 + 4 * (lane >> 5);
        if (a.c_half) gemm_epilogue16(acc[i], a.C, a.ldc, a.M - row0, a.alpha, a.beta, bv, a.act, atomic, 1, (int64_t)row0 * a.ldc + col);
        else gemm_epilogue16(acc[i], a.C + (int64_t)row0 * a.ldc + col, a.ldc, a.M - row0, a.alpha, a.beta, bv, a.act, atomic);
    }
}

template <bool AKC, bool BKC>
__global__ __launch_bounds__(512, 2) void gemm_swp_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) __bf16 sw_smem[];
    sw_body<AKC, BKC>(a, sw_smem, blockIdx.x, blockIdx.y, blockIdx.z);
}
template <bool AKC, bool BKC>
__global__ __launch_bounds__(512, 2) void gemm_swp_group_kernel(GemmGroupArgs G) {
    extern __shared__ __attribute__((aligned(16))) __bf16 sw_smem[];
    int p = 0;
    while (p + 1 < G.n && (int)blockIdx.x >= G.start[p + 1]) ++p;
    const GemmArgs& a = G.p[p];
    const int id = blockIdx.x - G.start[p];
    const int tn = (a.N + 127) / 128, tm = (a.M + 127) / 128;
    const int bx = id % tn, by = (id / tn) % tm, bz = id / (tn * tm);
    sw_body<AKC, BKC>(a, sw_smem, bx, by, bz);
}

// dynamic LDS above 64 KB needs the attribute once per kernel and device
template <typename K> bool sw_attr(K kernel) {
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (done.load(std::memory_order_acquire) & (1ull << dev)) return true;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SW_LDS_BYTES) != hipSuccess)
        return false;
    done.fetch_or(1ull << dev, std::memory_order_release);
    return true;
}

}  // namespace

#define SW_GO(KERNEL, ARG, GRID)                                                                      \
    { if (!sw_attr(KERNEL)) return VAG_EINVAL;                                                        \
      hipLaunchKernelGGL(KERNEL, GRID, dim3(512), SW_LDS_BYTES, s, ARG); }

int vag_gemm_swp_launch(const GemmArgs& g, bool akc, bool bkc, dim3 grid, hipStream_t s) {
    if (akc && bkc) SW_GO((gemm_swp_kernel<true, true>), g, grid)
    else if (akc && !bkc) SW_GO((gemm_swp_kernel<true, false>), g, grid)
    else if (!akc && bkc) SW_GO((gemm_swp_kernel<false, true>), g, grid)
    else SW_GO((gemm_swp_kernel<false, false>), g, grid)
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}
int vag_gemm_swp_group_launch(const GemmGroupArgs& G, bool akc, bool bkc, int total_blocks, hipStream_t s) {
    const dim3 grid((unsigned)total_blocks);
    if (akc && bkc) SW_GO((gemm_swp_group_kernel<true, true>), G, grid)
    else if (akc && !bkc) SW_GO((gemm_swp_group_kernel<true, false>), G, grid)
    else if (!akc && bkc) SW_GO((gemm_swp_group_kernel<false, true>), G, grid)
    else SW_GO((gemm_swp_group_kernel<false, false>), G, grid)
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}
