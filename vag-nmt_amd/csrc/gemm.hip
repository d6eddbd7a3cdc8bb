// fp32 GEMM kernels for gfx950 on the f32-input MFMA (exact f32 fma chains, no reduced precision).
//
//  * gemm_tiled_kernel : LDS-staged BMxBNx16 block tiles, 4 waves (2x2), v_mfma_f32_32x32x2_f32,
//                        double-buffered LDS with register prefetch, optional split-K (atomic accumulate).
//                        Used for the once-per-batch contractions (input projections, attention keys,
//                        output head, every weight-gradient GEMM).
//  * skinny_kernel     : M <= ~128 rows (one decoder/encoder time step).  One 16-row m-tile per workgroup,
//                        K split across the waves, operands loaded straight from L2 into MFMA fragment
//                        layout (no LDS staging: each weight element is used once per workgroup),
//                        v_mfma_f32_16x16x4_f32, LDS only for the cross-wave reduction.  Epilogues: plain
//                        (bias/addend/tanh) and the fused GRU cell.
#ifndef VAG_CHEAT_B
#define VAG_CHEAT_B 0
#endif
#include "gemm_shared.h"
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <functional>
int vag_copy2d_launch(const float* in, int64_t ldi, float* out, int64_t ldo, int64_t rows, int64_t cols, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// tiled GEMM
// ------------------------------------------------------------------------------------------------
// Load a (BT outer) x (BK k) operand tile into registers.  KC: k is the contiguous dimension.  NTH threads.
template <int BT, bool KC, bool VEC, int NTH>
__device__ __forceinline__ void tile_load(const float* __restrict__ P, int64_t so, int64_t sk, int o0, int k0,
                                          int OUT, int KEND, float4 (&r)[BT * (BK / 4) / NTH]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < BT * (BK / 4) / NTH; ++i) {
        const int idx = tid + i * NTH;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KC) {
            const int o = o0 + idx / (BK / 4), k = k0 + ((idx % (BK / 4)) << 2);
            if (o < OUT) {
                const float* p = P + (int64_t)o * so + k;
                if (VEC && k + 3 < KEND) {
                    v = *reinterpret_cast<const float4*>(p);
                } else {
                    if (k + 0 < KEND) v.x = p[0];
                    if (k + 1 < KEND) v.y = p[1];
                    if (k + 2 < KEND) v.z = p[2];
                    if (k + 3 < KEND) v.w = p[3];
                }
            }
        } else {
            const int k = k0 + idx / (BT / 4), o = o0 + ((idx % (BT / 4)) << 2);
            if (k < KEND) {
                const float* p = P + (int64_t)k * sk + o;
                if (VEC && o + 3 < OUT) {
                    v = *reinterpret_cast<const float4*>(p);
                } else {
                    if (o + 0 < OUT) v.x = p[0];
                    if (o + 1 < OUT) v.y = p[1];
                    if (o + 2 < OUT) v.z = p[2];
                    if (o + 3 < OUT) v.w = p[3];
                }
            }
        }
        r[i] = v;
    }
}

// LDS image: S[k][o], row stride BT+4 floats (MFMA operand reads are 32 consecutive floats -> conflict free).
template <int BT, bool KC, int NTH>
__device__ __forceinline__ void tile_store(float* __restrict__ S, const float4 (&r)[BT * (BK / 4) / NTH]) {
    const int tid = threadIdx.x;
    constexpr int LD = BT + 4;
#pragma unroll
    for (int i = 0; i < BT * (BK / 4) / NTH; ++i) {
        const int idx = tid + i * NTH;
        if (KC) {
            const int o = idx / (BK / 4), k = (idx % (BK / 4)) << 2;
            S[(k + 0) * LD + o] = r[i].x;
            S[(k + 1) * LD + o] = r[i].y;
            S[(k + 2) * LD + o] = r[i].z;
            S[(k + 3) * LD + o] = r[i].w;
        } else {
            const int k = idx / (BT / 4), o = (idx % (BT / 4)) << 2;
            *reinterpret_cast<float4*>(&S[k * LD + o]) = r[i];
        }
    }
}

// NTH = 256: 4 waves as 2x2, each (BM/2)x(BN/2).  NTH = 512: 8 waves as 2x4, each (BM/2)x(BN/4): two waves per SIMD,
// so one workgroup alone on a CU (the usual case for this model's mid-size products) still overlaps LDS reads with MFMAs.
template <int BM, int BN, bool AKC, bool BKC, bool VEC, int NTH>
__device__ __forceinline__ void gemm_tiled_body(const GemmArgs& a, float* smem, int bx, int by, int bz) {
    constexpr int WN = NTH / 128;                 // waves along N
    constexpr int TM = BM / 64, TN = BN / (32 * WN);
    constexpr int LDA = BM + 4, LDB = BN + 4;
    float* As = smem;
    float* Bs = smem + 2 * BK * LDA;

    const int m0 = by * BM, n0 = bx * BN;
    const int kbeg = bz * a.kchunk;
    const int kend = min(a.K, kbeg + a.kchunk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[BM * (BK / 4) / NTH], rb[BN * (BK / 4) / NTH];
    tile_load<BM, AKC, VEC, NTH>(a.A, a.sa_o, a.sa_k, m0, kbeg, a.M, kend, ra);
    tile_load<BN, BKC, VEC, NTH>(a.B, a.sb_o, a.sb_k, n0, kbeg, a.N, kend, rb);
    tile_store<BM, AKC, NTH>(As, ra);
    tile_store<BN, BKC, NTH>(Bs, rb);
    __syncthreads();

    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = (k0 + BK) < kend;
        if (more) {
            tile_load<BM, AKC, VEC, NTH>(a.A, a.sa_o, a.sa_k, m0, k0 + BK, a.M, kend, ra);
            tile_load<BN, BKC, VEC, NTH>(a.B, a.sb_o, a.sb_k, n0, k0 + BK, a.N, kend, rb);
        }
        const float* Ac = As + cur * BK * LDA + wm * (BM / 2) + (lane & 31);
        const float* Bc = Bs + cur * BK * LDB + wn * (BN / WN) + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int kr = kk + (lane >> 5);
            float av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = Ac[kr * LDA + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = Bc[kr * LDB + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            tile_store<BM, AKC, NTH>(As + (cur ^ 1) * BK * LDA, ra);
            tile_store<BN, BKC, NTH>(Bs + (cur ^ 1) * BK * LDB, rb);
        }
        __syncthreads();
        cur ^= 1;
    }

    // epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    const bool atomic = a.splitk > 1;
    const bool first = bz == 0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (BN / WN) + j * 32 + (lane & 31);
            if (col >= a.N) continue;
            const float bv = (a.bias && first) ? a.bias[col] : 0.f;
            const int row0 = m0 + wm * (BM / 2) + i * 32 + 4 * (lane >> 5);
            float* cbase = a.C + (int64_t)row0 * a.ldc + col;
            if (a.c_half) gemm_epilogue16(acc[i][j], a.C, a.ldc, a.M - row0, a.alpha, a.beta, bv, a.act, false, 1, (int64_t)row0 * a.ldc + col);
            else gemm_epilogue16(acc[i][j], cbase, a.ldc, a.M - row0, a.alpha, a.beta, bv, a.act, atomic);
        }
}
template <int BM, int BN, bool AKC, bool BKC, bool VEC, int NTH>
__global__ __launch_bounds__(NTH) void gemm_tiled_kernel(GemmArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (BM + 4) + 2 * BK * (BN + 4)];
    gemm_tiled_body<BM, BN, AKC, BKC, VEC, NTH>(a, smem, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Several small weight-gradient products g_W += dY^T X of ONE launch (the leaf queue below): 64 x 64 tiles of the exact f32 kernel,
// both operands outer-contiguous, tiles of all tasks numbered through (tile0[k] = first tile of task k).  A task's optional
// rowsum (the bias gradient sum_r dY[r,:] that goes with it) is taken by the task's first column of tiles straight from memory.
constexpr int LEAF_MAX = 8;
struct LeafTasks {
    GemmArgs g[LEAF_MAX];
    int tile0[LEAF_MAX + 1];
    int tiles_x[LEAF_MAX];
    int n;
};
__global__ __launch_bounds__(256) void gemm_tiled_multi_kernel(LeafTasks T) {
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * 68 + 2 * BK * 68];
    int k = 0;
    while (k + 1 < T.n && (int)blockIdx.x >= T.tile0[k + 1]) ++k;
    const int t = blockIdx.x - T.tile0[k];
    const GemmArgs& a = T.g[k];
    const int bx = t % T.tiles_x[k], by = t / T.tiles_x[k];
    if (a.rowsum && bx == 0 && threadIdx.x < 64) {
        const int m = by * 64 + threadIdx.x;
        if (m < a.M) {
            float s0 = 0.f, s1 = 0.f;
            int r = 0;
            for (; r + 1 < a.K; r += 2) { s0 += a.A[(int64_t)r * a.sa_k + m]; s1 += a.A[(int64_t)(r + 1) * a.sa_k + m]; }
            if (r < a.K) s0 += a.A[(int64_t)r * a.sa_k + m];
            a.rowsum[m] += s0 + s1;
        }
    }
    gemm_tiled_body<64, 64, false, false, true, 256>(a, smem, bx, by, 0);
}

// ------------------------------------------------------------------------------------------------
// fp32 GEMM on the bf16 matrix pipes: 3-way operand split, 6 products ("bf16x6")
//
// gfx950's f32-input MFMA runs at 1/16 of the bf16 MFMA rate.  Every fp32 value is the exact sum of three bf16 numbers
// x = x1 + x2 + x3 (8 significand bits each: x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2); the residual is
// below 2^-24 |x|), so a*b = sum_{i+j<=4} a_i b_j up to 2^-24 relative: the six bf16 x bf16 products are exact in fp32
// and are accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  That is fp32-grade arithmetic (same error class as a
// reassociated fp32 sum) at 6/16 of the f32-MFMA instruction time.  The split is done once per element on its way from
// global memory into LDS (three bf16 planes per operand), so global traffic is unchanged.
// Block 128x128x32, 8 waves (2x4, 64x32 each), LDS single-staged (61 KB: two blocks per CU), register prefetch.
// ------------------------------------------------------------------------------------------------
// (operand loaders, the split + LDS store and the fragment reads / MFMA step live in gemm_shared.h)
template <bool AKC, bool BKC, bool VEC, int PL = 3, bool F16 = false, bool ABF = false>
__device__ __forceinline__ void gemm_split_body(const GemmArgs& a, __bf16* smem, int bx, int by, int bz) {
    __bf16* As = smem;
    __bf16* Bs = smem + PL * SP_PLANE;
    const int m0 = by * 128, n0 = bx * 128;
    const int kbeg = bz * a.kchunk;
    const int kend = min(a.K, kbeg + a.kchunk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 2, wn = wave & 3;

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // fragment addresses: lane (r = lane&31, h = lane>>5) holds 8 consecutive k (8h..8h+7) of row r
    const __bf16* Af = As + (wm * 64 + (lane & 31)) * SP_LD + 8 * (lane >> 5);
    const __bf16* Bf = Bs + (wn * 32 + (lane & 31)) * SP_LD + 8 * (lane >> 5);

    SpRegs ra, rb;
    // full k-tiles through the precomputed-offset loads (sp_fast_*), a last partial one (and unaligned operands) through sp_load
    constexpr bool FAST = VEC && !ABF;
    const int nt = (kend - kbeg + SP_BK - 1) / SP_BK;
    const int nfull = FAST ? (kend - kbeg) / SP_BK : 0;
    SpFast<AKC> fa;
    SpFast<BKC> fb;
    if (FAST) {
        sp_fast_init<AKC>(fa, a.A, a.sa_o, a.sa_k, m0, kbeg, a.M);
        sp_fast_init<BKC>(fb, a.B, a.sb_o, a.sb_k, n0, kbeg, a.N);
    }
    auto load_tile = [&](int t) {
        if (FAST && t < nfull) {
            sp_fast_load<AKC>(fa, ra);
            sp_fast_load<BKC>(fb, rb);
        } else {
            const int k0 = kbeg + t * SP_BK;
            if (ABF) sp_load_bf16<AKC>(reinterpret_cast<const unsigned short*>(a.A), a.sa_o, a.sa_k, m0, k0, a.M, kend, ra);
            else sp_load<AKC, VEC>(a.A, a.sa_o, a.sa_k, m0, k0, a.M, kend, ra);
            sp_load<BKC, VEC>(a.B, a.sb_o, a.sb_k, n0, k0, a.N, kend, rb);
        }
    };
    load_tile(0);
    // row sums of A for the first column tile's blocks (outer-contiguous A: this thread's two items of a k-tile are the same
    // four rows m at two k): the bias gradient of a weight-gradient product without a second pass over dY
    // (three-plane kernels only: at the one-plane kernels' 80-VGPR cap the extra state spills; the launcher sends those
    // products' row sums to a column-sum launch instead)
    const bool do_rs = !AKC && PL == 3 && a.rowsum != nullptr && bx == 0;
    float4 rs = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < nt; ++t) {
        if (!AKC && PL == 3 && do_rs) {
            rs.x += ra.v[0] + ra.v[4]; rs.y += ra.v[1] + ra.v[5]; rs.z += ra.v[2] + ra.v[6]; rs.w += ra.v[3] + ra.v[7];
        }
        sp_store<AKC, PL, F16, ABF, 8, (VAG_CHEAT_B > 1)>(As, ra);
        sp_store<BKC, PL, F16, false, 8, VAG_CHEAT_B != 0>(Bs, rb);
        __syncthreads();
        if (t + 1 < nt) load_tile(t + 1);
        sp_compute<PL, AKC, BKC, F16>(Af, Bf, As, Bs, wm * 64, wn * 32, acc);
        __syncthreads();
    }
    if (!AKC && PL == 3 && do_rs) {           // (uniform over the block) 16 threads hold partial sums of the same four rows: meet in LDS
        float4* rs_s = reinterpret_cast<float4*>(smem);
        rs_s[threadIdx.x] = rs;
        __syncthreads();
        if (threadIdx.x < 32) {
            float4 t = rs_s[threadIdx.x];
#pragma unroll
            for (int q = 1; q < 16; ++q) {
                const float4 o = rs_s[threadIdx.x + 32 * q];
                t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
            }
            const int m = m0 + 4 * threadIdx.x;
            if (m + 0 < a.M) atomicAdd(a.rowsum + m + 0, t.x);
            if (m + 1 < a.M) atomicAdd(a.rowsum + m + 1, t.y);
            if (m + 2 < a.M) atomicAdd(a.rowsum + m + 2, t.z);
            if (m + 3 < a.M) atomicAdd(a.rowsum + m + 3, t.w);
        }
    }

    // epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    bool atomic = a.splitk > 1;
    bool first = bz == 0;
    if (PL == 3 && !F16 && !ABF && a.slab != nullptr && a.nslices > 1) {      // (three-plane kernels only: the one-plane ones sit at an 80-register cap)
        // ---- split-K through slabs (GemmArgs::slab).  Atomics execute at the memory side at ~1.3 TB/s chip-wide (MI355X_MICROARCH.md,
        // global float atomics): a 2560 x 512 output in six slices is 31 MB of them, 24 us -- more than the slices' MFMA work -- and
        // they arrive in one burst when the launch is a single round of blocks.  Here a slice stores its 128 x 128 accumulators as they
        // lie in the registers (16 bytes per lane and store, write-through), drains, and one lane takes the tile's ticket; whoever
        // draws the last ticket loads ALL slabs of the tile in slice order (its own included: the sum then does not depend on who was
        // last -- bitwise reproducible, unlike the atomics) and finishes the tile alone.  Hand-off as persist.hip's (sc1 stores,
        // vmcnt(0), barrier, one agent-scope add whose RETURN value names the last arriver, sc1 loads): no fences, and no block
        // ever waits for another, so the scheme holds whatever part of the grid is resident.
        const int tile = by * ((a.N + 127) >> 7) + bx;
        float4* mine = reinterpret_cast<float4*>(a.slab) + ((int64_t)tile * a.nslices + bz) * 4096 + (wave * 8) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = {acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]};
                // (s_nop: a store of more than 64 bits must not be followed at once by a write of its data registers -- the compiler
                // knows that hazard for its own stores, not for inline asm)
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(mine + (i * 4 + q) * 64), "v"(v) : "memory");
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                          // (also: every wave is done with the LDS planes)
        int* flag = reinterpret_cast<int*>(smem);
        if (threadIdx.x == 0) {
            const unsigned t = __hip_atomic_fetch_add(a.ticket + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool last = t == (unsigned)(a.nslices - 1);
            if (last) __hip_atomic_store(a.ticket + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
            flag[0] = last ? 1 : 0;
        }
        __syncthreads();
        if (flag[0] == 0) return;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const float4* all = reinterpret_cast<const float4*>(a.slab) + (int64_t)tile * a.nslices * 4096 + (wave * 8) * 64 + lane;
        // (One slice per round trip.  Two in flight -- loads of slice z + 1 issued before slice z is waited for -- was tried: at the
        // kernels' 128-register cap the second buffer spills, scratch traffic then counts in vmcnt and the compiler saves load
        // targets before their data has arrived: wrong sums.  The tail this leaves is nslices x ~1.5 us on the last block of a tile.)
        for (int z = 0; z < a.nslices; ++z) {
            f32x4 v[8];
            const float4* p = all + (int64_t)z * 4096;
            asm volatile("global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %8, off offset:1024 sc1\n\t"
                         "global_load_dwordx4 %2, %8, off offset:2048 sc1\n\tglobal_load_dwordx4 %3, %8, off offset:3072 sc1\n\t"
                         "global_load_dwordx4 %4, %9, off sc1\n\tglobal_load_dwordx4 %5, %9, off offset:1024 sc1\n\t"
                         "global_load_dwordx4 %6, %9, off offset:2048 sc1\n\tglobal_load_dwordx4 %7, %9, off offset:3072 sc1\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                         : "v"(p), "v"(p + 256) : "memory");
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][4 * q + e] += v[i * 4 + q][e];
        }
        // an accumulating product still ADDS its one result atomically.  (A plain read-modify-write where no other product of the
        // launch targets the same C was measured: +11 us per step -- the block then waits for sixteen loads per lane before it can
        // store, while the atomics are fire-and-forget and overlap the next blocks' work.)
        atomic = a.beta != 0.f;
        first = true;                      // (the bias rides with the one epilogue)
    }
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

// (Round 3, measured and dropped: an XCD-contiguous, panel-ordered block -> tile mapping (T1 + 8-column panels).  Single
// products: no change on hot operands (4096^3 NT 705 -> 710 us); grouped launches: SLOWER in the step (weight-gradient group
// 150 -> 174 us, logits/keys group 67 -> 86 us) because an XCD then works through one product's blocks and the products'
// K differ by 10x -- the round-robin dealing of blocks over the XCDs is what balances a group.)
// Waves per SIMD asked of the compiler: 4 (two 8-wave blocks per CU: <= 128 VGPRs -- without the bound the forward group kernel
// took 134 and two single-product layouts 130, i.e. ONE block per CU), 6 for the one-plane kernels of the 2-byte mode (20 KB of
// LDS: three blocks per CU at <= 80 VGPRs, a few spills; configs[4] 33.3 -> 32.8 ms; 8 spills everything: 122 ms).
template <bool AKC, bool BKC, bool VEC, int PL = 3, bool F16 = false, bool ABF = false>
__global__ __launch_bounds__(512, PL == 1 ? 6 : 4) void gemm_split_kernel(GemmArgs a) {
    __shared__ __attribute__((aligned(16))) __bf16 smem[2 * PL * SP_PLANE];      // 60 / 40 / 20 KB
    gemm_split_body<AKC, BKC, VEC, PL, F16, ABF>(a, smem, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Grouped launch: the blocks of up to GROUP_MAX independent products of one operand layout (e.g. all "TN": both operands
// outer-contiguous, the weight gradients g_W += dY^T X of one operator) in ONE grid.  Each of these products alone is a few dozen 128x128 tiles
// with K = Tt*B: launched one by one they need split-K by 5-10 (atomics) to fill the chip and still pay a ramp and a
// partial last wave each; together they fill it with split-K 1-2.
template <bool AKC, bool BKC, int PL = 3, bool F16 = false>
__global__ __launch_bounds__(512, PL == 1 ? 6 : 4) void gemm_split_group_kernel(GemmGroupArgs G) {
    __shared__ __attribute__((aligned(16))) __bf16 smem[2 * PL * SP_PLANE];
    int p = 0;
    while (p + 1 < G.n && (int)blockIdx.x >= G.start[p + 1]) ++p;
    const GemmArgs& a = G.p[p];
    const int id = blockIdx.x - G.start[p];
    const int tn = (a.N + 127) / 128, tm = (a.M + 127) / 128;
    const int bx = id % tn, by = (id / tn) % tm, bz = id / (tn * tm);
    gemm_split_body<AKC, BKC, true, PL, F16>(a, smem, bx, by, bz);
}

// ---- round 4: 256 x 256 block tile for the ONE-plane products of the 2-byte storage mode (configs[4]) ----------------------
// With one plane per operand a 128 x 128 x 32 k-tile is 8 MFMAs per wave (256 matrix-pipe cycles per SIMD and block) behind 32 KB of
// fp32 operands: at the ~70 GB/s a CU draws from its L2 that is 0.46 us of ingest per 0.11 us of MFMA work -- the kernel above runs at
// 280-560 TF/s on configs[4]'s shapes (11-22 % of the bf16 peak; tools/exp_gemm_oneplane.py), bound by operand BYTES, as DESIGN
// section 9 already said of the large products in that mode.  A 256 x 256 tile does four times the MFMA work per k-tile on twice
// the bytes.  Eight waves (2 x 4, 128 x 64 outputs each: 8 accumulators), operands still fp32 in memory and rounded on their way
// into LDS (sp_load / sp_store, one plane; an operand's 256 rows are two of the 128-row images), two LDS stages of 40 KB, the loads of
// k-tile t + 2 and the store of t + 1 issued around the MFMAs of t, ONE barrier per k-tile.
constexpr int BIG_STAGE = 4 * SP_PLANE;                 // bf16 elements per stage: A halves 0,1 then B halves 0,1 (4 x 10 KB)
constexpr int BIG_LDS_BYTES = 2 * BIG_STAGE * 2;        // 81920
__device__ __forceinline__ void big_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool AKC, bool BKC, bool F16, bool ABF>
__device__ __forceinline__ void gemm_big_body(const GemmArgs& a, __bf16* smem, int bx, int by, int bz) {
    const int m0 = by * 256, n0 = bx * 256;
    const int kbeg = bz * a.kchunk;
    const int kend = min(a.K, kbeg + a.kchunk);
    const int nt = (kend - kbeg + SP_BK - 1) / SP_BK;
    const int nfull = (kend - kbeg) / SP_BK;               // full k-tiles: precomputed-offset loads (a bf16-stored A keeps its own loader)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 2, wn = wave & 3;               // rows wm * 128 .. +128 (= A half wm), columns wn * 64 .. +64 (B half wn >> 1)
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // fragment bases inside a stage (k-contiguous operands) / first outer index inside the half (outer-contiguous ones)
    const int foff_a = wm * SP_PLANE + (lane & 31) * SP_LD + 8 * (lane >> 5);
    const int foff_b = (2 + (wn >> 1)) * SP_PLANE + ((wn & 1) * 64 + (lane & 31)) * SP_LD + 8 * (lane >> 5);
    SpRegs ra[2], rb[2];                                   // halves of the k-tile in flight
    SpFast<AKC> fa[2];
    SpFastH<AKC> fah[2];
    SpFast<BKC> fb[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        // a half that lies entirely outside the operand is clamped onto its last valid row / group like any other out-of-range row
        if (!ABF) sp_fast_init<AKC>(fa[h], a.A, a.sa_o, a.sa_k, min(m0 + 128 * h, AKC ? a.M - 1 : ((a.M - 1) & ~3)), kbeg, a.M);
        else sp_fast_init_bf16<AKC>(fah[h], reinterpret_cast<const unsigned short*>(a.A), a.sa_o, a.sa_k,
                                    min(m0 + 128 * h, AKC ? a.M - 1 : ((a.M - 1) & ~3)), kbeg, a.M);
        sp_fast_init<BKC>(fb[h], a.B, a.sb_o, a.sb_k, min(n0 + 128 * h, BKC ? a.N - 1 : ((a.N - 1) & ~3)), kbeg, a.N);
    }
    auto load_tile = [&](int t) {
        const int k0 = kbeg + t * SP_BK;
        if (t < nfull) {            // ONE branch around all of a k-tile's loads: a select per load makes hipcc wait for each load in turn
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (ABF) sp_fast_load_bf16<AKC>(fah[h], ra[h]);
                else sp_fast_load<AKC>(fa[h], ra[h]);
                sp_fast_load<BKC>(fb[h], rb[h]);
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (ABF) sp_load_bf16<AKC>(reinterpret_cast<const unsigned short*>(a.A), a.sa_o, a.sa_k, m0 + 128 * h, k0, a.M, kend, ra[h]);
                else sp_load<AKC, true>(a.A, a.sa_o, a.sa_k, m0 + 128 * h, k0, a.M, kend, ra[h]);
                sp_load<BKC, true>(a.B, a.sb_o, a.sb_k, n0 + 128 * h, k0, a.N, kend, rb[h]);
            }
        }
    };
    // row sums of an outer-contiguous fp32 A (the bias gradient of a weight-gradient product g_W += dY^T X), taken from the tiles on
    // their way into LDS by the first column of blocks, as the 128 x 128 three-plane kernel does (round 6: the 2-byte mode's products
    // ran a separate column-sum pass over dY for them, 0.45 ms per step at configs[4])
    const bool do_rs = !AKC && !ABF && a.rowsum != nullptr && bx == 0;
    float4 rs[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    auto store_tile = [&](int stage) {
        __bf16* S = smem + stage * BIG_STAGE;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (!AKC && !ABF && do_rs) {
                rs[h].x += ra[h].v[0] + ra[h].v[4]; rs[h].y += ra[h].v[1] + ra[h].v[5];
                rs[h].z += ra[h].v[2] + ra[h].v[6]; rs[h].w += ra[h].v[3] + ra[h].v[7];
            }
            sp_store<AKC, 1, F16, ABF>(S + h * SP_PLANE, ra[h]);
            sp_store<BKC, 1, F16>(S + (2 + h) * SP_PLANE, rb[h]);
        }
    };
    auto compute = [&](int stage) {
        const __bf16* S = smem + stage * BIG_STAGE;
        const __bf16* Ah = S + wm * SP_PLANE;                       // this wave's A half (outer-contiguous image base)
        const __bf16* Bh = S + (2 + (wn >> 1)) * SP_PLANE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], bf[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
                bf[j] = BKC ? sp_frag(S + foff_b + j * 32 * SP_LD + ks * 16) : sp_frag_tr(Bh, (wn & 1) * 64 + 32 * j, ks);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[i] = AKC ? sp_frag(S + foff_a + i * 32 * SP_LD + ks * 16) : sp_frag_tr(Ah, 32 * i, ks);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (F16)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i]), __builtin_bit_cast(f16x8, bf[j]),
                                                                           acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
                }
        }
    };
    load_tile(0);
    store_tile(0);
    if (nt > 1) load_tile(1);
    big_barrier();
    for (int t = 0; t < nt; ++t) {
        const int st = t & 1;
        if (t + 1 < nt) store_tile(st ^ 1);           // k-tile t + 1 (in registers since the previous iteration)
        if (t + 2 < nt) load_tile(t + 2);
        compute(st);
        big_barrier();
    }
    if (!AKC && !ABF && do_rs) {      // (uniform over the block) 16 threads hold partial sums of the same four rows of a half: meet in LDS
        float4* rs_s = reinterpret_cast<float4*>(smem);
        __syncthreads();
        rs_s[threadIdx.x] = rs[0];
        rs_s[512 + threadIdx.x] = rs[1];
        __syncthreads();
        if (threadIdx.x < 64) {
            const int h = threadIdx.x >> 5, g = threadIdx.x & 31;
            float4 t = rs_s[h * 512 + g];
#pragma unroll
            for (int q = 1; q < 16; ++q) {
                const float4 o = rs_s[h * 512 + g + 32 * q];
                t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
            }
            const int m = m0 + 128 * h + 4 * g;
            if (m + 0 < a.M) atomicAdd(a.rowsum + m + 0, t.x);
            if (m + 1 < a.M) atomicAdd(a.rowsum + m + 1, t.y);
            if (m + 2 < a.M) atomicAdd(a.rowsum + m + 2, t.z);
            if (m + 3 < a.M) atomicAdd(a.rowsum + m + 3, t.w);
        }
    }
    const bool atomic = a.splitk > 1;
    const bool first = bz == 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + (lane & 31);
        if (col >= a.N) continue;
        const float bv = (a.bias && first) ? a.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row0 = m0 + wm * 128 + i * 32 + 4 * (lane >> 5);
            if (a.c_half) gemm_epilogue16(acc[i][j], a.C, a.ldc, a.M - row0, a.alpha, a.beta, bv, a.act, atomic, 1, (int64_t)row0 * a.ldc + col);
            else gemm_epilogue16(acc[i][j], a.C + (int64_t)row0 * a.ldc + col, a.ldc, a.M - row0, a.alpha, a.beta, bv, a.act, atomic);
        }
    }
}
template <bool AKC, bool BKC, bool F16, bool ABF>
__global__ __launch_bounds__(512, 2) void gemm_big_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) __bf16 big_smem[];
    gemm_big_body<AKC, BKC, F16, ABF>(a, big_smem, blockIdx.x, blockIdx.y, blockIdx.z);
}
template <bool AKC, bool BKC, bool F16>
__global__ __launch_bounds__(512, 2) void gemm_big_group_kernel(GemmGroupArgs G) {
    extern __shared__ __attribute__((aligned(16))) __bf16 big_smem[];
    int p = 0;
    while (p + 1 < G.n && (int)blockIdx.x >= G.start[p + 1]) ++p;
    const GemmArgs& a = G.p[p];
    const int id = blockIdx.x - G.start[p];
    const int tn = (a.N + 255) / 256, tm = (a.M + 255) / 256;
    const int bx = id % tn, by = (id / tn) % tm, bz = id / (tn * tm);
    gemm_big_body<AKC, BKC, F16, false>(a, big_smem, bx, by, bz);
}
template <typename K> static bool big_attr(K kernel) {          // dynamic LDS above 64 KB: the attribute once per kernel and device
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (done.load(std::memory_order_acquire) & (1ull << dev)) return true;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS_BYTES) != hipSuccess)
        return false;
    done.fetch_or(1ull << dev, std::memory_order_release);
    return true;
}

// (Round 4, measured and dropped -- profiles/r04_exp_gemm_waves4.txt, _pp.txt, _swp.txt, _big3.txt: a four-wave block with 64 x 64 wave tiles;
// 256 x 256 tiles for the three-plane products (+8-10 % at 4096^3, 0.5-0.8x at configs[1]'s M = 2560 shapes);
// a ping-pong of the two wave halves between MFMA and split intervals; one software-pipelined MFMA + split stream per wave.  All within
// +-10 % of this kernel: two waves per SIMD carrying 24 MFMAs + ~90 vector instructions + 30 LDS accesses per k-tile saturate the SIMD's
// issue port at ~75 % matrix-pipe occupancy whatever the order.)
// (A double-buffered variant -- 110 KB of LDS, one block per CU, split/store of tile t+1 issued between the k-halves of
// tile t -- measured 6-9 % slower than this single-stage kernel at two blocks per CU, and was dropped.  So was a
// wave-specialised one -- 4 producer waves splitting into a second LDS stage while 4 consumer waves run 64x64 MFMA
// tiles: 128 vs 138 TFLOP/s at 4096^3, up to 35 % slower at K = 256 -- and a 3-tile register prefetch (154 VGPRs, one
// block per CU).  Reference points: PMC on this kernel shows MFMA 31 %, LDS 39 %, VALU 26 % busy; a pure MFMA loop
// (tools/mfma_probe.hip) sustains 1.9-2.1 PFLOP/s bf16, i.e. 315-350 TFLOP/s fp32-equivalent at six products.)
// Planes per operand of the bf16 split on the calling thread: 3 (default: six products, fp32-grade) or 2 (three products:
// the 2-byte storage mode, set by the step driver for the duration of a call).
// 11: ONE fp16 plane (v_mfma_f32_32x32x16_f16; operands rounded to fp16 on their way into LDS -- the 2-byte mode's forward
// products in the step driver), 1: one bf16 plane (that mode's gradient products: fp16 would flush small gradients).
static thread_local int g_gemm_planes = 3;
void vag_gemm_set_planes(int planes) { g_gemm_planes = (planes == 2 || planes == 1 || planes == 11) ? planes : 3; }

static int gemm_split_dispatch(const GemmArgs& g, bool akc, bool bkc, bool vec, dim3 grid, hipStream_t s) {
    if (g.a_bf16) {         // the head's two vocabulary-sized gradient products over a bf16 d(logits) chunk
        if (g_gemm_planes != 1 || bkc || !vec) return VAG_EINVAL;
        if (akc) hipLaunchKernelGGL((gemm_split_kernel<true, false, true, 1, false, true>), grid, dim3(512), 0, s, g);
        else hipLaunchKernelGGL((gemm_split_kernel<false, false, true, 1, false, true>), grid, dim3(512), 0, s, g);
        VAG_LAUNCH_CHECK();
        return VAG_OK;
    }
#define VAG_SPLIT_CASE(AK, BKc, V)                                                                    \
    if (akc == AK && bkc == BKc && vec == V) {                                                        \
        if (g_gemm_planes == 2) hipLaunchKernelGGL((gemm_split_kernel<AK, BKc, V, 2>), grid, dim3(512), 0, s, g);   \
        else if (g_gemm_planes == 1) hipLaunchKernelGGL((gemm_split_kernel<AK, BKc, V, 1>), grid, dim3(512), 0, s, g);   \
        else if (g_gemm_planes == 11) hipLaunchKernelGGL((gemm_split_kernel<AK, BKc, V, 1, true>), grid, dim3(512), 0, s, g);   \
        else hipLaunchKernelGGL((gemm_split_kernel<AK, BKc, V, 3>), grid, dim3(512), 0, s, g);        \
        VAG_LAUNCH_CHECK();                                                                           \
        return VAG_OK;                                                                                \
    }
    VAG_SPLIT_CASE(true, true, true)
    VAG_SPLIT_CASE(true, false, true)
    VAG_SPLIT_CASE(false, true, true)
    VAG_SPLIT_CASE(false, false, true)
    VAG_SPLIT_CASE(true, true, false)
    VAG_SPLIT_CASE(true, false, false)
    VAG_SPLIT_CASE(false, true, false)
    VAG_SPLIT_CASE(false, false, false)
#undef VAG_SPLIT_CASE
    return VAG_EINVAL;
}

template <int BM, int BN, int NTH>
static int gemm_dispatch(const GemmArgs& g, bool akc, bool bkc, bool vec, dim3 grid, hipStream_t s) {
#define VAG_GEMM_CASE(AK, BKc, V)                                                                          \
    if (akc == AK && bkc == BKc && vec == V) {                                                             \
        hipLaunchKernelGGL((gemm_tiled_kernel<BM, BN, AK, BKc, V, NTH>), grid, dim3(NTH), 0, s, g);        \
        VAG_LAUNCH_CHECK();                                                                                \
        return VAG_OK;                                                                                     \
    }
    VAG_GEMM_CASE(true, true, true)
    VAG_GEMM_CASE(true, false, true)
    VAG_GEMM_CASE(false, true, true)
    VAG_GEMM_CASE(false, false, true)
    VAG_GEMM_CASE(true, true, false)
    VAG_GEMM_CASE(true, false, false)
    VAG_GEMM_CASE(false, true, false)
    VAG_GEMM_CASE(false, false, false)
#undef VAG_GEMM_CASE
    return VAG_EINVAL;
}

__global__ __launch_bounds__(256) void fill2d_kernel(float* __restrict__ C, int64_t ldc, int64_t rows, int64_t cols) {
    const int64_t total = rows * cols;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols;
        C[r * ldc + (i - r * cols)] = 0.f;
    }
}

// Tile / split-K choice.  One 128x128 block keeps a CU's four MFMA pipes busy for 128*k cycles, so a launch is only
// efficient with >= 2 blocks per CU in flight (latency hiding across co-resident blocks) and every CU busy: small-output,
// deep-K products (every weight gradient: K = Tt*B) are split along K and accumulated with fp32 atomics.
// ---- grouped launch queue (host side; the library is driven by one host thread per process) ----
// thread_local: forward runs on the caller's thread, backward on the autograd engine's; a bracket never spans threads
// One queue per operand layout (index akc*2 + bkc).  Brackets nest: an inner end() flushes everything queued so far (its
// caller is about to consume those results) and the outer bracket keeps collecting afterwards.
static thread_local int g_group_depth = 0;
static thread_local int g_qn[4] = {0, 0, 0, 0};
static thread_local GemmArgs g_q[4][GROUP_MAX];
static int vag_gemm_launch_now(const GemmArgs& q, hipStream_t stream);
void vag_colsum_queue_begin();
static bool gemm_take_prezeroed(const float* C);
int vag_colsum_queue_flush(hipStream_t stream);
void vag_colsum_queue_abort();
void vag_gemm_group_begin() {
    if (g_group_depth++ == 0) {
        for (int l = 0; l < 4; ++l) g_qn[l] = 0;
        vag_colsum_queue_begin();
    }
}
void vag_gemm_group_abort() {        // error path: drop the queues
    g_group_depth = 0;
    for (int l = 0; l < 4; ++l) g_qn[l] = 0;
    vag_colsum_queue_abort();
}
// Split-K and block order of one grouped launch.  The chip runs 512 of these blocks at a time (two per CU) and hands out
// blocks in index order to whichever slot frees first, i.e. list scheduling: with the longest blocks first the launch ends
// on short ones.  A common target slice length L (k-steps of SP_BK) is tried over the slice lengths the products can have;
// products are cut into round(K / L) slices (never shorter than 256; one that overwrites its output needs a fill launch first).  Each candidate is
// priced by simulating that schedule (block cost = slice length + a fixed prologue / epilogue share) plus the extra atomic
// traffic of the slices.  (The first version aimed at ~512 blocks with one common split: totals of 528 / 576 blocks -- the
// decoder / encoder weight-gradient groups -- ran a full second round for 16 / 64 blocks: 204 and 108 us.)
struct GroupPlanEntry { int n, tile; bool slabs; int m[GROUP_MAX], nn[GROUP_MAX], k[GROUP_MAX]; bool acc[GROUP_MAX], half[GROUP_MAX]; int split[GROUP_MAX], order[GROUP_MAX]; };
static void group_plan_compute(const GemmArgs* q, int n, int* split, int* order, int tile, bool slabs = false);
// plans are remembered per list of shapes (a training run repeats a handful of them; the simulation costs ~1 ms of host time)
static void group_plan(const GemmArgs* q, int n, int* split, int* order, int tile = 128, bool slabs = false) {
    constexpr int CACHE = 64;
    static thread_local GroupPlanEntry cache[CACHE];
    static thread_local int used = 0, next = 0;
    for (int e = 0; e < used; ++e) {
        const GroupPlanEntry& c = cache[e];
        bool same = c.n == n && c.tile == tile && c.slabs == slabs;
        for (int i = 0; same && i < n; ++i)
            same = c.m[i] == q[i].M && c.nn[i] == q[i].N && c.k[i] == q[i].K && c.acc[i] == (q[i].beta != 0.f) &&
                   c.half[i] == (q[i].c_half != 0);
        if (same) {
            for (int i = 0; i < n; ++i) { split[i] = c.split[i]; order[i] = c.order[i]; }
            return;
        }
    }
    group_plan_compute(q, n, split, order, tile, slabs);
    GroupPlanEntry& c = cache[next];
    next = (next + 1) % CACHE;
    if (used < CACHE) ++used;
    c.n = n; c.tile = tile; c.slabs = slabs;
    for (int i = 0; i < n; ++i) {
        c.m[i] = q[i].M; c.nn[i] = q[i].N; c.k[i] = q[i].K; c.acc[i] = q[i].beta != 0.f; c.half[i] = q[i].c_half != 0;
        c.split[i] = split[i]; c.order[i] = order[i];
    }
}
static void group_plan_compute(const GemmArgs* q, int n, int* split, int* order, int tile, bool slabs) {
    constexpr int MAXSLOTS = 512;
    const int SLOTS = tile == 256 ? 256 : 512;  // 256 x 256 one-plane blocks (gemm_big_kernel): one per CU
    const double C0 = tile == 256 ? 8.0 : 4.0;  // k-steps a block spends outside its main loop
    const double US_PER_KSTEP = tile == 256 ? 1.0 : 2.5;        // one 128x128x32 step of a block sharing its CU / one 256x256x32 one-plane step
    int cand[GROUP_MAX * 10 + 1], nc = 0;
    for (int i = 0; i < n; ++i) {
        const int ks = (int)cdiv64(q[i].K, SP_BK);
        const int smax = q[i].c_half ? 1 : std::max(1, std::min(10, q[i].K / 256));      // fp16 outputs are stored whole
        for (int j = 1; j <= smax; ++j) {
            const int L = (ks + j - 1) / j;
            bool seen = false;
            for (int c = 0; c < nc; ++c) seen = seen || cand[c] == L;
            if (!seen) cand[nc++] = L;
        }
    }
    double best = 1e30;
    double load[MAXSLOTS];
    for (int c = 0; c < nc; ++c) {
        const int L = cand[c];
        int sp[GROUP_MAX], len[GROUP_MAX], ord[GROUP_MAX];
        double atomic_us = 0.0;
        for (int i = 0; i < n; ++i) {
            const int ks = (int)cdiv64(q[i].K, SP_BK);
            int s_i = (ks + L / 2) / L;
            const int smax = q[i].c_half ? 1 : std::max(1, q[i].K / 256);
            s_i = std::max(1, std::min(s_i, smax));
            const double out_bytes = (double)q[i].M * (double)q[i].N * 4.0;
            if (slabs && s_i > 1) atomic_us += (double)s_i * out_bytes / 4.0e6 + 1.5 * s_i / 4.0 + out_bytes / (q[i].beta != 0.f ? 3.0e6 : 4.0e6);
            else if (q[i].beta != 0.f) atomic_us += (double)s_i * out_bytes / 3.0e6;
            else if (s_i > 1) atomic_us += (double)s_i * out_bytes / 3.0e6 + 4.0 + out_bytes / 4.0e6;   // + a fill launch first
            sp[i] = s_i;
            len[i] = (ks + s_i - 1) / s_i;
            ord[i] = i;
        }
        std::stable_sort(ord, ord + n, [&](int a, int b) { return len[a] > len[b]; });
        // list scheduling of equal-cost batches onto the slots: a min-heap of slot loads
        for (int k = 0; k < SLOTS; ++k) load[k] = 0.0;
        std::make_heap(load, load + SLOTS, std::greater<double>());
        double makespan = 0.0;
        for (int j = 0; j < n; ++j) {
            const int i = ord[j];
            const int64_t blocks = cdiv64(q[i].M, tile) * cdiv64(q[i].N, tile) * sp[i];
            const double cost = (double)len[i] + C0;
            for (int64_t b = 0; b < blocks; ++b) {
                std::pop_heap(load, load + SLOTS, std::greater<double>());
                load[SLOTS - 1] += cost;
                makespan = std::max(makespan, load[SLOTS - 1]);
                std::push_heap(load, load + SLOTS, std::greater<double>());
            }
        }
        const double us = makespan * US_PER_KSTEP + atomic_us;
        if (us < best) {
            best = us;
            for (int i = 0; i < n; ++i) { split[i] = sp[i]; order[i] = ord[i]; }
        }
    }
}
// Host-only view of the plan (include/vag_nmt.h: vag_gemm_group_plan): tests and tuning scripts
int vag_gemm_group_plan_host(int n, const int64_t* M, const int64_t* N, const int64_t* K, const int* accumulate, int* split,
                             int* order) {
    VAG_CHECK_ARG(n >= 1 && n <= GROUP_MAX && M && N && K && accumulate && split && order);
    GemmArgs q[GROUP_MAX] = {};
    for (int i = 0; i < n; ++i) {
        VAG_CHECK_ARG(M[i] > 0 && N[i] > 0 && K[i] > 0 && M[i] < (1ll << 30) && N[i] < (1ll << 30) && K[i] < (1ll << 30));
        q[i].M = (int)M[i]; q[i].N = (int)N[i]; q[i].K = (int)K[i]; q[i].beta = accumulate[i] ? 1.f : 0.f; q[i].c_half = 0;
    }
    group_plan_compute(q, n, split, order, 128);
    return VAG_OK;
}
// Scratch of the slab form of split-K (GemmArgs::slab): caller-owned, handed over per thread for the duration of a call (the step
// driver's workspace: step.hip).  A launch takes what its products need from the start of it -- launches of one stream follow each
// other, so the next one may reuse the same floats; a launch that goes to ANOTHER stream gets none (atomics, as before).
struct GemmScratch { float* slab = nullptr; int64_t floats = 0; unsigned* tickets = nullptr; int64_t ntickets = 0; };
static thread_local GemmScratch g_gemm_scratch;
void vag_gemm_set_scratch(float* slab, int64_t floats, unsigned* tickets, int64_t ntickets) {
    g_gemm_scratch.slab = slab; g_gemm_scratch.floats = slab ? floats : 0;
    g_gemm_scratch.tickets = tickets; g_gemm_scratch.ntickets = tickets ? ntickets : 0;
}
// slabs and tickets for a product of `tiles` output tiles in `slices` k-slices, from running offsets; false: does not fit (or off)
static bool gemm_take_slabs(GemmArgs& a, int64_t tiles, int slices, int64_t& used_f, int64_t& used_t) {
    a.slab = nullptr; a.ticket = nullptr; a.nslices = slices;
    const GemmScratch& sc = g_gemm_scratch;
    if (slices <= 1 || !sc.slab || !sc.tickets || vag_opt().gemm_slabs == 0) return false;
    const int64_t need = tiles * slices * 16384;
    if (used_f + need > sc.floats || used_t + tiles > sc.ntickets) return false;
    a.slab = sc.slab + used_f; a.ticket = sc.tickets + used_t;
    used_f += need; used_t += tiles;
    return true;
}
static int gemm_group_flush_layout(int lay, hipStream_t stream, bool own_stream = true) {
    const int n = g_qn[lay];
    g_qn[lay] = 0;
    if (n == 0) return VAG_OK;
    if (n == 1) {
        const int depth = g_group_depth;       // launch directly, not back into the queue
        g_group_depth = 0;
        const int rc = vag_gemm_launch_now(g_q[lay][0], stream);
        g_group_depth = depth;
        return rc;
    }
    const GemmArgs* q = g_q[lay];
    GemmGroupArgs G;
    G.n = n;
    int split[GROUP_MAX], order[GROUP_MAX];
    // the 2-byte storage mode's one-plane products: 256 x 256 tiles when every product of the group is large enough for them
    bool big = (g_gemm_planes == 1 || g_gemm_planes == 11) && vag_opt().gemm_big != 0;
    for (int j = 0; j < n && big; ++j) big = q[j].M >= 192 && q[j].N >= 192 && !q[j].a_bf16;
    const int T = big ? 256 : 128;
    if (!big && g_gemm_planes != 3)             // row sums ride on the three-plane 128 x 128 kernel and on the 256 x 256 one only
        for (int j = 0; j < n; ++j)
            if (g_q[lay][j].rowsum) {
                GemmArgs& r = g_q[lay][j];
                VAG_TRY(vag_colsum_launch(r.A, r.K, r.M, r.sa_k, r.rowsum, stream));
                r.rowsum = nullptr;
            }
    const bool slabs_on = !big && own_stream && g_gemm_planes == 3 && g_gemm_scratch.slab != nullptr && vag_opt().gemm_slabs != 0;
    group_plan(q, n, split, order, T, slabs_on);
    int total = 0;
    int64_t slab_used = 0, ticket_used = 0;
    for (int j = 0; j < n; ++j) {
        GemmArgs& a = G.p[j];
        a = q[order[j]];
        int s_i = split[order[j]];
        const int kchunk = (int)(cdiv64(cdiv64(a.K, s_i), SP_BK) * SP_BK);
        s_i = (int)cdiv64(a.K, kchunk);
        a.kchunk = kchunk;
        // accumulating products always add atomically here (two of them may target the same gradient buffer);
        // splitk > 1 is what selects the atomic epilogue, the block count below uses the real number of k-slices
        a.splitk = a.beta != 0.f ? (s_i > 2 ? s_i : 2) : s_i;
        const bool slabs = !big && own_stream && (g_gemm_planes == 3) &&
                           gemm_take_slabs(a, cdiv64(a.M, T) * cdiv64(a.N, T), s_i, slab_used, ticket_used);
        if (!slabs) { a.slab = nullptr; a.ticket = nullptr; a.nslices = s_i; }
        if (slabs && a.beta == 0.f) (void)gemm_take_prezeroed(a.C);      // (a prezeroed mark on this output is spent either way)
        if (a.beta == 0.f && s_i > 1 && !slabs && !gemm_take_prezeroed(a.C)) {          // sliced overwrite: the slices add into a zeroed output
            int64_t nb = cdiv64((int64_t)a.M * a.N, 256 * 8);
            if (nb > 2048) nb = 2048;
            hipLaunchKernelGGL(fill2d_kernel, dim3((unsigned)nb), dim3(256), 0, stream, a.C, a.ldc, (int64_t)a.M, (int64_t)a.N);
            VAG_LAUNCH_CHECK();
        }
        G.start[j] = total;
        total += (int)(cdiv64(a.M, T) * cdiv64(a.N, T)) * s_i;
    }
    G.start[n] = total;
    const bool akc = (lay & 2) != 0, bkc = (lay & 1) != 0;
#define VAG_GROUP_GO(PLN, F)                                                                                                  \
    if (!akc && !bkc)                                                                                                         \
        hipLaunchKernelGGL((gemm_split_group_kernel<false, false, PLN, F>), dim3((unsigned)total), dim3(512), 0, stream, G);  \
    else if (akc && !bkc)                                                                                                     \
        hipLaunchKernelGGL((gemm_split_group_kernel<true, false, PLN, F>), dim3((unsigned)total), dim3(512), 0, stream, G);   \
    else if (akc && bkc)                                                                                                      \
        hipLaunchKernelGGL((gemm_split_group_kernel<true, true, PLN, F>), dim3((unsigned)total), dim3(512), 0, stream, G);    \
    else                                                                                                                      \
        hipLaunchKernelGGL((gemm_split_group_kernel<false, true, PLN, F>), dim3((unsigned)total), dim3(512), 0, stream, G);
    if (big) {
#define VAG_BIG_GO(AK, BKc, F)                                                                                                \
        { if (!big_attr(gemm_big_group_kernel<AK, BKc, F>)) return VAG_EINVAL;                                                \
          hipLaunchKernelGGL((gemm_big_group_kernel<AK, BKc, F>), dim3((unsigned)total), dim3(512), BIG_LDS_BYTES, stream, G); }
        const bool f16 = g_gemm_planes == 11;
        if (!akc && !bkc) { if (f16) VAG_BIG_GO(false, false, true) else VAG_BIG_GO(false, false, false) }
        else if (akc && !bkc) { if (f16) VAG_BIG_GO(true, false, true) else VAG_BIG_GO(true, false, false) }
        else if (akc && bkc) { if (f16) VAG_BIG_GO(true, true, true) else VAG_BIG_GO(true, true, false) }
        else { if (f16) VAG_BIG_GO(false, true, true) else VAG_BIG_GO(false, true, false) }
#undef VAG_BIG_GO
    }
    else if (g_gemm_planes == 11) { VAG_GROUP_GO(1, true) }
    else if (g_gemm_planes == 1) { VAG_GROUP_GO(1, false) }
    else if (g_gemm_planes == 2) { VAG_GROUP_GO(2, false) }
    else { VAG_GROUP_GO(3, false) }
#undef VAG_GROUP_GO
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}
// A side stream for the weight-gradient layout (TN: both operands outer-contiguous, gemm_tn_acc: g_W += dY^T X with its bias sums)
// of the group flushes of the calling thread, until taken back: the step driver sends the decoder's weight gradients there
// (step.hip, step_fork bit 2).  Only leaves have that layout -- nothing later in a step but the optimiser reads what they write --
// and it is flushed first, so it depends on nothing else in its flush; the event is recorded on the flushing stream right before,
// i.e. behind every launch that produced the operands.
static thread_local hipStream_t g_group_leaf_stream = nullptr;
static thread_local hipEvent_t g_group_leaf_event = nullptr;
static thread_local bool g_group_leaf_used = false;
void vag_gemm_group_leaf_stream(hipStream_t s, hipEvent_t ev) { g_group_leaf_stream = s; g_group_leaf_event = ev; if (s) g_group_leaf_used = false; }
bool vag_gemm_group_leaf_used() { return g_group_leaf_used; }
int vag_gemm_group_end(hipStream_t stream) {
    if (g_group_depth <= 0) return VAG_OK;
    int rc = vag_colsum_queue_flush(stream);
    for (int lay = 0; lay < 4 && rc == VAG_OK; ++lay) {
        hipStream_t to = stream;
        if (lay == 0 && g_group_leaf_stream && g_qn[0] > 0 && g_group_leaf_stream != stream) {
            if (hipEventRecord(g_group_leaf_event, stream) == hipSuccess && hipStreamWaitEvent(g_group_leaf_stream, g_group_leaf_event, 0) == hipSuccess) {
                to = g_group_leaf_stream;
                g_group_leaf_used = true;
            } else {
                (void)hipGetLastError();
            }
        }
        rc = gemm_group_flush_layout(lay, to, to == stream);
    }
    if (--g_group_depth == 0 || rc != VAG_OK) {
        if (rc != VAG_OK) vag_gemm_group_abort();
        else vag_colsum_queue_abort();           // bracket closed: later column sums launch at once
    }
    return rc;
}

// Leaf queue (step driver, backward of the VSE branch and of the initial state: VSE_Imagine_Enc.py:110-152, V11.py:118): the
// weight-gradient products of those operators are rank-B updates (K = B <= 128 rows) that nothing later in the step reads -- five
// launches of the 64 x 64 kernel and three column-sum launches, ~5 us each, strung between the kernels of a 28-launch chain.  Between
// vag_leaf_begin and vag_leaf_flush such products (and the column sums that go with them) are held back and go out as ONE launch
// at the flush; their operands must stay untouched until then (the step driver's do: api.hip).
static thread_local bool g_leaf_on = false;
static thread_local LeafTasks g_leaf;
// Outputs a caller has already zeroed (vag_train_step's prologue launch): a sliced (split-K) overwriting product into one of them
// skips its own fill launch.  An entry is used once.  Calling thread.
static thread_local const float* g_gemm_prezeroed[4] = {nullptr, nullptr, nullptr, nullptr};
void vag_gemm_prezeroed_set(int slot, const float* p) { if (slot >= 0 && slot < 4) g_gemm_prezeroed[slot] = p; }
static bool gemm_take_prezeroed(const float* C) {
    for (auto& q : g_gemm_prezeroed)
        if (q && q == C) { q = nullptr; return true; }
    return false;
}
void vag_leaf_begin() { g_leaf_on = vag_opt().leaf_queue != 0; g_leaf.n = 0; g_leaf.tile0[0] = 0; }
void vag_leaf_abort() { g_leaf_on = false; g_leaf.n = 0; }
bool vag_leaf_attach_rowsum(const float* X, int64_t rows, int64_t N, int64_t ld, float* out) {
    if (!g_leaf_on) return false;
    for (int k = 0; k < g_leaf.n; ++k) {
        GemmArgs& q = g_leaf.g[k];
        if (q.A == X && q.K == (int)rows && q.M == (int)N && q.sa_k == ld && !q.rowsum) { q.rowsum = out; return true; }
    }
    return false;
}
int vag_leaf_flush(hipStream_t stream) {
    const bool was = g_leaf_on;
    g_leaf_on = false;
    const int n = g_leaf.n;
    if (!was || n == 0) { g_leaf.n = 0; return VAG_OK; }
    hipLaunchKernelGGL(gemm_tiled_multi_kernel, dim3((unsigned)g_leaf.tile0[n]), dim3(256), 0, stream, g_leaf);   // (n travels in the struct)
    g_leaf.n = 0;
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

int vag_gemm_launch(int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t sam, int64_t sak,
                    const float* B, int64_t sbk, int64_t sbn, float beta, float* C, int64_t ldc,
                    const float* bias, int act, hipStream_t stream, int c_half, float* rowsum, int a_bf16) {
    const bool opt_f32mfma = vag_opt().gemm_f32mfma != 0;      // vag_set_option("gemm_f32mfma"): the bf16x6 bound test flips it
    const bool opt_nogroup = vag_opt().gemm_nogroup != 0;
    if (g_leaf_on && g_leaf.n < LEAF_MAX && M > 0 && N > 0 && K > 0 && K <= 128 && sam == 1 && sbn == 1 && alpha == 1.f && beta == 1.f &&
        !bias && act == VAG_ACT_NONE && !c_half && !a_bf16 && A && B && C && aligned16(A) && aligned16(B) && sak % 4 == 0 && sbk % 4 == 0 &&
        M < (1 << 20) && N < (1 << 20)) {
        const int k = g_leaf.n++;
        GemmArgs& g = g_leaf.g[k];
        g.A = A; g.B = B; g.C = C; g.bias = nullptr;
        g.sa_o = 1; g.sa_k = sak; g.sb_o = 1; g.sb_k = sbk; g.ldc = ldc;
        g.M = (int)M; g.N = (int)N; g.K = (int)K; g.kchunk = (int)(cdiv64(K, BK) * BK);
        g.alpha = 1.f; g.beta = 1.f; g.act = VAG_ACT_NONE; g.splitk = 1; g.c_half = 0; g.a_bf16 = 0; g.rowsum = rowsum;
        g_leaf.tiles_x[k] = (int)cdiv64(N, 64);
        g_leaf.tile0[k + 1] = g_leaf.tile0[k] + (int)(cdiv64(N, 64) * cdiv64(M, 64));
        return VAG_OK;
    }
    VAG_CHECK_ARG(M >= 0 && N >= 0 && K >= 0 && A && B && C);
    if (M == 0 || N == 0) return VAG_OK;
    VAG_CHECK_ARG(M < (1ll << 30) && N < (1ll << 30) && K < (1ll << 30));
    VAG_CHECK_ARG(sam == 1 || sak == 1);
    VAG_CHECK_ARG(sbk == 1 || sbn == 1);
    GemmArgs g;
    g.slab = nullptr; g.ticket = nullptr; g.nslices = 1;
    g.A = A; g.B = B; g.C = C; g.bias = bias;
    const bool akc = (sak == 1);        // A: k contiguous
    const bool bkc = (sbk == 1);        // B: k contiguous
    g.sa_o = sam; g.sa_k = sak; g.sb_o = sbn; g.sb_k = sbk;
    g.ldc = ldc; g.M = (int)M; g.N = (int)N; g.K = (int)K;
    g.alpha = alpha; g.beta = beta; g.act = act; g.c_half = c_half; g.rowsum = rowsum; g.a_bf16 = a_bf16;
    VAG_CHECK_ARG(!a_bf16 || (g_gemm_planes == 1 && g_group_depth == 0 && !rowsum));     // one-plane bf16 kernel, launched at once
    VAG_CHECK_ARG(!c_half || beta == 0.f);      // fp16 output: plain stores only (no split-K, no accumulation)
    VAG_CHECK_ARG(!rowsum || sam == 1);         // row sums ride on outer-contiguous A tiles only
    // ... and on the three-plane kernels, or (round 6) on the 256 x 256 one-plane kernel of the 2-byte mode's gradient products;
    // otherwise a column-sum pass over A
    const bool rs_big_ok = (g_gemm_planes == 1 || g_gemm_planes == 11) && !akc && !a_bf16 && vag_opt().gemm_big != 0 && M >= 192 && N >= 192 &&
                           !opt_f32mfma;
    if (rowsum && g_gemm_planes != 3 && !rs_big_ok) {
        VAG_TRY(vag_colsum_launch(A, K, M, sak, rowsum, stream));
        g.rowsum = rowsum = nullptr;
    }
    const int64_t lda = akc ? sam : sak, ldb = bkc ? sbn : sbk;
    const bool vec = aligned16(A) && aligned16(B) && (lda % 4 == 0) && (ldb % 4 == 0);
    const int lay = (akc ? 2 : 0) + (bkc ? 1 : 0);
    if (g_group_depth > 0 && g_qn[lay] < GROUP_MAX && vec && alpha == 1.f && (beta == 0.f || beta == 1.f) &&
        act == VAG_ACT_NONE && M > 64 && N > 64 && K >= 256 && !opt_f32mfma && !opt_nogroup) {
        g.kchunk = (int)K; g.splitk = 1;
        g_q[lay][g_qn[lay]++] = g;
        return VAG_OK;
    }
    // cost model (microseconds) over tile in {64,128} x split-K: MFMA time of the busiest CU + output traffic
    // (split-K partial sums as fp32 atomics ~3 TB/s chip-wide, plain stores ~4 TB/s) -- constants fitted to measured launches.
    const bool can_split = (act == VAG_ACT_NONE) && (beta == 0.f || beta == 1.f) && !c_half;
    double best = 1e30;
    int64_t T = 64, splitk = 1;
    // bytes/us of split-K partial sums landing as fp32 atomics (fitted: tools/exp_gemm_sweep.py; 1e6 was too pessimistic)
    const double atomic_rate = 3.0e6;
    for (int64_t t = 64; t <= 128; t *= 2) {
        if (t == 128 && (M <= 64 || N <= 64)) continue;
        if (t == 64 && a_bf16 && M > 64 && N > 64) continue;       // a bf16-stored operand: the 128 x 128 one-plane kernel only
        const double eff = (t == 128) ? (opt_f32mfma ? 0.62 : 0.64) : 0.42;   // fraction of the f32-MFMA peak
        const int64_t base = cdiv64(M, t) * cdiv64(N, t);
        for (int64_t sp = 1; sp <= 64; sp += (sp < 16 ? 1 : sp < 32 ? 4 : 8)) {      // few tiles, long K (a row chunk of d(tmid)): up to 64
            if (sp > 1 && (!can_split || K / sp < 128)) break;
            const int64_t kper = cdiv64(cdiv64(K, sp), BK) * BK;
            const int64_t blocks = base * sp;
            // a CU holds two 128x128 blocks, and a pair advances ~1.33x faster than two blocks one after the other:
            // full waves of 512 blocks count 1.5 "single-block rounds", a remainder of <= 256 blocks counts 1
            double rounds;
            if (t == 128 && !opt_f32mfma) {
                const int64_t full = blocks / 512, rem = blocks % 512;
                rounds = 1.5 * (double)full + (rem == 0 ? 0.0 : (rem <= 256 ? 1.0 : 1.5));
            } else {
                rounds = (double)cdiv64(blocks, 256);
            }
            const double t_mfma = rounds * (double)kper * (double)(t * t) * 2.0 / (256.0 * eff) / 2400.0;
            const double bytes = (double)M * (double)N * 4.0;
            // (slab form of split-K, when the caller's scratch is at hand: the slices' slabs as plain stores, then the last block
            // of a tile reads them back one round trip per slice, plus the one result)
            const bool slabs_on = g_gemm_scratch.slab != nullptr && vag_opt().gemm_slabs != 0 && t == 128 && g_gemm_planes == 3 && !opt_f32mfma;
            const double t_out = sp > 1 ? (slabs_on ? sp * bytes / 4.0e6 + 1.5 * sp + bytes / (beta != 0.f ? atomic_rate : 4.0e6)
                                                    : sp * bytes / atomic_rate + (beta == 0.f ? bytes / 4.0e6 + 2.0 : 0.0))
                                        : bytes * (beta != 0.f ? 2.0 : 1.0) / 4.0e6;
            const double cost = t_mfma + t_out;
            if (cost < best) { best = cost; T = t; splitk = sp; }
        }
    }
    {          // tuning hook: vag_set_option("gemm_force_tile" / "gemm_force_splitk")
        const int ft = vag_opt().gemm_force_tile, fs = vag_opt().gemm_force_splitk;
        if ((ft == 64 || ft == 128) && fs >= 1 && (fs == 1 || can_split)) { T = ft; splitk = fs; }
    }
#ifdef VAG_LAB
    if (vag_opt().gemm_debug)
        fprintf(stderr, "[vag_gemm] M=%lld N=%lld K=%lld akc=%d bkc=%d beta=%g -> T=%lld splitk=%lld model=%.1f us\n",
                (long long)M, (long long)N, (long long)K, (int)akc, (int)bkc, (double)beta, (long long)T, (long long)splitk, best);
#endif
    // the 2-byte storage mode's one-plane products on 256 x 256 tiles (gemm_big_kernel): split-K so that tiles x slices fill the
    // 256 CUs about once; a bf16-stored A (d(logits) chunks) included
    // (only where the cost model above chose the 128 x 128 one-plane kernel: products it sends to the exact f32 64 x 64 kernel stay there)
    if (T == 128 && (g_gemm_planes == 1 || g_gemm_planes == 11) && vec && !opt_f32mfma && vag_opt().gemm_big != 0 && M >= 192 && N >= 192 &&
        (!a_bf16 || !bkc) && vag_opt().gemm_force_tile == 0) {
        const int64_t tiles = cdiv64(M, 256) * cdiv64(N, 256);
        int64_t sp = 1;
        if (can_split) while (tiles * sp < 208 && K / (sp + 1) >= 512 && sp < 32) ++sp;
        int kc = (int)(cdiv64(cdiv64(K, sp), BK) * BK);
        sp = cdiv64(K, kc);
        g.splitk = (int)sp; g.kchunk = kc;
        if (sp > 1 && beta == 0.f && !gemm_take_prezeroed(C)) {
            int64_t nb = cdiv64(M * N, 256 * 8);
            if (nb > 2048) nb = 2048;
            hipLaunchKernelGGL(fill2d_kernel, dim3((unsigned)nb), dim3(256), 0, stream, C, ldc, M, N);
            VAG_LAUNCH_CHECK();
        }
        const dim3 grid((unsigned)cdiv64(N, 256), (unsigned)cdiv64(M, 256), (unsigned)sp);
        const bool f16 = g_gemm_planes == 11;
#define VAG_BIG1(AK, BKc, F, AB)                                                                                              \
        { if (!big_attr(gemm_big_kernel<AK, BKc, F, AB>)) return VAG_EINVAL;                                                  \
          hipLaunchKernelGGL((gemm_big_kernel<AK, BKc, F, AB>), grid, dim3(512), BIG_LDS_BYTES, stream, g); }
        if (a_bf16) { if (akc) VAG_BIG1(true, false, false, true) else VAG_BIG1(false, false, false, true) }
        else if (!akc && !bkc) { if (f16) VAG_BIG1(false, false, true, false) else VAG_BIG1(false, false, false, false) }
        else if (akc && !bkc) { if (f16) VAG_BIG1(true, false, true, false) else VAG_BIG1(true, false, false, false) }
        else if (akc && bkc) { if (f16) VAG_BIG1(true, true, true, false) else VAG_BIG1(true, true, false, false) }
        else { if (f16) VAG_BIG1(false, true, true, false) else VAG_BIG1(false, true, false, false) }
#undef VAG_BIG1
        VAG_LAUNCH_CHECK();
        return VAG_OK;
    }
    const bool big = (T == 128);
    VAG_CHECK_ARG(!a_bf16 || (big && !opt_f32mfma));       // a bf16-stored operand exists for the one-plane split kernel only
    if (g.rowsum && (!big || opt_f32mfma || g_gemm_planes != 3)) {          // the other kernels do not carry row sums: a column-sum pass over A instead
        VAG_TRY(vag_colsum_launch(A, K, M, sak, rowsum, stream));
        g.rowsum = nullptr;
    }
    int kchunk = (int)(cdiv64(cdiv64(K, splitk), BK) * BK);
    splitk = cdiv64(K, kchunk);
    g.splitk = (int)splitk; g.kchunk = kchunk;
    int64_t slab_used = 0, ticket_used = 0;
    const bool slabs = big && !opt_f32mfma && g_gemm_planes == 3 && !a_bf16 &&
                       gemm_take_slabs(g, cdiv64(M, T) * cdiv64(N, T), (int)splitk, slab_used, ticket_used);
    if (!slabs) { g.slab = nullptr; g.ticket = nullptr; g.nslices = (int)splitk; }
    if (slabs && beta == 0.f) (void)gemm_take_prezeroed(C);
    if (splitk > 1 && beta == 0.f && !slabs && !gemm_take_prezeroed(C)) {
        int64_t nb = cdiv64(M * N, 256 * 8);
        if (nb > 2048) nb = 2048;
        hipLaunchKernelGGL(fill2d_kernel, dim3((unsigned)nb), dim3(256), 0, stream, C, ldc, M, N);
        VAG_LAUNCH_CHECK();
    }
    dim3 grid((unsigned)cdiv64(N, T), (unsigned)cdiv64(M, T), (unsigned)splitk);
    if (big) {
        if (!opt_f32mfma) return gemm_split_dispatch(g, akc, bkc, vec, grid, stream);
        // f32-input MFMA path (v_mfma_f32_32x32x2_f32), 8 waves (2 per SIMD)
        return gemm_dispatch<128, 128, 512>(g, akc, bkc, vec, grid, stream);
    }
    return gemm_dispatch<64, 64, 256>(g, akc, bkc, vec, grid, stream);
}

// One product launched at once (not queued into an open group bracket) with `planes` bf16 planes per operand.
int vag_gemm_launch_planes(int planes, int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t sam, int64_t sak,
                           const float* B, int64_t sbk, int64_t sbn, float beta, float* C, int64_t ldc, hipStream_t stream,
                           int a_bf16) {
    const int depth = g_group_depth, pl = g_gemm_planes;
    g_group_depth = 0;
    g_gemm_planes = planes;
    const int rc = vag_gemm_launch(M, N, K, alpha, A, sam, sak, B, sbk, sbn, beta, C, ldc, nullptr, 0, stream, 0, nullptr, a_bf16);
    g_group_depth = depth;
    g_gemm_planes = pl;
    return rc;
}
static int vag_gemm_launch_now(const GemmArgs& q, hipStream_t stream) {
    return vag_gemm_launch(q.M, q.N, q.K, q.alpha, q.A, q.sa_o, q.sa_k, q.B, q.sb_k, q.sb_o, q.beta, q.C, q.ldc, q.bias, q.act,
                           stream, q.c_half, q.rowsum);
}

// ------------------------------------------------------------------------------------------------
// skinny GEMM  out[m,n] = sum_k A[m,k] W[n,k]   (both k-contiguous), fused epilogues
//
// These launches are bound by bytes fetched per CU (each CU pulls ~33 GB/s out of the Infinity Cache; L2 is dropped
// at every kernel boundary), so the workgroup tile is chosen to (a) put one workgroup on every one of the 256 CUs
// and (b) minimise (A rows + W rows) * K per workgroup: see pick_plain_tile().
// ------------------------------------------------------------------------------------------------
struct SkinnyArgs {
    const float* A; const float* W; int64_t lda, ldw;
    int M, N, K;
    const float* bias; const float* addend; int64_t ldadd; float* out; int64_t ldo; int act;
    int64_t bsA = 0, bsW = 0, bsO = 0;     // batched launches (blockIdx.z): element strides of A, W and out/addend
    // A rows taken through an index (A = embedding table, row m = A[row_idx[m]]); the column-tile-0 workgroups also write
    // the gathered rows out (the embedded inputs are needed again by the head and by the backward pass)
    const int64_t* row_idx = nullptr; float* gather_out = nullptr; int64_t ld_gather = 0;
    float a_scale = 1.f;                   // fp16-weight products only: power of two applied to A before its fp16 rounding (gradients)
    float* out2 = nullptr; int64_t ldo2 = 0; float scale2 = 0.f; int acc2 = 0;      // skinny_bt_kernel: out2 (+)= scale2 * result as well
};

// Load pattern.  The MFMA wants lane l to hold row l&15, k-group l>>4, but a wave request whose lane QUADS each touch
// a different row is processed at about half the rate of one whose quads read 64 contiguous bytes (measured,
// tools/skinny_probe.hip: 11.5 -> 6.5 us for 64x512x2560).  So lane l LOADS row skinny_ldrow(l), 16-byte segment
// l&3 (every quad = 64 contiguous bytes of one row), and the float4s are then moved to the MFMA's lanes with a fixed
// permutation through ds_bpermute (no LDS storage; it swaps lane bits {0,1} with {4,5}).
__device__ __forceinline__ int skinny_ldrow(int lane) { return 4 * ((lane >> 2) & 3) + (lane >> 4); }
__device__ __forceinline__ int skinny_ldseg(int lane) { return lane & 3; }
__device__ __forceinline__ float4 skinny_xpose(float4 v, int src4) {
    float4 o;
    o.x = __int_as_float(__builtin_amdgcn_ds_bpermute(src4, __float_as_int(v.x)));
    o.y = __int_as_float(__builtin_amdgcn_ds_bpermute(src4, __float_as_int(v.y)));
    o.z = __int_as_float(__builtin_amdgcn_ds_bpermute(src4, __float_as_int(v.z)));
    o.w = __int_as_float(__builtin_amdgcn_ds_bpermute(src4, __float_as_int(v.w)));
    return o;
}

// MT 16-row m-tiles x NT 16-col n-tiles per workgroup; K split over the WAVES waves.  ap/wp are this lane's LOAD
// pointers (row skinny_ldrow(lane) of each tile, already offset by 4*skinny_ldseg(lane) floats).  Partial sums of all
// waves end up in red[wave][tile][lane][4] (C/D map of the 16x16 MFMA: col = lane&15, row = 4*(lane>>4)+i).
template <int WAVES, int MT, int NT, int U = 4>
__device__ __forceinline__ void skinny_mma(const float* const (&ap)[MT], const float* const (&wp)[NT], int K, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = skinny_ldseg(lane);
    // MFMA lane (k-group G = lane>>4, row 4a+j = lane&15) takes the float4 loaded by lane 16j + 4a + G
    const int src4 = 4 * (16 * (lane & 3) + (lane & 12) + (lane >> 4));
    const int kper = ((K + WAVES - 1) / WAVES + 15) & ~15;
    const int kbeg = wave * kper;
    const int kend = min(K, kbeg + kper);
    f32x4 acc[MT][NT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            acc[i][j][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[i][j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    // U chunks (16 of K each) in flight per wave.  Chosen by the caller so that a wave's whole K share is requested at
    // once where it fits the registers: a second trip of the loop is a second dependent round of memory latency.
    for (int c0 = kbeg; c0 < kend; c0 += 16 * U) {
        float4 av[MT][U], wv[NT][U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = c0 + 16 * u + 4 * g;
            const bool ok = k < kend;
#pragma unroll
            for (int i = 0; i < MT; ++i)
                av[i][u] = ok ? *reinterpret_cast<const float4*>(ap[i] + c0 + 16 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < NT; ++j)      // a null row pointer = this lane's row of the tile is unused: zeros, no request
                wv[j][u] = (ok && wp[j]) ? *reinterpret_cast<const float4*>(wp[j] + c0 + 16 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int i = 0; i < MT; ++i) av[i][u] = skinny_xpose(av[i][u], src4);
#pragma unroll
            for (int j = 0; j < NT; ++j) wv[j][u] = skinny_xpose(wv[j][u], src4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    // lane group g supplies k = 4g+e to MFMA e; A and W use the same k order, so the sum is exact.
                    acc[i][j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][u].x, wv[j][u].x, acc[i][j][0], 0, 0, 0);
                    acc[i][j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][u].y, wv[j][u].y, acc[i][j][1], 0, 0, 0);
                    acc[i][j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][u].z, wv[j][u].z, acc[i][j][0], 0, 0, 0);
                    acc[i][j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][u].w, wv[j][u].w, acc[i][j][1], 0, 0, 0);
                }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            f32x4 s = acc[i][j][0] + acc[i][j][1];
            *reinterpret_cast<f32x4*>(&red[((wave * (MT * NT) + i * NT + j) * 64 + lane) * 4]) = s;
        }
    __syncthreads();
}

// The same product with the W operand stored as fp16 (the 2-byte storage mode: weights the recurrences re-read at every
// time step).  Chunks are 32 of K: lane (row r, segment g) loads 8 consecutive halves of W (one 16-byte request: quads
// still read 64 contiguous bytes) and the matching 8 floats of A as two float4 (k = 8g..8g+3, 8g+4..8g+7); both go through
// the same lane permutation, which leaves MFMA lane (row, k-group G) with k = 8G..8G+7 of its row -- exactly the operand
// layout of v_mfma_f32_16x16x32_f16.  Round 3: ONE fp16 MFMA per chunk (A rounded to fp16 in registers, fp32 accumulation)
// instead of eight v_mfma_f32_16x16x4_f32 on fp32 A x converted W: at configs[4] (M = 256, K = 1024..3072) the per-step
// kernels were bound by the f32 matrix pipe (1.6 GFLOP in 15-26 us = 60-110 TF/s of its 157), and the fp16 pipe runs the
// same chunk in 1/16 of the cycles.  a_scale (a power of two): gradients are small -- backward kernels pass 2^12 so that
// entries down to 1.5e-8 stay normal fp16 numbers and up to 16 stay finite; the sum is scaled back in fp32.  ap/wp are
// already offset by 8*skinny_ldseg(lane) elements.  K % 8 == 0.
typedef _Float16 sk_f16x8 __attribute__((ext_vector_type(8)));
template <int WAVES, int MT, int NT, int U = 2>
__device__ __forceinline__ void skinny_mma_h16(const float* const (&ap)[MT], const vag_half* const (&wp)[NT], int K, float* red,
                                               float a_scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = skinny_ldseg(lane);
    const int src4 = 4 * (16 * (lane & 3) + (lane & 12) + (lane >> 4));
    const int kper = ((K + WAVES - 1) / WAVES + 31) & ~31;
    const int kbeg = wave * kper;
    const int kend = min(K, kbeg + kper);
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c0 = kbeg; c0 < kend; c0 += 32 * U) {
        float4 a0[MT][U], a1[MT][U];
        uint4 wq[NT][U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = c0 + 32 * u + 8 * g;
            const bool ok = k < kend;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                a0[i][u] = ok ? *reinterpret_cast<const float4*>(ap[i] + c0 + 32 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
                a1[i][u] = ok ? *reinterpret_cast<const float4*>(ap[i] + c0 + 32 * u + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j)
                wq[j][u] = (ok && wp[j]) ? *reinterpret_cast<const uint4*>(wp[j] + c0 + 32 * u) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            sk_f16x8 af[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const float4 x = skinny_xpose(a0[i][u], src4), y = skinny_xpose(a1[i][u], src4);
                const u32x4 q = {pack_f16(x.x * a_scale, x.y * a_scale), pack_f16(x.z * a_scale, x.w * a_scale),
                                 pack_f16(y.x * a_scale, y.y * a_scale), pack_f16(y.z * a_scale, y.w * a_scale)};
                af[i] = __builtin_bit_cast(sk_f16x8, q);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                uint4 q = wq[j][u];
                q.x = (unsigned)__builtin_amdgcn_ds_bpermute(src4, (int)q.x);
                q.y = (unsigned)__builtin_amdgcn_ds_bpermute(src4, (int)q.y);
                q.z = (unsigned)__builtin_amdgcn_ds_bpermute(src4, (int)q.z);
                q.w = (unsigned)__builtin_amdgcn_ds_bpermute(src4, (int)q.w);
                const u32x4 t = {q.x, q.y, q.z, q.w};
                const sk_f16x8 wf = __builtin_bit_cast(sk_f16x8, t);
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], wf, acc[i][j], 0, 0, 0);
            }
        }
    }
    const float inv = 1.f / a_scale;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const f32x4 s = acc[i][j] * inv;
            *reinterpret_cast<f32x4*>(&red[((wave * (MT * NT) + i * NT + j) * 64 + lane) * 4]) = s;
        }
    __syncthreads();
}
// Dispatch on the W storage type.  koff = this lane's element offset inside a chunk (4g for fp32 W, 8g for fp16 W).
template <bool WH> __device__ __forceinline__ int skinny_koff(int g) { return WH ? 8 * g : 4 * g; }
template <int WAVES, int MT, int NT, int U, bool WH>
__device__ __forceinline__ void skinny_mma_any(const float* const (&ap)[MT], const float* const (&wp)[NT], int K, float* red,
                                               float a_scale = 1.f) {
    if constexpr (WH) {
        const vag_half* wh[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) wh[j] = reinterpret_cast<const vag_half*>(wp[j]);
        skinny_mma_h16<WAVES, MT, NT, (U + 1) / 2>(ap, wh, K, red, a_scale);
    } else {
        skinny_mma<WAVES, MT, NT, U>(ap, wp, K, red);
    }
}
// address of element (row, col) of a W matrix stored as fp32 or fp16 (returned as const float* either way: the half
// version is re-cast inside skinny_mma_any)
template <bool WH> __device__ __forceinline__ const float* skinny_wptr(const float* W, int64_t row, int64_t ldw, int col) {
    if (WH) return reinterpret_cast<const float*>(reinterpret_cast<const vag_half*>(W) + row * ldw + col);
    return W + row * ldw + col;
}

// sum over the waves of tile `tile`, for this lane's 4 accumulator rows
template <int WAVES, int TILES>
__device__ __forceinline__ f32x4 skinny_sum(const float* red, int tile, int lane) {
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < WAVES; ++w) s += *reinterpret_cast<const f32x4*>(&red[((w * TILES + tile) * 64 + lane) * 4]);
    return s;
}
// sum over the waves of ONE element (row mrow in [0,16), col in [0,16)) of tile `tile`
template <int WAVES, int TILES>
__device__ __forceinline__ float skinny_sum1(const float* red, int tile, int mrow, int col) {
    const int lane = (mrow >> 2) * 16 + col, i = mrow & 3;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) s += red[((w * TILES + tile) * 64 + lane) * 4 + i];
    return s;
}

// One 16x16 output tile per workgroup; the 256 outputs are finished by the first 256 threads (bias / addend requested
// before the product).
template <int WAVES, int U = 4, bool WH = false>
__device__ __forceinline__ void skinny_plain_body(const SkinnyArgs& a, float* red, int bx, int by) {
    const int lane = threadIdx.x & 63;
    const int r = skinny_ldrow(lane), g = skinny_ldseg(lane);
    const int m0 = by * 16, nb = bx * 16;
    const int erow = (threadIdx.x >> 4) & 15, ecol = threadIdx.x & 15;
    const int em = m0 + erow, ej = nb + ecol;
    const bool eok = threadIdx.x < 256 && em < a.M && ej < a.N;
    float pre = 0.f;
    if (eok) {
        if (a.bias) pre = a.bias[ej];
        if (a.addend) pre += a.addend[(int64_t)em * a.ldadd + ej];
    }
    const float* ap[1];
    const float* wp[1];
    const int64_t arow = a.row_idx ? a.row_idx[min(m0 + r, a.M - 1)] : (int64_t)min(m0 + r, a.M - 1);
    ap[0] = a.A + arow * a.lda + skinny_koff<WH>(g);
    wp[0] = skinny_wptr<WH>(a.W, min(nb + r, a.N - 1), a.ldw, skinny_koff<WH>(g));
    if (a.gather_out && bx == 0) {
        const int k4 = a.K >> 2;
        for (int i = threadIdx.x; i < 16 * k4; i += WAVES * 64) {
            const int row = i / k4, c4 = i - row * k4;
            if (m0 + row < a.M)
                reinterpret_cast<float4*>(a.gather_out + (int64_t)(m0 + row) * a.ld_gather)[c4] =
                    reinterpret_cast<const float4*>(a.A + a.row_idx[m0 + row] * a.lda)[c4];
        }
    }
    skinny_mma_any<WAVES, 1, 1, U, WH>(ap, wp, a.K, red, a.a_scale);
    if (!eok) return;
    float v = skinny_sum1<WAVES, 1>(red, 0, erow, ecol) + pre;
    if (a.act == VAG_ACT_TANH) v = vag_tanh(v);
    a.out[(int64_t)em * a.ldo + ej] = v;
}
// MT x NT 16 x 16 tiles per workgroup (round 4, for M >= 128 rows: configs[4]'s per-step products have 256).  A 16 x 16 tile
// requests 16 rows of each operand; at M = 256, N = 3072, K = 1024 that is 295 MB per product, and the launch runs at the
// chip's aggregate request rate (~6-8 TB/s), not at anything the product itself needs: 2 x 4 tiles request 98 MB.
// red: WAVES x MT NT KB.  No row gather (the decoding-step forms keep the 16 x 16 body).
template <int WAVES, int MT, int NT, int U, bool WH>
__device__ __forceinline__ void skinny_plain_body_mn(const SkinnyArgs& a, float* red, int bx, int by) {
    const int lane = threadIdx.x & 63;
    const int r = skinny_ldrow(lane), g = skinny_ldseg(lane);
    const int m0 = by * 16 * MT, nb = bx * 16 * NT;
    const float* ap[MT];
    const float* wp[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) ap[i] = a.A + (int64_t)min(m0 + 16 * i + r, a.M - 1) * a.lda + skinny_koff<WH>(g);
#pragma unroll
    for (int j = 0; j < NT; ++j) wp[j] = skinny_wptr<WH>(a.W, min(nb + 16 * j + r, a.N - 1), a.ldw, skinny_koff<WH>(g));
    skinny_mma_any<WAVES, MT, NT, U, WH>(ap, wp, a.K, red, a.a_scale);
    for (int o = threadIdx.x; o < 256 * MT * NT; o += WAVES * 64) {
        const int tile = o >> 8, erow = (o >> 4) & 15, ecol = o & 15;
        const int em = m0 + 16 * (tile / NT) + erow, ej = nb + 16 * (tile % NT) + ecol;
        if (em >= a.M || ej >= a.N) continue;
        float v = skinny_sum1<WAVES, MT * NT>(red, tile, erow, ecol);
        if (a.bias) v += a.bias[ej];
        if (a.addend) v += a.addend[(int64_t)em * a.ldadd + ej];
        if (a.act == VAG_ACT_TANH) v = vag_tanh(v);
        a.out[(int64_t)em * a.ldo + ej] = v;
    }
}
// WH: the W operand is stored as fp16 (batched launches, blockIdx.z, are fp32-only)
template <int WAVES, bool WH = false>
__global__ __launch_bounds__(WAVES * 64) void skinny_plain_kernel(SkinnyArgs a) {
    __shared__ __attribute__((aligned(16))) float red[WAVES * 64 * 4];
    if (blockIdx.z) {
        a.A += blockIdx.z * a.bsA; a.W += blockIdx.z * a.bsW; a.out += blockIdx.z * a.bsO;
        if (a.addend) a.addend += blockIdx.z * a.bsO;
    }
    skinny_plain_body<WAVES, 4, WH>(a, red, blockIdx.x, blockIdx.y);
}

// out (M,N) = act(A1 W1^T + A2 W2^T + A3 W3^T + b1 + b2 + b3) [* dropout]: the pre-activation of the output head for one
// decoding step (NMT_Decoder.py:137-141) in ONE launch instead of three products and a dropout pass.  16x16 tile per
// workgroup; every wave takes its share of each segment's K and keeps accumulating in registers, one reduction at the end.
struct Skinny3Args {
    const float* A[3]; const float* W[3]; const float* bias[3];
    int64_t lda[3], ldw[3];
    int K[3];
    int M, N;
    float* out; int64_t ldo; int act;
    const uint64_t* rng; int sid; float p; int64_t drop_idx0;      // dropout multiplier of element (m,n): index drop_idx0 + m*N + n
    const float* addend = nullptr; int64_t ldadd = 0;              // optional (M,N) term added before the activation
    const float* addend2 = nullptr; const int64_t* idx2 = nullptr; int64_t ldadd2 = 0;     // ... and a second one, row m from line idx2[m]
};
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void skinny3_kernel(Skinny3Args a) {
    __shared__ __attribute__((aligned(16))) float red[WAVES * 64 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = skinny_ldrow(lane), g = skinny_ldseg(lane);
    const int m0 = blockIdx.y * 16, nb = blockIdx.x * 16;
    const int erow = (threadIdx.x >> 4) & 15, ecol = threadIdx.x & 15;
    const int em = m0 + erow, ej = nb + ecol;
    const bool eok = threadIdx.x < 256 && em < a.M && ej < a.N;
    float pre = 0.f;
    if (eok) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (a.bias[q]) pre += a.bias[q][ej];
        if (a.addend) pre += a.addend[(int64_t)em * a.ldadd + ej];
        if (a.addend2) pre += a.addend2[a.idx2[em] * a.ldadd2 + ej];
    }
    const int src4 = 4 * (16 * (lane & 3) + (lane & 12) + (lane >> 4));
    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int K = a.K[q];
        if (K <= 0) continue;
        const float* ap = a.A[q] + (int64_t)min(m0 + r, a.M - 1) * a.lda[q] + 4 * g;
        const float* wp = a.W[q] + (int64_t)min(nb + r, a.N - 1) * a.ldw[q] + 4 * g;
        const int kper = ((K + WAVES - 1) / WAVES + 15) & ~15;
        const int kbeg = wave * kper, kend = min(K, kbeg + kper);
        constexpr int U = 4;
        for (int c0 = kbeg; c0 < kend; c0 += 16 * U) {
            float4 av[U], wv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = c0 + 16 * u + 4 * g < kend;
                av[u] = ok ? *reinterpret_cast<const float4*>(ap + c0 + 16 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
                wv[u] = ok ? *reinterpret_cast<const float4*>(wp + c0 + 16 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { av[u] = skinny_xpose(av[u], src4); wv[u] = skinny_xpose(wv[u], src4); }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, wv[u].x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, wv[u].y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, wv[u].z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, wv[u].w, acc1, 0, 0, 0);
            }
        }
    }
    *reinterpret_cast<f32x4*>(&red[(wave * 64 + lane) * 4]) = acc0 + acc1;
    __syncthreads();
    if (!eok) return;
    float v = skinny_sum1<WAVES, 1>(red, 0, erow, ecol) + pre;
    if (a.act == VAG_ACT_TANH) v = vag_tanh(v);
    if (a.rng && a.p > 0.f) v *= vag_drop_mul(a.rng, a.sid, (uint64_t)(a.drop_idx0 + (int64_t)em * a.N + ej), a.p);
    a.out[(int64_t)em * a.ldo + ej] = v;
}
static bool skinny_ok(const float* A, int64_t lda, const float* W, int64_t ldw, int64_t K);
// Three-segment product; every (A_q, W_q, K_q) must satisfy the skinny alignment rules, M <= 256.
int vag_skinny3_launch(int64_t M, int64_t N, const float* const* A, const int64_t* lda, const float* const* W, const int64_t* ldw,
                       const int64_t* K, const float* const* bias, float* out, int64_t ldo, int act, const uint64_t* rng, int sid,
                       float p, int64_t drop_idx0, hipStream_t stream, const float* addend, int64_t ldadd, const float* addend2,
                       const int64_t* idx2, int64_t ldadd2) {
    VAG_CHECK_ARG(M > 0 && M <= 256 && N > 0 && out);
    Skinny3Args a;
    for (int q = 0; q < 3; ++q) {
        VAG_CHECK_ARG(K[q] == 0 || (A[q] && W[q] && skinny_ok(A[q], lda[q], W[q], ldw[q], K[q])));      // K = 0: the segment is skipped (its bias still counts)
        a.A[q] = A[q]; a.W[q] = W[q]; a.bias[q] = bias[q]; a.lda[q] = lda[q]; a.ldw[q] = ldw[q]; a.K[q] = (int)K[q];
    }
    a.M = (int)M; a.N = (int)N; a.out = out; a.ldo = ldo; a.act = act;
    a.rng = rng; a.sid = sid; a.p = p; a.drop_idx0 = drop_idx0; a.addend = addend; a.ldadd = ldadd;
    VAG_CHECK_ARG(!addend2 || idx2);
    a.addend2 = addend2; a.idx2 = idx2; a.ldadd2 = ldadd2;
    const dim3 grid((unsigned)cdiv64(N, 16), (unsigned)cdiv64(M, 16));
    hipLaunchKernelGGL(skinny3_kernel<8>, grid, dim3(512), 0, stream, a);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// Horizontal fusion for the decoder steps: an attention dot-product pass (a streaming dot per (row, position), latency
// bound, most CUs half idle) and an INDEPENDENT skinny product in one grid.  Backward: the hidden-side part dgh2 W_hh2
// (+ carry) of the next launch's dh1 = [dq | dgh2] [attn_h; W_hh2] -- its operand is known one launch earlier than dq, so
// it runs beside d alpha instead of lengthening the critical path (K 2560 -> 1024 there).  Forward: W_hh2 h1, which the
// scores do not need (they only need attn_h h1), runs beside them instead of in front of them.
// Blocks [0, nscore) are (position chunk, row) pairs of the dot product, the rest are 16x16 tiles of the product.
static bool skinny_ok(const float* A, int64_t lda, const float* W, int64_t ldw, int64_t K);
struct DotArgs {
    const float* x; const float* q; const float* addend; float* out;      // x (B,Ts,W), q (N,ldq), addend/out (N,Ts)
    const float* v; const float* mask;                                     // MODE 0: out = v . tanh(x + q), masked to -inf
    int64_t ldq;
    int Ts, W, gx, nscore;
};
// DS_WAVES waves per block: that many (row, position) pairs, or the K split of one product tile
// S16: 2-byte storage mode -- the streamed operand x (attention keys / projected keys) and the side product's weights
// are fp16 in memory
template <int MODE, int DS_WAVES, bool S16 = false>
__global__ __launch_bounds__(64 * DS_WAVES) void attn_dot_side_kernel(DotArgs d, SkinnyArgs a, int tiles_x) {
    __shared__ __attribute__((aligned(16))) float red[DS_WAVES * 64 * 4];
    // the side product's blocks come FIRST in the grid: each is one long K loop, the reduction's blocks are many and short -- behind
    // them (round 2-3 order) the side product was the kernel's tail: 35.5 / 42.6 us at configs[4] whatever the reduction cost
    const int nside = gridDim.x - d.nscore;
    if ((int)blockIdx.x < nside) {
        const int t = blockIdx.x;
        skinny_plain_body<DS_WAVES, (MODE == 1 ? 6 : 4), S16>(a, red, t % tiles_x, t / tiles_x);
        return;
    }
    const int id = blockIdx.x - nside;
    const int lane = threadIdx.x & 63;
    const int s = (id % d.gx) * DS_WAVES + (threadIdx.x >> 6);
    if (s >= d.Ts) return;
    const int64_t n = id / d.gx;
    const int64_t xrow = (n * d.Ts + s) * d.W;
    const float* qr = d.q + n * d.ldq;
    float acc = 0.f;
    for (int c = lane * 4; c < d.W; c += 256) {
        const float4 pv = ld4_any<S16>(d.x, xrow + c);
        const float4 qv = *reinterpret_cast<const float4*>(qr + c);
        if (MODE == 0) {
            const float4 vv = *reinterpret_cast<const float4*>(d.v + c);
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
        if (d.addend) acc += d.addend[n * d.Ts + s];
        if (MODE == 0 && d.mask && d.mask[n * d.Ts + s] == 0.f) acc = -INFINITY;
        d.out[n * d.Ts + s] = acc;
    }
}
// Round 4: the same two reductions with the row-constant operands in REGISTERS.  The kernel above re-reads q (and v) for every
// position -- 16 + 16 bytes out of L1 per 8 streamed bytes in the 2-byte mode: the L1, not HBM, set its pace (2.4 / 3.0 TB/s on
// configs[4]'s 84 / 126 MB per step, against 4.1-5.6 TB/s of the neighbouring streaming kernels).  Here a wave owns positions
// s0 + wave, + DS_WAVES, ... of ONE query row and keeps its 8 NJ-float share of q (and v) -- elements 512 j + 8 lane + e -- for all
// of them; a position is NJ 16-byte (fp16 keys) or 2 NJ 16-byte (fp32) loads per lane and nothing else from memory, two positions
// in flight.  W = 512 NJ, NJ in {2, 3, 4, 6} (C = 2H and 3H at H = 512 / 1024); other widths keep the kernel above.
template <bool S16> __device__ __forceinline__ void ld8_any(const float* base, int64_t elem, float (&o)[8]) {
    if (S16) {
        const uint4 p = *reinterpret_cast<const uint4*>(reinterpret_cast<const vag_half*>(base) + elem);
        o[0] = h16_lo(p.x); o[1] = h16_hi(p.x); o[2] = h16_lo(p.y); o[3] = h16_hi(p.y);
        o[4] = h16_lo(p.z); o[5] = h16_hi(p.z); o[6] = h16_lo(p.w); o[7] = h16_hi(p.w);
    } else {
        const float4 a = *reinterpret_cast<const float4*>(base + elem), b = *reinterpret_cast<const float4*>(base + elem + 4);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    }
}
// WIDE: the side product on 2 x 4 tiles per workgroup (M >= 128 rows; 8 waves)
template <int MODE, int DS_WAVES, bool S16, int NJ, bool WIDE>
__global__ __launch_bounds__(64 * DS_WAVES) void attn_dot_side_reg_kernel(DotArgs d, SkinnyArgs a, int tiles_x) {
    constexpr int SMT = WIDE ? 2 : 1, SNT = WIDE ? 4 : 1;
    __shared__ __attribute__((aligned(16))) float red[DS_WAVES * 64 * 4 * SMT * SNT];
    // the side product's blocks come first in the grid (each is one long K loop, the reduction's blocks are many and short).
    // At configs[4] the side product IS the kernel's duration: 35.5 / 42.6 us whatever the reduction costs, until its tiles grew
    const int nside = gridDim.x - d.nscore;
    if ((int)blockIdx.x < nside) {
        const int t = blockIdx.x;
        if (WIDE) skinny_plain_body_mn<DS_WAVES, SMT, SNT, (MODE == 1 ? 4 : 4), S16>(a, red, t % tiles_x, t / tiles_x);
        else skinny_plain_body<DS_WAVES, (MODE == 1 ? 6 : 4), S16>(a, red, t % tiles_x, t / tiles_x);
        return;
    }
    const int id = blockIdx.x - nside;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n = id / d.gx;
    const int ppb = (d.Ts + d.gx - 1) / d.gx;                       // positions of this block: [s0, s1)
    const int s0 = (id % d.gx) * ppb, s1 = min(d.Ts, s0 + ppb);
    const float* qr = d.q + n * d.ldq + 8 * lane;
    float q[NJ][8], v[MODE == 0 ? NJ : 1][8];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        ld8_any<false>(qr, 512 * j, q[j]);
        if (MODE == 0) ld8_any<false>(d.v + 8 * lane, 512 * j, v[j]);
    }
    auto dot = [&](const float (&x)[NJ][8]) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += MODE == 0 ? v[j][e] * vag_tanh(x[j][e] + q[j][e]) : x[j][e] * q[j][e];
        return wave_sum(acc);
    };
    auto put = [&](int s, float acc) {
        if (lane == 0) {
            if (d.addend) acc += d.addend[n * d.Ts + s];
            if (MODE == 0 && d.mask && d.mask[n * d.Ts + s] == 0.f) acc = -INFINITY;
            d.out[n * d.Ts + s] = acc;
        }
    };
    const int64_t xbase = n * d.Ts * (int64_t)d.W + 8 * lane;
    int s = s0 + wave;
    for (; s + DS_WAVES < s1; s += 2 * DS_WAVES) {                   // two positions in flight
        float x0[NJ][8], x1[NJ][8];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            ld8_any<S16>(d.x, xbase + (int64_t)s * d.W + 512 * j, x0[j]);
            ld8_any<S16>(d.x, xbase + (int64_t)(s + DS_WAVES) * d.W + 512 * j, x1[j]);
        }
        put(s, dot(x0));
        put(s + DS_WAVES, dot(x1));
    }
    if (s < s1) {
        float x0[NJ][8];
#pragma unroll
        for (int j = 0; j < NJ; ++j) ld8_any<S16>(d.x, xbase + (int64_t)s * d.W + 512 * j, x0[j]);
        put(s, dot(x0));
    }
}
// mode 1: out (N,Ts) = x[n,s,:] . q[n,:] + addend;  mode 0: out = v . tanh(x[n,s,:] + q[n,:]), -inf where mask (N,Ts) == 0.
// One source row per query row (training: N = B).  Side product in the same grid:
// P (M,Np) = A (M,K) Wt^T + pbias + padd;  A row stride lda, Wt (Np,K) row stride ldw, padd (M,Np) contiguous, P row stride ldp.
int vag_attn_dot_side_launch(int mode, const float* x, const float* q, int64_t ldq, const float* v, const float* mask,
                             const float* addend, int64_t N, int64_t Ts, int64_t W, float* out, int64_t M, int64_t Np,
                             int64_t K, const float* A, int64_t lda, const float* Wt, int64_t ldw, const float* pbias,
                             const float* padd, float* P, int64_t ldp, hipStream_t stream, bool s16) {
    VAG_CHECK_ARG(x && q && out && N > 0 && Ts > 0 && W > 0 && W % 4 == 0 && ldq % 4 == 0 && aligned16(x) && aligned16(q));
    VAG_CHECK_ARG((mode == 1 || (mode == 0 && v && aligned16(v))) && A && Wt && P && M > 0 && Np > 0 &&
                  skinny_ok(A, lda, Wt, ldw, K));
    DotArgs d;
    d.x = x; d.q = q; d.addend = addend; d.out = out; d.v = v; d.mask = mask; d.ldq = ldq; d.Ts = (int)Ts; d.W = (int)W;
    const int DS_WAVES = mode == 0 ? 8 : 16;        // measured: forward side product K = H, backward K = 3H
    d.gx = (int)cdiv64(Ts, DS_WAVES);
    VAG_CHECK_ARG((int64_t)d.gx * N < (1ll << 30));
    d.nscore = (int)(d.gx * N);
    SkinnyArgs a;
    a.A = A; a.W = Wt; a.lda = lda; a.ldw = ldw; a.M = (int)M; a.N = (int)Np; a.K = (int)K;
    a.bias = pbias; a.addend = padd; a.ldadd = Np; a.out = P; a.ldo = ldp; a.act = VAG_ACT_NONE;
    a.a_scale = mode == 1 ? 4096.f : 1.f;           // backward: the riding product's A operand is a gate gradient
    const int tiles_x = (int)cdiv64(Np, 16), tiles_y = (int)cdiv64(M, 16);
    if (W % 512 == 0 && (W == 1024 || W == 1536 || W == 2048 || W == 3072) && aligned16(v ? v : q) && (!s16 || K % 8 == 0) &&
        vag_opt().attn_dot_reg != 0) {
        // row-constant operands in registers (attn_dot_side_reg_kernel): a block = positions [g ppb, (g + 1) ppb) of one row
        const bool wide = M >= 128;                 // the side product on 32 x 64 tiles (skinny_plain_body_mn), 8 waves either way
        const int WV = (mode == 0 || wide) ? 8 : 16;
        int64_t gx = cdiv64(512, N);
        const int64_t gmax = Ts / (2 * WV) > 1 ? Ts / (2 * WV) : 1;
        if (gx > gmax) gx = gmax;
        d.gx = (int)gx;
        d.nscore = (int)(gx * N);
        const int stx = wide ? (int)cdiv64(Np, 64) : tiles_x, sty = wide ? (int)cdiv64(M, 32) : tiles_y;
        const dim3 grid2((unsigned)(d.nscore + stx * sty));
#define VAG_DOTREG(MODE_, WV, S16_, NJ_)                                                                                        \
        { if (wide) hipLaunchKernelGGL((attn_dot_side_reg_kernel<MODE_, WV, S16_, NJ_, true>), grid2, dim3(64 * WV), 0, stream, d, a, stx); \
          else hipLaunchKernelGGL((attn_dot_side_reg_kernel<MODE_, WV, S16_, NJ_, false>), grid2, dim3(64 * WV), 0, stream, d, a, stx); }
#define VAG_DOTREG_NJ(MODE_, WV, S16_)                                                      \
        switch (W / 512) {                                                                  \
            case 2: VAG_DOTREG(MODE_, WV, S16_, 2); break;                                  \
            case 3: VAG_DOTREG(MODE_, WV, S16_, 3); break;                                  \
            case 4: VAG_DOTREG(MODE_, WV, S16_, 4); break;                                  \
            default: VAG_DOTREG(MODE_, WV, S16_, 6); break;                                 \
        }
        if (s16) { if (mode == 0) { VAG_DOTREG_NJ(0, 8, true) } else if (wide) { VAG_DOTREG_NJ(1, 8, true) } else { VAG_DOTREG_NJ(1, 16, true) } }
        else { if (mode == 0) { VAG_DOTREG_NJ(0, 8, false) } else if (wide) { VAG_DOTREG_NJ(1, 8, false) } else { VAG_DOTREG_NJ(1, 16, false) } }
#undef VAG_DOTREG_NJ
#undef VAG_DOTREG
        VAG_LAUNCH_CHECK();
        return VAG_OK;
    }
    const dim3 grid((unsigned)(d.nscore + tiles_x * tiles_y));
    if (s16) {
        VAG_CHECK_ARG(K % 8 == 0 && W % 4 == 0);
        if (mode == 0) hipLaunchKernelGGL((attn_dot_side_kernel<0, 8, true>), grid, dim3(512), 0, stream, d, a, tiles_x);
        else hipLaunchKernelGGL((attn_dot_side_kernel<1, 16, true>), grid, dim3(1024), 0, stream, d, a, tiles_x);
    } else {
        if (mode == 0) hipLaunchKernelGGL((attn_dot_side_kernel<0, 8>), grid, dim3(512), 0, stream, d, a, tiles_x);
        else hipLaunchKernelGGL((attn_dot_side_kernel<1, 16>), grid, dim3(1024), 0, stream, d, a, tiles_x);
    }
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// out[m,n] = sum_k A[m,k] B[k,n] with B stored (K,N) row-major (data gradients dX = dY W of the once-per-batch
// M <= 128-row operators: the LDS-tiled kernels need ~14 us for these however small they are).  A is loaded as in
// skinny_mma; B is read directly in MFMA layout, one dword per lane and k (16 consecutive n = 64 contiguous bytes).
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void skinny_bt_kernel(SkinnyArgs a) {
    __shared__ __attribute__((aligned(16))) float red[WAVES * 64 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = blockIdx.y * 16, nb = blockIdx.x * 16;
    const int erow = (threadIdx.x >> 4) & 15, ecol = threadIdx.x & 15;
    const int em = m0 + erow, ej = nb + ecol;
    const bool eok = threadIdx.x < 256 && em < a.M && ej < a.N;
    const float* ap = a.A + (int64_t)min(m0 + skinny_ldrow(lane), a.M - 1) * a.lda + 4 * skinny_ldseg(lane);
    const int src4 = 4 * (16 * (lane & 3) + (lane & 12) + (lane >> 4));
    const int g = lane >> 4;
    const float* bp = a.W + min(nb + (lane & 15), a.N - 1);
    const int kper = ((a.K + WAVES - 1) / WAVES + 15) & ~15;
    const int kbeg = wave * kper;
    const int kend = min(a.K, kbeg + kper);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    constexpr int U = 2;
    for (int c0 = kbeg; c0 < kend; c0 += 16 * U) {
        float4 av[U];
        float bv[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ka = c0 + 16 * u + 4 * skinny_ldseg(lane);
            av[u] = ka < kend ? *reinterpret_cast<const float4*>(ap + c0 + 16 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
            const int kb = c0 + 16 * u + 4 * g;
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[u][e] = kb < kend ? bp[(int64_t)(kb + e) * a.ldw] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float4 x = skinny_xpose(av[u], src4);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, bv[u][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, bv[u][1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, bv[u][2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, bv[u][3], acc1, 0, 0, 0);
        }
    }
    *reinterpret_cast<f32x4*>(&red[(wave * 64 + lane) * 4]) = acc0 + acc1;
    __syncthreads();
    if (!eok) return;
    float v = skinny_sum1<WAVES, 1>(red, 0, erow, ecol);
    if (a.bias) v += a.bias[ej];
    if (a.addend) v += a.addend[(int64_t)em * a.ldadd + ej];
    a.out[(int64_t)em * a.ldo + ej] = v;
    if (a.out2) {
        float* o2 = a.out2 + (int64_t)em * a.ldo2 + ej;
        *o2 = (a.acc2 ? *o2 : 0.f) + a.scale2 * v;
    }
}

// Fused GRU cell: 16 rows x 16 hidden units x 3 gates per workgroup.  The 256 (row, unit) outputs are finished by the
// first 256 threads, one each; their epilogue operands (the other projection, h_prev, bias) are requested BEFORE the
// product so that they arrive under it instead of costing a second memory round trip.
// MT 16-row tiles per workgroup (2 for wide batches: the three gate rows of W are then re-read by half as many workgroups).
template <int WAVES, bool WH = false, int MT = 1>
__global__ __launch_bounds__(WAVES * 64) void gru_step_kernel(GruStepArgs a) {
    constexpr int NT = 3;
    __shared__ __attribute__((aligned(16))) float red[WAVES * MT * NT * 64 * 4];
    const GruSide& sd = a.s[blockIdx.z];
    const int lane = threadIdx.x & 63;
    const int r = skinny_ldrow(lane), g = skinny_ldseg(lane);
    const int m0 = blockIdx.y * 16 * MT, u0 = blockIdx.x * 16;
    const int H = a.H;
    // epilogue operands: output q of thread t is (row m0 + 16*(t'/256) + (t'%256)/16, unit u0 + t'%16), t' = t + q*threads
    constexpr int OUTS = (MT * 256 + WAVES * 64 - 1) / (WAVES * 64);
    float o_r[OUTS], o_z[OUTS], o_n[OUTS], hp[OUTS], b_r[OUTS], b_z[OUTS], b_n[OUTS];
    bool active[OUTS], eok[OUTS];
#pragma unroll
    for (int q = 0; q < OUTS; ++q) {
        const int t = threadIdx.x + q * WAVES * 64;
        const int em = m0 + 16 * (t >> 8) + ((t >> 4) & 15), ej = u0 + (t & 15);
        eok[q] = t < MT * 256 && em < a.M && ej < H;
        o_r[q] = o_z[q] = o_n[q] = hp[q] = b_r[q] = b_z[q] = b_n[q] = 0.f;
        active[q] = true;
        if (eok[q]) {
            const float* op = sd.other + (sd.other_idx ? sd.other_idx[em] : (int64_t)em) * a.ldother + ej;
            o_r[q] = op[0]; o_z[q] = op[H]; o_n[q] = op[2 * H];
            hp[q] = sd.hprev[(int64_t)em * a.ldh + ej];
            if (sd.bias) { b_r[q] = sd.bias[ej]; b_z[q] = sd.bias[H + ej]; b_n[q] = sd.bias[2 * H + ej]; }
            if (a.lengths) active[q] = sd.t < a.lengths[em];
        }
    }
    const float* ap[MT];
    const float* wp[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) ap[i] = sd.A + (int64_t)min(m0 + 16 * i + r, a.M - 1) * a.lda + skinny_koff<WH>(g);
#pragma unroll
    for (int j = 0; j < NT; ++j) wp[j] = skinny_wptr<WH>(sd.W, min(j * H + u0 + r, 3 * H - 1), a.ldw, skinny_koff<WH>(g));
    skinny_mma_any<WAVES, MT, NT, 4, WH>(ap, wp, a.K, red);
#pragma unroll
    for (int q = 0; q < OUTS; ++q) {
        if (!eok[q]) continue;
        const int t = threadIdx.x + q * WAVES * 64;
        const int mt = t >> 8, erow = (t >> 4) & 15, ecol = t & 15;
        const int em = m0 + 16 * mt + erow, ej = u0 + ecol;
        const float c_r = skinny_sum1<WAVES, MT * NT>(red, mt * NT + 0, erow, ecol) + b_r[q];
        const float c_z = skinny_sum1<WAVES, MT * NT>(red, mt * NT + 1, erow, ecol) + b_z[q];
        const float c_n = skinny_sum1<WAVES, MT * NT>(red, mt * NT + 2, erow, ecol) + b_n[q];
        const float gi_n = a.comp_hidden ? o_n[q] : c_n;
        const float gh_n = a.comp_hidden ? c_n : o_n[q];
        const float rr = vag_sigmoid(c_r + o_r[q]);
        const float zz = vag_sigmoid(c_z + o_z[q]);
        const float nn = vag_tanh(gi_n + rr * gh_n);
        const float hn = (1.f - zz) * nn + zz * hp[q];
        const int64_t o = (int64_t)em * H + ej;
        if (sd.save) {
            const int64_t MH = (int64_t)a.M * H;
            sd.save[o] = rr; sd.save[MH + o] = zz; sd.save[2 * MH + o] = nn; sd.save[3 * MH + o] = gh_n;
        }
        sd.hout[o] = active[q] ? hn : hp[q];
        if (sd.out2) sd.out2[(int64_t)em * a.ld2 + ej] = active[q] ? hn : 0.f;
    }
}

// The same cell with 8 or 4 hidden units per workgroup (2x / 4x the workgroups).  UNITS = 8: tile 0 holds [r | z] of
// the 8 units, tile 1 [n | unused]; UNITS = 4: one tile [r | z | n | unused].  For single-direction launches with
// M <= 64 rows this puts a workgroup on every CU instead of every other one, with less MFMA time and fewer bytes per
// workgroup (unused tile rows issue no loads).
template <int WAVES, int UNITS, bool WH = false>
__global__ __launch_bounds__(WAVES * 64) void gru_step_small_kernel(GruStepArgs a) {
    constexpr int NT = UNITS == 8 ? 2 : 1;
    __shared__ __attribute__((aligned(16))) float red[WAVES * NT * 64 * 4];
    const GruSide& sd = a.s[blockIdx.z];
    const int lane = threadIdx.x & 63;
    const int r = skinny_ldrow(lane), g = skinny_ldseg(lane);
    const int m0 = blockIdx.y * 16, u0 = blockIdx.x * UNITS;
    const int H = a.H;
    // epilogue operands of thread t < 16*UNITS: output (row m0 + t/UNITS, unit u0 + t%UNITS)
    const int erow = (threadIdx.x / UNITS) & 15, ecol = threadIdx.x % UNITS;
    const int em = m0 + erow, ej = u0 + ecol;
    const bool eok = threadIdx.x < 16 * UNITS && em < a.M && ej < H;
    float o_r = 0.f, o_z = 0.f, o_n = 0.f, hp = 0.f, b_r = 0.f, b_z = 0.f, b_n = 0.f;
    bool active = true;
    if (eok) {
        const float* op = sd.other + (sd.other_idx ? sd.other_idx[em] : (int64_t)em) * a.ldother + ej;
        o_r = op[0]; o_z = op[H]; o_n = op[2 * H];
        hp = sd.hprev[(int64_t)em * a.ldh + ej];
        if (sd.bias) { b_r = sd.bias[ej]; b_z = sd.bias[H + ej]; b_n = sd.bias[2 * H + ej]; }
        if (a.lengths) active = sd.t < a.lengths[em];
    }
    const float* ap[1];
    const float* wp[NT];
    ap[0] = sd.A + (int64_t)min(m0 + r, a.M - 1) * a.lda + skinny_koff<WH>(g);
    const int unit = min(u0 + (r % UNITS), H - 1);
    const int gate = r / UNITS;                                   // gate of tile 0's row r
    const int ko = skinny_koff<WH>(g);
    if (UNITS == 8) {
        wp[0] = skinny_wptr<WH>(sd.W, gate * H + unit, a.ldw, ko);                                  // r | z
        wp[NT - 1] = r < 8 ? skinny_wptr<WH>(sd.W, 2 * H + unit, a.ldw, ko) : nullptr;              // n | -
    } else {
        wp[0] = gate < 3 ? skinny_wptr<WH>(sd.W, gate * H + unit, a.ldw, ko) : nullptr;             // r | z | n | -
    }
    skinny_mma_any<WAVES, 1, NT, 4, WH>(ap, wp, a.K, red);
    if (!eok) return;
    const float c_r = skinny_sum1<WAVES, NT>(red, 0, erow, ecol) + b_r;
    const float c_z = skinny_sum1<WAVES, NT>(red, 0, erow, UNITS + ecol) + b_z;
    const float c_n = (UNITS == 8 ? skinny_sum1<WAVES, NT>(red, NT - 1, erow, ecol)
                                  : skinny_sum1<WAVES, NT>(red, 0, erow, 2 * UNITS + ecol)) + b_n;
    const float gi_n = a.comp_hidden ? o_n : c_n;
    const float gh_n = a.comp_hidden ? c_n : o_n;
    const float rr = vag_sigmoid(c_r + o_r);
    const float zz = vag_sigmoid(c_z + o_z);
    const float nn = vag_tanh(gi_n + rr * gh_n);
    const float hn = (1.f - zz) * nn + zz * hp;
    const int64_t o = (int64_t)em * H + ej;
    if (sd.save) {
        const int64_t MH = (int64_t)a.M * H;
        sd.save[o] = rr; sd.save[MH + o] = zz; sd.save[2 * MH + o] = nn; sd.save[3 * MH + o] = gh_n;
    }
    sd.hout[o] = active ? hn : hp;
    if (sd.out2) sd.out2[(int64_t)em * a.ld2 + ej] = active ? hn : 0.f;
}

// dh = A WT^T + addend, then the GRU cell backward (elementwise) that consumes dh -- see common.h.
// Same structure: 256 outputs finished by 256 threads, epilogue operands prefetched under the product.
// (A half-tile variant -- 8 output columns per workgroup on twice the workgroups, as gru_step_small_kernel does for the
// forward cell -- measured no gain here: the MFMA count per workgroup stays that of a full tile.)
// MT x NT 16x16 tiles per workgroup (1 x 1 for M <= 64 rows: one workgroup per CU matters more there; 2 x 2 for the wide
// configuration, where each launch is bound by operand re-reads out of L2: 300 MB -> 150 MB at M = 256, H = 1024).
template <int WAVES, int U, bool WH = false, int MT = 1, int NT = 1>
__global__ __launch_bounds__(WAVES * 64) void gru_bwd_step_kernel(GruBwdStepArgs a) {
    constexpr int TILES = MT * NT;
    __shared__ __attribute__((aligned(16))) float red[WAVES * TILES * 64 * 4];
    const GruBwdStepSide& sd = a.s[blockIdx.z];
    const int lane = threadIdx.x & 63;
    const int r = skinny_ldrow(lane), g = skinny_ldseg(lane);
    const int m0 = blockIdx.y * 16 * MT, nb = blockIdx.x * 16 * NT;
    const int H = a.H;
    const int64_t MH = (int64_t)a.M * H;
    constexpr int OUTS = (TILES * 256 + WAVES * 64 - 1) / (WAVES * 64);      // outputs per thread
    float add[OUTS], e[OUTS], rr[OUTS], zz[OUTS], nn[OUTS], hn[OUTS], hp[OUTS];
    bool active[OUTS], eok[OUTS];
    int64_t oo[OUTS];
#pragma unroll
    for (int q = 0; q < OUTS; ++q) {
        const int t = threadIdx.x + q * WAVES * 64;
        const int tile = t >> 8, erow = (t >> 4) & 15, ecol = t & 15;
        const int em = m0 + 16 * (tile / NT) + erow, ej = nb + 16 * (tile % NT) + ecol;
        eok[q] = tile < TILES && em < a.M && ej < H;
        oo[q] = (int64_t)em * H + ej;
        add[q] = e[q] = rr[q] = zz[q] = nn[q] = hn[q] = hp[q] = 0.f;
        active[q] = true;
        if (eok[q]) {
            if (sd.addend) add[q] = sd.addend[oo[q]];
            if (a.has_cell) {
                if (a.lengths) active[q] = sd.t < a.lengths[em];
                if (sd.dh_add) {
                    e[q] = sd.dh_add[(int64_t)em * a.ld_add + ej];
                    if (a.rng && a.p > 0.f) e[q] *= vag_drop_mul(a.rng, a.sid, (uint64_t)em * a.ld_add + sd.drop_idx0 + ej, a.p);
                }
                rr[q] = sd.save[oo[q]]; zz[q] = sd.save[MH + oo[q]]; nn[q] = sd.save[2 * MH + oo[q]]; hn[q] = sd.save[3 * MH + oo[q]];
                hp[q] = sd.hprev[(int64_t)em * a.ldh + ej];
            }
        }
    }
    const float* ap[MT];
    const float* wp[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) ap[i] = sd.A + (int64_t)min(m0 + 16 * i + r, a.M - 1) * a.lda + skinny_koff<WH>(g);
#pragma unroll
    for (int j = 0; j < NT; ++j) wp[j] = skinny_wptr<WH>(sd.WT, min(nb + 16 * j + r, H - 1), a.ldw, skinny_koff<WH>(g));
    skinny_mma_any<WAVES, MT, NT, U, WH>(ap, wp, a.K, red, 4096.f);       // A = gate gradients: scaled into fp16's range
#pragma unroll
    for (int q = 0; q < OUTS; ++q) {
        if (!eok[q]) continue;
        const int t = threadIdx.x + q * WAVES * 64;
        const int tile = t >> 8, erow = (t >> 4) & 15, ecol = t & 15;
        const int em = m0 + 16 * (tile / NT) + erow, ej = nb + 16 * (tile % NT) + ecol;
        const int64_t o = oo[q];
        float dh = skinny_sum1<WAVES, TILES>(red, tile, erow, ecol) + add[q];
        if (!a.has_cell) {
            sd.dh_out[o] = dh;
            continue;
        }
        float* gi = sd.dgi + (int64_t)em * a.ldgi + ej;
        float* gh = sd.dgh + (int64_t)em * a.ldgh + ej;
        if (!active[q]) {
            gi[0] = 0.f; gi[H] = 0.f; gi[2 * H] = 0.f;
            gh[0] = 0.f; gh[H] = 0.f; gh[2 * H] = 0.f;
            sd.dh_direct[o] = dh;
            continue;
        }
        dh += e[q];
        const float dn_pre = dh * (1.f - zz[q]) * (1.f - nn[q] * nn[q]);
        const float dz_pre = dh * (hp[q] - nn[q]) * zz[q] * (1.f - zz[q]);
        const float dr_pre = dn_pre * hn[q] * rr[q] * (1.f - rr[q]);
        gi[0] = dr_pre; gi[H] = dz_pre; gi[2 * H] = dn_pre;
        gh[0] = dr_pre; gh[H] = dz_pre; gh[2 * H] = dn_pre * rr[q];
        sd.dh_direct[o] = dh * zz[q];
    }
}

static bool skinny_ok(const float* A, int64_t lda, const float* W, int64_t ldw, int64_t K) {
    return aligned16(A) && aligned16(W) && lda % 4 == 0 && ldw % 4 == 0 && K % 4 == 0 && K >= 4;
}

// 32 x 64 outputs per workgroup (skinny_plain_body_mn): wide batches
template <int WAVES, bool WH>
__global__ __launch_bounds__(WAVES * 64) void skinny_plain_mn_kernel(SkinnyArgs a) {
    __shared__ __attribute__((aligned(16))) float red[WAVES * 64 * 4 * 8];
    skinny_plain_body_mn<WAVES, 2, 4, 4, WH>(a, red, blockIdx.x, blockIdx.y);
}
static void skinny_plain_go(const SkinnyArgs& a, hipStream_t stream, bool w16 = false) {
    // wide batches (configs[4]: 256 rows): 32 x 64 tiles request a third of the operand bytes; as long as >= 192 workgroups remain (beam decoding has 192 rows)
    if (a.M >= 128 && a.K > 256 && a.K <= 1024 && !a.row_idx && !a.gather_out && cdiv64(a.N, 64) * cdiv64(a.M, 32) >= 192) {
        dim3 gw((unsigned)cdiv64(a.N, 64), (unsigned)cdiv64(a.M, 32), 1);
        if (w16) hipLaunchKernelGGL((skinny_plain_mn_kernel<8, true>), gw, dim3(512), 0, stream, a);
        else hipLaunchKernelGGL((skinny_plain_mn_kernel<8, false>), gw, dim3(512), 0, stream, a);
        return;
    }
    dim3 grid((unsigned)cdiv64(a.N, 16), (unsigned)cdiv64(a.M, 16), 1);
    if (w16) {
        if (a.K <= 256) hipLaunchKernelGGL((skinny_plain_kernel<4, true>), grid, dim3(256), 0, stream, a);
        else if (a.K <= 1024) hipLaunchKernelGGL((skinny_plain_kernel<8, true>), grid, dim3(512), 0, stream, a);
        else hipLaunchKernelGGL((skinny_plain_kernel<16, true>), grid, dim3(1024), 0, stream, a);
        return;
    }
    if (a.K <= 256) hipLaunchKernelGGL((skinny_plain_kernel<4>), grid, dim3(256), 0, stream, a);
    else if (a.K <= 1024) hipLaunchKernelGGL((skinny_plain_kernel<8>), grid, dim3(512), 0, stream, a);
    else hipLaunchKernelGGL((skinny_plain_kernel<16>), grid, dim3(1024), 0, stream, a);
}

// out (M,N) = table[idx[m], :] W^T + bias, and gathered (M,K) = table[idx[m], :]: embedding lookup and input projection of one
// decoding step in one launch.  M <= 256, K % 4 == 0.
int vag_skinny_gather_launch(int64_t M, int64_t N, int64_t K, const float* table, int64_t ldt, const int64_t* idx, const float* W,
                             int64_t ldw, const float* bias, float* out, int64_t ldo, float* gathered, int64_t ldg,
                             hipStream_t stream) {
    VAG_CHECK_ARG(M > 0 && M <= 256 && N > 0 && table && idx && W && out && gathered && skinny_ok(table, ldt, W, ldw, K) &&
                  aligned16(gathered) && ldg % 4 == 0);
    SkinnyArgs a;
    a.A = table; a.W = W; a.lda = ldt; a.ldw = ldw; a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.bias = bias; a.addend = nullptr; a.ldadd = 0; a.out = out; a.ldo = ldo; a.act = VAG_ACT_NONE;
    a.row_idx = idx; a.gather_out = gathered; a.ld_gather = ldg;
    skinny_plain_go(a, stream);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// ------------------------------------------------------------------------------------------------
// Tall-skinny product for the per-step vocabulary head of decoding and of free-running training steps (round 4):
// out[M, N] = A[M, K] W[N, K]^T + bias, M <= 256 rows (B*k hypotheses, or the batch), N = V large, K = E small.
// (NMT_Decoder.py:143 at one time step: models/...V11.py:148-160 free running, :259-313 beam search.)
// The 16x16-tile skinny kernel requests M*N*K/2 bytes chip-wide (230 MB at 192 x 9391 x 256: 38 us) and the 64x64 LDS-tiled f32
// kernel took 17.5 us in the beam step / 16 us in the greedy step (profiles/r04_decode_kernel_stats.csv): both pay for
// re-reading operands, W above all.  Here a workgroup owns 64 columns for ALL rows: its W tile is read once, split exactly into
// three bf16 planes and kept in LDS (K chunks of 256: 99 KB), wave (row tile, k split) streams its 32 rows of A straight from
// memory in MFMA operand layout, splits them in registers and runs the six products (bf16x6, fp32-grade: the arithmetic of the
// training step's logits product, gemm.hip above).  Requested bytes: N/64 x (M + 64) x K x 4 (38 MB instead of 230).
struct TallArgs {
    const float* A; const float* W; const float* bias; float* out;
    float* parts;              // NULL, or (gridDim.x, M, 2): per row the (max, sum of exp(. - max)) of this workgroup's 64 columns -- the
                               // pieces of the row's log-sum-exp, so that decoding needs no pass over the logits to normalise them
    int64_t lda, ldw, ldo;
    int M, N, K, RT, KS;       // RT = ceil(M / 32) row tiles, KS = k splits: RT * KS <= 8 waves
};
constexpr int TALL_KC = 256;                       // K chunk staged in LDS
constexpr int TALL_LD = TALL_KC + 8;               // bf16 elements per LDS row: 528 B = 132 dwords = 4 mod 64 banks: conflict-free b128 reads
constexpr int TALL_PLANE = 64 * TALL_LD;
constexpr int TALL_LDS_BYTES = 3 * TALL_PLANE * 2; // 101376
__global__ __launch_bounds__(512, 2) void skinny_tall_kernel(TallArgs a) {
    extern __shared__ __attribute__((aligned(16))) __bf16 tall_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 64;
    const int rt = wave / a.KS, ks = wave % a.KS;
    const bool active = rt < a.RT;
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int arow = min(rt * 32 + (lane & 31), a.M - 1);
    const float* Ap = a.A + (int64_t)arow * a.lda + 8 * (lane >> 5);
    // staging map: thread -> (W row = tid >> 3, 32 consecutive k at (tid & 7) * 32)
    const int srow = threadIdx.x >> 3, sk = (threadIdx.x & 7) * 32;
    const float* Wp = a.W + (int64_t)min(n0 + srow, a.N - 1) * a.ldw + sk;
    for (int kc = 0; kc < a.K; kc += TALL_KC) {
        const int klen = min(TALL_KC, a.K - kc);              // multiple of 32 (launcher)
        if (kc > 0) __syncthreads();
        if (sk < klen) {
            float4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const float4*>(Wp + kc + 4 * i);
#pragma unroll
            for (int i = 0; i < 4; ++i) {                      // 8 consecutive k -> one 16-byte store per plane
                unsigned p[3][4];
                split3(v[2 * i].x, v[2 * i].y, p[0][0], p[1][0], p[2][0]);
                split3(v[2 * i].z, v[2 * i].w, p[0][1], p[1][1], p[2][1]);
                split3(v[2 * i + 1].x, v[2 * i + 1].y, p[0][2], p[1][2], p[2][2]);
                split3(v[2 * i + 1].z, v[2 * i + 1].w, p[0][3], p[1][3], p[2][3]);
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    *reinterpret_cast<uint4*>(tall_smem + q * TALL_PLANE + srow * TALL_LD + sk + 8 * i) =
                        make_uint4(p[q][0], p[q][1], p[q][2], p[q][3]);
            }
        }
        __syncthreads();
        if (active) {
            const int per = klen / a.KS;                       // k range of this wave inside the chunk: 64, 128 or 256 (launcher)
            const int kb = ks * per;
            const __bf16* Bf = tall_smem + (lane & 31) * TALL_LD + 8 * (lane >> 5);
            // the wave's rows of A in groups of 64 k (8 x 16-byte loads per lane), the next group in flight under the current
            // one's MFMAs: a k-step that waited for its own two loads cost a memory round trip each (16 us at 16 k-steps)
            const int ng = per >> 6;
            float4 xa[2][8];
            auto load_group = [&](int g, float4 (&x)[8]) {
                const float* src = Ap + kc + kb + 64 * g;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    x[2 * i] = *reinterpret_cast<const float4*>(src + 16 * i);
                    x[2 * i + 1] = *reinterpret_cast<const float4*>(src + 16 * i + 4);
                }
            };
            auto compute_group = [&](int g, const float4 (&x)[8]) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k0 = kb + 64 * g + 16 * i;
                    const float4 x0 = x[2 * i], x1 = x[2 * i + 1];
                    unsigned q[3][4];
                    split3(x0.x, x0.y, q[0][0], q[1][0], q[2][0]);
                    split3(x0.z, x0.w, q[0][1], q[1][1], q[2][1]);
                    split3(x1.x, x1.y, q[0][2], q[1][2], q[2][2]);
                    split3(x1.z, x1.w, q[0][3], q[1][3], q[2][3]);
                    bf16x8 af[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const u32x4 t = {q[p][0], q[p][1], q[p][2], q[p][3]};
                        af[p] = __builtin_bit_cast(bf16x8, t);
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        bf16x8 bf[3];
#pragma unroll
                        for (int p = 0; p < 3; ++p) bf[p] = sp_frag(Bf + p * TALL_PLANE + j * 32 * TALL_LD + k0);
                        // smallest terms first (the order of sp_compute)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[2], acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[1], acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bf[0], acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[1], acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[0], acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0], acc[j], 0, 0, 0);
                    }
                }
            };
            load_group(0, xa[0]);
            for (int g = 0; g < ng; g += 2) {
                if (g + 1 < ng) load_group(g + 1, xa[1]);
                compute_group(g, xa[0]);
                if (g + 1 < ng) {
                    if (g + 2 < ng) load_group(g + 2, xa[0]);
                    compute_group(g + 1, xa[1]);
                }
            }
        }
    }
    // the k splits of a row tile meet in LDS (the W planes are done with)
    if (a.KS > 1) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(tall_smem);      // [wave][j][r][lane]
        if (active && ks > 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((wave * 2 + j) * 16 + r) * 64 + lane] = acc[j][r];
        }
        __syncthreads();
        if (active && ks == 0) {
            for (int o = 1; o < a.KS; ++o)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][r] += red[(((wave + o) * 2 + j) * 16 + r) * 64 + lane];
        }
    }
    if (a.parts) __syncthreads();      // the epilogue below transposes through the LDS the other waves' MFMAs may still be reading
    if (!active || ks != 0) return;
    const int row0 = rt * 32 + 4 * (lane >> 5);
    float bvj[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + j * 32 + (lane & 31);
        bvj[j] = (a.bias && col < a.N) ? a.bias[col] : 0.f;
        if (col >= a.N) continue;
        gemm_epilogue16(acc[j], a.out + (int64_t)row0 * a.ldo + col, a.ldo, a.M - row0, 1.f, 0.f, bvj[j], 0, false);
    }
    if (a.parts) {
        // A row's 64 values sit in the 32 lanes of one half of the wave.  Through LDS (the W planes are done with; 4.2 KB per wave
        // beyond the k-split reduction area, row stride 33) so that lane l owns row l & 31, columns 16 (l >> 5) .. + 15 of a 32-column
        // half, serially; the two column halves and the two lane halves are merged as (max, sum) pairs.  Layout [workgroup][row]:
        // a wave's 32 results are 256 contiguous bytes.
        float* T = reinterpret_cast<float*>(tall_smem) + 16384 + wave * (32 * 33);
        const int rl = lane & 31, ch = lane >> 5;
        float m = -INFINITY, sm = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool cok = n0 + j * 32 + rl < a.N;
#pragma unroll
            for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * ch) * 33 + rl] = cok ? acc[j][r] + bvj[j] : -INFINITY;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            float v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = T[rl * 33 + 16 * ch + e];
            float mj = v[0];
#pragma unroll
            for (int e = 1; e < 16; ++e) mj = fmaxf(mj, v[e]);
            float sj = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) sj += v[e] == -INFINITY ? 0.f : __expf(v[e] - mj);
            const float mn = fmaxf(m, mj);
            sm = (m == -INFINITY ? 0.f : sm * __expf(m - mn)) + (mj == -INFINITY ? 0.f : sj * __expf(mj - mn));
            m = mn;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        const float mo = __shfl_xor(m, 32, 64), so = __shfl_xor(sm, 32, 64);
        const float mn = fmaxf(m, mo);
        sm = (m == -INFINITY ? 0.f : sm * __expf(m - mn)) + (mo == -INFINITY ? 0.f : so * __expf(mo - mn));
        const int row = rt * 32 + rl;
        if (ch == 0 && row < a.M) *reinterpret_cast<float2*>(a.parts + ((int64_t)blockIdx.x * a.M + row) * 2) = make_float2(mn, sm);
    }
}
static bool skinny_tall_ok(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* W, int64_t ldw) {
    // measured (tools/exp_tall.py, 20 launches per graph; us): M = 192: V = 9391 16.2 vs 18.3 on the LDS-tiled f32 kernel, V = 40000 45 vs 49;
    // M = 16: 7.1 vs 4.1 on the 16x16 skinny kernel -- one workgroup per CU is a chain of memory round trips, so only the wide cases come here
    return M > 96 && M <= 256 && N >= 4096 && K >= 256 && K % 256 == 0 && K <= 4096 && lda % 4 == 0 && ldw % 4 == 0 && aligned16(A) &&
           aligned16(W);
}
static int skinny_tall_launch(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* W, int64_t ldw,
                              const float* bias, float* out, int64_t ldo, hipStream_t stream, float* parts = nullptr) {
    TallArgs a;
    a.A = A; a.W = W; a.bias = bias; a.out = out; a.parts = parts; a.lda = lda; a.ldw = ldw; a.ldo = ldo;
    a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.RT = (int)cdiv64(M, 32);
    a.KS = a.RT <= 2 ? 4 : (a.RT <= 4 ? 2 : 1);     // k range per wave and chunk: a multiple of 64 (K % 256 == 0, skinny_tall_ok)
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return VAG_EINVAL;
    if (!(done.load(std::memory_order_acquire) & (1ull << dev))) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(skinny_tall_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                TALL_LDS_BYTES) != hipSuccess) return VAG_EINVAL;
        done.fetch_or(1ull << dev, std::memory_order_release);
    }
    hipLaunchKernelGGL(skinny_tall_kernel, dim3((unsigned)cdiv64(N, 64)), dim3(512), TALL_LDS_BYTES, stream, a);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// The vocabulary product of a decoding step with the pieces of every row's log-sum-exp (TallArgs::parts): number of pieces per
// row (= workgroups), or 0 when the tall-skinny kernel does not take the shape (the caller then normalises with lse_nll_kernel).
int64_t vag_logits_parts_count(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* W, int64_t ldw) {
    return vag_opt().gemm_f32mfma == 0 && skinny_tall_ok(M, N, K, A, lda, W, ldw) ? cdiv64(N, 64) : 0;
}
int vag_logits_parts_launch(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* W, int64_t ldw,
                            const float* bias, float* out, int64_t ldo, float* parts, hipStream_t stream) {
    VAG_CHECK_ARG(A && W && out && parts && vag_logits_parts_count(M, N, K, A, lda, W, ldw) > 0);
    return skinny_tall_launch(M, N, K, A, lda, W, ldw, bias, out, ldo, stream, parts);
}

int vag_skinny_launch(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* W, int64_t ldw,
                      const float* bias, const float* addend, int64_t ldadd, float* out, int64_t ldo, int act,
                      hipStream_t stream, bool w16) {
    VAG_CHECK_ARG(M >= 0 && N >= 0 && K > 0 && A && W && out);
    if (M == 0 || N == 0) return VAG_OK;
    if (w16) {      // W stored as fp16: skinny kernels only (the per-time-step products of the 2-byte storage mode)
        VAG_CHECK_ARG(M <= 256 && K % 8 == 0 && lda % 4 == 0 && ldw % 8 == 0 && aligned16(A) && aligned16(W));
        SkinnyArgs a;
        a.A = A; a.W = W; a.lda = lda; a.ldw = ldw; a.M = (int)M; a.N = (int)N; a.K = (int)K;
        a.bias = bias; a.addend = addend; a.ldadd = ldadd; a.out = out; a.ldo = ldo; a.act = act;
        skinny_plain_go(a, stream, true);
        VAG_LAUNCH_CHECK();
        return VAG_OK;
    }
    // vocabulary-sized products of one decoding / free-running step: the tall-skinny bf16x6 kernel above
    if (!addend && act == 0 && vag_opt().gemm_f32mfma == 0 && skinny_tall_ok(M, N, K, A, lda, W, ldw))
        return skinny_tall_launch(M, N, K, A, lda, W, ldw, bias, out, ldo, stream);
    // every 16x16 output tile re-reads its operand rows: M*N*K/2 bytes requested in all.  Measured at the beam-decode
    // shape (B*k = 192 rows): the 192x2560x512 query/gate product (126 MB) is still faster here (5 vs 16 us), the
    // 192x9391x256 vocabulary product (230 MB) is faster on the LDS-tiled kernel (19 vs 38 us).
    if (!skinny_ok(A, lda, W, ldw, K) || M > 256 || (M > 64 && (double)M * (double)N * (double)K > 350e6)) {
        // generic path through the tiled kernel; an addend is folded in with beta = 1
        if (addend) {
            if (addend != out) VAG_TRY(vag_copy2d_launch(addend, ldadd, out, ldo, M, N, stream));
            return vag_gemm_launch(M, N, K, 1.f, A, lda, 1, W, 1, ldw, 1.f, out, ldo, bias, act, stream);
        }
        return vag_gemm_launch(M, N, K, 1.f, A, lda, 1, W, 1, ldw, 0.f, out, ldo, bias, act, stream);
    }
    SkinnyArgs a;
    a.A = A; a.W = W; a.lda = lda; a.ldw = ldw; a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.bias = bias; a.addend = addend; a.ldadd = ldadd; a.out = out; a.ldo = ldo; a.act = act;
    // Tile choice, from measurements (tools/exp_tiles.py, tools/skinny_probe.hip): these launches are bound by the bytes
    // REQUESTED chip-wide (L2 is dropped at every kernel boundary; ~6.3 TB/s aggregate, ~50 GB/s per CU), duplicates
    // included, so wider/taller tiles or half tiles do not pay at M <= 128; 16x16 with K split over the waves it is.
    skinny_plain_go(a, stream);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// A second destination for the NEXT vag_skinny_nn_launch of the calling thread: out2 (M,N) (+)= scale * (A B) -- an axpy launch saved
// (the initial state's backward: d_ctx += split * dx beside dx itself).  Consumed by that launch whatever path it takes.
struct SkinnyOut2 { float* out2 = nullptr; int64_t ld = 0; float scale = 0.f; int acc = 0; };
static thread_local SkinnyOut2 g_sk_out2;
void vag_skinny_nn_out2(float* out2, int64_t ld, float scale, int accumulate) { g_sk_out2 = SkinnyOut2{out2, ld, scale, accumulate}; }
int vag_axpy_launch(float a, const float* x, float* y, int64_t n, int accumulate, hipStream_t s);      // elem.hip
static int vag_axpy2d_launch(float a, const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int64_t cols, int accumulate, hipStream_t s) {
    if (ldx == cols && ldy == cols) return vag_axpy_launch(a, x, y, rows * cols, accumulate, s);
    for (int64_t r = 0; r < rows; ++r) VAG_TRY(vag_axpy_launch(a, x + r * ldx, y + r * ldy, cols, accumulate, s));     // (not a shape of this library)
    return VAG_OK;
}

// C (M,N) = beta C + A (M,K) B (K,N), B row-major with row stride ldb; M <= 128 (skinny path), else the tiled kernels.
int vag_skinny_nn_launch(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                         float beta, float* C, int64_t ldc, hipStream_t stream) {
    VAG_CHECK_ARG(M >= 0 && N >= 0 && K > 0 && A && B && C && (beta == 0.f || beta == 1.f));
    const SkinnyOut2 o2 = g_sk_out2;
    g_sk_out2 = SkinnyOut2();
    if (M == 0 || N == 0) return VAG_OK;
    if (M > 128 || !aligned16(A) || lda % 4 != 0 || K % 4 != 0 || (double)M * (double)N * (double)K > 350e6) {
        VAG_TRY(vag_gemm_launch(M, N, K, 1.f, A, lda, 1, B, ldb, 1, beta, C, ldc, nullptr, VAG_ACT_NONE, stream));
        if (o2.out2) return vag_axpy2d_launch(o2.scale, C, ldc, o2.out2, o2.ld, M, N, o2.acc, stream);
        return VAG_OK;
    }
    SkinnyArgs a;
    a.out2 = o2.out2; a.ldo2 = o2.ld; a.scale2 = o2.scale; a.acc2 = o2.acc;
    a.A = A; a.W = B; a.lda = lda; a.ldw = ldb; a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.bias = nullptr; a.addend = beta != 0.f ? C : nullptr; a.ldadd = ldc; a.out = C; a.ldo = ldc; a.act = VAG_ACT_NONE;
    dim3 grid((unsigned)cdiv64(N, 16), (unsigned)cdiv64(M, 16), 1);
    if (K <= 256) hipLaunchKernelGGL((skinny_bt_kernel<4>), grid, dim3(256), 0, stream, a);
    else if (K <= 1024) hipLaunchKernelGGL((skinny_bt_kernel<8>), grid, dim3(512), 0, stream, a);
    else hipLaunchKernelGGL((skinny_bt_kernel<16>), grid, dim3(1024), 0, stream, a);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// nb independent products out_z (M,N) = A_z (M,K) W_z^T, operands and outputs at fixed element strides (one launch)
int vag_skinny_batched_launch(int64_t nb, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, int64_t bsA,
                              const float* W, int64_t ldw, int64_t bsW, float* out, int64_t ldo, int64_t bsO,
                              hipStream_t stream) {
    VAG_CHECK_ARG(nb > 0 && nb < 65536 && M > 0 && N > 0 && K > 0 && A && W && out && skinny_ok(A, lda, W, ldw, K));
    VAG_CHECK_ARG(bsA % 4 == 0 && bsW % 4 == 0);
    SkinnyArgs a;
    a.A = A; a.W = W; a.lda = lda; a.ldw = ldw; a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.bias = nullptr; a.addend = nullptr; a.ldadd = 0; a.out = out; a.ldo = ldo; a.act = VAG_ACT_NONE;
    a.bsA = bsA; a.bsW = bsW; a.bsO = bsO;
    dim3 grid((unsigned)cdiv64(N, 16), (unsigned)cdiv64(M, 16), (unsigned)nb);
    if (K <= 256) hipLaunchKernelGGL((skinny_plain_kernel<4>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((skinny_plain_kernel<8>), grid, dim3(512), 0, stream, a);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

int vag_gru_step_launch(const GruStepArgs& a, int nz, hipStream_t stream, bool w16) {
    VAG_CHECK_ARG(a.H > 0 && a.M > 0 && a.K > 0 && (nz == 1 || nz == 2));
    for (int z = 0; z < nz; ++z) {
        VAG_CHECK_ARG(a.s[z].A && a.s[z].W && a.s[z].other && a.s[z].hprev && a.s[z].hout);
        VAG_CHECK_ARG(skinny_ok(a.s[z].A, a.lda, a.s[z].W, a.ldw, a.K));
    }
    VAG_CHECK_ARG(!w16 || (a.K % 8 == 0 && a.ldw % 8 == 0));
    // tile choice from measurements (tools/exp_tiles.py, tools/skinny_probe.hip): 16 units x 16 rows; half tiles on
    // twice the CUs were no faster (the launch is paced by load requests issued chip-wide, duplicates included).
    dim3 grid((unsigned)cdiv64(a.H, 16), (unsigned)cdiv64(a.M, 16), (unsigned)nz);
    constexpr int small = 8;
    const int64_t wgs = (int64_t)grid.x * grid.y * grid.z;
    constexpr int64_t maxwg = 160;
    if (a.K > 256 && wgs <= maxwg && (small == 8 || small == 4)) {
        // fewer than ~2/3 of the CUs would get a workgroup: fewer units each on more workgroups
        const dim3 g8((unsigned)cdiv64(a.H, 8), grid.y, grid.z), g4((unsigned)cdiv64(a.H, 4), grid.y, grid.z);
        if (small == 8) {
            if (w16) hipLaunchKernelGGL((gru_step_small_kernel<8, 8, true>), g8, dim3(512), 0, stream, a);
            else hipLaunchKernelGGL((gru_step_small_kernel<8, 8>), g8, dim3(512), 0, stream, a);
        } else {
            if (w16) hipLaunchKernelGGL((gru_step_small_kernel<8, 4, true>), g4, dim3(512), 0, stream, a);
            else hipLaunchKernelGGL((gru_step_small_kernel<8, 4>), g4, dim3(512), 0, stream, a);
        }
        VAG_LAUNCH_CHECK();
        return VAG_OK;
    }
    // wide batches: two row tiles per workgroup (half the W re-reads) as long as >= 192 workgroups remain (beam decoding has 192 rows)
    constexpr int opt_wide = 1;
    if (opt_wide && a.M >= 128 && a.K > 256 && (int64_t)grid.x * cdiv64(a.M, 32) * nz >= 192) {
        const dim3 gw(grid.x, (unsigned)cdiv64(a.M, 32), grid.z);
        if (w16) hipLaunchKernelGGL((gru_step_kernel<8, true, 2>), gw, dim3(512), 0, stream, a);
        else hipLaunchKernelGGL((gru_step_kernel<8, false, 2>), gw, dim3(512), 0, stream, a);
        VAG_LAUNCH_CHECK();
        return VAG_OK;
    }
    if (w16) {
        if (a.K <= 256) hipLaunchKernelGGL((gru_step_kernel<4, true>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((gru_step_kernel<8, true>), grid, dim3(512), 0, stream, a);
    } else {
        if (a.K <= 256) hipLaunchKernelGGL((gru_step_kernel<4>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((gru_step_kernel<8>), grid, dim3(512), 0, stream, a);
    }
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

int vag_gru_bwd_step_launch(const GruBwdStepArgs& a, int nz, hipStream_t stream, bool w16) {
    VAG_CHECK_ARG(a.H > 0 && a.M > 0 && a.K > 0 && (nz == 1 || nz == 2));
    for (int z = 0; z < nz; ++z) {
        VAG_CHECK_ARG(a.s[z].A && a.s[z].WT && skinny_ok(a.s[z].A, a.lda, a.s[z].WT, a.ldw, a.K));
        if (a.has_cell) VAG_CHECK_ARG(a.s[z].save && a.s[z].hprev && a.s[z].dgi && a.s[z].dgh && a.s[z].dh_direct);
        else VAG_CHECK_ARG(a.s[z].dh_out != nullptr);
    }
    VAG_CHECK_ARG(!w16 || (a.K % 8 == 0 && a.ldw % 8 == 0));
    // wide shapes (M >= 128 rows and still >= 256 workgroups): 2 x 2 tiles per workgroup halve the operand re-reads
    constexpr int opt_wide = 1;
    if (opt_wide && a.M >= 128 && a.K > 1024 && cdiv64(a.H, 32) * cdiv64(a.M, 32) * nz >= 256) {
        dim3 gw((unsigned)cdiv64(a.H, 32), (unsigned)cdiv64(a.M, 32), (unsigned)nz);
        if (w16) hipLaunchKernelGGL((gru_bwd_step_kernel<16, 4, true, 2, 2>), gw, dim3(1024), 0, stream, a);
        else hipLaunchKernelGGL((gru_bwd_step_kernel<16, 2, false, 2, 2>), gw, dim3(1024), 0, stream, a);
        VAG_LAUNCH_CHECK();
        return VAG_OK;
    }
    dim3 grid((unsigned)cdiv64(a.H, 16), (unsigned)cdiv64(a.M, 16), (unsigned)nz);
    // waves by K; chunks in flight so that a wave's K share (K / waves, in 16s) is one round of requests when it fits
#define VAG_BWD_GO(WH)                                                                                                \
    if (a.K <= 256) hipLaunchKernelGGL((gru_bwd_step_kernel<4, 4, WH>), grid, dim3(256), 0, stream, a);               \
    else if (a.K <= 512) hipLaunchKernelGGL((gru_bwd_step_kernel<8, 4, WH>), grid, dim3(512), 0, stream, a);          \
    else if (a.K <= 1024) hipLaunchKernelGGL((gru_bwd_step_kernel<8, 8, WH>), grid, dim3(512), 0, stream, a);         \
    else if (a.K <= 1536) hipLaunchKernelGGL((gru_bwd_step_kernel<16, 6, WH>), grid, dim3(1024), 0, stream, a);       \
    else hipLaunchKernelGGL((gru_bwd_step_kernel<16, 4, WH>), grid, dim3(1024), 0, stream, a);
    if (w16) { VAG_BWD_GO(true) } else { VAG_BWD_GO(false) }
#undef VAG_BWD_GO
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// ------------------------------------------------------------------------------------------------
// column sums (bias gradients) and transpose
// ------------------------------------------------------------------------------------------------
// out (and, when given, out2 / out3: parameters whose gradients are the same column sums) += column sums of X
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int M, int N, int64_t ld,
                                                     int rows_per, float* __restrict__ out, float* __restrict__ out2,
                                                     float* __restrict__ out3) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int m0 = blockIdx.y * rows_per, m1 = min(M, m0 + rows_per);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int m = m0;
    for (; m + 3 < m1; m += 4) {
        s0 += X[(int64_t)m * ld + n];
        s1 += X[(int64_t)(m + 1) * ld + n];
        s2 += X[(int64_t)(m + 2) * ld + n];
        s3 += X[(int64_t)(m + 3) * ld + n];
    }
    for (; m < m1; ++m) s0 += X[(int64_t)m * ld + n];
    const float t = (s0 + s1) + (s2 + s3);
    atomicAdd(out + n, t);
    if (out2) atomicAdd(out2 + n, t);
    if (out3) atomicAdd(out3 + n, t);
}

// Several column sums in one grid (blockIdx.z = task): inside a vag_gemm_group_begin/end bracket the bias-gradient sums
// of an operator are queued like its weight-gradient products and go out together.
constexpr int COLSUM_MAX = 8;
struct ColsumTasks {
    const float* X[COLSUM_MAX]; float* out[COLSUM_MAX]; float* out2[COLSUM_MAX]; float* out3[COLSUM_MAX];
    int64_t ld[COLSUM_MAX];
    int M[COLSUM_MAX], N[COLSUM_MAX], rows_per[COLSUM_MAX];
};
__global__ __launch_bounds__(256) void colsum_multi_kernel(ColsumTasks T) {
    const int k = blockIdx.z;
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int M = T.M[k], N = T.N[k], rows_per = T.rows_per[k];
    const int m0 = blockIdx.y * rows_per;
    if (n >= N || m0 >= M) return;
    const int m1 = min(M, m0 + rows_per);
    const float* X = T.X[k];
    const int64_t ld = T.ld[k];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int m = m0;
    for (; m + 3 < m1; m += 4) {
        s0 += X[(int64_t)m * ld + n];
        s1 += X[(int64_t)(m + 1) * ld + n];
        s2 += X[(int64_t)(m + 2) * ld + n];
        s3 += X[(int64_t)(m + 3) * ld + n];
    }
    for (; m < m1; ++m) s0 += X[(int64_t)m * ld + n];
    const float t = (s0 + s1) + (s2 + s3);
    atomicAdd(T.out[k] + n, t);
    if (T.out2[k]) atomicAdd(T.out2[k] + n, t);
    if (T.out3[k]) atomicAdd(T.out3[k] + n, t);
}
static thread_local bool g_colsum_queue_on = false;
static thread_local int g_colsum_n = 0;
static thread_local ColsumTasks g_colsum;
static thread_local unsigned g_colsum_gx = 0, g_colsum_gy = 0;
void vag_colsum_queue_begin() { g_colsum_queue_on = true; g_colsum_n = 0; g_colsum_gx = g_colsum_gy = 0; }
void vag_colsum_queue_abort() { g_colsum_queue_on = false; g_colsum_n = 0; }
int vag_colsum_queue_flush(hipStream_t stream) {       // launches what is queued; the queue stays open
    const int n = g_colsum_n;
    g_colsum_n = 0;
    const unsigned gx = g_colsum_gx, gy = g_colsum_gy;
    g_colsum_gx = g_colsum_gy = 0;
    if (n == 0) return VAG_OK;
    hipLaunchKernelGGL(colsum_multi_kernel, dim3(gx, gy, (unsigned)n), dim3(256), 0, stream, g_colsum);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

int vag_colsum3_launch(const float* X, int64_t M, int64_t N, int64_t ld, float* out, float* out2, float* out3,
                       hipStream_t stream) {
    VAG_CHECK_ARG(X && out && M >= 0 && N >= 0);
    if (M == 0 || N == 0) return VAG_OK;
    if (!out2 && !out3 && vag_leaf_attach_rowsum(X, M, N, ld, out)) return VAG_OK;     // rides in the held-back product that reads X
    if (g_colsum_queue_on && g_colsum_n < COLSUM_MAX && M < (1ll << 30) && N < (1ll << 30)) {
        const int64_t nbx = cdiv64(N, 256);
        int64_t splits = cdiv64(1024, nbx);
        if (splits > cdiv64(M, 8)) splits = cdiv64(M, 8);
        if (splits < 1) splits = 1;
        const int rows_per = (int)cdiv64(M, splits);
        const int k = g_colsum_n++;
        g_colsum.X[k] = X; g_colsum.out[k] = out; g_colsum.out2[k] = out2; g_colsum.out3[k] = out3; g_colsum.ld[k] = ld;
        g_colsum.M[k] = (int)M; g_colsum.N[k] = (int)N; g_colsum.rows_per[k] = rows_per;
        if ((unsigned)nbx > g_colsum_gx) g_colsum_gx = (unsigned)nbx;
        const unsigned gy = (unsigned)cdiv64(M, rows_per);
        if (gy > g_colsum_gy) g_colsum_gy = gy;
        return VAG_OK;
    }
    const int64_t nbx = cdiv64(N, 256);
    int64_t splits = cdiv64(1024, nbx);
    if (splits > cdiv64(M, 8)) splits = cdiv64(M, 8);
    if (splits < 1) splits = 1;
    const int rows_per = (int)cdiv64(M, splits);
    dim3 grid((unsigned)nbx, (unsigned)cdiv64(M, rows_per));
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, stream, X, (int)M, (int)N, ld, rows_per, out, out2, out3);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}
int vag_colsum_launch(const float* X, int64_t M, int64_t N, int64_t ld, float* out, hipStream_t stream) {
    return vag_colsum3_launch(X, M, N, ld, out, nullptr, nullptr, stream);
}

__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, int M, int N,
                                                        float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int m = m0 + ty + i, n = n0 + tx;
        if (m < M && n < N) tile[ty + i][tx] = in[(int64_t)m * N + n];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int n = n0 + ty + i, m = m0 + tx;
        if (m < M && n < N) out[(int64_t)n * M + m] = tile[tx][ty + i];
    }
}

int vag_transpose_launch(const float* in, int64_t M, int64_t N, float* out, hipStream_t stream) {
    VAG_CHECK_ARG(in && out && M > 0 && N > 0);
    dim3 grid((unsigned)cdiv64(N, 32), (unsigned)cdiv64(M, 32));
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, stream, in, (int)M, (int)N, out);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

