// fp32-grade products on operands that are ALREADY split into bf16 planes in memory ("plane form").
//
// The in-kernel split of gemm.hip (gemm_split_kernel) spends a quarter of every block iteration in VALU work that is
// redone by every block that touches a tile (a logits product splits its (Tt*B, E) operand 74 times) and in the 8-byte LDS
// stores that follow it; MFMA, VALU and LDS phases of an iteration add up instead of overlapping.
// Here the split is done ONCE per matrix by a streaming pass (plane_split_kernel; weights: once per optimiser step) and
// the product's main loop is LDS-DMA + transposed/row fragment reads + MFMA only:
//   * global -> LDS by global_load_lds_dwordx4 (no registers, no ds_write): the LDS image of a tile is lane-linear per
//     wave instruction, so the bank swizzles live in the per-lane SOURCE address (cdna_hip_programming.md T2, rule 21);
//   * k-contiguous operand: image [128 outer rows][64 B] per plane, 16-byte chunk c of row r at slot c ^ ((r >> 2) & 3):
//     a fragment (8 consecutive k of one row) is one ds_read_b128 and 16 rows tile the 64 banks once;
//   * outer-contiguous operand: image [32 k rows][256 B] per plane with the XOR of sp_oc_off (T10 image (b)), fragments by
//     ds_read_b64_tr_b16 -- same reads as gemm.hip;
//   * out-of-range pieces are fetched from a zero page (a lane cannot be masked off in an LDS-DMA without leaving stale
//     bytes in the image), so K, M, N need no padding beyond the planes' own 8-element row granularity;
//   * two stages (96 KB) with the next tile's DMA in flight under the current tile's MFMAs and ONE barrier per k-tile
//     (raw s_barrier + counted waits: a __syncthreads() would drain the DMA), or one stage (48 KB, three blocks per CU).
// Same arithmetic as gemm_split_kernel: x = x1 + x2 + x3 exactly, six products a_i b_j (i + j <= 4), fp32 accumulation.
#include "gemm_shared.h"
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

__device__ __attribute__((aligned(256))) unsigned g_zero_page[64];       // 256 bytes of zeros (never written)

constexpr int PP_PLANE_B = 8192;                 // bytes per plane of one operand tile (128 x 32 bf16)
constexpr int PP_STAGE_B = 6 * PP_PLANE_B;       // A planes then B planes

// ------------------------------------------------------------------------------------------------
// fp32 -> bf16 planes (streaming, 8 elements per thread): out plane p, element (r, c) at out[p*ps + r*ldp + c], c < ldp
// = round8(cols); columns cols .. ldp-1 are written as zeros.
// ------------------------------------------------------------------------------------------------
constexpr int SPLIT_JOBS = 16;
struct SplitJobs {
    const float* src[SPLIT_JOBS]; __bf16* dst[SPLIT_JOBS];
    int64_t ld[SPLIT_JOBS], ps[SPLIT_JOBS];
    int rows[SPLIT_JOBS], cols[SPLIT_JOBS], ldp[SPLIT_JOBS], planes[SPLIT_JOBS];
    int64_t start[SPLIT_JOBS + 1];        // first 8-element chunk of each job in the flat chunk index space
    int n;
};
__global__ __launch_bounds__(256) void plane_split_kernel(SplitJobs J) {
    const int64_t total = J.start[J.n];
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int j = 0;
        while (j + 1 < J.n && i >= J.start[j + 1]) ++j;
        const int64_t c = i - J.start[j];
        const int cpr = J.ldp[j] >> 3;
        const int r = (int)(c / cpr), c0 = (int)(c - (int64_t)r * cpr) << 3;
        const float* s = J.src[j] + (int64_t)r * J.ld[j] + c0;
        float v[8];
        if (c0 + 7 < J.cols[j] && ((reinterpret_cast<uintptr_t>(s) & 15) == 0)) {
            const float4 a = *reinterpret_cast<const float4*>(s), b = *reinterpret_cast<const float4*>(s + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (c0 + e < J.cols[j]) ? s[e] : 0.f;
        }
        unsigned p1[4], p2[4], p3[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split3(v[2 * e], v[2 * e + 1], p1[e], p2[e], p3[e]);
        __bf16* d = J.dst[j] + (int64_t)r * J.ldp[j] + c0;
        *reinterpret_cast<uint4*>(d) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
        if (J.planes[j] >= 2) *reinterpret_cast<uint4*>(d + J.ps[j]) = make_uint4(p2[0], p2[1], p2[2], p2[3]);
        if (J.planes[j] >= 3) *reinterpret_cast<uint4*>(d + 2 * J.ps[j]) = make_uint4(p3[0], p3[1], p3[2], p3[3]);
    }
}

// ------------------------------------------------------------------------------------------------
// main loop
// ------------------------------------------------------------------------------------------------
// One LDS-DMA piece: 64 lanes x 16 bytes from per-lane global addresses to LDS bytes [lds_addr, lds_addr + 1024) (lane l at
// + 16 l).  Inline asm, not __builtin_amdgcn_global_load_lds: hipcc treats the builtin as an LDS store that every later
// ds_read may alias and drains it with s_waitcnt vmcnt(0) in front of the first fragment read of the CURRENT tile, which
// serialises the next tile's transfer with this tile's MFMAs.  An asm piece is outside its bookkeeping
// (cdna_hip_programming.md 5.7 item 1): the loop below counts it by hand (s_waitcnt vmcnt(0) + barrier before the reads).
__device__ __forceinline__ void dma16(const void* g, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(uintptr_t)(lds_void_t*)p;
}

// Per-thread source description of one operand: every thread moves one 16-byte chunk per plane and k-tile.
struct PlaneSrc {
    const __bf16* base;     // plane 0, this thread's chunk at k-tile 0 (already includes row/chunk offsets)
    int64_t ps;             // plane stride (elements)
    int64_t kstep;          // elements to advance per k-tile (KC: 32, OC: 32 * ld)
    int kofs;               // KC: k offset of this thread's chunk inside a tile; OC: this thread's k row inside a tile
    bool oob;               // this thread's chunk lies outside the operand's outer range for the whole product
};
template <bool KC>
__device__ __forceinline__ PlaneSrc plane_src(const __bf16* P, int64_t ps, int64_t ld, int o0, int OUT, int kbeg) {
    const int tid = threadIdx.x;
    PlaneSrc s;
    s.ps = ps;
    if (KC) {
        const int row = tid >> 2, slot = tid & 3, ch = slot ^ ((row >> 2) & 3);
        const int gr = min(o0 + row, OUT - 1);            // rows past the edge: a valid duplicate (those outputs are never stored)
        s.kofs = 8 * ch;
        s.base = P + (int64_t)gr * ld + kbeg + s.kofs;
        s.kstep = 32;
        s.oob = false;
    } else {
        const int kr = tid >> 4, slot = tid & 15, ch = slot ^ (((kr & 3) << 2) | ((kr >> 2) & 3));
        s.kofs = kr;
        s.base = P + (int64_t)(kbeg + kr) * ld + o0 + 8 * ch;
        s.kstep = 32 * ld;
        s.oob = o0 + 8 * ch >= ((OUT + 7) & ~7);
    }
    return s;
}
// issue the DMAs of k-tile starting at k0 (this thread: PL chunks of this operand) into the stage at lds_op
template <bool KC, int PL>
__device__ __forceinline__ void plane_issue(const PlaneSrc& s, int64_t tile, int k0, int kend, unsigned lds_op_wave) {
    // validity of this thread's chunk in this tile: its first k (KC) / its k row (OC) must lie below kend; the rest of a
    // straddling KC chunk is zero in the planes themselves (rows are zero-padded to 8 elements)
    const bool ok = !s.oob && (k0 + s.kofs < kend);
    const __bf16* g = s.base + tile * s.kstep;
#pragma unroll
    for (int p = 0; p < PL; ++p) {
        const void* src = ok ? static_cast<const void*>(g + p * s.ps) : static_cast<const void*>(g_zero_page);
        dma16(src, lds_op_wave + p * PP_PLANE_B);
    }
}

__device__ __forceinline__ bf16x8 pp_frag_kc(const unsigned char* plane, int row, int chunk) {
    return sp_frag(reinterpret_cast<const __bf16*>(plane + 64 * row + 16 * (chunk ^ ((row >> 2) & 3))));
}

template <int PL, bool AKC, bool BKC>
__device__ __forceinline__ void pp_compute(const unsigned char* As, const unsigned char* Bs, int wm, int wn, f32x16 (&acc)[2]) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[2][PL], bf[PL];
#pragma unroll
        for (int p = 0; p < PL; ++p) {
            bf[p] = BKC ? pp_frag_kc(Bs + p * PP_PLANE_B, wn * 32 + r, 2 * ks + h)
                        : sp_frag_tr(reinterpret_cast<const __bf16*>(Bs + p * PP_PLANE_B), wn * 32, ks);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                af[i][p] = AKC ? pp_frag_kc(As + p * PP_PLANE_B, wm * 64 + 32 * i + r, 2 * ks + h)
                               : sp_frag_tr(reinterpret_cast<const __bf16*>(As + p * PP_PLANE_B), wm * 64 + 32 * i, ks);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (PL == 3) {          // smallest terms first
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[PL - 1], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[1], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PL - 1], bf[0], acc[i], 0, 0, 0);
            }
            if (PL >= 2) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[PL >= 2 ? 1 : 0], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PL >= 2 ? 1 : 0], bf[0], acc[i], 0, 0, 0);
            }
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[0], acc[i], 0, 0, 0);
        }
    }
}

template <bool AKC, bool BKC, int PL, int NST>
__device__ __forceinline__ void gemm_planes_body(const GemmArgs& a, unsigned char* smem, int bx, int by, int bz) {
    const int m0 = by * 128, n0 = bx * 128;
    const int kbeg = bz * a.kchunk;
    const int kend = min(a.K, kbeg + a.kchunk);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const PlaneSrc sa = plane_src<AKC>(a.Ap, a.a_ps, a.a_ld, m0, a.M, kbeg);
    const PlaneSrc sb = plane_src<BKC>(a.Bp, a.b_ps, a.b_ld, n0, a.N, kbeg);
    const int nt = (kend - kbeg + 31) >> 5;
    // LDS byte address of this wave's 1 KiB slice of every plane image (wave-uniform: it goes to M0)
    const unsigned wbase = __builtin_amdgcn_readfirstlane(lds_addr_of(smem) + wave * 1024);

    plane_issue<AKC, PL>(sa, 0, kbeg, kend, wbase);
    plane_issue<BKC, PL>(sb, 0, kbeg, kend, wbase + 3 * PP_PLANE_B);
    for (int t = 0; t < nt; ++t) {
        unsigned char* cur = smem + (NST == 2 ? (t & 1) * PP_STAGE_B : 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of tile t have landed
        __builtin_amdgcn_s_barrier();                                // ... everybody's have, and everybody has left tile t-1
        if (NST == 2 && t + 1 < nt) {
            const unsigned nxt = wbase + ((t + 1) & 1) * PP_STAGE_B;
            plane_issue<AKC, PL>(sa, t + 1, kbeg + 32 * (t + 1), kend, nxt);
            plane_issue<BKC, PL>(sb, t + 1, kbeg + 32 * (t + 1), kend, nxt + 3 * PP_PLANE_B);
        }
        pp_compute<PL, AKC, BKC>(cur, cur + 3 * PP_PLANE_B, wm, wn, acc);
        if (NST == 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                            // everybody has read tile t before it is overwritten
            if (t + 1 < nt) {
                plane_issue<AKC, PL>(sa, t + 1, kbeg + 32 * (t + 1), kend, wbase);
                plane_issue<BKC, PL>(sb, t + 1, kbeg + 32 * (t + 1), kend, wbase + 3 * PP_PLANE_B);
            }
        }
    }

    // epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
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

// Workgroup i runs on XCD i % 8 (MI355X_MICROARCH.md): give every XCD a contiguous range of tiles so that the blocks that
// share an operand panel also share an L2 (T1, bijective form).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

template <bool AKC, bool BKC, int PL, int NST>
__global__ __launch_bounds__(512) void gemm_planes_kernel(GemmArgs a, int tn, int tm) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * PP_STAGE_B];
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int bx = id % tn, by = (id / tn) % tm, bz = id / (tn * tm);
    gemm_planes_body<AKC, BKC, PL, NST>(a, smem, bx, by, bz);
}
template <bool AKC, bool BKC, int PL, int NST>
__global__ __launch_bounds__(512) void gemm_planes_group_kernel(GemmGroupArgs G) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * PP_STAGE_B];
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    int p = 0;
    while (p + 1 < G.n && bid >= G.start[p + 1]) ++p;
    const GemmArgs& a = G.p[p];
    const int id = bid - G.start[p];
    const int tn = (a.N + 127) / 128, tm = (a.M + 127) / 128;
    const int bx = id % tn, by = (id / tn) % tm, bz = id / (tn * tm);
    gemm_planes_body<AKC, BKC, PL, NST>(a, smem, bx, by, bz);
}

// ------------------------------------------------------------------------------------------------
// host side: where an operand's planes come from
// ------------------------------------------------------------------------------------------------
// (1) registered regions: the owner of a contiguous fp32 region keeps an element-wise split of the WHOLE region up to date
//     (the step driver: the flat parameter buffer and the derived weights, refreshed once per optimiser step); any operand
//     that lies inside with 8-element-aligned rows uses those planes in place;
// (2) the arena: a caller-owned scratch range (part of the step workspace) into which operands produced during the step
//     are split on demand -- one streaming launch per product group, right before it -- and that is reused as soon as the
//     products reading it have been launched.
// All of it is per host thread (one thread drives a stream), like the group queues of gemm.hip.
namespace {
struct Region { const float* base; int64_t n; const __bf16* planes; int64_t ps; };
struct Made { const float* X; int64_t rows, cols, ld; int planes; const __bf16* p; int64_t ps, ldp; bool queued; };
thread_local std::vector<Region> g_regions;
thread_local std::vector<Made> g_made;
thread_local unsigned char* g_arena = nullptr;
thread_local int64_t g_arena_bytes = 0, g_arena_top = 0;
thread_local SplitJobs g_jobs;
thread_local int g_njobs = 0;
thread_local std::vector<SplitJobs> g_job_batches;       // filled batches waiting for the flush

void push_job(const float* X, int64_t rows, int64_t cols, int64_t ld, __bf16* dst, int64_t ps, int64_t ldp, int planes) {
    if (g_njobs == SPLIT_JOBS) { g_jobs.n = g_njobs; g_job_batches.push_back(g_jobs); g_njobs = 0; }
    const int j = g_njobs++;
    g_jobs.src[j] = X; g_jobs.dst[j] = dst; g_jobs.ld[j] = ld; g_jobs.ps[j] = ps;
    g_jobs.rows[j] = (int)rows; g_jobs.cols[j] = (int)cols; g_jobs.ldp[j] = (int)ldp; g_jobs.planes[j] = planes;
}

// planes of the row-major fp32 view (rows, cols, ld) at X; k_is_col: the product's k runs along the columns
bool planes_for(const float* X, int64_t rows, int64_t cols, int64_t ld, bool k_is_col, int planes, bool queued,
                const __bf16** out, int64_t* ps, int64_t* ldp) {
    for (const Region& r : g_regions) {
        if (X >= r.base && X + (rows - 1) * ld + cols <= r.base + r.n) {
            const int64_t off = X - r.base;
            // rows must start on 8-element boundaries; when k runs along the columns the row must also END on one (a
            // straddling chunk would pull the neighbour's elements into the sum instead of zeros)
            if ((off & 7) || (ld & 7) || (k_is_col && (cols & 7))) break;
            *out = r.planes + off; *ps = r.ps; *ldp = ld;
            return true;
        }
    }
    for (const Made& m : g_made)
        if (m.X == X && m.rows == rows && m.cols == cols && m.ld == ld && m.planes >= planes) {
            *out = m.p; *ps = m.ps; *ldp = m.ldp;
            return true;
        }
    if (!g_arena || rows <= 0 || cols <= 0 || rows >= (1ll << 30) || cols >= (1ll << 30)) return false;
    const int64_t lp = (cols + 7) & ~7ll;
    const int64_t pstride = (rows * lp + 127) & ~127ll;            // planes start 256-byte aligned
    const int64_t bytes = pstride * 2 * planes + 512;              // tail: an outer-contiguous tile may read past the last row
    if (g_arena_top + bytes > g_arena_bytes) return false;
    __bf16* dst = reinterpret_cast<__bf16*>(g_arena + g_arena_top);
    g_arena_top += (bytes + 255) & ~255ll;
    push_job(X, rows, cols, ld, dst, pstride, lp, planes);
    g_made.push_back(Made{X, rows, cols, ld, planes, dst, pstride, lp, queued});
    *out = dst; *ps = pstride; *ldp = lp;
    return true;
}
}  // namespace

void vag_planes_set_arena(void* p, int64_t bytes) {
    g_arena = reinterpret_cast<unsigned char*>(p);
    g_arena_bytes = p ? bytes : 0;
    g_arena_top = 0;
    g_made.clear();
    g_njobs = 0;
    g_job_batches.clear();
}
void vag_planes_registry_clear() { g_regions.clear(); }
void vag_planes_register(const float* base, int64_t n, const void* planes, int64_t ps) {
    if (base && planes && n > 0) g_regions.push_back(Region{base, n, reinterpret_cast<const __bf16*>(planes), ps});
}
bool vag_planes_active() { return g_arena != nullptr || !g_regions.empty(); }

bool vag_planes_attach(GemmArgs& g, bool akc, bool bkc, int planes, bool queued) {
    if (!vag_planes_active()) return false;
    const int64_t top = g_arena_top;
    const size_t nmade = g_made.size();
    const int njobs = g_njobs;
    const size_t nbatches = g_job_batches.size();
    const __bf16 *ap = nullptr, *bp = nullptr;
    int64_t aps = 0, ald = 0, bps = 0, bld = 0;
    // A(m,k): k-contiguous -> rows = M, cols = K, ld = sa_o;  outer-contiguous -> rows = K, cols = M, ld = sa_k
    const bool oka = akc ? planes_for(g.A, g.M, g.K, g.sa_o, true, planes, queued, &ap, &aps, &ald)
                         : planes_for(g.A, g.K, g.M, g.sa_k, false, planes, queued, &ap, &aps, &ald);
    const bool okb = oka && (bkc ? planes_for(g.B, g.N, g.K, g.sb_o, true, planes, queued, &bp, &bps, &bld)
                                 : planes_for(g.B, g.K, g.N, g.sb_k, false, planes, queued, &bp, &bps, &bld));
    if (!okb) {                 // roll back whatever the first operand queued
        if (nbatches == g_job_batches.size()) g_njobs = njobs;         // (a batch boundary was crossed: leave the extra job, harmless)
        if (nbatches == g_job_batches.size()) { g_arena_top = top; g_made.resize(nmade); }
        return false;
    }
    g.Ap = ap; g.a_ps = aps; g.a_ld = ald;
    g.Bp = bp; g.b_ps = bps; g.b_ld = bld;
    return true;
}

int vag_planes_flush_jobs(hipStream_t s) {
    if (g_njobs > 0) { g_jobs.n = g_njobs; g_job_batches.push_back(g_jobs); g_njobs = 0; }
    for (SplitJobs& J : g_job_batches) {
        int64_t total = 0;
        for (int j = 0; j < J.n; ++j) { J.start[j] = total; total += (int64_t)J.rows[j] * (J.ldp[j] >> 3); }
        J.start[J.n] = total;
        int64_t nb = cdiv64(total, 256 * 2);
        if (nb > 4096) nb = 4096;
        if (nb < 1) nb = 1;
        hipLaunchKernelGGL(plane_split_kernel, dim3((unsigned)nb), dim3(256), 0, s, J);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { g_job_batches.clear(); return (int)e; }
    }
    g_job_batches.clear();
    return VAG_OK;
}

// all: every product that reads the arena has been launched -> the arena and the list of planes made so far start over;
// otherwise only the planes made for directly launched products are forgotten (their sources may change again at once)
void vag_planes_release(bool all) {
    if (all) { g_arena_top = 0; g_made.clear(); return; }
    size_t w = 0;
    for (size_t i = 0; i < g_made.size(); ++i)
        if (g_made[i].queued) g_made[w++] = g_made[i];
    g_made.resize(w);
}

// stages: two (96 KB, one block per CU, DMA under the MFMAs) unless the option says one (48 KB, three blocks per CU)
static int planes_stages() { return vag_opt().gemm_plane_stages == 1 ? 1 : 2; }

#define VAG_PP_LAUNCH(KERNEL, AK, BK_, ...)                                                                    \
    do {                                                                                                       \
        const int nst = planes_stages();                                                                       \
        if (planes == 3) {                                                                                     \
            if (nst == 2) hipLaunchKernelGGL((KERNEL<AK, BK_, 3, 2>), __VA_ARGS__);                            \
            else hipLaunchKernelGGL((KERNEL<AK, BK_, 3, 1>), __VA_ARGS__);                                     \
        } else if (planes == 2) {                                                                              \
            if (nst == 2) hipLaunchKernelGGL((KERNEL<AK, BK_, 2, 2>), __VA_ARGS__);                            \
            else hipLaunchKernelGGL((KERNEL<AK, BK_, 2, 1>), __VA_ARGS__);                                     \
        } else {                                                                                               \
            if (nst == 2) hipLaunchKernelGGL((KERNEL<AK, BK_, 1, 2>), __VA_ARGS__);                            \
            else hipLaunchKernelGGL((KERNEL<AK, BK_, 1, 1>), __VA_ARGS__);                                     \
        }                                                                                                      \
    } while (0)

int vag_gemm_planes_dispatch(const GemmArgs& g, bool akc, bool bkc, int planes, dim3 grid, hipStream_t s) {
    VAG_CHECK_ARG(g.Ap && g.Bp && (planes >= 1 && planes <= 3));
    const int tn = (int)grid.x, tm = (int)grid.y;
    const dim3 flat(grid.x * grid.y * grid.z);
    if (akc && bkc) VAG_PP_LAUNCH(gemm_planes_kernel, true, true, flat, dim3(512), 0, s, g, tn, tm);
    else if (akc && !bkc) VAG_PP_LAUNCH(gemm_planes_kernel, true, false, flat, dim3(512), 0, s, g, tn, tm);
    else if (!akc && bkc) VAG_PP_LAUNCH(gemm_planes_kernel, false, true, flat, dim3(512), 0, s, g, tn, tm);
    else VAG_PP_LAUNCH(gemm_planes_kernel, false, false, flat, dim3(512), 0, s, g, tn, tm);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}
int vag_gemm_planes_group_dispatch(const GemmGroupArgs& G, bool akc, bool bkc, int planes, int total, hipStream_t s) {
    VAG_CHECK_ARG(planes >= 1 && planes <= 3 && total > 0);
    const dim3 flat((unsigned)total);
    if (akc && bkc) VAG_PP_LAUNCH(gemm_planes_group_kernel, true, true, flat, dim3(512), 0, s, G);
    else if (akc && !bkc) VAG_PP_LAUNCH(gemm_planes_group_kernel, true, false, flat, dim3(512), 0, s, G);
    else if (!akc && bkc) VAG_PP_LAUNCH(gemm_planes_group_kernel, false, true, flat, dim3(512), 0, s, G);
    else VAG_PP_LAUNCH(gemm_planes_group_kernel, false, false, flat, dim3(512), 0, s, G);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// Element-wise split of a whole contiguous region (a registered region's refresh): planes[p * ps + i] for i < n.
int vag_planes_split_region(const float* src, int64_t n, void* planes, int64_t ps, hipStream_t s) {
    VAG_CHECK_ARG(src && planes && n > 0 && ps >= ((n + 7) & ~7ll) && aligned16(src) && aligned16(planes) && (ps & 7) == 0);
    // as a matrix of 8192-element rows (the last one shorter)
    const int64_t W = 8192;
    SplitJobs J;
    J.n = 0;
    const int64_t full = n / W, rem = n - full * W;
    auto add = [&](const float* p, __bf16* d, int64_t rows, int64_t cols) {
        const int j = J.n++;
        J.src[j] = p; J.dst[j] = d; J.ld[j] = W; J.ps[j] = ps;
        J.rows[j] = (int)rows; J.cols[j] = (int)cols; J.ldp[j] = (int)((cols + 7) & ~7ll); J.planes[j] = 3;
    };
    __bf16* d = reinterpret_cast<__bf16*>(planes);
    for (int64_t r0 = 0; r0 < full; r0 += (1 << 20)) add(src + r0 * W, d + r0 * W, (full - r0 < (1 << 20)) ? full - r0 : (1 << 20), W);
    if (rem) add(src + full * W, d + full * W, 1, rem);
    int64_t total = 0;
    for (int j = 0; j < J.n; ++j) { J.start[j] = total; total += (int64_t)J.rows[j] * (J.ldp[j] >> 3); }
    J.start[J.n] = total;
    int64_t nb = cdiv64(total, 256 * 2);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(plane_split_kernel, dim3((unsigned)nb), dim3(256), 0, s, J);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}
