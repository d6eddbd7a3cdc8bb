// Declarations of the LDS-tiled products of gemm.hip: argument structs, the epilogue, the exact 3-way bf16 split and the LDS
// fragment reads (kept apart so that probes under tools/ can build against the same pieces).
#pragma once
#include "common.h"

struct GemmArgs {
    const float* A; const float* B; float* C; const float* bias;
    int64_t sa_o, sa_k;   // A(m,k) = A[m*sa_o + k*sa_k]
    int64_t sb_o, sb_k;   // B(k,n) = B[n*sb_o + k*sb_k]
    int64_t ldc;
    int M, N, K, kchunk;
    float alpha, beta;
    int act, splitk;
    int c_half;           // 1: C is stored as fp16 (outputs that the recurrences re-read every step); needs beta == 0, no split-K
    int a_bf16;           // 1: A is stored as bf16 (2 bytes per element, same element strides): one-plane bf16 kernel only
    float* rowsum;        // optional (A outer-contiguous only): rowsum[m] += sum_k A(m,k), i.e. the bias gradient sum_r dY[r,m] of a
                          // weight-gradient product g_W += dY^T X, taken from the A tiles the product loads anyway
    // Split-K WITHOUT one atomic per element and slice (round 6, bf16x6 128 x 128 kernels): every k-slice of an output tile parks
    // its accumulators in a slab of its own (64 KB, write-through stores), takes a ticket, and the block that arrives LAST sums all
    // `nslices` slabs in slice order and runs the ordinary epilogue once.  NULL: the slices add into C with atomics (rounds 1-5).
    float* slab;          // [tiles][nslices][128 x 128] floats of the caller's scratch (vag_gemm_set_scratch)
    unsigned* ticket;     // [tiles] arrival counters, zero between launches (the last block puts its tile's back to zero)
    int nslices;          // k-slices per output tile of THIS product (grid z of a single launch)
};


constexpr int BK = 32;      // k-depth of one LDS stage (one barrier per 32 of K)

// Epilogue of one 32x32 accumulator (this lane: 16 rows row0 + (r&3) + 8*(r>>2) of one column; rows_left = M - row0).
// The beta path requests all 16 old values BEFORE using any of them (a per-element load-use-store sequence costs one
// memory round trip per element: ~15 us for a 64x64-tile product however small it is).
__device__ __forceinline__ void gemm_epilogue16(const f32x16& acc, float* __restrict__ cbase, int64_t ldc, int rows_left,
                                                float alpha, float beta, float bv, int act, bool atomic, int c_half = 0,
                                                int64_t c_elem0 = 0) {
    if (rows_left <= 0) return;
    if (c_half) {          // fp16 output: cbase is the matrix base, c_elem0 the element index of (row0, col)
        vag_half* ch = reinterpret_cast<vag_half*>(cbase) + c_elem0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            float v = alpha * acc[r] + bv;
            if (act == VAG_ACT_TANH) v = vag_tanh(v);
            if (dr < rows_left) ch[(int64_t)dr * ldc] = (vag_half)v;
        }
        return;
    }
    if (atomic) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            if (dr < rows_left) atomicAdd(cbase + (int64_t)dr * ldc, alpha * acc[r] + bv);
        }
        return;
    }
    float old[16];
    if (beta != 0.f) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            old[r] = cbase[(int64_t)min(dr, rows_left - 1) * ldc];       // clamped: no branch around the load
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        float v = alpha * acc[r] + bv;
        if (beta != 0.f) v += beta * old[r];
        if (act == VAG_ACT_TANH) v = vag_tanh(v);
        if (dr < rows_left) cbase[(int64_t)dr * ldc] = v;
    }
}


typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int SP_BK = 32;
constexpr int SP_LD = SP_BK + 8;            // bf16 elements per LDS row of a k-contiguous operand: 80 B = 20 dwords, 16-byte aligned:
                                            // a fragment (8 consecutive k) is ONE ds_read_b128, and 16 consecutive rows tile the 64
                                            // banks exactly once (20 r mod 64 are 16 distinct multiples of 4).  (72-byte rows made the
                                            // compiler pair the two 8-byte halves into ds_read2_b64, which is banked over 32.)
constexpr int SP_PLANE = 128 * SP_LD;       // bf16 elements per plane

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    f32x2 v = {a, b};
    bf16x2 h = __builtin_convertvector(v, bf16x2);     // v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ unsigned pack_f16(float a, float b) {     // two fp16 (round to nearest even), low half = a
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const f16x2 h = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(unsigned, h);
}
// a - b as ONE scalar v_sub_f32: under -O3 the SLP vectoriser pairs the two residuals of a split into v_pk_add_f32, and
// packed f32 VALU beside MFMAs is an anti-lever on gfx950 (MI355X_MICROARCH.md price list: +13 cycles each): measured
// +8-10 % on the k-contiguous products (4096^3 NT 147 -> 160 TF/s, d out.weight 96 -> 88 us).  Inline asm keeps it scalar
// here without switching SLP off for the rest of the file (that cost the recurrent-step kernels more than it gained).
__device__ __forceinline__ float sub_f32(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// split two floats into three packed bf16 pairs (low half = first element)
__device__ __forceinline__ void split3(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = pack_bf16(a, b);
    const float a1 = __builtin_bit_cast(float, p1 << 16), b1 = __builtin_bit_cast(float, p1 & 0xffff0000u);
    const float ra = sub_f32(a, a1), rb = sub_f32(b, b1);
    p2 = pack_bf16(ra, rb);
    const float a2 = __builtin_bit_cast(float, p2 << 16), b2 = __builtin_bit_cast(float, p2 & 0xffff0000u);
    p3 = pack_bf16(sub_f32(ra, a2), sub_f32(rb, b2));
}


// ---- outer-contiguous operands (round 2): LDS image [k][outer], 32 rows of 128 bf16 (256 B), filled with 8-byte stores of
// four consecutive outer elements and read back TRANSPOSED by gfx950's ds_read_b64_tr_b16 (a 16-lane group fetches a
// 4 (k) x 16 (outer) block and each lane receives one outer column's four k values): the MFMA operand's eight consecutive
// k of one row are two such reads.  Before: eight scalar global loads per thread and operand and twelve 4-byte LDS
// stores scattering (k, k+1) pairs into an [outer][k] image.  16-byte chunks of a row are XOR-swizzled with the row
// (cdna_hip_programming.md T10 image (b)): without it the four rows of a transposed read hit the same banks.
// Element offset of columns col..col+3 (col % 4 == 0) of row `row` inside a plane:
__device__ __forceinline__ int sp_oc_off(int row, int col) {
    const int ch = col >> 3;
    return (256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) + 8 * ((col >> 2) & 1)) >> 1;
}
typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
typedef bf16x4v __attribute__((address_space(3))) lds_bf16x4v;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 sp_frag(const __bf16* p) {      // 8 consecutive k of one row: one 16-byte LDS read
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    const u32x4 q = {v.x, v.y, v.z, v.w};
    return __builtin_bit_cast(bf16x8, q);
}

// One k-tile of MFMA work from the LDS planes: 2 k-steps of 16; PL = 3: six bf16 products (fp32-grade), PL = 2: three
// (x = x1 + x2 exactly to 16 significand bits: the 2-byte storage mode, whose operands carry no more than that), PL = 1:
// plain bf16 operands, one product (2-byte mode, the two vocabulary-sized gradient products of the head only).
// fragment of an outer-contiguous operand: rows (outer) ob + (lane & 31), k = ks*16 + 8*(lane >> 5) .. + 7
__device__ __forceinline__ bf16x8 sp_frag_tr(const __bf16* plane, int ob, int ks) {
    const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
    const int col = ob + 16 * (g & 1) + 4 * pp, kb = ks * 16 + 8 * (g >> 1);
    const bf16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4v*)(plane + sp_oc_off(kb + q, col)));
    const bf16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4v*)(plane + sp_oc_off(kb + 4 + q, col)));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// global -> registers (8 floats per thread per operand tile).
// KC (k contiguous): two float4 = (row, 4 consecutive k) items.  OC (outer contiguous): two float4 = (k row, 4 consecutive
// outer) items, stored as they come into the [k][outer] image (sp_oc_off) and transposed by the fragment reads.
template <int NW = 8> struct SpRegsT { float v[4 * (16 / NW)]; };      // 1024 float4 items per operand tile / threads
typedef SpRegsT<8> SpRegs;
// k-contiguous operands: the four rows a 32-lane group writes to LDS with one 8-byte store per lane are 4 apart, not
// consecutive: with the 80-byte row stride rows r, r+4, r+8, r+12 start at banks 0, 16, 0, 16 (mod 32) and tile the 32 banks
// exactly twice (consecutive rows: PMC had 20 % of the LDS-active cycles as bank conflicts).
__device__ __forceinline__ int sp_row(int q) { return (q & ~15) | ((q & 3) << 2) | ((q >> 2) & 3); }

template <bool KC, bool VEC, int NW = 8>
__device__ __forceinline__ void sp_load(const float* __restrict__ P, int64_t so, int64_t sk, int o0, int k0, int OUT,
                                        int KEND, SpRegsT<NW>& r) {
    constexpr int NT = 64 * NW, NI = 1024 / NT;
    const int tid = threadIdx.x;
    if (KC) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int idx = tid + i * NT;
            const int o = o0 + sp_row(idx >> 3), k = k0 + ((idx & 7) << 2);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
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
            r.v[4 * i + 0] = v.x; r.v[4 * i + 1] = v.y; r.v[4 * i + 2] = v.z; r.v[4 * i + 3] = v.w;
        }
    } else {
        // two float4 along the outer dimension per thread: item idx -> k row idx>>5, outer group (idx&31)*4; a wave reads two
        // 512-byte row segments per instruction.  r.v[4i..4i+3] = the four outer elements of item i.
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int idx = tid + i * NT;
            const int k = k0 + (idx >> 5), o = o0 + ((idx & 31) << 2);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
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
            r.v[4 * i + 0] = v.x; r.v[4 * i + 1] = v.y; r.v[4 * i + 2] = v.z; r.v[4 * i + 3] = v.w;
        }
    }
}

// Round 4, the main loop's loads.  sp_load above spends ~45 instructions per 16-byte load on bounds checks and 64-bit address
// arithmetic, every k-tile again -- and these kernels turned out to be bound by vector-instruction ISSUE, not by the matrix pipe,
// the LDS or memory (a ping-pong variant that kept one wave per SIMD purely on MFMAs ran at the speed of its partner's ~330 VALU
// instructions per k-tile; profiles/r04_exp_gemm_pp.txt).  Here everything that does not change from k-tile to k-tile is computed
// once per block: a 32-bit element offset per item relative to a block-uniform base pointer that advances by a constant per
// k-tile.  Rows (or 4-column groups) outside the operand are CLAMPED to a valid one instead of being zero-filled: what they
// feed are accumulator rows / columns the epilogue never stores.  An outer-contiguous group that straddles the edge (o < OUT <=
// o + 3) is loaded whole -- the vectorised kernels require the k-row stride to be a multiple of 4 floats, so the group lies
// inside its row -- and its surplus lanes again only reach outputs that are not stored.  Only FULL k-tiles come this way; a
// last partial tile (K % 32 != 0) takes sp_load, which zero-fills along k.
template <bool KC, int NW = 8> struct SpFast { unsigned off[1024 / (64 * NW)]; const float* base; int64_t step; };
template <bool KC, int NW = 8>
__device__ __forceinline__ void sp_fast_init(SpFast<KC, NW>& f, const float* __restrict__ P, int64_t so, int64_t sk, int o0, int kbeg,
                                             int OUT) {
    constexpr int NT = 64 * NW, NI = 1024 / NT;
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int idx = tid + i * NT;
        if (KC) {
            const int r = min(o0 + sp_row(idx >> 3), OUT - 1) - o0;
            f.off[i] = (unsigned)(r * (int)so + ((idx & 7) << 2));
        } else {
            const int og = (idx & 31) << 2;
            f.off[i] = (unsigned)((idx >> 5) * (int)sk + (o0 + og < OUT ? og : 0));
        }
    }
    f.base = KC ? P + (int64_t)o0 * so + kbeg : P + o0 + (int64_t)kbeg * sk;
    f.step = KC ? (int64_t)SP_BK : (int64_t)SP_BK * sk;
}
template <bool KC, int NW = 8>
__device__ __forceinline__ void sp_fast_load(SpFast<KC, NW>& f, SpRegsT<NW>& r) {
    constexpr int NI = 1024 / (64 * NW);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(f.base + f.off[i]);
        r.v[4 * i + 0] = v.x; r.v[4 * i + 1] = v.y; r.v[4 * i + 2] = v.z; r.v[4 * i + 3] = v.w;
    }
    f.base += f.step;
}

// the same for an operand STORED as bf16 (d(logits) chunks of the 2-byte storage mode): 8-byte items, kept as two packed pairs in
// r.v[4i], r.v[4i+1] exactly as sp_load_bf16 leaves them
template <bool KC> struct SpFastH { unsigned off[2]; const unsigned short* base; int64_t step; };
template <bool KC>
__device__ __forceinline__ void sp_fast_init_bf16(SpFastH<KC>& f, const unsigned short* __restrict__ P, int64_t so, int64_t sk, int o0, int kbeg,
                                                  int OUT) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * 512;
        if (KC) {
            const int r = min(o0 + sp_row(idx >> 3), OUT - 1) - o0;
            f.off[i] = (unsigned)(r * (int)so + ((idx & 7) << 2));
        } else {
            const int og = (idx & 31) << 2;
            f.off[i] = (unsigned)((idx >> 5) * (int)sk + (o0 + og < OUT ? og : 0));
        }
    }
    f.base = KC ? P + (int64_t)o0 * so + kbeg : P + o0 + (int64_t)kbeg * sk;
    f.step = KC ? (int64_t)SP_BK : (int64_t)SP_BK * sk;
}
template <bool KC>
__device__ __forceinline__ void sp_fast_load_bf16(SpFastH<KC>& f, SpRegs& r) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint2 v = *reinterpret_cast<const uint2*>(f.base + f.off[i]);
        r.v[4 * i + 0] = __builtin_bit_cast(float, v.x);
        r.v[4 * i + 1] = __builtin_bit_cast(float, v.y);
        r.v[4 * i + 2] = 0.f; r.v[4 * i + 3] = 0.f;
    }
    f.base += f.step;
}

// The same two items per thread of an operand STORED as bf16 (2-byte storage mode: d(logits) as its producer writes it): eight
// bytes per item, kept as two packed pairs in r.v[4i], r.v[4i+1] -- they ARE the one bf16 plane, sp_store<.., PRE> passes them on.
template <bool KC>
__device__ __forceinline__ void sp_load_bf16(const unsigned short* __restrict__ P, int64_t so, int64_t sk, int o0, int k0, int OUT,
                                             int KEND, SpRegs& r) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * 512;
        unsigned lo = 0u, hi = 0u;
        if (KC) {
            const int o = o0 + sp_row(idx >> 3), k = k0 + ((idx & 7) << 2);
            if (o < OUT) {
                const unsigned short* p = P + (int64_t)o * so + k;
                if (k + 3 < KEND) {
                    const uint2 v = *reinterpret_cast<const uint2*>(p);
                    lo = v.x; hi = v.y;
                } else {
                    if (k + 0 < KEND) lo |= (unsigned)p[0];
                    if (k + 1 < KEND) lo |= (unsigned)p[1] << 16;
                    if (k + 2 < KEND) hi |= (unsigned)p[2];
                    if (k + 3 < KEND) hi |= (unsigned)p[3] << 16;
                }
            }
        } else {
            const int k = k0 + (idx >> 5), o = o0 + ((idx & 31) << 2);
            if (k < KEND) {
                const unsigned short* p = P + (int64_t)k * sk + o;
                if (o + 3 < OUT) {
                    const uint2 v = *reinterpret_cast<const uint2*>(p);
                    lo = v.x; hi = v.y;
                } else {
                    if (o + 0 < OUT) lo |= (unsigned)p[0];
                    if (o + 1 < OUT) lo |= (unsigned)p[1] << 16;
                    if (o + 2 < OUT) hi |= (unsigned)p[2];
                    if (o + 3 < OUT) hi |= (unsigned)p[3] << 16;
                }
            }
        }
        r.v[4 * i + 0] = __builtin_bit_cast(float, lo);
        r.v[4 * i + 1] = __builtin_bit_cast(float, hi);
        r.v[4 * i + 2] = 0.f; r.v[4 * i + 3] = 0.f;
    }
}

// registers -> PL (3, 2 or 1) bf16 planes in LDS, image [outer][k] per plane; F16 (PL = 1 only): one fp16 plane instead
// CHEAT (measurement builds only, -DVAG_CHEAT_B=1: tools/exp_gemm_halfsplit.py): planes 2 and 3 are copies of plane 1 -- the WRONG
// numbers at the instruction count of an operand that arrives pre-split (one pack per pair instead of the 11-instruction split): an
// upper bound on what splitting the weight operand once per optimiser step could buy, before its 6-bytes-per-element ingest
template <bool KC, int PL = 3, bool F16 = false, bool PRE = false, int NW = 8, bool CHEAT = false>
__device__ __forceinline__ void sp_store(__bf16* __restrict__ S, const SpRegsT<NW>& r) {
    constexpr int NT = 64 * NW, NI = 1024 / NT;
    static_assert(!F16 || PL == 1, "the fp16 image is a single plane");
    static_assert(!PRE || (PL == 1 && !F16), "pre-packed bf16 pairs are the one bf16 plane");
    const int tid = threadIdx.x;
    if (KC) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int idx = tid + i * NT;
            const int o = sp_row(idx >> 3), k = (idx & 7) << 2;
            unsigned a1, a2, a3, b1, b2, b3;
            if (PRE) {
                a1 = __builtin_bit_cast(unsigned, r.v[4 * i + 0]); b1 = __builtin_bit_cast(unsigned, r.v[4 * i + 1]);
                a2 = a3 = b2 = b3 = 0;
            } else if (F16) {
                a1 = pack_f16(r.v[4 * i + 0], r.v[4 * i + 1]); b1 = pack_f16(r.v[4 * i + 2], r.v[4 * i + 3]);
                a2 = a3 = b2 = b3 = 0;
            } else if (PL == 1) {
                a1 = pack_bf16(r.v[4 * i + 0], r.v[4 * i + 1]); b1 = pack_bf16(r.v[4 * i + 2], r.v[4 * i + 3]);
                a2 = a3 = b2 = b3 = 0;
            } else if (CHEAT) {
                a1 = a2 = a3 = pack_bf16(r.v[4 * i + 0], r.v[4 * i + 1]); b1 = b2 = b3 = pack_bf16(r.v[4 * i + 2], r.v[4 * i + 3]);
            } else {
                split3(r.v[4 * i + 0], r.v[4 * i + 1], a1, a2, a3);
                split3(r.v[4 * i + 2], r.v[4 * i + 3], b1, b2, b3);
            }
            __bf16* d = S + o * SP_LD + k;
            *reinterpret_cast<uint2*>(d) = make_uint2(a1, b1);
            if (PL >= 2) *reinterpret_cast<uint2*>(d + SP_PLANE) = make_uint2(a2, b2);
            if (PL == 3) *reinterpret_cast<uint2*>(d + 2 * SP_PLANE) = make_uint2(a3, b3);
        }
    } else {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int idx = tid + i * NT;
            unsigned a1, a2, a3, b1, b2, b3;
            if (PRE) {
                a1 = __builtin_bit_cast(unsigned, r.v[4 * i + 0]); b1 = __builtin_bit_cast(unsigned, r.v[4 * i + 1]);
                a2 = a3 = b2 = b3 = 0;
            } else if (F16) {
                a1 = pack_f16(r.v[4 * i + 0], r.v[4 * i + 1]); b1 = pack_f16(r.v[4 * i + 2], r.v[4 * i + 3]);
                a2 = a3 = b2 = b3 = 0;
            } else if (PL == 1) {
                a1 = pack_bf16(r.v[4 * i + 0], r.v[4 * i + 1]); b1 = pack_bf16(r.v[4 * i + 2], r.v[4 * i + 3]);
                a2 = a3 = b2 = b3 = 0;
            } else if (CHEAT) {
                a1 = a2 = a3 = pack_bf16(r.v[4 * i + 0], r.v[4 * i + 1]); b1 = b2 = b3 = pack_bf16(r.v[4 * i + 2], r.v[4 * i + 3]);
            } else {
                split3(r.v[4 * i + 0], r.v[4 * i + 1], a1, a2, a3);
                split3(r.v[4 * i + 2], r.v[4 * i + 3], b1, b2, b3);
            }
            __bf16* d = S + sp_oc_off(idx >> 5, (idx & 31) << 2);
            *reinterpret_cast<uint2*>(d) = make_uint2(a1, b1);
            if (PL >= 2) *reinterpret_cast<uint2*>(d + SP_PLANE) = make_uint2(a2, b2);
            if (PL == 3) *reinterpret_cast<uint2*>(d + 2 * SP_PLANE) = make_uint2(a3, b3);
        }
    }
}

// Af / Bf: fragment base of a k-contiguous operand; As / Bs + (oa, obn): plane base and first outer index of this wave's
// rows for an outer-contiguous one.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int PL, bool AKC, bool BKC, bool F16 = false>
__device__ __forceinline__ void sp_compute(const __bf16* Af, const __bf16* Bf, const __bf16* As, const __bf16* Bs, int oa, int obn,
                                           f32x16 (&acc)[2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[2][PL], bf[PL];
#pragma unroll
        for (int p = 0; p < PL; ++p) {
            bf[p] = BKC ? sp_frag(Bf + p * SP_PLANE + ks * 16) : sp_frag_tr(Bs + p * SP_PLANE, obn, ks);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                af[i][p] = AKC ? sp_frag(Af + p * SP_PLANE + i * 32 * SP_LD + ks * 16) : sp_frag_tr(As + p * SP_PLANE, oa + 32 * i, ks);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // smallest terms first
            if (PL == 3) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[PL - 1], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[1], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PL - 1], bf[0], acc[i], 0, 0, 0);
            }
            if (PL >= 2) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[PL >= 2 ? 1 : 0], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PL >= 2 ? 1 : 0], bf[0], acc[i], 0, 0, 0);
            }
            if (F16)            // the plane holds fp16 bit patterns (sp_store<.., 1, true>): same fragments, the f16 instruction
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i][0]), __builtin_bit_cast(f16x8, bf[0]),
                                                                acc[i], 0, 0, 0);
            else
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[0], acc[i], 0, 0, 0);
        }
    }
}


constexpr int GROUP_MAX = 12;
struct GemmGroupArgs {
    GemmArgs p[GROUP_MAX];
    int start[GROUP_MAX + 1];      // first block of each product
    int n;
};

