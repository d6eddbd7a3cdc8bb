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

constexpr int GROUP_MAX = 12;
struct GemmGroupArgs {
    GemmArgs p[GROUP_MAX];
    int start[GROUP_MAX + 1];      // first block of each product
    int n;
};
