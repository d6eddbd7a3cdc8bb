// Probe: do the matrix pipe, the vector ALU and LDS overlap across the waves of one CU on gfx950?  A block of 8 waves (2 per
// SIMD); role of a wave by its index: MFMA loop (the bf16x6 GEMM's 32x32x16 products on 2 accumulators), VALU loop (its
// fp32 -> three bf16 planes split) or LDS loop (its fragment reads).  Times: each role alone (other waves exit at once),
// then mixes with the roles on DIFFERENT waves of every SIMD.  Build: hipcc --offload-arch=gfx950 -O3 overlap_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    f32x2 v = {a, b};
    bf16x2 h = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ void split3(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = pack_bf16(a, b);
    const float a1 = __builtin_bit_cast(float, p1 << 16), b1 = __builtin_bit_cast(float, p1 & 0xffff0000u);
    const float ra = a - a1, rb = b - b1;
    p2 = pack_bf16(ra, rb);
    const float a2 = __builtin_bit_cast(float, p2 << 16), b2 = __builtin_bit_cast(float, p2 & 0xffff0000u);
    p3 = pack_bf16(ra - a2, rb - b2);
}

// roles: bit0 of (mask >> wave) ... role[w] = (roles >> (2*w)) & 3: 0 idle, 1 MFMA, 2 VALU, 3 LDS
__global__ __launch_bounds__(512) void probe(float* out, int iters, unsigned roles) {
    __shared__ __attribute__((aligned(16))) unsigned lds[8 * 64 * 8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int role = (roles >> (2 * wave)) & 3;
    for (int i = threadIdx.x; i < 8 * 64 * 8; i += 512) lds[i] = i * 2654435761u;
    __syncthreads();
    float s = 0.f;
    if (role == 1) {
        f32x16 acc[2];
        for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane * 3 + i); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 12; ++j) {       // 24 MFMAs = one k-tile of the GEMM wave
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[1], 0, 0, 0);
            }
        }
        for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else if (role == 2) {
        float x[16];
        for (int i = 0; i < 16; ++i) x[i] = (float)(lane * 16 + i) * 1.0001f;
        unsigned acc = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {        // 16 floats = one k-tile's share of both operands
                unsigned p1, p2, p3;
                split3(x[2 * i], x[2 * i + 1], p1, p2, p3);
                acc ^= p1 + p2 * 3 + p3 * 5;
                x[2 * i] += 1.f; x[2 * i + 1] += 2.f;
            }
        }
        s = (float)acc;
    } else if (role == 3) {
        uint2 acc = make_uint2(0, 0);
        const uint2* p = reinterpret_cast<const uint2*>(lds) + lane;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 36; ++j) {       // 18 fragments of 2 x b64 per k-tile
                const uint2 v = p[((j * 7 + it) & 7) * 64];
                acc.x ^= v.x; acc.y += v.y;
            }
        }
        s = (float)(acc.x + acc.y);
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

static float run(unsigned roles, int blocks, int iters) {
    static float* out = nullptr;
    if (!out) hipMalloc(&out, (size_t)4096 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, out, iters, roles);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, out, iters, roles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
static unsigned roles_of(const int r[8]) { unsigned m = 0; for (int w = 0; w < 8; ++w) m |= (unsigned)r[w] << (2 * w); return m; }
int main() {
    const int iters = 4000, blocks = 256;      // one block per CU; waves w and w+4 share a SIMD
    struct { const char* name; int r[8]; } cases[] = {
        {"MFMA on waves 0-3 (1 per SIMD)          ", {1, 1, 1, 1, 0, 0, 0, 0}},
        {"MFMA on all 8 waves (2 per SIMD)        ", {1, 1, 1, 1, 1, 1, 1, 1}},
        {"VALU split on waves 4-7                 ", {0, 0, 0, 0, 2, 2, 2, 2}},
        {"VALU split on all 8 waves               ", {2, 2, 2, 2, 2, 2, 2, 2}},
        {"LDS reads on waves 4-7                  ", {0, 0, 0, 0, 3, 3, 3, 3}},
        {"LDS reads on all 8 waves                ", {3, 3, 3, 3, 3, 3, 3, 3}},
        {"MFMA 0-3 + VALU 4-7                     ", {1, 1, 1, 1, 2, 2, 2, 2}},
        {"MFMA 0-3 + LDS 4-7                      ", {1, 1, 1, 1, 3, 3, 3, 3}},
        {"VALU 0-3 + LDS 4-7                      ", {2, 2, 2, 2, 3, 3, 3, 3}},
        {"MFMA 0,1 VALU 2,3 | LDS 4,5 MFMA 6,7    ", {1, 1, 2, 2, 3, 3, 1, 1}},
    };
    for (auto& c : cases) {
        const float ms = run(roles_of(c.r), blocks, iters);
        printf("%s %.3f ms  = %.0f cycles per k-tile\n", c.name, ms, ms * 1e-3 * 2.4e9 / iters);
    }
    return 0;
}
