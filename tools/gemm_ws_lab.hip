// Lab (round 2): wave-specialised bf16x6 GEMM, NT layout (both operands k-contiguous), full tiles only.
// 12 waves per block: waves 0-7 consume (LDS fragments -> MFMA, 64x32 each, as the shipped kernel), waves 8-11 produce
// (global -> registers -> three bf16 planes -> the OTHER LDS stage).  One barrier per k-tile.  Motivation:
// tools/overlap_probe.hip shows MFMA waves overlap LDS waves completely and VALU waves by half, while the shipped kernel's
// phases simply add up.  Build: hipcc --offload-arch=gfx950 -O3 -o gemm_ws_lab gemm_ws_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32, LD = BK + 4, PLANE = 128 * LD, STAGE = 6 * PLANE;

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
__device__ __forceinline__ bf16x8 frag(const __bf16* p) {
    const uint2 lo = *reinterpret_cast<const uint2*>(p);
    const uint2 hi = *reinterpret_cast<const uint2*>(p + 4);
    const u32x4 q = {lo.x, lo.y, hi.x, hi.y};
    return __builtin_bit_cast(bf16x8, q);
}
__device__ __forceinline__ void store4(__bf16* S, int row, int k, float4 v) {
    unsigned a1, a2, a3, b1, b2, b3;
    split3(v.x, v.y, a1, a2, a3);
    split3(v.z, v.w, b1, b2, b3);
    __bf16* d = S + row * LD + k;
    *reinterpret_cast<uint2*>(d) = make_uint2(a1, b1);
    *reinterpret_cast<uint2*>(d + PLANE) = make_uint2(a2, b2);
    *reinterpret_cast<uint2*>(d + 2 * PLANE) = make_uint2(a3, b3);
}

template <int PROD>       // producer waves: 4 or 8
__global__ __launch_bounds__(512 + 64 * PROD) void gemm_ws(const float* __restrict__ A, const float* __restrict__ B,
                                                           float* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) __bf16 smem[];      // 2 stages x (3 A planes + 3 B planes)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    const int T = K / BK;
    const bool producer = wave >= 8;
    constexpr int PT = 64 * PROD;                 // producer threads
    constexpr int PER = 1024 / PT;                // float4 per operand per producer thread (4 or 2)
    float4 ra[PER], rb[PER];
    const int p = tid - 512;
    auto gload = [&](int t) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = p + i * PT, row = idx >> 3, k4 = (idx & 7) << 2;
            ra[i] = *reinterpret_cast<const float4*>(A + (size_t)(m0 + row) * K + t * BK + k4);
            rb[i] = *reinterpret_cast<const float4*>(B + (size_t)(n0 + row) * K + t * BK + k4);
        }
    };
    auto lstore = [&](int stage) {
        __bf16* As = smem + stage * STAGE;
        __bf16* Bs = As + 3 * PLANE;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = p + i * PT, row = idx >> 3, k4 = (idx & 7) << 2;
            store4(As, row, k4, ra[i]);
            store4(Bs, row, k4, rb[i]);
        }
    };
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int wm = (wave & 7) >> 2, wn = wave & 3;
    if (producer) {
        gload(0);
        lstore(0);
        if (T > 1) gload(1);
    }
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        if (producer) {
            if (t + 1 < T) {
                lstore((t + 1) & 1);
                if (t + 2 < T) gload(t + 2);
            }
        } else {
            const __bf16* As = smem + (t & 1) * STAGE;
            const __bf16* Af = As + (wm * 64 + (lane & 31)) * LD + 8 * (lane >> 5);
            const __bf16* Bf = As + 3 * PLANE + (wn * 32 + (lane & 31)) * LD + 8 * (lane >> 5);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 af[2][3], bf[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    bf[q] = frag(Bf + q * PLANE + ks * 16);
#pragma unroll
                    for (int i = 0; i < 2; ++i) af[i][q] = frag(Af + q * PLANE + i * 32 * LD + ks * 16);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[2], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[1], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[0], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[1], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[0], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[0], acc[i], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    if (producer) return;
    const int col = n0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row0 = m0 + wm * 64 + i * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 16; ++r) C[(size_t)(row0 + (r & 3) + 8 * (r >> 2)) * N + col] = acc[i][r];
    }
}

template <int PROD>
static void run(int M, int N, int K) {
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    srand(1);
    for (auto& x : hA) x = (float)rand() / RAND_MAX - 0.5f;
    for (auto& x : hB) x = (float)rand() / RAND_MAX - 0.5f;
    float *A, *B, *C;
    hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4); hipMalloc(&C, (size_t)M * N * 4);
    hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    const size_t lds = 2 * STAGE * sizeof(__bf16);
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ws<PROD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid(N / 128, M / 128), block(512 + 64 * PROD);
    hipLaunchKernelGGL(gemm_ws<PROD>, grid, block, lds, 0, A, B, C, M, N, K);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(gemm_ws<PROD>, grid, block, lds, 0, A, B, C, M, N, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    std::vector<float> hC((size_t)M * N);
    hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int s = 0; s < 64; ++s) {
        const int m = (s * 977) % M, n = (s * 4099) % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)m * K + k] * hB[(size_t)n * K + k];
        worst = fmax(worst, fabs(ref - hC[(size_t)m * N + n]));
    }
    printf("producers=%d  %dx%dx%d  %.1f us  %.1f TF/s fp32-equivalent  max abs err %.2e (hipError %d)\n", PROD, M, N, K, ms * 1e3,
           2.0 * M * N * K / ms / 1e9, worst, (int)hipGetLastError());
    hipFree(A); hipFree(B); hipFree(C);
}
int main() {
    for (int rep = 0; rep < 2; ++rep) {
        run<4>(4096, 4096, 4096);
        run<4>(4096, 4096, 1024);
        run<4>(2560, 9472, 256);
        run<4>(2560, 1536, 256);
        run<8>(4096, 4096, 4096);
        run<8>(4096, 4096, 1024);
        run<8>(2560, 9472, 256);
    }
    return 0;
}
