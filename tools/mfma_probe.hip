// Probe: sustained issue rate of v_mfma_f32_32x32x16_bf16 (no memory traffic), to calibrate the GEMM kernels' MFMA-busy fraction.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(threadIdx.x * 3 + i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int threads, int blocks, int iters) {
    float* out; hipMalloc(&out, (size_t)blocks * threads * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * (threads / 64) * iters * NACC;
    const double fl = n * 32.0 * 32 * 16 * 2;
    printf("NACC=%d waves/block=%d blocks=%d: %.3f ms  %.1f TFLOP/s  (%.1f cycles@2.4GHz per MFMA per SIMD)\n", NACC, threads / 64, blocks, ms,
           fl / ms / 1e9, ms * 1e-3 * 2.4e9 / (n / 1024.0));
    hipFree(out);
}
int main() {
    run<4>(256, 256, 20000);     // 1 wave per SIMD
    run<4>(512, 256, 20000);     // 2 waves per SIMD
    run<2>(512, 256, 20000);
    run<1>(512, 256, 20000);
    run<4>(512, 512, 20000);     // 4 waves per SIMD
    return 0;
}
