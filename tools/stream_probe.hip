// Probe: how fast can W workgroups each stream a private contiguous slab (float4 loads, U in flight per thread)?
// Decides whether "one workgroup per batch row" attention kernels (scores+context fused) can beat two full-grid launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int U>
__global__ void stream_kernel(const float4* __restrict__ src, int64_t f4_per_wg, float* __restrict__ out) {
    const float4* p = src + (int64_t)blockIdx.x * f4_per_wg;
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < f4_per_wg; i += (int64_t)blockDim.x * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t j = i + (int64_t)u * blockDim.x;
            v[u] = j < f4_per_wg ? p[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ void touch_kernel(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }

template <int U>
void run(int wgs, int threads, int kb_per_wg, float4* src, float* out, float* flag) {
    const int64_t f4 = (int64_t)kb_per_wg * 1024 / 16;
    hipGraph_t graph; hipGraphExec_t exec; hipStream_t s; hipStreamCreate(&s);
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 50; ++i) {
        hipLaunchKernelGGL(stream_kernel<U>, dim3(wgs), dim3(threads), 0, s, src, f4, out);
        hipLaunchKernelGGL(touch_kernel, dim3(1), dim3(64), 0, s, flag);       // a dependent tiny kernel in between
    }
    hipStreamEndCapture(s, &graph); hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphLaunch(exec, s); hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s); hipGraphLaunch(exec, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / 50 - 1.55;      // minus the tiny kernel's boundary cost
    printf("wgs=%4d threads=%4d U=%d  %4d KB/WG (total %.1f MB): %.2f us -> %.0f GB/s per WG, %.2f TB/s\n", wgs, threads, U, kb_per_wg,
           wgs * kb_per_wg / 1024.0, us, kb_per_wg * 1024.0 / us / 1e3, wgs * kb_per_wg * 1024.0 / us / 1e6);
    hipGraphExecDestroy(exec); hipGraphDestroy(graph); hipStreamDestroy(s);
}

int main() {
    float4* src; float* out; float* flag;
    hipMalloc(&src, 64ll << 20); hipMemset(src, 0, 64ll << 20);
    hipMalloc(&out, 4 << 20); hipMalloc(&flag, 256); hipMemset(flag, 0, 256);
    run<4>(64, 1024, 328, src, out, flag);
    run<8>(64, 1024, 328, src, out, flag);
    run<4>(128, 1024, 246, src, out, flag);
    run<8>(128, 1024, 246, src, out, flag);
    run<8>(128, 512, 246, src, out, flag);
    run<4>(256, 1024, 164, src, out, flag);
    run<8>(256, 512, 164, src, out, flag);
    run<4>(256, 1024, 82, src, out, flag);
    run<4>(640, 256, 16, src, out, flag);        // today's scores kernel shape: 10.5 MB over 640 WGs
    run<4>(2560, 64, 4, src, out, flag);
    run<8>(256, 64, 41, src, out, flag);         // today's ctx kernel shape: 256 single-wave WGs
    return 0;
}
