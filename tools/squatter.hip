// A kernel with the footprint of a collective's device kernel (RCCL: tens of workgroups of 256-512 threads, ~64 VGPRs, some LDS,
// resident for the whole transfer): it occupies CU resources for a given time and does nothing else.  Used by
// tools/exp_squatter.py to probe what happens to a persistent recurrence kernel -- which needs all its workgroups resident, one
// per CU -- when such a kernel runs beside it (graph B's encoder backward beside bucket 0's all-reduce).
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/squatter.hip -o tools/libsquatter.so
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ void squat_kernel(unsigned long long ticks, float* sink) {
    extern __shared__ float lds[];
    // ~64 live VGPRs: a small register-resident state that the loop keeps rotating
    float r[48];
#pragma unroll
    for (int i = 0; i < 48; ++i) r[i] = (float)(threadIdx.x + i);
    lds[threadIdx.x] = 1.f;
    const unsigned long long t0 = wall_clock64();          // 100 MHz
    while (wall_clock64() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < 48; ++i) r[i] = r[i] * 1.0001f + r[(i + 7) % 48] * 1e-6f;
        __builtin_amdgcn_s_sleep(8);
    }
    float acc = lds[threadIdx.x];
#pragma unroll
    for (int i = 0; i < 48; ++i) acc += r[i];
    if (acc == 123.456f) sink[0] = acc;                    // never true: keeps the registers alive
}

extern "C" int squat(void* stream, int wgs, int threads, int lds_bytes, double us, float* sink) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(squat_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
            hipSuccess) return -1;
        attr = true;
    }
    hipLaunchKernelGGL(squat_kernel, dim3(wgs), dim3(threads), (size_t)lds_bytes, (hipStream_t)stream,
                       (unsigned long long)(us * 100.0), sink);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
