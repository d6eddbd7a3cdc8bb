// hipcc -O3 --offload-arch=gfx950 tools/scratch_probe.hip -o scratch_probe
// Does a kernel that uses scratch memory cost more per launch?  (MI355X, 50 launches per graph: 2.03 us without, 4.85 us with 48 bytes
// of scratch per lane, 1 KB or 149 KB of dynamic LDS alike -- ~2.8 us, not the ~18 us of a persistent decoder launch that the
// kernel's own stamps do not account for.)  256 workgroups x 512 threads, 149 KB dynamic LDS (the persistent
// decoder kernels' footprint), trivial work; with and without a scratch-resident array.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <bool SCRATCH>
__global__ __launch_bounds__(512, 1) void k(float* out, int n, int sel) {
    extern __shared__ float lds[];
    float acc = 0.f;
    if (SCRATCH) {
        volatile float a[8];
        for (int i = 0; i < 8; ++i) a[i] = (float)(threadIdx.x + i);
        acc = a[(sel + threadIdx.x) & 7] + a[(sel * 7) & 7] + out[threadIdx.x % n];
    } else {
        acc = out[threadIdx.x % n];
    }
    lds[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[1] + acc;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    float* d; CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int lds = 149 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int mode = 0; mode < 4; ++mode) {
        const bool scratch = mode & 1;
        const int l = (mode & 2) ? lds : 1024;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 50; ++i) {
            if (scratch) hipLaunchKernelGGL(k<true>, dim3(256), dim3(512), l, s, d, 1024, i);
            else hipLaunchKernelGGL(k<false>, dim3(256), dim3(512), l, s, d, 1024, i);
        }
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("scratch %d  lds %6d B: %.2f us per launch\n", (int)scratch, l, ms * 1000.f / 500.f);
    }
    return 0;
}
