#include <hip/hip_runtime.h>
__global__ void k(unsigned* out) {
    unsigned x = __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11));
    if (threadIdx.x == 0) out[blockIdx.x] = x;
}
int main() {
    unsigned* d; hipMalloc(&d, 512 * 4);
    hipLaunchKernelGGL(k, dim3(512), dim3(64), 0, 0, d);
    unsigned h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 32; ++i) printf("%u ", h[i]); printf("\n");
    int cnt[16] = {0}; for (int i = 0; i < 256; ++i) cnt[h[i] & 15]++;
    for (int i = 0; i < 16; ++i) printf("%d ", cnt[i]); printf("\n");
    return 0;
}
