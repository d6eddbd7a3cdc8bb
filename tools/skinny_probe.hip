// Diagnostic: where does a skinny (M=64) product spend its time?  s_memtime stamps per wave.
// Build: hipcc -O3 --offload-arch=gfx950 tools/skinny_probe.hip -o gpurun_out/skinny_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WAVES, int U, int FULL = 0>
__global__ __launch_bounds__(WAVES * 64) void probe(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ out,
                                                    int M, int N, int K, unsigned long long* stamps, int do_stamp) {
    __shared__ __attribute__((aligned(16))) float red[WAVES * 64 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * 16, nb = blockIdx.x * 16;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const int kper = ((K + WAVES - 1) / WAVES + 15) & ~15;
    const int kbeg = wave * kper, kend = min(K, kbeg + kper);
    // FULL: timing-only access pattern, 8 rows x 128 contiguous bytes per wave request (results are NOT a product)
    // FULL=1: 8 rows x 128 contiguous bytes per wave request; FULL=2: 16 rows x 64 B with every lane QUAD contiguous
    const int qrow = 4 * ((lane >> 2) & 3) + (lane >> 4), qseg = lane & 3;
    const float* ap = FULL == 1 ? A + (long)min(m0 + (lane >> 3), M - 1) * K + 4 * (lane & 7)
                    : FULL == 2 ? A + (long)min(m0 + qrow, M - 1) * K + 4 * qseg
                                : A + (long)min(m0 + r, M - 1) * K + 4 * g;
    const float* wp = FULL == 1 ? W + (long)min(nb + (lane >> 3), N - 1) * K + 4 * (lane & 7)
                    : FULL == 2 ? W + (long)min(nb + qrow, N - 1) * K + 4 * qseg
                                : W + (long)min(nb + r, N - 1) * K + 4 * g;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    unsigned long long t1 = 0, t2 = 0;
    for (int c0 = kbeg; c0 < kend; c0 += 16 * U) {
        float4 av[U], wv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = c0 + 16 * u + 4 * g < kend;
            const long off = FULL == 1 ? (long)(c0 + 32 * (u >> 1)) + (long)(u & 1) * 8 * K : (long)(c0 + 16 * u);
            av[u] = ok ? *reinterpret_cast<const float4*>(ap + off) : make_float4(0, 0, 0, 0);
            wv[u] = ok ? *reinterpret_cast<const float4*>(wp + off) : make_float4(0, 0, 0, 0);
        }
        if (c0 == kbeg) t1 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_waitcnt(0);
        if (c0 == kbeg) t2 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, wv[u].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, wv[u].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, wv[u].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, wv[u].w, acc1, 0, 0, 0);
        }
    }
    f32x4 s = acc0 + acc1;
    unsigned long long t3 = __builtin_amdgcn_s_memtime();
    *reinterpret_cast<f32x4*>(&red[(wave * 64 + lane) * 4]) = s;
    __syncthreads();
    unsigned long long t4 = __builtin_amdgcn_s_memtime();
    if (wave == 0) {
        f32x4 tot = {0, 0, 0, 0};
        for (int w = 0; w < WAVES; ++w) tot += *reinterpret_cast<const f32x4*>(&red[(w * 64 + lane) * 4]);
        for (int i = 0; i < 4; ++i) {
            int m = m0 + 4 * g + i, col = nb + r;
            if (m < M && col < N) out[(long)m * N + col] = tot[i];
        }
    }
    unsigned long long t5 = __builtin_amdgcn_s_memtime();
    if (do_stamp && lane == 0) {
        unsigned long long* p = stamps + ((long)(blockIdx.y * gridDim.x + blockIdx.x) * WAVES + wave) * 6;
        p[0] = t0; p[1] = t1; p[2] = t2; p[3] = t3; p[4] = t4; p[5] = t5;
    }
}

template <int WAVES, int U, int FULL = 0>
void run(int M, int N, int K) {
    float *A, *W, *out; unsigned long long* st;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&out, (size_t)M * N * 4);
    dim3 grid((N + 15) / 16, (M + 15) / 16);
    const int nw = grid.x * grid.y * WAVES;
    hipMalloc(&st, (size_t)nw * 6 * 8);
    hipMemset(A, 0, (size_t)M * K * 4); hipMemset(W, 0, (size_t)N * K * 4);
    hipStream_t s; hipStreamCreate(&s);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((probe<WAVES, U, FULL>), grid, dim3(WAVES * 64), 0, s, A, W, out, M, N, K, st, 0);
    hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((probe<WAVES, U, FULL>), grid, dim3(WAVES * 64), 0, s, A, W, out, M, N, K, st, 0);
    hipEventRecord(e1, s); hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipLaunchKernelGGL((probe<WAVES, U, FULL>), grid, dim3(WAVES * 64), 0, s, A, W, out, M, N, K, st, 1);
    hipStreamSynchronize(s);
    std::vector<unsigned long long> h((size_t)nw * 6);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int i = 0; i < nw; ++i) { tmin = std::min(tmin, h[i * 6]); tmax = std::max(tmax, h[i * 6 + 5]); }
    double d[5] = {0, 0, 0, 0, 0}; double start_spread = 0;
    for (int i = 0; i < nw; ++i) {
        for (int j = 0; j < 5; ++j) d[j] += (double)(h[i * 6 + j + 1] - h[i * 6 + j]);
        start_spread = std::max(start_spread, (double)(h[i * 6] - tmin));
    }
    // s_memtime ticks at 100 MHz (constant clock) on gfx9: report in ns
    printf("%sM=%d N=%d K=%d WAVES=%d U=%d: %.2f us/launch (eager stream) | WGs=%d | first->last start %.0f ns | total span %.0f ns | "
           "avg per-wave ns: issue %.0f, wait %.0f, mfma %.0f, barrier %.0f, epilogue %.0f\n",
           FULL == 1 ? "[8x128B] " : FULL == 2 ? "[quad-contiguous 16x64B] " : "", M, N, K, WAVES, U, ms * 1000 / 200, grid.x * grid.y, start_spread * 10, (double)(tmax - tmin) * 10,
           d[0] / nw * 10, d[1] / nw * 10, d[2] / nw * 10, d[3] / nw * 10, d[4] / nw * 10);
    hipFree(A); hipFree(W); hipFree(out); hipFree(st);
}

int main() {
    run<16, 10, 1>(64, 512, 2560);
    run<16, 10, 2>(64, 512, 2560);
    run<16, 4, 2>(64, 512, 2560);
    run<8, 4, 1>(64, 512, 512);
    run<8, 4, 2>(64, 512, 512);
    run<8, 8, 2>(64, 2560, 512);
    run<16, 10>(64, 512, 2560);
    run<16, 4>(64, 512, 2560);
    run<8, 8>(64, 512, 2560);
    run<4, 10>(64, 512, 2560);
    run<8, 8>(64, 2560, 512);
    run<8, 8>(64, 512, 512);
    run<4, 8>(64, 512, 512);
    run<8, 8>(64, 512, 1024);
    return 0;
}
