// Probe: cost of a barrier among persistent workgroups on gfx950 -- all 256 (one per CU) against 8 independent groups of
// 32 whose members share an XCD (workgroup i runs on XCD i % 8), and a hand-off of data through memory between the phases.
// Decides whether a persistent "one batch slice per XCD" recurrence (no kernel boundaries, XCD-local exchanges) can beat
// four launches per decoder step at 1.45 us a boundary.  Build: hipcc --offload-arch=gfx950 -O3 barrier_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

// sense-reversing counter barrier; `cnt` and `gen` are in device memory, one pair per group
__device__ __forceinline__ void group_barrier(unsigned* cnt, unsigned* gen, unsigned members, unsigned& my_gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned g = my_gen + 1;
        if (atomicAdd(cnt, 1u) == members - 1) {
            atomicExch(cnt, 0u);
            __threadfence();
            __hip_atomic_store(gen, g, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != g) __builtin_amdgcn_s_sleep(1);
        }
    }
    my_gen += 1;
    __syncthreads();
}

// mode 0: one group of all workgroups; mode 1: groups = blockIdx % 8 (same XCD); mode 2: groups = blockIdx / 32 (spread over XCDs)
__global__ __launch_bounds__(512) void probe(unsigned* state, float* data, int iters, int mode, long long* cycles) {
    const unsigned nwg = gridDim.x;
    unsigned group = 0, members = nwg;
    if (mode == 1) { group = blockIdx.x % 8; members = nwg / 8; }
    if (mode == 2) { group = blockIdx.x / 32; members = 32; }
    unsigned* cnt = state + group * 64;           // separate cache lines
    unsigned* gen = state + group * 64 + 32;
    unsigned my_gen = 0;
    float acc = 0.f;
    const long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        // hand-off: every workgroup writes a value, after the barrier reads its neighbour's (same group)
        if (threadIdx.x == 0) __hip_atomic_store(data + blockIdx.x * 32, (float)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        group_barrier(cnt, gen, members, my_gen);
        const unsigned nb = mode == 1 ? (blockIdx.x + 8) % nwg : (mode == 2 ? (blockIdx.x / 32) * 32 + (blockIdx.x + 1) % 32 : (blockIdx.x + 1) % nwg);
        if (threadIdx.x == 0) acc += __hip_atomic_load(data + nb * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { cycles[blockIdx.x] = t1 - t0; data[blockIdx.x * 32 + 1] = acc; }
}

int main() {
    unsigned* state; float* data; long long* cyc;
    hipMalloc(&state, 8 * 64 * 4); hipMalloc(&data, 256 * 32 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    const char* names[3] = {"all 256 workgroups", "8 groups of 32, same XCD (id % 8)", "8 groups of 32, spread over the XCDs (id / 32)"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            hipMemset(state, 0, 8 * 64 * 4);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, state, data, iters, mode, cyc);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-52s %.3f us per barrier + hand-off (hipError %d)\n", names[mode], ms * 1e3 / iters, (int)hipGetLastError());
        }
    return 0;
}
