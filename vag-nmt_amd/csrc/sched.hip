// Compute-unit partitioned streams for the overlapped training step.
//
// The recurrences are chains of ~400 dependent launches of 128-256 workgroups each that leave most of the chip's issue
// slots idle; the dense work around them (output head, weight-gradient products) is independent of the chain for most of
// its length.  Two plain streams do not help (measured, profiles/r02_exp_overlap.txt: a 128x128 product block holds half
// a CU's registers for 20-70 us and the chain's 1024-thread workgroups wait for whole blocks to retire), but two streams
// created with DISJOINT CU MASKS do (profiles/r02_exp_cumask.txt: chain on 160 CUs 0.558 ms -- unchanged --, six
// weight-gradient products on the other 96 CUs 0.46 ms, together 0.61 ms against 0.86 ms in series).  This file owns the
// two masked streams (per calling thread, created on first use, alive for the life of the process) and a ring of events
// for the hand-offs between them and the caller's stream.
//
// A masked stream keeps its mask inside a captured graph only when that graph is linear and replayed on the stream itself;
// a captured fork/join is replayed on unmasked internal streams.  The overlapped step is therefore launched eagerly -- the
// host enqueues a step's ~500 launches in well under the step's GPU time (measured: eager 4.28 ms vs graph 4.27 ms).
#include "kernels.h"
#include <cstdlib>

namespace {

constexpr int kEvents = 256;

struct Sched {
    bool ready = false, failed = false;
    int chain_cus = 0, total_cus = 0;
    hipStream_t chain = nullptr, side = nullptr;
    hipEvent_t ev[kEvents];
    int next = 0;
};
thread_local Sched g_sched;

int sched_init() {
    Sched& z = g_sched;
    if (z.ready) return VAG_OK;
    if (z.failed) return VAG_EINVAL;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { z.failed = true; return VAG_EINVAL; }
    z.total_cus = prop.multiProcessorCount;
    // split: 5/8 of the CUs for the chains (160 of 256: no chain kernel of the headline shape has more than 128 workgroups
    // of 1024 threads or 256 of 512, and none got slower on 160), 3/8 for the side work.  VAG_OVERLAP_CUS overrides.
    int nc = z.total_cus * 5 / 8;
    if (const char* e = getenv("VAG_OVERLAP_CUS")) nc = atoi(e);
    if (nc < 8 || nc > z.total_cus - 8) { z.failed = true; return VAG_EINVAL; }
    z.chain_cus = nc;
    const int words = (z.total_cus + 31) / 32;
    uint32_t ma[32] = {0}, mb[32] = {0};
    if (words > 32) { z.failed = true; return VAG_EINVAL; }
    // bit i = CU i in the driver's numbering, which deals consecutive bits out over the XCDs: both streams get an equal
    // share of every XCD (and of every XCD's L2)
    for (int i = 0; i < z.total_cus; ++i) (i < nc ? ma : mb)[i / 32] |= 1u << (i % 32);
    if (getenv("VAG_OVERLAP_NOMASK")) {       // experiment: the same schedule on two ordinary streams
        if (hipStreamCreateWithFlags(&z.chain, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&z.side, hipStreamNonBlocking) != hipSuccess) { z.failed = true; return VAG_EINVAL; }
    } else if (hipExtStreamCreateWithCUMask(&z.chain, (uint32_t)words, ma) != hipSuccess ||
               hipExtStreamCreateWithCUMask(&z.side, (uint32_t)words, mb) != hipSuccess) {
        z.failed = true;
        return VAG_EINVAL;
    }
    for (int i = 0; i < kEvents; ++i)
        if (hipEventCreateWithFlags(&z.ev[i], hipEventDisableTiming) != hipSuccess) { z.failed = true; return VAG_EINVAL; }
    z.ready = true;
    return VAG_OK;
}

}  // namespace

int vag_sched_streams(hipStream_t* chain, hipStream_t* side) {
    VAG_TRY(sched_init());
    *chain = g_sched.chain;
    *side = g_sched.side;
    return VAG_OK;
}

// Everything enqueued on `from` so far happens before anything enqueued on `to` from now on.  The events come from a
// ring: a wait refers to the record that precedes it, so re-recording an event later does not disturb earlier waits.
int vag_sched_order(hipStream_t from, hipStream_t to) {
    if (from == to) return VAG_OK;
    VAG_TRY(sched_init());
    Sched& z = g_sched;
    hipEvent_t e = z.ev[z.next];
    z.next = (z.next + 1) % kEvents;
    hipError_t r = hipEventRecord(e, from);
    if (r != hipSuccess) return (int)r;
    r = hipStreamWaitEvent(to, e, 0);
    return r == hipSuccess ? VAG_OK : (int)r;
}

// A point on `from` that several later waits may refer to.
int vag_sched_mark(hipStream_t from, hipEvent_t* out) {
    VAG_TRY(sched_init());
    Sched& z = g_sched;
    hipEvent_t e = z.ev[z.next];
    z.next = (z.next + 1) % kEvents;
    hipError_t r = hipEventRecord(e, from);
    if (r != hipSuccess) return (int)r;
    *out = e;
    return VAG_OK;
}
int vag_sched_wait(hipStream_t to, hipEvent_t e) {
    hipError_t r = hipStreamWaitEvent(to, e, 0);
    return r == hipSuccess ? VAG_OK : (int)r;
}

// ---- diagnostics: VAG_OVERLAP_TRACE=1 prints, per call, when each marked point of the three streams was reached ----
#include <vector>
#include <string>
#include <cstdio>
namespace {
struct TracePt { std::string label; hipEvent_t ev; };
thread_local std::vector<TracePt> g_trace;
thread_local std::vector<hipEvent_t> g_trace_pool;
}
bool vag_sched_tracing() {
    static const bool on = getenv("VAG_OVERLAP_TRACE") != nullptr;
    return on;
}
void vag_sched_trace(hipStream_t s, const char* label, int64_t idx) {
    if (!vag_sched_tracing()) return;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return;      // timing events cannot be captured
    hipEvent_t e;
    if (!g_trace_pool.empty()) { e = g_trace_pool.back(); g_trace_pool.pop_back(); }
    else if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, s);
    char buf[96];
    snprintf(buf, sizeof buf, idx >= 0 ? "%s[%lld]" : "%s", label, (long long)idx);
    g_trace.push_back({buf, e});
}
void vag_sched_trace_dump() {
    if (!vag_sched_tracing() || g_trace.empty()) return;
    (void)hipDeviceSynchronize();
    static thread_local int calls = 0;
    const bool print = (++calls % 20) == 10;
    for (size_t i = 0; i < g_trace.size(); ++i) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, g_trace[0].ev, g_trace[i].ev);
        if (print) fprintf(stderr, "[trace] %9.1f us  %s\n", ms * 1e3, g_trace[i].label.c_str());
    }
    for (auto& p : g_trace) g_trace_pool.push_back(p.ev);
    g_trace.clear();
}

static thread_local const VagLoopHooks* g_hooks = nullptr;
void vag_set_loop_hooks(const VagLoopHooks* h) { g_hooks = h; }
const VagLoopHooks* vag_loop_hooks() { return g_hooks; }
