// Bandwidth-bound helper kernels: embedding gather / scatter-add, GRU cell backward (elementwise part),
// dropout, mean-pool.  All accesses are 16-byte vectors where the layout allows (E, H, C are multiples of 4).
#include "kernels.h"

static inline dim3 grid1d(int64_t n, int bs = 256) {
    int64_t b = cdiv64(n, bs);
    if (b > 65535ll * 16) b = 65535ll * 16;
    return dim3((unsigned)(b < 1 ? 1 : b));
}

// ------------------------------------------------------------------ embedding
__global__ __launch_bounds__(256) void embed_gather_kernel(const int64_t* __restrict__ idx, int64_t ist, int64_t isb,
                                                           int T, int B, const float* __restrict__ W, int E,
                                                           float* __restrict__ out, const uint64_t* rng, int sid,
                                                           float p, float* __restrict__ mask_out) {
    const int E4 = E >> 2;
    const int64_t total = (int64_t)T * B * E4;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / E4;
        const int e = (int)(i - row * E4) << 2;
        const int t = (int)(row / B), b = (int)(row - (int64_t)t * B);
        const int64_t tok = idx[b * isb + t * ist];
        if (mask_out && e == 0) mask_out[b * isb + t * ist] = tok != 0 ? 1.f : 0.f;     // the source mask (V11.py:100) on the way
        float4 v = *reinterpret_cast<const float4*>(W + tok * E + e);
        if (rng && p > 0.f) {
            const uint64_t o = (uint64_t)row * E + e;
            v.x *= vag_drop_mul(rng, sid, o + 0, p);
            v.y *= vag_drop_mul(rng, sid, o + 1, p);
            v.z *= vag_drop_mul(rng, sid, o + 2, p);
            v.w *= vag_drop_mul(rng, sid, o + 3, p);
        }
        *reinterpret_cast<float4*>(out + row * E + e) = v;
    }
}

int vag_embed_gather_launch(const int64_t* idx, int64_t ist, int64_t isb, int64_t T, int64_t B, const float* W,
                            int64_t E, float* out, const uint64_t* rng, int sid, float p, hipStream_t s, float* mask_out) {
    VAG_CHECK_ARG(idx && W && out && E > 0 && E % 4 == 0 && T >= 0 && B >= 0);
    if (T * B == 0) return VAG_OK;
    hipLaunchKernelGGL(embed_gather_kernel, grid1d(T * B * E / 4), dim3(256), 0, s, idx, ist, isb, (int)T, (int)B, W,
                       (int)E, out, rng, sid, p, mask_out);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

__global__ __launch_bounds__(256) void embed_scatter_kernel(const int64_t* __restrict__ idx, int64_t ist, int64_t isb,
                                                            int T, int B, const float* __restrict__ g, int E,
                                                            float* __restrict__ gW, const uint64_t* rng, int sid,
                                                            float p, unsigned* poison, int consume) {
    // a persistent recurrence of this step gave up a wait (persist.hip: g_persist_poison): its gradient is void.  The padding
    // row's first entry (never touched otherwise) becomes non-finite, so that the norm pass -- after the all-reduce, on every
    // replica -- sees it and the optimiser skips the step (optim.hip)
    // `consume`: the flag is the PROCESS-WIDE one (a step without a guard pair of its own, vag_step_cfg.guard == NULL): nothing else
    // ever clears that word -- adam_prep_kernel resets only its driver's pair -- so the step that turns it into a skipped update takes
    // it down as well; one transient give-up anywhere in the process must not void every later unguarded step.
    if (poison && blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned f = consume ? atomicExch(poison, 0u) : __hip_atomic_load(poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (f != 0u) gW[0] = __builtin_inff();
    }
    // One WAVE per (row, 64-float column block): its lanes add 64 CONSECUTIVE floats, i.e. every atomic wave-instruction covers
    // 256 contiguous bytes -- the shape float atomics run at full rate for (MI355X_MICROARCH.md, global float atomics).  Round 3
    // gave every lane a float4 and issued four atomics of one dword every 16 bytes: a quarter of each 64-byte atomic request
    // carried data (16 us for the decoder's 2560 x 256 rows; now ~5).
    const int EB = (E + 63) >> 6;                                  // 64-float blocks per row
    const int64_t total = (int64_t)T * B * EB;                     // wave tasks
    const int lane = threadIdx.x & 63;
    for (int64_t w = blockIdx.x * 4ll + (threadIdx.x >> 6); w < total; w += (int64_t)gridDim.x * 4) {
        const int64_t row = w / EB;
        const int e = (int)(w - row * EB) * 64 + lane;
        const int t = (int)(row / B), b = (int)(row - (int64_t)t * B);
        const int64_t tok = idx[b * isb + t * ist];
        if (tok == 0 || e >= E) continue;   // padding_idx: no gradient
        float v = g[row * E + e];
        if (rng && p > 0.f) v *= vag_drop_mul(rng, sid, (uint64_t)row * E + e, p);
        atomicAdd(gW + tok * E + e, v);
    }
}

int vag_embed_scatter_launch(const int64_t* idx, int64_t ist, int64_t isb, int64_t T, int64_t B, const float* g,
                             int64_t E, float* gW, const uint64_t* rng, int sid, float p, hipStream_t s,
                             const unsigned* poison) {
    VAG_CHECK_ARG(idx && g && gW && E > 0 && E % 4 == 0 && T >= 0 && B >= 0);
    if (T * B == 0) return VAG_OK;
    hipLaunchKernelGGL(embed_scatter_kernel, grid1d(T * B * ((E + 63) / 64) * 64), dim3(256), 0, s, idx, ist, isb, (int)T, (int)B, g,
                       (int)E, gW, rng, sid, p, const_cast<unsigned*>(poison), (poison && vag_persist_guard_peek() == nullptr) ? 1 : 0);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// ------------------------------------------------------------------ GRU cell backward (elementwise part)
// Given dh' and the saved gates:  dn = dh'(1-z); dz = dh'(h-n); dh_direct = dh' z;
// dn_pre = dn(1-n^2); dr = dn_pre*hn; dr_pre = dr r(1-r); dz_pre = dz z(1-z);
// dgi = [dr_pre, dz_pre, dn_pre];  dgh = [dr_pre, dz_pre, dn_pre*r].
__global__ __launch_bounds__(256) void gru_bwd_elem_kernel(GruBwdArgs a) {
    const GruBwdSide& sd = a.s[blockIdx.y];
    const int H = a.H;
    const int64_t total = (int64_t)a.M * H;
    const int64_t MH = total;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int m = (int)(i / H), j = (int)(i - (int64_t)m * H);
        const bool active = a.lengths ? (sd.t < a.lengths[m]) : true;
        float dh = sd.dh_carry ? sd.dh_carry[i] : 0.f;
        float* gi = sd.dgi + (int64_t)m * a.ldgi + j;
        float* gh = sd.dgh + (int64_t)m * a.ldgh + j;
        if (!active) {
            gi[0] = 0.f; gi[H] = 0.f; gi[2 * H] = 0.f;
            gh[0] = 0.f; gh[H] = 0.f; gh[2 * H] = 0.f;
            sd.dh_prev[i] = dh;      // state was carried through unchanged
            continue;
        }
        if (sd.dh_add) {
            float e = sd.dh_add[(int64_t)m * a.ld_add + j];
            if (a.rng && a.p > 0.f) e *= vag_drop_mul(a.rng, a.sid, (uint64_t)m * a.ld_add + sd.drop_idx0 + j, a.p);
            dh += e;
        }
        const float r = sd.save[i], z = sd.save[MH + i], n = sd.save[2 * MH + i], hn = sd.save[3 * MH + i];
        const float hp = sd.hprev[(int64_t)m * a.ldh + j];
        const float dn_pre = dh * (1.f - z) * (1.f - n * n);
        const float dz_pre = dh * (hp - n) * z * (1.f - z);
        const float dr_pre = dn_pre * hn * r * (1.f - r);
        gi[0] = dr_pre; gi[H] = dz_pre; gi[2 * H] = dn_pre;
        gh[0] = dr_pre; gh[H] = dz_pre; gh[2 * H] = dn_pre * r;
        sd.dh_prev[i] = dh * z;
    }
}

int vag_gru_bwd_elem_launch(const GruBwdArgs& a, int nz, hipStream_t s) {
    VAG_CHECK_ARG(a.M > 0 && a.H > 0 && (nz == 1 || nz == 2));
    dim3 grid(grid1d((int64_t)a.M * a.H).x, (unsigned)nz);
    hipLaunchKernelGGL(gru_bwd_elem_kernel, grid, dim3(256), 0, s, a);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// ------------------------------------------------------------------ small elementwise kernels
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                       float* __restrict__ dx, int64_t n, const uint64_t* rng,
                                                       int sid, float p, int64_t idx0) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float yy = y[i];
        float d = dy[i];
        if (rng && p > 0.f) {
            const float mlt = vag_drop_mul(rng, sid, (uint64_t)(idx0 + i), p);
            d *= mlt;
            yy = mlt > 0.f ? yy / mlt : 0.f;     // y holds tanh(.)*mul; undo the scale for kept elements
        }
        dx[i] = d * (1.f - yy * yy);
    }
}
int vag_tanh_bwd_launch(const float* y, const float* dy, float* dx, int64_t n, const uint64_t* rng, int sid, float p,
                        hipStream_t s, int64_t idx0) {
    if (n == 0) return VAG_OK;
    hipLaunchKernelGGL(tanh_bwd_kernel, grid1d(n), dim3(256), 0, s, y, dy, dx, n, rng, sid, p, idx0);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

__global__ __launch_bounds__(256) void dropout_apply_kernel(float* __restrict__ x, int64_t n, int64_t idx0,
                                                            const uint64_t* rng, int sid, float p,
                                                            float* __restrict__ mask_out) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float mlt = vag_drop_mul(rng, sid, (uint64_t)(idx0 + i), p);
        if (mask_out) mask_out[i] = mlt;
        else x[i] *= mlt;
    }
}
int vag_dropout_apply_launch(float* x, int64_t n, int64_t idx0, const uint64_t* rng, int sid, float p, hipStream_t s) {
    if (n == 0 || !rng || p <= 0.f) return VAG_OK;
    hipLaunchKernelGGL(dropout_apply_kernel, grid1d(n), dim3(256), 0, s, x, n, idx0, rng, sid, p, (float*)nullptr);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}
// x = tanh(x) * dropout multiplier (the head's pre-activation, accumulated by three grouped products)
__global__ __launch_bounds__(256) void tanh_dropout_kernel(float* __restrict__ x, int64_t n, int64_t idx0,
                                                           const uint64_t* rng, int sid, float p) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        x[i] = vag_tanh(x[i]) * vag_drop_mul(rng, sid, (uint64_t)(idx0 + i), p);
}
int vag_tanh_dropout_launch(float* x, int64_t n, int64_t idx0, const uint64_t* rng, int sid, float p, hipStream_t s) {
    if (n == 0) return VAG_OK;
    hipLaunchKernelGGL(tanh_dropout_kernel, grid1d(n), dim3(256), 0, s, x, n, idx0, rng, sid, p);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}
int vag_dropout_mask_launch(const uint64_t* rng, int sid, int64_t n, float p, float* out, hipStream_t s) {
    VAG_CHECK_ARG(out && n >= 0);
    if (n == 0) return VAG_OK;
    hipLaunchKernelGGL(dropout_apply_kernel, grid1d(n), dim3(256), 0, s, (float*)nullptr, n, (int64_t)0, rng, sid, p, out);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

__global__ __launch_bounds__(256) void axpy_kernel(float a, const float* __restrict__ x, float* __restrict__ y,
                                                   int64_t n, int acc) {
    // acc: 0 -> y = a*x, 1 -> y += a*x, 2 -> y = 0 (fill; x is not read)
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        y[i] = acc == 2 ? 0.f : (acc ? y[i] + a * x[i] : a * x[i]);
}
int vag_axpy_launch(float a, const float* x, float* y, int64_t n, int accumulate, hipStream_t s) {
    if (n == 0) return VAG_OK;
    hipLaunchKernelGGL(axpy_kernel, grid1d(n), dim3(256), 0, s, a, x, y, n, accumulate);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

__global__ __launch_bounds__(256) void src_mask_kernel(const int64_t* __restrict__ src, int64_t n,
                                                       float* __restrict__ mask) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        mask[i] = src[i] != 0 ? 1.f : 0.f;
}
int vag_src_mask_launch(const int64_t* src, int64_t n, float* mask, hipStream_t s) {
    if (n == 0) return VAG_OK;
    hipLaunchKernelGGL(src_mask_kernel, grid1d(n), dim3(256), 0, s, src, n, mask);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// ------------------------------------------------------------------ mean-pool (+ mix with the attended context)
__global__ __launch_bounds__(256) void meanpool_mix_kernel(const float* __restrict__ enc, const float* __restrict__ mask,
                                                           const float* __restrict__ ctx, float split, int Ts, int C,
                                                           float* __restrict__ xmix) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float cnt = 0.f;
    for (int t = 0; t < Ts; ++t) cnt += mask[(int64_t)b * Ts + t];
    float s = 0.f;
    const float* e = enc + (int64_t)b * Ts * C + c;
    for (int t = 0; t < Ts; ++t) s += e[(int64_t)t * C];
    const float mean = s / cnt;
    xmix[(int64_t)b * C + c] = ctx ? split * ctx[(int64_t)b * C + c] + (1.f - split) * mean : mean;
}
int vag_meanpool_mix_launch(const float* enc, const float* mask, const float* ctx, float split, int64_t B, int64_t Ts,
                            int64_t C, float* xmix, hipStream_t s) {
    VAG_CHECK_ARG(enc && mask && xmix && B > 0 && Ts > 0 && C > 0);
    dim3 grid((unsigned)cdiv64(C, 256), (unsigned)B);
    hipLaunchKernelGGL(meanpool_mix_kernel, grid, dim3(256), 0, s, enc, mask, ctx, split, (int)Ts, (int)C, xmix);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

__global__ __launch_bounds__(256) void meanpool_bwd_kernel(const float* __restrict__ mask, const float* __restrict__ dx,
                                                           float coef, int Ts, int C, float* __restrict__ d_enc,
                                                           int acc) {
    // blockIdx.z = a chunk of 4 positions, all of a thread's read-modify-writes in flight together (round 3: one thread walked
    // all Ts positions of its column one after the other: 15 us for 21 MB)
    const int b = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float cnt = 0.f;
    for (int t = 0; t < Ts; ++t) cnt += mask[(int64_t)b * Ts + t];
    const float v = coef * dx[(int64_t)b * C + c] / cnt;
    float* d = d_enc + (int64_t)b * Ts * C + c;
    const int t0 = blockIdx.z * 4;
    float old[4] = {0.f, 0.f, 0.f, 0.f};
    if (acc) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (t0 + i < Ts) old[i] = d[(int64_t)(t0 + i) * C];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (t0 + i < Ts) d[(int64_t)(t0 + i) * C] = old[i] + v;
}
int vag_meanpool_bwd_launch(const float* mask, const float* dx, float coef, int64_t B, int64_t Ts, int64_t C,
                            float* d_enc, int accumulate, hipStream_t s) {
    VAG_CHECK_ARG(mask && dx && d_enc && B > 0 && Ts > 0 && C > 0);
    if (vag_rmw_defer_meanpool(mask, dx, coef, B, Ts, C, d_enc, accumulate)) return VAG_OK;      // attn.hip: held back for one merged pass
    dim3 grid((unsigned)cdiv64(C, 256), (unsigned)B, (unsigned)cdiv64(Ts, 4));
    hipLaunchKernelGGL(meanpool_bwd_kernel, grid, dim3(256), 0, s, mask, dx, coef, (int)Ts, (int)C, d_enc, accumulate);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

__global__ void rng_advance_kernel(uint64_t* rng) {
    if (threadIdx.x == 0 && blockIdx.x == 0) rng[1] += 1;
}
int vag_rng_advance_launch(uint64_t* rng, hipStream_t s) {
    VAG_CHECK_ARG(rng);
    hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(64), 0, s, rng);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

__global__ __launch_bounds__(256) void copy2d_kernel(const float* __restrict__ in, int64_t ldi, float* __restrict__ out,
                                                     int64_t ldo, int64_t rows, int64_t cols) {
    const int64_t total = rows * cols;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols, c = i - r * cols;
        out[r * ldo + c] = in[r * ldi + c];
    }
}
int vag_copy2d_launch(const float* in, int64_t ldi, float* out, int64_t ldo, int64_t rows, int64_t cols, hipStream_t s) {
    if (rows * cols == 0) return VAG_OK;
    hipLaunchKernelGGL(copy2d_kernel, grid1d(rows * cols), dim3(256), 0, s, in, ldi, out, ldo, rows, cols);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}

// out[i, 0:w] = in[idx[i], 0:w]   (int64 token matrices; batch assembly from the device-resident corpus)
__global__ __launch_bounds__(256) void gather_rows_i64_kernel(const int64_t* __restrict__ in, int64_t ld,
                                                              const int64_t* __restrict__ idx, int64_t rows, int64_t w,
                                                              int64_t* __restrict__ out) {
    const int64_t total = rows * w;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / w, c = i - r * w;
        out[i] = in[idx[r] * ld + c];
    }
}
int vag_gather_rows_i64_launch(const int64_t* in, int64_t ld, const int64_t* idx, int64_t rows, int64_t w, int64_t* out,
                               hipStream_t s) {
    VAG_CHECK_ARG(in && idx && out && rows >= 0 && w >= 0 && w <= ld);
    if (rows * w == 0) return VAG_OK;
    hipLaunchKernelGGL(gather_rows_i64_kernel, grid1d(rows * w), dim3(256), 0, s, in, ld, idx, rows, w, out);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}


// ------------------------------------------------------------------ several small fills / copies / transposes, one launch
// Each job covers a (rows x cols) fp32 matrix: kind 0 zero-fill, 1 copy, 2 transpose (dst[c*ld_dst + r] = src[r*ld_src + c]),
// 3 / 4 = copy / transpose into an fp16 destination (ld_dst in halves).
// A block handles one 32x32 tile of one job.  Used for the per-optimiser-step derived weights and the zero rows of the
// recurrent state buffers: a handful of tiny dependent-free operations that would otherwise be a launch each.
__global__ __launch_bounds__(256) void jobs_kernel(VagJobs J) {
    __shared__ float tile[32][33];
    int k = 0;
    while (k + 1 < J.n && (int)blockIdx.x >= J.start[k + 1]) ++k;
    const VagJob& j = J.j[k];
    const int id = blockIdx.x - J.start[k];
    const int tc = (int)((j.cols + 31) / 32);
    const int64_t r0 = (int64_t)(id / tc) * 32, c0 = (int64_t)(id % tc) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const bool to_half = j.kind >= 3;                            // kinds 3 / 4: copy / transpose into an fp16 destination
    vag_half* dh = reinterpret_cast<vag_half*>(j.dst);
    if (j.kind != 2 && j.kind != 4) {
#pragma unroll
        for (int i = 0; i < 32; i += 8) {
            const int64_t r = r0 + ty + i, c = c0 + tx;
            if (r < j.rows && c < j.cols) {
                const float v = j.kind == 0 ? 0.f : j.src[r * j.ld_src + c];
                if (to_half) dh[r * j.ld_dst + c] = (vag_half)v;
                else j.dst[r * j.ld_dst + c] = v;
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int64_t r = r0 + ty + i, c = c0 + tx;
        if (r < j.rows && c < j.cols) tile[ty + i][tx] = j.src[r * j.ld_src + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int64_t c = c0 + ty + i, r = r0 + tx;
        if (r < j.rows && c < j.cols) {
            if (to_half) dh[c * j.ld_dst + r] = (vag_half)tile[tx][ty + i];
            else j.dst[c * j.ld_dst + r] = tile[tx][ty + i];
        }
    }
}
int vag_jobs_launch(const VagJob* jobs, int n, hipStream_t s) {
    VAG_CHECK_ARG(jobs && n >= 1 && n <= VAG_MAXJOBS);
    VagJobs J;
    J.n = n;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        VAG_CHECK_ARG(jobs[i].dst && (jobs[i].kind == 0 || jobs[i].src) && jobs[i].rows >= 0 && jobs[i].cols >= 0);
        J.j[i] = jobs[i];
        J.start[i] = total;
        total += (int)(cdiv64(jobs[i].rows, 32) * cdiv64(jobs[i].cols, 32));
    }
    J.start[n] = total;
    if (total == 0) return VAG_OK;
    hipLaunchKernelGGL(jobs_kernel, dim3((unsigned)total), dim3(256), 0, s, J);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}


// ------------------------------------------------------------------ up to four contiguous byte ranges, one launch
// (a batch's src / lengths / tgt / image rows into the step driver's static buffers: one launch instead of four copies)
struct Copy4 { const unsigned char* s[4]; unsigned char* d[4]; int64_t n[4]; };
__global__ __launch_bounds__(256) void copy4_kernel(Copy4 c) {
    const int k = blockIdx.y;
    const int64_t n = c.n[k];
    const unsigned char* s = c.s[k];
    unsigned char* d = c.d[k];
    if (((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15) == 0) {
        const int64_t n16 = n >> 4;
        const uint4* s4 = reinterpret_cast<const uint4*>(s);
        uint4* d4 = reinterpret_cast<uint4*>(d);
        // a workgroup moves contiguous 16 KB pieces (256 threads x 4 x 16 bytes, four loads in flight per thread), pieces dealt
        // round-robin over the grid; non-temporal both ways: every byte is touched once (round 2's four grid-strided streams
        // per thread, default cache policy: 4.4 TB/s of the 6.3 TB/s a float4 copy reaches on this part)
        typedef unsigned u4v __attribute__((ext_vector_type(4)));
        const u4v* sv = reinterpret_cast<const u4v*>(s);
        u4v* dv = reinterpret_cast<u4v*>(d);
        const int64_t pieces = n16 >> 10;
        for (int64_t pc = blockIdx.x; pc < pieces; pc += gridDim.x) {
            const int64_t i = (pc << 10) + threadIdx.x;
            const u4v a = __builtin_nontemporal_load(sv + i), b = __builtin_nontemporal_load(sv + i + 256);
            const u4v c2 = __builtin_nontemporal_load(sv + i + 512), e = __builtin_nontemporal_load(sv + i + 768);
            __builtin_nontemporal_store(a, dv + i); __builtin_nontemporal_store(b, dv + i + 256);
            __builtin_nontemporal_store(c2, dv + i + 512); __builtin_nontemporal_store(e, dv + i + 768);
        }
        const int64_t stride = (int64_t)gridDim.x * 256;
        for (int64_t i = (pieces << 10) + blockIdx.x * 256ll + threadIdx.x; i < n16; i += stride) d4[i] = s4[i];
        for (int64_t j = (n16 << 4) + blockIdx.x * 256ll + threadIdx.x; j < n; j += stride) d[j] = s[j];
    } else {
        for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) d[i] = s[i];
    }
}
int vag_copy4_launch(const void* const* src, void* const* dst, const int64_t* bytes, int n, hipStream_t s) {
    VAG_CHECK_ARG(src && dst && bytes && n >= 1 && n <= 4);
    Copy4 c;
    int64_t mx = 0;
    for (int i = 0; i < 4; ++i) {
        c.s[i] = i < n ? reinterpret_cast<const unsigned char*>(src[i]) : nullptr;
        c.d[i] = i < n ? reinterpret_cast<unsigned char*>(dst[i]) : nullptr;
        c.n[i] = i < n ? bytes[i] : 0;
        VAG_CHECK_ARG(c.n[i] >= 0 && (c.n[i] == 0 || (c.s[i] && c.d[i])));
        if (c.n[i] > mx) mx = c.n[i];
    }
    if (mx == 0) return VAG_OK;
    int64_t nb = cdiv64(mx, 256 * 16 * 4);
    if (nb > 8192) nb = 8192;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(copy4_kernel, dim3((unsigned)nb, (unsigned)n), dim3(256), 0, s, c);
    VAG_LAUNCH_CHECK();
    return VAG_OK;
}
