// SAC / TD3 head + target arithmetic and the flat-buffer optimizer tail.
//
// These are the many tiny element-wise ops and the ~10 `.item()` host syncs of the reference update
// (sac_full_length_rnn_ensembleQ.py:387,423-454; q_value_guard.py:29-38; rnn_base.py:490-491,531-532) folded
// into a few HBM-bound kernels whose scalar state (Q-guard min/max, reductions) stays on the device.
#include "resel_common.h"
#include <math.h>

namespace {
using namespace resel;

constexpr int RED_BLOCKS = 256;
constexpr float HALF_LOG_2PI = 0.9189385332046727f;

// ------------------------------------------------------------------------------------ tanh-Gaussian head
__global__ void tanh_gaussian_fwd_kernel(const float* __restrict__ out2, const float* __restrict__ noise,
                                         float* __restrict__ amean, float* __restrict__ asample, float* __restrict__ logp,
                                         int M, int A) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float lp = 0.f;
    for (int a = 0; a < A; ++a) {
        const float ls = fminf(fmaxf(out2[(int64_t)m * 2 * A + a], -20.f), 2.f);
        const float mu = out2[(int64_t)m * 2 * A + A + a];
        const float nz = noise[(int64_t)m * A + a];
        const float pre = mu + nz * expf(ls);
        // softplus(-2 pre) = max(-2pre, 0) + log1p(exp(-|2 pre|))
        const float sp = fmaxf(-2.f * pre, 0.f) + log1pf(expf(-fabsf(2.f * pre)));
        lp += -0.5f * nz * nz - (ls + HALF_LOG_2PI) - 2.f * (RESEL_LN2 - pre - sp);
        if (amean) amean[(int64_t)m * A + a] = tanhf(mu);
        asample[(int64_t)m * A + a] = tanhf(pre);
    }
    logp[m] = lp;
}

__global__ void tanh_gaussian_bwd_kernel(const float* __restrict__ out2, const float* __restrict__ noise,
                                         const float* __restrict__ d_sample, const float* __restrict__ d_logp,
                                         float* __restrict__ d_out2, int M, int A) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const float dlp = d_logp ? d_logp[m] : 0.f;
    for (int a = 0; a < A; ++a) {
        const float raw = out2[(int64_t)m * 2 * A + a];
        const float ls = fminf(fmaxf(raw, -20.f), 2.f);
        const float mu = out2[(int64_t)m * 2 * A + A + a];
        const float nz = noise[(int64_t)m * A + a];
        const float sd = expf(ls);
        const float smp = tanhf(mu + nz * sd);
        const float ds = d_sample ? d_sample[(int64_t)m * A + a] : 0.f;
        const float dpre = ds * (1.f - smp * smp) + dlp * 2.f * smp;       // d logp / d pre = 2 tanh(pre)
        const float inside = (raw >= -20.f && raw <= 2.f) ? 1.f : 0.f;      // torch.clamp passes gradient on [min, max]
        d_out2[(int64_t)m * 2 * A + a] = inside * (dpre * nz * sd - dlp);
        d_out2[(int64_t)m * 2 * A + A + a] = dpre;
    }
}

// ------------------------------------------------------------------------------------ block reductions
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// 256-thread block: combine (min, max, max2, sum) and leave the result in out[0..3] (thread 0 writes)
__device__ __forceinline__ void block_reduce4(float mn, float mx, float mx2, float sm, float* out) {
    __shared__ float s[4][4];
    mn = wave_min(mn); mx = wave_max(mx); mx2 = wave_max(mx2); sm = wave_sum(sm);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s[0][w] = mn; s[1][w] = mx; s[2][w] = mx2; s[3][w] = sm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = fminf(fminf(s[0][0], s[0][1]), fminf(s[0][2], s[0][3]));
        out[1] = fmaxf(fmaxf(s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3]));
        out[2] = fmaxf(fmaxf(s[2][0], s[2][1]), fmaxf(s[2][2], s[2][3]));
        out[3] = (s[3][0] + s[3][1]) + (s[3][2] + s[3][3]);
    }
}

// ------------------------------------------------------------------------------------ REDQ target + Q guard
__global__ __launch_bounds__(256) void target_v_kernel(const float* __restrict__ q, const int32_t* __restrict__ subset, int m,
                                                       const float* __restrict__ next_logp, const float* __restrict__ log_alpha,
                                                       float* __restrict__ v, float* __restrict__ part, int E, int M) {
    const float alpha = (next_logp && log_alpha) ? expf(log_alpha[0]) : 0.f;
    float mn = INFINITY, mx = -INFINITY;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < M; i += gridDim.x * 256) {
        float val = INFINITY;
        for (int k = 0; k < m; ++k) val = fminf(val, q[(int64_t)subset[k] * M + i]);
        if (next_logp) val -= alpha * next_logp[i];
        v[i] = val;
        mn = fminf(mn, val); mx = fmaxf(mx, val);
    }
    block_reduce4(mn, mx, 0.f, 0.f, part + 4 * blockIdx.x);
}
__global__ __launch_bounds__(256) void guard_init_kernel(const float* __restrict__ part, int nblk, float* guard) {
    float mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < nblk; i += 256) { mn = fminf(mn, part[4 * i]); mx = fmaxf(mx, part[4 * i + 1]); }
    __shared__ float res[4];
    block_reduce4(mn, mx, 0.f, 0.f, res);
    __syncthreads();
    if (threadIdx.x == 0 && guard[2] == 0.f) { guard[0] = res[0]; guard[1] = res[1]; guard[2] = 1.f; }   // q_value_guard.py:23-26
}
__global__ __launch_bounds__(256) void target_y_kernel(const float* __restrict__ v, const float* __restrict__ reward,
                                                       const float* __restrict__ done, const float* __restrict__ mask, float gamma,
                                                       const float* __restrict__ guard, float* __restrict__ target,
                                                       float* __restrict__ part, int M) {
    const bool ready = guard[2] != 0.f;              // an uninitialised guard does not clamp (its first interval is the batch's own range)
    const float lo = ready ? guard[0] : -INFINITY, hi = ready ? guard[1] : INFINITY;
    float mn = INFINITY, mx = -INFINITY, ma = 0.f, sm = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < M; i += gridDim.x * 256) {
        const float c = fminf(fmaxf(v[i], lo), hi);
        const float y = reward[i] + (1.f - done[i]) * gamma * c;
        target[i] = y;
        const float mk = mask ? mask[i] : 1.f;
        const float ym = y * mk;
        mn = fminf(mn, ym); mx = fmaxf(mx, ym); ma = fmaxf(ma, fabsf(y)); sm += mk;
    }
    block_reduce4(mn, mx, ma, sm, part + 4 * blockIdx.x);
}
// data-parallel form: the batch extrema are written out ({-min, max}: one MAX all-reduce serves both) so that every rank
// initialises / updates its guard from the extrema of the GLOBAL batch
__global__ __launch_bounds__(256) void extrema_out_kernel(const float* __restrict__ part, int nblk, float* __restrict__ ext, float* stats) {
    float mn = INFINITY, mx = -INFINITY, ma = 0.f, sm = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) {
        mn = fminf(mn, part[4 * i]); mx = fmaxf(mx, part[4 * i + 1]); ma = fmaxf(ma, part[4 * i + 2]); sm += part[4 * i + 3];
    }
    __shared__ float res[4];
    block_reduce4(mn, mx, ma, sm, res);
    __syncthreads();
    if (threadIdx.x == 0) {
        ext[0] = -res[0]; ext[1] = res[1];
        if (stats) { stats[0] = res[2]; stats[1] = res[3]; }
    }
}
// data-parallel, one collective per step: slots [world][4] = every rank's {-min v, max v, -min(y mask), max(y mask)} (each rank
// wrote its own row into a zero-filled tail of the gradient bucket; the SUM all-reduce delivered all rows everywhere).  The guard
// is initialised (first call) and updated from the extrema of the GLOBAL batch, exactly as q_value_guard.py:22-38 on one process.
__global__ void guard_apply_slots_kernel(const float* __restrict__ slots, int world, float* guard) {
    if (threadIdx.x != 0) return;
    float e[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int r = 0; r < world; ++r)
        for (int j = 0; j < 4; ++j) e[j] = fmaxf(e[j], slots[4 * r + j]);
    if (guard[2] == 0.f) { guard[0] = -e[0]; guard[1] = e[1]; guard[2] = 1.f; }
    const float decay = guard[3], bmin = -e[2], bmax = e[3];
    float gmin = fminf(guard[0], bmin), gmax = fmaxf(guard[1], bmax);
    if (decay < 1.f) {
        gmin = decay * gmin + (1.f - decay) * bmin;
        gmax = decay * gmax + (1.f - decay) * bmax;
    }
    guard[0] = gmin; guard[1] = gmax;
}
__global__ void guard_init_from_kernel(const float* __restrict__ ext, float* guard) {
    if (threadIdx.x == 0 && guard[2] == 0.f) { guard[0] = -ext[0]; guard[1] = ext[1]; guard[2] = 1.f; }
}
__global__ void guard_update_from_kernel(const float* __restrict__ ext, float* guard) {
    if (threadIdx.x == 0) {
        const float decay = guard[3], bmin = -ext[0], bmax = ext[1];
        float gmin = fminf(guard[0], bmin), gmax = fmaxf(guard[1], bmax);
        if (decay < 1.f) {
            gmin = decay * gmin + (1.f - decay) * bmin;
            gmax = decay * gmax + (1.f - decay) * bmax;
        }
        guard[0] = gmin; guard[1] = gmax;
    }
}
__global__ __launch_bounds__(256) void guard_update_kernel(const float* __restrict__ part, int nblk, float* guard, float* stats) {
    float mn = INFINITY, mx = -INFINITY, ma = 0.f, sm = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) {
        mn = fminf(mn, part[4 * i]); mx = fmaxf(mx, part[4 * i + 1]); ma = fmaxf(ma, part[4 * i + 2]); sm += part[4 * i + 3];
    }
    __shared__ float res[4];
    block_reduce4(mn, mx, ma, sm, res);
    __syncthreads();
    if (threadIdx.x == 0) {                                    // q_value_guard.py:29-38
        const float decay = guard[3];
        float gmin = fminf(guard[0], res[0]), gmax = fmaxf(guard[1], res[1]);
        if (decay < 1.f) {
            gmin = decay * gmin + (1.f - decay) * res[0];
            gmax = decay * gmax + (1.f - decay) * res[1];
        }
        guard[0] = gmin; guard[1] = gmax;
        if (stats) { stats[0] = res[2]; stats[1] = res[3]; }
    }
}

// ------------------------------------------------------------------------------------ flat optimizer tail
__global__ void soft_update_kernel(float* __restrict__ tgt, const float* __restrict__ src, float tau, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        float4 t = ld4(tgt + i);
        const float4 s = ld4(src + i);
        t.x = t.x * tau + (1.f - tau) * s.x; t.y = t.y * tau + (1.f - tau) * s.y;
        t.z = t.z * tau + (1.f - tau) * s.z; t.w = t.w * tau + (1.f - tau) * s.w;
        st4(tgt + i, t);
    } else {
        for (int64_t k = i; k < n; ++k) tgt[k] = tgt[k] * tau + (1.f - tau) * src[k];
    }
}

__global__ void adamw_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                  int64_t n, const int64_t* __restrict__ seg_end, const float* __restrict__ seg_lr,
                                  const float* __restrict__ seg_wd, int nseg, float beta1, float beta2, float eps,
                                  float bc1, float bc2_sqrt, const float* __restrict__ grad_scale, const float* __restrict__ bc_dev) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (bc_dev) { bc1 = bc_dev[0]; bc2_sqrt = bc_dev[1]; }       // captured updates: the step-dependent factors live in device memory
    int sidx = 0;
    while (sidx < nseg - 1 && i >= seg_end[sidx]) ++sidx;
    const float lr = seg_lr[sidx], wd = seg_wd[sidx];
    const float gs = grad_scale ? grad_scale[0] : 1.f;
    const float gi = g[i] * gs;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi;
}

__global__ __launch_bounds__(256) void sumsq_part_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ part) {
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += x[i] * x[i];
    __shared__ float s[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ part, int nblk, float* out) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) acc += part[i];
    __shared__ float s[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (s[0] + s[1]) + (s[2] + s[3]);
}

}  // namespace

extern "C" int resel_tanh_gaussian_fwd(const float* out2, const float* noise, float* action_mean, float* action_sample,
                                       float* logp, int M, int A, resel_stream_t stream) {
    if (!out2 || !noise || !action_sample || !logp || M <= 0 || A <= 0) return RESEL_EINVAL;
    hipLaunchKernelGGL(tanh_gaussian_fwd_kernel, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, out2, noise,
                       action_mean, action_sample, logp, M, A);
    return launch_status();
}
extern "C" int resel_tanh_gaussian_bwd(const float* out2, const float* noise, const float* d_sample, const float* d_logp,
                                       float* d_out2, int M, int A, resel_stream_t stream) {
    if (!out2 || !noise || !d_out2 || M <= 0 || A <= 0) return RESEL_EINVAL;
    hipLaunchKernelGGL(tanh_gaussian_bwd_kernel, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, out2, noise,
                       d_sample, d_logp, d_out2, M, A);
    return launch_status();
}

extern "C" size_t resel_sac_target_workspace_bytes(int M) { return ((size_t)M + 8 * RED_BLOCKS) * sizeof(float); }

extern "C" int resel_sac_target(const float* q, const int32_t* subset, int m, const float* next_logp, const float* log_alpha,
                                const float* reward, const float* done, const float* mask, float gamma, float* guard,
                                float* target, float* stats, void* workspace, int E, int M, resel_stream_t stream) {
    if (!q || !subset || m <= 0 || !reward || !done || !guard || !target || !workspace || E <= 0 || M <= 0) return RESEL_EINVAL;
    if (next_logp && !log_alpha) return RESEL_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    float* v = (float*)workspace;
    float* part1 = v + M;
    float* part2 = part1 + 4 * RED_BLOCKS;
    const int nblk = (M + 255) / 256 < RED_BLOCKS ? (M + 255) / 256 : RED_BLOCKS;
    hipLaunchKernelGGL(target_v_kernel, dim3(nblk), dim3(256), 0, s, q, subset, m, next_logp, log_alpha, v, part1, E, M);
    hipLaunchKernelGGL(guard_init_kernel, dim3(1), dim3(256), 0, s, part1, nblk, guard);
    hipLaunchKernelGGL(target_y_kernel, dim3(nblk), dim3(256), 0, s, v, reward, done, mask, gamma, guard, target, part2, M);
    hipLaunchKernelGGL(guard_update_kernel, dim3(1), dim3(256), 0, s, part2, nblk, guard, stats);
    return launch_status();
}

extern "C" int resel_sac_target_phase(int phase, const float* q, const int32_t* subset, int m, const float* next_logp, const float* log_alpha,
                                      const float* reward, const float* done, const float* mask, float gamma, float* guard,
                                      float* target, float* stats, float* extrema, void* workspace, int E, int M, resel_stream_t stream) {
    if (phase < 0 || phase > 2 || !guard || !extrema || !workspace || M <= 0) return RESEL_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    float* v = (float*)workspace;
    float* part1 = v + M;
    float* part2 = part1 + 4 * RED_BLOCKS;
    const int nblk = (M + 255) / 256 < RED_BLOCKS ? (M + 255) / 256 : RED_BLOCKS;
    if (phase == 0) {                  // v = min over the subset (- alpha log pi); extrema[0:2] = {-min v, max v} of this rank's rows
        if (!q || !subset || m <= 0 || E <= 0 || (next_logp && !log_alpha)) return RESEL_EINVAL;
        hipLaunchKernelGGL(target_v_kernel, dim3(nblk), dim3(256), 0, s, q, subset, m, next_logp, log_alpha, v, part1, E, M);
        hipLaunchKernelGGL(extrema_out_kernel, dim3(1), dim3(256), 0, s, part1, nblk, extrema, (float*)nullptr);
    } else if (phase == 1) {           // extrema[0:2] now global: first-call initialisation, clamp, y; extrema[2:4] = {-min, max} of y * mask
        if (!reward || !done || !target) return RESEL_EINVAL;
        hipLaunchKernelGGL(guard_init_from_kernel, dim3(1), dim3(64), 0, s, extrema, guard);
        hipLaunchKernelGGL(target_y_kernel, dim3(nblk), dim3(256), 0, s, v, reward, done, mask, gamma, guard, target, part2, M);
        hipLaunchKernelGGL(extrema_out_kernel, dim3(1), dim3(256), 0, s, part2, nblk, extrema + 2, stats);
    } else {                           // extrema[2:4] now global: running min / max update
        hipLaunchKernelGGL(guard_update_from_kernel, dim3(1), dim3(64), 0, s, extrema + 2, guard);
    }
    return launch_status();
}

extern "C" int resel_sac_target_local(const float* q, const int32_t* subset, int m, const float* next_logp, const float* log_alpha,
                                      const float* reward, const float* done, const float* mask, float gamma, const float* guard,
                                      float* target, float* stats, float* extrema, void* workspace, int E, int M, resel_stream_t stream) {
    if (!q || !subset || m <= 0 || !reward || !done || !guard || !target || !extrema || !workspace || E <= 0 || M <= 0) return RESEL_EINVAL;
    if (next_logp && !log_alpha) return RESEL_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    float* v = (float*)workspace;
    float* part1 = v + M;
    float* part2 = part1 + 4 * RED_BLOCKS;
    const int nblk = (M + 255) / 256 < RED_BLOCKS ? (M + 255) / 256 : RED_BLOCKS;
    hipLaunchKernelGGL(target_v_kernel, dim3(nblk), dim3(256), 0, s, q, subset, m, next_logp, log_alpha, v, part1, E, M);
    hipLaunchKernelGGL(extrema_out_kernel, dim3(1), dim3(256), 0, s, part1, nblk, extrema, (float*)nullptr);
    hipLaunchKernelGGL(target_y_kernel, dim3(nblk), dim3(256), 0, s, v, reward, done, mask, gamma, guard, target, part2, M);
    hipLaunchKernelGGL(extrema_out_kernel, dim3(1), dim3(256), 0, s, part2, nblk, extrema + 2, stats);
    return launch_status();
}

extern "C" int resel_guard_apply_slots(const float* slots, int world, float* guard, resel_stream_t stream) {
    if (!slots || world <= 0 || !guard) return RESEL_EINVAL;
    hipLaunchKernelGGL(guard_apply_slots_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, slots, world, guard);
    return launch_status();
}

extern "C" int resel_soft_update(float* target, const float* online, float tau, int64_t n, resel_stream_t stream) {
    if (!target || !online || n <= 0 || !aligned16(target) || !aligned16(online)) return RESEL_EINVAL;
    const int64_t nthr = (n + 3) / 4;
    hipLaunchKernelGGL(soft_update_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, target, online, tau, n);
    return launch_status();
}

extern "C" int resel_adamw_flat(float* p, const float* g, float* m, float* v, int64_t n, const int64_t* seg_end,
                                const float* seg_lr, const float* seg_wd, int nseg, float beta1, float beta2, float eps,
                                int step, const float* grad_scale, resel_stream_t stream) {
    if (!p || !g || !m || !v || n <= 0 || !seg_end || !seg_lr || !seg_wd || nseg <= 0 || step <= 0) return RESEL_EINVAL;
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_flat_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
                       seg_end, seg_lr, seg_wd, nseg, beta1, beta2, eps, bc1, bc2_sqrt, grad_scale, (const float*)nullptr);
    return launch_status();
}

extern "C" int resel_adamw_flat_dev(float* p, const float* g, float* m, float* v, int64_t n, const int64_t* seg_end,
                                    const float* seg_lr, const float* seg_wd, int nseg, float beta1, float beta2, float eps,
                                    const float* bias_corrections, const float* grad_scale, resel_stream_t stream) {
    if (!p || !g || !m || !v || n <= 0 || !seg_end || !seg_lr || !seg_wd || nseg <= 0 || !bias_corrections) return RESEL_EINVAL;
    hipLaunchKernelGGL(adamw_flat_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
                       seg_end, seg_lr, seg_wd, nseg, beta1, beta2, eps, 1.f, 1.f, grad_scale, bias_corrections);
    return launch_status();
}

extern "C" size_t resel_sumsq_workspace_bytes(int64_t n) { (void)n; return RED_BLOCKS * sizeof(float); }

extern "C" int resel_sumsq(const float* x, int64_t n, float* out, void* workspace, resel_stream_t stream) {
    if (!x || n <= 0 || !out || !workspace) return RESEL_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int nblk = (int)((n + 255) / 256 < RED_BLOCKS ? (n + 255) / 256 : RED_BLOCKS);
    hipLaunchKernelGGL(sumsq_part_kernel, dim3(nblk), dim3(256), 0, s, x, n, (float*)workspace);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, nblk, out);
    return launch_status();
}
