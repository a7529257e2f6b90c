// Masked losses of the update as one forward and one backward pass each (reference algorithm/sac_full_length_rnn_ensembleQ.py:80-81
// `_mask_mean`, :105-114 `_Q_loss`, sac_full_length_rnn_redq.py:37-49 / td3_full_length_rnn_redq.py:39-51 `_policy_loss`, :130-132
// `_alpha_loss`).  The trainer keeps the sums UN-normalised (the global valid count divides the gradient inside AdamW), so:
//   critic:  L = sum_m mask[m] sum_e (q[e, m] - y[m])^2                          dq[e, m] = 2 g mask[m] (q[e, m] - y[m])
//   actor:   L = sum_m mask[m] (use_logp alpha logp[m] - red_e q[e, m])           red = mean (REDQ) or min (ensemble-min)
//            S = sum_m mask[m] logp[m]   (the entropy-coefficient gradient and the logged log-prob come from it)
//            dlogp[m] = g use_logp alpha mask[m];  dq[e, m] = -g mask[m] / E  (mean)  or  -g mask[m] [e = argmin, first on ties]
// torch's autograd spelt these as ~35 element-wise / reduction launches of ~5 us per update over [E, M] = 8 x 66 752 operands.
// Two-stage sums in a fixed order (per-block partials, one finishing block): bitwise reproducible, no atomics.
#include "resel_common.h"
#include <algorithm>

namespace {
using namespace resel;
constexpr int LOSS_BLOCKS = 256;

__device__ __forceinline__ void block_sum2(float a, float b, float* out2) {
    __shared__ float s[2][4];
    a = wave_sum(a); b = wave_sum(b);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s[0][w] = a; s[1][w] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out2[0] = (s[0][0] + s[0][1]) + (s[0][2] + s[0][3]);
        out2[1] = (s[1][0] + s[1][1]) + (s[1][2] + s[1][3]);
    }
}

__global__ __launch_bounds__(256) void q_loss_fwd_kernel(const float* __restrict__ q, const float* __restrict__ y, const float* __restrict__ mask,
                                                         float* __restrict__ part, int E, int M) {
    float acc = 0.f;
    for (int m = blockIdx.x * 256 + threadIdx.x; m < M; m += gridDim.x * 256) {
        const float mk = mask ? mask[m] : 1.f, t = y[m];
        float s = 0.f;
        for (int e = 0; e < E; ++e) { const float d = q[(int64_t)e * M + m] - t; s = __builtin_fmaf(d, d, s); }
        acc = __builtin_fmaf(mk, s, acc);
    }
    block_sum2(acc, 0.f, part + 2 * blockIdx.x);
}
__global__ __launch_bounds__(256) void q_loss_bwd_kernel(const float* __restrict__ q, const float* __restrict__ y, const float* __restrict__ mask,
                                                         const float* __restrict__ g, float* __restrict__ dq, int E, int M) {
    const float g2 = 2.f * g[0];
    for (int m = blockIdx.x * 256 + threadIdx.x; m < M; m += gridDim.x * 256) {
        const float c = g2 * (mask ? mask[m] : 1.f), t = y[m];
        for (int e = 0; e < E; ++e) dq[(int64_t)e * M + m] = c * (q[(int64_t)e * M + m] - t);
    }
}
__global__ __launch_bounds__(256) void actor_loss_fwd_kernel(const float* __restrict__ logp, const float* __restrict__ q, const float* __restrict__ mask,
                                                             const float* __restrict__ log_alpha, float* __restrict__ part, int E, int M,
                                                             int use_logp, int reduce_min) {
    const float alpha = use_logp ? expf(log_alpha[0]) : 0.f;
    float acc = 0.f, sl = 0.f;
    for (int m = blockIdx.x * 256 + threadIdx.x; m < M; m += gridDim.x * 256) {
        const float mk = mask ? mask[m] : 1.f;
        float r = q[m];
        for (int e = 1; e < E; ++e) { const float v = q[(int64_t)e * M + m]; r = reduce_min ? fminf(r, v) : r + v; }
        if (!reduce_min) r *= 1.f / (float)E;
        const float lp = logp ? logp[m] : 0.f;
        acc = __builtin_fmaf(mk, alpha * lp - r, acc);
        sl = __builtin_fmaf(mk, lp, sl);
    }
    block_sum2(acc, sl, part + 2 * blockIdx.x);
}
__global__ __launch_bounds__(256) void actor_loss_bwd_kernel(const float* __restrict__ q, const float* __restrict__ mask, const float* __restrict__ log_alpha,
                                                             const float* __restrict__ g, float* __restrict__ dlogp, float* __restrict__ dq, int E, int M,
                                                             int use_logp, int reduce_min) {
    const float gg = g[0], alpha = use_logp ? expf(log_alpha[0]) : 0.f;
    for (int m = blockIdx.x * 256 + threadIdx.x; m < M; m += gridDim.x * 256) {
        const float c = gg * (mask ? mask[m] : 1.f);
        if (dlogp) dlogp[m] = alpha * c;
        if (reduce_min) {
            int am = 0;
            float r = q[m];
            for (int e = 1; e < E; ++e) { const float v = q[(int64_t)e * M + m]; if (v < r) { r = v; am = e; } }
            for (int e = 0; e < E; ++e) dq[(int64_t)e * M + m] = e == am ? -c : 0.f;
        } else {
            const float v = -c / (float)E;
            for (int e = 0; e < E; ++e) dq[(int64_t)e * M + m] = v;
        }
    }
}
__global__ __launch_bounds__(256) void loss_final_kernel(const float* __restrict__ part, int nblk, float* __restrict__ out2) {
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) { a += part[2 * i]; b += part[2 * i + 1]; }
    block_sum2(a, b, out2);
}
inline int nblocks(int M) { return std::min(LOSS_BLOCKS, (M + 255) / 256); }
}  // namespace

extern "C" size_t resel_masked_loss_workspace_bytes(void) { return (size_t)2 * LOSS_BLOCKS * sizeof(float); }

extern "C" int resel_q_loss_fwd(const float* q, const float* y, const float* mask, float* out2, void* workspace, int E, int M, resel_stream_t stream) {
    if (!q || !y || !out2 || !workspace || E <= 0 || M <= 0) return RESEL_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int nb = nblocks(M);
    hipLaunchKernelGGL(q_loss_fwd_kernel, dim3(nb), dim3(256), 0, s, q, y, mask, (float*)workspace, E, M);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, nb, out2);
    return launch_status();
}
extern "C" int resel_q_loss_bwd(const float* q, const float* y, const float* mask, const float* g, float* dq, int E, int M, resel_stream_t stream) {
    if (!q || !y || !g || !dq || E <= 0 || M <= 0) return RESEL_EINVAL;
    hipLaunchKernelGGL(q_loss_bwd_kernel, dim3(nblocks(M)), dim3(256), 0, (hipStream_t)stream, q, y, mask, g, dq, E, M);
    return launch_status();
}
extern "C" int resel_actor_loss_fwd(const float* logp, const float* q, const float* mask, const float* log_alpha, float* out2, void* workspace,
                                    int E, int M, int use_logp, int reduce_min, resel_stream_t stream) {
    if (!q || !out2 || !workspace || E <= 0 || M <= 0 || (use_logp && (!logp || !log_alpha))) return RESEL_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int nb = nblocks(M);
    hipLaunchKernelGGL(actor_loss_fwd_kernel, dim3(nb), dim3(256), 0, s, logp, q, mask, log_alpha, (float*)workspace, E, M, use_logp, reduce_min);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, nb, out2);
    return launch_status();
}
extern "C" int resel_actor_loss_bwd(const float* q, const float* mask, const float* log_alpha, const float* g, float* dlogp, float* dq,
                                    int E, int M, int use_logp, int reduce_min, resel_stream_t stream) {
    if (!q || !g || !dq || E <= 0 || M <= 0 || (use_logp && (!log_alpha || !dlogp))) return RESEL_EINVAL;
    hipLaunchKernelGGL(actor_loss_bwd_kernel, dim3(nblocks(M)), dim3(256), 0, (hipStream_t)stream, q, mask, log_alpha, g, dlogp, dq, E, M, use_logp,
                       reduce_min);
    return launch_status();
}
