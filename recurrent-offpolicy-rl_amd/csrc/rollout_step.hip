// One-token (T = 1) rollout kernels: what `policy.forward` runs once per environment step between updates
// (reference algorithm/sac.py:319-326 -> models/rnn_base.py:437-452).  At B = 1..few rows these are latency-bound, not
// bandwidth-bound: each kernel does the whole per-layer state update in ONE launch, reads every operand once, and takes
// its step counter from device memory so that a whole policy step can be replayed as a hipGraph.
//   mamba_conv_step          conv window roll + depthwise conv + bias + SiLU          (smamba/mamba.py:262-271)
//   selective_state_update   dt_proj + softplus + h <- h exp(dt A) + dt B x, y = C.h + D x, * silu(z)
//                                                                                     (smamba/mamba.py:281-294,
//                                                                                      mamba_ssm/ops/triton/selective_state_update.py:123-154)
//   attn_decode              KV-cache append + one causal ALiBi attention row           (flash_attn MHA with inference_params,
//                                                                                      TransformerFlashAttention.py:76-81)
#include <hip/hip_bf16.h>
#include "resel_common.h"

namespace {
using namespace resel;

// ---- smamba: conv window ----------------------------------------------------------------------------------------
// one thread per (row, channel).  The stored window holds W taps of channel d at st[d * sd + j * sk], oldest first:
//   smamba (mamba.py:138-141): [Di, K] rows, W = K (the oldest tap is dropped);  s6 `mamba` (s6/mamba.py:166-176): time-major
//   [K - 1, Di], W = K - 1.  The conv reads the newest K - 1 stored taps + x; the new window is (stored[1:], x).
__global__ __launch_bounds__(256) void conv_step_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ st_in,
                                                        int64_t ld_in, float* __restrict__ st_out, int64_t ld_out, int64_t sd, int64_t sk,
                                                        int W, const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ xc, int B, int Di, int K, int act) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Di) return;
    const int b = i / Di, d = i % Di;
    const float* si = st_in + (int64_t)b * ld_in + (int64_t)d * sd;
    float* so = st_out + (int64_t)b * ld_out + (int64_t)d * sd;
    const float* wd = w + (int64_t)d * K;
    const float xn = x[(int64_t)b * ldx + d];
    float acc = bias ? bias[d] : 0.f;
    const int off = W - (K - 1);
    for (int j = 1; j < W; ++j) {
        const float v = si[(int64_t)j * sk];
        so[(int64_t)(j - 1) * sk] = v;
        if (j >= off) acc = __builtin_fmaf(v, wd[j - off], acc);
    }
    if (off == 0 && W > 0) acc = __builtin_fmaf(si[0], wd[0], acc);
    if (W > 0) so[(int64_t)(W - 1) * sk] = xn;
    acc = __builtin_fmaf(xn, wd[K - 1], acc);
    xc[(int64_t)b * Di + d] = act ? siluf_(acc) : acc;
}

// ---- smamba: state update -----------------------------------------------------------------------------------------
// 8 lanes per (row, channel): lane `sub` owns states n = sub, sub + 8, ... and the dt_proj terms r = sub, sub + 8, ...
__device__ __forceinline__ float sum8(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    return v;
}

__global__ __launch_bounds__(256) void state_update_kernel(const float* __restrict__ st_in, int64_t ld_in, float* __restrict__ st_out,
                                                           int64_t ld_out, const float* __restrict__ xc,
                                                           const float* __restrict__ xdb, int64_t ld_xdb,
                                                           const float* __restrict__ w_dt, const float* __restrict__ dt_bias,
                                                           const float* __restrict__ A_log, const float* __restrict__ Dskip,
                                                           const float* __restrict__ z, int64_t ldz, float* __restrict__ y,
                                                           int B, int Di, int N, int R) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 3, sub = threadIdx.x & 7;
    const bool live = i < B * Di;                       // whole 8-lane groups live or die together; keep them for the shuffles
    const int ic = live ? i : B * Di - 1;
    const int b = ic / Di, d = ic % Di;
    const float* row = xdb + (int64_t)b * ld_xdb;
    float dtp = 0.f;
    for (int r = sub; r < R; r += 8) dtp = __builtin_fmaf(row[r], w_dt[(int64_t)d * R + r], dtp);
    const float dt = softplusf_(sum8(dtp) + dt_bias[d]);
    const float xv = xc[(int64_t)b * Di + d];
    const float* si = st_in + (int64_t)b * ld_in + (int64_t)d * N;
    float* so = st_out + (int64_t)b * ld_out + (int64_t)d * N;
    const float* Bm = row + R;
    const float* Cm = row + R + N;
    float acc = 0.f;
    for (int n = sub; n < N; n += 8) {
        const float a = -__expf(A_log[(int64_t)d * N + n]);
        const float h = __builtin_fmaf(si[n], fast_exp(dt * a), dt * Bm[n] * xv);
        if (live) so[n] = h;
        acc = __builtin_fmaf(h, Cm[n], acc);
    }
    acc = sum8(acc);
    if (live && sub == 0) {
        float o = __builtin_fmaf(Dskip[d], xv, acc);
        if (z) o *= siluf_(z[(int64_t)b * ldz + d]);
        y[(int64_t)b * Di + d] = o;
    }
}

// ---- cgpt: KV-cache append + one attention row --------------------------------------------------------------------
// grid (H, B), 256 threads = 4 waves; lane-strided keys with a per-lane online softmax, combined through the wave and LDS.
__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
struct alignas(16) bf16x8 { uint16_t v[8]; };

template <int HD>
__global__ __launch_bounds__(256) void attn_decode_kernel(const uint16_t* __restrict__ qkv, int64_t ld_qkv, uint16_t* __restrict__ cache,
                                                          const int* __restrict__ pos_dev, int pos_host, const float* __restrict__ slopes,
                                                          uint16_t* __restrict__ out, float scale, int H, int S) {
    __shared__ float s_m[4], s_l[4], s_acc[4][HD];
    const int h = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int pos = pos_dev ? *pos_dev : pos_host;      // tokens already in the cache == position of this token
    const uint16_t* qrow = qkv + (int64_t)b * ld_qkv + (int64_t)h * HD;
    const uint16_t* krow = qrow + (int64_t)H * HD;
    const uint16_t* vrow = krow + (int64_t)H * HD;
    uint16_t* cb = cache + (int64_t)b * S * 2 * H * HD;  // [S, 2, H, HD]
    if (pos < 0 || pos >= S) {                            // cache full: the reference's flash-attn asserts; poison the output
        if (threadIdx.x < HD) out[((int64_t)b * H + h) * HD + threadIdx.x] = 0x7fc0;
        return;
    }
    if (threadIdx.x < HD / 8) {                           // append this token's k, v
        const int c = threadIdx.x * 8;
        *reinterpret_cast<bf16x8*>(cb + (((int64_t)pos * 2 + 0) * H + h) * HD + c) = *reinterpret_cast<const bf16x8*>(krow + c);
        *reinterpret_cast<bf16x8*>(cb + (((int64_t)pos * 2 + 1) * H + h) * HD + c) = *reinterpret_cast<const bf16x8*>(vrow + c);
    }
    float q[HD];
#pragma unroll
    for (int c = 0; c < HD; c += 8) {
        const bf16x8 t = *reinterpret_cast<const bf16x8*>(qrow + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) q[c + e] = bf2f(t.v[e]);
    }
    const float c1 = scale * RESEL_LOG2E, slope2 = slopes ? slopes[h] * RESEL_LOG2E : 0.f;
    float m = -1e30f, l = 0.f, acc[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) acc[c] = 0.f;
    // keys 0..pos-1 from the cache, key `pos` (this token) from the qkv row: handled by the thread that would own it
    for (int j = threadIdx.x; j <= pos; j += 256) {
        const uint16_t* kp = j < pos ? cb + (((int64_t)j * 2 + 0) * H + h) * HD : krow;
        const uint16_t* vp = j < pos ? cb + (((int64_t)j * 2 + 1) * H + h) * HD : vrow;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < HD; c += 8) {
            const bf16x8 t = *reinterpret_cast<const bf16x8*>(kp + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) s = __builtin_fmaf(q[c + e], bf2f(t.v[e]), s);
        }
        s = s * c1 - slope2 * (float)(pos - j);
        const float mn = fmaxf(m, s);
        const float r = fast_exp2(m - mn), p = fast_exp2(s - mn);
        l = l * r + p;
#pragma unroll
        for (int c = 0; c < HD; c += 8) {
            const bf16x8 t = *reinterpret_cast<const bf16x8*>(vp + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[c + e] = __builtin_fmaf(p, bf2f(t.v[e]), acc[c + e] * r);
        }
        m = mn;
    }
    float mw = m;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mw = fmaxf(mw, __shfl_xor(mw, o, 64));
    const float r = fast_exp2(m - mw);                   // lanes without keys: m = -1e30 -> r = 0 (mw is finite: key `pos` exists)
    l = wave_sum(l * r);
#pragma unroll
    for (int c = 0; c < HD; ++c) {
        const float t = wave_sum(acc[c] * r);
        if (lane == 0) s_acc[w][c] = t;
    }
    if (lane == 0) { s_m[w] = mw; s_l[w] = l; }
    __syncthreads();
    if (threadIdx.x < HD) {
        float mt = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
        float lt = 0.f, at = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float rk = fast_exp2(s_m[k] - mt);
            lt += s_l[k] * rk;
            at += s_acc[k][threadIdx.x] * rk;
        }
        out[((int64_t)b * H + h) * HD + threadIdx.x] = f2bf(at / lt);
    }
}

}  // namespace

extern "C" {

int resel_mamba_conv_step(const float* x, int64_t ldx, const float* state_in, int64_t ld_in, float* state_out, int64_t ld_out,
                          int64_t stride_d, int64_t stride_k, int W, const float* w, const float* bias, float* xc, int B, int Di,
                          int K, int act, resel_stream_t stream) {
    if (!x || !state_in || !state_out || !w || !xc) return RESEL_EINVAL;
    if (B <= 0 || Di <= 0 || K <= 0 || (W != K && W != K - 1)) return RESEL_EINVAL;
    const int n = B * Di;
    hipLaunchKernelGGL(conv_step_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, ldx, state_in, ld_in, state_out,
                       ld_out, stride_d, stride_k, W, w, bias, xc, B, Di, K, act);
    return launch_status();
}

int resel_selective_state_update(const float* state_in, int64_t ld_in, float* state_out, int64_t ld_out, const float* xc,
                                 const float* x_db, int64_t ld_xdb, const float* w_dt, const float* dt_bias, const float* A_log,
                                 const float* D, const float* z, int64_t ldz, float* y, int B, int Di, int N, int R,
                                 resel_stream_t stream) {
    if (!state_in || !state_out || !xc || !x_db || !w_dt || !dt_bias || !A_log || !D || !y) return RESEL_EINVAL;
    if (B <= 0 || Di <= 0 || N <= 0 || R <= 0) return RESEL_EINVAL;
    const int64_t n = (int64_t)B * Di * 8;
    hipLaunchKernelGGL(state_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, state_in, ld_in,
                       state_out, ld_out, xc, x_db, ld_xdb, w_dt, dt_bias, A_log, D, z, ldz, y, B, Di, N, R);
    return launch_status();
}

int resel_attn_decode(const uint16_t* qkv, int64_t ld_qkv, uint16_t* kv_cache, const int32_t* pos_dev, int pos_host, const float* slopes,
                      uint16_t* out, float scale, int B, int H, int head_dim, int max_seqlen, resel_stream_t stream) {
    if (!qkv || !kv_cache || !out) return RESEL_EINVAL;
    if (B <= 0 || H <= 0 || max_seqlen <= 0) return RESEL_EINVAL;
    if (!aligned16(qkv) || !aligned16(kv_cache) || (ld_qkv % 8) != 0) return RESEL_EINVAL;
    if (!pos_dev && (pos_host < 0 || pos_host >= max_seqlen)) return RESEL_EINVAL;
    const dim3 grid(H, B), block(256);
    auto q = qkv;
    auto c = kv_cache;
    auto o = out;
    hipStream_t s = (hipStream_t)stream;
    switch (head_dim) {
        case 32: hipLaunchKernelGGL(attn_decode_kernel<32>, grid, block, 0, s, q, ld_qkv, c, pos_dev, pos_host, slopes, o, scale, H, max_seqlen); break;
        case 64: hipLaunchKernelGGL(attn_decode_kernel<64>, grid, block, 0, s, q, ld_qkv, c, pos_dev, pos_host, slopes, o, scale, H, max_seqlen); break;
        default: return RESEL_EINVAL;
    }
    return launch_status();
}

}  // extern "C"
