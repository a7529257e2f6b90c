// Depthwise causal conv1d + bias + SiLU on the masked input, token-major [B*L, Di] - forward and backward.
//
// Memory-bound (2 floats of HBM traffic per element): each block stages a [TT + K - 1][64-channel] tile of the
// masked input in LDS with coalesced float4 loads (16 lanes x 16 B = one 256-byte row segment), every thread
// owns 4 channels and keeps their K taps in registers (the tap count is padded to KT in {4, 8, 16, 32} with
// zero weights), and produces float4 outputs from K ds_read_b128 + 4K FMAs each.  The zero left pad is per
// row b only: packed trajectories inside a row are separated by the mask and the reset gap, exactly like the
// reference's nn.Conv1d over the whole row (offpolicy_rnn/models/smamba/mamba.py:210-212).
#include "resel_common.h"

namespace {
using namespace resel;

constexpr int TT = 64;      // output time steps per tile
constexpr int TILE_C = 64;

struct ConvParams {
    const float *x, *w, *bias, *mask, *dy;
    float *y, *dx, *dw_part, *db_part;
    int64_t ld_x, ld_y, ld_dy, ld_dx;
    int B, L, Di, K, silu;
};

template <int KT>
__device__ __forceinline__ void load_taps(const ConvParams& p, int c, bool ok, float4 (&wr)[KT]) {
    // wr[k'] holds tap k = k' - (KT - K) of the 4 channels c..c+3 (zero for the padded leading taps)
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {
        const int k = kk - (KT - p.K);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok && k >= 0) {
            v.x = p.w[(int64_t)(c + 0) * p.K + k];
            v.y = p.w[(int64_t)(c + 1) * p.K + k];
            v.z = p.w[(int64_t)(c + 2) * p.K + k];
            v.w = p.w[(int64_t)(c + 3) * p.K + k];
        }
        wr[kk] = v;
    }
}

// rows [t0 - (KT-1), t0 + nrows - (KT-1)) of the masked input -> s_x[0..nrows)
template <int ROWS>
__device__ __forceinline__ void stage_x(const ConvParams& p, float (*s_x)[TILE_C], int64_t tok0, int t_first, int nrows,
                                        int d0, int tid, int nthreads) {
    const int tc4 = (tid & 15) * 4;
    const bool c_ok = (d0 + tc4) < p.Di;
    for (int r = tid >> 4; r < nrows; r += nthreads / 16) {
        const int t = t_first + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t >= 0 && t < p.L && c_ok) {
            v = ld4(p.x + (tok0 + t) * p.ld_x + d0 + tc4);
            if (p.mask) {
                const float m = p.mask[tok0 + t];
                v.x *= m; v.y *= m; v.z *= m; v.w *= m;
            }
        }
        st4(&s_x[r][tc4], v);
    }
}

template <int KT>
__global__ __launch_bounds__(256) void conv_fwd_kernel(ConvParams p) {
    __shared__ __attribute__((aligned(16))) float s_x[TT + KT - 1][TILE_C];
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * TT, b = blockIdx.y, d0 = blockIdx.z * TILE_C;
    const int64_t tok0 = (int64_t)b * p.L;
    const int tc4 = (tid & 15) * 4, tr = tid >> 4;
    const bool c_ok = (d0 + tc4) < p.Di;
    float4 wr[KT];
    load_taps<KT>(p, d0 + tc4, c_ok, wr);
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c_ok && p.bias) bv = ld4(p.bias + d0 + tc4);
    stage_x<TT + KT - 1>(p, s_x, tok0, t0 - (KT - 1), TT + KT - 1, d0, tid, 256);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TT / 16; ++i) {
        const int r = tr + i * 16, t = t0 + r;
        if (t < p.L && c_ok) {
            float4 acc = bv;
#pragma unroll
            for (int kk = 0; kk < KT; ++kk) {
                const float4 xv = ld4(&s_x[r + kk][tc4]);
                acc.x = __builtin_fmaf(wr[kk].x, xv.x, acc.x); acc.y = __builtin_fmaf(wr[kk].y, xv.y, acc.y);
                acc.z = __builtin_fmaf(wr[kk].z, xv.z, acc.z); acc.w = __builtin_fmaf(wr[kk].w, xv.w, acc.w);
            }
            if (p.silu) { acc.x = siluf_(acc.x); acc.y = siluf_(acc.y); acc.z = siluf_(acc.z); acc.w = siluf_(acc.w); }
            st4(p.y + (tok0 + t) * p.ld_y + d0 + tc4, acc);
        }
    }
}

// Backward: one block per (row b, channel tile) walks the time tiles, so dw / dbias accumulate in registers and
// only per-row partials [B, Di, KT] / [B, Di] leave the block (summed by conv_reduce_kernel, no atomics).
//   g[t]   = dy[t] * silu'(pre[t])                      pre = conv output before the activation (recomputed)
//   dx[t]  = mask[t] * sum_k w[k] * g[t + (K-1) - k]
//   dw[k]  = sum_t g[t] * xm[t - (K-1) + k]
template <int KT>
__global__ __launch_bounds__(256) void conv_bwd_kernel(ConvParams p) {
    __shared__ __attribute__((aligned(16))) float s_x[TT + 2 * (KT - 1)][TILE_C];   // masked x rows [t0-(KT-1), t0+TT+KT-1)
    __shared__ __attribute__((aligned(16))) float s_g[TT + KT - 1][TILE_C];         // g rows [t0, t0+TT+KT-1)
    const int tid = threadIdx.x;
    const int b = blockIdx.x, d0 = blockIdx.y * TILE_C;
    const int64_t tok0 = (int64_t)b * p.L;
    const int tc4 = (tid & 15) * 4, tr = tid >> 4;
    const bool c_ok = (d0 + tc4) < p.Di;
    float4 wr[KT], dwr[KT];
    load_taps<KT>(p, d0 + tc4, c_ok, wr);
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) dwr[kk] = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), dbr = bv;
    if (c_ok && p.bias) bv = ld4(p.bias + d0 + tc4);

    for (int t0 = 0; t0 < p.L; t0 += TT) {
        __syncthreads();
        stage_x<TT + 2 * (KT - 1)>(p, s_x, tok0, t0 - (KT - 1), TT + 2 * (KT - 1), d0, tid, 256);
        __syncthreads();
        // g for rows [t0, t0 + TT + KT - 1); rows >= t0 + TT are halo recomputed for dx only
        for (int r = tr; r < TT + KT - 1; r += 16) {
            const int t = t0 + r;
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < p.L && c_ok) {
                float4 acc = bv;
#pragma unroll
                for (int kk = 0; kk < KT; ++kk) {
                    const float4 xv = ld4(&s_x[r + kk][tc4]);
                    acc.x = __builtin_fmaf(wr[kk].x, xv.x, acc.x); acc.y = __builtin_fmaf(wr[kk].y, xv.y, acc.y);
                    acc.z = __builtin_fmaf(wr[kk].z, xv.z, acc.z); acc.w = __builtin_fmaf(wr[kk].w, xv.w, acc.w);
                }
                g = ld4(p.dy + (tok0 + t) * p.ld_dy + d0 + tc4);
                if (p.silu) { g.x *= dsiluf_(acc.x); g.y *= dsiluf_(acc.y); g.z *= dsiluf_(acc.z); g.w *= dsiluf_(acc.w); }
                if (r < TT) {                       // rows owned by this tile contribute to dw / dbias
                    dbr.x += g.x; dbr.y += g.y; dbr.z += g.z; dbr.w += g.w;
#pragma unroll
                    for (int kk = 0; kk < KT; ++kk) {
                        const float4 xv = ld4(&s_x[r + kk][tc4]);
                        dwr[kk].x = __builtin_fmaf(g.x, xv.x, dwr[kk].x); dwr[kk].y = __builtin_fmaf(g.y, xv.y, dwr[kk].y);
                        dwr[kk].z = __builtin_fmaf(g.z, xv.z, dwr[kk].z); dwr[kk].w = __builtin_fmaf(g.w, xv.w, dwr[kk].w);
                    }
                }
            }
            st4(&s_g[r][tc4], g);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TT / 16; ++i) {
            const int r = tr + i * 16, t = t0 + r;
            if (t < p.L && c_ok) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int kk = 0; kk < KT; ++kk) {       // tap kk pairs x[t] with the output at t + (KT-1) - kk
                    const float4 gv = ld4(&s_g[r + (KT - 1) - kk][tc4]);
                    acc.x = __builtin_fmaf(wr[kk].x, gv.x, acc.x); acc.y = __builtin_fmaf(wr[kk].y, gv.y, acc.y);
                    acc.z = __builtin_fmaf(wr[kk].z, gv.z, acc.z); acc.w = __builtin_fmaf(wr[kk].w, gv.w, acc.w);
                }
                if (p.mask) {
                    const float m = p.mask[tok0 + t];
                    acc.x *= m; acc.y *= m; acc.z *= m; acc.w *= m;
                }
                st4(p.dx + (tok0 + t) * p.ld_dx + d0 + tc4, acc);
            }
        }
    }
    // reduce the 16 row-threads that share a channel quad, then write the per-row partial
    __syncthreads();
    float* s_red = &s_x[0][0];                       // [16][64] floats per tap, reused tap by tap
#pragma unroll
    for (int kk = 0; kk <= KT; ++kk) {               // unrolled: dwr[] must be indexed with compile-time constants
        const float4 v = (kk < KT) ? dwr[kk < KT ? kk : 0] : dbr;
        st4(&s_red[tr * TILE_C + tc4], v);
        __syncthreads();
        if (tid < TILE_C && d0 + tid < p.Di) {
            float acc = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc += s_red[r * TILE_C + tid];
            if (kk < KT) p.dw_part[((int64_t)b * p.Di + d0 + tid) * KT + kk] = acc;
            else p.db_part[(int64_t)b * p.Di + d0 + tid] = acc;
        }
        __syncthreads();
    }
}

// dw[d, k] = sum_b dw_part[b, d, KT - K + k] ; dbias[d] = sum_b db_part[b, d]
__global__ void conv_reduce_kernel(const float* dw_part, const float* db_part, float* dw, float* dbias,
                                   int B, int Di, int K, int KT) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Di * K) {
        const int d = i / K, k = i % K;
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc += dw_part[((int64_t)b * Di + d) * KT + (KT - K) + k];
        dw[i] = acc;
    }
    if (dbias != nullptr && i < Di) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc += db_part[(int64_t)b * Di + i];
        dbias[i] = acc;
    }
}

inline int pad_taps(int K) { return K <= 4 ? 4 : (K <= 8 ? 8 : (K <= 16 ? 16 : 32)); }
inline bool conv_args_ok(const float* x, int64_t ld_x, const float* o, int64_t ld_o, int B, int L, int Di, int K) {
    return x && o && B > 0 && L > 0 && Di > 0 && Di % 4 == 0 && K >= 1 && K <= 32 && ld_x % 4 == 0 && ld_o % 4 == 0 &&
           aligned16(x) && aligned16(o);
}

}  // namespace

extern "C" int resel_causal_conv1d_fwd(const float* x, int64_t ld_x, const float* w, const float* bias, const float* mask,
                                       float* y, int64_t ld_y, int B, int L, int Di, int K, int silu,
                                       resel_stream_t stream) {
    if (!conv_args_ok(x, ld_x, y, ld_y, B, L, Di, K) || !w || (bias && !aligned16(bias))) return RESEL_EINVAL;
    ConvParams p{x, w, bias, mask, nullptr, y, nullptr, nullptr, nullptr, ld_x, ld_y, 0, 0, B, L, Di, K, silu};
    dim3 grid((L + TT - 1) / TT, B, (Di + TILE_C - 1) / TILE_C);
    hipStream_t s = (hipStream_t)stream;
    switch (pad_taps(K)) {
        case 4: hipLaunchKernelGGL(conv_fwd_kernel<4>, grid, dim3(256), 0, s, p); break;
        case 8: hipLaunchKernelGGL(conv_fwd_kernel<8>, grid, dim3(256), 0, s, p); break;
        case 16: hipLaunchKernelGGL(conv_fwd_kernel<16>, grid, dim3(256), 0, s, p); break;
        default: hipLaunchKernelGGL(conv_fwd_kernel<32>, grid, dim3(256), 0, s, p); break;
    }
    return launch_status();
}

extern "C" size_t resel_causal_conv1d_bwd_workspace_bytes(int B, int L, int Di, int K) {
    (void)L;
    return ((size_t)B * Di * pad_taps(K) + (size_t)B * Di) * sizeof(float);
}

extern "C" int resel_causal_conv1d_bwd(const float* x, int64_t ld_x, const float* w, const float* bias, const float* mask,
                                       const float* dy, int64_t ld_dy, float* dx, int64_t ld_dx, float* dw, float* dbias,
                                       void* workspace, int B, int L, int Di, int K, int silu, resel_stream_t stream) {
    if (!conv_args_ok(x, ld_x, dx, ld_dx, B, L, Di, K) || !w || !dy || !dw || !workspace || ld_dy % 4 || !aligned16(dy) ||
        (bias && !aligned16(bias)))
        return RESEL_EINVAL;
    const int KT = pad_taps(K);
    float* dw_part = (float*)workspace;
    float* db_part = dw_part + (size_t)B * Di * KT;
    ConvParams p{x, w, bias, mask, dy, nullptr, dx, dw_part, db_part, ld_x, 0, ld_dy, ld_dx, B, L, Di, K, silu};
    dim3 grid(B, (Di + TILE_C - 1) / TILE_C);
    hipStream_t s = (hipStream_t)stream;
    switch (KT) {
        case 4: hipLaunchKernelGGL(conv_bwd_kernel<4>, grid, dim3(256), 0, s, p); break;
        case 8: hipLaunchKernelGGL(conv_bwd_kernel<8>, grid, dim3(256), 0, s, p); break;
        case 16: hipLaunchKernelGGL(conv_bwd_kernel<16>, grid, dim3(256), 0, s, p); break;
        default: hipLaunchKernelGGL(conv_bwd_kernel<32>, grid, dim3(256), 0, s, p); break;
    }
    const int n = Di * K;
    hipLaunchKernelGGL(conv_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dw_part, db_part, dw, dbias, B, Di, K, KT);
    return launch_status();
}
