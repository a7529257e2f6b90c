// Fused residual add + LayerNorm / RMSNorm over rows of width C, forward and backward.
//
// HBM-bound: one wave per row, 16 B per lane per access (a 256-wide row is exactly one float4 per lane),
// statistics by wave-level xor reductions (no LDS, no second pass over memory).  The backward accumulates the
// weight / bias gradients of the rows a wave visits in registers and leaves one partial per block; a second
// kernel sums the partials in a fixed order (no atomics, bitwise reproducible).
#include "resel_common.h"
#include <algorithm>

namespace {
using namespace resel;

__device__ __forceinline__ float elu1(float y) { return y > 0.f ? y : expm1f(y); }         // ELU, alpha = 1 (torch's expm1 form)
__device__ __forceinline__ float delu1(float y) { return y > 0.f ? 1.f : expf(y); }         // its derivative from the pre-activation

constexpr int WAVES = 4;            // rows in flight per block
constexpr int BWD_BLOCKS = 512;     // 2 per CU

template <int VPL>                  // float4 per lane: C <= VPL * 256
__global__ __launch_bounds__(WAVES * 64) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ residual,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            float* __restrict__ y, float* __restrict__ res_out,
                                                            float* __restrict__ stats, int M, int C, float eps, int rms, int act, AmaxOut amax) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (row >= M) return;
    const int c4 = C / 4;
    const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * C);
    const float4* rr = residual ? reinterpret_cast<const float4*>(residual + (int64_t)row * C) : nullptr;
    float4 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = i * 64 + lane;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < c4) {
            v[i] = xr[c];
            if (rr) { const float4 q = rr[c]; v[i].x += q.x; v[i].y += q.y; v[i].z += q.z; v[i].w += q.w; }
            if (res_out) reinterpret_cast<float4*>(res_out + (int64_t)row * C)[c] = v[i];
        }
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = rms ? 0.f : wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = i * 64 + lane;
        if (c < c4) {
            const float a = v[i].x - mean, bq = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + bq * bq) + (cc * cc + d * d);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    if (stats && lane == 0) { stats[2 * (int64_t)row] = mean; stats[2 * (int64_t)row + 1] = rstd; }
    float wmax = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = i * 64 + lane;
        if (c < c4) {
            const float4 wv = reinterpret_cast<const float4*>(w)[c];
            float4 o;
            o.x = (v[i].x - mean) * rstd * wv.x; o.y = (v[i].y - mean) * rstd * wv.y;
            o.z = (v[i].z - mean) * rstd * wv.z; o.w = (v[i].w - mean) * rstd * wv.w;
            if (b) { const float4 bb = reinterpret_cast<const float4*>(b)[c]; o.x += bb.x; o.y += bb.y; o.z += bb.z; o.w += bb.w; }
            if (act) { o.x = elu1(o.x); o.y = elu1(o.y); o.z = elu1(o.z); o.w = elu1(o.w); }      // |elu(y)| <= |y|: the bound below still holds
            reinterpret_cast<float4*>(y + (int64_t)row * C)[c] = o;
            wmax = amax4(wmax, wv);
        }
    }
    // Magnitude of the output for the projection that reads it (GEMM mode 2): a normalised row obeys |x_i - mean| * rstd <= sqrt(C)
    // (LayerNorm and RMSNorm alike), so sqrt(C) max|w| + max|b| bounds every |y| - no per-row work, ONE wave publishes it.  The bound
    // is ~4x the typical maximum at C = 256: two of the 29 bits of element range of the scaled operand (include/resel_hip.h).
    if (amax.slot != nullptr && row == 0) {
        float bmax = 0.f;
        if (b) {
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const int c = i * 64 + lane;
                if (c < c4) bmax = amax4(bmax, reinterpret_cast<const float4*>(b)[c]);
            }
        }
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) { wmax = fmaxf(wmax, __shfl_xor(wmax, sh, 64)); bmax = fmaxf(bmax, __shfl_xor(bmax, sh, 64)); }
        amax_publish_wave(sqrtf((float)C) * wmax + bmax, amax);
    }
}

template <int VPL, bool ACT>          // ACT: the forward stored elu(y) (a compile-time switch: as a run-time one it cost the plain form 61.8 -> 69.2 us)
__global__ __launch_bounds__(WAVES * 64) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ dres_in,
                                                            const float* __restrict__ res, const float* __restrict__ w,
                                                            const float* __restrict__ b, const float* __restrict__ stats, float* __restrict__ dx,
                                                            float* __restrict__ dw_part, float* __restrict__ db_part,
                                                            int M, int C, int rms, AmaxOut amax) {
    __shared__ __attribute__((aligned(16))) float s_acc[2][WAVES][VPL * 256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c4 = C / 4;
    float4 dwa[VPL], dba[VPL], wreg[VPL], breg[ACT ? VPL : 1];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        dwa[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        dba[i] = dwa[i];
        const int c = i * 64 + lane;
        wreg[i] = c < c4 ? reinterpret_cast<const float4*>(w)[c] : dwa[i];
        if constexpr (ACT) breg[i] = (b && c < c4) ? reinterpret_cast<const float4*>(b)[c] : dwa[i];
    }
    float dxmax = 0.f;
    for (int row = blockIdx.x * WAVES + wv; row < M; row += gridDim.x * WAVES) {
        const float mean = stats[2 * (int64_t)row], rstd = stats[2 * (int64_t)row + 1];
        float4 xh[VPL], g[VPL];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = i * 64 + lane;
            xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            g[i] = xh[i];
            if (c < c4) {
                const float4 r = reinterpret_cast<const float4*>(res + (int64_t)row * C)[c];
                float4 d = reinterpret_cast<const float4*>(dy + (int64_t)row * C)[c];
                xh[i].x = (r.x - mean) * rstd; xh[i].y = (r.y - mean) * rstd; xh[i].z = (r.z - mean) * rstd; xh[i].w = (r.w - mean) * rstd;
                if constexpr (ACT) {          // the forward stored elu(y), y = xh w + b: its derivative from the recomputed y (1 for y > 0, e^y below)
                    d.x *= delu1(xh[i].x * wreg[i].x + breg[i].x); d.y *= delu1(xh[i].y * wreg[i].y + breg[i].y);
                    d.z *= delu1(xh[i].z * wreg[i].z + breg[i].z); d.w *= delu1(xh[i].w * wreg[i].w + breg[i].w);
                }
                dwa[i].x += d.x * xh[i].x; dwa[i].y += d.y * xh[i].y; dwa[i].z += d.z * xh[i].z; dwa[i].w += d.w * xh[i].w;
                dba[i].x += d.x; dba[i].y += d.y; dba[i].z += d.z; dba[i].w += d.w;
                g[i].x = d.x * wreg[i].x; g[i].y = d.y * wreg[i].y; g[i].z = d.z * wreg[i].z; g[i].w = d.w * wreg[i].w;
                c1 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
                c2 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
            }
        }
        c1 = wave_sum(c1) / (float)C;
        c2 = rms ? 0.f : wave_sum(c2) / (float)C;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = i * 64 + lane;
            if (c < c4) {
                float4 o;
                o.x = (g[i].x - (xh[i].x * c1 + c2)) * rstd; o.y = (g[i].y - (xh[i].y * c1 + c2)) * rstd;
                o.z = (g[i].z - (xh[i].z * c1 + c2)) * rstd; o.w = (g[i].w - (xh[i].w * c1 + c2)) * rstd;
                if (dres_in) {
                    const float4 q = reinterpret_cast<const float4*>(dres_in + (int64_t)row * C)[c];
                    o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w;
                }
                reinterpret_cast<float4*>(dx + (int64_t)row * C)[c] = o;
                dxmax = amax4(dxmax, o);
            }
        }
    }
    amax_publish_wave(dxmax, amax);                  // persistent waves: one publication each (dx feeds the out_proj gradient GEMMs of the block below)
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        st4(&s_acc[0][wv][(i * 64 + lane) * 4], dwa[i]);
        st4(&s_acc[1][wv][(i * 64 + lane) * 4], dba[i]);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += WAVES * 64) {
        float a = 0.f, bq = 0.f;
#pragma unroll
        for (int k = 0; k < WAVES; ++k) { a += s_acc[0][k][c]; bq += s_acc[1][k][c]; }
        dw_part[(int64_t)blockIdx.x * C + c] = a;
        db_part[(int64_t)blockIdx.x * C + c] = bq;
    }
}

inline int bwd_blocks(int M) { const int need = (M + WAVES - 1) / WAVES; return need < BWD_BLOCKS ? need : BWD_BLOCKS; }
inline bool ln_ok(int M, int C) { return M > 0 && C > 0 && C % 4 == 0 && C <= 2048; }

}  // namespace

extern "C" int resel_add_layernorm_fwd(const float* x, const float* residual, const float* w, const float* b,
                                       float* y, float* res_out, float* stats, int M, int C, float eps, int rms, int act,
                                       void* amax_y, unsigned amax_epoch, resel_stream_t stream) {
    if (!x || !w || !y || !ln_ok(M, C) || (act != 0 && act != 1) || (amax_y && (reinterpret_cast<uintptr_t>(amax_y) & 7u))) return RESEL_EINVAL;
    const AmaxOut ao{(unsigned long long*)amax_y, amax_epoch};
    if (!aligned16(x) || !aligned16(y) || !aligned16(w) || (residual && !aligned16(residual)) || (b && !aligned16(b)) ||
        (res_out && !aligned16(res_out)))
        return RESEL_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((M + WAVES - 1) / WAVES), blk(WAVES * 64);
    if (C <= 256) hipLaunchKernelGGL(ln_fwd_kernel<1>, grid, blk, 0, s, x, residual, w, b, y, res_out, stats, M, C, eps, rms, act, ao);
    else if (C <= 512) hipLaunchKernelGGL(ln_fwd_kernel<2>, grid, blk, 0, s, x, residual, w, b, y, res_out, stats, M, C, eps, rms, act, ao);
    else if (C <= 1024) hipLaunchKernelGGL(ln_fwd_kernel<4>, grid, blk, 0, s, x, residual, w, b, y, res_out, stats, M, C, eps, rms, act, ao);
    else hipLaunchKernelGGL(ln_fwd_kernel<8>, grid, blk, 0, s, x, residual, w, b, y, res_out, stats, M, C, eps, rms, act, ao);
    return launch_status();
}

extern "C" size_t resel_add_layernorm_bwd_workspace_bytes(int M, int C) {
    return (size_t)2 * bwd_blocks(M) * C * sizeof(float);
}

extern "C" int resel_add_layernorm_bwd(const float* dy, const float* dres_in, const float* res, const float* w, const float* b,
                                       const float* stats, float* dx, float* dw, float* db, void* workspace,
                                       int M, int C, int rms, int has_bias, int act, void* amax_dx, unsigned amax_epoch, resel_stream_t stream) {
    if (!dy || !res || !w || !stats || !dx || !dw || !workspace || !ln_ok(M, C) || (reinterpret_cast<uintptr_t>(amax_dx) & 7u)) return RESEL_EINVAL;
    if ((act != 0 && act != 1) || (b && !aligned16(b)) || (has_bias && act && !b)) return RESEL_EINVAL;
    const AmaxOut ao{(unsigned long long*)amax_dx, amax_epoch};
    if (!aligned16(dy) || !aligned16(res) || !aligned16(w) || !aligned16(dx) || (dres_in && !aligned16(dres_in)))
        return RESEL_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int nblk = bwd_blocks(M);
    float* dw_part = (float*)workspace;
    float* db_part = dw_part + (size_t)nblk * C;
#define LN_BWD(V) do { if (act) hipLaunchKernelGGL((ln_bwd_kernel<V, true>), grid, blk, 0, s, dy, dres_in, res, w, b, stats, dx, dw_part, db_part, M, C, rms, ao); \
                       else hipLaunchKernelGGL((ln_bwd_kernel<V, false>), grid, blk, 0, s, dy, dres_in, res, w, b, stats, dx, dw_part, db_part, M, C, rms, ao); } while (0)
    dim3 grid(nblk), blk(WAVES * 64);
    if (C <= 256) LN_BWD(1);
    else if (C <= 512) LN_BWD(2);
    else if (C <= 1024) LN_BWD(4);
    else LN_BWD(8);
    launch_colsum(dw_part, C, nblk, C, dw, s);
    if (has_bias && db) launch_colsum(db_part, C, nblk, C, db, s);
    return launch_status();
}
