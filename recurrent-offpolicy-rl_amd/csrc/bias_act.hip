// Bias + activation tail of the fc / efc-E layers, fused around the library GEMM (HBM-bound, one pass each way).
//   forward  (in place on the GEMM output):   a[r, c] = act(y[r, c] + bias[seg(r), c])
//   backward (from the OUTPUT, so the pre-activation is never kept):
//            gy[r, c] = g[r, c] * act'(.)        ELU: a > 0 ? 1 : a + 1
//            dbias[s, c] = sum_{r in segment s} gy[r, c]   (per-block partials -> colsum_kernel, fixed order, no atomics)
// Rows are grouped in `nseg` segments of `rows_per_seg` consecutive rows, each with its own bias row: nseg = E for the
// per-member layers of the ensemble critic ([E, M, out] contiguous, reference EnsembleLinear bias [E, 1, out],
// ensemble_linear_model.py:23-26), nseg = 1 for nn.Linear and for the shared-input ensemble layer ([M, E*out]).
// Replaces torch's broadcast-bias copy in front of baddbmm (547 MB written per critic layer at config 2), the separate
// ELU pass over a saved pre-activation and the column-sum pass of the bias gradient.
#include "resel_common.h"

namespace {
using namespace resel;

constexpr int BWD_ROWS = 128;        // rows per backward block of the head kernel; bias_act_bwd picks 128 / 64 / 32 (block_rows) to fill the chip

__device__ __forceinline__ float elu_(float x) { return x > 0.f ? x : fast_exp(x) - 1.f; }

__global__ __launch_bounds__(256) void bias_act_fwd_kernel(float* __restrict__ y, const float* __restrict__ bias, int64_t rows,
                                                           int C, int64_t rows_per_seg, int act) {
    const int c4n = C / 4;
    const int64_t n4 = rows * c4n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c4n;
        const int c = (int)(i % c4n) * 4;
        float4 v = ld4(y + r * C + c);
        if (bias) {
            const float4 b = ld4(bias + (r / rows_per_seg) * C + c);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        if (act == 1) { v.x = elu_(v.x); v.y = elu_(v.y); v.z = elu_(v.z); v.w = elu_(v.w); }
        st4(y + r * C + c, v);
    }
}

// grid = (row blocks inside a segment, column blocks of 256 floats, segments); 256 threads = 64 float4 columns x 4 row lanes
__global__ __launch_bounds__(256) void bias_act_bwd_kernel(const float* __restrict__ g, int64_t ldg, const float* __restrict__ a, int64_t lda,
                                                           float* __restrict__ gy, float* __restrict__ db_part,
                                                           int C, int64_t rows_per_seg, int act, int nrb, int brows, AmaxOut amax) {
    __shared__ __attribute__((aligned(16))) float s_red[4][256];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.y * 256 + cl * 4;
    const int64_t seg = blockIdx.z;
    const int64_t r0 = (int64_t)blockIdx.x * brows;
    const int64_t r1 = r0 + brows < rows_per_seg ? r0 + brows : rows_per_seg;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float gmax = 0.f;
    if (c < C) {
        for (int64_t r = r0 + rl; r < r1; r += 4) {
            const int64_t o = (seg * rows_per_seg + r) * C + c;
            float4 v = ld4(g + (seg * rows_per_seg + r) * ldg + c);          // g may be a column block of a wider gradient (row stride ldg)
            if (act == 1) {
                const float4 av = ld4(a + (seg * rows_per_seg + r) * lda + c);   // so may the activation (a block of a row buffer)
                v.x *= av.x > 0.f ? 1.f : av.x + 1.f; v.y *= av.y > 0.f ? 1.f : av.y + 1.f;
                v.z *= av.z > 0.f ? 1.f : av.z + 1.f; v.w *= av.w > 0.f ? 1.f : av.w + 1.f;
            }
            if (gy != nullptr) st4(gy + o, v);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            gmax = amax4(gmax, v);
        }
    }
    amax_publish_wave(gmax, amax);
    if (db_part == nullptr) return;
    st4(&s_red[rl][cl * 4], acc);
    __syncthreads();
    if (rl == 0 && c < C) {
        float4 t = ld4(&s_red[0][cl * 4]);
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            const float4 q = ld4(&s_red[k][cl * 4]);
            t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
        }
        st4(db_part + ((seg * nrb + blockIdx.x) * (int64_t)C) + c, t);
    }
}

// ---- ensemble head: hidden layer tail + the width-1 output layer in the same passes -------------------------------
// forward, one wave per row r of segment e:  a = elu(y + b2[e]) (in place),  q[r] = sum_c a[c] * w3[e, c] + b3[e]
__global__ __launch_bounds__(256) void head_fwd_kernel(float* __restrict__ y, const float* __restrict__ b2, const float* __restrict__ w3,
                                                       const float* __restrict__ b3, float* __restrict__ q, int64_t rows, int H,
                                                       int64_t rows_per_seg) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int64_t e = r / rows_per_seg;
    float acc = 0.f;
    for (int c = lane * 4; c < H; c += 256) {
        float4 v = ld4(y + r * H + c);
        const float4 b = ld4(b2 + e * H + c), w = ld4(w3 + e * H + c);
        v.x = elu_(v.x + b.x); v.y = elu_(v.y + b.y); v.z = elu_(v.z + b.z); v.w = elu_(v.w + b.w);
        st4(y + r * H + c, v);
        acc += v.x * w.x + v.y * w.y + v.z * w.z + v.w * w.w;
    }
    acc = wave_sum(acc);
    if (lane == 0) q[r] = acc + (b3 ? b3[e] : 0.f);
}

// backward: gy[r, c] = gq[r] * w3[e, c] * elu'(a[r, c]);  partials of db2[e, c] = sum_r gy and dw3[e, c] = sum_r a * gq
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ gq, const float* __restrict__ a,
                                                       const float* __restrict__ w3, float* __restrict__ gy,
                                                       float* __restrict__ db_part, float* __restrict__ dw_part, int H,
                                                       int64_t rows_per_seg, int nrb, AmaxOut amax) {
    __shared__ __attribute__((aligned(16))) float s_red[2][4][256];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.y * 256 + cl * 4;
    const int64_t seg = blockIdx.z;
    const int64_t r0 = (int64_t)blockIdx.x * BWD_ROWS;
    const int64_t r1 = r0 + BWD_ROWS < rows_per_seg ? r0 + BWD_ROWS : rows_per_seg;
    float4 accb = make_float4(0.f, 0.f, 0.f, 0.f), accw = accb;
    float gmax = 0.f;
    if (c < H) {
        const float4 w = ld4(w3 + seg * H + c);
        for (int64_t r = r0 + rl; r < r1; r += 4) {
            const int64_t row = seg * rows_per_seg + r;
            const float g = gq[row];
            const float4 av = ld4(a + row * H + c);
            float4 v;
            v.x = g * w.x * (av.x > 0.f ? 1.f : av.x + 1.f); v.y = g * w.y * (av.y > 0.f ? 1.f : av.y + 1.f);
            v.z = g * w.z * (av.z > 0.f ? 1.f : av.z + 1.f); v.w = g * w.w * (av.w > 0.f ? 1.f : av.w + 1.f);
            st4(gy + row * H + c, v);
            gmax = amax4(gmax, v);
            accb.x += v.x; accb.y += v.y; accb.z += v.z; accb.w += v.w;
            accw.x += g * av.x; accw.y += g * av.y; accw.z += g * av.z; accw.w += g * av.w;
        }
    }
    amax_publish_wave(gmax, amax);
    st4(&s_red[0][rl][cl * 4], accb);
    st4(&s_red[1][rl][cl * 4], accw);
    __syncthreads();
    if (rl < 2 && c < H) {                           // row lane 0 finishes db2, row lane 1 finishes dw3
        float4 t = ld4(&s_red[rl][0][cl * 4]);
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            const float4 qv = ld4(&s_red[rl][k][cl * 4]);
            t.x += qv.x; t.y += qv.y; t.z += qv.z; t.w += qv.w;
        }
        st4((rl == 0 ? db_part : dw_part) + ((seg * nrb + blockIdx.x) * (int64_t)H) + c, t);
    }
}

inline int row_blocks(int64_t rows_per_seg) { return (int)((rows_per_seg + BWD_ROWS - 1) / BWD_ROWS); }
// rows per block of bias_act_bwd: 128.  (64 / 32 for gradients that give fewer than 4 blocks per CU - 257 blocks at configs[2]'s 32 832 x 256 -
// measured SLOWER on the whole update: 30.22 against 29.80 ms, same box, three runs each: more partial rows to write and to sum.)
inline int block_rows(int64_t, int, int64_t) { return BWD_ROWS; }
inline int row_blocks_of(int64_t rows_per_seg, int brows) { return (int)((rows_per_seg + brows - 1) / brows); }
inline bool shape_ok(int64_t rows, int C, int64_t rows_per_seg) {
    return rows > 0 && C > 0 && C % 4 == 0 && rows_per_seg > 0 && rows % rows_per_seg == 0;
}

}  // namespace

extern "C" int resel_bias_act_fwd(float* y, const float* bias, int64_t rows, int C, int64_t rows_per_seg, int act,
                                  resel_stream_t stream) {
    if (!y || !shape_ok(rows, C, rows_per_seg) || act < 0 || act > 1 || !aligned16(y) || (bias && !aligned16(bias))) return RESEL_EINVAL;
    const int64_t n4 = rows * (C / 4);
    const int64_t want = (n4 + 255) / 256;
    const unsigned grid = (unsigned)(want < 256 * 32 ? want : 256 * 32);      // <= 32 blocks per CU, grid-stride beyond
    hipLaunchKernelGGL(bias_act_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, y, bias, rows, C, rows_per_seg, act);
    return launch_status();
}

extern "C" size_t resel_bias_act_bwd_workspace_bytes(int64_t rows, int C, int64_t rows_per_seg) {
    if (!shape_ok(rows, C, rows_per_seg)) return 0;
    return (size_t)(rows / rows_per_seg) * row_blocks_of(rows_per_seg, block_rows(rows, C, rows_per_seg)) * C * sizeof(float);
}

extern "C" int resel_bias_act_bwd(const float* g, int64_t ldg, const float* a, int64_t lda, float* gy, float* dbias, void* workspace, int64_t rows, int C,
                                  int64_t rows_per_seg, int act, void* amax_gy, unsigned amax_epoch, resel_stream_t stream) {
    if (amax_gy && (reinterpret_cast<uintptr_t>(amax_gy) & 7u)) return RESEL_EINVAL;
    if (!g || !shape_ok(rows, C, rows_per_seg) || act < 0 || act > 1 || (act == 1 && (!a || !gy)) || (dbias && !workspace) || ldg < C || (ldg & 3))
        return RESEL_EINVAL;
    if (a && (lda < C || (lda & 3))) return RESEL_EINVAL;
    if (!aligned16(g) || (gy && !aligned16(gy)) || (a && !aligned16(a)) || (workspace && !aligned16(workspace))) return RESEL_EINVAL;
    const int brows = block_rows(rows, C, rows_per_seg);
    const int nseg = (int)(rows / rows_per_seg), nrb = row_blocks_of(rows_per_seg, brows);
    hipStream_t s = (hipStream_t)stream;
    float* part = dbias ? (float*)workspace : nullptr;
    hipLaunchKernelGGL(bias_act_bwd_kernel, dim3(nrb, (C + 255) / 256, nseg), dim3(256), 0, s, g, ldg, a, lda, gy, part, C, rows_per_seg, act, nrb, brows,
                       AmaxOut{(unsigned long long*)amax_gy, amax_epoch});
    if (dbias) launch_colsum(part, C, nrb, C, dbias, s, 1, 0, nseg);      // dbias[sg, :] = sum over the segment's row blocks
    return launch_status();
}

extern "C" int resel_ensemble_head_fwd(float* y, const float* b2, const float* w3, const float* b3, float* q, int64_t rows, int H,
                                       int64_t rows_per_seg, resel_stream_t stream) {
    if (!y || !b2 || !w3 || !q || !shape_ok(rows, H, rows_per_seg) || !aligned16(y) || !aligned16(b2) || !aligned16(w3)) return RESEL_EINVAL;
    hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, y, b2, w3, b3, q, rows, H,
                       rows_per_seg);
    return launch_status();
}

extern "C" size_t resel_ensemble_head_bwd_workspace_bytes(int64_t rows, int H, int64_t rows_per_seg) {
    return 2 * resel_bias_act_bwd_workspace_bytes(rows, H, rows_per_seg);
}

extern "C" int resel_ensemble_head_bwd(const float* gq, const float* a, const float* w3, float* gy, float* db2, float* dw3,
                                       void* workspace, int64_t rows, int H, int64_t rows_per_seg, void* amax_gy, unsigned amax_epoch,
                                       resel_stream_t stream) {
    if (amax_gy && (reinterpret_cast<uintptr_t>(amax_gy) & 7u)) return RESEL_EINVAL;
    if (!gq || !a || !w3 || !gy || !db2 || !dw3 || !workspace || !shape_ok(rows, H, rows_per_seg)) return RESEL_EINVAL;
    if (!aligned16(a) || !aligned16(w3) || !aligned16(gy) || !aligned16(workspace)) return RESEL_EINVAL;
    const int nseg = (int)(rows / rows_per_seg), nrb = row_blocks(rows_per_seg);
    hipStream_t s = (hipStream_t)stream;
    float* db_part = (float*)workspace;
    float* dw_part = db_part + (size_t)nseg * nrb * H;
    hipLaunchKernelGGL(head_bwd_kernel, dim3(nrb, (H + 255) / 256, nseg), dim3(256), 0, s, gq, a, w3, gy, db_part, dw_part, H, rows_per_seg, nrb,
                       AmaxOut{(unsigned long long*)amax_gy, amax_epoch});
    launch_colsum(db_part, H, nrb, H, db2, s, 1, 0, nseg);
    launch_colsum(dw_part, H, nrb, H, dw3, s, 1, 0, nseg);
    return launch_status();
}
