// Depthwise causal conv1d + bias + SiLU on the masked input, token-major [B*L, Di] - forward and backward.
//
// Memory-bound (2 floats of HBM traffic per element).  The K taps of a thread's channels live in registers (the tap
// count is padded to KT in {4, 8, 16, 32} with zero weights).  The zero left pad is per
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
    AmaxOut amax;                    // optional: publish max |y| (forward) / max |dx| (backward)
    const float* dy2 = nullptr;      // backward: optional second gradient of the output, summed with dy on load (the Mamba mixer's conv output
    int64_t ld_dy2 = 0;              // receives the scan's du AND the x_proj input gradient: no accumulating GEMM, no add pass)
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

typedef float f2 __attribute__((ext_vector_type(2)));

// Forward: no LDS.  A thread owns 2 adjacent channels and walks `tt` consecutive time steps of one row with the last KT
// inputs in a rotating register window (slot indices are compile-time: the step loop is unrolled KT-fold), so every
// input element is fetched once (plus the KT-1 step warm-up halo at the head of a time chunk) and every global access
// of a wave is one fully used 512-byte row segment; the two waves of a block sit side by side on the channel axis and
// blockIdx.x (fastest) continues it, so concurrently running blocks stream whole rows.  [The first version staged
// 64-channel x 64-step tiles in LDS: its 256-byte row pieces at a 4 KB token stride reached only 1.07 TB/s.]
// Loads run P steps ahead of their use in a small register ring (P = min(KT, 16): 16 against 8 ahead measured 66.5 against 69.5 us at K = 16).  The step groups are straight-line code with exactly
// one load and (after the warm-up group) one store per step: any conditional memory operation inside them would make
// the compiler fall back to s_waitcnt vmcnt(0) per step and serialise the ring on the load latency - hence the clamped
// (always valid) load addresses, the mask values pre-loaded for the whole chunk (one lane per step, fetched back with
// v_readlane) and the three instances of the group body (warm-up / full / ragged tail).
constexpr int CONV_TT = 64;          // time steps per chunk (<= 64: one mask register pair covers chunk + halo)

template <int KT, int MODE>          // MODE 0: warm-up group (only its last step produces an output), 1: full, 2: guarded tail
__device__ __forceinline__ void conv_fwd_group(const ConvParams& p, const float* xrow, float* yrow, int tg, int t_end,
                                               int ig, float mlo, float mhi, const f2 (&wr)[KT], f2 (&win)[KT],
                                               f2 (&pre)[KT < 16 ? KT : 16], f2 bv, float& ymax) {
    constexpr int P = KT < 16 ? KT : 16;
    const float mreg = ig < 64 ? mlo : mhi;          // KT divides 64: a group never straddles the two mask registers
#pragma unroll
    for (int s = 0; s < KT; ++s) {
        const int t = tg + s;
        const float m = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mreg), (ig & 63) + s));
        win[s] = pre[s % P] * f2{m, m};
        const int tl = min(max(t + P, 0), t_end - 1);                        // clamped: always a valid token of this row
        pre[s % P] = *reinterpret_cast<const f2*>(xrow + (int64_t)tl * p.ld_x);
        if (MODE == 1 || (MODE == 0 && s == KT - 1) || (MODE == 2 && t < t_end)) {
            f2 acc = bv;
#pragma unroll
            for (int kk = 0; kk < KT; ++kk)             // tap kk pairs with x[t - (KT-1) + kk] = slot (s + 1 + kk) mod KT
                acc = __builtin_elementwise_fma(wr[kk], win[(s + 1 + kk) % KT], acc);
            if (p.silu) { acc.x = siluf_(acc.x); acc.y = siluf_(acc.y); }
            *reinterpret_cast<f2*>(yrow + (int64_t)t * p.ld_y) = acc;
            ymax = fmaxf(ymax, fmaxf(__builtin_fabsf(acc.x), __builtin_fabsf(acc.y)));
        }
    }
}

template <int KT>
__global__ __launch_bounds__(128) void conv_fwd_kernel(ConvParams p) {
    constexpr int P = KT < 16 ? KT : 16;
    const int lane = threadIdx.x & 63;
    if ((blockIdx.x * 128 + (threadIdx.x & 64)) * 2 >= p.Di) return;      // whole wave past the last channel (no barriers below)
    // lanes past the last channel of a partly filled wave shadow the last pair (same loads, same stores, same values):
    // every lane stays active, which the v_readlane mask fetches and the exact vmcnt accounting rely on
    const int c = min((int)(blockIdx.x * 128 + threadIdx.x) * 2, p.Di - 2);
    const int b = blockIdx.z, t0 = blockIdx.y * CONV_TT;
    const int t_end = min(p.L, t0 + CONV_TT);
    const int64_t tok0 = (int64_t)b * p.L;
    const float* xrow = p.x + tok0 * p.ld_x + c;
    float* yrow = p.y + tok0 * p.ld_y + c;
    f2 wr[KT], win[KT];
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {
        const int k = kk - (KT - p.K);
        wr[kk] = f2{0.f, 0.f};
        win[kk] = f2{0.f, 0.f};
        if (k >= 0) wr[kk] = f2{p.w[(int64_t)c * p.K + k], p.w[(int64_t)(c + 1) * p.K + k]};
    }
    f2 bv = {0.f, 0.f};
    if (p.bias) bv = *reinterpret_cast<const f2*>(p.bias + c);
    const int ts = t0 - (KT - 1);                    // time of step 0 (warm-up starts in the left halo)
    // mask of step i = ts + i in lane i of (mlo, mhi); 0 left of the row start (zero padding) and right of the chunk
    float mlo, mhi;
    {
        const int ta = ts + lane, tb = ts + 64 + lane;
        const bool oka = ta >= 0 && ta < t_end, okb = tb >= 0 && tb < t_end;
        const float va = p.mask ? p.mask[tok0 + min(max(ta, 0), t_end - 1)] : 1.f;
        const float vb = p.mask ? p.mask[tok0 + min(max(tb, 0), t_end - 1)] : 1.f;
        mlo = oka ? va : 0.f;
        mhi = okb ? vb : 0.f;
    }
    f2 pre[P];
#pragma unroll
    for (int i = 0; i < P; ++i) pre[i] = *reinterpret_cast<const f2*>(xrow + (int64_t)min(max(ts + i, 0), t_end - 1) * p.ld_x);
    float ymax = 0.f;
    conv_fwd_group<KT, 0>(p, xrow, yrow, ts, t_end, 0, mlo, mhi, wr, win, pre, bv, ymax);
    int ig = KT;
    for (; ts + ig + KT <= t_end; ig += KT) conv_fwd_group<KT, 1>(p, xrow, yrow, ts + ig, t_end, ig, mlo, mhi, wr, win, pre, bv, ymax);
    if (ts + ig < t_end) conv_fwd_group<KT, 2>(p, xrow, yrow, ts + ig, t_end, ig, mlo, mhi, wr, win, pre, bv, ymax);
    amax_publish_wave(ymax, p.amax);
}

// Backward, register-window form (K <= 16, L >= 64): the same thread ownership as the forward - 2 channels, one 64-step
// chunk of one row - with FOUR rotating windows in registers: the last KT masked inputs, the last KT gate gradients g, the
// taps and the tap gradients.
//   g[t]   = dy[t] * silu'(pre[t])                      pre = conv output before the activation (recomputed)
//   dx[s]  = mask[s] * sum_k w[k] * g[s + (KT-1) - k]   (g = 0 at and beyond the row end)
//   dw[k]  = sum_t g[t] * xm[t - (KT-1) + k],  db = sum_t g[t]      over the steps t the chunk OWNS
// Step i of a chunk is time t = t0 - (KT+1) + i: KT steps of input warm-up, KT steps that start the g window, then TT / KT
// groups in which every step also emits dx[t - (KT-1)] - the warm-up length is chosen so that the dx phase starts on a
// group boundary, which leaves three straight-line group bodies (exact vmcnt accounting, see the forward).  The last
// chunk of a row is shifted back to end exactly at the row end (it recomputes a few dx of its neighbour, same values) and
// owns only its own steps.  Per-chunk dw / db partials are summed by colsum_kernel (fixed order, no atomics).
// The chunk length TT is one of 64 / 80 / 96 steps, whichever walks the fewest steps over the row (`bwd_chunk`): a chunk walks TT + 2 KT steps
// to own TT, and the shifted last chunk repeats what it overlaps - L = 1043, KT = 16: 17 x 96 = 1632 steps at 64, 11 x 128 = 1408 at 96
// (137 -> 117 us per launch; longer chunks leave too few waves for a latency-bound walk: 128 steps 137 us).
constexpr int CB_NM = 2;                        // mask registers: step i of a chunk walk in lane i % 64 of register i / 64 (TT + 2 KT <= 128)

template <int KT, int MODE, bool TWO, int CB_TT>          // MODE 0: inputs only, 1: + g window and dw / db, 2: + dx; TWO: dy + dy2
__device__ __forceinline__ void conv_bwd_group(const ConvParams& p, const float* xrow, const float* dyrow, const float* dy2row, float* dxrow, int tg,
                                               int ig, int t_own0, const float (&mk)[CB_NM], const f2 (&wr)[KT], f2 (&xwin)[KT],
                                               f2 (&gwin)[KT], f2 (&dwr)[KT], f2& dbr, f2 (&prex)[KT < 4 ? KT : 4],
                                               f2 (&predy)[KT < 4 ? KT : 4], f2 (&predy2)[KT < 4 ? KT : 4], f2 bv, float& dxmax) {
    constexpr int P = KT < 4 ? KT : 4;
    float mreg = mk[0];
#pragma unroll
    for (int q = 1; q < CB_NM; ++q) mreg = ig >= 64 * q ? mk[q] : mreg;
#pragma unroll
    for (int s = 0; s < KT; ++s) {
        const int t = tg + s;
        const float m = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mreg), (ig & 63) + s));
        xwin[s] = prex[s % P] * f2{m, m};
        const int tl = min(max(t + P, 0), p.L - 1);
        prex[s % P] = *reinterpret_cast<const f2*>(xrow + (int64_t)tl * p.ld_x);
        if (MODE >= 1) {
            f2 dyv = predy[s % P];
            predy[s % P] = *reinterpret_cast<const f2*>(dyrow + (int64_t)tl * p.ld_dy);
            if (TWO) {                                                         // a second load per step, as static as the first
                dyv += predy2[s % P];
                predy2[s % P] = *reinterpret_cast<const f2*>(dy2row + (int64_t)tl * p.ld_dy2);
            }
            f2 acc = bv;
#pragma unroll
            for (int kk = 0; kk < KT; ++kk) acc = __builtin_elementwise_fma(wr[kk], xwin[(s + 1 + kk) % KT], acc);
            const float live = (t >= 0 && t < p.L) ? 1.f : 0.f;                 // g is zero outside the row
            f2 g = dyv * f2{live, live};
            if (p.silu) { g.x *= dsiluf_(acc.x); g.y *= dsiluf_(acc.y); }
            gwin[s] = g;
            const float own = (t >= t_own0 && t < t_own0 + CB_TT) ? 1.f : 0.f;   // halo steps serve dx only
            const f2 go = g * f2{own, own};
            dbr += go;
#pragma unroll
            for (int kk = 0; kk < KT; ++kk) dwr[kk] = __builtin_elementwise_fma(go, xwin[(s + 1 + kk) % KT], dwr[kk]);
        }
        if (MODE == 2) {
            f2 acc = {0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KT; ++kk) acc = __builtin_elementwise_fma(wr[kk], gwin[(s - kk + KT) % KT], acc);
            const int im = ig + s - (KT - 1);                                    // step index whose time is t - (KT-1)
            float mm = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mk[0]), im & 63));
#pragma unroll
            for (int q = 1; q < CB_NM; ++q) {
                const float mq = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mk[q]), im & 63));
                mm = im >= 64 * q ? mq : mm;
            }
            const f2 dxv = acc * f2{mm, mm};
            *reinterpret_cast<f2*>(dxrow + (int64_t)(t - (KT - 1)) * p.ld_dx) = dxv;
            dxmax = fmaxf(dxmax, fmaxf(__builtin_fabsf(dxv.x), __builtin_fabsf(dxv.y)));
        }
    }
}

template <int KT, bool TWO, int CB_TT>
__global__ __launch_bounds__(128) void conv_bwd_win_kernel(ConvParams p, int nchunk) {
    static_assert(CB_TT % KT == 0 && CB_TT + 2 * KT <= 64 * CB_NM, "chunk length");
    constexpr int P = KT < 4 ? KT : 4;
    const int lane = threadIdx.x & 63;
    if ((blockIdx.x * 128 + (threadIdx.x & 64)) * 2 >= p.Di) return;      // whole wave past the last channel (no barriers below)
    const int c_raw = (int)(blockIdx.x * 128 + threadIdx.x) * 2;
    const int c = min(c_raw, p.Di - 2);             // shadow lanes (see the forward kernel)
    const int b = blockIdx.z, chunk = blockIdx.y;
    const int t_own0 = chunk * CB_TT;               // first step this chunk accounts for in dw / db
    const int t0 = min(t_own0, p.L - CB_TT);        // the last chunk is shifted back to end at the row end
    const int64_t tok0 = (int64_t)b * p.L;
    const float* xrow = p.x + tok0 * p.ld_x + c;
    const float* dyrow = p.dy + tok0 * p.ld_dy + c;
    const float* dy2row = TWO ? p.dy2 + tok0 * p.ld_dy2 + c : nullptr;
    float* dxrow = p.dx + tok0 * p.ld_dx + c;
    f2 wr[KT], xwin[KT], gwin[KT], dwr[KT];
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {
        const int k = kk - (KT - p.K);
        wr[kk] = f2{0.f, 0.f};
        xwin[kk] = gwin[kk] = dwr[kk] = f2{0.f, 0.f};
        if (k >= 0) wr[kk] = f2{p.w[(int64_t)c * p.K + k], p.w[(int64_t)(c + 1) * p.K + k]};
    }
    f2 bv = {0.f, 0.f}, dbr = {0.f, 0.f};
    if (p.bias) bv = *reinterpret_cast<const f2*>(p.bias + c);
    const int ts = t0 - (KT + 1);                   // time of step 0
    float mk[CB_NM];                                // mask of step i in lane i % 64 of mk[i / 64]; 0 outside the row
#pragma unroll
    for (int q = 0; q < CB_NM; ++q) {
        const int ta = ts + 64 * q + lane;
        const float va = p.mask ? p.mask[tok0 + min(max(ta, 0), p.L - 1)] : 1.f;
        mk[q] = (ta >= 0 && ta < p.L) ? va : 0.f;
    }
    f2 prex[P], predy[P], predy2[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        prex[i] = *reinterpret_cast<const f2*>(xrow + (int64_t)min(max(ts + i, 0), p.L - 1) * p.ld_x);
        predy[i] = *reinterpret_cast<const f2*>(dyrow + (int64_t)min(max(ts + KT + i, 0), p.L - 1) * p.ld_dy);
        predy2[i] = TWO ? *reinterpret_cast<const f2*>(dy2row + (int64_t)min(max(ts + KT + i, 0), p.L - 1) * p.ld_dy2) : f2{0.f, 0.f};
    }
    float dxmax = 0.f;
    conv_bwd_group<KT, 0, TWO, CB_TT>(p, xrow, dyrow, dy2row, dxrow, ts, 0, t_own0, mk, wr, xwin, gwin, dwr, dbr, prex, predy, predy2, bv, dxmax);
    conv_bwd_group<KT, 1, TWO, CB_TT>(p, xrow, dyrow, dy2row, dxrow, ts + KT, KT, t_own0, mk, wr, xwin, gwin, dwr, dbr, prex, predy, predy2, bv, dxmax);
    for (int ig = 2 * KT; ig < 2 * KT + CB_TT; ig += KT)
        conv_bwd_group<KT, 2, TWO, CB_TT>(p, xrow, dyrow, dy2row, dxrow, ts + ig, ig, t_own0, mk, wr, xwin, gwin, dwr, dbr, prex, predy, predy2, bv, dxmax);
    amax_publish_wave(dxmax, p.amax);
    if (c_raw < p.Di) {
        const int64_t row = (int64_t)b * nchunk + chunk;
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
            p.dw_part[(row * p.Di + c) * KT + kk] = dwr[kk].x;
            p.dw_part[(row * p.Di + c + 1) * KT + kk] = dwr[kk].y;
        }
        p.db_part[row * p.Di + c] = dbr.x;
        p.db_part[row * p.Di + c + 1] = dbr.y;
    }
}

// Backward: one block per (row b, channel tile) walks the time tiles, so dw / dbias accumulate in registers and
// only per-row partials [B, Di, KT] / [B, Di] leave the block (summed by colsum_kernel, no atomics).
//   g[t]   = dy[t] * silu'(pre[t])                      pre = conv output before the activation (recomputed)
//   dx[t]  = mask[t] * sum_k w[k] * g[t + (K-1) - k]
//   dw[k]  = sum_t g[t] * xm[t - (K-1) + k]
template <int KT>
__global__ __launch_bounds__(256) void conv_bwd_kernel(ConvParams p) {
    float dxmax = 0.f;
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
                if (p.dy2) {
                    const float4 g2 = ld4(p.dy2 + (tok0 + t) * p.ld_dy2 + d0 + tc4);
                    g.x += g2.x; g.y += g2.y; g.z += g2.z; g.w += g2.w;
                }
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
                dxmax = amax4(dxmax, acc);
            }
        }
    }
    amax_publish_wave(dxmax, p.amax);
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

inline int pad_taps(int K) { return K <= 4 ? 4 : (K <= 8 ? 8 : (K <= 16 ? 16 : 32)); }
inline bool conv_args_ok(const float* x, int64_t ld_x, const float* o, int64_t ld_o, int B, int L, int Di, int K) {
    return x && o && B > 0 && L > 0 && Di > 0 && Di % 4 == 0 && K >= 1 && K <= 32 && ld_x % 4 == 0 && ld_o % 4 == 0 &&
           aligned16(x) && aligned16(o);
}

}  // namespace

extern "C" int resel_causal_conv1d_fwd(const float* x, int64_t ld_x, const float* w, const float* bias, const float* mask,
                                       float* y, int64_t ld_y, int B, int L, int Di, int K, int silu,
                                       void* amax_y, unsigned amax_epoch, resel_stream_t stream) {
    if (amax_y && (reinterpret_cast<uintptr_t>(amax_y) & 7u)) return RESEL_EINVAL;
    if (!conv_args_ok(x, ld_x, y, ld_y, B, L, Di, K) || !w || (bias && !aligned16(bias))) return RESEL_EINVAL;
    ConvParams p{x, w, bias, mask, nullptr, y, nullptr, nullptr, nullptr, ld_x, ld_y, 0, 0, B, L, Di, K, silu,
                 AmaxOut{(unsigned long long*)amax_y, amax_epoch}};
    const int KT = pad_taps(K);
    dim3 grid((Di + 255) / 256, (L + CONV_TT - 1) / CONV_TT, B);
    hipStream_t s = (hipStream_t)stream;
    switch (KT) {
        case 4: launch_timed(RESEL_PROF_CONV_FWD, conv_fwd_kernel<4>, grid, dim3(128), 0, s, p); break;
        case 8: launch_timed(RESEL_PROF_CONV_FWD, conv_fwd_kernel<8>, grid, dim3(128), 0, s, p); break;
        case 16: launch_timed(RESEL_PROF_CONV_FWD, conv_fwd_kernel<16>, grid, dim3(128), 0, s, p); break;
        default: launch_timed(RESEL_PROF_CONV_FWD, conv_fwd_kernel<32>, grid, dim3(128), 0, s, p); break;
    }
    return launch_status();
}

// chunk length of the windowed backward for rows of L steps: the candidate that walks the fewest steps (ties: the shorter); 0: not windowed
inline int bwd_chunk(int L, int KT) {
    if (KT > 16 || L < 64) return 0;
    int best = 0;
    long best_steps = 0;
    for (int tt = 64; tt <= 96 && tt <= L; tt += 16) {
        const long steps = (long)((L + tt - 1) / tt) * (tt + 2 * KT);
        if (!best || steps < best_steps) { best = tt; best_steps = steps; }
    }
    return best;
}
inline int bwd_rows(int B, int L, int KT) {
    const int tt = bwd_chunk(L, KT);
    return tt ? B * ((L + tt - 1) / tt) : B;
}

extern "C" size_t resel_causal_conv1d_bwd_workspace_bytes(int B, int L, int Di, int K) {
    const int KT = pad_taps(K);
    return (size_t)bwd_rows(B, L, KT) * Di * (KT + 1) * sizeof(float);
}

extern "C" int resel_causal_conv1d_bwd(const float* x, int64_t ld_x, const float* w, const float* bias, const float* mask,
                                       const float* dy, int64_t ld_dy, float* dx, int64_t ld_dx, float* dw, float* dbias,
                                       void* workspace, int B, int L, int Di, int K, int silu, void* amax_dx, unsigned amax_epoch,
                                       resel_stream_t stream) {
    return resel_causal_conv1d_bwd2(x, ld_x, w, bias, mask, dy, ld_dy, nullptr, 0, dx, ld_dx, dw, dbias, workspace, B, L, Di, K, silu, amax_dx,
                                    amax_epoch, stream);
}

extern "C" int resel_causal_conv1d_bwd2(const float* x, int64_t ld_x, const float* w, const float* bias, const float* mask,
                                        const float* dy, int64_t ld_dy, const float* dy2, int64_t ld_dy2, float* dx, int64_t ld_dx,
                                        float* dw, float* dbias, void* workspace, int B, int L, int Di, int K, int silu,
                                        void* amax_dx, unsigned amax_epoch, resel_stream_t stream) {
    if (amax_dx && (reinterpret_cast<uintptr_t>(amax_dx) & 7u)) return RESEL_EINVAL;
    if (dy2 && (ld_dy2 % 4 || !aligned16(dy2))) return RESEL_EINVAL;
    if (!conv_args_ok(x, ld_x, dx, ld_dx, B, L, Di, K) || !w || !dy || !dw || !workspace || ld_dy % 4 || !aligned16(dy) ||
        (bias && !aligned16(bias)))
        return RESEL_EINVAL;
    const int KT = pad_taps(K);
    const int rows = bwd_rows(B, L, KT);            // partial rows: one per (b, chunk) or one per b
    float* dw_part = (float*)workspace;
    float* db_part = dw_part + (size_t)rows * Di * KT;
    ConvParams p{x, w, bias, mask, dy, nullptr, dx, dw_part, db_part, ld_x, 0, ld_dy, ld_dx, B, L, Di, K, silu,
                 AmaxOut{(unsigned long long*)amax_dx, amax_epoch}, dy2, ld_dy2};
    hipStream_t s = (hipStream_t)stream;
    if (const int tt = bwd_chunk(L, KT)) {
        const int nchunk = (L + tt - 1) / tt;
        dim3 grid((Di + 255) / 256, nchunk, B);
#define CONV_BWD_WIN_T(KTv, TTv) do { if (dy2) launch_timed(RESEL_PROF_CONV_BWD, conv_bwd_win_kernel<KTv, true, TTv>, grid, dim3(128), 0, s, p, nchunk); \
                                      else launch_timed(RESEL_PROF_CONV_BWD, conv_bwd_win_kernel<KTv, false, TTv>, grid, dim3(128), 0, s, p, nchunk); } while (0)
#define CONV_BWD_WIN(KTv) do { if (tt == 64) CONV_BWD_WIN_T(KTv, 64); else if (tt == 80) CONV_BWD_WIN_T(KTv, 80); else CONV_BWD_WIN_T(KTv, 96); } while (0)
        switch (KT) {
            case 4: CONV_BWD_WIN(4); break;
            case 8: CONV_BWD_WIN(8); break;
            default: CONV_BWD_WIN(16); break;
        }
#undef CONV_BWD_WIN
#undef CONV_BWD_WIN_T
    } else {
        dim3 grid(B, (Di + TILE_C - 1) / TILE_C);
        switch (KT) {
            case 4: launch_timed(RESEL_PROF_CONV_BWD, conv_bwd_kernel<4>, grid, dim3(256), 0, s, p); break;
            case 8: launch_timed(RESEL_PROF_CONV_BWD, conv_bwd_kernel<8>, grid, dim3(256), 0, s, p); break;
            case 16: launch_timed(RESEL_PROF_CONV_BWD, conv_bwd_kernel<16>, grid, dim3(256), 0, s, p); break;
            default: launch_timed(RESEL_PROF_CONV_BWD, conv_bwd_kernel<32>, grid, dim3(256), 0, s, p); break;
        }
    }
    launch_colsum(dw_part, (int64_t)Di * KT, rows, Di * KT, dw, s, KT, K);   // dw[d, k] = sum_rows dw_part[row, d, KT - K + k]
    if (dbias) launch_colsum(db_part, Di, rows, Di, dbias, s);
    return launch_status();
}
