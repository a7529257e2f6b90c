// Linear-recurrence scans for gilr (real, gated) and lru (complex diagonal), dense [B, L, C] - forward and backward.
//
// The reference runs these as strictly sequential Triton loops over L with one program per (b, 256 channels)
// (real_rnn_tie_input_gate.py:9-34, complex_rnn.py:44-87): at B = 16, T' = 2003 that is 16 programs.  Here the
// time axis itself is parallel: a workgroup owns (row b, 64 channels), its 16 waves own 16 contiguous time
// segments, lanes are channels (every access is a fully used 256-byte segment).  Pass 1 reduces each segment to
// the affine map h_out = a * h_in + c, the 16 maps are composed through LDS, pass 2 replays the segment from
// its true incoming state and writes the result.  The composition is exact algebra on the same fp32
// operations, start-resets are folded into the gate (f := 0), and nothing is atomic.
// Small batches (configs[4]: B = 16, C = 256 is 64 such workgroups on 256 CUs): the lanes of a wave are split CL channels x
// 64 / CL time segments (CL = 64, 32 or 16, chosen by the launcher so that (C / CL) * B workgroups fill the chip): a workgroup
// then owns CL channels and 16 * 64 / CL time segments; a wave access is 64 / CL row pieces of 4 CL bytes.
#include "resel_common.h"

namespace {
using namespace resel;

constexpr int NSEG = 16;            // waves (= time segments) per workgroup

__device__ __forceinline__ float tanhf_(float x) {
    const float e = fast_exp(-2.0f * fabsf(x));                 // tanh|x| = (1 - e) / (1 + e)
    const float t = (1.0f - e) * fast_rcp(1.0f + e);
    return copysignf(t, x);
}

struct Seg {
    int t0, t1;
};
template <int NST>
__device__ __forceinline__ Seg segment(int w, int L) {
    const int len = (L + NST - 1) / NST;
    Seg s;
    s.t0 = min(L, w * len);
    s.t1 = min(L, s.t0 + len);
    return s;
}

// keep_t = 1 - start_t of the row (1 everywhere without a start tensor, 0 at the sentinel slot L): staged once per
// workgroup in LDS.  Read from global inside the time loops it was a load + wait + branch per step, which serialised every
// wave on two dependent memory round trips per step (gilr forward: 130 us, 83 % of the wave time parked in s_waitcnt).
__device__ __forceinline__ void stage_keep(float* s_keep, const float* __restrict__ start, int b, int L) {
    for (int t = threadIdx.x; t <= L; t += blockDim.x) s_keep[t] = t < L ? (start ? 1.f - start[(int64_t)b * L + t] : 1.f) : 0.f;
    __syncthreads();
}

// ------------------------------------------------------------------------------------------ real (gilr)
__device__ __forceinline__ void gilr_gate(float vraw, float fraw, float keep, int act, float& v, float& fe) {
    v = act ? tanhf_(vraw) : vraw;
    fe = (act ? sigmoidf_(fraw) : fraw) * keep;
}

template <bool ACT, int CL>
__global__ __launch_bounds__(NSEG * 64) void linrec_real_fwd_kernel(const float* __restrict__ v, const float* __restrict__ f,
                                                                    int64_t ld_u, const float* __restrict__ start, const float* __restrict__ h0,
                                                                    float* __restrict__ h, int B, int L, int C, AmaxOut amax_h) {
    constexpr int act = ACT ? 1 : 0;
    constexpr int NST = NSEG * 64 / CL;                          // time segments of this workgroup
    __shared__ float s_a[NST][CL], s_c[NST][CL];
    extern __shared__ float s_keep[];
    const int lane = threadIdx.x & (CL - 1), w = (threadIdx.x >> 6) * (64 / CL) + ((threadIdx.x & 63) / CL);   // channel, time segment
    const int b = blockIdx.y, c = blockIdx.x * CL + lane;
    stage_keep(s_keep, start, b, L);
    const bool ok = c < C;
    const int64_t base = (int64_t)b * L * C + c, ubase = (int64_t)b * L * ld_u + c;     // h is dense, (v, f) rows are ld_u apart
    const Seg sg = segment<NST>(w, L);
    float a = 1.f, hl = 0.f;
    if (ok) {
#pragma unroll 8
        for (int t = sg.t0; t < sg.t1; ++t) {
            float vv, fe;
            gilr_gate(v[ubase + (int64_t)t * ld_u], f[ubase + (int64_t)t * ld_u], s_keep[t], act, vv, fe);
            hl = __builtin_fmaf(fe, hl - vv, vv);                 // f h + (1 - f) v
            a *= fe;
        }
    }
    s_a[w][lane] = a;
    s_c[w][lane] = hl;
    __syncthreads();
    float hin = (h0 && ok) ? h0[(int64_t)b * C + c] : 0.f;
    for (int ww = 0; ww < w; ++ww) hin = __builtin_fmaf(s_a[ww][lane], hin, s_c[ww][lane]);
    float hmax = 0.f;
    if (ok) {
        float hc = hin;
#pragma unroll 8
        for (int t = sg.t0; t < sg.t1; ++t) {
            float vv, fe;
            gilr_gate(v[ubase + (int64_t)t * ld_u], f[ubase + (int64_t)t * ld_u], s_keep[t], act, vv, fe);
            hc = __builtin_fmaf(fe, hc - vv, vv);
            h[base + (int64_t)t * C] = hc;
            hmax = fmaxf(hmax, __builtin_fabsf(hc));
        }
    }
    amax_publish_wave(hmax, amax_h);
}

// g_t = dh_t + f_{t+1} g_{t+1} ;  dv_t = g_t (1 - f_t) ;  df_t = g_t (h_{t-1} - v_t)   (then through tanh / sigmoid)
template <bool ACT, int CL>
__global__ __launch_bounds__(NSEG * 64) void linrec_real_bwd_kernel(const float* __restrict__ v, const float* __restrict__ f,
                                                                    int64_t ld_u, const float* __restrict__ start, const float* __restrict__ h0,
                                                                    const float* __restrict__ h, const float* __restrict__ dh,
                                                                    float* __restrict__ dv, float* __restrict__ df, int64_t ld_du,
                                                                    int B, int L, int C, AmaxOut amax_du) {
    constexpr int act = ACT ? 1 : 0;
    constexpr int NST = NSEG * 64 / CL;                          // time segments of this workgroup
    __shared__ float s_a[NST][CL], s_c[NST][CL];
    extern __shared__ float s_keep[];
    const int lane = threadIdx.x & (CL - 1), w = (threadIdx.x >> 6) * (64 / CL) + ((threadIdx.x & 63) / CL);   // channel, time segment
    const int b = blockIdx.y, c = blockIdx.x * CL + lane;
    stage_keep(s_keep, start, b, L);
    const bool ok = c < C;
    const int64_t base = (int64_t)b * L * C + c, ubase = (int64_t)b * L * ld_u + c, dbase = (int64_t)b * L * ld_du + c;
    const Seg sg = segment<NST>(w, L);
    auto gate_f = [&](int t) -> float {                          // effective gate f_t (0 beyond the row end: keep sentinel)
        const float fr = f[ubase + (int64_t)min(t, L - 1) * ld_u];
        return (act ? sigmoidf_(fr) : fr) * s_keep[t];
    };
    float a = 1.f, gl = 0.f;
    if (ok && sg.t1 > sg.t0) {
        float fnext = gate_f(sg.t1);
#pragma unroll 8
        for (int t = sg.t1 - 1; t >= sg.t0; --t) {
            gl = __builtin_fmaf(fnext, gl, dh[base + (int64_t)t * C]);
            a *= fnext;
            fnext = gate_f(t);
        }
    }
    s_a[w][lane] = a;
    s_c[w][lane] = gl;
    __syncthreads();
    float dmax = 0.f;
    float gin = 0.f;                                              // g just right of this segment
    for (int ww = NST - 1; ww > w; --ww) gin = __builtin_fmaf(s_a[ww][lane], gin, s_c[ww][lane]);
    if (ok && sg.t1 > sg.t0) {
        float g = gin;
        float fnext = gate_f(sg.t1);
        const float hzero = h0 ? h0[(int64_t)b * C + c] : 0.f;
#pragma unroll 8
        for (int t = sg.t1 - 1; t >= sg.t0; --t) {
            g = __builtin_fmaf(fnext, g, dh[base + (int64_t)t * C]);
            const float keep = s_keep[t];
            const float vraw = v[ubase + (int64_t)t * ld_u], fraw = f[ubase + (int64_t)t * ld_u];
            const float vv = act ? tanhf_(vraw) : vraw;
            const float sg_ = act ? sigmoidf_(fraw) : fraw;
            const float fe = sg_ * keep;
            const float hload = h[base + (int64_t)max(t - 1, 0) * C];      // clamped address: no branch around the load
            const float hprev = t > 0 ? hload : hzero;
            float gv = g * (1.f - fe);
            float gf = g * (hprev - vv) * keep;
            if (act) {
                gv *= (1.f - vv * vv);
                gf *= sg_ * (1.f - sg_);
            }
            dv[dbase + (int64_t)t * ld_du] = gv;
            df[dbase + (int64_t)t * ld_du] = gf;
            dmax = fmaxf(dmax, fmaxf(__builtin_fabsf(gv), __builtin_fabsf(gf)));
            fnext = fe;
        }
    }
    amax_publish_wave(dmax, amax_du);
}

// --------------------------------------------------------------------------------------- complex (lru)
template <int CL>
__global__ __launch_bounds__(NSEG * 64) void linrec_complex_fwd_kernel(const float* __restrict__ vr, const float* __restrict__ vi,
                                                                       const float* __restrict__ lam_re, const float* __restrict__ lam_im,
                                                                       const float* __restrict__ gamma, const float* __restrict__ start,
                                                                       const float* __restrict__ h0r, const float* __restrict__ h0i,
                                                                       float* __restrict__ hr, float* __restrict__ hi, int B, int L, int C,
                                                                       int64_t ld_u, AmaxOut amax_h) {
    constexpr int NST = NSEG * 64 / CL;
    __shared__ float s_ar[NST][CL], s_ai[NST][CL], s_cr[NST][CL], s_ci[NST][CL];
    extern __shared__ float s_keep[];
    const int lane = threadIdx.x & (CL - 1), w = (threadIdx.x >> 6) * (64 / CL) + ((threadIdx.x & 63) / CL);   // channel, time segment
    const int b = blockIdx.y, c = blockIdx.x * CL + lane;
    stage_keep(s_keep, start, b, L);
    const bool ok = c < C;
    const int64_t base = (int64_t)b * L * C + c, ubase = (int64_t)b * L * ld_u + c;
    const Seg sg = segment<NST>(w, L);
    const float lr = ok ? lam_re[c] : 0.f, li = ok ? lam_im[c] : 0.f, gm = (ok && gamma) ? gamma[c] : 1.f;
    float ar = 1.f, ai = 0.f, cr = 0.f, ci = 0.f;
    if (ok) {
#pragma unroll 8
        for (int t = sg.t0; t < sg.t1; ++t) {
            const float keep = s_keep[t];
            const float fr = lr * keep, fi = li * keep;
            const float xr = gm * vr[ubase + (int64_t)t * ld_u], xi = gm * vi[ubase + (int64_t)t * ld_u];
            const float nr = cr * fr - ci * fi + xr, ni = cr * fi + ci * fr + xi;
            cr = nr; ci = ni;
            const float pr = ar * fr - ai * fi, pi = ar * fi + ai * fr;
            ar = pr; ai = pi;
        }
    }
    s_ar[w][lane] = ar; s_ai[w][lane] = ai; s_cr[w][lane] = cr; s_ci[w][lane] = ci;
    __syncthreads();
    float xr0 = (h0r && ok) ? h0r[(int64_t)b * C + c] : 0.f, xi0 = (h0i && ok) ? h0i[(int64_t)b * C + c] : 0.f;
    for (int ww = 0; ww < w; ++ww) {
        const float pr = s_ar[ww][lane], pi = s_ai[ww][lane];
        const float nr = pr * xr0 - pi * xi0 + s_cr[ww][lane], ni = pr * xi0 + pi * xr0 + s_ci[ww][lane];
        xr0 = nr; xi0 = ni;
    }
    float hmax = 0.f;
    if (ok) {
        float cr2 = xr0, ci2 = xi0;
#pragma unroll 8
        for (int t = sg.t0; t < sg.t1; ++t) {
            const float keep = s_keep[t];
            const float fr = lr * keep, fi = li * keep;
            const float xr = gm * vr[ubase + (int64_t)t * ld_u], xi = gm * vi[ubase + (int64_t)t * ld_u];
            const float nr = cr2 * fr - ci2 * fi + xr, ni = cr2 * fi + ci2 * fr + xi;
            cr2 = nr; ci2 = ni;
            hr[base + (int64_t)t * C] = cr2;
            hi[base + (int64_t)t * C] = ci2;
            hmax = fmaxf(hmax, fmaxf(__builtin_fabsf(cr2), __builtin_fabsf(ci2)));
        }
    }
    amax_publish_wave(hmax, amax_h);
}

// g_t = dh_t + conj(f_{t+1}) g_{t+1} ; dv = gamma g ; dgamma += Re(g conj(v_raw)) ; dlambda += (1 - s_t) g conj(h_{t-1})
template <int CL>
__global__ __launch_bounds__(NSEG * 64) void linrec_complex_bwd_kernel(const float* __restrict__ vr, const float* __restrict__ vi,
                                                                       const float* __restrict__ lam_re, const float* __restrict__ lam_im,
                                                                       const float* __restrict__ gamma, const float* __restrict__ start,
                                                                       const float* __restrict__ h0r, const float* __restrict__ h0i,
                                                                       const float* __restrict__ hr, const float* __restrict__ hi,
                                                                       const float* __restrict__ dhr, const float* __restrict__ dhi,
                                                                       float* __restrict__ dvr, float* __restrict__ dvi,
                                                                       float* __restrict__ part, int B, int L, int C, int64_t ld_u, int64_t ld_du) {
    constexpr int NST = NSEG * 64 / CL;
    __shared__ float s_ar[NST][CL], s_ai[NST][CL], s_cr[NST][CL], s_ci[NST][CL];
    extern __shared__ float s_keep[];
    const int lane = threadIdx.x & (CL - 1), w = (threadIdx.x >> 6) * (64 / CL) + ((threadIdx.x & 63) / CL);   // channel, time segment
    const int b = blockIdx.y, c = blockIdx.x * CL + lane;
    stage_keep(s_keep, start, b, L);
    const bool ok = c < C;
    const int64_t base = (int64_t)b * L * C + c, ubase = (int64_t)b * L * ld_u + c, dbase = (int64_t)b * L * ld_du + c;
    const Seg sg = segment<NST>(w, L);
    const float lr = ok ? lam_re[c] : 0.f, li = ok ? lam_im[c] : 0.f, gm = (ok && gamma) ? gamma[c] : 1.f;
    auto keep_at = [&](int t) -> float { return s_keep[t]; };          // sentinel slot L holds 0
    float ar = 1.f, ai = 0.f, gr = 0.f, gi = 0.f;
    if (ok && sg.t1 > sg.t0) {
        float kn = keep_at(sg.t1);
#pragma unroll 8
        for (int t = sg.t1 - 1; t >= sg.t0; --t) {
            const float fr = lr * kn, fi = -li * kn;                 // conj(f_{t+1})
            const float nr = gr * fr - gi * fi + dhr[base + (int64_t)t * C], ni = gr * fi + gi * fr + dhi[base + (int64_t)t * C];
            gr = nr; gi = ni;
            const float pr = ar * fr - ai * fi, pi = ar * fi + ai * fr;
            ar = pr; ai = pi;
            kn = keep_at(t);
        }
    }
    s_ar[w][lane] = ar; s_ai[w][lane] = ai; s_cr[w][lane] = gr; s_ci[w][lane] = gi;
    __syncthreads();
    float xr0 = 0.f, xi0 = 0.f;
    for (int ww = NST - 1; ww > w; --ww) {
        const float pr = s_ar[ww][lane], pi = s_ai[ww][lane];
        const float nr = pr * xr0 - pi * xi0 + s_cr[ww][lane], ni = pr * xi0 + pi * xr0 + s_ci[ww][lane];
        xr0 = nr; xi0 = ni;
    }
    float dlr = 0.f, dli = 0.f, dgm = 0.f;
    if (ok && sg.t1 > sg.t0) {
        float g_r = xr0, g_i = xi0;
        float kn = keep_at(sg.t1);
        const float h0r_ = h0r ? h0r[(int64_t)b * C + c] : 0.f, h0i_ = h0i ? h0i[(int64_t)b * C + c] : 0.f;
#pragma unroll 8
        for (int t = sg.t1 - 1; t >= sg.t0; --t) {
            const float fr = lr * kn, fi = -li * kn;
            const float nr = g_r * fr - g_i * fi + dhr[base + (int64_t)t * C], ni = g_r * fi + g_i * fr + dhi[base + (int64_t)t * C];
            g_r = nr; g_i = ni;
            const float kt = keep_at(t);
            const float plr = hr[base + (int64_t)max(t - 1, 0) * C], pli = hi[base + (int64_t)max(t - 1, 0) * C];
            const float pr = t > 0 ? plr : h0r_, pi = t > 0 ? pli : h0i_;
            dlr += kt * (g_r * pr + g_i * pi);                      // d/d lam_re : g . h_prev
            dli += kt * (g_i * pr - g_r * pi);                      // d/d lam_im
            dgm += g_r * vr[ubase + (int64_t)t * ld_u] + g_i * vi[ubase + (int64_t)t * ld_u];
            dvr[dbase + (int64_t)t * ld_du] = gm * g_r;
            dvi[dbase + (int64_t)t * ld_du] = gm * g_i;
            kn = kt;
        }
    }
    // per-(b, segment) partials of the per-channel parameter gradients: [B, NST, 3, C]
    if (ok) {
        float* o = part + (((int64_t)b * NST + w) * 3) * C + c;
        o[0] = dlr; o[(int64_t)C] = dli; o[2 * (int64_t)C] = dgm;
    }
}

// lru's per-channel parameters (reference lru.py:104-110): rows of p_log = (nu_log, theta_log, gamma_log) ->
//   lambda = exp(-exp(nu_log)) (cos, sin)(exp(theta_log)), gamma = exp(gamma_log)            out rows = (lam_re, lam_im, gamma)
// One launch each way instead of ~10 element-wise launches on 256-element tensors forward and ~15 backward.
__global__ void lru_params_fwd_kernel(const float* __restrict__ p_log, float* __restrict__ out, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float nu = expf(p_log[c]), th = expf(p_log[C + c]), gm = expf(p_log[2 * C + c]);
    const float mag = expf(-nu);
    out[c] = mag * cosf(th);
    out[C + c] = mag * sinf(th);
    out[2 * C + c] = gm;
}
__global__ void lru_params_bwd_kernel(const float* __restrict__ p_log, const float* __restrict__ dout, float* __restrict__ dp, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float nu = expf(p_log[c]), th = expf(p_log[C + c]), gm = expf(p_log[2 * C + c]);
    const float mag = expf(-nu), cs = cosf(th), sn = sinf(th);
    const float dlr = dout[c], dli = dout[C + c], dg = dout[2 * C + c];
    dp[c] = -(dlr * cs + dli * sn) * mag * nu;                   // d mag / d nu = -mag, d nu / d nu_log = nu
    dp[C + c] = (dli * cs - dlr * sn) * mag * th;                // d theta / d theta_log = theta
    dp[2 * C + c] = dg * gm;
}

// channels per wave: the widest split whose grid still fills the chip
inline int pick_cl(int B, int C) {
    if ((int64_t)((C + 63) / 64) * B >= 256) return 64;
    if ((int64_t)((C + 31) / 32) * B >= 256) return 32;
    return 16;
}
#define LINREC_DISPATCH(CLV, CALL) do { if ((CLV) == 64) { constexpr int CL = 64; CALL; } else if ((CLV) == 32) { constexpr int CL = 32; CALL; } \
                                        else { constexpr int CL = 16; CALL; } } while (0)

}  // namespace

// (v, f) / (vr, vi) and their gradients may be column blocks of a wider token-major matrix - the [M, E C] output of a shared-input
// EnsembleLinear, whose [E, B, T, C] view the layers slice: ld_u / ld_du = floats between consecutive tokens (>= C, multiples of 4 not needed:
// every access is a 4-byte lane access).  h / dh are dense [B, L, C].
inline bool amax_arg_ok(const void* a) { return !a || !(reinterpret_cast<uintptr_t>(a) & 7u); }

extern "C" int resel_linrec_real_fwd(const float* v, const float* f, int64_t ld_u, const float* start, const float* h0, float* h,
                                     int B, int L, int C, int fuse_act, void* amax_h, unsigned amax_epoch, resel_stream_t stream) {
    if (!v || !f || !h || B <= 0 || L <= 0 || C <= 0 || ld_u < C || !amax_arg_ok(amax_h)) return RESEL_EINVAL;
    const int cl = pick_cl(B, C);
    const dim3 grid((C + cl - 1) / cl, B), blk(NSEG * 64);
    const size_t lds = (size_t)(L + 1) * sizeof(float);          // keep table (dynamic LDS)
    hipStream_t s = (hipStream_t)stream;
    const AmaxOut ao{(unsigned long long*)amax_h, amax_epoch};
    if (fuse_act) LINREC_DISPATCH(cl, (launch_timed(RESEL_PROF_LINREC_REAL_FWD, linrec_real_fwd_kernel<true, CL>, grid, blk, lds, s, v, f, ld_u, start, h0, h, B, L, C, ao)));
    else LINREC_DISPATCH(cl, (launch_timed(RESEL_PROF_LINREC_REAL_FWD, linrec_real_fwd_kernel<false, CL>, grid, blk, lds, s, v, f, ld_u, start, h0, h, B, L, C, ao)));
    return launch_status();
}

extern "C" int resel_linrec_real_bwd(const float* v, const float* f, int64_t ld_u, const float* start, const float* h0, const float* h,
                                     const float* dh, float* dv, float* df, int64_t ld_du, int B, int L, int C, int fuse_act,
                                     void* amax_du, unsigned amax_epoch, resel_stream_t stream) {
    if (!v || !f || !h || !dh || !dv || !df || B <= 0 || L <= 0 || C <= 0 || ld_u < C || ld_du < C || !amax_arg_ok(amax_du)) return RESEL_EINVAL;
    const int cl = pick_cl(B, C);
    const dim3 grid((C + cl - 1) / cl, B), blk(NSEG * 64);
    const size_t lds = (size_t)(L + 1) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    const AmaxOut ao{(unsigned long long*)amax_du, amax_epoch};
    if (fuse_act) LINREC_DISPATCH(cl, (launch_timed(RESEL_PROF_LINREC_REAL_BWD, linrec_real_bwd_kernel<true, CL>, grid, blk, lds, s, v, f, ld_u, start, h0, h, dh, dv, df, ld_du, B, L, C, ao)));
    else LINREC_DISPATCH(cl, (launch_timed(RESEL_PROF_LINREC_REAL_BWD, linrec_real_bwd_kernel<false, CL>, grid, blk, lds, s, v, f, ld_u, start, h0, h, dh, dv, df, ld_du, B, L, C, ao)));
    return launch_status();
}

extern "C" int resel_linrec_complex_fwd(const float* vr, const float* vi, int64_t ld_u, const float* lam_re, const float* lam_im,
                                        const float* gamma, const float* start, const float* h0r, const float* h0i,
                                        float* hr, float* hi, int B, int L, int C, void* amax_h, unsigned amax_epoch, resel_stream_t stream) {
    if (!vr || !vi || !lam_re || !lam_im || !hr || !hi || B <= 0 || L <= 0 || C <= 0 || ld_u < C || !amax_arg_ok(amax_h)) return RESEL_EINVAL;
    const int cl = pick_cl(B, C);
    const AmaxOut ao{(unsigned long long*)amax_h, amax_epoch};
    LINREC_DISPATCH(cl, (launch_timed(RESEL_PROF_LINREC_COMPLEX_FWD, linrec_complex_fwd_kernel<CL>, dim3((C + cl - 1) / cl, B), dim3(NSEG * 64),
                                      (size_t)(L + 1) * sizeof(float), (hipStream_t)stream, vr, vi, lam_re, lam_im, gamma, start, h0r, h0i, hr, hi, B, L, C, ld_u, ao)));
    return launch_status();
}

extern "C" size_t resel_linrec_complex_bwd_workspace_bytes(int B, int L, int C) {
    (void)L;
    return (size_t)B * (NSEG * 4) * 3 * C * sizeof(float);      // up to 64 time segments per row (CL = 16)
}

extern "C" int resel_linrec_complex_bwd(const float* vr, const float* vi, int64_t ld_u, const float* lam_re, const float* lam_im,
                                        const float* gamma, const float* start, const float* h0r, const float* h0i,
                                        const float* hr, const float* hi, const float* dhr, const float* dhi,
                                        float* dvr, float* dvi, int64_t ld_du, float* dlam_re, float* dlam_im, float* dgamma,
                                        void* workspace, int B, int L, int C, resel_stream_t stream) {
    if (!vr || !vi || !lam_re || !lam_im || !hr || !hi || !dhr || !dhi || !dvr || !dvi || !dlam_re || !dlam_im || !workspace ||
        B <= 0 || L <= 0 || C <= 0 || ld_u < C || ld_du < C)
        return RESEL_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int cl = pick_cl(B, C), nst = NSEG * 64 / cl;
    LINREC_DISPATCH(cl, (launch_timed(RESEL_PROF_LINREC_COMPLEX_BWD, linrec_complex_bwd_kernel<CL>, dim3((C + cl - 1) / cl, B), dim3(NSEG * 64),
                                      (size_t)(L + 1) * sizeof(float), s, vr, vi, lam_re, lam_im, gamma, start, h0r, h0i, hr, hi, dhr, dhi, dvr, dvi,
                                      (float*)workspace, B, L, C, ld_u, ld_du)));
    // per-(row, segment) partials [B * nst][3][C] -> d lambda_re, d lambda_im, d gamma (fixed summation order)
    const float* part = (const float*)workspace;
    launch_colsum(part, 3 * (int64_t)C, B * nst, C, dlam_re, s);
    launch_colsum(part + C, 3 * (int64_t)C, B * nst, C, dlam_im, s);
    if (dgamma) launch_colsum(part + 2 * (int64_t)C, 3 * (int64_t)C, B * nst, C, dgamma, s);
    return launch_status();
}

extern "C" int resel_lru_params_fwd(const float* params_log, float* out, int C, resel_stream_t stream) {
    if (!params_log || !out || C <= 0) return RESEL_EINVAL;
    hipLaunchKernelGGL(lru_params_fwd_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, params_log, out, C);
    return launch_status();
}

extern "C" int resel_lru_params_bwd(const float* params_log, const float* dout, float* dparams_log, int C, resel_stream_t stream) {
    if (!params_log || !dout || !dparams_log || C <= 0) return RESEL_EINVAL;
    hipLaunchKernelGGL(lru_params_bwd_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, params_log, dout, dparams_log, C);
    return launch_status();
}
