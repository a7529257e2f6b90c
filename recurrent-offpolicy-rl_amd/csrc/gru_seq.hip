// GRU recurrence on a hoisted input projection (gi = x W_ih^T + b_ih), forward and backward.
//
// The recurrence is T-sequential and each step needs every hidden unit of the previous step, so a step is a
// grid-wide dependency.  On MI355X a dependent kernel boundary (~1.5 us) is cheaper than a software grid
// barrier (4-7 us, MI355X_MICROARCH "barrier-xcd"), so every step is ONE small launch enqueued back-to-back
// from C (no Python in the loop): grid = (H/16 hidden-unit slices) x (B/16 row groups).  A block keeps its
// 16-unit slice of W_hh (pre-laid-out once per call as [slice][k/4][48][4] so that one ds_read_b128 yields 4
// consecutive k for one gate column) and its 16 rows of h_{t-1} in LDS and forms the three gate dot products
// with fp32 FMAs (exact fp32, no reduced-precision path: parity target is 1e-4 against ATen's CPU GRU).
// Neither roofline is tight for this layer; the reported figure is the achieved step rate.
#include "resel_common.h"

namespace {
using namespace resel;

constexpr int US = 16;    // hidden units per block
constexpr int RG = 16;    // batch rows per block
constexpr int HMAX = 256; // LDS budget of the backward step (s_dgh 16 x 3H + W slice 3H x 16 floats)

// wf[s][k4][g*16 + j][kk] = w_hh[(g*H + s*16 + j) * H + 4*k4 + kk]
__global__ void gru_layout_fwd_kernel(const float* __restrict__ w_hh, float* __restrict__ wf, int H) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)3 * H * H) return;
    const int kk = i & 3;
    int64_t r = i >> 2;
    const int col = r % 48; r /= 48;
    const int k4 = r % (H / 4);
    const int s = r / (H / 4);
    const int g = col / 16, j = col % 16;
    wf[i] = w_hh[((int64_t)g * H + s * 16 + j) * H + 4 * k4 + kk];
}
// wb[s][g4][j][gg] = w_hh[(4*g4 + gg) * H + s*16 + j]      (g4 over 3H/4 gate rows)
__global__ void gru_layout_bwd_kernel(const float* __restrict__ w_hh, float* __restrict__ wb, int H) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)3 * H * H) return;
    const int gg = i & 3;
    int64_t r = i >> 2;
    const int j = r % 16; r /= 16;
    const int g4 = r % (3 * H / 4);
    const int s = r / (3 * H / 4);
    wb[i] = w_hh[((int64_t)4 * g4 + gg) * H + s * 16 + j];
}

struct GruFwd {
    const float *gi, *wf, *b_hh, *h0;
    float *h_all, *gates;
    int B, L, H, t;
};

__global__ __launch_bounds__(256) void gru_fwd_step_kernel(GruFwd p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_w = smem;                              // [H/4][48][4]
    float* s_h = smem + (size_t)p.H * 48;           // [RG][H]
    const int tid = threadIdx.x;
    const int s = blockIdx.x, b0 = blockIdx.y * RG;
    const int H = p.H;
    // stage the weight slice (contiguous 48*H floats) and the 16 rows of h_{t-1}
    const float4* wsrc = reinterpret_cast<const float4*>(p.wf + (size_t)s * 48 * H);
    for (int i = tid; i < 12 * H; i += 256) reinterpret_cast<float4*>(s_w)[i] = wsrc[i];
    for (int i = tid; i < RG * H / 4; i += 256) {
        const int r = i / (H / 4), c4 = (i % (H / 4)) * 4;
        const int b = b0 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b < p.B) {
            if (p.t > 0) v = ld4(p.h_all + ((int64_t)b * p.L + (p.t - 1)) * H + c4);
            else if (p.h0) v = ld4(p.h0 + (int64_t)b * H + c4);
        }
        st4(&s_h[r * H + c4], v);
    }
    __syncthreads();
    const int r = tid >> 4, j = tid & 15;
    const int b = b0 + r, u = s * US + j;
    float ar = 0.f, az = 0.f, an = 0.f;
    for (int k4 = 0; k4 < H / 4; ++k4) {
        const float4 hv = ld4(&s_h[r * H + 4 * k4]);
        const float4 wr = ld4(&s_w[((size_t)k4 * 48 + j) * 4]);
        const float4 wz = ld4(&s_w[((size_t)k4 * 48 + 16 + j) * 4]);
        const float4 wn = ld4(&s_w[((size_t)k4 * 48 + 32 + j) * 4]);
        ar = __builtin_fmaf(hv.x, wr.x, ar); ar = __builtin_fmaf(hv.y, wr.y, ar); ar = __builtin_fmaf(hv.z, wr.z, ar); ar = __builtin_fmaf(hv.w, wr.w, ar);
        az = __builtin_fmaf(hv.x, wz.x, az); az = __builtin_fmaf(hv.y, wz.y, az); az = __builtin_fmaf(hv.z, wz.z, az); az = __builtin_fmaf(hv.w, wz.w, az);
        an = __builtin_fmaf(hv.x, wn.x, an); an = __builtin_fmaf(hv.y, wn.y, an); an = __builtin_fmaf(hv.z, wn.z, an); an = __builtin_fmaf(hv.w, wn.w, an);
    }
    if (b < p.B) {
        const int64_t tok = (int64_t)b * p.L + p.t;
        const float* g = p.gi + tok * 3 * H;
        const float rg = 1.f / (1.f + expf(-(g[u] + ar + p.b_hh[u])));
        const float zg = 1.f / (1.f + expf(-(g[H + u] + az + p.b_hh[H + u])));
        const float hn = an + p.b_hh[2 * H + u];
        const float ng = tanhf(g[2 * H + u] + rg * hn);
        const float hp = s_h[r * H + u];
        p.h_all[tok * H + u] = (1.f - zg) * ng + zg * hp;
        if (p.gates) {
            float* o = p.gates + tok * 4 * H;
            o[u] = rg; o[H + u] = zg; o[2 * H + u] = ng; o[3 * H + u] = hn;
        }
    }
}

struct GruBwd {
    const float *wb, *h0, *h_all, *gates, *dh_all, *carry_in;
    float *carry_out, *dgi, *dgh;
    int B, L, H, t;
};

__global__ __launch_bounds__(256) void gru_bwd_step_kernel(GruBwd p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = p.H;
    float* s_w = smem;                              // [3H/4][16][4]
    float* s_g = smem + (size_t)3 * H * 16;         // [RG][3H]
    float* s_dz = s_g + (size_t)RG * 3 * H;         // [RG][16]  dh * z for the owned units
    const int tid = threadIdx.x;
    const int s = blockIdx.x, b0 = blockIdx.y * RG;
    const float4* wsrc = reinterpret_cast<const float4*>(p.wb + (size_t)s * 48 * H);
    for (int i = tid; i < 12 * H; i += 256) reinterpret_cast<float4*>(s_w)[i] = wsrc[i];
    // gate gradients for the 16 rows x all H units (recomputed by every unit slice of the row group: elementwise)
    for (int i = tid; i < RG * H; i += 256) {
        const int r = i / H, u = i % H;
        const int b = b0 + r;
        float dr_ = 0.f, dz_ = 0.f, dn_ = 0.f, dhn = 0.f, dhz = 0.f;
        if (b < p.B) {
            const int64_t tok = (int64_t)b * p.L + p.t;
            const float dh = p.dh_all[tok * H + u] + p.carry_in[(int64_t)b * H + u];
            const float* g = p.gates + tok * 4 * H;
            const float rg = g[u], zg = g[H + u], ng = g[2 * H + u], hn = g[3 * H + u];
            const float hp = p.t > 0 ? p.h_all[(tok - 1) * H + u] : (p.h0 ? p.h0[(int64_t)b * H + u] : 0.f);
            const float dn = dh * (1.f - zg);
            dz_ = dh * (hp - ng) * zg * (1.f - zg);
            dn_ = dn * (1.f - ng * ng);
            dr_ = dn_ * hn * rg * (1.f - rg);
            dhn = dn_ * rg;
            dhz = dh * zg;
            if (u / US == s) {                       // owned slice: publish the projection gradients once
                float* o1 = p.dgi + tok * 3 * H;
                float* o2 = p.dgh + tok * 3 * H;
                o1[u] = dr_; o1[H + u] = dz_; o1[2 * H + u] = dn_;
                o2[u] = dr_; o2[H + u] = dz_; o2[2 * H + u] = dhn;
            }
        }
        s_g[r * 3 * H + u] = dr_;
        s_g[r * 3 * H + H + u] = dz_;
        s_g[r * 3 * H + 2 * H + u] = dhn;
        if (u / US == s) s_dz[r * US + (u % US)] = dhz;
    }
    __syncthreads();
    const int r = tid >> 4, j = tid & 15;
    float acc = 0.f;
    for (int g4 = 0; g4 < 3 * H / 4; ++g4) {
        const float4 gv = ld4(&s_g[r * 3 * H + 4 * g4]);
        const float4 wv = ld4(&s_w[((size_t)g4 * 16 + j) * 4]);
        acc = __builtin_fmaf(gv.x, wv.x, acc); acc = __builtin_fmaf(gv.y, wv.y, acc);
        acc = __builtin_fmaf(gv.z, wv.z, acc); acc = __builtin_fmaf(gv.w, wv.w, acc);
    }
    const int b = b0 + r;
    if (b < p.B) p.carry_out[(int64_t)b * H + s * US + j] = s_dz[r * US + j] + acc;
}

inline bool gru_ok(int B, int L, int H) { return B > 0 && L > 0 && H > 0 && H % 16 == 0 && H <= HMAX; }
inline size_t wlayout_floats(int H) { return (size_t)3 * H * H; }

}  // namespace

extern "C" size_t resel_gru_workspace_bytes(int B, int L, int H) {
    (void)L;
    return (wlayout_floats(H) + (size_t)2 * B * H) * sizeof(float);
}

extern "C" int resel_gru_seq_fwd(const float* gi, const float* w_hh, const float* b_hh, const float* h0,
                                 float* h_all, float* gates, void* workspace, int B, int L, int H, resel_stream_t stream) {
    if (!gi || !w_hh || !b_hh || !h_all || !workspace || !gru_ok(B, L, H)) return RESEL_EINVAL;
    if (!aligned16(h_all) || !aligned16(workspace) || (h0 && !aligned16(h0))) return RESEL_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    float* wf = (float*)workspace;
    const int64_t nw = (int64_t)3 * H * H;
    hipLaunchKernelGGL(gru_layout_fwd_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s, w_hh, wf, H);
    const size_t lds = ((size_t)H * 48 + (size_t)RG * H) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(gru_fwd_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    GruFwd p{gi, wf, b_hh, h0, h_all, gates, B, L, H, 0};
    dim3 grid(H / US, (B + RG - 1) / RG);
    for (int t = 0; t < L; ++t) {
        p.t = t;
        hipLaunchKernelGGL(gru_fwd_step_kernel, grid, dim3(256), lds, s, p);
    }
    return launch_status();
}

extern "C" int resel_gru_seq_bwd(const float* w_hh, const float* h0, const float* h_all, const float* gates,
                                 const float* dh_all, float* dgi, float* dgh, void* workspace,
                                 int B, int L, int H, resel_stream_t stream) {
    if (!w_hh || !h_all || !gates || !dh_all || !dgi || !dgh || !workspace || !gru_ok(B, L, H)) return RESEL_EINVAL;
    if (!aligned16(workspace)) return RESEL_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    float* wb = (float*)workspace;
    float* carry = wb + wlayout_floats(H);
    const int64_t nw = (int64_t)3 * H * H;
    hipLaunchKernelGGL(gru_layout_bwd_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s, w_hh, wb, H);
    if (hipMemsetAsync(carry, 0, (size_t)2 * B * H * sizeof(float), s) != hipSuccess) return RESEL_ELAUNCH;
    const size_t lds = ((size_t)3 * H * 16 + (size_t)RG * 3 * H + RG * US) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(gru_bwd_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    GruBwd p{wb, h0, h_all, gates, dh_all, nullptr, nullptr, dgi, dgh, B, L, H, 0};
    dim3 grid(H / US, (B + RG - 1) / RG);
    for (int t = L - 1; t >= 0; --t) {
        p.t = t;
        p.carry_in = carry + (size_t)((t + 1) & 1) * B * H;
        p.carry_out = carry + (size_t)(t & 1) * B * H;
        hipLaunchKernelGGL(gru_bwd_step_kernel, grid, dim3(256), lds, s, p);
    }
    return launch_status();
}
