// GRU recurrence on a hoisted input projection (gi = x W_ih^T + b_ih), forward and backward.
//
// The recurrence is T-sequential and each step needs every hidden unit of the previous step.  Two drivers share the same
// per-step arithmetic: (a) the PERSISTENT kernels further down - one launch per pass, the 16 unit-slice workgroups of a row
// group hand h_t (backward: the carry) to each other through tagged 8-byte granules, 3.0 / 3.9 us per step forward /
// backward and no per-step host work - used whenever the whole grid is co-resident (<= 256 workgroups); (b) one small
// launch per step enqueued back-to-back from C (a dependent kernel boundary, ~1.5 us, is cheaper than a full software
// grid barrier, 4-7 us, MI355X_MICROARCH "barrier-xcd"), 4.4 / 6.8 us per step, for larger grids.  Either way the
// LATENCY of one step is what matters, so the step is spread over the whole chip and keeps its operands close:
//   grid = (H/16 hidden-unit slices) x (B/4 row groups)  [16 x 16 = 256 workgroups at H = 256, B = 64: one per CU];
//   thread (j, kq) of a workgroup owns output unit j of the slice and the kq-th 1/16 of the reduction axis, and holds
//   that piece of W_hh for all three gates IN REGISTERS (3H/16 <= 96 floats, fetched with fully coalesced float4 loads
//   from a copy laid out once per call as [slice][float4 chunk][thread]); the 4 rows of h_{t-1} (forward) / of the gate
//   gradients (backward) sit in LDS and are read as broadcast float4s; the 16 partial sums of an output meet in LDS.
// All arithmetic is exact fp32 FMA (parity target 1e-4 against ATen's CPU GRU).  [First version: 64 workgroups with
// the W slice and 16 rows in LDS, every thread a full-length dot product - LDS-read bound at 9 us (fwd) / 23 us (bwd)
// per step.]  Neither roofline is tight for this layer; the reported figure is the achieved step rate.
#include "resel_common.h"
#include <cstdlib>

namespace {
using namespace resel;

// gate non-linearities on v_exp_f32 / v_rcp_f32 (1 ulp each: 2e-7 on a gate, parity bar 1e-4).  They sit on the serial chain of
// every step - 64 threads, everyone else waiting at a barrier - where libm's expf / tanhf cost ~0.25 us of a 3.4 us step.
__device__ __forceinline__ float gru_tanh(float x) {
    const float e = fast_exp(-2.0f * fabsf(x));                 // tanh|x| = (1 - e) / (1 + e)
    return copysignf((1.0f - e) * fast_rcp(1.0f + e), x);
}

constexpr int US = 16;    // hidden units per block
constexpr int RG = 4;     // batch rows per block
constexpr int KQMAX = 16; // reduction-axis pieces: H = KC * KQ with KQ <= 16; a block has 16 * KQ threads
constexpr int HMAX = 512;

// forward copy: thread (kq, j) of slice s needs W_hh[g*H + s*16 + j][kq*KC + i], g < 3, i < KC = H/16, as 3*KC/4 float4s
//   wf[((s * NC + c) * NT + tid) * 4 + e]   with chunk c = g * (KC/4) + i/4, e = i % 4, tid = kq * 16 + j, NT = 16 KQ
__global__ void gru_layout_fwd_kernel(const float* __restrict__ w_hh, float* __restrict__ wf, int H, int KQ) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)3 * H * H) return;
    const int KC = H / KQ, NC = 3 * KC / 4, NT = 16 * KQ;
    const int e = i & 3;
    int64_t r = i >> 2;
    const int tid = r % NT; r /= NT;
    const int c = r % NC;
    const int s = r / NC;
    const int g = c / (KC / 4), i4 = c % (KC / 4);
    const int kq = tid >> 4, j = tid & 15;
    wf[i] = w_hh[((int64_t)g * H + s * 16 + j) * H + kq * KC + i4 * 4 + e];
}
// backward copy: thread (kq, j) needs W_hh[q][s*16 + j] for the kq-th 1/16 of the 3H gate rows q: RC = 3H/16 values
//   wb[((s * NC + c) * NT + tid) * 4 + e]   with q = kq * RC + c * 4 + e
__global__ void gru_layout_bwd_kernel(const float* __restrict__ w_hh, float* __restrict__ wb, int H, int KQ) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)3 * H * H) return;
    const int RC = 3 * H / KQ, NC = RC / 4, NT = 16 * KQ;
    const int e = i & 3;
    int64_t r = i >> 2;
    const int tid = r % NT; r /= NT;
    const int c = r % NC;
    const int s = r / NC;
    const int kq = tid >> 4, j = tid & 15;
    wb[i] = w_hh[((int64_t)kq * RC + c * 4 + e) * H + s * 16 + j];
}

struct GruFwd {
    const float *gi, *wf, *b_hh, *h0;
    float *h_all, *gates;
    int B, L, H, t;
};

template <int KC>                                    // KC = H / 16: reduction elements per thread and gate
__global__ __launch_bounds__(256) void gru_fwd_step_kernel(GruFwd p) {
    constexpr int NC = 3 * KC / 4;
    __shared__ __attribute__((aligned(16))) float s_h[RG][KC * KQMAX];
    __shared__ float s_p[KQMAX][RG * 3][US + 1];
    const int tid = threadIdx.x, NT = blockDim.x, KQ = NT >> 4, H = p.H;
    const int s = blockIdx.x, b0 = blockIdx.y * RG;
    const int j = tid & 15, kq = tid >> 4;
    // this thread's piece of W_hh (registers) and the gate inputs of its output (issued first: longest latency)
    float4 w[NC];
    const float4* wsrc = reinterpret_cast<const float4*>(p.wf) + (size_t)s * NC * NT + tid;
#pragma unroll
    for (int c = 0; c < NC; ++c) w[c] = wsrc[(size_t)c * NT];
    const int ro = tid >> 4, uo = s * US + j;        // output mapping of the first RG*16 threads: (row ro, unit uo)
    const bool out_thr = tid < RG * US && b0 + ro < p.B;
    float gir = 0.f, giz = 0.f, gin = 0.f, br = 0.f, bz = 0.f, bn = 0.f;
    if (out_thr) {
        const float* g = p.gi + ((int64_t)(b0 + ro) * p.L + p.t) * 3 * H;
        gir = g[uo]; giz = g[H + uo]; gin = g[2 * H + uo];
        br = p.b_hh[uo]; bz = p.b_hh[H + uo]; bn = p.b_hh[2 * H + uo];
    }
    for (int i = tid; i < RG * H / 4; i += NT) {
        const int r = i / (H / 4), c4 = (i % (H / 4)) * 4;
        const int b = b0 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b < p.B) {
            if (p.t > 0) v = ld4(p.h_all + ((int64_t)b * p.L + (p.t - 1)) * H + c4);
            else if (p.h0) v = ld4(p.h0 + (int64_t)b * H + c4);
        }
        st4(&s_h[r][c4], v);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RG; ++r) {
        float a[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int i4 = 0; i4 < KC / 4; ++i4) {
            const float4 hv = ld4(&s_h[r][kq * KC + i4 * 4]);
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const float4 wv = w[g * (KC / 4) + i4];
                a[g] = __builtin_fmaf(hv.x, wv.x, a[g]); a[g] = __builtin_fmaf(hv.y, wv.y, a[g]);
                a[g] = __builtin_fmaf(hv.z, wv.z, a[g]); a[g] = __builtin_fmaf(hv.w, wv.w, a[g]);
            }
        }
        s_p[kq][r * 3 + 0][j] = a[0];
        s_p[kq][r * 3 + 1][j] = a[1];
        s_p[kq][r * 3 + 2][j] = a[2];
    }
    __syncthreads();
    if (out_thr) {
        float ar = 0.f, az = 0.f, an = 0.f;
        for (int q = 0; q < KQ; ++q) { ar += s_p[q][ro * 3][j]; az += s_p[q][ro * 3 + 1][j]; an += s_p[q][ro * 3 + 2][j]; }
        const int64_t tok = (int64_t)(b0 + ro) * p.L + p.t;
        const float rg = sigmoidf_(gir + ar + br);
        const float zg = sigmoidf_(giz + az + bz);
        const float hn = an + bn;
        const float ng = gru_tanh(gin + rg * hn);
        const float hp = s_h[ro][uo];
        p.h_all[tok * H + uo] = (1.f - zg) * ng + zg * hp;
        if (p.gates) {
            float* o = p.gates + tok * 4 * H;
            o[uo] = rg; o[H + uo] = zg; o[2 * H + uo] = ng; o[3 * H + uo] = hn;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Persistent forward: ONE launch walks all L steps.  The 16 unit-slice workgroups of a row group exchange h_t through 8-byte
// {tag = step + 1, value} granules in global memory (cdna_hip_programming.md section 6 Guideline 16, form R2: the data is
// the flag - relaxed agent-scope 8-byte atomic stores are write-through, the consumer re-reads its granules with relaxed
// agent-scope loads until every tag matches; no flag word, no fence, placement-independent).  Two granule buffers
// alternate by step parity: a workgroup can publish h_{t+1} only after it has read every slice of h_t, so a slow reader
// of h_t is never overtaken by more than one step.  W_hh pieces stay in registers for the whole sequence and the host
// issues one launch instead of L.  All workgroups must be resident (they are tiny: the launcher caps the grid); every spin
// is bounded and raises `err` instead of hanging.
typedef unsigned long long u64;
#define RESEL_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr unsigned SPIN_LIMIT = 1u << 22;

template <int KC>
__global__ __launch_bounds__(256) void gru_fwd_persistent_kernel(GruFwd p, u64* __restrict__ xchg, int* __restrict__ err) {
    constexpr int NC = 3 * KC / 4;
    __shared__ __attribute__((aligned(16))) float s_h[RG][KC * KQMAX];
    __shared__ float s_p[KQMAX][RG * 3][US + 1];
    __shared__ int s_fail;
    const int tid = threadIdx.x, NT = blockDim.x, KQ = NT >> 4, H = p.H;
    const int nslice = H / US;
    // ids congruent mod 8 share an XCD (round-robin dispatch): give a row group's slices one XCD when the grid allows it
    int id = blockIdx.x;
    const int total = gridDim.x;
    if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
    const int s = id % nslice, rg = id / nslice, b0 = rg * RG;
    const int j = tid & 15, kq = tid >> 4;
    float4 w[NC];
    const float4* wsrc = reinterpret_cast<const float4*>(p.wf) + (size_t)s * NC * NT + tid;
#pragma unroll
    for (int c = 0; c < NC; ++c) w[c] = wsrc[(size_t)c * NT];
    const int ro = tid >> 4, uo = s * US + j;
    const bool out_thr = tid < RG * US && b0 + ro < p.B;
    float br = 0.f, bz = 0.f, bn = 0.f;
    if (out_thr) { br = p.b_hh[uo]; bz = p.b_hh[H + uo]; bn = p.b_hh[2 * H + uo]; }
    u64* xg = xchg + (size_t)rg * 2 * RG * H;        // [2][RG][H] granules of this row group
    if (tid == 0) s_fail = 0;
    __syncthreads();
    for (int t = 0; t < p.L; ++t) {
        float gir = 0.f, giz = 0.f, gin = 0.f;
        if (out_thr) {                               // independent of the recurrence: in flight during the wait below
            const float* g = p.gi + ((int64_t)(b0 + ro) * p.L + t) * 3 * H;
            gir = g[uo]; giz = g[H + uo]; gin = g[2 * H + uo];
        }
        if (t == 0) {
            for (int i = tid; i < RG * H; i += NT) {
                const int r = i / H, u = i % H;
                s_h[r][u] = (p.h0 && b0 + r < p.B) ? p.h0[(int64_t)(b0 + r) * H + u] : 0.f;
            }
        } else {
            const u64* src = xg + (size_t)((t - 1) & 1) * RG * H;
            const unsigned epoch = (unsigned)t;      // h_{t-1} was published with tag t
            // all granules of this thread are requested before the first tag is looked at: one L2 round trip in the common
            // case instead of one per granule (the polls were the longest part of a step)
            constexpr int NPOLL = (RG * KC + 15) / 16;           // = RG * H / NT  (H = KC * KQ, NT = 16 * KQ)
            u64 xv[NPOLL];
            // The first poll is held back by ~770 cycles where a step is long enough (H >= 256): h_{t-1} of the other slices becomes
            // visible 0.4-0.5 us after this workgroup's own publish, and a poll that arrives in L2 before the data costs a whole second
            // round trip.  B = 64, H = 256, same box, us per step: no delay 2.70; s_sleep 2 / 4 / 8 / 12 / 16 / 20 / 24 / 32: 2.61 / 2.62 /
            // 2.65 / 2.20 / 2.24 / 2.35 / 2.46 / 2.67.  B = 32: 2.58 -> 2.15.  H = 128: no change; H = 64: 2.08 -> 2.25 (no delay there).
            // (The backward's poll already sits behind the loads of its carry-independent operands: any delay there measured slower.)
            if (KC >= 16) __builtin_amdgcn_s_sleep(12);
#pragma unroll
            for (int q = 0; q < NPOLL; ++q) {
                const int i = tid + q * NT;
                xv[q] = i < RG * H ? __hip_atomic_load(src + i, RESEL_RLX_AGENT) : 0;
            }
#pragma unroll
            for (int q = 0; q < NPOLL; ++q) {
                const int i = tid + q * NT;
                if (i >= RG * H) continue;
                u64 x = xv[q];
                unsigned spins = 0;
                while ((unsigned)(x >> 32) != epoch) {
                    if (++spins > SPIN_LIMIT) { s_fail = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                    x = __hip_atomic_load(src + i, RESEL_RLX_AGENT);
                }
                (&s_h[0][0])[(i / H) * (KC * KQMAX) + (i % H)] = __uint_as_float((unsigned)x);
            }
        }
        __syncthreads();
        if (s_fail) {                                // uniform: written before the barrier.  Fail LOUDLY: poison the output
            if (tid == 0) atomicExch(err, 1);
            if (out_thr) p.h_all[((int64_t)(b0 + ro) * p.L + t) * H + uo] = __builtin_nanf("");
            return;
        }
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            float a[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int i4 = 0; i4 < KC / 4; ++i4) {
                const float4 hv = ld4(&s_h[r][kq * KC + i4 * 4]);
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const float4 wv = w[g * (KC / 4) + i4];
                    a[g] = __builtin_fmaf(hv.x, wv.x, a[g]); a[g] = __builtin_fmaf(hv.y, wv.y, a[g]);
                    a[g] = __builtin_fmaf(hv.z, wv.z, a[g]); a[g] = __builtin_fmaf(hv.w, wv.w, a[g]);
                }
            }
            s_p[kq][r * 3 + 0][j] = a[0];
            s_p[kq][r * 3 + 1][j] = a[1];
            s_p[kq][r * 3 + 2][j] = a[2];
        }
        // h_{t-1} of the owned unit is fetched on THIS side of the barrier: behind it nothing reads s_h any more, so the next step's
        // poll may overwrite it while slower waves still finish this step - one barrier per step less
        const float hp = tid < RG * US ? s_h[ro][uo] : 0.f;
        __syncthreads();
        if (tid < RG * US) {
            float hnew = 0.f;
            if (out_thr) {
                float ar = 0.f, az = 0.f, an = 0.f;
                for (int q = 0; q < KQ; ++q) { ar += s_p[q][ro * 3][j]; az += s_p[q][ro * 3 + 1][j]; an += s_p[q][ro * 3 + 2][j]; }
                const int64_t tok = (int64_t)(b0 + ro) * p.L + t;
                const float rgt = sigmoidf_(gir + ar + br);
                const float zg = sigmoidf_(giz + az + bz);
                const float hn = an + bn;
                const float ng = gru_tanh(gin + rgt * hn);
                hnew = (1.f - zg) * ng + zg * hp;
                // publish FIRST (the other slices wait for it), the step's outputs behind it
                __hip_atomic_store(xg + (size_t)(t & 1) * RG * H + (size_t)ro * H + uo, ((u64)(unsigned)(t + 1) << 32) | __float_as_uint(hnew),
                                   RESEL_RLX_AGENT);
                p.h_all[tok * H + uo] = hnew;
                if (p.gates) {
                    float* o = p.gates + tok * 4 * H;
                    o[uo] = rgt; o[H + uo] = zg; o[2 * H + uo] = ng; o[3 * H + uo] = hn;
                }
            } else {
                // rows past B publish zeros so that every granule of the row group gets its tag
                __hip_atomic_store(xg + (size_t)(t & 1) * RG * H + (size_t)ro * H + uo, ((u64)(unsigned)(t + 1) << 32), RESEL_RLX_AGENT);
            }
        }
        // A third barrier is not needed for correctness any more (nothing behind the second one reads s_h; s_p is rewritten behind the
        // next step's first barrier) but the forward is FASTER with it: 2.71 against 3.00 us per step on the same box - the waves stay in
        // step for the next poll.  (The backward is faster without: 3.35 against 3.44.)
#ifndef GRU_AB_NOB3
        __syncthreads();
#endif
    }
}

struct GruBwd {
    const float *wb, *h0, *h_all, *gates, *dh_all, *carry_in;
    float *carry_out, *dgi, *dgh;
    int B, L, H, t;
};

template <int KC>
__global__ __launch_bounds__(256) void gru_bwd_step_kernel(GruBwd p) {
    constexpr int RC = 3 * KC;                       // gate rows per thread (3H / KQ)
    constexpr int NC = RC / 4;
    __shared__ __attribute__((aligned(16))) float s_gbuf[RG * 3 * KC * KQMAX];   // [RG][3H]: (dr | dz | dhn) of the RG rows, all H units
    __shared__ float s_dz[RG][US];                                    // dh * z of the owned units
    __shared__ float s_p[KQMAX][RG][US + 1];
    const int tid = threadIdx.x, NT = blockDim.x, KQ = NT >> 4, H = p.H;
    auto s_g = [&](int r) { return s_gbuf + (size_t)r * 3 * H; };
    const int s = blockIdx.x, b0 = blockIdx.y * RG;
    const int j = tid & 15, kq = tid >> 4;
    float4 w[NC];
    const float4* wsrc = reinterpret_cast<const float4*>(p.wb) + (size_t)s * NC * NT + tid;
#pragma unroll
    for (int c = 0; c < NC; ++c) w[c] = wsrc[(size_t)c * NT];
    // gate gradients for the RG rows x all H units (recomputed by every unit slice of the row group: elementwise)
    for (int i = tid; i < RG * H; i += NT) {
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
        s_g(r)[u] = dr_;
        s_g(r)[H + u] = dz_;
        s_g(r)[2 * H + u] = dhn;
        if (u / US == s) s_dz[r][u % US] = dhz;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RG; ++r) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float4 gv = ld4(s_g(r) + kq * RC + c * 4);
            acc = __builtin_fmaf(gv.x, w[c].x, acc); acc = __builtin_fmaf(gv.y, w[c].y, acc);
            acc = __builtin_fmaf(gv.z, w[c].z, acc); acc = __builtin_fmaf(gv.w, w[c].w, acc);
        }
        s_p[kq][r][j] = acc;
    }
    __syncthreads();
    if (tid < RG * US) {
        const int r = tid >> 4;
        const int b = b0 + r;
        float acc = s_dz[r][j];
        for (int q = 0; q < KQ; ++q) acc += s_p[q][r][j];
        if (b < p.B) p.carry_out[(int64_t)b * H + s * US + j] = acc;
    }
}

// Persistent backward: same exchange protocol as the forward; step t publishes the carry d h_{t-1} (its 16 units x RG rows)
// with tag (L - t), the step for t-1 collects all H units of its rows.  The gate-gradient operands of the NEXT step do not
// depend on the carry and are loaded before the wait.
template <int KC>
__global__ __launch_bounds__(256) void gru_bwd_persistent_kernel(GruBwd p, u64* __restrict__ xchg, int* __restrict__ err) {
    constexpr int RC = 3 * KC;
    constexpr int NC = RC / 4;
    constexpr int PER = (RG * KC * KQMAX + 255) / 256;       // (row, unit) items per thread at the widest block
    __shared__ __attribute__((aligned(16))) float s_gbuf[RG * 3 * KC * KQMAX];
    __shared__ float s_dz[RG][US], s_dn[RG][US];
    __shared__ float s_p[KQMAX][RG][US + 1];
    __shared__ int s_fail;
    const int tid = threadIdx.x, NT = blockDim.x, KQ = NT >> 4, H = p.H;
    auto s_g = [&](int r) { return s_gbuf + (size_t)r * 3 * H; };
    const int nslice = H / US;
    int id = blockIdx.x;
    const int total = gridDim.x;
    if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
    const int s = id % nslice, rg = id / nslice, b0 = rg * RG;
    const int j = tid & 15, kq = tid >> 4;
    float4 w[NC];
    const float4* wsrc = reinterpret_cast<const float4*>(p.wb) + (size_t)s * NC * NT + tid;
#pragma unroll
    for (int c = 0; c < NC; ++c) w[c] = wsrc[(size_t)c * NT];
    u64* xg = xchg + (size_t)rg * 2 * RG * H;
    if (tid == 0) s_fail = 0;
    __syncthreads();
    for (int t = p.L - 1, k = 0; t >= 0; --t, ++k) {          // k = steps already done; the carry consumed here has tag k
        // operands that do not depend on the carry
        float dho[PER], rgv[PER], zgv[PER], ngv[PER], hnv[PER], hpv[PER];
#pragma unroll
        for (int n = 0; n < PER; ++n) {
            const int i = tid + n * NT;
            dho[n] = rgv[n] = zgv[n] = ngv[n] = hnv[n] = hpv[n] = 0.f;
            if (i < RG * H && b0 + i / H < p.B) {
                const int r = i / H, u = i % H;
                const int64_t tok = (int64_t)(b0 + r) * p.L + t;
                const float* g = p.gates + tok * 4 * H;
                dho[n] = p.dh_all[tok * H + u];
                rgv[n] = g[u]; zgv[n] = g[H + u]; ngv[n] = g[2 * H + u]; hnv[n] = g[3 * H + u];
                hpv[n] = t > 0 ? p.h_all[(tok - 1) * H + u] : (p.h0 ? p.h0[(int64_t)(b0 + r) * H + u] : 0.f);
            }
        }
        const u64* src = xg + (size_t)((k - 1) & 1) * RG * H;
        u64 xv[PER];                                             // every carry granule requested before the first tag check
#pragma unroll
        for (int n = 0; n < PER; ++n) {
            const int i = tid + n * NT;
            xv[n] = (k > 0 && i < RG * H) ? __hip_atomic_load(src + i, RESEL_RLX_AGENT) : 0;
        }
#pragma unroll
        for (int n = 0; n < PER; ++n) {
            const int i = tid + n * NT;
            if (i >= RG * H) continue;
            const int r = i / H, u = i % H;
            float carry = 0.f;
            if (k > 0) {
                u64 x = xv[n];
                unsigned spins = 0;
                while ((unsigned)(x >> 32) != (unsigned)k) {
                    if (++spins > SPIN_LIMIT) { s_fail = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                    x = __hip_atomic_load(src + i, RESEL_RLX_AGENT);
                }
                carry = __uint_as_float((unsigned)x);
            }
            float dr_ = 0.f, dz_ = 0.f, dn_ = 0.f, dhn = 0.f, dhz = 0.f;
            if (b0 + r < p.B) {
                const float dh = dho[n] + carry;
                const float rgt = rgv[n], zg = zgv[n], ng = ngv[n], hn = hnv[n];
                const float dn = dh * (1.f - zg);
                dz_ = dh * (hpv[n] - ng) * zg * (1.f - zg);
                dn_ = dn * (1.f - ng * ng);
                dr_ = dn_ * hn * rgt * (1.f - rgt);
                dhn = dn_ * rgt;
                dhz = dh * zg;
            }
            s_g(r)[u] = dr_;
            s_g(r)[H + u] = dz_;
            s_g(r)[2 * H + u] = dhn;
            if (u / US == s) { s_dz[r][u % US] = dhz; s_dn[r][u % US] = dn_; }
        }
        __syncthreads();
        if (s_fail) {                                // fail loudly: poison this step's projection gradients
            if (tid == 0) atomicExch(err, 2);
            if (tid < RG * US && b0 + (tid >> 4) < p.B) p.dgi[((int64_t)(b0 + (tid >> 4)) * p.L + t) * 3 * H + s * US + j] = __builtin_nanf("");
            return;
        }
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float4 gv = ld4(s_g(r) + kq * RC + c * 4);
                acc = __builtin_fmaf(gv.x, w[c].x, acc); acc = __builtin_fmaf(gv.y, w[c].y, acc);
                acc = __builtin_fmaf(gv.z, w[c].z, acc); acc = __builtin_fmaf(gv.w, w[c].w, acc);
            }
            s_p[kq][r][j] = acc;
        }
        // read on this side of the barrier (as h_{t-1} in the forward): dh z of the owned unit, and the owned projection gradients, which
        // are stored BEHIND the publish (24 stores of one quarter-wave used to sit in front of the first barrier of every step)
        float dz_own = 0.f, o_dr = 0.f, o_dz = 0.f, o_dhn = 0.f, o_dn = 0.f;
        if (tid < RG * US) {
            const int r = tid >> 4, u = s * US + j;
            dz_own = s_dz[r][j]; o_dn = s_dn[r][j];
            o_dr = s_g(r)[u]; o_dz = s_g(r)[H + u]; o_dhn = s_g(r)[2 * H + u];
        }
        __syncthreads();
        if (tid < RG * US) {
            const int r = tid >> 4;
            float acc = dz_own;
            for (int q = 0; q < KQ; ++q) acc += s_p[q][r][j];
            if (b0 + r >= p.B) acc = 0.f;
            if (t == 0) { if (b0 + r < p.B) p.carry_out[(int64_t)(b0 + r) * H + s * US + j] = acc; }     // d h_0 (unused by the caller)
            else __hip_atomic_store(xg + (size_t)(k & 1) * RG * H + (size_t)r * H + s * US + j,
                                    ((u64)(unsigned)(k + 1) << 32) | __float_as_uint(acc), RESEL_RLX_AGENT);
            if (b0 + r < p.B) {
                const int u = s * US + j;
                const int64_t tok = (int64_t)(b0 + r) * p.L + t;
                float* o1 = p.dgi + tok * 3 * H;
                float* o2 = p.dgh + tok * 3 * H;
                o1[u] = o_dr; o1[H + u] = o_dz; o1[2 * H + u] = o_dn;
                o2[u] = o_dr; o2[H + u] = o_dz; o2[2 * H + u] = o_dhn;
            }
        }
        // no third barrier: s_g / s_dz are rewritten before, s_p behind the next step's first barrier - neither is read above (3.44 -> 3.35 us per step)
    }
}

// H = KC * KQ with KC in {4, 8, 12, 16, 24, 32} (templated) and 4 <= KQ <= 16: every multiple of 16 up to 256, and
// 320 / 384 / 448 / 512 above
inline int pick_kc(int H) {
    const int cand[6] = {4, 8, 12, 16, 24, 32};
    for (int kc : cand)
        if (H % kc == 0 && H / kc <= KQMAX && H / kc >= 4) return kc;
    return 0;
}
inline bool gru_ok(int B, int L, int H) { return B > 0 && L > 0 && H > 0 && H % 16 == 0 && H <= HMAX && pick_kc(H) > 0; }
inline size_t wlayout_floats(int H) { return (size_t)3 * H * H; }

}  // namespace

inline size_t xchg_granules(int B, int H) { return (size_t)((B + RG - 1) / RG) * 2 * RG * H; }
// [W copy | carry (2 B H floats) | exchange granules (8 bytes each) | error word]
inline size_t xchg_offset_bytes(int B, int H) { return (wlayout_floats(H) + (size_t)2 * B * H) * sizeof(float); }

extern "C" size_t resel_gru_workspace_bytes(int B, int L, int H) {
    (void)L;
    return xchg_offset_bytes(B, H) + xchg_granules(B, H) * sizeof(unsigned long long) + 64;
}

// persistent form: all workgroups must be co-resident.  The backward kernel runs at 2 waves per SIMD (2 workgroups per CU),
// so the grid is capped at one workgroup per CU of the smallest configuration this library targets (256 CUs); larger
// problems keep the launch-per-step form.  RESEL_GRU_PERSISTENT=0 forces the launch-per-step form.
inline bool persistent_ok(int B, int H) {
    static const int mode = getenv("RESEL_GRU_PERSISTENT") ? atoi(getenv("RESEL_GRU_PERSISTENT")) : 1;
    return mode != 0 && (H / US) * ((B + RG - 1) / RG) <= 256;
}

extern "C" int resel_gru_seq_fwd(const float* gi, const float* w_hh, const float* b_hh, const float* h0,
                                 float* h_all, float* gates, void* workspace, int B, int L, int H, resel_stream_t stream) {
    if (!gi || !w_hh || !b_hh || !h_all || !workspace || !gru_ok(B, L, H)) return RESEL_EINVAL;
    if (!aligned16(h_all) || !aligned16(workspace) || (h0 && !aligned16(h0))) return RESEL_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    float* wf = (float*)workspace;
    const int64_t nw = (int64_t)3 * H * H;
    const int KC = pick_kc(H), KQ = H / KC;
    hipLaunchKernelGGL(gru_layout_fwd_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s, w_hh, wf, H, KQ);
    GruFwd p{gi, wf, b_hh, h0, h_all, gates, B, L, H, 0};
    if (persistent_ok(B, H)) {
        char* base = (char*)workspace + xchg_offset_bytes(B, H);
        u64* xchg = (u64*)base;
        int* err = (int*)(base + xchg_granules(B, H) * sizeof(u64));
        if (hipMemsetAsync(base, 0, xchg_granules(B, H) * sizeof(u64) + 64, s) != hipSuccess) return RESEL_ELAUNCH;
        const dim3 pgrid((H / US) * ((B + RG - 1) / RG));
        switch (KC) {
            case 4: launch_timed(RESEL_PROF_GRU_FWD, gru_fwd_persistent_kernel<4>, pgrid, dim3(16 * KQ), 0, s, p, xchg, err); break;
            case 8: launch_timed(RESEL_PROF_GRU_FWD, gru_fwd_persistent_kernel<8>, pgrid, dim3(16 * KQ), 0, s, p, xchg, err); break;
            case 12: launch_timed(RESEL_PROF_GRU_FWD, gru_fwd_persistent_kernel<12>, pgrid, dim3(16 * KQ), 0, s, p, xchg, err); break;
            case 16: launch_timed(RESEL_PROF_GRU_FWD, gru_fwd_persistent_kernel<16>, pgrid, dim3(16 * KQ), 0, s, p, xchg, err); break;
            case 24: launch_timed(RESEL_PROF_GRU_FWD, gru_fwd_persistent_kernel<24>, pgrid, dim3(16 * KQ), 0, s, p, xchg, err); break;
            case 32: launch_timed(RESEL_PROF_GRU_FWD, gru_fwd_persistent_kernel<32>, pgrid, dim3(16 * KQ), 0, s, p, xchg, err); break;
            default: return RESEL_EINVAL;
        }
        return launch_status();
    }
    dim3 grid(H / US, (B + RG - 1) / RG);
    for (int t = 0; t < L; ++t) {
        p.t = t;
        switch (KC) {
            case 4: launch_timed(RESEL_PROF_GRU_FWD, gru_fwd_step_kernel<4>, grid, dim3(16 * KQ), 0, s, p); break;
            case 8: launch_timed(RESEL_PROF_GRU_FWD, gru_fwd_step_kernel<8>, grid, dim3(16 * KQ), 0, s, p); break;
            case 12: launch_timed(RESEL_PROF_GRU_FWD, gru_fwd_step_kernel<12>, grid, dim3(16 * KQ), 0, s, p); break;
            case 16: launch_timed(RESEL_PROF_GRU_FWD, gru_fwd_step_kernel<16>, grid, dim3(16 * KQ), 0, s, p); break;
            case 24: launch_timed(RESEL_PROF_GRU_FWD, gru_fwd_step_kernel<24>, grid, dim3(16 * KQ), 0, s, p); break;
            case 32: launch_timed(RESEL_PROF_GRU_FWD, gru_fwd_step_kernel<32>, grid, dim3(16 * KQ), 0, s, p); break;
            default: return RESEL_EINVAL;
        }
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
    const int KC = pick_kc(H), KQ = H / KC;
    hipLaunchKernelGGL(gru_layout_bwd_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s, w_hh, wb, H, KQ);
    if (hipMemsetAsync(carry, 0, (size_t)2 * B * H * sizeof(float), s) != hipSuccess) return RESEL_ELAUNCH;
    GruBwd p{wb, h0, h_all, gates, dh_all, nullptr, nullptr, dgi, dgh, B, L, H, 0};
    if (persistent_ok(B, H)) {
        char* base = (char*)workspace + xchg_offset_bytes(B, H);
        u64* xchg = (u64*)base;
        int* err = (int*)(base + xchg_granules(B, H) * sizeof(u64));
        if (hipMemsetAsync(base, 0, xchg_granules(B, H) * sizeof(u64) + 64, s) != hipSuccess) return RESEL_ELAUNCH;
        p.carry_out = carry;
        const dim3 pgrid((H / US) * ((B + RG - 1) / RG));
        switch (KC) {
            case 4: launch_timed(RESEL_PROF_GRU_BWD, gru_bwd_persistent_kernel<4>, pgrid, dim3(16 * KQ), 0, s, p, xchg, err); break;
            case 8: launch_timed(RESEL_PROF_GRU_BWD, gru_bwd_persistent_kernel<8>, pgrid, dim3(16 * KQ), 0, s, p, xchg, err); break;
            case 12: launch_timed(RESEL_PROF_GRU_BWD, gru_bwd_persistent_kernel<12>, pgrid, dim3(16 * KQ), 0, s, p, xchg, err); break;
            case 16: launch_timed(RESEL_PROF_GRU_BWD, gru_bwd_persistent_kernel<16>, pgrid, dim3(16 * KQ), 0, s, p, xchg, err); break;
            case 24: launch_timed(RESEL_PROF_GRU_BWD, gru_bwd_persistent_kernel<24>, pgrid, dim3(16 * KQ), 0, s, p, xchg, err); break;
            case 32: launch_timed(RESEL_PROF_GRU_BWD, gru_bwd_persistent_kernel<32>, pgrid, dim3(16 * KQ), 0, s, p, xchg, err); break;
            default: return RESEL_EINVAL;
        }
        return launch_status();
    }
    dim3 grid(H / US, (B + RG - 1) / RG);
    for (int t = L - 1; t >= 0; --t) {
        p.t = t;
        p.carry_in = carry + (size_t)((t + 1) & 1) * B * H;
        p.carry_out = carry + (size_t)(t & 1) * B * H;
        switch (KC) {
            case 4: launch_timed(RESEL_PROF_GRU_BWD, gru_bwd_step_kernel<4>, grid, dim3(16 * KQ), 0, s, p); break;
            case 8: launch_timed(RESEL_PROF_GRU_BWD, gru_bwd_step_kernel<8>, grid, dim3(16 * KQ), 0, s, p); break;
            case 12: launch_timed(RESEL_PROF_GRU_BWD, gru_bwd_step_kernel<12>, grid, dim3(16 * KQ), 0, s, p); break;
            case 16: launch_timed(RESEL_PROF_GRU_BWD, gru_bwd_step_kernel<16>, grid, dim3(16 * KQ), 0, s, p); break;
            case 24: launch_timed(RESEL_PROF_GRU_BWD, gru_bwd_step_kernel<24>, grid, dim3(16 * KQ), 0, s, p); break;
            case 32: launch_timed(RESEL_PROF_GRU_BWD, gru_bwd_step_kernel<32>, grid, dim3(16 * KQ), 0, s, p); break;
            default: return RESEL_EINVAL;
        }
    }
    return launch_status();
}
