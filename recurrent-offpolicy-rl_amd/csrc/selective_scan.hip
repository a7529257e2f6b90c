// Selective scan (Mamba S6 with `start` resets) for gfx950 - forward and backward.
//
// Layout / mapping (DESIGN.md section "selective_scan"):
//   * activations token-major [B*L, Di]; one workgroup owns (row b, 64-channel tile) for the whole sequence;
//   * lane  <-> channel (64 lanes = 64 consecutive channels: every global access of a wave is one or more
//     fully used 256-byte segments);
//   * wave  <-> group of NS states: the N = NS*NW state columns of a channel are split over the NW waves, so
//     B_t / C_t of a step are WAVE-UNIFORM and are fetched with scalar loads (s_load_dwordx{2,4,8}) and used
//     as SGPR operands of v_fma - no LDS, no per-lane traffic for them;
//   * time is processed in chunks staged through LDS: a coalesced float4 tile load applies softplus / the
//     delta*u product once per element, the scan phase reads one dword per lane per step, the per-wave
//     partial sums meet again in LDS and the output tile leaves with float4 stores;
//   * block id -> (b, channel tile) is XCD-aware: blocks are dealt round-robin over the 8 XCDs, so the
//     Di/64 channel tiles of one row b are given ids that differ by multiples of 8 and share one L2 for the
//     B_t / C_t rows they all read (speed only; any placement is correct).
// Backward: reverse-time recurrence with two-level recomputation - the forward leaves a state checkpoint
// every 64 steps, the backward rebuilds 16-step sub-checkpoints (registers) and then a 16-step state history
// (registers) per sub-chunk.  Reductions over channels (dB, dC) use an in-wave multi-value butterfly plus
// per-tile partial slabs summed by a second kernel: no float atomics, bitwise reproducible.
#include "resel_common.h"
#include <hip/hip_ext.h>
#include <vector>

namespace {
using namespace resel;

// Diagnostics (off by default): when enabled, the scan kernels are dispatched with hipExtLaunchKernelGGL and a
// (start, stop) HIP event pair bound to that dispatch, i.e. timed on the very stream they run on.
struct ProfSlot { std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; };
bool g_prof_on = false;
ProfSlot g_prof[2];                   // 0 = sscan_fwd_kernel, 1 = sscan_bwd_kernel

template <typename K, typename P>
void launch_maybe_timed(int slot, K kernel, dim3 grid, dim3 block, hipStream_t s, const P& p) {
    if (!g_prof_on) {
        hipLaunchKernelGGL(kernel, grid, block, 0, s, p);
        return;
    }
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    hipExtLaunchKernelGGL(kernel, grid, block, 0, s, a, b, 0, p);
    g_prof[slot].ev.emplace_back(a, b);
}

constexpr int TILE_C = 64;            // channels per workgroup
constexpr int TC = 32;                // time steps per LDS chunk (forward)
constexpr int CKS = RESEL_SSCAN_CKPT; // checkpoint stride
constexpr int SC = 16;                // backward sub-chunk (state history kept in registers)
constexpr int NSUB = CKS / SC;

struct FwdParams {
    const float *u, *delta, *z, *A, *Bm, *Cm, *D, *delta_bias, *start;
    float *out, *ckpt, *last_state;
    int64_t ld_u, ld_delta, ld_z, ld_b, ld_c, ld_out;
    int B, L, Di, N, nck, softplus, nd;
};

template <int NS>
__device__ __forceinline__ void load_coef(cfloat_p p, float (&dst)[NS]) {
#pragma unroll
    for (int j = 0; j < NS; ++j) dst[j] = p[j];
}

// XCD-aware decode of a 1-D block id into (row b, channel tile dt): ids congruent mod 8 share an XCD.
__device__ __forceinline__ bool decode_block(int id, int nd, int B, int& b, int& dt) {
    const int xcd = id & 7, k = id >> 3;
    dt = k % nd;
    b = (k / nd) * 8 + xcd;
    return b < B;
}

template <int NS, int NW>
__global__ __launch_bounds__(NW * 64) void sscan_fwd_kernel(FwdParams p) {
    constexpr int NT = NW * 64;
    constexpr int PER_T = (TC * 16) / NT;          // float4 tile items per thread
    __shared__ __attribute__((aligned(16))) float s_dl[TC][TILE_C];
    __shared__ __attribute__((aligned(16))) float s_du[TC][TILE_C];
    __shared__ __attribute__((aligned(16))) float s_y[NW][TC][TILE_C];

    int b, dt;
    if (!decode_block(blockIdx.x, p.nd, p.B, b, dt)) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int d0 = dt * TILE_C;
    const int d = d0 + lane;
    const bool d_ok = d < p.Di;
    const int N = NS * NW;
    const int64_t tok0 = (int64_t)b * p.L;

    // per-lane recurrence constants: A (pre-scaled for exp2; clamped below zero so that the reset trick
    // delta := +inf always yields exp2(-inf) = 0) for this lane's channel and this wave's states
    float A2[NS], h[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        A2[j] = d_ok ? fminf(p.A[(int64_t)d * N + w * NS + j] * RESEL_LOG2E, -1e-30f) : -1.f;
        h[j] = 0.f;
    }
    const int tc4 = (tid & 15) * 4;
    const int tr0 = tid >> 4;                       // first tile row of this thread
    const bool c_ok = (d0 + tc4) < p.Di;
    float4 Dv = make_float4(0.f, 0.f, 0.f, 0.f), bv = Dv;
    if (c_ok) {
        if (p.D) Dv = ld4(p.D + d0 + tc4);
        if (p.delta_bias) bv = ld4(p.delta_bias + d0 + tc4);
    }
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 pu[PER_T], pd[PER_T], pz[PER_T];         // register prefetch of the NEXT chunk's tile
    float4 u_r[PER_T], z_r[PER_T];

    auto prefetch = [&](int c0) {
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int t = c0 + tr0 + i * (NT / 16);
            pu[i] = zero4; pd[i] = zero4; pz[i] = zero4;
            if (t < p.L && c_ok) {
                const int64_t tok = tok0 + t;
                pu[i] = ld4(p.u + tok * p.ld_u + d0 + tc4);
                pd[i] = ld4(p.delta + tok * p.ld_delta + d0 + tc4);
                if (p.z) pz[i] = ld4(p.z + tok * p.ld_z + d0 + tc4);
            }
        }
    };
    prefetch(0);

    for (int c0 = 0; c0 < p.L; c0 += TC) {
        // ---- stage: softplus(delta + bias), delta * u -> LDS; u and z stay in registers for the output phase
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int r = tr0 + i * (NT / 16);
            float4 dv = pd[i];
            const float4 uv = pu[i];
            dv.x += bv.x; dv.y += bv.y; dv.z += bv.z; dv.w += bv.w;
            if (p.softplus) {
                dv.x = softplusf_(dv.x); dv.y = softplusf_(dv.y); dv.z = softplusf_(dv.z); dv.w = softplusf_(dv.w);
            }
            u_r[i] = uv;
            z_r[i] = pz[i];
            st4(&s_dl[r][tc4], dv);
            st4(&s_du[r][tc4], make_float4(dv.x * uv.x, dv.y * uv.y, dv.z * uv.z, dv.w * uv.w));
        }
        __syncthreads();
        if (c0 + TC < p.L) prefetch(c0 + TC);        // in flight during the whole scan phase

        // ---- scan: lane = channel, wave = state group, sequential in time; B_t / C_t / start_t are scalar loads
        const int nst = min(TC, p.L - c0);
        float Bc[NS], Cc[NS];
        {
            const int64_t tok = tok0 + c0;
            load_coef<NS>(as_uniform(p.Bm + tok * p.ld_b + w * NS), Bc);
            load_coef<NS>(as_uniform(p.Cm + tok * p.ld_c + w * NS), Cc);
        }
        float sflag = p.start ? as_uniform(p.start + tok0 + c0)[0] : 0.f;
        float dl = s_dl[0][lane], du = s_du[0][lane];
        for (int t = 0; t < nst; ++t) {
            // prefetch the next step's coefficients (clamped: the last prefetch of a chunk is a harmless re-read)
            const int tn = min(t + 1, nst - 1);
            const int64_t tokn = tok0 + c0 + tn;
            float Bn[NS], Cn[NS];
            load_coef<NS>(as_uniform(p.Bm + tokn * p.ld_b + w * NS), Bn);
            load_coef<NS>(as_uniform(p.Cm + tokn * p.ld_c + w * NS), Cn);
            const float sn = p.start ? as_uniform(p.start + tokn)[0] : 0.f;
            const float dln = s_dl[tn][lane], dun = s_du[tn][lane];

            const float dle = (sflag != 0.f) ? __builtin_inff() : dl;   // reset: exp2(-inf) = 0 wipes h_{t-1}
            float y = 0.f;
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const float dA = fast_exp2(dle * A2[j]);
                h[j] = __builtin_fmaf(dA, h[j], du * Bc[j]);
                y = __builtin_fmaf(Cc[j], h[j], y);
            }
            s_y[w][t][lane] = y;
            const int tabs = c0 + t + 1;
            if (p.ckpt != nullptr && (tabs % CKS) == 0 && tabs < p.L && d_ok) {
                float* ck = p.ckpt + (((int64_t)b * p.nck + (tabs / CKS - 1)) * N + w * NS) * p.Di + d;
#pragma unroll
                for (int j = 0; j < NS; ++j) ck[(int64_t)j * p.Di] = h[j];
            }
#pragma unroll
            for (int j = 0; j < NS; ++j) { Bc[j] = Bn[j]; Cc[j] = Cn[j]; }
            sflag = sn; dl = dln; du = dun;
        }
        __syncthreads();

        // ---- output tile: sum the NW partials, skip term, gate, float4 store
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int r = tr0 + i * (NT / 16);
            const int t = c0 + r;
            if (t < p.L && c_ok) {
                float4 y = ld4(&s_y[0][r][tc4]);
#pragma unroll
                for (int ww = 1; ww < NW; ++ww) {
                    const float4 q = ld4(&s_y[ww][r][tc4]);
                    y.x += q.x; y.y += q.y; y.z += q.z; y.w += q.w;
                }
                const float4 uv = u_r[i];
                y.x = __builtin_fmaf(Dv.x, uv.x, y.x); y.y = __builtin_fmaf(Dv.y, uv.y, y.y);
                y.z = __builtin_fmaf(Dv.z, uv.z, y.z); y.w = __builtin_fmaf(Dv.w, uv.w, y.w);
                if (p.z) {
                    const float4 zv = z_r[i];
                    y.x *= siluf_(zv.x); y.y *= siluf_(zv.y); y.z *= siluf_(zv.z); y.w *= siluf_(zv.w);
                }
                st4(p.out + (tok0 + t) * p.ld_out + d0 + tc4, y);
            }
        }
        // no barrier needed here: the next stage only writes s_dl/s_du (last read before the barrier above),
        // and s_y is rewritten only after the barrier that follows that stage.
    }
    if (p.last_state != nullptr && d_ok) {
#pragma unroll
        for (int j = 0; j < NS; ++j) p.last_state[((int64_t)b * p.Di + d) * N + w * NS + j] = h[j];
    }
}

// =====================================================================================================
// backward
// =====================================================================================================
struct BwdParams {
    const float *u, *delta, *z, *A, *Bm, *Cm, *D, *delta_bias, *start, *dout, *ckpt;
    float *du, *ddelta, *dz;
    float *dB_part, *dC_part, *dA_part, *dD_part, *dbias_part;     // workspace slabs
    int64_t ld_u, ld_delta, ld_z, ld_b, ld_c, ld_dout, ld_du, ld_ddelta, ld_dz;
    int B, L, Di, N, nck, softplus, nd;
};

// value held by lane (l ^ K), DPP where the pairing stays inside a 16-lane row
template <int K>
__device__ __forceinline__ float lane_xor(float v) {
    const int x = __builtin_bit_cast(int, v);
    int r;
    if constexpr (K == 1) {
        r = __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true);            // quad_perm [1,0,3,2]
    } else if constexpr (K == 2) {
        r = __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true);            // quad_perm [2,3,0,1]
    } else if constexpr (K == 4) {
        r = __builtin_amdgcn_update_dpp(0, x, 0x104, 0xF, 0x5, false);    // row_shl:4 -> banks 0,2
        r = __builtin_amdgcn_update_dpp(r, x, 0x114, 0xF, 0xA, false);    // row_shr:4 -> banks 1,3
    } else if constexpr (K == 8) {
        r = __builtin_amdgcn_update_dpp(0, x, 0x108, 0xF, 0x3, false);    // row_shl:8 -> banks 0,1
        r = __builtin_amdgcn_update_dpp(r, x, 0x118, 0xF, 0xC, false);    // row_shr:8 -> banks 2,3
    } else {
        r = __shfl_xor(x, K, 64);
    }
    return __builtin_bit_cast(float, r);
}

// Sum v[0..NS) over the 64 lanes of the wave with a multi-value butterfly: each of the first log2(NS) stages
// halves the number of values a lane carries.  On return v[0] of lane l is the wave total of the state
// j = state_of_lane<NS>(l) (all 64 lanes hold a total; lanes l < NS cover every j exactly once).
template <int NS>
__device__ __forceinline__ float butterfly_sum(float (&v)[NS], int lane) {
    if constexpr (NS >= 8) {
        const bool hi = lane & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float send = hi ? v[i] : v[i + 4];
            const float keep = hi ? v[i + 4] : v[i];
            v[i] = keep + lane_xor<1>(send);
        }
    }
    if constexpr (NS >= 4) {
        constexpr int K = NS >= 8 ? 2 : 1;
        const bool hi = lane & K;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float send = hi ? v[i] : v[i + 2];
            const float keep = hi ? v[i + 2] : v[i];
            v[i] = keep + lane_xor<K>(send);
        }
    }
    if constexpr (NS >= 2) {
        constexpr int K = NS >= 8 ? 4 : (NS >= 4 ? 2 : 1);
        const bool hi = lane & K;
        const float send = hi ? v[0] : v[1];
        const float keep = hi ? v[1] : v[0];
        v[0] = keep + lane_xor<K>(send);
    }
    float r = v[0];
    if constexpr (NS < 2) r += lane_xor<1>(r);
    if constexpr (NS < 4) r += lane_xor<2>(r);
    if constexpr (NS < 8) r += lane_xor<4>(r);
    r += lane_xor<8>(r);
    r += lane_xor<16>(r);
    r += lane_xor<32>(r);
    return r;
}
template <int NS>
__device__ __forceinline__ int state_of_lane(int lane) {
    if constexpr (NS == 8) return ((lane & 1) << 2) | (lane & 2) | ((lane >> 2) & 1);
    if constexpr (NS == 4) return ((lane & 1) << 1) | ((lane >> 1) & 1);
    if constexpr (NS == 2) return lane & 1;
    return 0;
}

template <int NS, int NW>
__global__ __launch_bounds__(NW * 64) void sscan_bwd_kernel(BwdParams p) {
    constexpr int NT = NW * 64;
    __shared__ __attribute__((aligned(16))) float s_dl[SC][TILE_C];
    __shared__ __attribute__((aligned(16))) float s_du[SC][TILE_C];
    __shared__ __attribute__((aligned(16))) float s_dy[SC][TILE_C];
    __shared__ __attribute__((aligned(16))) float s_part[3][NW][SC][TILE_C];

    int b, dt;
    if (!decode_block(blockIdx.x, p.nd, p.B, b, dt)) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int d0 = dt * TILE_C;
    const int d = d0 + lane;
    const bool d_ok = d < p.Di;
    const int N = NS * NW;
    const int64_t tok0 = (int64_t)b * p.L;
    const bool tile_thr = tid < SC * 16;            // threads that own one float4 of the [SC][64] tile
    const int tc4 = (tid & 15) * 4;
    const int tr = (tid >> 4) & (SC - 1);
    const bool c_ok = tile_thr && (d0 + tc4) < p.Di;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    float A2[NS], Aj[NS], dh[NS], dAacc[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const float a = d_ok ? p.A[(int64_t)d * N + w * NS + j] : -1.f;
        Aj[j] = a;
        A2[j] = fminf(a * RESEL_LOG2E, -1e-30f);
        dh[j] = 0.f;
        dAacc[j] = 0.f;
    }
    float4 Dv = zero4, bv = zero4, dDacc = zero4, dbacc = zero4;
    if (c_ok) {
        if (p.D) Dv = ld4(p.D + d0 + tc4);
        if (p.delta_bias) bv = ld4(p.delta_bias + d0 + tc4);
    }

    const int nchunk = (p.L + CKS - 1) / CKS;
    for (int k = nchunk - 1; k >= 0; --k) {
        const int tbase = k * CKS;
        const int cl = min(CKS, p.L - tbase);
        const int nsub = (cl + SC - 1) / SC;
        float hs[NSUB][NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            hs[0][j] = (k > 0 && d_ok)
                           ? p.ckpt[(((int64_t)b * p.nck + (k - 1)) * N + w * NS + j) * p.Di + d] : 0.f;
        }
        // ---------------- phase A: rebuild the sub-checkpoints hs[1..nsub-1] ----------------
#pragma unroll
        for (int s = 0; s < NSUB - 1; ++s) {
            if (s < nsub - 1) {
                __syncthreads();
                if (tile_thr) {
                    float4 dv = zero4, uv = zero4;
                    const int t = tbase + s * SC + tr;       // always < L here (a later sub-chunk exists)
                    if (c_ok) {
                        uv = ld4(p.u + (tok0 + t) * p.ld_u + d0 + tc4);
                        dv = ld4(p.delta + (tok0 + t) * p.ld_delta + d0 + tc4);
                        dv.x += bv.x; dv.y += bv.y; dv.z += bv.z; dv.w += bv.w;
                        if (p.softplus) {
                            dv.x = softplusf_(dv.x); dv.y = softplusf_(dv.y);
                            dv.z = softplusf_(dv.z); dv.w = softplusf_(dv.w);
                        }
                    }
                    st4(&s_dl[tr][tc4], dv);
                    st4(&s_du[tr][tc4], make_float4(dv.x * uv.x, dv.y * uv.y, dv.z * uv.z, dv.w * uv.w));
                }
                __syncthreads();
                float h[NS];
#pragma unroll
                for (int j = 0; j < NS; ++j) h[j] = hs[s][j];
                for (int i = 0; i < SC; ++i) {
                    const int64_t tok = tok0 + tbase + s * SC + i;
                    float Bc[NS];
                    load_coef<NS>(as_uniform(p.Bm + tok * p.ld_b + w * NS), Bc);
                    const float sf = p.start ? as_uniform(p.start + tok)[0] : 0.f;
                    const float dl = s_dl[i][lane], du = s_du[i][lane];
                    const float dle = (sf != 0.f) ? __builtin_inff() : dl;
#pragma unroll
                    for (int j = 0; j < NS; ++j) h[j] = __builtin_fmaf(fast_exp2(dle * A2[j]), h[j], du * Bc[j]);
                }
#pragma unroll
                for (int j = 0; j < NS; ++j) hs[s + 1][j] = h[j];
            }
        }
        // ---------------- phase B: sub-chunks in reverse ----------------
#pragma unroll
        for (int s = NSUB - 1; s >= 0; --s) {
            if (s < nsub) {
                const int sl = min(SC, cl - s * SC);          // steps in this sub-chunk
                const int ts = tbase + s * SC;
                __syncthreads();
                float4 u4 = zero4, draw4 = zero4, z4 = zero4, do4 = zero4, dl4 = zero4, dy4 = zero4;
                if (tile_thr) {
                    if (c_ok && tr < sl) {
                        const int64_t tok = tok0 + ts + tr;
                        u4 = ld4(p.u + tok * p.ld_u + d0 + tc4);
                        draw4 = ld4(p.delta + tok * p.ld_delta + d0 + tc4);
                        do4 = ld4(p.dout + tok * p.ld_dout + d0 + tc4);
                        if (p.z) z4 = ld4(p.z + tok * p.ld_z + d0 + tc4);
                        draw4.x += bv.x; draw4.y += bv.y; draw4.z += bv.z; draw4.w += bv.w;
                        dl4 = draw4;
                        if (p.softplus) {
                            dl4.x = softplusf_(dl4.x); dl4.y = softplusf_(dl4.y);
                            dl4.z = softplusf_(dl4.z); dl4.w = softplusf_(dl4.w);
                        }
                        dy4 = do4;
                        if (p.z) {
                            dy4.x *= siluf_(z4.x); dy4.y *= siluf_(z4.y); dy4.z *= siluf_(z4.z); dy4.w *= siluf_(z4.w);
                        }
                    }
                    st4(&s_dl[tr][tc4], dl4);
                    st4(&s_du[tr][tc4], make_float4(dl4.x * u4.x, dl4.y * u4.y, dl4.z * u4.z, dl4.w * u4.w));
                    st4(&s_dy[tr][tc4], dy4);
                }
                __syncthreads();
                // forward recomputation with the full state history in registers
                float hist[SC][NS];
                {
                    float h[NS];
#pragma unroll
                    for (int j = 0; j < NS; ++j) h[j] = hs[s][j];
#pragma unroll
                    for (int i = 0; i < SC; ++i) {
                        if (i < sl) {
                            const int64_t tok = tok0 + ts + i;
                            float Bc[NS];
                            load_coef<NS>(as_uniform(p.Bm + tok * p.ld_b + w * NS), Bc);
                            const float sf = p.start ? as_uniform(p.start + tok)[0] : 0.f;
                            const float dl = s_dl[i][lane], du = s_du[i][lane];
                            const float dle = (sf != 0.f) ? __builtin_inff() : dl;
#pragma unroll
                            for (int j = 0; j < NS; ++j) {
                                h[j] = __builtin_fmaf(fast_exp2(dle * A2[j]), h[j], du * Bc[j]);
                                hist[i][j] = h[j];
                            }
                            __builtin_amdgcn_sched_barrier(0);   // keep the unrolled steps in order (register pressure)
                        }
                    }
                }
                // reverse time
#pragma unroll
                for (int i = SC - 1; i >= 0; --i) {
                    if (i < sl) {
                        const int64_t tok = tok0 + ts + i;
                        float Bc[NS], Cc[NS];
                        load_coef<NS>(as_uniform(p.Bm + tok * p.ld_b + w * NS), Bc);
                        load_coef<NS>(as_uniform(p.Cm + tok * p.ld_c + w * NS), Cc);
                        const float sf = p.start ? as_uniform(p.start + tok)[0] : 0.f;
                        const float dl = s_dl[i][lane], du = s_du[i][lane], dy = s_dy[i][lane];
                        const float dle = (sf != 0.f) ? __builtin_inff() : dl;
                        float P1 = 0.f, P2 = 0.f, P3 = 0.f;
                        float dBv[NS], dCv[NS];
#pragma unroll
                        for (int j = 0; j < NS; ++j) {
                            const float hj = hist[i][j];
                            const float hp = (i == 0) ? hs[s][j] : hist[i > 0 ? i - 1 : 0][j];
                            dh[j] = __builtin_fmaf(dy, Cc[j], dh[j]);
                            P3 = __builtin_fmaf(Cc[j], hj, P3);
                            const float dA = fast_exp2(dle * A2[j]);          // 0 at a reset step
                            const float tmp = dh[j] * hp * dA;                // dL/d(dA) * dA
                            P2 = __builtin_fmaf(tmp, Aj[j], P2);
                            dAacc[j] = __builtin_fmaf(tmp, dl, dAacc[j]);
                            dBv[j] = dh[j] * du;
                            dCv[j] = dy * hj;
                            P1 = __builtin_fmaf(dh[j], Bc[j], P1);
                            dh[j] *= dA;
                        }
                        s_part[0][w][i][lane] = P1;
                        s_part[1][w][i][lane] = P2;
                        s_part[2][w][i][lane] = P3;
                        const float rb = butterfly_sum<NS>(dBv, lane);
                        const float rc = butterfly_sum<NS>(dCv, lane);
                        if (lane < NS) {
                            const int64_t o = ((int64_t)dt * p.B * p.L + tok) * N + w * NS + state_of_lane<NS>(lane);
                            p.dB_part[o] = rb;
                            p.dC_part[o] = rc;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __syncthreads();
                // epilogue on the tile mapping
                if (c_ok && tr < sl) {
                    float4 P1 = zero4, P2 = zero4, P3 = zero4;
#pragma unroll
                    for (int ww = 0; ww < NW; ++ww) {
                        const float4 a = ld4(&s_part[0][ww][tr][tc4]);
                        const float4 bq = ld4(&s_part[1][ww][tr][tc4]);
                        const float4 c = ld4(&s_part[2][ww][tr][tc4]);
                        P1.x += a.x; P1.y += a.y; P1.z += a.z; P1.w += a.w;
                        P2.x += bq.x; P2.y += bq.y; P2.z += bq.z; P2.w += bq.w;
                        P3.x += c.x; P3.y += c.y; P3.z += c.z; P3.w += c.w;
                    }
                    const int64_t tok = tok0 + ts + tr;
                    float4 o;
                    // du = delta' * sum_n dh B + D * dy
                    o.x = dl4.x * P1.x + Dv.x * dy4.x; o.y = dl4.y * P1.y + Dv.y * dy4.y;
                    o.z = dl4.z * P1.z + Dv.z * dy4.z; o.w = dl4.w * P1.w + Dv.w * dy4.w;
                    st4(p.du + tok * p.ld_du + d0 + tc4, o);
                    // d delta' = sum_n (dh h_prev dA) A + u * sum_n dh B ; chain through softplus
                    float4 g;
                    g.x = P2.x + u4.x * P1.x; g.y = P2.y + u4.y * P1.y; g.z = P2.z + u4.z * P1.z; g.w = P2.w + u4.w * P1.w;
                    if (p.softplus) {
                        g.x *= sigmoidf_(draw4.x); g.y *= sigmoidf_(draw4.y); g.z *= sigmoidf_(draw4.z); g.w *= sigmoidf_(draw4.w);
                    }
                    st4(p.ddelta + tok * p.ld_ddelta + d0 + tc4, g);
                    dbacc.x += g.x; dbacc.y += g.y; dbacc.z += g.z; dbacc.w += g.w;
                    dDacc.x += dy4.x * u4.x; dDacc.y += dy4.y * u4.y; dDacc.z += dy4.z * u4.z; dDacc.w += dy4.w * u4.w;
                    if (p.z) {
                        float4 yy;                     // pre-gate output y = sum_n C h + D u
                        yy.x = P3.x + Dv.x * u4.x; yy.y = P3.y + Dv.y * u4.y; yy.z = P3.z + Dv.z * u4.z; yy.w = P3.w + Dv.w * u4.w;
                        float4 gz;
                        gz.x = do4.x * yy.x * dsiluf_(z4.x); gz.y = do4.y * yy.y * dsiluf_(z4.y);
                        gz.z = do4.z * yy.z * dsiluf_(z4.z); gz.w = do4.w * yy.w * dsiluf_(z4.w);
                        st4(p.dz + tok * p.ld_dz + d0 + tc4, gz);
                    }
                }
            }
        }
    }
    // ---- per-(b) partials of the parameter gradients
    if (d_ok) {
#pragma unroll
        for (int j = 0; j < NS; ++j) p.dA_part[((int64_t)b * p.Di + d) * N + w * NS + j] = dAacc[j];
    }
    __syncthreads();
    float* s_red = &s_part[0][0][0][0];               // [2][SC][64] scratch
    if (tile_thr) {
        st4(&s_red[(0 * SC + tr) * TILE_C + tc4], dDacc);
        st4(&s_red[(1 * SC + tr) * TILE_C + tc4], dbacc);
    }
    __syncthreads();
    if (tid < 2 * TILE_C) {
        const int which = tid >> 6, c = tid & 63;
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < SC; ++r) acc += s_red[(which * SC + r) * TILE_C + c];
        if (d0 + c < p.Di) (which ? p.dbias_part : p.dD_part)[(int64_t)b * p.Di + d0 + c] = acc;
    }
}

// dB / dC: sum the per-channel-tile slabs; parameter gradients: sum the per-row partials.
__global__ void sscan_reduce_bc_kernel(const float* __restrict__ part, int nd, int64_t ntok, int N, float* out, int64_t ld) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // float4 index over [ntok, N]
    const int n4 = N / 4;
    if (i >= ntok * n4) return;
    const int64_t tok = i / n4;
    const int c = (int)(i % n4) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < nd; ++t) {
        const float4 v = ld4(part + ((int64_t)t * ntok + tok) * N + c);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    float* o = out + tok * ld + c;
    o[0] = acc.x; o[1] = acc.y; o[2] = acc.z; o[3] = acc.w;
}
__global__ void sscan_reduce_rows_kernel(const float* __restrict__ part, int B, int64_t n, float* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc += part[(int64_t)b * n + i];
    out[i] = acc;
}

template <int NS, int NW>
int launch_fwd(const FwdParams& p, hipStream_t s) {
    const int bp = (p.B + 7) / 8 * 8;
    launch_maybe_timed(0, sscan_fwd_kernel<NS, NW>, dim3(bp * p.nd), dim3(NW * 64), s, p);
    return launch_status();
}
template <int NS, int NW>
int launch_bwd(const BwdParams& p, hipStream_t s) {
    const int bp = (p.B + 7) / 8 * 8;
    launch_maybe_timed(1, sscan_bwd_kernel<NS, NW>, dim3(bp * p.nd), dim3(NW * 64), s, p);
    return launch_status();
}

inline int n_ckpt(int L) { return (L - 1) / CKS; }   // checkpoints after steps CKS, 2*CKS, ... (< L)
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct BwdWs { size_t dB, dC, dA, dD, dbias, total; };
inline BwdWs bwd_ws(int B, int L, int Di, int N) {
    const size_t nd = (Di + TILE_C - 1) / TILE_C;
    BwdWs w;
    size_t o = 0;
    w.dB = o; o += align256(nd * (size_t)B * L * N * 4);
    w.dC = o; o += align256(nd * (size_t)B * L * N * 4);
    w.dA = o; o += align256((size_t)B * Di * N * 4);
    w.dD = o; o += align256((size_t)B * Di * 4);
    w.dbias = o; o += align256((size_t)B * Di * 4);
    w.total = o;
    return w;
}

}  // namespace

extern "C" int resel_profile_enable(int on) {
    g_prof_on = on != 0;
    return RESEL_OK;
}

extern "C" int resel_profile_collect(int kernel_id, double* total_us, int* launches) {
    if (kernel_id < 0 || kernel_id > 1 || !total_us || !launches) return RESEL_EINVAL;
    double tot = 0.0;
    int n = 0;
    for (auto& pr : g_prof[kernel_id].ev) {
        float ms = 0.f;
        if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
            tot += 1e3 * ms;
            ++n;
        }
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    g_prof[kernel_id].ev.clear();
    *total_us = tot;
    *launches = n;
    return RESEL_OK;
}

extern "C" size_t resel_selective_scan_ckpt_bytes(int B, int L, int Di, int N) {
    return (size_t)B * (size_t)(n_ckpt(L) > 0 ? n_ckpt(L) : 0) * (size_t)N * (size_t)Di * sizeof(float);
}

extern "C" int resel_selective_scan_fwd(const float* u, int64_t ld_u, const float* delta, int64_t ld_delta,
                                        const float* z, int64_t ld_z, const float* A,
                                        const float* Bm, int64_t ld_b, const float* Cm, int64_t ld_c,
                                        const float* D, const float* delta_bias, const float* start,
                                        float* out, int64_t ld_out, float* ckpt, float* last_state,
                                        int B, int L, int Di, int N, int delta_softplus, resel_stream_t stream) {
    if (!u || !delta || !A || !Bm || !Cm || !out || B <= 0 || L <= 0 || Di <= 0) return RESEL_EINVAL;
    if (Di % 4 != 0 || ld_u % 4 || ld_delta % 4 || ld_out % 4 || (z && ld_z % 4)) return RESEL_EINVAL;
    if (!aligned16(u) || !aligned16(delta) || !aligned16(out) || (z && !aligned16(z))) return RESEL_EINVAL;
    if ((D && !aligned16(D)) || (delta_bias && !aligned16(delta_bias))) return RESEL_EINVAL;
    FwdParams p{u, delta, z, A, Bm, Cm, D, delta_bias, start, out, ckpt, last_state,
                ld_u, ld_delta, ld_z, ld_b, ld_c, ld_out, B, L, Di, N, n_ckpt(L), delta_softplus,
                (Di + TILE_C - 1) / TILE_C};
    hipStream_t s = (hipStream_t)stream;
    switch (N) {
        case 4: return launch_fwd<1, 4>(p, s);
        case 8: return launch_fwd<2, 4>(p, s);
        case 16: return launch_fwd<4, 4>(p, s);
        case 32: return launch_fwd<8, 4>(p, s);
        case 64: return launch_fwd<8, 8>(p, s);
        default: return RESEL_EINVAL;
    }
}

extern "C" size_t resel_selective_scan_bwd_workspace_bytes(int B, int L, int Di, int N) {
    return bwd_ws(B, L, Di, N).total;
}

extern "C" int resel_selective_scan_bwd(const float* u, int64_t ld_u, const float* delta, int64_t ld_delta,
                                        const float* z, int64_t ld_z, const float* A,
                                        const float* Bm, int64_t ld_b, const float* Cm, int64_t ld_c,
                                        const float* D, const float* delta_bias, const float* start,
                                        const float* dout, int64_t ld_dout, const float* ckpt,
                                        float* du, int64_t ld_du, float* ddelta, int64_t ld_ddelta,
                                        float* dz, int64_t ld_dz, float* dBm, int64_t ld_db, float* dCm, int64_t ld_dc,
                                        float* dA, float* dD, float* ddelta_bias, void* workspace,
                                        int B, int L, int Di, int N, int delta_softplus, resel_stream_t stream) {
    if (!u || !delta || !A || !Bm || !Cm || !dout || !du || !ddelta || !dBm || !dCm || !dA || !workspace)
        return RESEL_EINVAL;
    if (B <= 0 || L <= 0 || Di <= 0 || Di % 4 != 0 || N % 4 != 0) return RESEL_EINVAL;
    if ((z != nullptr) != (dz != nullptr)) return RESEL_EINVAL;
    if (n_ckpt(L) > 0 && !ckpt) return RESEL_EINVAL;
    if (ld_u % 4 || ld_delta % 4 || ld_dout % 4 || ld_du % 4 || ld_ddelta % 4 || (z && (ld_z % 4 || ld_dz % 4)))
        return RESEL_EINVAL;
    if (!aligned16(u) || !aligned16(delta) || !aligned16(dout) || !aligned16(du) || !aligned16(ddelta) ||
        (z && (!aligned16(z) || !aligned16(dz))) || (D && !aligned16(D)) || (delta_bias && !aligned16(delta_bias)) ||
        !aligned16(workspace))
        return RESEL_EINVAL;
    const BwdWs ws = bwd_ws(B, L, Di, N);
    char* base = (char*)workspace;
    BwdParams p{u, delta, z, A, Bm, Cm, D, delta_bias, start, dout, ckpt, du, ddelta, dz,
                (float*)(base + ws.dB), (float*)(base + ws.dC), (float*)(base + ws.dA), (float*)(base + ws.dD),
                (float*)(base + ws.dbias),
                ld_u, ld_delta, ld_z, ld_b, ld_c, ld_dout, ld_du, ld_ddelta, ld_dz,
                B, L, Di, N, n_ckpt(L), delta_softplus, (Di + TILE_C - 1) / TILE_C};
    hipStream_t s = (hipStream_t)stream;
    int rc;
    switch (N) {
        case 4: rc = launch_bwd<1, 4>(p, s); break;
        case 8: rc = launch_bwd<2, 4>(p, s); break;
        case 16: rc = launch_bwd<4, 4>(p, s); break;
        case 32: rc = launch_bwd<4, 8>(p, s); break;     // 8 waves x 4 states: halves the register-resident history
        case 64: rc = launch_bwd<8, 8>(p, s); break;
        default: return RESEL_EINVAL;
    }
    if (rc != RESEL_OK) return rc;
    const int64_t ntok = (int64_t)B * L;
    const int64_t n4 = ntok * (N / 4);
    hipLaunchKernelGGL(sscan_reduce_bc_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s,
                       p.dB_part, p.nd, ntok, N, dBm, ld_db);
    hipLaunchKernelGGL(sscan_reduce_bc_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s,
                       p.dC_part, p.nd, ntok, N, dCm, ld_dc);
    const int64_t na = (int64_t)Di * N;
    hipLaunchKernelGGL(sscan_reduce_rows_kernel, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, s, p.dA_part, B, na, dA);
    if (dD) hipLaunchKernelGGL(sscan_reduce_rows_kernel, dim3((Di + 255) / 256), dim3(256), 0, s, p.dD_part, B, (int64_t)Di, dD);
    if (ddelta_bias)
        hipLaunchKernelGGL(sscan_reduce_rows_kernel, dim3((Di + 255) / 256), dim3(256), 0, s, p.dbias_part, B, (int64_t)Di, ddelta_bias);
    return launch_status();
}
