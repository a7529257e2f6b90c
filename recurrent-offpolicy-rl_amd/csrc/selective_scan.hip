// Selective scan (Mamba S6 with `start` resets) for gfx950 - forward and backward.
//
// Layout / mapping (DESIGN.md section "selective_scan"):
//   * activations token-major [B*L, Di]; one workgroup owns (row b, 64-channel tile) for the whole sequence;
//   * lane  <-> channel (64 lanes = 64 consecutive channels: every global access of a wave is one or more
//     fully used 256-byte segments);
//   * wave  <-> group of NS states: the N = NS*NW state columns of a channel are split over the NW waves, so
//     B_t / C_t of a step are WAVE-UNIFORM: their rows are staged in LDS with the chunk's tiles and read with
//     broadcast ds_read_b128 one step ahead of their use (scalar loads into SGPR operands were tried and dropped:
//     SMEM latency is not covered by one step of look-ahead);
//   * time is processed in chunks staged through LDS: a coalesced float4 tile load applies softplus / the
//     delta*u product once per element, the scan phase reads one dword per lane per step, the per-wave
//     partial sums meet again in LDS and the output tile leaves with float4 stores;
//   * block id -> (b, channel tile) is XCD-aware: blocks are dealt round-robin over the 8 XCDs, so the
//     Di/64 channel tiles of one row b are given ids that differ by multiples of 8 and share one L2 for the
//     B_t / C_t rows they all read (speed only; any placement is correct).
// Backward: reverse-time recurrence with recomputation - the forward leaves a state checkpoint every 8 steps
// (straight from the state registers; the forward is VALU-bound, not HBM-bound), the backward stages a 16-step sub-chunk in
// LDS, replays each of its 8-step halves ONCE from that half's checkpoint with the (state, decay) history in registers, and
// sweeps the half in reverse.  Reductions over channels (dB, dC) use an in-wave multi-value butterfly (DPP) plus per-tile partial
// slabs summed by a second kernel: no float atomics, bitwise reproducible.
// Rates that shape both kernels (tools/micro/valu_rate2.hip, MI355X, issue cycles per wave64 instruction on one SIMD):
// v_fma/v_mul_f32 2.25 with two waves per SIMD (4.5 alone), v_pk_fma/v_pk_mul_f32 4.5 (two results), v_exp_f32 8.2 -
// the per-state arithmetic is written on float2 pairs, and with N = 32 states the kernels are bound by vector issue
// (and, with every CU busy, by the clock the chip holds under that load) well below the HBM roofline.
#include "resel_common.h"
#include <hip/hip_ext.h>
#include <type_traits>
#include <vector>

namespace {
using namespace resel;

template <typename K, typename P>
void launch_maybe_timed(int slot, K kernel, dim3 grid, dim3 block, hipStream_t s, const P& p) {
    launch_timed(slot, kernel, grid, block, 0, s, p);
}
#ifdef SSCAN_STAMP
unsigned long long* g_stamps = nullptr;
#endif

constexpr int TILE_C = 64;            // channels per workgroup
constexpr int CKS = RESEL_SSCAN_CKPT; // checkpoint stride
constexpr int SC = 16;                // backward sub-chunk = one checkpoint interval staged in LDS
constexpr int SCH = 8;                // steps of (h, dA) history kept in registers
static_assert(2 * CKS == SC && CKS == SCH, "a backward sub-chunk is two checkpoint intervals: each 8-step half starts from a stored state");

struct FwdParams {
    const float *u, *delta, *z, *A, *Bm, *Cm, *D, *delta_bias, *start;
    float *out, *ckpt, *last_state;
    int64_t ld_u, ld_delta, ld_z, ld_b, ld_c, ld_out;
    int B, L, Di, N, nck, softplus, nd, bc_vec;
    int a_log;                      // A holds A_log: the kernels form A = -exp(A_log) themselves (delta_softplus bit 2)
    int seg_len, nseg;              // time-parallel form (MODE 1 / 2): steps per segment (multiple of the chunk), segments per row
    float *h_carry, *sdl;           // [B][nseg][N][Di] local end states -> entry states; [B][nseg][Di] delta sums
    AmaxOut amax_out;               // optional: publish max |out| (the out_proj GEMM scales its operand with it)
#ifdef SSCAN_STAMP
    unsigned long long* stamps;     // diagnostic build only (tools/micro/sscan_lab.hip): per-wave phase cycle sums
#endif
};

#ifdef SSCAN_STAMP
#define STAMP(acc, last) do { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    (acc) += t_ - (last); (last) = t_; } while (0)
#else
#define STAMP(acc, last) do { } while (0)
#endif

template <int NS>
__device__ __forceinline__ void load_coef(cfloat_p p, float (&dst)[NS]) {
#pragma unroll
    for (int j = 0; j < NS; ++j) dst[j] = p[j];
}

// A[i], or -exp(A_log[i]) when the caller passes the parameter itself (reference smamba/mamba.py:187 `A = -torch.exp(self.A_log.float())`:
// two element-wise launches per mixer call, and a third for dA_log = dA * A in the backward)
__device__ __forceinline__ float load_A(const float* A, int64_t i, int a_log) {
    const float a = A[i];
    return a_log ? -expf(a) : a;
}

// XCD-aware decode of a 1-D block id into (row b, channel tile dt): ids congruent mod 8 share an XCD.
__device__ __forceinline__ bool decode_block(int id, int nd, int B, int& b, int& dt) {
    const int xcd = id & 7, k = id >> 3;
    dt = k % nd;
    b = (k / nd) * 8 + xcd;
    return b < B;
}

// NS wave-uniform coefficients from LDS (every lane reads the same address: one broadcast ds_read_b128 / b64 / b32)
template <int NS>
__device__ __forceinline__ void lds_coef(const float* p, float (&dst)[NS]) {
    if constexpr (NS % 4 == 0) {
#pragma unroll
        for (int j = 0; j < NS; j += 4) {
            const float4 v = ld4(p + j);
            dst[j] = v.x; dst[j + 1] = v.y; dst[j + 2] = v.z; dst[j + 3] = v.w;
        }
    } else if constexpr (NS == 2) {
        const float2 v = *reinterpret_cast<const float2*>(p);
        dst[0] = v.x; dst[1] = v.y;
    } else {
#pragma unroll
        for (int j = 0; j < NS; ++j) dst[j] = p[j];
    }
}

typedef float f2 __attribute__((ext_vector_type(2)));

// the same as float2 pairs (NS odd: the upper lane of the last pair is zero)
template <int NS>
__device__ __forceinline__ void lds_coef2(const float* p, f2 (&dst)[(NS + 1) / 2]) {
    if constexpr (NS % 4 == 0) {
#pragma unroll
        for (int j = 0; j < NS; j += 4) {
            const float4 v = ld4(p + j);
            dst[j / 2] = f2{v.x, v.y};
            dst[j / 2 + 1] = f2{v.z, v.w};
        }
    } else if constexpr (NS == 2) {
        const float2 v = *reinterpret_cast<const float2*>(p);
        dst[0] = f2{v.x, v.y};
    } else {
        dst[0] = f2{p[0], 0.f};
    }
}

// 4 consecutive floats of a [tok, N] coefficient matrix with token stride ld (vector load when the layout allows it)
__device__ __forceinline__ float4 load_bc4(const float* base, int64_t tok, int64_t ld, int c, bool vec) {
    const float* q = base + tok * ld + c;
    if (vec) return ld4(q);
    return make_float4(q[0], q[1], q[2], q[3]);
}

// packed multiplies whose first operand is ONE half of a register pair, broadcast to both results (op_sel): {p.x, p.x} * x and {p.y, p.y} * x
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2v pk_mul_lo(f2v p, f2v x) {         // {p.x * x.x, p.x * x.y}
    f2v r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(p), "v"(x));
    return r;
}
__device__ __forceinline__ f2v pk_mul_hi(f2v p, f2v x) {         // {p.y * x.x, p.y * x.y}
    f2v r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(p), "v"(x));
    return r;
}

// a lane's NS states -> its checkpoint slot (NS contiguous floats; 16-byte stores when NS is a multiple of 4)
template <int NS>
__device__ __forceinline__ void store_ckpt(float* q, const f2 (&hp)[(NS + 1) / 2]) {
    if constexpr (NS % 4 == 0) {
#pragma unroll
        for (int i = 0; i < NS / 4; ++i) st4(q + 4 * i, make_float4(hp[2 * i].x, hp[2 * i].y, hp[2 * i + 1].x, hp[2 * i + 1].y));
    } else {
#pragma unroll
        for (int j = 0; j < NS; ++j) q[j] = (j & 1) ? hp[j / 2].y : hp[j / 2].x;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Forward.  Ownership: workgroup = (row b, 64 channels[, time segment]), lane = channel, wave = NS-state group, time in
// TC-step chunks staged through LDS.  What keeps the per-step overhead down (phase stamps / PMC: tools/micro/sscan_lab.hip):
//   * resets are applied when the chunk is STAGED (delta := +inf at a start step, so exp2(delta * A) = 0): no select, no
//     mask arithmetic in the step;
//   * the step loop is straight-line over the whole chunk (rows past the end of the sequence are staged as identity steps:
//     delta = 0, delta * u = 0), so the compiler software-pipelines across steps and pairs the y stores;
//   * checkpoints are collected in LDS and leave with the output tile as whole 256-byte rows;
//   * the prefetched tiles of the next chunk are waited for BEFORE this chunk's output stores are issued (hipcc otherwise
//     guards their first use with s_waitcnt vmcnt(0) and every chunk eats a full store round trip), and the data registers
//     of those stores stay live until the end of the next scan phase;
//   * softplus / SiLU of the tile passes are branch-free.
// MODE 0: one workgroup scans a whole row (B * Di / 64 workgroups: enough to fill the chip from B ~ 32 at Di = 512).
// Time-parallel form for small batches (north_star: parallel scan over the sequence): the row is cut into `nseg` segments of
// `seg_len` steps (a multiple of TC), each segment is its own workgroup and the scan becomes
//   MODE 1  local pass: every segment scans from a zero state and leaves its end state h_loc and the sum S of its deltas
//           (+inf after a reset) - no output, no checkpoints, no C operand;
//   carry   (sscan_carry_kernel) per (row, channel, state): h_in[s] = state entering segment s, by the segment recurrence
//           h_in[s + 1] = exp2(A2 * S[s]) * h_in[s] + h_loc[s]  (the product of a segment's decays is the exp of the sum);
//   MODE 2  final pass: every segment rescans from its true entry state and produces outputs and checkpoints exactly as MODE 0.
// ~1.9x the arithmetic of MODE 0 on nseg x the workgroups; chosen by the launcher when the one-pass grid cannot fill the CUs.
// OCC: waves per SIMD the register allocation must leave room for (the second argument of HIP's __launch_bounds__).  2 = the shipped form
// (TC = 32: 216 VGPRs, 56 KB of LDS).  3 with TC = 16 (162 VGPRs, 28 KB): a THIRD workgroup per CU, for grids that have one to give -
// B x Di / 64 >= 768 workgroups, i.e. B >= 96 rows at Di = 512.  At configs[1] (B = 64: exactly two workgroups per CU, profiles/r06_sscan_pmc.md)
// it cannot help and the shorter chunks cost two more barriers per 32 steps, so the launcher keeps TC = 32 there.
template <int NS, int NW, int TC, int MODE, int OCC = 2>
__global__ __launch_bounds__(NW * 64, OCC) void sscan_fwd2_kernel(FwdParams p) {
    constexpr int NT = NW * 64;
    constexpr int N = NS * NW;
    constexpr int NP = (NS + 1) / 2;
    constexpr int PER_T = (TC * 16 + NT - 1) / NT;
    constexpr int BC_ITEMS = TC * N / 4;
    constexpr int PER_BC = (BC_ITEMS + NT - 1) / NT;
    static_assert(TC % 2 == 0 && (TC * 16) % NT == 0, "whole tile rows per thread pass");
    // per (step, channel) ONE 8-byte word {exp argument softplus(delta + bias) (+inf at a reset), that * u}: a lane's step operands
    // arrive as a register pair whose halves feed the packed multiplies through op_sel (pk_mul_lo / pk_mul_hi: no splat moves)
    __shared__ __attribute__((aligned(16))) float s_p[TC][TILE_C][2];
    __shared__ __attribute__((aligned(16))) float s_y[MODE == 1 ? 1 : NW][TC][TILE_C];
    __shared__ __attribute__((aligned(16))) float s_B[TC][N];
    __shared__ __attribute__((aligned(16))) float s_C[MODE == 1 ? 1 : TC][N];

    int b, dt;
    if (!decode_block(blockIdx.x, p.nd, p.B, b, dt)) return;
    const int seg = MODE == 0 ? 0 : (int)blockIdx.y;
    const int t_begin = MODE == 0 ? 0 : seg * p.seg_len;
    const int t_end = MODE == 0 ? p.L : min(p.L, t_begin + p.seg_len);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int d0 = dt * TILE_C;
    const int d = d0 + lane;
    const bool d_ok = d < p.Di;
    const int64_t tok0 = (int64_t)b * p.L;

    f2 A2p[NP], hp[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        float a[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int j = 2 * k + e;
            a[e] = (d_ok && j < NS) ? fminf(load_A(p.A, (int64_t)d * N + w * NS + j, p.a_log) * RESEL_LOG2E, -1e-30f) : -1.f;
        }
        A2p[k] = f2{a[0], a[1]};
        hp[k] = f2{0.f, 0.f};
        if (MODE == 2 && d_ok) {                     // the state entering this segment (sscan_carry_kernel)
            const float* hin = p.h_carry + (((int64_t)b * p.nseg + seg) * N + w * NS) * p.Di + d;
            hp[k] = f2{2 * k < NS ? hin[(int64_t)(2 * k) * p.Di] : 0.f, 2 * k + 1 < NS ? hin[(int64_t)(2 * k + 1) * p.Di] : 0.f};
        }
    }
    float sdl = 0.f;                                 // MODE 1: sum of this lane's deltas over the segment (+inf after a reset)
    const int tc4 = (tid & 15) * 4;
    const int tr0 = tid >> 4;
    const bool c_ok = (d0 + tc4) < p.Di;
    float4 Dv = make_float4(0.f, 0.f, 0.f, 0.f), bv = Dv;
    if (c_ok) {
        if (p.D) Dv = ld4(p.D + d0 + tc4);
        if (p.delta_bias) bv = ld4(p.delta_bias + d0 + tc4);
    }
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 pu[PER_T], pd[PER_T], pz[PER_T];
    float4 pB[PER_BC], pC[PER_BC];
    float pst[PER_T];
    float4 u_r[PER_T], z_r[PER_T];

    const bool tile_full = d0 + TILE_C <= p.Di;     // uniform: every thread's four channels exist
    auto prefetch = [&](int c0) {
        if (tile_full && c0 + TC <= t_end) {        // a whole chunk of a whole channel tile (uniform): no zero fill, no guards
#pragma unroll
            for (int i = 0; i < PER_T; ++i) {
                const int64_t tok = tok0 + c0 + tr0 + i * (NT / 16);
                pu[i] = ld4(p.u + tok * p.ld_u + d0 + tc4);
                pd[i] = ld4(p.delta + tok * p.ld_delta + d0 + tc4);
                pz[i] = (MODE != 1 && p.z) ? ld4(p.z + tok * p.ld_z + d0 + tc4) : zero4;
                pst[i] = p.start ? p.start[tok] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < PER_BC; ++i) {
                const int it = tid + i * NT;
                const int t = c0 + it / (N / 4), c = (it % (N / 4)) * 4;
                if (PER_BC * NT == BC_ITEMS || it < BC_ITEMS) {
                    pB[i] = load_bc4(p.Bm, tok0 + t, p.ld_b, c, p.bc_vec);
                    if (MODE != 1) pC[i] = load_bc4(p.Cm, tok0 + t, p.ld_c, c, p.bc_vec);
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int t = c0 + tr0 + i * (NT / 16);
            pu[i] = zero4; pd[i] = zero4; pz[i] = zero4; pst[i] = 0.f;
            if (t < t_end) {
                const int64_t tok = tok0 + t;
                if (c_ok) {
                    pu[i] = ld4(p.u + tok * p.ld_u + d0 + tc4);
                    pd[i] = ld4(p.delta + tok * p.ld_delta + d0 + tc4);
                    if (MODE != 1 && p.z) pz[i] = ld4(p.z + tok * p.ld_z + d0 + tc4);
                }
                if (p.start) pst[i] = p.start[tok];
            }
        }
#pragma unroll
        for (int i = 0; i < PER_BC; ++i) {
            const int it = tid + i * NT;
            const int t = c0 + it / (N / 4), c = (it % (N / 4)) * 4;
            pB[i] = zero4; pC[i] = zero4;
            if (it < BC_ITEMS && t < t_end) {
                pB[i] = load_bc4(p.Bm, tok0 + t, p.ld_b, c, p.bc_vec);
                if (MODE != 1) pC[i] = load_bc4(p.Cm, tok0 + t, p.ld_c, c, p.bc_vec);
            }
        }
    };
    prefetch(t_begin);
    auto retire_prefetch = [&]() {                 // make hipcc wait for the prefetched tiles HERE (see the header comment)
#pragma unroll
        for (int i = 0; i < PER_T; ++i) asm volatile("" :: "v"(pu[i].x), "v"(pd[i].x), "v"(pz[i].x), "v"(pst[i]));
#pragma unroll
        for (int i = 0; i < PER_BC; ++i) asm volatile("" :: "v"(pB[i].x), "v"(pC[i].x));
    };
    retire_prefetch();
    // Data registers of the output-phase stores stay LIVE until the end of the next scan phase: hipcc guards the first
    // overwrite of a store's data register with s_waitcnt vmcnt - placed there, the stores have long completed.
    float4 y_keep[PER_T];
    float omax = 0.f;                                // max |out| of this thread's stores
#pragma unroll
    for (int i = 0; i < PER_T; ++i) y_keep[i] = zero4;
    auto release_store_data = [&]() {
#pragma unroll
        for (int i = 0; i < PER_T; ++i) asm volatile("" :: "v"(y_keep[i].x), "v"(y_keep[i].y), "v"(y_keep[i].z), "v"(y_keep[i].w));
    };
    // checkpoints (the state after every CKS = 8 steps, so that each 8-step half of a backward sub-chunk starts from a stored state
    // instead of a replay): one dword per lane and state straight from the state registers - a wave's row of a state is 256
    // contiguous bytes, and gfx950 needs no wait between a store and the next write of its data register
    // layout [row][checkpoint][wave][channel][NS states]: a lane's NS states are contiguous (two 16-byte stores at NS = 8, the
    // backward fetches them with two LDS-DMA pieces), a wave's store is one contiguous 64 * 4 NS byte run
    float* ck_row = (MODE != 1 && p.ckpt != nullptr && d_ok) ? p.ckpt + ((((int64_t)b * p.nck + t_begin / CKS) * NW + w) * p.Di + d) * NS : nullptr;
#ifdef SSCAN_STAMP
    unsigned long long st_stage = 0, st_b1 = 0, st_scan = 0, st_b2 = 0, st_out = 0, st_last = 0, st_dummy = 0;
    const unsigned long long st_r0 = __builtin_amdgcn_s_memrealtime();
    STAMP(st_dummy, st_last);
    const unsigned long long st_t0 = st_last;
#endif

    for (int c0 = t_begin; c0 < t_end; c0 += TC) {
        // ---- stage
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int r = tr0 + i * (NT / 16);
            float4 dv = pd[i];
            const float4 uv = pu[i];
            dv.x += bv.x; dv.y += bv.y; dv.z += bv.z; dv.w += bv.w;
            if (p.softplus == 1) {
                dv.x = softplus_nb(dv.x); dv.y = softplus_nb(dv.y); dv.z = softplus_nb(dv.z); dv.w = softplus_nb(dv.w);
            }
            float4 du4 = make_float4(dv.x * uv.x, dv.y * uv.y, dv.z * uv.z, dv.w * uv.w);
            if (c0 + TC > t_end) {                                            // last chunk only (uniform branch)
                if (c0 + r >= t_end) { dv = zero4; du4 = zero4; }            // identity step past the end of the row / segment
            }
            {                                                                 // reset: exp2(-inf * |A|) = 0 wipes the carried state
                const bool rs = pst[i] != 0.f;
                const float inf = __builtin_inff();
                dv.x = rs ? inf : dv.x; dv.y = rs ? inf : dv.y; dv.z = rs ? inf : dv.z; dv.w = rs ? inf : dv.w;
            }
            u_r[i] = uv;
            z_r[i] = pz[i];
            st4(&s_p[r][tc4][0], make_float4(dv.x, du4.x, dv.y, du4.y));
            st4(&s_p[r][tc4 + 2][0], make_float4(dv.z, du4.z, dv.w, du4.w));
        }
#pragma unroll
        for (int i = 0; i < PER_BC; ++i) {
            const int it = tid + i * NT;
            if (it < BC_ITEMS) {
                st4(&s_B[0][0] + it * 4, pB[i]);
                if (MODE != 1) st4(&s_C[0][0] + it * 4, pC[i]);
            }
        }
        STAMP(st_stage, st_last);
        __syncthreads();
        STAMP(st_b1, st_last);
        prefetch(c0 + TC);                           // in flight during the whole scan phase (past the end: every load is guarded off)

        // ---- scan
        f2v Pr[TC];
#pragma unroll
        for (int t = 0; t < TC; ++t) Pr[t] = *reinterpret_cast<const f2v*>(&s_p[t][lane][0]);
        f2 Bq[2][NP], Cq[2][NP];                      // operand rows one step ahead of their use (two register sets)
        lds_coef2<NS>(&s_B[0][w * NS], Bq[0]);
        if (MODE != 1) lds_coef2<NS>(&s_C[0][w * NS], Cq[0]);
#pragma unroll
        for (int t = 0; t < TC; ++t) {
            if (t + 1 < TC) {
                lds_coef2<NS>(&s_B[t + 1][w * NS], Bq[(t + 1) & 1]);
                if (MODE != 1) lds_coef2<NS>(&s_C[t + 1][w * NS], Cq[(t + 1) & 1]);
            }
            const f2v Pt = Pr[t];
            f2 yacc = {0.f, 0.f};
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const f2v arg = pk_mul_lo(Pt, A2p[k]);
                f2 dA;
                dA.x = fast_exp2(arg.x);
                dA.y = fast_exp2(arg.y);
                hp[k] = __builtin_elementwise_fma(dA, hp[k], (f2)pk_mul_hi(Pt, Bq[t & 1][k]));
                if (MODE != 1) yacc = (k == 0) ? Cq[t & 1][k] * hp[k] : __builtin_elementwise_fma(Cq[t & 1][k], hp[k], yacc);
            }
            if (MODE == 1) {
                sdl += Pt.x;
            } else {
                s_y[w][t][lane] = yacc.x + yacc.y;
                if ((t + 1) % CKS == 0) {                                    // c0 is a multiple of TC, TC of CKS
                    const int tabs = c0 + t + 1;                             // the state after `tabs` steps = checkpoint tabs / CKS - 1
                    if (ck_row != nullptr && tabs < p.L && tabs <= t_end) store_ckpt<NS>(ck_row, hp);
                    if (ck_row != nullptr) ck_row += (int64_t)N * p.Di;
                }
            }
        }
        // the next chunk's tiles have been in flight for the whole scan: retire them here, ahead of the output stores
        retire_prefetch();
        release_store_data();
        STAMP(st_scan, st_last);
        __syncthreads();
        STAMP(st_b2, st_last);

        // ---- output tile
        if (MODE == 1) continue;
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int r = tr0 + i * (NT / 16);
            const int t = c0 + r;
            if (t < t_end && c_ok) {
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
                    y.x *= silu_nb(zv.x); y.y *= silu_nb(zv.y); y.z *= silu_nb(zv.z); y.w *= silu_nb(zv.w);
                }
                y_keep[i] = y;
                omax = amax4(omax, y);
                st4(p.out + (tok0 + t) * p.ld_out + d0 + tc4, y_keep[i]);
            }
        }
        STAMP(st_out, st_last);
    }
#ifdef SSCAN_STAMP
    if (p.stamps && lane == 0) {
        unsigned long long* o = p.stamps + ((size_t)blockIdx.x * NW + w) * 8;
        o[0] = st_stage; o[1] = st_b1; o[2] = st_scan; o[3] = st_b2; o[4] = st_out;
        o[5] = st_last - st_t0; o[6] = __builtin_amdgcn_s_memrealtime() - st_r0; o[7] = st_r0;
    }
#endif
    if (MODE == 1) {                                 // local end state and delta sum of this segment
        if (d_ok) {
            float* ho = p.h_carry + (((int64_t)b * p.nseg + seg) * N + w * NS) * p.Di + d;
#pragma unroll
            for (int j = 0; j < NS; ++j) ho[(int64_t)j * p.Di] = (j & 1) ? hp[j / 2].y : hp[j / 2].x;
            if (w == 0) p.sdl[((int64_t)b * p.nseg + seg) * p.Di + d] = sdl;
        }
        return;
    }
    if (p.last_state != nullptr && d_ok && t_end == p.L) {
#pragma unroll
        for (int j = 0; j < NS; ++j)
            p.last_state[((int64_t)b * p.Di + d) * N + w * NS + j] = (j & 1) ? hp[j / 2].y : hp[j / 2].x;
    }
    amax_publish_wave(omax, p.amax_out);
}

// h_carry[b][s] (local end states on entry) -> states ENTERING segment s; one thread per (row, state, channel)
__global__ void sscan_carry_kernel(float* __restrict__ h_carry, const float* __restrict__ sdl, const float* __restrict__ A, int a_log, int B, int nseg, int N, int Di) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * N * Di) return;
    const int d = (int)(i % Di), n = (int)((i / Di) % N), b = (int)(i / ((int64_t)Di * N));
    const float a2 = fminf(load_A(A, (int64_t)d * N + n, a_log) * RESEL_LOG2E, -1e-30f);
    float h = 0.f;
    for (int s = 0; s < nseg; ++s) {
        float* q = h_carry + (((int64_t)b * nseg + s) * N + n) * Di + d;
        const float loc = *q;
        *q = h;
        h = fast_exp2(a2 * sdl[((int64_t)b * nseg + s) * Di + d]) * h + loc;
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
    int B, L, Di, N, nck, softplus, nd, bc_vec;
    int a_log;                      // A holds A_log (see FwdParams)
    int seg_len, nseg;              // time-parallel form: steps per segment (multiple of 32), segments per row (1 = whole row)
    float *dh_carry, *sdl;          // [B][nseg][N][Di]: dL/dh flowing INTO the end of each segment; [B][nseg][Di] delta sums
    AmaxOut amax_dz, amax_ddelta;   // optional: publish max |dz|, max |ddelta| (operands of the in_proj / dt_proj gradient GEMMs)
};

// Cross-lane sums for the dB / dC channel reductions.
// butterfly_sum<V>(v, lane): sums v[0..V) over the 64 lanes with a multi-value butterfly whose stages go from the widest
// pairing down: each halving stage pairs every lane with one lane of the other half of its 64 / 32 / 16 / 8-lane group,
// the "low" lane keeps the lower half of the values and receives the partner's lower half, the "high" lane the upper
// half.  The two widest stages are gfx950's v_permlane32_swap / v_permlane16_swap (swap + add = 2 instructions for
// TWO outputs, no select), the 8- and 4-lane stages a select pair plus a DPP add (row_ror:8, row_half_mirror: any
// involution across the group boundary will do for a sum).  After the halving stages a single value is left and is
// summed over the remaining lane bits with DPP adds.  On return every lane holds the wave total of value index
// value_of_lane<V>(lane); the lanes with (lane & (64 / V - 1)) == 0 cover every index exactly once.  V in {4, 8, 16}.
__device__ __forceinline__ void permlane32_swap(float& a, float& c) {   // a[32..63] <-> c[0..31]
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(c));
}
__device__ __forceinline__ void permlane16_swap(float& a, float& c) {   // odd 16-lane rows of a <-> even rows of c
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(c));
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
constexpr int DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_ROR8 = 0x128;

template <int V, int K = 32>
__device__ __forceinline__ float butterfly_sum(float (&v)[V > 0 ? V : 1], int lane) {
    if constexpr (V > 1) {
        constexpr int H = V / 2;
        float nv[H];
        if constexpr (K == 32 || K == 16) {
#pragma unroll
            for (int i = 0; i < H; ++i) {
                float a = v[i], c = v[i + H];
                if constexpr (K == 32) permlane32_swap(a, c); else permlane16_swap(a, c);
                nv[i] = a + c;
            }
        } else {
            static_assert(K == 8 || K == 4, "halving stages: 32, 16, 8, 4");
            const bool hi = lane & K;
#pragma unroll
            for (int i = 0; i < H; ++i) {
                const float send = hi ? v[i] : v[i + H];
                const float keep = hi ? v[i + H] : v[i];
                nv[i] = keep + (K == 8 ? dpp_mov<DPP_ROW_ROR8>(send) : dpp_mov<DPP_ROW_HALF_MIRROR>(send));
            }
        }
        return butterfly_sum<H, K / 2>(nv, lane);
    } else {
        float r = v[0];
        if constexpr (K >= 8) r += dpp_mov<DPP_ROW_ROR8>(r);
        if constexpr (K >= 4) r += dpp_mov<DPP_ROW_HALF_MIRROR>(r);
        r += dpp_mov<DPP_QUAD_XOR2>(r);
        r += dpp_mov<DPP_QUAD_XOR1>(r);
        return r;
    }
}
template <int V>
__device__ __forceinline__ int value_of_lane(int lane) {
    int idx = 0, h = V / 2;
#pragma unroll
    for (int bit = 32; h >= 1; bit >>= 1) {
        if (lane & bit) idx += h;
        h >>= 1;
    }
    return idx;
}

// Asynchronous global -> LDS copies (gfx950 global_load_lds_dwordx4 / _dword): no VGPR destination.  The LDS address is
// the wave-uniform `lds_wave_base` (in M0) + lane * size, so the LDS image of a wave-instruction is lane-linear.
// Issued as inline assembly on purpose: for the builtin form hipcc guards EVERY later LDS read that may alias the
// destination with s_waitcnt vmcnt(0), which in the sweep also waits for the global store of the previous step (a store
// round trip per step).  Written this way the copies are invisible to the compiler's counters; the kernel retires
// them itself with barrier_vm<N>() below.
__device__ __forceinline__ unsigned lds_offset(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ void glds16(const float* g, float* lds_wave_base) {
    const unsigned l = __builtin_amdgcn_readfirstlane(lds_offset(lds_wave_base));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(l) : "memory");
}
__device__ __forceinline__ void glds4(const float* g, float* lds_wave_base) {
    const unsigned l = __builtin_amdgcn_readfirstlane(lds_offset(lds_wave_base));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(l) : "memory");
}

// Workgroup barriers of the backward kernel.  While an LDS-DMA may be in flight hipcc turns every __syncthreads() into
// s_waitcnt vmcnt(0) + s_barrier, which also drains the GLOBAL STORES a wave has just issued (du / ddelta / dz rows of an
// epilogue, the dB / dC partials of a sweep) - a full store round trip per barrier, ~20 % of the kernel.  The kernel
// therefore keeps its own books: barrier_lds() orders LDS traffic only, barrier_vm<N>() additionally waits until at most
// N vector-memory operations are outstanding - the N stores issued AFTER the DMA being waited for (in-order counter).
__device__ __forceinline__ void barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int NPEND>
__device__ __forceinline__ void barrier_vm() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NPEND) : "memory");
}

// One workgroup = (row b, 64-channel tile); NW waves x NS states; time in 16-step sub-chunks, last to first.
// Per sub-chunk: the raw operand tiles (u, delta, z, dout, B, C rows and the checkpointed state) arrive in LDS by
// LDS-DMA, issued one sub-chunk ahead into the other half of a double buffer - the kernel is bound by VALU issue and
// needs two waves per SIMD, i.e. <= 256 registers per lane, and the 16-step (h, dA) history of a lane already takes
// 128 of them: a register-staged prefetch (40 more) pushed it to 302 and one wave per SIMD at half the issue rate.
// Then: transform pass on the tile (softplus, gated dout) | forward replay from the checkpoint and reverse sweep, in
// two 8-step halves | epilogue of each half on the tile mapping.
template <int NS, int NW>
__global__ __launch_bounds__(NW * 64) void sscan_bwd_kernel(BwdParams p) {
    constexpr int NT = NW * 64;
    constexpr int N = NS * NW;
    constexpr int NP = (NS + 1) / 2;
    constexpr int BC_ITEMS = SC * N / 4;                       // float4 items of a [SC][N] coefficient tile
    constexpr int PER_BC = (BC_ITEMS + NT - 1) / NT;
    constexpr int PER_BC1 = (SC * N + NT - 1) / NT;            // dword items (unaligned B / C rows)
    // [buffer][u | delta -> softplus(delta + bias) | z -> dout * silu'(z) | dout -> dout * silu(z)] (transformed in place)
    __shared__ __attribute__((aligned(16))) float s_raw[2][4][SC][TILE_C];
    __shared__ __attribute__((aligned(16))) float s_acc[2][SCH][TILE_C];     // running dD / dbias sums of the tile threads
    __shared__ __attribute__((aligned(16))) float s_B[2][SC][N];
    __shared__ __attribute__((aligned(16))) float s_C[2][SC][N];
    __shared__ __attribute__((aligned(16))) float s_h0[N][TILE_C];           // checkpoint in front of the staged sub-chunk
    __shared__ __attribute__((aligned(16))) float s_part[3][NW][SCH][TILE_C];
    __shared__ float s_st[2][SC];
    __shared__ __attribute__((aligned(16))) float s_cvec[2][TILE_C];        // delta_bias | D of this channel tile

    int b, dt;
    if (!decode_block(blockIdx.x, p.nd, p.B, b, dt)) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int d0 = dt * TILE_C;
    const int d = d0 + lane;
    const bool d_ok = d < p.Di;
    const int64_t tok0 = (int64_t)b * p.L;
    const bool tile_thr = tid < SC * 16;            // threads that own one float4 of the [SC][64] tile
    const int tc4 = (tid & 15) * 4;
    const int tr = (tid >> 4) & (SC - 1);
    const bool c_ok = tile_thr && (d0 + tc4) < p.Di;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // LDS-DMA never writes the slots of channels >= Di (those lanes are masked off) nor the z tile without a gate:
    // clear them once, so that the idle lanes of the cross-channel reductions carry zeros
    for (int i = tid; i < 2 * 4 * SC * TILE_C / 4; i += NT) st4(&s_raw[0][0][0][0] + i * 4, zero4);
    for (int i = tid; i < N * TILE_C / 4; i += NT) st4(&s_h0[0][0] + i * 4, zero4);
    for (int i = tid; i < 2 * SCH * TILE_C / 4; i += NT) st4(&s_acc[0][0][0] + i * 4, zero4);
    if (tid < 2 * SC) (&s_st[0][0])[tid] = 0.f;
    if (tid < 2 * TILE_C) {
        const float* src = (tid < TILE_C) ? p.delta_bias : p.D;
        const int c = tid & (TILE_C - 1);
        (&s_cvec[0][0])[tid] = (src != nullptr && d0 + c < p.Di) ? src[d0 + c] : 0.f;
    }

    float dzmax = 0.f, ddmax = 0.f;                  // max |dz|, max |ddelta| of this thread's stores
    f2 A2p[NP], dh[NP], dAacc[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        float a[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int j = 2 * k + e;
            a[e] = (d_ok && j < NS) ? load_A(p.A, (int64_t)d * N + w * NS + j, p.a_log) : -1.f;
        }
        A2p[k] = f2{fminf(a[0] * RESEL_LOG2E, -1e-30f), fminf(a[1] * RESEL_LOG2E, -1e-30f)};   // A * log2(e)
        dh[k] = f2{0.f, 0.f};
        dAacc[k] = f2{0.f, 0.f};
    }
    // time-parallel form: this workgroup owns sub-chunks [sc_begin, sc_end) of the row and starts from the adjoint state that
    // the later segments send back (sscan_bwd_local_kernel + sscan_carry_rev_kernel)
    const int seg = p.nseg > 1 ? (int)blockIdx.y : 0;
    const int prow = b * p.nseg + seg;              // row of the per-(row, segment) partial slabs
    if (p.nseg > 1 && d_ok) {
        const float* q = p.dh_carry + ((int64_t)prow * N + w * NS) * p.Di + d;
#pragma unroll
        for (int k = 0; k < NP; ++k)
            dh[k] = f2{2 * k < NS ? q[(int64_t)(2 * k) * p.Di] : 0.f, 2 * k + 1 < NS ? q[(int64_t)(2 * k + 1) * p.Di] : 0.f};
    }

    const int nsc_row = (p.L + SC - 1) / SC;
    const int sc_begin = p.nseg > 1 ? seg * (p.seg_len / SC) : 0;
    const int nsc = p.nseg > 1 ? min(nsc_row, sc_begin + p.seg_len / SC) : nsc_row;      // one past this segment's last sub-chunk
    // issue the LDS-DMA of sub-chunk sc into buffer sc & 1 (plus its start flags into a register)
    auto stage_issue = [&](int sc) {
        const int ts = sc * SC, buf = sc & 1;
        if (c_ok && ts + tr < p.L) {
            const int64_t tok = tok0 + ts + tr;
            float* base = &s_raw[buf][0][0][0] + w * 256;                    // this wave's 4 rows of the [SC][64] tile
            glds16(p.u + tok * p.ld_u + d0 + tc4, base);
            glds16(p.delta + tok * p.ld_delta + d0 + tc4, base + SC * TILE_C);
            if (p.z) glds16(p.z + tok * p.ld_z + d0 + tc4, base + 2 * SC * TILE_C);
            glds16(p.dout + tok * p.ld_dout + d0 + tc4, base + 3 * SC * TILE_C);
        }
        if (p.bc_vec) {
#pragma unroll
            for (int i = 0; i < PER_BC; ++i) {
                const int it = tid + i * NT;
                const int r = it / (N / 4), c = (it % (N / 4)) * 4;
                if (it < BC_ITEMS && ts + r < p.L) {
                    glds16(p.Bm + (tok0 + ts + r) * p.ld_b + c, &s_B[buf][0][0] + (w * 64 + i * NT) * 4);
                    glds16(p.Cm + (tok0 + ts + r) * p.ld_c + c, &s_C[buf][0][0] + (w * 64 + i * NT) * 4);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < PER_BC1; ++i) {
                const int it = tid + i * NT;
                const int r = it / N, c = it % N;
                if (it < SC * N && ts + r < p.L) {
                    glds4(p.Bm + (tok0 + ts + r) * p.ld_b + c, &s_B[buf][0][0] + w * 64 + i * NT);
                    glds4(p.Cm + (tok0 + ts + r) * p.ld_c + c, &s_C[buf][0][0] + w * 64 + i * NT);
                }
            }
        }
        if (p.start && tid < SC && ts + tid < p.L) glds4(p.start + tok0 + ts + tid, &s_st[buf][0]);   // wave 0, lanes < SC
    };
    // Checkpoint c holds the state after (c + 1) * CKS steps.  A sub-chunk (SC = 2 CKS steps) is swept in two halves, each starting
    // from a stored state: the first half from checkpoint 2 sc - 1 (sc = 0: the zero state), the second half from checkpoint 2 sc.
    // Each wave fetches (LDS-DMA) and later reads only its own NS rows of s_h0, so the one buffer is refilled as soon as the wave has
    // its rows in registers: the second half's state arrives during the previous sub-chunk's first half, the first half's state
    // during this sub-chunk's second half.
    float* const s_h0w = &s_h0[0][0] + w * NS * TILE_C;                       // this wave's NS * 64 floats of s_h0
    auto issue_ck = [&](int c) {
        if (c >= 0 && d_ok) {
            const float* src = p.ckpt + ((((int64_t)b * p.nck + c) * NW + w) * p.Di + d) * NS;   // this lane's NS contiguous states
            if constexpr (NS % 4 == 0) {
#pragma unroll
                for (int i = 0; i < NS / 4; ++i) glds16(src + 4 * i, s_h0w + i * 4 * TILE_C);      // LDS image [piece][lane][4]
            } else {
#pragma unroll
                for (int j = 0; j < NS; ++j) glds4(src + j, s_h0w + j * TILE_C);                   // LDS image [state][lane]
            }
        }
    };
    auto read_ck = [&](bool valid, f2 (&h)[NP]) {
        if constexpr (NS % 4 == 0) {
#pragma unroll
            for (int i = 0; i < NS / 4; ++i) {
                const float4 v = valid ? ld4(s_h0w + (i * TILE_C + lane) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                h[2 * i] = f2{v.x, v.y};
                h[2 * i + 1] = f2{v.z, v.w};
            }
        } else {
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                float a[2] = {0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int j = 2 * k + e;
                    if (valid && j < NS) a[e] = s_h0w[j * TILE_C + lane];
                }
                h[k] = f2{a[0], a[1]};
            }
        }
    };
    __syncthreads();                                   // the clears above precede the first DMA
    stage_issue(nsc - 1);
    // the state the LAST sub-chunk needs first: its second half's (checkpoint 2 sc) - or, on a short tail without a second half, its own start
    issue_ck(min(SC, p.L - (nsc - 1) * SC) > SCH ? 2 * (nsc - 1) : 2 * (nsc - 1) - 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // later iterations find their DMA retired by barrier_vm (C) / (E)

    for (int sc = nsc - 1; sc >= sc_begin; --sc) {
        const int ts = sc * SC, buf = sc & 1;
        const int sl = min(SC, p.L - ts);                 // steps in this sub-chunk
        float (*s_u)[TILE_C] = s_raw[buf][0];
        float (*s_dl)[TILE_C] = s_raw[buf][1];
        float (*s_gz)[TILE_C] = s_raw[buf][2];
        float (*s_dy)[TILE_C] = s_raw[buf][3];
        float (*sB)[N] = s_B[buf];
        float (*sC)[N] = s_C[buf];
        const float* sst = s_st[buf];
        // ---------------- (A) DMA of this sub-chunk has landed; transform pass ----------------
        barrier_lds();
        if (c_ok && tr < sl) {
            const float4 bv = ld4(&s_cvec[0][tc4]);
            float4 dl4 = ld4(&s_dl[tr][tc4]);
            dl4.x += bv.x; dl4.y += bv.y; dl4.z += bv.z; dl4.w += bv.w;
            if (p.softplus == 1) {
                dl4.x = softplus_nb(dl4.x); dl4.y = softplus_nb(dl4.y); dl4.z = softplus_nb(dl4.z); dl4.w = softplus_nb(dl4.w);
            }
            st4(&s_dl[tr][tc4], dl4);
            if (p.z) {
                const float4 do4 = ld4(&s_dy[tr][tc4]), pz = ld4(&s_gz[tr][tc4]);
                st4(&s_dy[tr][tc4], make_float4(do4.x * siluf_(pz.x), do4.y * siluf_(pz.y), do4.z * siluf_(pz.z), do4.w * siluf_(pz.w)));
                st4(&s_gz[tr][tc4], make_float4(do4.x * dsiluf_(pz.x), do4.y * dsiluf_(pz.y), do4.z * dsiluf_(pz.z), do4.w * dsiluf_(pz.w)));
            }
        }
        barrier_lds();                                     // (B)
        if (sc > sc_begin) stage_issue(sc - 1);            // in flight during the whole replay / reverse phase

        // ---------------- replay + reverse, in two halves of SCH steps (later half first) ----------------
        // The (h, dA) history of SCH = 8 steps x NS states lives in registers.  Each half starts from a STORED state (the forward
        // leaves one every CKS = 8 steps; until round 4 it left one every 16 and the second half's start was obtained by
        // replaying the first half without history - 8 exp + 12 packed instructions per step and wave, executed twice).
        // Operand sets are fetched from LDS one step ahead of their use (the LDS round trip would otherwise sit at the
        // head of every step of a wave that has only one partner on its SIMD): the replay keeps two sets in flight, the
        // reverse sweep refills its single set between a step's arithmetic and its channel reduction, which no longer
        // needs B / C / delta - so the prefetch costs no extra registers at the sweep's pressure peak.
        struct RepOps { f2 Bq[NP]; float sf, dl, u; };
        auto fetch_rep = [&](int i, RepOps& o) {
            lds_coef2<NS>(&sB[i][w * NS], o.Bq);
            o.sf = sst[i];
            o.dl = s_dl[i][lane];
            o.u = s_u[i][lane];
        };
        auto replay_step = [&](const RepOps& o, f2 (&h)[NP], f2 (&dAo)[NP]) {
            const float du = o.dl * o.u;
            const float dle = (o.sf != 0.f) ? __builtin_inff() : o.dl;
            const f2 dle2 = {dle, dle}, du2 = {du, du};
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const f2 arg = dle2 * A2p[k];
                dAo[k].x = fast_exp2(arg.x);
                dAo[k].y = fast_exp2(arg.y);
                h[k] = __builtin_elementwise_fma(dAo[k], h[k], du2 * o.Bq[k]);
            }
        };
        struct RevOps { f2 Bq[NP], Cq[NP]; float dl, dy, u; };
        auto fetch_rev = [&](int r, RevOps& o) {
            lds_coef2<NS>(&sB[r][w * NS], o.Bq);
            lds_coef2<NS>(&sC[r][w * NS], o.Cq);
            o.dl = s_dl[r][lane];
            o.dy = s_dy[r][lane];
            o.u = s_u[r][lane];
        };
        auto half = [&](int base, int n, const f2 (&hstart)[NP]) {
            f2 hist_h[SCH][NP], hist_a[SCH][NP];
            {
                f2 h[NP];
#pragma unroll
                for (int k = 0; k < NP; ++k) h[k] = hstart[k];
                RepOps o0, o1;
                fetch_rep(base, o0);
#pragma unroll
                for (int i = 0; i < SCH; i += 2) {
                    if (i < n) {
                        f2 dA[NP];
                        fetch_rep(min(base + i + 1, SC - 1), o1);
                        __builtin_amdgcn_sched_barrier(0);
                        replay_step(o0, h, dA);
#pragma unroll
                        for (int k = 0; k < NP; ++k) { hist_h[i][k] = h[k]; hist_a[i][k] = dA[k]; }
                    }
                    if (i + 1 < n) {
                        f2 dA[NP];
                        fetch_rep(min(base + i + 2, SC - 1), o0);
                        __builtin_amdgcn_sched_barrier(0);
                        replay_step(o1, h, dA);
#pragma unroll
                        for (int k = 0; k < NP; ++k) { hist_h[i + 1][k] = h[k]; hist_a[i + 1][k] = dA[k]; }
                    }
                }
            }
            RevOps o;
            fetch_rev(base + n - 1, o);
#pragma unroll
            for (int i = SCH - 1; i >= 0; --i) {
                if (i < n) {
                    const int r = base + i;
                    const float du = o.dl * o.u;
                    const f2 dl2 = {o.dl, o.dl}, du2 = {du, du}, dy2 = {o.dy, o.dy};
                    f2 P1 = {0.f, 0.f}, P2 = P1, P3 = P1;
                    float red[4 * NP];                     // [dB pairs | dC pairs] for the channel reduction
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        const f2 hk = hist_h[i][k];
                        const f2 hp = (i == 0) ? hstart[k] : hist_h[i > 0 ? i - 1 : 0][k];
                        const f2 ak = hist_a[i][k];                                   // 0 at a reset step
                        const f2 dhk = __builtin_elementwise_fma(dy2, o.Cq[k], dh[k]);  // dL/dh_t
                        P3 = __builtin_elementwise_fma(o.Cq[k], hk, P3);
                        dh[k] = dhk * ak;                                             // carried to step t-1
                        const f2 tmp = dh[k] * hp;                                    // dL/d(dA) * dA
                        P2 = __builtin_elementwise_fma(tmp, A2p[k], P2);              // in units of log2(e): rescaled in the epilogue
                        dAacc[k] = __builtin_elementwise_fma(tmp, dl2, dAacc[k]);
                        const f2 gb = dhk * du2, gc = dy2 * hk;
                        red[2 * k] = gb.x; red[2 * k + 1] = gb.y;
                        red[2 * NP + 2 * k] = gc.x; red[2 * NP + 2 * k + 1] = gc.y;
                        P1 = __builtin_elementwise_fma(dhk, o.Bq[k], P1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (i > 0) fetch_rev(base + i - 1, o);                            // lands during the reduction below
                    s_part[0][w][i][lane] = P1.x + P1.y;
                    s_part[1][w][i][lane] = P2.x + P2.y;
                    s_part[2][w][i][lane] = P3.x + P3.y;
                    const float tot = butterfly_sum<4 * NP>(red, lane);
                    if ((lane & (64 / (4 * NP) - 1)) == 0) {
                        const int vi = value_of_lane<4 * NP>(lane);                   // < 2NP: dB state vi ; else dC state vi - 2NP
                        const int st = vi < 2 * NP ? vi : vi - 2 * NP;
                        if (st < NS) {
                            const int64_t o2 = ((int64_t)dt * p.B * p.L + tok0 + ts + r) * N + w * NS + st;
                            (vi < 2 * NP ? p.dB_part : p.dC_part)[o2] = tot;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        // epilogue of rows [base, base + n): one float2 (2 channels of one row) per thread, so that all 256 tile threads
        // take part in each half's epilogue (operands come back from LDS)
        auto epilogue = [&](int base, int n) {
            const int hr = (tid >> 5) & (SCH - 1), c2 = (tid & 31) * 2;
            const int row = base + hr;
            if (tile_thr && hr < n && d0 + c2 < p.Di) {
                f2 P1 = {0.f, 0.f}, P2 = P1, P3 = P1;
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) {
                    P1 += *reinterpret_cast<const f2*>(&s_part[0][ww][hr][c2]);
                    P2 += *reinterpret_cast<const f2*>(&s_part[1][ww][hr][c2]);
                    P3 += *reinterpret_cast<const f2*>(&s_part[2][ww][hr][c2]);
                }
                const f2 dl2 = *reinterpret_cast<const f2*>(&s_dl[row][c2]), u2 = *reinterpret_cast<const f2*>(&s_u[row][c2]);
                const f2 dy2 = *reinterpret_cast<const f2*>(&s_dy[row][c2]);
                const f2 Dv = *reinterpret_cast<const f2*>(&s_cvec[1][c2]);
                const int64_t tok = tok0 + ts + row;
                // du = delta' * sum_n dh B + D * dy
                *reinterpret_cast<f2*>(p.du + tok * p.ld_du + d0 + c2) = dl2 * P1 + Dv * dy2;
                // d delta' = sum_n (dh h_prev dA) A + u * sum_n dh B
                f2 g = RESEL_LN2 * P2 + u2 * P1;
                if (p.softplus) {                           // softplus'(x) = sigmoid(x) = 1 - exp(-softplus(x))
                    g.x *= 1.f - fast_exp(-dl2.x);
                    g.y *= 1.f - fast_exp(-dl2.y);
                }
                *reinterpret_cast<f2*>(p.ddelta + tok * p.ld_ddelta + d0 + c2) = g;
                ddmax = fmaxf(ddmax, fmaxf(__builtin_fabsf(g.x), __builtin_fabsf(g.y)));
                f2* aD = reinterpret_cast<f2*>(&s_acc[0][hr][c2]);      // this thread's own slots
                f2* ab = reinterpret_cast<f2*>(&s_acc[1][hr][c2]);
                *aD += dy2 * u2;
                *ab += g;
                if (p.z) {                                  // dz = dout * silu'(z) * (pre-gate output y = sum_n C h + D u)
                    const f2 gc = *reinterpret_cast<const f2*>(&s_gz[row][c2]);
                    const f2 dzv = gc * (P3 + Dv * u2);
                    *reinterpret_cast<f2*>(p.dz + tok * p.ld_dz + d0 + c2) = dzv;
                    dzmax = fmaxf(dzmax, fmaxf(__builtin_fabsf(dzv.x), __builtin_fabsf(dzv.y)));
                }
            }
        };
        if (sl > SCH) {
            f2 hm[NP];
            read_ck(true, hm);                             // checkpoint 2 sc: the state after the first half's steps
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // rows in registers: the slots take the first half's start state
            issue_ck(2 * sc - 1);
            half(SCH, sl - SCH, hm);
            // (C) the tile DMA of the next sub-chunk was issued before this half's sl - SCH partial-slab stores
            if (sl == SC) barrier_vm<SCH>(); else barrier_vm<0>();
            epilogue(SCH, sl - SCH);
            barrier_lds();                                 // (D) s_part is rewritten by the first half
        }
        {
            f2 h0[NP];
            read_ck(sc > 0, h0);                           // checkpoint 2 sc - 1 (retired by (C); a short tail fetched it in the prologue)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the checkpoint rows are in registers: their LDS slots may be refilled
            if (sc > sc_begin) issue_ck(2 * (sc - 1));     // the next sub-chunk (always a whole one) starts with its second half
            half(0, min(sl, SCH), h0);
        }
        // (E) retires the checkpoint DMA (and, on a short tail sub-chunk that skipped (C), the tile DMA)
        if (sl >= SCH) barrier_vm<SCH>(); else barrier_vm<0>();
        epilogue(0, min(sl, SCH));
    }
    // ---- per-(b) partials of the parameter gradients
    if (d_ok) {
#pragma unroll
        for (int j = 0; j < NS; ++j)
            p.dA_part[((int64_t)prow * p.Di + d) * N + w * NS + j] = (j & 1) ? dAacc[j / 2].y : dAacc[j / 2].x;
    }
    __syncthreads();
    if (tid < 2 * TILE_C) {
        const int which = tid >> 6, c = tid & 63;
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < SCH; ++r) acc += s_acc[which][r][c];
        if (d0 + c < p.Di) (which ? p.dbias_part : p.dD_part)[(int64_t)prow * p.Di + d0 + c] = acc;
    }
    amax_publish_wave(dzmax, p.amax_dz);
    amax_publish_wave(ddmax, p.amax_ddelta);
}

// ---- time-parallel backward, local pass: the adjoint recurrence alone (dL/dh_t = dy_t C_t + carried; carried = dL/dh_t * dA_t),
// run over one time segment from a zero adjoint.  No state replay, no gradients: it leaves, per segment, the adjoint at the
// segment's start (dh_loc) and the sum of the segment's deltas, from which sscan_carry_rev_kernel forms the adjoint entering
// the END of every segment (the map across a segment is dh_start = exp2(A2 * S) * dh_end + dh_loc).  Same staging as the
// forward: TC-step chunks, resets as delta = +inf (the decay 0 cuts the adjoint exactly where the forward cut the state).
template <int NS, int NW, int TC>
__global__ __launch_bounds__(NW * 64) void sscan_bwd_local_kernel(BwdParams p) {
    constexpr int NT = NW * 64;
    constexpr int N = NS * NW;
    constexpr int NP = (NS + 1) / 2;
    constexpr int PER_T = (TC * 16 + NT - 1) / NT;
    constexpr int BC_ITEMS = TC * N / 4;
    constexpr int PER_BC = (BC_ITEMS + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) float s_dl[TC][TILE_C];
    __shared__ __attribute__((aligned(16))) float s_dy[TC][TILE_C];
    __shared__ __attribute__((aligned(16))) float s_C[TC][N];
    int b, dt;
    if (!decode_block(blockIdx.x, p.nd, p.B, b, dt)) return;
    const int seg = blockIdx.y;
    const int t_begin = seg * p.seg_len, t_end = min(p.L, t_begin + p.seg_len);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int d0 = dt * TILE_C, d = d0 + lane;
    const bool d_ok = d < p.Di;
    const int64_t tok0 = (int64_t)b * p.L;
    f2 A2p[NP], dh[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        float a[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int j = 2 * k + e;
            a[e] = (d_ok && j < NS) ? fminf(load_A(p.A, (int64_t)d * N + w * NS + j, p.a_log) * RESEL_LOG2E, -1e-30f) : -1.f;
        }
        A2p[k] = f2{a[0], a[1]};
        dh[k] = f2{0.f, 0.f};
    }
    float sdl = 0.f;
    const int tc4 = (tid & 15) * 4, tr0 = tid >> 4;
    const bool c_ok = (d0 + tc4) < p.Di;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c_ok && p.delta_bias) bv = ld4(p.delta_bias + d0 + tc4);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int nchunk = (t_end - t_begin + TC - 1) / TC;
    for (int ci = nchunk - 1; ci >= 0; --ci) {
        const int c0 = t_begin + ci * TC;
        __syncthreads();                             // the previous chunk's tiles are consumed
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int r = tr0 + i * (NT / 16);
            const int t = c0 + r;
            float4 dv = zero4, dy = zero4;
            if (r < TC && t < t_end && c_ok) {
                const int64_t tok = tok0 + t;
                dv = ld4(p.delta + tok * p.ld_delta + d0 + tc4);
                dy = ld4(p.dout + tok * p.ld_dout + d0 + tc4);
                dv.x += bv.x; dv.y += bv.y; dv.z += bv.z; dv.w += bv.w;
                if (p.softplus == 1) { dv.x = softplus_nb(dv.x); dv.y = softplus_nb(dv.y); dv.z = softplus_nb(dv.z); dv.w = softplus_nb(dv.w); }
                if (p.z) {
                    const float4 zv = ld4(p.z + tok * p.ld_z + d0 + tc4);
                    dy.x *= silu_nb(zv.x); dy.y *= silu_nb(zv.y); dy.z *= silu_nb(zv.z); dy.w *= silu_nb(zv.w);
                }
                if (p.start && p.start[tok] != 0.f) { const float inf = __builtin_inff(); dv = make_float4(inf, inf, inf, inf); }
            }
            if (r < TC) { st4(&s_dl[r][tc4], dv); st4(&s_dy[r][tc4], dy); }
        }
#pragma unroll
        for (int i = 0; i < PER_BC; ++i) {
            const int it = tid + i * NT;
            const int t = c0 + it / (N / 4), c = (it % (N / 4)) * 4;
            if (it < BC_ITEMS) st4(&s_C[0][0] + it * 4, t < t_end ? load_bc4(p.Cm, tok0 + t, p.ld_c, c, p.bc_vec) : zero4);
        }
        __syncthreads();
#pragma unroll
        for (int t = TC - 1; t >= 0; --t) {
            f2 Cq[NP];
            lds_coef2<NS>(&s_C[t][w * NS], Cq);
            const float dl = s_dl[t][lane], dy = s_dy[t][lane];
            const f2 dl2 = {dl, dl}, dy2 = {dy, dy};
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const f2 arg = dl2 * A2p[k];
                f2 dA;
                dA.x = fast_exp2(arg.x);
                dA.y = fast_exp2(arg.y);
                dh[k] = __builtin_elementwise_fma(dy2, Cq[k], dh[k]) * dA;
            }
            sdl += dl;
        }
    }
    if (d_ok) {
        float* ho = p.dh_carry + (((int64_t)b * p.nseg + seg) * N + w * NS) * p.Di + d;
#pragma unroll
        for (int j = 0; j < NS; ++j) ho[(int64_t)j * p.Di] = (j & 1) ? dh[j / 2].y : dh[j / 2].x;
        if (w == 0) p.sdl[((int64_t)b * p.nseg + seg) * p.Di + d] = sdl;
    }
}
// dh_carry[b][s] (adjoint at the START of segment s for a zero adjoint at its end) -> adjoint entering the END of segment s
__global__ void sscan_carry_rev_kernel(float* __restrict__ dh_carry, const float* __restrict__ sdl, const float* __restrict__ A, int a_log, int B, int nseg, int N, int Di) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * N * Di) return;
    const int d = (int)(i % Di), n = (int)((i / Di) % N), b = (int)(i / ((int64_t)Di * N));
    const float a2 = fminf(load_A(A, (int64_t)d * N + n, a_log) * RESEL_LOG2E, -1e-30f);
    float carry = 0.f;
    for (int s = nseg - 1; s >= 0; --s) {
        float* q = dh_carry + (((int64_t)b * nseg + s) * N + n) * Di + d;
        const float loc = *q;
        *q = carry;
        carry = fast_exp2(a2 * sdl[((int64_t)b * nseg + s) * Di + d]) * carry + loc;
    }
}

// dB / dC: sum the per-channel-tile slabs; parameter gradients: sum the per-row partials.
__global__ void sscan_reduce_bc_kernel(const float* __restrict__ partB, const float* __restrict__ partC, int nd, int64_t ntok, int N,
                                       float* outB, int64_t ldB, float* outC, int64_t ldC) {
    const float* part = blockIdx.y ? partC : partB;
    float* out = blockIdx.y ? outC : outB;
    const int64_t ld = blockIdx.y ? ldC : ldB;
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
// dA [Di * N] (blockIdx.y = 0; with A_log: dA_log = dA * A, A = -exp(A_log)), dD [Di] (1), ddelta_bias [Di] (2): column sums of the per-row
// partial slabs [K][C] in a fixed order - the body of resel_common.h's colsum_kernel, three outputs per launch
__global__ __launch_bounds__(256) void sscan_param_grads_kernel(const float* __restrict__ pA, const float* __restrict__ pD, const float* __restrict__ pb, int K,
                                                                int na, int Di, const float* __restrict__ A, int a_log, float* __restrict__ dA,
                                                                float* __restrict__ dD, float* __restrict__ dbias) {
    __shared__ float s_acc[16][17];
    const int which = blockIdx.y;
    const float* part = which == 0 ? pA : which == 1 ? pD : pb;
    float* out = which == 0 ? dA : which == 1 ? dD : dbias;
    const int C = which == 0 ? na : Di;
    if (out == nullptr || (int)blockIdx.x * 16 >= C) return;
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float acc = 0.f;
    if (c < C) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int k = rg;
        for (; k + 48 < K; k += 64) {
            a0 += part[(int64_t)k * C + c];
            a1 += part[(int64_t)(k + 16) * C + c];
            a2 += part[(int64_t)(k + 32) * C + c];
            a3 += part[(int64_t)(k + 48) * C + c];
        }
        for (; k < K; k += 16) a0 += part[(int64_t)k * C + c];
        acc = (a0 + a1) + (a2 + a3);
    }
    s_acc[rg][cl] = acc;
    __syncthreads();
    if (rg == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += s_acc[r][cl];
        if (which == 0 && a_log) t *= -expf(A[c]);
        out[c] = t;
    }
}
// Time segments of the forward: one pass while the one-pass grid (B * nd workgroups, two fit a CU) covers ~3/4 of the chip's
// 512 slots, otherwise enough segments to fill them (each at least two chunks long).  `force` > 0 overrides (tests).
inline int fwd_segments(int B, int L, int nd, int TC, int force) {
    const int max_seg = L / (2 * TC) > 0 ? L / (2 * TC) : 1;
    int nseg = force > 0 ? force : (B * nd >= 384 ? 1 : (512 + B * nd - 1) / (B * nd));
    if (nseg > max_seg) nseg = max_seg;
    if (nseg <= 1) return 1;
    const int seg_len = ((L + nseg - 1) / nseg + TC - 1) / TC * TC;
    return (L + seg_len - 1) / seg_len;
}
int g_fwd_big = [] { const char* e = getenv("RESEL_SSCAN_FWD_BIG"); return e ? atoi(e) : 1; }();   // 0: never take the three-workgroup edition (A/B)
template <int NS, int NW, int TC>
int launch_fwd(FwdParams p, int force_seg, void* workspace, hipStream_t s) {
    const int bp = (p.B + 7) / 8 * 8;
    const int nseg = fwd_segments(p.B, p.L, p.nd, TC, force_seg);
    if (nseg <= 1) {
        if (NS == 8 && NW == 4 && TC == 32 && g_fwd_big && (int64_t)bp * p.nd >= 768)      // three workgroups per CU exist: the TC = 16 edition holds them
            launch_maybe_timed(RESEL_PROF_SSCAN_FWD, sscan_fwd2_kernel<NS, NW, (NS == 8 && NW == 4 ? 16 : TC), 0, 3>, dim3(bp * p.nd), dim3(NW * 64), s, p);
        else
            launch_maybe_timed(RESEL_PROF_SSCAN_FWD, sscan_fwd2_kernel<NS, NW, TC, 0>, dim3(bp * p.nd), dim3(NW * 64), s, p);
        return launch_status();
    }
    if (!workspace) return RESEL_EINVAL;
    p.nseg = nseg;
    p.seg_len = ((p.L + nseg - 1) / nseg + TC - 1) / TC * TC;
    p.h_carry = (float*)workspace;
    p.sdl = p.h_carry + (size_t)p.B * nseg * p.N * p.Di;
    launch_maybe_timed(RESEL_PROF_SSCAN_FWD_LOCAL, sscan_fwd2_kernel<NS, NW, TC, 1>, dim3(bp * p.nd, nseg), dim3(NW * 64), s, p);
    const int64_t n = (int64_t)p.B * p.N * p.Di;
    hipLaunchKernelGGL(sscan_carry_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p.h_carry, p.sdl, p.A, p.a_log, p.B, nseg, p.N, p.Di);
    launch_maybe_timed(RESEL_PROF_SSCAN_FWD, sscan_fwd2_kernel<NS, NW, TC, 2>, dim3(bp * p.nd, nseg), dim3(NW * 64), s, p);
    return launch_status();
}
template <int NS, int NW>
int launch_bwd(const BwdParams& p, hipStream_t s) {
    const int bp = (p.B + 7) / 8 * 8;
    if (p.nseg > 1) {                                // time-parallel form: local adjoint pass, carry, then the full pass per segment
        launch_maybe_timed(RESEL_PROF_SSCAN_BWD_LOCAL, sscan_bwd_local_kernel<NS, NW, 32>, dim3(bp * p.nd, p.nseg), dim3(NW * 64), s, p);
        const int64_t n = (int64_t)p.B * p.N * p.Di;
        hipLaunchKernelGGL(sscan_carry_rev_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p.dh_carry, p.sdl, p.A, p.a_log, p.B, p.nseg, p.N, p.Di);
    }
    launch_maybe_timed(RESEL_PROF_SSCAN_BWD, sscan_bwd_kernel<NS, NW>, dim3(bp * p.nd, p.nseg), dim3(NW * 64), s, p);
    return launch_status();
}

inline int n_ckpt(int L) { return (L - 1) / CKS; }   // checkpoints after steps CKS, 2*CKS, ... (< L)
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct BwdWs { size_t dB, dC, dA, dD, dbias, carry, sdl, total; int nseg; };
inline BwdWs bwd_ws(int B, int L, int Di, int N, int force_seg) {
    const size_t nd = (Di + TILE_C - 1) / TILE_C;
    BwdWs w;
    w.nseg = fwd_segments(B, L, (int)nd, 32, force_seg);
    const size_t rows = (size_t)B * w.nseg;          // per-(row, segment) partials of the parameter gradients
    size_t o = 0;
    w.dB = o; o += align256(nd * (size_t)B * L * N * 4);
    w.dC = o; o += align256(nd * (size_t)B * L * N * 4);
    w.dA = o; o += align256(rows * Di * N * 4);
    w.dD = o; o += align256(rows * Di * 4);
    w.dbias = o; o += align256(rows * Di * 4);
    w.carry = o; o += w.nseg > 1 ? align256(rows * N * Di * 4) : 0;
    w.sdl = o; o += w.nseg > 1 ? align256(rows * Di * 4) : 0;
    w.total = o;
    return w;
}

}  // namespace

extern "C" size_t resel_selective_scan_ckpt_bytes(int B, int L, int Di, int N) {
    return (size_t)B * (size_t)(n_ckpt(L) > 0 ? n_ckpt(L) : 0) * (size_t)N * (size_t)Di * sizeof(float);
}

extern "C" size_t resel_selective_scan_fwd_workspace_bytes(int B, int L, int Di, int N, int time_segments) {
    const int nseg = fwd_segments(B, L, (Di + TILE_C - 1) / TILE_C, 32, time_segments);
    return nseg > 1 ? ((size_t)B * nseg * N * Di + (size_t)B * nseg * Di) * sizeof(float) : 0;
}

extern "C" int resel_selective_scan_fwd(const float* u, int64_t ld_u, const float* delta, int64_t ld_delta,
                                        const float* z, int64_t ld_z, const float* A,
                                        const float* Bm, int64_t ld_b, const float* Cm, int64_t ld_c,
                                        const float* D, const float* delta_bias, const float* start,
                                        float* out, int64_t ld_out, float* ckpt, float* last_state, void* workspace,
                                        int B, int L, int Di, int N, int delta_softplus, int time_segments,
                                        void* amax_out, unsigned amax_epoch, resel_stream_t stream) {
    if (!u || !delta || !A || !Bm || !Cm || !out || B <= 0 || L <= 0 || Di <= 0) return RESEL_EINVAL;
    if (amax_out && (reinterpret_cast<uintptr_t>(amax_out) & 7u)) return RESEL_EINVAL;
    if (Di % 4 != 0 || ld_u % 4 || ld_delta % 4 || ld_out % 4 || (z && ld_z % 4)) return RESEL_EINVAL;
    if (!aligned16(u) || !aligned16(delta) || !aligned16(out) || (z && !aligned16(z))) return RESEL_EINVAL;
    if ((D && !aligned16(D)) || (delta_bias && !aligned16(delta_bias))) return RESEL_EINVAL;
    FwdParams p{u, delta, z, A, Bm, Cm, D, delta_bias, start, out, ckpt, last_state,
                ld_u, ld_delta, ld_z, ld_b, ld_c, ld_out, B, L, Di, N, n_ckpt(L), delta_softplus & 3,
                (Di + TILE_C - 1) / TILE_C,
                (ld_b % 4 == 0 && ld_c % 4 == 0 && aligned16(Bm) && aligned16(Cm)) ? 1 : 0, (delta_softplus >> 2) & 1, 0, 1, nullptr, nullptr,
                AmaxOut{(unsigned long long*)amax_out, amax_epoch}};
#ifdef SSCAN_STAMP
    p.stamps = g_stamps;
#endif
    hipStream_t s = (hipStream_t)stream;
    switch (N) {
        case 4: return launch_fwd<1, 4, 32>(p, time_segments, workspace, s);
        case 8: return launch_fwd<2, 4, 32>(p, time_segments, workspace, s);
        case 16: return launch_fwd<4, 4, 32>(p, time_segments, workspace, s);
        case 32: return launch_fwd<8, 4, 32>(p, time_segments, workspace, s);
        case 64: return launch_fwd<8, 8, 32>(p, time_segments, workspace, s);
        default: return RESEL_EINVAL;
    }
}

extern "C" size_t resel_selective_scan_bwd_workspace_bytes(int B, int L, int Di, int N, int time_segments) {
    return bwd_ws(B, L, Di, N, time_segments).total;
}

extern "C" int resel_selective_scan_bwd(const float* u, int64_t ld_u, const float* delta, int64_t ld_delta,
                                        const float* z, int64_t ld_z, const float* A,
                                        const float* Bm, int64_t ld_b, const float* Cm, int64_t ld_c,
                                        const float* D, const float* delta_bias, const float* start,
                                        const float* dout, int64_t ld_dout, const float* ckpt,
                                        float* du, int64_t ld_du, float* ddelta, int64_t ld_ddelta,
                                        float* dz, int64_t ld_dz, float* dBm, int64_t ld_db, float* dCm, int64_t ld_dc,
                                        float* dA, float* dD, float* ddelta_bias, void* workspace,
                                        int B, int L, int Di, int N, int delta_softplus, int time_segments,
                                        void* amax_dz, void* amax_ddelta, unsigned amax_epoch, resel_stream_t stream) {
    if (!u || !delta || !A || !Bm || !Cm || !dout || !du || !ddelta || !dBm || !dCm || !dA || !workspace)
        return RESEL_EINVAL;
    if ((reinterpret_cast<uintptr_t>(amax_dz) & 7u) || (reinterpret_cast<uintptr_t>(amax_ddelta) & 7u)) return RESEL_EINVAL;
    if (B <= 0 || L <= 0 || Di <= 0 || Di % 4 != 0 || N % 4 != 0) return RESEL_EINVAL;
    if ((z != nullptr) != (dz != nullptr)) return RESEL_EINVAL;
    if (n_ckpt(L) > 0 && !ckpt) return RESEL_EINVAL;
    if (ld_u % 4 || ld_delta % 4 || ld_dout % 4 || ld_du % 4 || ld_ddelta % 4 || (z && (ld_z % 4 || ld_dz % 4)))
        return RESEL_EINVAL;
    if (!aligned16(u) || !aligned16(delta) || !aligned16(dout) || !aligned16(du) || !aligned16(ddelta) ||
        (z && (!aligned16(z) || !aligned16(dz))) || (D && !aligned16(D)) || (delta_bias && !aligned16(delta_bias)) ||
        !aligned16(workspace))
        return RESEL_EINVAL;
    const BwdWs ws = bwd_ws(B, L, Di, N, time_segments);
    char* base = (char*)workspace;
    BwdParams p{u, delta, z, A, Bm, Cm, D, delta_bias, start, dout, ckpt, du, ddelta, dz,
                (float*)(base + ws.dB), (float*)(base + ws.dC), (float*)(base + ws.dA), (float*)(base + ws.dD),
                (float*)(base + ws.dbias),
                ld_u, ld_delta, ld_z, ld_b, ld_c, ld_dout, ld_du, ld_ddelta, ld_dz,
                B, L, Di, N, n_ckpt(L), delta_softplus & 3, (Di + TILE_C - 1) / TILE_C,
                (ld_b % 4 == 0 && ld_c % 4 == 0 && aligned16(Bm) && aligned16(Cm)) ? 1 : 0, (delta_softplus >> 2) & 1,
                ws.nseg > 1 ? ((L + ws.nseg - 1) / ws.nseg + 31) / 32 * 32 : L, ws.nseg,
                ws.nseg > 1 ? (float*)(base + ws.carry) : nullptr, ws.nseg > 1 ? (float*)(base + ws.sdl) : nullptr,
                AmaxOut{(unsigned long long*)amax_dz, amax_epoch}, AmaxOut{(unsigned long long*)amax_ddelta, amax_epoch}};
    hipStream_t s = (hipStream_t)stream;
    int rc;
    switch (N) {
        case 4: rc = launch_bwd<1, 4>(p, s); break;
        case 8: rc = launch_bwd<2, 4>(p, s); break;
        case 16: rc = launch_bwd<4, 4>(p, s); break;
        case 32: rc = launch_bwd<8, 4>(p, s); break;     // 4 waves x 8 states: the 8-step (h, dA) history is 128 VGPRs
        case 64: rc = launch_bwd<8, 8>(p, s); break;
        default: return RESEL_EINVAL;
    }
    if (rc != RESEL_OK) return rc;
    const int64_t ntok = (int64_t)B * L;
    const int64_t n4 = ntok * (N / 4);
    // two tail launches (were five): dB | dC slabs in one grid, the three parameter-gradient column sums in another
    hipLaunchKernelGGL(sscan_reduce_bc_kernel, dim3((unsigned)((n4 + 255) / 256), 2), dim3(256), 0, s,
                       p.dB_part, p.dC_part, p.nd, ntok, N, dBm, ld_db, dCm, ld_dc);
    const int64_t na = (int64_t)Di * N;
    hipLaunchKernelGGL(sscan_param_grads_kernel, dim3((unsigned)((na + 15) / 16), 3), dim3(256), 0, s,
                       p.dA_part, p.dD_part, p.dbias_part, B * ws.nseg, (int)na, Di, A, p.a_log, dA, dD, ddelta_bias);
    return launch_status();
}
