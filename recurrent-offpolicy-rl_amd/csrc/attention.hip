// cgpt attention core for gfx950: packed variable-length causal attention with ALiBi, bf16 MFMA, fp32 softmax.
//
// Tokens of all sequences are packed [T, 3, H, hd] (q | k | v per token); cu_seqlens delimits the sequences.
// All three kernels are built on v_mfma_f32_32x32x16_bf16 and keep the SCORE tile transposed so that softmax
// statistics are per LANE (MFMA C/D layout: column = lane & 31, 16 rows in registers, row = (r&3) + 8(r>>2) + 4(lane>>5)):
//   forward / dQ kernel : S^T[key][q] = K Q^T      -> column = query: running max / sum / lse / delta are one value per lane,
//                         and the probability tile feeds the next product (O^T = V^T P^T, dQ^T = K^T dS^T) straight from
//                         registers as the B operand (rows = summed index; k-order 16s + 8(j>>2) + 4h + (j&3)), while the
//                         A operand (V^T / K^T) is read from a transposed LDS tile in that same k-order with two ds_read_b64;
//   dK/dV kernel        : S[q][key] = Q K^T        -> column = key: dV^T = dO^T P and dK^T = Q^T dS sum over the row index q.
// Arithmetic intensity is low for hd = 32 (4 MFMAs per 32x32 tile against ~16 exp2 per lane), so these kernels are
// VALU/exp-bound rather than MFMA-bound; the whole attention share of a cgpt update is ~1 TFLOP.
// Semantics restated from flash-attn's MHA(causal, alibi) - see oracle/kernels.py attention_alibi_varlen_ref: PARITY UNPINNED.
//
// Attention-probability dropout (MHA(dropout=p), reference TransformerFlashAttention.py:67-70): the keep mask is a pure
// function of (seed, offset, head, packed query token, key position), so the forward, the dQ and the dK/dV kernel each
// regenerate it for the elements they hold and nothing is stored:
//     word(q_tok, key >> 2) = mix32((q_tok * 0x9E3779B1) ^ ((key >> 2) * 0x85EBCA77) ^ head_key)      (32 bits -> 4 keys)
//     keep(q_tok, key)      = byte (key & 3) of that word < thr,   thr = floor((1 - p) * 255) + 1      (8-bit threshold and
//     the 1 / (1 - p) rescale of the kept probabilities are flash-attn's; the counter function itself is this build's and
//     is restated in oracle/kernels.py `attn_dropout_keep`).  Softmax statistics (running max, sum, lse, delta) are those
//     of the un-dropped probabilities; only the P V, dV and dP products see the mask.
#include "resel_common.h"

namespace {
using namespace resel;

typedef __bf16 bf16_t;
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float NEG_BIG = -1e30f;
constexpr int KT = 32;          // keys per MFMA score tile
constexpr int KTB = 128;        // keys staged in LDS per barrier pair of the forward / dQ kernel (KTB / KT sub-tiles)
constexpr int PADE = 8;         // bf16 elements of row padding in LDS tiles (16 bytes)

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = (bf16_t)0.f;
    return z;
}
__device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// accumulator registers 8s .. 8s+7 -> bf16 B-operand fragment of k-step s
__device__ __forceinline__ bf16x8 acc_to_frag(const f32x16& x, int s) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const f32x2 v = {x[8 * s + j], x[8 * s + j + 1]};
        const bf16x2 p = __builtin_convertvector(v, bf16x2);
        r[j] = p[0];
        r[j + 1] = p[1];
    }
    return r;
}
// A-operand fragment in the accumulator's k-order from a transposed LDS tile row: elements [16s + 4h, +4) and [16s + 8 + 4h, +4)
__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* row, int s, int hh) {
    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(row + 16 * s + 4 * hh);
    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(row + 16 * s + 8 + 4 * hh);
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) { r[j] = lo[j]; r[4 + j] = hi[j]; }
    return r;
}
__device__ __forceinline__ int acc_row(int reg, int hh) { return (reg & 3) + 8 * (reg >> 2) + 4 * hh; }

__device__ __forceinline__ uint32_t mix32(uint32_t x) {          // 2-multiply avalanche finaliser ("lowbias32")
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
constexpr uint32_t DROP_CQ = 0x9E3779B1u, DROP_CK = 0x85EBCA77u, DROP_CH = 0xC2B2AE3Du;
__device__ __forceinline__ uint32_t drop_head_key(uint64_t seed, uint64_t offset, int h) {
    uint32_t x = mix32((uint32_t)h * DROP_CH ^ (uint32_t)(offset >> 32));
    x = mix32(x ^ (uint32_t)offset);
    x = mix32(x ^ (uint32_t)(seed >> 32));
    return mix32(x ^ (uint32_t)seed);
}
// lane j's value within each quad of lanes (DPP quad_perm broadcast)
template <int J>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, J | (J << 2) | (J << 4) | (J << 6), 0xf, 0xf, true);
}

struct AttnParams {
    const bf16_t* qkv;
    const int32_t* cu;
    const float* slopes;
    bf16_t* out;
    float* lse;                 // [H, T] base-2 log-sum-exp of the scaled + biased scores
    const bf16_t* dout;
    const float* delta;         // [H, T]
    bf16_t* dqkv;
    int T, S, H;
    float scale;
    uint32_t thr;               // dropout: keep iff random byte < thr (256 = keep all)
    float rp;                   // 1 / (1 - p)
    uint64_t seed, offset;
};

// stage a [KT keys][HD] tile of K or V (which = 1 / 2) row-major and / or transposed into LDS
template <int HD, bool ROW, bool TR>
__device__ __forceinline__ void stage_kv(const AttnParams& p, int which, int t0, int len, int k0, int h,
                                         bf16_t (*rowm)[HD + PADE], bf16_t (*trm)[KT + PADE], int tid, int nthr) {
    for (int i = tid; i < KT * HD / 8; i += nthr) {
        const int key = i / (HD / 8), c8 = (i % (HD / 8)) * 8;
        bf16x8 v = zero8();
        if (k0 + key < len) v = *reinterpret_cast<const bf16x8*>(p.qkv + (((int64_t)(t0 + k0 + key) * 3 + which) * p.H + h) * HD + c8);
        if (ROW) *reinterpret_cast<bf16x8*>(&rowm[key][c8]) = v;
        if (TR) {
#pragma unroll
            for (int e = 0; e < 8; ++e) trm[c8 + e][key] = v[e];
        }
    }
}

// K / V tile staging split in two so that the global loads of tile kt+1 are in flight while tile kt is being consumed:
// load_kv fetches this thread's 16-byte pieces into registers, store_kv writes them row-major and / or transposed to LDS.
template <int HD>
struct KVRegs { bf16x8 k[(KTB * HD / 8 + 255) / 256], v[(KTB * HD / 8 + 255) / 256]; };

template <int HD>
__device__ __forceinline__ void load_kv(const AttnParams& p, int t0, int len, int k0, int h, int tid, KVRegs<HD>& r) {
    constexpr int NI = (KTB * HD / 8 + 255) / 256;
#pragma unroll
    for (int n = 0; n < NI; ++n) {
        const int i = tid + n * 256;
        const int key = i / (HD / 8), c8 = (i % (HD / 8)) * 8;
        r.k[n] = zero8();
        r.v[n] = zero8();
        if (i < KTB * HD / 8 && k0 + key < len) {
            const bf16_t* base = p.qkv + ((int64_t)(t0 + k0 + key) * 3 * p.H + h) * HD + c8;
            r.k[n] = *reinterpret_cast<const bf16x8*>(base + (int64_t)1 * p.H * HD);
            r.v[n] = *reinterpret_cast<const bf16x8*>(base + (int64_t)2 * p.H * HD);
        }
    }
}
template <int HD, bool KTR, bool VROW, bool VTR>     // K always row-major (+ transposed: KTR); V row-major and / or transposed
__device__ __forceinline__ void store_kv(const KVRegs<HD>& r, int tid, bf16_t (*k_lds)[HD + PADE], bf16_t (*v_lds)[HD + PADE],
                                         bf16_t (*tr_lds)[KTB + PADE]) {
    constexpr int NI = (KTB * HD / 8 + 255) / 256;
#pragma unroll
    for (int n = 0; n < NI; ++n) {
        const int i = tid + n * 256;
        if (i < KTB * HD / 8) {
            const int key = i / (HD / 8), c8 = (i % (HD / 8)) * 8;
            *reinterpret_cast<bf16x8*>(&k_lds[key][c8]) = r.k[n];
            if (VROW) *reinterpret_cast<bf16x8*>(&v_lds[key][c8]) = r.v[n];
            if (KTR || VTR) {
                const bf16x8 t = KTR ? r.k[n] : r.v[n];
#pragma unroll
                for (int e = 0; e < 8; ++e) tr_lds[c8 + e][key] = t[e];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------- forward
// MODE 0: forward (O, lse).  MODE 1: dQ (needs dout, lse, delta).
// The softmax work per score element is what bounds these kernels at hd = 32 (4 MFMAs per 32x32 tile), so it is kept
// minimal: the ALiBi term is one add of a per-lane constant and a per-tile offset folded into the scale FMA, the causal /
// length mask is evaluated only on tiles that cross the diagonal or the sequence end (wave-uniform branch), and the
// accumulator is rescaled only when some lane's running maximum actually moved.
template <int HD, int MODE, bool DROP>
__global__ __launch_bounds__(256) void attn_q_kernel(AttnParams p) {
    constexpr int KS = HD / 16, ND = HD / 32;
    __shared__ __attribute__((aligned(16))) bf16_t k_lds[KTB][HD + PADE];
    __shared__ __attribute__((aligned(16))) bf16_t v_lds[MODE == 1 ? KTB : 1][HD + PADE];         // row-major V (dQ only)
    __shared__ __attribute__((aligned(16))) bf16_t tr_lds[HD][KTB + PADE];                         // V^T (fwd) or K^T (dQ)
    const int s = blockIdx.y, h = blockIdx.z;
    const int t0 = p.cu[s], len = p.cu[s + 1] - t0;
    const int qb0 = blockIdx.x * 128;
    if (qb0 >= len) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int q = qb0 + w * 32 + r;
    const bool q_ok = q < len;
    const float c1 = p.scale * RESEL_LOG2E;
    const float slope2 = (p.slopes ? p.slopes[h] : 0.f) * RESEL_LOG2E;
    bf16x8 qf[KS], dof[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qf[ks] = zero8();
        dof[ks] = zero8();
        if (q_ok) {
            qf[ks] = *reinterpret_cast<const bf16x8*>(p.qkv + (((int64_t)(t0 + q) * 3 + 0) * p.H + h) * HD + 16 * ks + 8 * hh);
            if (MODE == 1) dof[ks] = *reinterpret_cast<const bf16x8*>(p.dout + ((int64_t)(t0 + q) * p.H + h) * HD + 16 * ks + 8 * hh);
        }
    }
    float m = NEG_BIG, l = 0.f;
    float lse2 = 0.f, dlt = 0.f;
    if (MODE == 1 && q_ok) {
        lse2 = p.lse[(int64_t)h * p.T + t0 + q];
        dlt = p.delta[(int64_t)h * p.T + t0 + q];
    }
    f32x16 acc[ND];
#pragma unroll
    for (int t = 0; t < ND; ++t) acc[t] = zero16();
    // The score accumulator STARTS from the constants of its row and tile, in raw (pre-scale) units: the ALiBi term
    // -slope (q - key) = slope row(i) - slope (q - k0) (a per-lane constant per register + a per-tile scalar) and, in the dQ
    // pass, -lse; exp2(acc * c1 [- m]) then needs one multiply-add per score and no separate bias / lse arithmetic (the
    // constants replace the zero fill, instruction for instruction).  dP starts from -delta the same way.
    const float inv_c1 = 1.f / c1;
    float crow[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) crow[i] = slope2 * inv_c1 * (float)acc_row(i, hh);
    const float lse_c = MODE == 1 ? lse2 * inv_c1 : 0.f;
    constexpr float RESCALE_THR = 5.f;              // forward: rescale the running sums only when a maximum grows by > 2^5

    // dropout: per-lane query word and the four key-quad offsets of a tile (rows 8g + 4hh + {0..3} = keys of one quad)
    uint32_t dq_word = 0, dk_off[4] = {0, 0, 0, 0};
    if (DROP) {
        dq_word = ((uint32_t)(t0 + q) * DROP_CQ) ^ drop_head_key(p.seed, p.offset, h);
#pragma unroll
        for (int g = 0; g < 4; ++g) dk_off[g] = (uint32_t)(2 * g + hh) * DROP_CK;
    }
    const int q_hi = min(len, qb0 + 128) - 1;                    // last query of this block
    const int nkeys = q_hi + 1;                                  // causal: keys <= q_hi
    const int wq_lo = qb0 + w * 32, wq_hi = wq_lo + 31;          // this wave's queries
    KVRegs<HD> kv;
    load_kv<HD>(p, t0, len, 0, h, tid, kv);
    for (int kb = 0; kb < nkeys; kb += KTB) {
        __syncthreads();
        store_kv<HD, MODE == 1, MODE == 1, MODE == 0>(kv, tid, k_lds, v_lds, tr_lds);
        __syncthreads();
        if (kb + KTB < nkeys) load_kv<HD>(p, t0, len, kb + KTB, h, tid, kv);   // in flight while this stage is consumed
      for (int sub = 0; sub < KTB / KT; ++sub) {
        const int k0 = kb + sub * KT, ko = sub * KT;
        if (k0 > wq_hi || k0 >= nkeys) break;                    // the rest lies in this wave's future (uniform per wave)
        const bool masked = (k0 + KT - 1 > wq_lo) || (wq_hi >= len);          // tile crosses the diagonal / the sequence end
        const float base = -slope2 * inv_c1 * (float)(q - k0) - lse_c;
        f32x16 st;
#pragma unroll
        for (int i = 0; i < 16; ++i) st[i] = crow[i] + base;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            st = mfma(*reinterpret_cast<const bf16x8*>(&k_lds[ko + r][16 * ks + 8 * hh]), qf[ks], st);
        if (masked) {
#pragma unroll
            for (int i = 0; i < 16; ++i) st[i] = (q_ok && (k0 + acc_row(i, hh)) <= q) ? st[i] : NEG_BIG;      // (key <= q < len)
        }
        uint32_t dw[4] = {0, 0, 0, 0};
        if (DROP) {
            const uint32_t kbase = (uint32_t)(k0 >> 2) * DROP_CK;
#pragma unroll
            for (int g = 0; g < 4; ++g) dw[g] = mix32(dq_word ^ (kbase + dk_off[g]));
        }
        f32x16 pt;
        if (MODE == 0) {
            float mloc = fmaxf(fmaxf(st[0], st[1]), st[2]);
#pragma unroll
            for (int i = 3; i < 15; i += 2) mloc = fmaxf(fmaxf(mloc, st[i]), st[i + 1]);
            mloc = fmaxf(mloc, st[15]);
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64)) * c1;
            if (__any(mloc > m + RESCALE_THR)) {                 // some lane's maximum grew past the threshold: rescale the running sums
                const float mnew = fmaxf(m, mloc);
                const float alpha = fast_exp2(m - mnew);
                l *= alpha;
#pragma unroll
                for (int t = 0; t < ND; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[t][i] *= alpha;
                m = mnew;
            }
            const float negm = -m;
            float psum = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {                       // masked scores (-1e30) give exp2(-huge) = 0 by themselves
                pt[i] = fast_exp2(__builtin_fmaf(st[i], c1, negm));
                psum += pt[i];
            }
            l += psum;
            if (DROP) {
#pragma unroll
                for (int i = 0; i < 16; ++i) pt[i] = ((dw[i >> 2] >> (8 * (i & 3))) & 0xffu) < p.thr ? pt[i] : 0.f;
            }
        } else {
            // dP^T = V dO^T ; dS^T = P^T (dP^T - delta)   (the 1/sqrt(d) factor of dS is applied once, to dQ, in the epilogue)
            f32x16 dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) dp[i] = DROP ? 0.f : -dlt;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                dp = mfma(*reinterpret_cast<const bf16x8*>(&v_lds[ko + r][16 * ks + 8 * hh]), dof[ks], dp);
            if (DROP) {                                          // dP = mask / (1 - p) * (dO V^T), then - delta
#pragma unroll
                for (int i = 0; i < 16; ++i) dp[i] = (((dw[i >> 2] >> (8 * (i & 3))) & 0xffu) < p.thr ? dp[i] * p.rp : 0.f) - dlt;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) pt[i] = fast_exp2(st[i] * c1) * dp[i];
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 bfr = acc_to_frag(pt, s2);
#pragma unroll
            for (int t = 0; t < ND; ++t) acc[t] = mfma(tr_frag(&tr_lds[32 * t + r][ko], s2, hh), bfr, acc[t]);
        }
      }
    }
    // epilogue: acc[t][reg] = X^T[d = 32t + row(reg)][q]
    float inv = MODE == 1 ? p.scale : 1.f;
    if (MODE == 0) {
        const float ltot = l + __shfl_xor(l, 32, 64);
        inv = ltot > 0.f ? (DROP ? p.rp : 1.f) / ltot : 0.f;
        if (q_ok && hh == 0) p.lse[(int64_t)h * p.T + t0 + q] = m + __log2f(fmaxf(ltot, 1e-37f));
    }
    if (q_ok) {
        bf16_t* dst = MODE == 0 ? p.out + ((int64_t)(t0 + q) * p.H + h) * HD
                                : p.dqkv + (((int64_t)(t0 + q) * 3 + 0) * p.H + h) * HD;
#pragma unroll
        for (int t = 0; t < ND; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {acc[t][4 * g] * inv, acc[t][4 * g + 1] * inv, acc[t][4 * g + 2] * inv, acc[t][4 * g + 3] * inv};
                *reinterpret_cast<bf16x4*>(dst + 32 * t + 8 * g + 4 * hh) = __builtin_convertvector(v, bf16x4);
            }
    }
}

// delta[h, tok] = sum_d dO * O
template <int HD>
__global__ void attn_delta_kernel(const bf16_t* __restrict__ out, const bf16_t* __restrict__ dout, float* __restrict__ delta, int T, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;         // over T * H
    if (i >= T * H) return;
    const int tok = i / H, h = i % H;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < HD; c += 8) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(out + (int64_t)i * HD + c);
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(dout + (int64_t)i * HD + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += (float)a[e] * (float)b[e];
    }
    delta[(int64_t)h * T + tok] = acc;
}

// ------------------------------------------------------------------------------------------------- dK / dV
template <int HD, bool DROP>
__global__ __launch_bounds__(256) void attn_dkv_kernel(AttnParams p) {
    constexpr int KS = HD / 16, ND = HD / 32;
    __shared__ __attribute__((aligned(16))) bf16_t qT[4][HD][KT + PADE];       // per-wave transposed Q tile  [d][q]
    __shared__ __attribute__((aligned(16))) bf16_t doT[4][HD][KT + PADE];      // per-wave transposed dO tile [d][q]
    __shared__ float s_red[4][2 * ND][16][64];
    __shared__ float s_lse[4][KT], s_dlt[4][KT];               // per-wave lse / delta of the query tile (one coalesced load)
    const int s = blockIdx.y, h = blockIdx.z;
    const int t0 = p.cu[s], len = p.cu[s + 1] - t0;
    const int k0 = blockIdx.x * KT;
    if (k0 >= len) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int key = k0 + r;
    const bool k_ok = key < len;
    const float c1 = p.scale * RESEL_LOG2E;
    const float slope2 = (p.slopes ? p.slopes[h] : 0.f) * RESEL_LOG2E;
    bf16x8 kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        kf[ks] = zero8();
        vf[ks] = zero8();
        if (k_ok) {
            kf[ks] = *reinterpret_cast<const bf16x8*>(p.qkv + (((int64_t)(t0 + key) * 3 + 1) * p.H + h) * HD + 16 * ks + 8 * hh);
            vf[ks] = *reinterpret_cast<const bf16x8*>(p.qkv + (((int64_t)(t0 + key) * 3 + 2) * p.H + h) * HD + 16 * ks + 8 * hh);
        }
    }
    f32x16 dkt[ND], dvt[ND];
#pragma unroll
    for (int t = 0; t < ND; ++t) { dkt[t] = zero16(); dvt[t] = zero16(); }
    // as in attn_q_kernel the accumulators start from the row constants (raw units): ALiBi -slope (q - key) = -slope row(i) -
    // slope (q0 - key), -lse of the query row (staged per tile as -lse / c1) for the scores and -delta for dP
    const float inv_c1 = 1.f / c1;
    float crow[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) crow[i] = -slope2 * inv_c1 * (float)acc_row(i, hh);
    // dropout: the word of (query row, this lane's key quad) serves the four lanes of a quad, one byte each; a lane
    // hashes the rows with (row & 3) == (lane & 3) and the quad exchanges them by DPP
    uint32_t dk_word = 0, dq_off[4] = {0, 0, 0, 0};
    const int dsh = 8 * (r & 3);
    if (DROP) {
        dk_word = ((uint32_t)(key >> 2) * DROP_CK) ^ drop_head_key(p.seed, p.offset, h);
#pragma unroll
        for (int g = 0; g < 4; ++g) dq_off[g] = (uint32_t)((r & 3) + 8 * g + 4 * hh) * DROP_CQ;
    }
    const int nqt = (len + KT - 1) / KT;
    const int qt_first = k0 / KT;                                // first query tile that can see these keys
    const int niter = (nqt - qt_first + 3) / 4;
    for (int it = 0; it < niter; ++it) {
        const int qt = qt_first + it * 4 + w;
        const bool active = qt < nqt;
        const int q0 = qt * KT;
        const int qa = q0 + r;                                   // the query this lane loads as an A-operand row
        bf16x8 qa_f[KS], doa_f[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qa_f[ks] = zero8();
            doa_f[ks] = zero8();
            if (active && qa < len) {
                qa_f[ks] = *reinterpret_cast<const bf16x8*>(p.qkv + (((int64_t)(t0 + qa) * 3 + 0) * p.H + h) * HD + 16 * ks + 8 * hh);
                doa_f[ks] = *reinterpret_cast<const bf16x8*>(p.dout + ((int64_t)(t0 + qa) * p.H + h) * HD + 16 * ks + 8 * hh);
            }
        }
        float l2q = 0.f, dlq = 0.f;
        if (active && hh == 0 && qa < len) {
            l2q = p.lse[(int64_t)h * p.T + t0 + qa];
            dlq = p.delta[(int64_t)h * p.T + t0 + qa];
        }
        __syncthreads();                                         // previous iteration's transposed tiles fully consumed
        if (hh == 0) { s_lse[w][r] = -l2q * inv_c1; s_dlt[w][r] = -dlq; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                qT[w][16 * ks + 8 * hh + e][r] = qa_f[ks][e];
                doT[w][16 * ks + 8 * hh + e][r] = doa_f[ks][e];
            }
        __syncthreads();
        if (!active) continue;
        const float offk = -slope2 * inv_c1 * (float)(q0 - key);
        f32x16 sacc, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = acc_row(i, hh);
            sacc[i] = crow[i] + offk + s_lse[w][row];
            dp[i] = DROP ? 0.f : s_dlt[w][row];
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            sacc = mfma(qa_f[ks], kf[ks], sacc);                 // S[q][key] (+ ALiBi - lse, raw units)
            dp = mfma(doa_f[ks], vf[ks], dp);                    // dP[q][key] - delta
        }
        f32x16 pm, ds;
        bool keep[16];
        if (DROP) {
            const uint32_t qbase = (uint32_t)(t0 + q0) * DROP_CQ;
            uint32_t mine[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) mine[g] = mix32((qbase + dq_off[g]) ^ dk_word);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                keep[4 * g + 0] = ((quad_bcast<0>(mine[g]) >> dsh) & 0xffu) < p.thr;
                keep[4 * g + 1] = ((quad_bcast<1>(mine[g]) >> dsh) & 0xffu) < p.thr;
                keep[4 * g + 2] = ((quad_bcast<2>(mine[g]) >> dsh) & 0xffu) < p.thr;
                keep[4 * g + 3] = ((quad_bcast<3>(mine[g]) >> dsh) & 0xffu) < p.thr;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) dp[i] = (keep[i] ? dp[i] * p.rp : 0.f) + s_dlt[w][acc_row(i, hh)];
        }
        const bool masked = (q0 < k0 + KT - 1) || (q0 + KT > len) || (k0 + KT > len);      // diagonal tile or ragged end (wave-uniform)
        if (masked) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int qq = q0 + acc_row(i, hh);
                const bool ok = k_ok && qq < len && key <= qq;
                const float e = ok ? fast_exp2(sacc[i] * c1) : 0.f;
                pm[i] = e;
                ds[i] = e * dp[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float e = fast_exp2(sacc[i] * c1);
                pm[i] = e;
                ds[i] = e * dp[i];
            }
        }
        if (DROP) {                                              // dV sees the dropped, rescaled probabilities
#pragma unroll
            for (int i = 0; i < 16; ++i) pm[i] = keep[i] ? pm[i] * p.rp : 0.f;
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pb = acc_to_frag(pm, s2), dsb = acc_to_frag(ds, s2);
#pragma unroll
            for (int t = 0; t < ND; ++t) {
                dvt[t] = mfma(tr_frag(&doT[w][32 * t + r][0], s2, hh), pb, dvt[t]);     // dV^T[d][key] += dO^T[d][q] P[q][key]
                dkt[t] = mfma(tr_frag(&qT[w][32 * t + r][0], s2, hh), dsb, dkt[t]);     // dK^T[d][key] += Q^T[d][q] dS[q][key]
            }
        }
    }
    // sum the four waves' partial accumulators, then store dK / dV rows
    __syncthreads();
#pragma unroll
    for (int t = 0; t < ND; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            s_red[w][t][i][lane] = dkt[t][i];
            s_red[w][ND + t][i][lane] = dvt[t][i];
        }
    __syncthreads();
    for (int pair = w; pair < 2 * ND; pair += 4) {
        const int which = pair < ND ? 1 : 2, t = pair < ND ? pair : pair - ND;
        if (!k_ok) continue;
        bf16_t* dst = p.dqkv + (((int64_t)(t0 + key) * 3 + which) * p.H + h) * HD + 32 * t;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * g + e;
                v[e] = ((s_red[0][pair][i][lane] + s_red[1][pair][i][lane]) + (s_red[2][pair][i][lane] + s_red[3][pair][i][lane])) *
                       (which == 1 ? p.scale : 1.f);                  // dK carries the 1/sqrt(d) of dS
            }
            *reinterpret_cast<bf16x4*>(dst + 8 * g + 4 * hh) = __builtin_convertvector(v, bf16x4);
        }
    }
}

inline bool attn_ok(int T, int S, int H, int hd, int max_seqlen, float p_drop) {
    return T > 0 && S > 0 && H > 0 && (hd == 32 || hd == 64) && max_seqlen > 0 && p_drop >= 0.f && p_drop < 1.f;
}
inline void set_dropout(AttnParams& p, float p_drop, uint64_t seed, uint64_t offset) {
    p.thr = (uint32_t)floorf((1.f - p_drop) * 255.f) + 1u;       // flash-attn's 8-bit keep threshold
    p.rp = 1.f / (1.f - p_drop);
    p.seed = seed;
    p.offset = offset;
}

template <int HD, bool DROP>
void launch_fwd(const AttnParams& p, dim3 grid, hipStream_t s) {
    launch_timed(RESEL_PROF_ATTN_FWD, attn_q_kernel<HD, 0, DROP>, grid, dim3(256), 0, s, p);
}
template <int HD, bool DROP>
void launch_bwd(const AttnParams& p, const bf16_t* out, dim3 gq, dim3 gk, hipStream_t s) {
    const int n = p.T * p.H;
    hipLaunchKernelGGL(attn_delta_kernel<HD>, dim3((n + 255) / 256), dim3(256), 0, s, out, p.dout, const_cast<float*>(p.delta), p.T, p.H);
    launch_timed(RESEL_PROF_ATTN_DQ, attn_q_kernel<HD, 1, DROP>, gq, dim3(256), 0, s, p);
    launch_timed(RESEL_PROF_ATTN_DKV, attn_dkv_kernel<HD, DROP>, gk, dim3(256), 0, s, p);
}

}  // namespace

extern "C" int resel_attn_varlen_fwd(const uint16_t* qkv, const int32_t* cu_seqlens, const float* slopes, uint16_t* out, float* lse,
                                     int T, int S, int H, int hd, int max_seqlen, float scale,
                                     float p_drop, uint64_t seed, uint64_t offset, resel_stream_t stream) {
    if (!qkv || !cu_seqlens || !out || !lse || !attn_ok(T, S, H, hd, max_seqlen, p_drop) || !aligned16(qkv) || !aligned16(out)) return RESEL_EINVAL;
    AttnParams p{(const bf16_t*)qkv, cu_seqlens, slopes, (bf16_t*)out, lse, nullptr, nullptr, nullptr, T, S, H, scale, 256u, 1.f, 0, 0};
    set_dropout(p, p_drop, seed, offset);
    dim3 grid((max_seqlen + 127) / 128, S, H);
    hipStream_t s = (hipStream_t)stream;
    const bool drop = p_drop > 0.f;
    if (hd == 32) { if (drop) launch_fwd<32, true>(p, grid, s); else launch_fwd<32, false>(p, grid, s); }
    else          { if (drop) launch_fwd<64, true>(p, grid, s); else launch_fwd<64, false>(p, grid, s); }
    return launch_status();
}

extern "C" size_t resel_attn_varlen_bwd_workspace_bytes(int T, int H, int hd) {
    (void)hd;
    return (size_t)T * H * sizeof(float);
}

extern "C" int resel_attn_varlen_bwd(const uint16_t* qkv, const int32_t* cu_seqlens, const float* slopes, const uint16_t* out,
                                     const float* lse, const uint16_t* dout, uint16_t* dqkv, void* workspace,
                                     int T, int S, int H, int hd, int max_seqlen, float scale,
                                     float p_drop, uint64_t seed, uint64_t offset, resel_stream_t stream) {
    if (!qkv || !cu_seqlens || !out || !lse || !dout || !dqkv || !workspace || !attn_ok(T, S, H, hd, max_seqlen, p_drop)) return RESEL_EINVAL;
    if (!aligned16(qkv) || !aligned16(out) || !aligned16(dout) || !aligned16(dqkv)) return RESEL_EINVAL;
    float* delta = (float*)workspace;
    AttnParams p{(const bf16_t*)qkv, cu_seqlens, slopes, nullptr, const_cast<float*>(lse), (const bf16_t*)dout, delta, (bf16_t*)dqkv, T, S, H, scale,
                 256u, 1.f, 0, 0};
    set_dropout(p, p_drop, seed, offset);
    hipStream_t s = (hipStream_t)stream;
    dim3 gq((max_seqlen + 127) / 128, S, H), gk((max_seqlen + KT - 1) / KT, S, H);
    const bool drop = p_drop > 0.f;
    if (hd == 32) { if (drop) launch_bwd<32, true>(p, (const bf16_t*)out, gq, gk, s); else launch_bwd<32, false>(p, (const bf16_t*)out, gq, gk, s); }
    else          { if (drop) launch_bwd<64, true>(p, (const bf16_t*)out, gq, gk, s); else launch_bwd<64, false>(p, (const bf16_t*)out, gq, gk, s); }
    return launch_status();
}
