// cgpt attention core for gfx950: packed variable-length causal attention with ALiBi, bf16 MFMA, fp32 softmax.
//
// Tokens of all sequences are packed [T, 3, H, hd] (q | k | v per token); cu_seqlens delimits the sequences.
// All three kernels are built on v_mfma_f32_32x32x16_bf16 and keep the SCORE tile transposed so that softmax
// statistics are per LANE (MFMA C/D layout: column = lane & 31, 16 rows in registers, row = (r&3) + 8(r>>2) + 4(lane>>5)):
//   forward / dQ kernel : S^T[key][q] = K Q^T      -> column = query: running max / sum / lse / delta are one value per lane,
//                         and the probability tile feeds the next product (O^T = V^T P^T, dQ^T = K^T dS^T) straight from
//                         registers as the B operand (rows = summed index; k-order 16s + 8(j>>2) + 4h + (j&3));
//   dK/dV kernel        : S[q][key] = Q K^T        -> column = key: dV^T = dO^T P and dK^T = Q^T dS sum over the row index q.
// Second edition (round 3).  What streams through LDS (K / V in the forward and dQ kernel, Q / dO in the dK/dV kernel) is kept
// as ONE swizzled row-major image per tensor: the products that want rows read it with ds_read_b128, the ones that want the
// transpose (V^T, K^T, dO^T, Q^T in the accumulator's k-order) read the SAME image with ds_read_b64_tr_b16 - no transposed
// copy, no 2-byte LDS stores, zero bank conflicts on either form (struct Img).  The images are filled by LDS-DMA
// (global_load_lds_dwordx4: no staging register, no LDS store instruction), double-buffered; the workgroups take their
// (sequence, block) from a device-built longest-first work list (attn_worklist_kernel), which is what a ragged causal batch
// needs to keep 256 CUs busy.  Measured at 32 sequences x 1026 tokens (+ 32 of one token) x 8 heads x 32: forward 87 -> 64 us
// (269 TFLOP/s of causal-half flops), dQ 87 -> 64 us, dK/dV 200 -> 77 us; profiles/r03_attention.md has the steps and the
// things that did not help (key-split wave pairs, a third LDS buffer, 5 waves per SIMD with spills, fp32-MFMA bias start).
// Arithmetic intensity is low for hd = 32 (4 MFMAs per 32x32 tile against ~16 exp2 per lane), so these kernels are
// VALU-issue-bound rather than MFMA-bound (PMC: VALU 58 % busy, MFMA 21 %); the whole attention share of a cgpt update is ~1 TFLOP.
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
    const float* delta;         // [H, T]  MINUS delta = -sum_d dO O (backward pre-pass)
    const float* comb;          // [H, T]  -(slope_h q + lse) / (scale log2 e): the score accumulator's per-query start value (dK/dV kernel)
    bf16_t* dqkv;
    int T, S, H;
    int nqb;                    // 128-token blocks of the longest sequence (grid bound)
    const uint32_t* wl;         // work list (attn_worklist): [0] = items, then (sequence << 16 | level), longest first; NULL: static order
    float scale;
    uint32_t thr;               // dropout: keep iff random byte < thr (256 = keep all)
    float rp;                   // 1 / (1 - p)
    uint64_t seed, offset;
    const unsigned long long* obase;   // device word added to `offset` when the kernel runs (resel_dropout_offset_base), or nullptr
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

// ------------------------------------------------------------------------------------------------- work list
// Causal attention over a ragged batch is a set of very unequal workgroups: the block of 128 queries (keys, in the dK/dV kernel)
// number b of a sequence streams b + 1 (blocks - b) blocks of the other side.  A grid in (sequence, block) order runs short and
// long ones as they come and idles most of the chip behind the last long ones; empty blocks of short sequences in front of real
// ones cost dispatch slots.  The work list holds exactly the real (sequence, block) items, LONGEST FIRST within chunks of
// WL_CHUNK consecutive sequences (level k = blocks streamed: every sequence with >= k blocks has one item per level), so the
// dispatcher - which hands workgroups out in id order to whichever CU has room - does longest-processing-time-first scheduling.
// Chunks, not the whole batch: the blocks of one (sequence, head) re-read the same K / V rows, and with every sequence's long
// blocks in flight at once those rows fall out of the L2 between uses (128 sequences of 1026: dQ 20 % slower than in sequence
// order).  Built on the device from cu_seqlens by one wave (ballot ranks: the order is deterministic).
constexpr int WL_MAX_LEVELS = 1024;
constexpr int WL_CHUNK = 32;
__global__ __launch_bounds__(64) void attn_worklist_kernel(const int32_t* __restrict__ cu, int S, int nlev, uint32_t* __restrict__ wl) {
    const int lane = threadIdx.x;
    uint32_t run = 0;
    for (int c0 = 0; c0 < S; c0 += WL_CHUNK) {
        const int s = c0 + lane;
        const int nb = (lane < WL_CHUNK && s < S) ? min((cu[s + 1] - cu[s] + KTB - 1) / KTB, nlev) : 0;
        int kmax = nb;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) kmax = max(kmax, __shfl_xor(kmax, d, 64));
        for (int k = kmax; k >= 1; --k) {
            const uint64_t m = __ballot(nb >= k);
            if (nb >= k) wl[1 + run + __popcll(m & ((1ull << lane) - 1ull))] = ((uint32_t)s << 16) | (uint32_t)k;
            run += (uint32_t)__popcll(m);
        }
    }
    if (lane == 0) wl[0] = run;
}
inline bool worklist_ok(int S, int max_seqlen) { return S < 65536 && (max_seqlen + KTB - 1) / KTB <= WL_MAX_LEVELS; }
inline size_t worklist_bytes(int S, int max_seqlen) {
    return worklist_ok(S, max_seqlen) ? ((size_t)S * ((max_seqlen + KTB - 1) / KTB) + 1) * sizeof(uint32_t) : 0;
}

// workgroup id -> (sequence, head, level).  XCD-aware: consecutive ids go round the 8 XCDs, so head h runs on XCD h % 8 for every
// sequence - the blocks of one (sequence, head), which all read the same K / V (Q / dO) rows, share one L2.  With a work list the
// items follow in its order; without, (sequence, block) order with each sequence's longest block first.  false: nothing to do.
__device__ __forceinline__ bool attn_block(const AttnParams& p, int& s, int& h, int& level) {
    const int hg = (p.H + 7) >> 3;
    const int j = blockIdx.x >> 3;
    h = (j % hg) * 8 + (blockIdx.x & 7);
    if (h >= p.H) return false;
    const int item = j / hg;
    if (p.wl != nullptr) {
        if ((uint32_t)item >= p.wl[0]) return false;
        const uint32_t e = p.wl[1 + item];
        s = (int)(e >> 16);
        level = (int)(e & 0xffffu);
    } else {
        s = item / p.nqb;
        level = p.nqb - item % p.nqb;
        if ((level - 1) * KTB >= p.cu[s + 1] - p.cu[s]) return false;
    }
    return true;
}

// ------------------------------------------------------------------------------------------------- forward / dQ
// Swizzled row-major LDS image of a [rows][HD] bf16 tile.  ONE image serves both operand forms (guide T10):
//   * row reads  (ds_read_b128: lane (r, hh) takes row r, 16-byte chunk 2 ks + hh) - the A operand of S^T = K Q^T, dP^T = V dO^T;
//   * transposed reads (ds_read_b64_tr_b16: a 16-lane group takes a 4-row x 16-column block and each lane receives a COLUMN)
//     - the A operand of O^T = V^T P^T and dQ^T = K^T dS^T in the accumulator's k-order 16 s + 8 (j >> 2) + 4 h + (j & 3).
// chunk ch of row `row` lives at chunk position ch ^ swz(row); with swz below both read forms are bank-conflict free
// (64-byte rows: 4 rows span the 64 banks, the 16-lane groups of ds_read_b128 hold 4 rows of each residue mod 4 whose
// (row >> 2) & 3 differ; 128-byte rows: 2 rows span the banks and the row bits 1..3 separate the 8 rows of a group).
#define LDS_AS __attribute__((address_space(3)))
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int HD>
struct Img {
    static constexpr int ROWB = HD * 2;
    static __device__ __forceinline__ int swz(int row) {
        return HD == 32 ? ((row >> 2) & 3) : ((((row >> 1) & 1) << 2) | ((row >> 2) & 3));
    }
    static __device__ __forceinline__ int off(int row, int ch) { return row * ROWB + ((ch ^ swz(row)) << 4); }
    // lane-constant part of the row-read address of k-step ks (tile rows start at a multiple of 32: swz does not see them)
    static __device__ __forceinline__ int row_off(int r, int hh, int ks) { return off(r, 2 * ks + hh); }
    // lane-constant part of the transposed-read address: k-step s2, first / second block of the step (sec), column tile t
    static __device__ __forceinline__ int tr_off(int lane, int s2, int sec, int t) {
        const int li = lane & 15, dh = (lane >> 4) & 1, hh = lane >> 5;
        const int row = 16 * s2 + 8 * sec + 4 * hh + (li >> 2);
        return off(row, 4 * t + 2 * dh + ((li & 3) >> 1)) + 8 * (li & 1);
    }
};
__device__ __forceinline__ bf16x8 lds_row(LDS_AS const char* a) { return *reinterpret_cast<LDS_AS const bf16x8*>(a); }
__device__ __forceinline__ bf16x8 lds_tr(LDS_AS const char* lo, LDS_AS const char* hi) {      // EXEC must be all ones
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)lo);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)hi);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}
// max / sum over the two 32-lane halves of a wave in the VALU (v_permlane32_swap; no LDS round trip)
__device__ __forceinline__ void halves(float x, float& a, float& b) {      // a = lower half's value, b = upper half's, in all lanes
    // inline asm: with both operands the same value, hipcc (ROCm 7.2) folds the two results of
    // __builtin_amdgcn_permlane32_swap into one register (a + b became 2 a)
    a = x;
    b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float half_max(float x) { float a, b; halves(x, a, b); return fmaxf(a, b); }
__device__ __forceinline__ float half_sum(float x) { float a, b; halves(x, a, b); return a + b; }

template <int V> struct IntC { static constexpr int value = V; };
// Asynchronous global -> LDS copies (global_load_lds_dwordx4 / _dword): the LDS address is the wave-uniform base (M0) + lane * size.
// Inline assembly on purpose: hipcc guards every LDS read that may alias a builtin copy's destination with s_waitcnt vmcnt(0);
// written this way the copies are invisible to its counters and the kernel retires them itself with vm_wait<N>().
__device__ __forceinline__ void glds16(const char* g, LDS_AS char* lds_wave_base) {
    const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(l) : "memory");
}
__device__ __forceinline__ void glds4(const float* g, LDS_AS char* lds_wave_base) {
    const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(l) : "memory");
}
template <int N>
__device__ __forceinline__ void vm_wait() {                      // s_waitcnt vmcnt(N) only (expcnt / lgkmcnt left alone)
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ float sgpr(float x) {                 // a wave-uniform value, held in a scalar register
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}

// MODE 0: forward (O, lse).  MODE 1: dQ (needs dout, lse, delta).
// Workgroup = 128 queries x 4 * KSPL waves: wave w works on query tile w & 3 (32 queries, one per lane column) and, with KSPL = 2,
// on every second 32-key tile of a staged key block - the two waves of a query tile keep private running statistics /
// accumulators and are merged through LDS at the end (halves the serial tile chain of the late query blocks, which is what
// bounds a launch of few long sequences).  Key blocks (KTB keys of K and V) are double-buffered in LDS: one barrier per block,
// the global loads of block i + 2 are in flight during block i + 1.
// The softmax work per score element is what bounds these kernels at hd = 32 (4 MFMAs per 32x32 tile), so it is kept
// minimal: the ALiBi term is one add of a per-lane constant and a per-tile offset folded into the accumulator's initial
// value, the causal / length mask is evaluated only on tiles that cross the diagonal or the sequence end (wave-uniform
// branch), the accumulator is rescaled only when some lane's running maximum actually moved, and the forward takes two
// tiles per step (one maximum exchange / rescale test / loop turn per 64 keys, two independent MFMA chains in flight).
template <int HD, int MODE, bool DROP, int KSPL>
#ifdef ATTN_AB_OCC5
__global__ __launch_bounds__(256 * KSPL, HD == 32 && MODE == 0 && !DROP ? 5 : 1) void attn_q_kernel(AttnParams p) {
#else
__global__ __launch_bounds__(256 * KSPL, KSPL == 2 && HD == 32 ? 4 : 1) void attn_q_kernel(AttnParams p) {
#endif
    constexpr int KS = HD / 16, ND = HD / 32, NTHR = 256 * KSPL;
    constexpr int ROWB = HD * 2, IMG = KTB * ROWB;
    constexpr int NI = KTB * HD / 8 / NTHR;                      // 16-byte pieces of K (and of V) per thread and key block
#ifdef ATTN_AB_NT1
    constexpr int NTS = 1;
#else
    constexpr int NTS = MODE == 0 ? 2 : 1;                       // tiles per step
#endif
    static_assert(NI >= 1 && KTB / KT == 4, "staging split");
    constexpr int COMB = 4 * 64 * (16 * ND + 2) * 4;             // merge buffer of the key-split pairs (aliases the images)
    static_assert(KSPL == 1 || COMB <= 4 * IMG, "merge buffer");     // NBUF * 2 * IMG >= 4 * IMG
#ifdef ATTN_AB_NBUF3
    constexpr int NBUF = 3;                                      // one barrier per key block, but 48 KB: 3 workgroups per CU - measured slower
#else
    constexpr int NBUF = 2;
#endif
    __shared__ __attribute__((aligned(16))) char smem[NBUF * 2 * IMG];   // [buffer][K image | V image]
    LDS_AS char* const lds = (LDS_AS char*)smem;

    int s, h, level;
    if (!attn_block(p, s, h, level)) return;
    const int t0 = p.cu[s], len = p.cu[s + 1] - t0;
    const int qb0 = (level - 1) * 128;                           // level = key blocks this query block streams
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qt = w & 3, kh = w >> 2;
    const int r = lane & 31, hh = lane >> 5;
    const int q = qb0 + qt * 32 + r;
    const bool q_ok = q < len;
    const float c1 = p.scale * RESEL_LOG2E;
    const float slope2 = (p.slopes ? p.slopes[h] : 0.f) * RESEL_LOG2E;
    bf16x8 qf[KS], dof[MODE == 1 ? KS : 1];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qf[ks] = zero8();
        if (MODE == 1) dof[ks] = zero8();
        if (q_ok) {
            qf[ks] = *reinterpret_cast<const bf16x8*>(p.qkv + (((int64_t)(t0 + q) * 3 + 0) * p.H + h) * HD + 16 * ks + 8 * hh);
            if (MODE == 1) dof[ks] = *reinterpret_cast<const bf16x8*>(p.dout + ((int64_t)(t0 + q) * p.H + h) * HD + 16 * ks + 8 * hh);
        }
    }
    float m = NEG_BIG, l = 0.f;
    float lse2 = 0.f, dlt = 0.f;
    if (MODE == 1 && q_ok) {
        lse2 = p.lse[(int64_t)h * p.T + t0 + q];
        dlt = -p.delta[(int64_t)h * p.T + t0 + q];
    }
    f32x16 acc[ND];
#pragma unroll
    for (int t = 0; t < ND; ++t) acc[t] = zero16();
    // The score accumulator STARTS from the constants of its row and tile, in raw (pre-scale) units: the ALiBi term
    // -slope (q - key) = slope row(i) - slope (q - k0) (a per-lane constant per register + a per-tile scalar) and, in the dQ
    // pass, -lse; exp2(acc * c1 [- m]) then needs one multiply-add per score and no separate bias / lse arithmetic (the
    // constants replace the zero fill, instruction for instruction).  dP starts from -delta the same way.
    const float inv_c1 = 1.f / c1;
    const float sl = slope2 * inv_c1;
    // ... through ONE v_mfma_f32_32x32x2_f32 per tile (k = 2, fp32 operands: exact): A[key row][0] = slope * row, B[0][q] = 1 and
    // A[key row][1] = 1, B[1][q] = -slope (q - k0) [- lse]; lane (r, hh) holds A[r][hh] / B[hh][r].  16 VALU adds per tile become
    // one matrix instruction on a pipe that is ~15 % busy here, and one VALU operation for the B value.
    const float bias_a = hh == 0 ? sl * (float)r : 1.f;
    const float lse_c = MODE == 1 ? lse2 * inv_c1 : 0.f;
    constexpr float RESCALE_THR = 5.f;              // forward: rescale the running sums only when a maximum grows by > 2^5

    // dropout: per-lane query word and the four key-quad offsets of a tile (rows 8g + 4hh + {0..3} = keys of one quad)
    uint32_t dq_word = 0, dk_off[4] = {0, 0, 0, 0};
    if (DROP) {
        dq_word = ((uint32_t)(t0 + q) * DROP_CQ) ^ drop_head_key(p.seed, p.offset + (p.obase ? *p.obase : 0ull), h);
#pragma unroll
        for (int g = 0; g < 4; ++g) dk_off[g] = (uint32_t)(2 * g + hh) * DROP_CK;
    }
    const int q_hi = min(len, qb0 + 128) - 1;                    // last query of this block
    const int nkeys = q_hi + 1;                                  // causal: keys <= q_hi
    const int nkb = (nkeys + KTB - 1) / KTB;
    const int wq_lo = qb0 + qt * 32;                             // this wave's queries; a wave past the sequence end only stages
    const int wq_hi = wq_lo < len ? wq_lo + 31 : -1;

    // lane-constant LDS offsets of the operand reads (the tile's first row and the image select are added per tile)
    int roff[KS], toff[2][2][ND];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) roff[ks] = Img<HD>::row_off(r, hh, ks);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int sec = 0; sec < 2; ++sec)
#pragma unroll
            for (int t = 0; t < ND; ++t) toff[s2][sec][t] = Img<HD>::tr_off(lane, s2, sec, t);

    // staging: thread -> NI (key row, chunk) pieces of K and of V
    // staging by LDS-DMA (global_load_lds: no register in between, no LDS store instruction).  A wave-instruction fills 64
    // consecutive 16-byte slots of an image, so the swizzle is applied on the SOURCE side: slot (row, chunk') is fed from
    // chunk' ^ swz(row) of that row.  Keys past the sequence end re-read its last row (finite data; their scores are masked).
    int srow[NI];
    uint32_t sch[NI];
#pragma unroll
    for (int n = 0; n < NI; ++n) {
        const int slot = (n * (NTHR / 64) + w) * 64 + lane;
        srow[n] = slot / (HD / 8);
        sch[n] = (uint32_t)((slot % (HD / 8)) ^ Img<HD>::swz(srow[n])) * 16u;
    }
    auto issue_kv = [&](int kb, int buf) {
        const char* const kbase = reinterpret_cast<const char*>(p.qkv + (((int64_t)(t0 + kb) * 3 + 1) * p.H + h) * HD);
        const char* const vbase = kbase + (int64_t)p.H * HD * 2;
        const int last = len - kb - 1;
        LDS_AS char* const base = lds + buf * 2 * IMG;
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            const int64_t go = (int64_t)min(srow[n], last) * (3 * p.H * HD * 2) + sch[n];
            glds16(kbase + go, base + (n * (NTHR / 64) + w) * 1024);
            glds16(vbase + go, base + IMG + (n * (NTHR / 64) + w) * 1024);
        }
    };
    constexpr int NVM = 2 * NI;                                  // vector-memory operations per thread and key block

    // the per-query operands have landed before the key loop starts (the compiler would otherwise place its wait at their first
    // use inside the loop, where vmcnt - an in-order counter - also covers the copies in flight)
    __builtin_amdgcn_s_waitcnt(0x0F70);                          // vmcnt(0)
    issue_kv(0, 0);
    if (nkb > 1) issue_kv(KTB, 1);

    for (int ib = 0; ib < nkb; ++ib) {
        const int kb = ib * KTB;
        // block ib has landed (the copies of block ib + 1 may stay in flight: vmcnt retires in order); with three buffers the
        // barrier also says that every wave is done with block ib - 1, whose buffer the copies of block ib + 2 overwrite
        if (ib + 1 < nkb) vm_wait<NVM>(); else vm_wait<0>();
        __syncthreads();
        if (NBUF == 3 && ib + 2 < nkb) issue_kv(kb + 2 * KTB, (ib + 2) % 3);
        LDS_AS const char* const kimg = lds + (ib % NBUF) * 2 * IMG;
        LDS_AS const char* const vimg = kimg + IMG;

        // One instantiation serves every step: the accumulators then live in the same registers on every path (two code paths
        // - a one-tile and a two-tile step - made hipcc copy the 16 accumulator registers behind the last MFMA of each step, a
        // full MFMA-pipeline drain per step).  An odd tile count (diagonal key block only) runs the second tile fully masked.
        auto step = [&](auto NTc, int sub_a, int sub_b, bool b_on) {
            constexpr int NT = decltype(NTc)::value;
            const int sub[2] = {sub_a, sub_b};
            f32x16 st[NT];
            uint32_t dw[NT][4];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int ko = sub[n] * KT, k0 = kb + ko;
                const float bias_b = hh == 0 ? 1.f : -sl * (float)(q - k0) - lse_c;
                st[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bias_a, bias_b, zero16(), 0, 0, 0);
                if (n == 0 || b_on) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
#if defined(ATTN_AB_NOQK)
                        st[n][ks] += (float)qf[ks][0];
#elif defined(ATTN_AB_NOKROW)
                        st[n] = mfma(qf[KS - 1 - ks], qf[ks], st[n]);
#else
                        st[n] = mfma(lds_row(kimg + ko * ROWB + roff[ks]), qf[ks], st[n]);
#endif
                    }
                }
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int k0 = kb + sub[n] * KT;
                if ((k0 + KT - 1 > wq_lo) || (wq_hi >= len)) {   // tile crosses the diagonal / the sequence end (wave-uniform)
                    int qm = q_ok && (n == 0 || b_on) ? q - k0 : -1;                 // opaque to the optimiser: it hoists the 16 compares of this rare
                    asm volatile("" : "+v"(qm));                 // branch into every turn of the tile loop otherwise
#pragma unroll
                    for (int i = 0; i < 16; ++i) st[n][i] = acc_row(i, hh) <= qm ? st[n][i] : NEG_BIG;                      // (key <= q < len)
                }
                if (DROP) {
                    const uint32_t kbase = (uint32_t)(k0 >> 2) * DROP_CK;
#pragma unroll
                    for (int g = 0; g < 4; ++g) dw[n][g] = mix32(dq_word ^ (kbase + dk_off[g]));
                }
            }
            f32x16 pt[NT];
            if (MODE == 0) {
                float mloc = fmaxf(fmaxf(st[0][0], st[0][1]), st[0][2]);
#ifndef ATTN_AB_NOMAX
#pragma unroll
                for (int i = 3; i < 15; i += 2) mloc = fmaxf(fmaxf(mloc, st[0][i]), st[0][i + 1]);
                mloc = fmaxf(mloc, st[0][15]);
                if (NT == 2) {
                    float m1 = fmaxf(fmaxf(st[NT - 1][0], st[NT - 1][1]), st[NT - 1][2]);
#pragma unroll
                    for (int i = 3; i < 15; i += 2) m1 = fmaxf(fmaxf(m1, st[NT - 1][i]), st[NT - 1][i + 1]);
                    mloc = fmaxf(fmaxf(mloc, m1), st[NT - 1][15]);
                }
#endif
#ifdef ATTN_AB_NOMAX
                mloc = st[0][0] * c1;
#else
                mloc = half_max(mloc) * c1;
#endif
#ifdef ATTN_AB_NORESCALE
                if (__any(mloc > m + 1e30f)) {
#else
                if (__any(mloc > m + RESCALE_THR)) {
#endif             // some lane's maximum grew past the threshold: rescale the running sums
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
                float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {               // masked scores (-1e30) give exp2(-huge) = 0 by themselves
#ifdef ATTN_AB_NOEXP
                        pt[n][i] = __builtin_fmaf(st[n][i], c1, negm);
#else
                        pt[n][i] = fast_exp2(__builtin_fmaf(st[n][i], c1, negm));
#endif
#ifndef ATTN_AB_NOSUM
                        ps[i & 3] += pt[n][i];
#endif
                    }
                l += (ps[0] + ps[1]) + (ps[2] + ps[3]);
                if (DROP) {
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int i = 0; i < 16; ++i) pt[n][i] = ((dw[n][i >> 2] >> (8 * (i & 3))) & 0xffu) < p.thr ? pt[n][i] : 0.f;
                }
            } else {
                // dP^T = V dO^T ; dS^T = P^T (dP^T - delta)   (the 1/sqrt(d) factor of dS is applied once, to dQ, in the epilogue)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int ko = sub[n] * KT;
                    f32x16 dp;
#pragma unroll
                    for (int i = 0; i < 16; ++i) dp[i] = DROP ? 0.f : -dlt;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) dp = mfma(lds_row(vimg + ko * ROWB + roff[ks]), dof[ks], dp);
                    if (DROP) {                                  // dP = mask / (1 - p) * (dO V^T), then - delta
#pragma unroll
                        for (int i = 0; i < 16; ++i) dp[i] = (((dw[n][i >> 2] >> (8 * (i & 3))) & 0xffu) < p.thr ? dp[i] * p.rp : 0.f) - dlt;
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) pt[n][i] = fast_exp2(st[n][i] * c1) * dp[i];
                }
            }
            LDS_AS const char* const timg = MODE == 0 ? vimg : kimg;       // V^T (forward) / K^T (dQ) through transposed reads
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int ko = sub[n] * KT;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8 bfr = acc_to_frag(pt[n], s2);
#pragma unroll
                    for (int t = 0; t < ND; ++t) {
#if defined(ATTN_AB_NOPV)
                        acc[t][4 * s2 + n] += pt[n][8 * s2] + (float)bfr[0];
#elif defined(ATTN_AB_NOTR)
                        acc[t] = mfma(qf[s2 % KS], bfr, acc[t]);
#else
                        acc[t] = mfma(lds_tr(timg + ko * ROWB + toff[s2][0][t], timg + ko * ROWB + toff[s2][1][t]), bfr, acc[t]);
#endif
                    }
                }
            }
        };
        auto valid = [&](int sub) { const int k0 = kb + sub * KT; return k0 <= wq_hi && k0 < nkeys; };      // wave-uniform
        if (NTS == 2) {
            // KSPL = 2: one step on the tiles {kh, kh + 2}; KSPL = 1: steps {0, 1}, {2, 3}
#pragma unroll 1
            for (int g = 0; g < 2 / KSPL; ++g) {
                const int a = KSPL == 2 ? kh : 2 * g, b = KSPL == 2 ? kh + 2 : 2 * g + 1;
                if (valid(a)) step(IntC<2>{}, a, b, valid(b));
            }
        } else {
#pragma unroll 1
            for (int g = 0; g < 4 / KSPL; ++g) {
                const int a = KSPL == 2 ? kh + 2 * g : g;
                if (valid(a)) step(IntC<1>{}, a, a, true);
            }
        }
        if (NBUF == 2) {                                         // two buffers (hd 64: LDS): a second barrier frees this block's buffer
            __syncthreads();
            if (ib + 2 < nkb) issue_kv(kb + 2 * KTB, ib & 1);
        }
    }

    // merge the key-split pair: the kh = 1 wave hands its statistics / accumulator to its kh = 0 partner through LDS
    if (KSPL == 2) {
        LDS_AS float* const cb = (LDS_AS float*)lds + (qt * (16 * ND + 2)) * 64 + lane;
        __syncthreads();                                         // every wave is done with the last key block's images
        if (kh == 1) {
#pragma unroll
            for (int t = 0; t < ND; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) cb[(16 * t + i) * 64] = acc[t][i];
            cb[16 * ND * 64] = m;
            cb[(16 * ND + 1) * 64] = l;
        }
        __syncthreads();
        if (kh == 1) return;
        if (MODE == 0) {
            const float m1 = cb[16 * ND * 64], l1 = cb[(16 * ND + 1) * 64];
            const float mn = fmaxf(m, m1);
            const float a0 = fast_exp2(m - mn), a1 = fast_exp2(m1 - mn);
            l = l * a0 + l1 * a1;
#pragma unroll
            for (int t = 0; t < ND; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][i] = acc[t][i] * a0 + cb[(16 * t + i) * 64] * a1;
            m = mn;
        } else {
#pragma unroll
            for (int t = 0; t < ND; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][i] += cb[(16 * t + i) * 64];
        }
    }
    // epilogue: acc[t][reg] = X^T[d = 32t + row(reg)][q]
    float inv = MODE == 1 ? p.scale : 1.f;
    if (MODE == 0) {
        const float ltot = half_sum(l);
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

// Backward pre-pass, per (sequence, token, head): ndelta[h, tok] = -sum_d dO * O and comb[h, tok] = -(slope_h q + lse) / c1 with
// q the token's position in its sequence - the per-query start values of the dP / score accumulators of the dK/dV kernel
// (which copies them global -> LDS without touching a register).
template <int HD>
__global__ void attn_delta_kernel(const bf16_t* __restrict__ out, const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                  const int32_t* __restrict__ cu, const float* __restrict__ slopes, float* __restrict__ ndelta,
                                  float* __restrict__ comb, int T, int H, float inv_c1) {
    const int s = blockIdx.y;
    const int t0 = cu[s], len = cu[s + 1] - t0;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;         // over len * H
    if (j >= len * H) return;
    const int q = j / H, h = j % H;
    const int64_t i = (int64_t)(t0 + q) * H + h;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < HD; c += 8) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(out + i * HD + c);
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(dout + i * HD + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += (float)a[e] * (float)b[e];
    }
    const int64_t at = (int64_t)h * T + t0 + q;
    ndelta[at] = -acc;
    comb[at] = -((slopes ? slopes[h] : 0.f) * RESEL_LOG2E * (float)q + lse[at]) * inv_c1;
}

// ------------------------------------------------------------------------------------------------- dK / dV
// Workgroup = 128 keys x 4 waves: wave w owns the keys [k0 + 32 w, +32) - K and V fragments of its keys stay in registers as B
// operands for the whole launch, dK^T / dV^T of its keys in its accumulators (no cross-wave reduction).  The query side
// streams through LDS in blocks of 128 queries, double-buffered (one barrier per block), from the diagonal block to the end of
// the sequence: Q and dO as swizzled row-major images (row reads feed S = Q K^T and dP = dO V^T, transposed reads of the SAME
// images feed dV^T = dO^T P and dK^T = Q^T dS), plus one fp32 value pair per query: -(slope q + lse) / c1 (ALiBi row part and
// log-sum-exp folded into the score accumulator's initial value) and -delta (initial value of dP).
template <int HD, bool DROP>
__global__ __launch_bounds__(256) void attn_dkv_kernel(AttnParams p) {
    constexpr int KS = HD / 16, ND = HD / 32, NTHR = 256;
    constexpr int ROWB = HD * 2, IMG = KTB * ROWB;
    constexpr int NI = KTB * HD / 8 / NTHR;                      // 16-byte pieces of Q (and of dO) per thread and query block
    constexpr int BUF = 2 * IMG + 2 * KTB * 4;                   // Q image | dO image | comb[128] | -delta[128]
    __shared__ __attribute__((aligned(16))) char smem[3 * BUF];  // three buffers: one barrier per query block
    LDS_AS char* const lds = (LDS_AS char*)smem;

    int s, h, level;
    if (!attn_block(p, s, h, level)) return;
    const int t0 = p.cu[s], len = p.cu[s + 1] - t0;
    const int kb0 = ((len + KTB - 1) / KTB - level) * KTB;       // level = query blocks this key block streams (diagonal .. end)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int k0 = kb0 + w * KT;                                 // this wave's keys
    const int key = k0 + r;
    const bool k_ok = key < len;
    const bool w_on = k0 < len;                                  // a wave past the sequence end only stages
    const float c1 = p.scale * RESEL_LOG2E;
    const float slope2 = (p.slopes ? p.slopes[h] : 0.f) * RESEL_LOG2E;
    const float inv_c1 = 1.f / c1;
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
    const float lane_c = slope2 * inv_c1 * (float)key;           // ALiBi -slope (q - key): the key part (the query part is staged)
    // dropout: the word of (query row, this lane's key quad) serves the four lanes of a quad, one byte each; a lane
    // hashes the rows with (row & 3) == (lane & 3) and the quad exchanges them by DPP
    uint32_t dk_word = 0, dq_off[4] = {0, 0, 0, 0};
    const int dsh = 8 * (r & 3);
    if (DROP) {
        dk_word = ((uint32_t)(key >> 2) * DROP_CK) ^ drop_head_key(p.seed, p.offset + (p.obase ? *p.obase : 0ull), h);
#pragma unroll
        for (int g = 0; g < 4; ++g) dq_off[g] = (uint32_t)((r & 3) + 8 * g + 4 * hh) * DROP_CQ;
    }
    int roff[KS], toff[2][ND];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) roff[ks] = Img<HD>::row_off(r, hh, ks);
#pragma unroll
    for (int sec = 0; sec < 2; ++sec)
#pragma unroll
        for (int t = 0; t < ND; ++t) toff[sec][t] = Img<HD>::tr_off(lane, 0, sec, t);     // k-step s2: + 16 s2 rows (swz does not see them)

    // staging by LDS-DMA (global_load_lds: no register in between, no LDS store instruction).  A wave-instruction fills 64
    // consecutive 16-byte slots of an image, so the swizzle is applied on the SOURCE side: slot (row, chunk') is fed from
    // chunk' ^ swz(row) of that row.  Rows past the sequence end re-read its last row (finite data; their scores are masked).
    // Per query block and thread: NI slots of Q, NI of dO (16 bytes) and one of the 256 per-query floats (4 bytes).
    int srow[NI];
    uint32_t sq[NI], so[NI];                                     // byte offset of the slot's chunk within its row
#pragma unroll
    for (int n = 0; n < NI; ++n) {
        const int slot = (n * 4 + w) * 64 + lane;
        srow[n] = slot / (HD / 8);
        const uint32_t ch = (uint32_t)((slot % (HD / 8)) ^ Img<HD>::swz(srow[n]));
        sq[n] = ch * 16u;
        so[n] = ch * 16u;
    }
    const float* const fsrc = (tid < KTB ? p.comb : p.delta) + (int64_t)h * p.T + t0;
    const int frow = tid & (KTB - 1);
    auto issue_q = [&](int qb, int buf) {
        const char* const qbase = reinterpret_cast<const char*>(p.qkv + ((int64_t)(t0 + qb) * 3 * p.H + h) * HD);
        const char* const obase = reinterpret_cast<const char*>(p.dout + ((int64_t)(t0 + qb) * p.H + h) * HD);
        const int last = len - qb - 1;
        LDS_AS char* const base = lds + buf * BUF;
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            const int row = min(srow[n], last);
            glds16(qbase + (int64_t)row * (3 * p.H * HD * 2) + sq[n], base + (n * 4 + w) * 1024);
            glds16(obase + (int64_t)row * (p.H * HD * 2) + so[n], base + IMG + (n * 4 + w) * 1024);
        }
        glds4(fsrc + qb + min(frow, last), base + 2 * IMG + w * 256);
    };
    constexpr int NVM = 2 * NI + 1;                              // vector-memory operations per thread and query block

    const int nblk = (len - kb0 + KTB - 1) / KTB;                // query blocks from the diagonal one to the end of the sequence
    __builtin_amdgcn_s_waitcnt(0x0F70);                          // vmcnt(0): K / V fragments landed (see attn_q_kernel)
    issue_q(kb0, 0);
    if (nblk > 1) issue_q(kb0 + KTB, 1);

    for (int ib = 0; ib < nblk; ++ib) {
        const int qb = kb0 + ib * KTB;
        // block ib has landed (the copies of block ib + 1 may stay in flight: vmcnt retires in order); the barrier also says that
        // every wave is done with block ib - 1, whose buffer the copies of block ib + 2 now overwrite
        if (ib + 1 < nblk) vm_wait<NVM>(); else vm_wait<0>();
        __syncthreads();
        if (ib + 2 < nblk) issue_q(qb + 2 * KTB, (ib + 2) % 3);
        LDS_AS const char* const qimg = lds + (ib % 3) * BUF;
        LDS_AS const char* const oimg = qimg + IMG;
        LDS_AS const float* const comb = reinterpret_cast<LDS_AS const float*>(qimg + 2 * IMG);
        LDS_AS const float* const sdl = comb + KTB;
#pragma unroll 1
        for (int sub = 0; sub < KTB / KT; ++sub) {
            const int ro = sub * KT, q0 = qb + ro;
            if (!w_on || q0 + KT - 1 < k0 || q0 >= len) continue;        // wave-uniform: tile before the diagonal / past the end
            f32x16 sacc, dp;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 cv = *reinterpret_cast<LDS_AS const f32x4*>(comb + ro + 8 * g + 4 * hh);
                const f32x4 dv = *reinterpret_cast<LDS_AS const f32x4*>(sdl + ro + 8 * g + 4 * hh);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sacc[4 * g + e] = cv[e] + lane_c;
                    dp[4 * g + e] = DROP ? 0.f : dv[e];
                }
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                sacc = mfma(lds_row(qimg + ro * ROWB + roff[ks]), kf[ks], sacc);          // S[q][key] (+ ALiBi - lse, raw units)
                dp = mfma(lds_row(oimg + ro * ROWB + roff[ks]), vf[ks], dp);              // dP[q][key] - delta
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
                for (int g = 0; g < 4; ++g) {
                    const f32x4 dv = *reinterpret_cast<LDS_AS const f32x4*>(sdl + ro + 8 * g + 4 * hh);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dp[4 * g + e] = (keep[4 * g + e] ? dp[4 * g + e] * p.rp : 0.f) + dv[e];
                }
            }
            if ((q0 < k0 + KT - 1) || (q0 + KT > len) || (k0 + KT > len)) {              // diagonal tile or ragged end (wave-uniform)
                int km = k_ok ? key - q0 : 1 << 20;              // opaque to the optimiser (see attn_q_kernel)
                int qn = len - q0;
                asm volatile("" : "+v"(km), "+s"(qn));
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = acc_row(i, hh);
                    const float e = (row >= km && row < qn) ? fast_exp2(sacc[i] * c1) : 0.f;      // key <= q < len
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
            if (DROP) {                                          // dV sees the dropped, rescaled probabilities
#pragma unroll
                for (int i = 0; i < 16; ++i) pm[i] = keep[i] ? pm[i] * p.rp : 0.f;
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pb = acc_to_frag(pm, s2), dsb = acc_to_frag(ds, s2);
                const int so = (ro + 16 * s2) * ROWB;
#pragma unroll
                for (int t = 0; t < ND; ++t) {
                    dvt[t] = mfma(lds_tr(oimg + so + toff[0][t], oimg + so + toff[1][t]), pb, dvt[t]);     // dV^T[d][key] += dO^T[d][q] P[q][key]
                    dkt[t] = mfma(lds_tr(qimg + so + toff[0][t], qimg + so + toff[1][t]), dsb, dkt[t]);    // dK^T[d][key] += Q^T[d][q] dS[q][key]
                }
            }
        }
    }
    // epilogue: dkt / dvt [t][reg] = X^T[d = 32 t + row(reg)][key]
    if (k_ok) {
#pragma unroll
        for (int which = 1; which <= 2; ++which) {
            bf16_t* dst = p.dqkv + (((int64_t)(t0 + key) * 3 + which) * p.H + h) * HD;
            const float sc = which == 1 ? p.scale : 1.f;         // dK carries the 1/sqrt(d) of dS
#pragma unroll
            for (int t = 0; t < ND; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x16& x = which == 1 ? dkt[t] : dvt[t];
                    const f32x4 v = {x[4 * g] * sc, x[4 * g + 1] * sc, x[4 * g + 2] * sc, x[4 * g + 3] * sc};
                    *reinterpret_cast<bf16x4*>(dst + 32 * t + 8 * g + 4 * hh) = __builtin_convertvector(v, bf16x4);
                }
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
    p.obase = p_drop > 0.f ? resel::dropout_offset_base() : nullptr;
}

template <int HD, bool DROP>
void launch_fwd(const AttnParams& p, dim3 grid, hipStream_t s) {
#ifdef ATTN_AB_KSPL2
    constexpr int KSPL = HD == 32 ? 2 : 1;
#else
    constexpr int KSPL = 1;                                      // key-split pairs (2): measured slower at every batch size tried (r03 profile notes)
#endif
    launch_timed(RESEL_PROF_ATTN_FWD, attn_q_kernel<HD, 0, DROP, KSPL>, grid, dim3(256 * KSPL), 0, s, p);
}
template <int HD, bool DROP>
void launch_bwd(const AttnParams& p, const bf16_t* out, dim3 gq, dim3 gk, hipStream_t s) {
    hipLaunchKernelGGL(attn_delta_kernel<HD>, dim3((p.nqb * 128 * p.H + 255) / 256, p.S), dim3(256), 0, s, out, p.dout, p.lse, p.cu, p.slopes,
                       const_cast<float*>(p.delta), const_cast<float*>(p.comb), p.T, p.H, 1.f / (p.scale * RESEL_LOG2E));
    launch_timed(RESEL_PROF_ATTN_DQ, attn_q_kernel<HD, 1, DROP, 1>, gq, dim3(256), 0, s, p);
    launch_timed(RESEL_PROF_ATTN_DKV, attn_dkv_kernel<HD, DROP>, gk, dim3(256), 0, s, p);
}

}  // namespace

extern "C" int resel_attn_varlen_fwd(const uint16_t* qkv, const int32_t* cu_seqlens, const float* slopes, uint16_t* out, float* lse,
                                     void* workspace, int T, int S, int H, int hd, int max_seqlen, float scale,
                                     float p_drop, uint64_t seed, uint64_t offset, resel_stream_t stream) {
    if (!qkv || !cu_seqlens || !out || !lse || !attn_ok(T, S, H, hd, max_seqlen, p_drop) || !aligned16(qkv) || !aligned16(out)) return RESEL_EINVAL;
    AttnParams p{(const bf16_t*)qkv, cu_seqlens, slopes, (bf16_t*)out, lse, nullptr, nullptr, nullptr, nullptr, T, S, H, 0, nullptr, scale, 256u, 1.f, 0, 0};
    set_dropout(p, p_drop, seed, offset);
    p.nqb = (max_seqlen + 127) / 128;
    dim3 grid((H + 7) / 8 * 8 * S * p.nqb);
    hipStream_t s = (hipStream_t)stream;
    if (workspace && worklist_ok(S, max_seqlen)) {
        hipLaunchKernelGGL(attn_worklist_kernel, dim3(1), dim3(64), 0, s, cu_seqlens, S, p.nqb, (uint32_t*)workspace);
        p.wl = (const uint32_t*)workspace;
    }
    const bool drop = p_drop > 0.f;
    if (hd == 32) { if (drop) launch_fwd<32, true>(p, grid, s); else launch_fwd<32, false>(p, grid, s); }
    else          { if (drop) launch_fwd<64, true>(p, grid, s); else launch_fwd<64, false>(p, grid, s); }
    return launch_status();
}

extern "C" size_t resel_attn_varlen_fwd_workspace_bytes(int S, int max_seqlen) { return worklist_bytes(S, max_seqlen); }

extern "C" size_t resel_attn_varlen_bwd_workspace_bytes(int T, int S, int H, int hd, int max_seqlen) {
    (void)hd;
    return (size_t)2 * T * H * sizeof(float) + worklist_bytes(S, max_seqlen);          // -delta | comb | work list
}

extern "C" int resel_attn_varlen_bwd(const uint16_t* qkv, const int32_t* cu_seqlens, const float* slopes, const uint16_t* out,
                                     const float* lse, const uint16_t* dout, uint16_t* dqkv, void* workspace,
                                     int T, int S, int H, int hd, int max_seqlen, float scale,
                                     float p_drop, uint64_t seed, uint64_t offset, resel_stream_t stream) {
    if (!qkv || !cu_seqlens || !out || !lse || !dout || !dqkv || !workspace || !attn_ok(T, S, H, hd, max_seqlen, p_drop)) return RESEL_EINVAL;
    if (!aligned16(qkv) || !aligned16(out) || !aligned16(dout) || !aligned16(dqkv)) return RESEL_EINVAL;
    float* delta = (float*)workspace;
    AttnParams p{(const bf16_t*)qkv, cu_seqlens, slopes, nullptr, const_cast<float*>(lse), (const bf16_t*)dout, delta, delta + (size_t)T * H, (bf16_t*)dqkv, T, S, H, 0, nullptr, scale,
                 256u, 1.f, 0, 0};
    set_dropout(p, p_drop, seed, offset);
    hipStream_t s = (hipStream_t)stream;
    p.nqb = (max_seqlen + 127) / 128;
    dim3 gq((H + 7) / 8 * 8 * S * p.nqb), gk((H + 7) / 8 * 8 * S * p.nqb);
    if (worklist_ok(S, max_seqlen)) {
        uint32_t* wl = (uint32_t*)(delta + (size_t)2 * T * H);
        hipLaunchKernelGGL(attn_worklist_kernel, dim3(1), dim3(64), 0, s, cu_seqlens, S, p.nqb, wl);
        p.wl = wl;
    }
    const bool drop = p_drop > 0.f;
    if (hd == 32) { if (drop) launch_bwd<32, true>(p, (const bf16_t*)out, gq, gk, s); else launch_bwd<32, false>(p, (const bf16_t*)out, gq, gk, s); }
    else          { if (drop) launch_bwd<64, true>(p, (const bf16_t*)out, gq, gk, s); else launch_bwd<64, false>(p, (const bf16_t*)out, gq, gk, s); }
    return launch_status();
}
