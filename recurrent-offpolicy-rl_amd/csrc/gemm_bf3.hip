// fp32 GEMM on the bf16 matrix cores of gfx950, second edition: the operands are split ONCE PER BLOCK, on their way into LDS.
//
//     C[b][m][n] = epi( sum_k A[b](m, k) * B[b](n, k) )          same contract and operand layouts as gemm_f32.hip
//
// Product formation (resel_gemm_f32 modes 6 / 9, gemm_f32.hip 'SPLIT'): every fp32 operand element x is written exactly as
// x1 + x2 + x3, three bf16 values obtained by truncation; a bf16 x bf16 product is exact in fp32 and v_mfma_f32_32x32x16_bf16
// accumulates the plane products in fp32.  Mode 9 keeps all nine plane products, mode 6 drops a2 b3, a3 b2, a3 b3 (each at
// most 2^-24 |a b|).  Mode 3 ("bf16x3", what torch.set_float32_matmul_precision('high') names) keeps two planes per operand -
// 16 significant bits - and the products a1 b1 + a1 b2 + a2 b1: every dropped term is at most 2^-16 |a b|; half the matrix
// instructions and two thirds of the split / LDS work of mode 6.  The first edition kept fp32 tiles in LDS and every wave split the fragments it read: with 2 x 2 waves
// per block each element was split twice, 352 vector instructions per wave and K step beside 48 matrix instructions - the
// vector pipe, not the matrix pipe, set the pace (PMC: matrix pipe 50 % busy, profiles/r02_gemm.md).  Here
//   * the thread that LOADS an element splits it (22 vector instructions per four elements, once) and stores the three planes
//     to LDS as bf16; fragments are read from LDS as finished MFMA operands (ds_read_b128 = 8 k of one plane);
//   * the block tile is 256 x 128 (8 waves as 4 x 2, wave tile 64 x 64 = four 32 x 32 tiles): a thread splits 24 elements per
//     K step (132 vector instructions) against 48 matrix instructions of 32 cycles - inside their shadow;
//   * LDS image per plane: [row][32 k] bf16 = 64 bytes per row, no padding; the 16-byte chunk index is XOR-ed with bits 2..3
//     of the row and rows 4..7 of every 8 swap inside their pairs, which makes the fragment reads (ds_read_b128), the
//     [rows][K] stores (ds_write_b64) and the transposed [K][rows] stores conflict-free or 2-way at worst;
//     2 stages x (3 x 16 KB + 3 x 8 KB) = 144 KB: one block of 512 threads per CU, two waves per SIMD.
// Pipeline per K step s (one barrier): MFMAs of k slab 0 | split + LDS stores of step s + 1 (its global loads were issued a step
// earlier) | fragment reads of slab 1 | global loads of step s + 2 | first half of slab 1's MFMAs | barrier | fragment reads of
// step s + 1's slab 0 under the second half of slab 1's MFMAs.
// A third edition for mode 2 (producer / consumer waves, `gemm_ws_kernel` further down) takes the shapes with whole K steps; this one keeps the rest.
// Blocks are persistent (256 = one per CU) and walk items exactly as in gemm_f32.hip: whole tiles, and K slices for the last
// partly filled round and for weight gradients, summed by a fix-up kernel in a fixed order (deterministic, no atomics).
#include "resel_common.h"
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace {
using namespace resel;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int BM = 256, BN = 128, BK = 32;
constexpr int NTH = 512;
constexpr int GRID = 256;
constexpr int TILE = BM * BN;
constexpr int ROWB = 64;                         // bytes of one row of one plane: 32 k x bf16
constexpr int PLA = BM * ROWB, PLB = BN * ROWB;  // one plane of the A / B tile
constexpr int STAGE = 3 * PLA + 3 * PLB;         // 73 728 bytes

struct Params {
    const float *A, *B, *bias;
    float *C, *slab;
    int64_t lda, ldb, ldc, sA, sB, sC, sBias;
    int M, N, K;
    int act;
    int mt, nt;
    int nfull, nsplit, nsl, kslice;
    const float *amaxA, *amaxB;                  // mode 2: device scalars >= max |A|, max |B| (resel_amax); nullptr otherwise
    AmaxOut amaxC;                               // optional: publish max |C| (the values stored, after bias / activation / accumulate)
    int ntst;                                    // third edition: non-temporal C stores
    // fused epilogues of the third edition (resel_gemm_f32_dact / resel_gemm_f32_head, whole-K items only):
    //   act 4: C = product * elu'(Y) with Y the OUTPUT of the layer below (y > 0 ? 1 : y + 1); red = per-wave column sums of C
    //          [batch][2 mt][N] (the bias gradient of that layer, folded by colsum_kernel);
    //   act 5: C = a = elu(product + bias); red = per-wave row dots sum_n a[m][n] aux[z][n] over the wave's 64 columns [batch][2 nt][M]
    //          (the width-1 output layer of the critic head, folded by head_fold_kernel)
    const float* aux;
    int64_t ldaux, sAux;
    float* red;
    int redrows;                                 // act 4: partial rows per batch member in `red` (2 mt + one for the rows the host entry handles apart)
};

__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : fast_exp(x) - 1.f; }

__device__ __forceinline__ void tile_origin(const Params& p, int t, int& z, int& m0, int& n0) {
    const int ntile = p.mt * p.nt;
    z = t / ntile;
    const int tt = t - z * ntile;
    const int q = ntile / 8, r = ntile % 8, x = tt & 7, j = tt >> 3;
    const int bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    m0 = (bid / p.nt) * BM;
    n0 = (bid % p.nt) * BN;
}
struct Item { int m0, n0, z, kbeg, kend, split; };
__device__ __forceinline__ Item decode(const Params& p, int it) {
    Item o;
    int t = it;
    o.kbeg = 0; o.kend = p.K; o.split = 0;
    if (it >= p.nfull) {
        const int idx = it - p.nfull, tr = idx / p.nsl, sl = idx - tr * p.nsl;
        t = p.nfull + tr;
        o.kbeg = sl * p.kslice; o.kend = min(p.K, o.kbeg + p.kslice); o.split = idx + 1;
    }
    tile_origin(p, t, o.z, o.m0, o.n0);
    return o;
}

// byte offset of (row, 16-byte chunk c = k / 8) inside one plane
__device__ __forceinline__ int plane_off(int row, int c) {
    const int q = row >> 2;
    return ((row ^ (q & 1)) << 6) + ((c ^ (q & 3)) << 4);
}

// ---- four fp32 values -> three planes of four bf16 (8 bytes each).  NP = 2 (mode 3): the second plane is the residual ROUNDED to
// nearest even (v_cvt_pk_bf16_f32) instead of truncated - the dropped part is then unbiased and at most 2^-17 |x|
__device__ __forceinline__ uint32_t rne_pair(float lo, float hi) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
struct P3 { uint2 p1, p2, p3; };
template <int NP>
__device__ __forceinline__ P3 split4(const float4& v) {
#ifdef BF3_AB_NOSPLIT                  // ablation (wrong results): no vector work for the split
    return P3{make_uint2(__float_as_uint(v.x), __float_as_uint(v.y)), make_uint2(__float_as_uint(v.z), __float_as_uint(v.w)),
              make_uint2(__float_as_uint(v.x), __float_as_uint(v.w))};
#endif
    const float x[4] = {v.x, v.y, v.z, v.w};
    uint32_t w1[2], w2[2], w3[2] = {0u, 0u};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const uint32_t u0 = __float_as_uint(x[2 * q]), u1 = __float_as_uint(x[2 * q + 1]);
        w1[q] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);        // {hi16(x[2q+1]), hi16(x[2q])}: element 2q in the low half
        const float r0 = x[2 * q] - __uint_as_float(u0 & 0xffff0000u), r1 = x[2 * q + 1] - __uint_as_float(u1 & 0xffff0000u);
        if (NP == 2) {
            w2[q] = rne_pair(r0, r1);
        } else {
            const uint32_t v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
            w2[q] = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
            const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
            w3[q] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
        }
    }
    return P3{make_uint2(w1[0], w1[1]), make_uint2(w2[0], w2[1]), make_uint2(w3[0], w3[1])};
}
struct P3h { uint32_t p1, p2, p3; };                 // two values -> three planes of two bf16
template <int NP>
__device__ __forceinline__ P3h split2(float x0, float x1) {
    const uint32_t u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    P3h o;
    o.p1 = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
    if (NP == 2) {
        o.p2 = rne_pair(r0, r1);
        o.p3 = 0u;
        return o;
    }
    const uint32_t v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    o.p2 = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    o.p3 = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
    return o;
}

// ---- mode 2 ("f16x3"): fp16 planes of the SCALED operands, three plane products, one accumulator.  Each operand is multiplied by
// a power of two s that brings its largest magnitude into [2^14, 2^15) (resel_amax; exact), then
//   A (the wide-range operand: activations, gradients):  a1 = fp16(x s),  a2 = fp16(2^11 (x s - a1))        - two planes;
//   B (weights; the second operand of a weight gradient): b1 = fp16(x s),  b2 = fp16(x s - b1),  b1s = 2^-11 b1 - three planes;
//   C = (a1 b1 + a1 b2 + a2 b1s) / (sA sB):  the residuals are exact in fp32, every plane product is exact in fp32, the dropped
//   term a2 b2 is <= 2^-22 |a b|.  The 2^11 between the planes of one element sits on the A side for the a-residual (a2 is
//   normal down to |x| = 2^-29 max|A|) and on the B side for the b-residual (b2, b1s normal down to 2^-18 max|B|; below that
//   the element keeps 40 - d bits, d = log2(max|B| / |x|)) - a single fp16 accumulator scale cannot give both operands the
//   full range, two accumulator sets do not fit 256 registers beside the fragments (tried: 11-31 spilled dwords, slower than mode 6).
// sc = {s, 2048 s}.  7 vector instructions per element pair.
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <bool WIDE>
__device__ __forceinline__ void split_pair_f16(float x0, float x1, f32x2_t sc, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
#ifdef BF3_AB_NOSPLIT                  // ablation (wrong results)
    p1 = __float_as_uint(x0); p2 = __float_as_uint(x1); p3 = __float_as_uint(x0) ^ __float_as_uint(sc.x);
    return;
#endif
    const f32x2_t xs = {x0 * sc.x, x1 * sc.x};
    const f16x2_t h = __builtin_convertvector(xs, f16x2_t);
    p1 = __builtin_bit_cast(uint32_t, h);
    if (WIDE) {
        const f32x2_t r = {__builtin_fmaf((float)h.x, -2048.f, x0 * sc.y), __builtin_fmaf((float)h.y, -2048.f, x1 * sc.y)};
        p2 = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2_t));
        p3 = 0u;
    } else {
        const f32x2_t r = {xs.x - (float)h.x, xs.y - (float)h.y};
        p2 = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2_t));
        p3 = 0u;                                   // 2^-11 b1 is formed by the consumer of the fragment (mfma_lead_f16)
    }
}
template <bool WIDE>
__device__ __forceinline__ P3 split4_f16(const float4& v, f32x2_t sc) {
    P3 o;
    split_pair_f16<WIDE>(v.x, v.y, sc, o.p1.x, o.p2.x, o.p3.x);
    split_pair_f16<WIDE>(v.z, v.w, sc, o.p1.y, o.p2.y, o.p3.y);
    return o;
}
template <bool WIDE>
__device__ __forceinline__ P3h split2_f16(float x0, float x1, f32x2_t sc) {
    P3h o;
    split_pair_f16<WIDE>(x0, x1, sc, o.p1, o.p2, o.p3);
    return o;
}
// scale of an operand from its (upper bound of the) largest magnitude a: 2^(14 - floor(log2 a)), so a s is in [2^14, 2^15);
// a = 0 or tiny: a large finite scale (the operand is zero / negligible either way)
__device__ __forceinline__ float f16_scale(float amax) {
    const int f = (int)((__float_as_uint(amax) >> 23) & 0xffu);
    const int e = min(max(268 - f, 1), 240);         // 2048 s stays finite
    return __uint_as_float((uint32_t)e << 23);
}

// ---- one operand's share of a thread in a K step.
// KC (P[row][k]): NPC pieces, piece i = row (tid / 8 + 64 i), k = 4 (tid % 8) .. + 3 (one float4, 8 bytes per plane).
// !KC (P[k][row]), the 256-row operand: one 4 (k) x 4 (rows) patch, rows 4 g .., k = 4 k4 ..: four float4 along rows,
//   transposed in registers, four ds_write_b64 per plane.  tid = [g_lo:3][k4:3][g_hi:3]: a wave-load covers 8 k rows x 128 B.
// !KC, the 128-row operand: one 2 (k) x 4 (rows) patch, k = 2 k2 ..: two float4, four ds_write_b32 per plane.
//   tid = [g_lo:3][k2:4][g_hi:2].
// Rows beyond the operand's extent are read from row 0 (their products land in rows / columns that are never stored); k beyond
// the slice end contributes zeros (guarded loads on the last step of a K that is not a multiple of 32).
template <bool KC, int ROWS>
struct Src {
    static constexpr int NPC = ROWS / 64;          // KC pieces
    static constexpr bool is_kc = KC;
    static constexpr int NR = KC ? NPC : (ROWS == 256 ? 4 : 2);
    const char* base;
    const char* base0;                             // the tile's step at k = 0: always a whole, valid K step (K >= 32)
    uint32_t off[NR];
    uint32_t loff[KC ? NPC : 4];                   // LDS byte offsets inside a plane (KC: per piece; !KC: per patch row)
    int64_t step;
    int rl0, rlim, ld4, kofs4;                     // KC, third edition: this thread's row in piece 0, the last valid row of the tile (uniform),
                                                   // the row stride and this thread's k offset in bytes: offset of piece i = min(rl0 + 64 i, rlim) ld4 + kofs4
    int kofs;
    float4 r[NR];
    __device__ __forceinline__ void init(const float* P, int64_t ld, int rows, int r0, int k0, int tid) {
        if (KC) {
            kofs = 4 * (tid & 7);
            base = (const char*)(P + (int64_t)r0 * ld + k0);
            base0 = (const char*)(P + (int64_t)r0 * ld);
#pragma unroll
            for (int i = 0; i < NPC; ++i) {
                const int rl = (tid >> 3) + 64 * i;
                off[i] = (uint32_t)(((r0 + rl < rows ? rl : 0) * ld + kofs) * 4);
            }
            step = BK * 4;
            rl0 = tid >> 3;
            rlim = min(rows - 1 - r0, ROWS - 1);         // rows past the operand read its last row (their products are never stored)
            ld4 = (int)(ld * 4);
            kofs4 = kofs * 4;
        } else {
            const int g = ROWS == 256 ? (tid & 7) + 8 * (tid >> 6) : (tid & 7) + 8 * (tid >> 7);
            kofs = ROWS == 256 ? 4 * ((tid >> 3) & 7) : 2 * ((tid >> 3) & 15);
            base = (const char*)(P + (int64_t)k0 * ld + r0);
            base0 = (const char*)(P + r0);
            const int rl = 4 * g;
#pragma unroll
            for (int j = 0; j < NR; ++j) off[j] = (uint32_t)(((kofs + j) * ld + (r0 + rl < rows ? rl : 0)) * 4);
            step = (int64_t)BK * ld * 4;
        }
    }
    __device__ __forceinline__ void init_lds(int tid) {
        if (KC) {
#pragma unroll
            for (int i = 0; i < NPC; ++i) loff[i] = plane_off((tid >> 3) + 64 * i, (tid & 7) >> 1) + 8 * (tid & 1);
        } else if (ROWS == 256) {
            const int g = (tid & 7) + 8 * (tid >> 6), k4 = (tid >> 3) & 7;
#pragma unroll
            for (int j = 0; j < 4; ++j) loff[j] = plane_off(4 * g + j, k4 >> 1) + 8 * (k4 & 1);
        } else {
            const int g = (tid & 7) + 8 * (tid >> 7), k2 = (tid >> 3) & 15;
#pragma unroll
            for (int j = 0; j < 4; ++j) loff[j] = plane_off(4 * g + j, k2 >> 2) + 4 * (k2 & 3);
        }
    }
    // branch-free form for the scheduled region of the K loop: `full` (uniform) = a whole K step is due - load it and advance;
    // otherwise the same instructions re-read the tile's step at k = 0 (in bounds) and the caller's guarded `load` follows
    __device__ __forceinline__ void load_sched(bool full) {
#ifdef BF3_AB_NOLOAD                   // ablation (wrong results): the staging registers are never refreshed
        asm volatile("" : "+v"(r[0].x));
        return;
#endif
        const char* b = full ? base : base0;
#pragma unroll
        for (int i = 0; i < NR; ++i) r[i] = *reinterpret_cast<const float4*>(b + off[i]);
        if (full) base += step;
    }
    __device__ __forceinline__ void load(int k0, int kend) { load_to(r, k0, kend); }
    __device__ __forceinline__ void load_to(float4 (&r)[NR], int k0, int kend) {
#ifdef BF3_AB_NOLOAD
        asm volatile("" : "+v"(r[0].x));
        return;
#endif
        if (k0 + BK <= kend) {
#pragma unroll
            for (int i = 0; i < NR; ++i) r[i] = *reinterpret_cast<const float4*>(base + off[i]);
        } else {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k0 + kofs + (KC ? 0 : i) < kend) r[i] = *reinterpret_cast<const float4*>(base + off[i]);
            }
        }
        base += step;
    }
    // third edition (producer waves): one piece of a register set at a time, so that a register is reloaded for two K steps ahead
    // as soon as its values have been split (KC layouts; the transposed layouts need all NR registers of a patch together)
    __device__ __forceinline__ void load_piece(float4& v, int i, int k0, int kend) const {
#ifdef BF3_AB_NOLOAD
        asm volatile("" : "+v"(v.x));
        return;
#endif
        if (k0 + BK <= kend) {
            v = *reinterpret_cast<const float4*>(base + off[i]);
        } else {
            v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k0 + kofs + (KC ? 0 : i) < kend) v = *reinterpret_cast<const float4*>(base + off[i]);
        }
    }
    __device__ __forceinline__ void advance() { base += step; }
    __device__ __forceinline__ uint32_t piece_off(int i) const {        // two vector instructions per piece instead of a register per piece
        return __umul24((uint32_t)min(rl0 + 64 * i, rlim), (uint32_t)ld4) + (uint32_t)kofs4;
    }
    template <int PL, int NP, int F16>
    __device__ __forceinline__ void store_piece(const float4& v, int i, char* pl, f32x2_t sc) const {
        static_assert(KC, "pieces exist in the [rows][K] layout only");
        const P3 s = F16 == 1 ? split4_f16<true>(v, sc) : F16 == 2 ? split4_f16<false>(v, sc) : split4<NP>(v);
        *reinterpret_cast<uint2*>(pl + loff[i]) = s.p1;
        *reinterpret_cast<uint2*>(pl + PL + loff[i]) = s.p2;
        if (NP == 3) *reinterpret_cast<uint2*>(pl + 2 * PL + loff[i]) = s.p3;
    }
    // split the staged values and store the planes of this thread's pieces into the plane set at `pl` (plane stride PL bytes)
    template <int PL, int NP, int F16 = 0>
    __device__ __forceinline__ void store(char* pl, f32x2_t sc = f32x2_t{1.f, 2048.f}) const { store_from<PL, NP, F16>(r, pl, sc); }
    template <int PL, int NP, int F16 = 0>
    __device__ __forceinline__ void store_from(const float4 (&r)[NR], char* pl, f32x2_t sc = f32x2_t{1.f, 2048.f}) const {
        if (KC) {
#pragma unroll
            for (int i = 0; i < NPC; ++i) {
                const P3 s = F16 == 1 ? split4_f16<true>(r[i], sc) : F16 == 2 ? split4_f16<false>(r[i], sc) : split4<NP>(r[i]);
                *reinterpret_cast<uint2*>(pl + loff[i]) = s.p1;
                *reinterpret_cast<uint2*>(pl + PL + loff[i]) = s.p2;
                if (NP == 3) *reinterpret_cast<uint2*>(pl + 2 * PL + loff[i]) = s.p3;
            }
        } else if (ROWS == 256) {
            const float c[4][4] = {{r[0].x, r[1].x, r[2].x, r[3].x}, {r[0].y, r[1].y, r[2].y, r[3].y},
                                   {r[0].z, r[1].z, r[2].z, r[3].z}, {r[0].w, r[1].w, r[2].w, r[3].w}};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 cj = make_float4(c[j][0], c[j][1], c[j][2], c[j][3]);
                const P3 s = F16 == 1 ? split4_f16<true>(cj, sc) : F16 == 2 ? split4_f16<false>(cj, sc) : split4<NP>(cj);
                *reinterpret_cast<uint2*>(pl + loff[j]) = s.p1;
                *reinterpret_cast<uint2*>(pl + PL + loff[j]) = s.p2;
                if (NP == 3) *reinterpret_cast<uint2*>(pl + 2 * PL + loff[j]) = s.p3;
            }
        } else {
            const float c[4][2] = {{r[0].x, r[1].x}, {r[0].y, r[1].y}, {r[0].z, r[1].z}, {r[0].w, r[1].w}};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const P3h s = F16 == 1 ? split2_f16<true>(c[j][0], c[j][1], sc) : F16 == 2 ? split2_f16<false>(c[j][0], c[j][1], sc) : split2<NP>(c[j][0], c[j][1]);
                *reinterpret_cast<uint32_t*>(pl + loff[j]) = s.p1;
                *reinterpret_cast<uint32_t*>(pl + PL + loff[j]) = s.p2;
                if (NP == 3) *reinterpret_cast<uint32_t*>(pl + 2 * PL + loff[j]) = s.p3;
            }
        }
    }
};

struct Frag { bf16x8 a[3][2], b[3][2]; };            // [plane][32-row tile] of one k slab (16 k)

__device__ __forceinline__ bf16x8 lds16(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }
// fa / fb: this lane's fragment address of slab `s` in plane 0, tile 0 of the stage (byte pointers into LDS)
template <int NP>
__device__ __forceinline__ void read_a(Frag& f, const char* fa) {
#pragma unroll
    for (int pi = 0; pi < NP; ++pi)
#pragma unroll
        for (int t = 0; t < 2; ++t) f.a[pi][t] = lds16(fa + pi * PLA + t * 32 * ROWB);
}
template <int NP>
__device__ __forceinline__ void read_b(Frag& f, const char* fb) {
#pragma unroll
    for (int pi = 0; pi < NP; ++pi)
#pragma unroll
        for (int t = 0; t < 2; ++t) f.b[pi][t] = lds16(fb + pi * PLB + t * 32 * ROWB);
}
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int P, int Q>
__device__ __forceinline__ void mfma_term_f16(f32x16 (&acc)[2][2], const Frag& f) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#ifdef BF3_AB_NOMFMA
            asm volatile("" :: "v"(f.a[P][a]), "v"(f.b[Q][b]));
#else
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f.a[P][a]), __builtin_bit_cast(f16x8, f.b[Q][b]), acc[a][b], 0, 0, 0);
#endif
        }
}
template <int P, int Q>
__device__ __forceinline__ void mfma_term(f32x16 (&acc)[2][2], const Frag& f) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#ifdef BF3_AB_NOMFMA                   // ablation (wrong results): operands consumed, no matrix instruction
            asm volatile("" :: "v"(f.a[P][a]), "v"(f.b[Q][b]));
#elif defined(BF3_AB_M16)              // ablation (wrong results): the same matrix-pipe time as TWO v_mfma_f32_16x16x32_bf16 on accumulator quarters
            {
                typedef float f32x4v __attribute__((ext_vector_type(4)));
                constexpr int h = ((P + Q) & 1) * 2;
                f32x4v q0 = __builtin_shufflevector(acc[a][b], acc[a][b], 4 * h, 4 * h + 1, 4 * h + 2, 4 * h + 3);
                f32x4v q1 = __builtin_shufflevector(acc[a][b], acc[a][b], 4 * h + 4, 4 * h + 5, 4 * h + 6, 4 * h + 7);
                q0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a[P][a], f.b[Q][b], q0, 0, 0, 0);
                q1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a[P][a], f.b[Q][b], q1, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[a][b][4 * h + e] = q0[e]; acc[a][b][4 * h + 4 + e] = q1[e]; }
            }
#else
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[P][a], f.b[Q][b], acc[a][b], 0, 0, 0);
#endif
        }
}
// the small terms are added first
template <int SPLIT>
__device__ __forceinline__ void mfma_small(f32x16 (&acc)[2][2], const Frag& f) {
    if (SPLIT == 3 || SPLIT == 2) return;           // two planes per operand: a1 b1 + a1 b2 + a2 b1 only (mfma_lead)
    if (SPLIT == 9) { mfma_term<2, 2>(acc, f); mfma_term<2, 1>(acc, f); mfma_term<1, 2>(acc, f); }
    mfma_term<2, 0>(acc, f); mfma_term<0, 2>(acc, f); mfma_term<1, 1>(acc, f);
}
__device__ __forceinline__ void mfma_lead(f32x16 (&acc)[2][2], const Frag& f) {
    mfma_term<1, 0>(acc, f); mfma_term<0, 1>(acc, f); mfma_term<0, 0>(acc, f);
}
// mode 2: a2 b1s + a1 b2 + a1 b1 (A planes {a1, a2}, B planes {b1, b2, b1s}), small terms first
// the third B "plane" 2^-11 b1 never goes through LDS: four packed fp16 multiplies per fragment (exact unless the product is subnormal -
// the same values the producer's packed multiply stored before), a fifth less fragment traffic and a plane less to store
__device__ __forceinline__ void mfma_lead_f16(f32x16 (&acc)[2][2], Frag& f) {
    const f16x8 k = {(_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f,
                     (_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f};
#pragma unroll
    for (int t = 0; t < 2; ++t) f.b[2][t] = __builtin_bit_cast(bf16x8, __builtin_bit_cast(f16x8, f.b[0][t]) * k);
    mfma_term_f16<1, 2>(acc, f); mfma_term_f16<0, 1>(acc, f); mfma_term_f16<0, 0>(acc, f);
}

#define BF3_FENCE() __builtin_amdgcn_sched_barrier(0)
#ifdef BF3_AB_CLOCK                    // diagnostic build: in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz per block
__device__ unsigned long long g_bf3_clock[2 * GRID];
#endif

template <bool AKC, bool BKC, int SPLIT>
__global__ __launch_bounds__(NTH, 2) void gemm_bf3_kernel(Params p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];           // 2 stages
    constexpr bool F16 = SPLIT == 2;                                     // fp16 planes of the scaled operands (f16x3)
    constexpr int NP = (SPLIT == 3 || SPLIT == 2) ? 2 : 3;               // planes per element of A
    constexpr int NPB = (SPLIT == 3 || SPLIT == 2) ? 2 : 3;              // planes of B in LDS (mode 2: b1, b2; 2^-11 b1 is formed from b1's fragment)
    f32x2_t scA = {1.f, 2048.f}, scB = {1.f, 2048.f};
    float unscale = 1.f;
    if (F16) {
        const float sa_ = f16_scale(amax_read(p.amaxA)), sb_ = f16_scale(amax_read(p.amaxB));
        scA = f32x2_t{sa_, 2048.f * sa_};
        scB = f32x2_t{sb_, 2048.f * sb_};
        unscale = (1.f / sa_) * (1.f / sb_);                             // powers of two: exact
    }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
    const int li = lane & 31, lh = lane >> 5;
    const int total = p.nfull + p.nsplit * p.nsl;
    const int G = gridDim.x;
    if ((int)blockIdx.x >= total) return;

    Src<AKC, BM> sa;
    Src<BKC, BN> sb;
    sa.init_lds(tid);
    sb.init_lds(tid);
    int p_item = blockIdx.x, p_k0, p_kend;
    bool p_live = true;
    auto p_open = [&]() {
        const Item it = decode(p, p_item);
        sa.init(p.A + (int64_t)it.z * p.sA, p.lda, p.M, it.m0, it.kbeg, tid);
        sb.init(p.B + (int64_t)it.z * p.sB, p.ldb, p.N, it.n0, it.kbeg, tid);
        p_k0 = it.kbeg; p_kend = it.kend;
    };
    auto produce = [&]() {                          // global loads of the next K step in program order (also across items)
        if (!p_live) return;
        sa.load(p_k0, p_kend);
        sb.load(p_k0, p_kend);
        p_k0 += BK;
        if (p_k0 >= p_kend) {
            p_item += G;
            if (p_item < total) p_open(); else p_live = false;
        }
    };
    auto stage_store = [&](int st) {
        sa.template store<PLA, NP, F16 ? 1 : 0>(lds + st * STAGE, scA);
        sb.template store<PLB, NPB, F16 ? 2 : 0>(lds + st * STAGE + 3 * PLA, scB);
    };
    // fragment addresses: slab s of the lane = chunk 2 s + lh of row li of the wave's tile rows
    const char* fa[2];
    const char* fb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        fa[s] = lds + wm * ROWB + plane_off(li, 2 * s + lh);
        fb[s] = lds + 3 * PLA + wn * ROWB + plane_off(li, 2 * s + lh);
    }

#ifdef BF3_AB_STAGGER                     // experiment: BF3_AB_STAGGER groups of CUs start a fraction of a tile time apart (write bursts desynchronised)
    {
        const int grp = ((int)blockIdx.x >> 3) % BF3_AB_STAGGER;
        const long long wait = (long long)((p.K + BK - 1) / BK) * 5300 * grp / BF3_AB_STAGGER;
        const long long t0 = __builtin_amdgcn_s_memtime();
        while ((long long)__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
    }
#endif
#ifdef BF3_AB_CLOCK
    const unsigned long long ck_t0 = __builtin_amdgcn_s_memtime(), ck_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    float cmax = 0.f;                               // max |C| over everything this lane stores (published once, at the end)
    Frag f0, f1;
    p_open();
    produce();
    stage_store(0);
    produce();
    __syncthreads();
    read_a<NP>(f0, fa[0]); read_b<NPB>(f0, fb[0]);
    int cur_st = 0;
    for (int c_item = blockIdx.x; c_item < total; c_item += G) {
        const Item cur = decode(p, c_item);
        float zero = 0.f;
        asm volatile("" : "+v"(zero));              // opaque: or 64 registers of hoisted zeros stay live across the K loop
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] = zero;
        float bv[2] = {0.f, 0.f};

        for (int c_k0 = cur.kbeg; c_k0 < cur.kend; c_k0 += BK) {
            const int so = cur_st * STAGE, sn = (cur_st ^ 1) * STAGE;
            const bool fast = p_live && p_k0 + BK <= p_kend;
            if (c_k0 + BK >= cur.kend && p.bias && !cur.split) {        // requested most of a step before the epilogue needs them
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int n = cur.n0 + wn + 32 * b + li;
                    bv[b] = p.bias[(int64_t)cur.z * p.sBias + (n < p.N ? n : 0)];
                }
            }
            BF3_FENCE();
            // ---- region A: 36 (54) matrix instructions; in their shadow slab 1's fragment reads, the split + LDS stores of step
            // s + 1's tile and the global loads of step s + 2 into the registers the split has just released
            read_a<NP>(f1, fa[1] + so); read_b<NPB>(f1, fb[1] + so);
            mfma_small<SPLIT>(acc, f0);
            stage_store(cur_st ^ 1);
            if (F16) mfma_lead_f16(acc, f0); else mfma_lead(acc, f0);
            sa.load_sched(fast);
            sb.load_sched(fast);
            mfma_small<SPLIT>(acc, f1);
            if (SPLIT == 3 || SPLIT == 2) {         // 12 matrix instructions carry 8 (10) fragment reads, the split, 12 (14) LDS writes, the loads
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i < 2 * (NP + NPB)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    if (i >= 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
            } else {
                constexpr int NMA = SPLIT == 9 ? 60 : 36;
#pragma unroll
                for (int i = 0; i < NMA; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 // one MFMA
                    if (i < 12) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // one LDS read
                    __builtin_amdgcn_sched_group_barrier(0x002, SPLIT == 9 ? 3 : 5, 0);   // vector instructions
                    if (i >= 4 && i < 4 + 18) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // one LDS write
                    if (i >= NMA - 8) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);          // one global load
                }
            }
            BF3_FENCE();
#ifdef BF3_AB_NOBAR                    // ablation (wrong results)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // stage s + 1 is complete; every wave has read stage s
#endif
            BF3_FENCE();
            // ---- region B: the leading terms of slab 1 with the fragment reads of step s + 1's slab 0
            read_a<NP>(f0, fa[0] + sn); read_b<NPB>(f0, fb[0] + sn);
            if (F16) mfma_lead_f16(acc, f1); else mfma_lead(acc, f1);
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i < 2 * (NP + NPB)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            BF3_FENCE();
            if (fast) {
                p_k0 += BK;
                if (p_k0 >= p_kend) {
                    p_item += G;
                    if (p_item < total) p_open(); else p_live = false;
                }
            } else {
                produce();
            }
            BF3_FENCE();
            cur_st ^= 1;
        }
        if (F16) {                                   // C = (a1 b1 + a1 b2 + a2 b1s) / (sA sB): powers of two, exact
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[a][b][e] *= unscale;
        }
#ifdef BF3_AB_NOEPI                    // ablation (wrong results): one store per wave and tile keeps the accumulators alive
        if (lane == 0) p.C[(int64_t)cur.m0 * p.ldc + cur.n0 + w] = acc[0][0][0] + acc[0][1][1] + acc[1][0][2] + acc[1][1][3];
        continue;
#endif
        // epilogue: D layout col = lane & 31 (n), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (m).  One dword per lane and store:
        // 32 consecutive n of one row = a full 128-byte line (x 2 rows per instruction).  The transposed form (B fragment as the
        // first MFMA operand: four consecutive n per lane, 16 float4 stores instead of 64 dword stores) was measured SLOWER
        // (583 -> 618 us at 66 752 x 2048 x 384): a store costs per line touched (32 rows x 32 bytes per instruction there).
        if (cur.split) {
            float* o = p.slab + (int64_t)(cur.split - 1) * TILE + (wm + 4 * lh) * BN + wn + li;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[(32 * a + (e & 3) + 8 * (e >> 2)) * BN + 32 * b] = acc[a][b][e];
        } else {
            float* C = p.C + (int64_t)cur.z * p.sC;
            const bool full_m = cur.m0 + BM <= p.M;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int n = cur.n0 + wn + 32 * b + li;
                if (n >= p.N) continue;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    float v[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = acc[a][b][e] + bv[b];
                    if (p.act == 1) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) v[e] = elu1(v[e]);
                    }
                    if (p.act == 3) {                                // softplus (dt_proj of the Mamba mixer: the scan then reads delta itself)
#pragma unroll
                        for (int e = 0; e < 16; ++e) v[e] = softplus_nb(v[e]);
                    }
                    const int mb = cur.m0 + wm + 32 * a + 4 * lh;
                    float* crow = C + (int64_t)mb * p.ldc + n;
                    if (p.act == 2) {                                // C += product (the accumulating form of an input gradient)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int dm = (e & 3) + 8 * (e >> 2);
                            if (mb + dm < p.M) v[e] += crow[(int64_t)dm * p.ldc];
                        }
                    }
                    if (full_m) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) crow[(int64_t)((e & 3) + 8 * (e >> 2)) * p.ldc] = v[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int dm = (e & 3) + 8 * (e >> 2);
                            if (mb + dm < p.M) crow[(int64_t)dm * p.ldc] = v[e];
                        }
                    }
                    if (p.amaxC.slot) {              // rows past M repeat row 0's products: in range of the real data
#pragma unroll
                        for (int e = 0; e < 16; e += 2) cmax = fmaxf(cmax, fmaxf(__builtin_fabsf(v[e]), __builtin_fabsf(v[e + 1])));
                    }
                }
            }
        }
    }
    amax_publish_wave(cmax, p.amaxC);
#ifdef BF3_AB_CLOCK
    if (tid == 0) {
        g_bf3_clock[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - ck_t0;
        g_bf3_clock[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - ck_r0;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Third edition (mode 2, whole K steps): the same block tile, planes, LDS image and item walk, with the two jobs of a K step given to
// different waves.  In the second edition every wave loads, splits, stores planes, reads fragments, issues matrix instructions and stores
// C in program order - a wave parked on `vmcnt` issues no matrix instruction either - and its parts ADD (profiles/r04_gemm.md: data path
// 192 us + split 50 + C stores 113 + matrix instructions 158 = 513 against 455 measured at 66 752 x 2048 x 384).  Here, 512 threads:
//   * waves 0..3 (consumers, 2 x 2 over the 256 x 128 tile, wave tile 128 x 64 = 4 x 2 tiles of 32 x 32, 128 accumulator registers) only
//     read fragments (12 ds_read_b128 per 16-k slab, two register sets), issue matrix instructions and store C; they never wait for a
//     load.  2^-11 b1 - the third B "plane" of mode 2 - is formed from b1's fragment with four packed fp16 multiplies (the values
//     the producer's packed multiply used to store): LDS holds two planes per operand;
//   * waves 4..7 (producers, one per SIMD) hold two register sets = two K steps of the fp32 tiles; ALL their tile loads are inline
//     assembly the compiler does not count, and every unit (a piece of a [rows][K] operand, an instance of a transposed one) waits for
//     ITS loads with a hand-written `s_waitcnt vmcnt(24 - n)` tied to its registers through "+v" operands (24 loads in flight, the
//     unit's n are the oldest; foreign vector-memory operations only make a wait stricter), splits, stores its planes and reloads its
//     registers for the step two further on.  Each producer thread does the work of second-edition threads t and t + 256 (same
//     addresses, same LDS image).  What hipcc does to a fine-grained load / split / store pipeline written in plain C++ - branches
//     per piece, `vmcnt(0)` after every barrier or in front of every load, whole register sets copied at loop edges (= read while
//     their loads are in flight) - is in profiles/r04_gemm.md; hence: a SCALAR role test (two separate paths to the waitcnt pass),
//     native 128-bit vectors as asm operands, ONE path through the steady-state loop, the tail behind a drain;
//   * one `s_barrier` per K step for all eight waves: stage s + 1 written, stage s read;
//   * whole C tiles leave through a wave-private 5 KB LDS scratch (the third A plane of the stages, unused in mode 2) as 16-byte
//     stores of eight whole 128-byte lines: 32 store instructions per wave and tile instead of 128 (a wave may have 63 in flight).
// 243-245 registers, no scratch, two waves per SIMD, one block per CU, persistent.  Measured against the second edition (same box,
// us): fwd [T,384] -> 256 77 -> 67, [T,256] -> 256 55 -> 48, [T,256] -> 1024 171 -> 150, [T,384] -> 2048 420 -> 365.
constexpr int NTH_WS = 512;
constexpr int WS_LDS = 2 * STAGE;
#define WS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <bool AKC, bool BKC, int SPLIT, int EPI = 0>
__global__ __launch_bounds__(NTH_WS, 1) void gemm_ws_kernel(Params p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];           // 2 stages
    constexpr bool F16 = SPLIT == 2;
    constexpr int NP = (SPLIT == 3 || SPLIT == 2) ? 2 : 3;
    constexpr int NPB = (SPLIT == 3 || SPLIT == 2) ? 2 : 3;
    const int tid = threadIdx.x;
    const int total = p.nfull + p.nsplit * p.nsl;
    const int G = gridDim.x;
    if ((int)blockIdx.x >= total) return;
    float sa_ = 1.f, sb_ = 1.f;
    if (F16) { sa_ = f16_scale(amax_read(p.amaxA)); sb_ = f16_scale(amax_read(p.amaxB)); }
    asm volatile("" :: "v"(sa_), "v"(sb_));            // the scales have landed HERE: no compiler-counted load is pending inside the producers' loop

    if (__builtin_amdgcn_readfirstlane(tid) >= 256) {                   // a scalar condition: the two roles are separate paths to the compiler too
        // ------------------------------------------------------------------------------------------------ producers
        const int pt = tid - 256;
        const f32x2_t scA = {sa_, 2048.f * sa_}, scB = {sb_, 2048.f * sb_};
        typedef Src<AKC, BM> SA;
        typedef Src<BKC, BN> SB;
        SA a0, a1;
        SB b0, b1;
        a0.init_lds(pt); a1.init_lds(pt + 256);
        b0.init_lds(pt); b1.init_lds(pt + 256);
        // Register sets as native 128-bit values: they are operands of the inline assembly below.  ALL tile loads of this wave are inline
        // assembly the compiler does not count, and every unit (a piece of a [rows][K] operand, an instance of a transposed one) waits for
        // its own loads with a hand-written `s_waitcnt vmcnt(24 - n)` tied to its registers through "+v" operands (24 loads = two register
        // sets in flight; the n loads of the unit are the oldest), splits, stores its planes and reloads its registers for the step two
        // further on right away.  Issue order = unit order: a0, a1, b0, b1.  Whole K steps only (the host sends other shapes to edition 2).
        typedef float f32x4a __attribute__((ext_vector_type(4)));
        f32x4a ra[2][2][SA::NR], rb[2][2][SB::NR];                      // [register set][virtual thread]
        int nsteps = 0;
        for (int it = blockIdx.x; it < total; it += G) {
            const Item i = decode(p, it);
            nsteps += (i.kend - i.kbeg + BK - 1) / BK;
        }
        int p_item = blockIdx.x, p_k0 = 0, p_kend = 0;
        bool p_live = true;
        auto p_open = [&]() {
            const Item it = decode(p, p_item);
            const float* Ab = p.A + (int64_t)it.z * p.sA;
            const float* Bb = p.B + (int64_t)it.z * p.sB;
            a0.init(Ab, p.lda, p.M, it.m0, it.kbeg, pt); a1.init(Ab, p.lda, p.M, it.m0, it.kbeg, pt + 256);
            b0.init(Bb, p.ldb, p.N, it.n0, it.kbeg, pt); b1.init(Bb, p.ldb, p.N, it.n0, it.kbeg, pt + 256);
            p_k0 = it.kbeg; p_kend = it.kend;
        };
        auto advance = [&]() {
            a0.base += a0.step; a1.base += a1.step; b0.base += b0.step; b1.base += b1.step;
            p_k0 += BK;
            if (p_k0 >= p_kend) {
                p_item += G;
                if (p_item < total) p_open(); else p_live = false;
            }
        };
        auto aload = [&](f32x4a& r, const char* base, uint32_t off) {
#ifdef BF3_AB_NOLOAD
            asm volatile("" : "+v"(r) : "v"(off), "s"(base) : "memory");
#else
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(off), "s"(base) : "memory");
#endif
        };
        auto issue_set = [&](auto SET) {                                // prologue: a whole step into set S
            constexpr int S = decltype(SET)::value;
            auto all = [&](auto& src, auto& regs) {
                typedef std::remove_reference_t<decltype(src)> ST;
#pragma unroll
                for (int i = 0; i < ST::NR; ++i) {
                    if constexpr (ST::is_kc) aload(regs[i], src.base, src.piece_off(i));
                    else aload(regs[i], src.base, src.off[i]);
                }
            };
            all(a0, ra[S][0]); all(a1, ra[S][1]); all(b0, rb[S][0]); all(b1, rb[S][1]);
            advance();
        };
        constexpr int FA = F16 ? 1 : 0, FB = F16 ? 2 : 0;
        // MODE 0: steady state - two sets (24 loads) in flight, every unit waits for its own loads and reloads.  MODE 1: everything has
        // landed (tail): split and store only.  MODE 2: as 1, then reload the set and wait for it (a tail step that still has a successor).
        auto recycle = [&](auto SET, auto MODEC) {
            constexpr int S = decltype(SET)::value;
            constexpr int MODE = decltype(MODEC)::value;
            constexpr bool LOAD = MODE == 0;
            uint32_t st_off = S * STAGE;
            asm volatile("" : "+s"(st_off));               // opaque: the plane addresses are formed per piece (one add), not kept in eight registers across the loop
            char* st = lds + st_off;
            auto one = [&](auto& src, auto& regs, char* pl, f32x2_t sc, auto PLc, auto NPc, auto Fc) {
                constexpr int PL = decltype(PLc)::value, NPP = decltype(NPc)::value, FF = decltype(Fc)::value;
                typedef std::remove_reference_t<decltype(src)> ST;
                if constexpr (ST::is_kc) {
#pragma unroll
                    for (int i = 0; i < ST::NR; ++i) {
                        if (LOAD) asm volatile("s_waitcnt vmcnt(23)" : "+v"(regs[i]) :: "memory");
                        const float4 v = make_float4(regs[i].x, regs[i].y, regs[i].z, regs[i].w);
                        const P3 sp = FF == 1 ? split4_f16<true>(v, sc) : FF == 2 ? split4_f16<false>(v, sc) : split4<NPP>(v);
                        char* d = pl + src.loff[0] + 4096 * i;
                        *reinterpret_cast<uint2*>(d) = sp.p1;
                        *reinterpret_cast<uint2*>(d + PL) = sp.p2;
                        if (NPP == 3) *reinterpret_cast<uint2*>(d + 2 * PL) = sp.p3;
                        if (MODE != 1) aload(regs[i], src.base, src.piece_off(i));
                    }
                } else {
                    float4 t[ST::NR];
                    if constexpr (ST::NR == 4) { if (LOAD) asm volatile("s_waitcnt vmcnt(20)" : "+v"(regs[0]), "+v"(regs[1]), "+v"(regs[2]), "+v"(regs[3]) :: "memory"); }
                    else { if (LOAD) asm volatile("s_waitcnt vmcnt(22)" : "+v"(regs[0]), "+v"(regs[1]) :: "memory"); }
#pragma unroll
                    for (int i = 0; i < ST::NR; ++i) t[i] = make_float4(regs[i].x, regs[i].y, regs[i].z, regs[i].w);
                    src.template store_from<PL, NPP, FF>(t, pl, sc);
                    if (MODE != 1) {
#pragma unroll
                        for (int i = 0; i < ST::NR; ++i) aload(regs[i], src.base, src.off[i]);
                    }
                }
            };
            using IA = std::integral_constant<int, PLA>;
            using IB = std::integral_constant<int, PLB>;
            one(a0, ra[S][0], st, scA, IA{}, std::integral_constant<int, NP>{}, std::integral_constant<int, FA>{});
            one(a1, ra[S][1], st, scA, IA{}, std::integral_constant<int, NP>{}, std::integral_constant<int, FA>{});
            one(b0, rb[S][0], st + 3 * PLA, scB, IB{}, std::integral_constant<int, NPB>{}, std::integral_constant<int, FB>{});
            one(b1, rb[S][1], st + 3 * PLA, scB, IB{}, std::integral_constant<int, NPB>{}, std::integral_constant<int, FB>{});
            if (MODE != 1) advance();
        };
        auto landed = [&]() {                            // every load issued so far has landed, in every register of both sets
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int v = 0; v < 2; ++v) {
#pragma unroll
                    for (int i = 0; i < SA::NR; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[q][v][i]) :: "memory");
#pragma unroll
                    for (int i = 0; i < SB::NR; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rb[q][v][i]) :: "memory");
                }
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        using M0 = std::integral_constant<int, 0>;
        using M1 = std::integral_constant<int, 1>;
        using M2 = std::integral_constant<int, 2>;
        p_open();
        issue_set(S0{});                                // step 0
        if (nsteps > 1) issue_set(S1{});                // step 1
        int j = 0;
        // ONE path through the steady state (pairs of steps that both reload): the loop-carried registers keep their places - with
        // the tail's variants inside this loop hipcc copied whole register sets at the loop edges, i.e. read registers still in flight
        for (; j + 3 < nsteps; j += 2) {                // step j -> stage j & 1 from set j & 1, one K step ahead of the consumers
            recycle(S0{}, M0{});
            WS_BARRIER();
            recycle(S1{}, M0{});
            WS_BARRIER();
        }
        landed();
        for (; j < nsteps; ++j) {                       // at most three steps: synchronous
            const bool more = j + 2 < nsteps;
            if (j & 1) { if (more) recycle(S1{}, M2{}); else recycle(S1{}, M1{}); }
            else { if (more) recycle(S0{}, M2{}); else recycle(S0{}, M1{}); }
            if (more) landed();
            WS_BARRIER();
        }
        WS_BARRIER();
        return;
    }

    // ---------------------------------------------------------------------------------------------------- consumers
    // wave w: rows 128 (w >> 1) .. + 127, columns 64 (w & 1) .. + 63 of the 256 x 128 block tile = 4 x 2 tiles of 32 x 32 = 128 accumulator
    // registers; fragments of a 16-k slab: A 2 planes x 4 tiles + B 2 planes x 2 tiles = 12 ds_read_b128 (48 registers), two sets
    static_assert(SPLIT == 2, "the 4 + 4 wave form exists for mode 2");
    const float unscale = (1.f / sa_) * (1.f / sb_);
    const int lane = tid & 63, w = tid >> 6;
    const int wm = (w >> 1) * 128, wn = (w & 1) * 64;
    const int li = lane & 31, lh = lane >> 5;
    const char* fa[2];
    const char* fb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        fa[s] = lds + wm * ROWB + plane_off(li, 2 * s + lh);
        fb[s] = lds + 3 * PLA + wn * ROWB + plane_off(li, 2 * s + lh);
    }
    struct F4 { f16x8 a[2][4], b[2][2]; };
    auto rd = [&](F4& f, const char* pa, const char* pb) {
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
#pragma unroll
            for (int t = 0; t < 4; ++t) f.a[pi][t] = *reinterpret_cast<const f16x8*>(pa + pi * PLA + t * 32 * ROWB);
#pragma unroll
            for (int t = 0; t < 2; ++t) f.b[pi][t] = *reinterpret_cast<const f16x8*>(pb + pi * PLB + t * 32 * ROWB);
        }
    };
    f32x16 acc[4][2];
    auto mm = [&](const F4& f) {                    // a2 (2^-11 b1) + a1 b2 + a1 b1, the small terms first
        const f16x8 k = {(_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f,
                         (_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f};
        const f16x8 bs0 = f.b[0][0] * k, bs1 = f.b[0][1] * k;
#ifdef BF3_AB_NOMFMA
#pragma unroll
        for (int a = 0; a < 4; ++a) asm volatile("" :: "v"(f.a[0][a]), "v"(f.a[1][a]), "v"(bs0), "v"(bs1), "v"(f.b[0][0]), "v"(f.b[0][1]), "v"(f.b[1][0]), "v"(f.b[1][1]));
        return;
#endif
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[1][a], bs0, acc[a][0], 0, 0, 0);
            acc[a][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[1][a], bs1, acc[a][1], 0, 0, 0);
        }
#pragma unroll
        for (int q = 1; q >= 0; --q)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[0][a], f.b[q][0], acc[a][0], 0, 0, 0);
                acc[a][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[0][a], f.b[q][1], acc[a][1], 0, 0, 0);
            }
    };
    float cmax = 0.f;
    F4 f0, f1;
    WS_BARRIER();
    rd(f0, fa[0], fb[0]);
    int cur_st = 0;
    // EPI 4 keeps four tiles of Y in flight through its epilogue (64 registers): the first fragments of the NEXT item are then read behind
    // the epilogue instead of under the item's last matrix instructions (their stage stays valid until this wave passes the next barrier)
    constexpr bool DEFER_F0 = EPI == 4;
    for (int c_item = blockIdx.x; c_item < total; c_item += G) {
        const Item cur = decode(p, c_item);
        if (DEFER_F0 && c_item != (int)blockIdx.x) rd(f0, fa[0] + cur_st * STAGE, fb[0] + cur_st * STAGE);
        float zero = 0.f;
        asm volatile("" : "+v"(zero));
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] = zero;
        float bv[2] = {0.f, 0.f};
        for (int c_k0 = cur.kbeg; c_k0 < cur.kend; c_k0 += BK) {
            const int so = cur_st * STAGE, sn = (cur_st ^ 1) * STAGE;
            if (c_k0 + BK >= cur.kend && p.bias && !cur.split) {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int n = cur.n0 + wn + 32 * b + li;
                    bv[b] = p.bias[(int64_t)cur.z * p.sBias + (n < p.N ? n : 0)];
                }
            }
            BF3_FENCE();
            rd(f1, fa[1] + so, fb[1] + so);
            mm(f0);
#pragma unroll
            for (int i = 0; i < 12; ++i) {              // one fragment read behind each of the first matrix instructions
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            BF3_FENCE();
            WS_BARRIER();
            BF3_FENCE();
            if (!(DEFER_F0 && c_k0 + BK >= cur.kend)) rd(f0, fa[0] + sn, fb[0] + sn);
            mm(f1);
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            BF3_FENCE();
            cur_st ^= 1;
        }
#ifdef BF3_AB_NOEPI
        if (lane == 0) p.C[(int64_t)cur.m0 * p.ldc + cur.n0 + w] = acc[0][0][0] + acc[1][1][1] + acc[2][0][2] + acc[3][1][3];
        continue;
#endif
        // epilogue (as the second edition's, four row tiles)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] *= unscale;
        if (cur.split) {
            float* o = p.slab + (int64_t)(cur.split - 1) * TILE + (wm + 4 * lh) * BN + wn + li;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[(32 * a + (e & 3) + 8 * (e >> 2)) * BN + 32 * b] = acc[a][b][e];
            continue;
        }
        if constexpr (EPI == 0) {                                    // the plain epilogue, exactly as before the fused forms existed
            float* C = p.C + (int64_t)cur.z * p.sC;
            const bool full_m = cur.m0 + BM <= p.M;
            if (full_m && cur.n0 + BN <= p.N && p.act != 2) {
                // whole tiles: every 32 x 32 tile is turned through a wave-private 5 KB LDS scratch (rows of 40 words: the two half-waves of a
                // dword write hit disjoint banks) and leaves as FOUR 16-byte-per-lane stores of eight whole 128-byte lines each - 32 store
                // instructions per wave and tile instead of 128: a wave may have 63 vector-memory operations in flight, and with four
                // consumer waves 128 dword stores each stalled on that limit (160 of 415 us at 66 752 x 2048 x 384).
                // Scratch: the third A plane of the stages, which mode 2 does not use (waves 0, 1 in stage 0's, 2, 3 in stage 1's).
                char* sc = lds + (w >> 1) * STAGE + 2 * PLA + (w & 1) * 5120;
                float* wr = reinterpret_cast<float*>(sc) + (4 * lh) * 40 + li;
                const float4* rdp = reinterpret_cast<const float4*>(sc + (lane >> 3) * 160 + (lane & 7) * 16);
#pragma unroll
                for (int b = 0; b < 2; ++b) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        float v[16];
#pragma unroll
                        for (int e = 0; e < 16; ++e) v[e] = acc[a][b][e] + bv[b];
                        if (p.act == 1) {
#pragma unroll
                            for (int e = 0; e < 16; ++e) v[e] = elu1(v[e]);
                        }
                        if (p.act == 3) {
#pragma unroll
                            for (int e = 0; e < 16; ++e) v[e] = softplus_nb(v[e]);
                        }
                        if (p.amaxC.slot) {
#pragma unroll
                            for (int e = 0; e < 16; e += 2) cmax = fmaxf(cmax, fmaxf(__builtin_fabsf(v[e]), __builtin_fabsf(v[e + 1])));
                        }
#pragma unroll
                        for (int e = 0; e < 16; ++e) wr[((e & 3) + 8 * (e >> 2)) * 40] = v[e];
                        float* q = C + (int64_t)(cur.m0 + wm + 32 * a + (lane >> 3)) * p.ldc + cur.n0 + wn + 32 * b + 4 * (lane & 7);
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const float4 t = rdp[g * 8 * 10];                   // rows 8 g + (lane >> 3): 8 rows x 160 bytes = 80 float4
                            if (p.ntst) {                                       // streaming stores (RESEL_GEMM_NT): -4 % on the kernel alone at N = 2048
                                typedef float f4v __attribute__((ext_vector_type(4)));
                                __builtin_nontemporal_store(f4v{t.x, t.y, t.z, t.w}, reinterpret_cast<f4v*>(q + (int64_t)(8 * g) * p.ldc));
                            } else {
                                *reinterpret_cast<float4*>(q + (int64_t)(8 * g) * p.ldc) = t;
                            }
                        }
                    }
                }
                continue;
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int n = cur.n0 + wn + 32 * b + li;
                if (n >= p.N) continue;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    float v[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = acc[a][b][e] + bv[b];
                    if (p.act == 1) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) v[e] = elu1(v[e]);
                    }
                    if (p.act == 3) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) v[e] = softplus_nb(v[e]);
                    }
                    const int mb = cur.m0 + wm + 32 * a + 4 * lh;
                    float* crow = C + (int64_t)mb * p.ldc + n;
                    if (p.act == 2) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int dm = (e & 3) + 8 * (e >> 2);
                            if (mb + dm < p.M) v[e] += crow[(int64_t)dm * p.ldc];
                        }
                    }
                    if (full_m) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) crow[(int64_t)((e & 3) + 8 * (e >> 2)) * p.ldc] = v[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int dm = (e & 3) + 8 * (e >> 2);
                            if (mb + dm < p.M) crow[(int64_t)dm * p.ldc] = v[e];
                        }
                    }
                    if (p.amaxC.slot) {
#pragma unroll
                        for (int e = 0; e < 16; e += 2) cmax = fmaxf(cmax, fmaxf(__builtin_fabsf(v[e]), __builtin_fabsf(v[e + 1])));
                    }
                }
            }
        } else {
            float* C = p.C + (int64_t)cur.z * p.sC;
            const bool full_m = cur.m0 + BM <= p.M;
            constexpr bool DACT = EPI == 4, HEAD = EPI == 5;               // fused epilogues: Params::aux / red
            const int act = EPI ? (HEAD ? 1 : 0) : p.act;
            if constexpr (DACT) {
                // Whole tiles only (the host entry sends the rows past the last whole 256-row tile elsewhere).  Two phases:
                //   A  the Y values are loaded IN THE ACCUMULATOR LAYOUT (column = li of block b, rows 4 lh + (e & 3) + 8 (e >> 2): a dword load
                //      covers two whole 128-byte lines) and multiplied into the accumulators in place, tile by tile with PF tiles of loads in
                //      flight; column sums and the magnitude come from the same registers;
                //   B  the tiles leave exactly as in the plain epilogue (LDS turn, 16-byte full-line non-temporal stores).
                // No store is issued between the loads of an item: a wait for a load also waits for every OLDER store (one in-order counter
                // per wave), and with loads and stores interleaved per tile every tile waited for the write acknowledgements of the tiles
                // before it (measured: 650 us against 305 for the product alone; this form: see profiles/r05_gemm.md).  The loads are inline
                // assembly with a scalar base + ONE per-lane offset register (the compiler formed a 64-bit address pair per load and spilled
                // accumulators; every scratch reload is a vmcnt(0)); it does not count them: each tile waits for ITS sixteen loads with a
                // hand-written vmcnt = the loads issued since (foreign vector-memory operations only make a wait stricter).
                constexpr int PF = 3;
                // per-lane constants are formed HERE from the lane id (v_mbcnt) and the wave id in a scalar register: kept across the K loop
                // they were spilled
                const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
                const int li = lane & 31, lh = lane >> 5;
                const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
                const int wm = (w >> 1) * 128, wn = (w & 1) * 64;
                typedef float f32x4a __attribute__((ext_vector_type(4)));
                f32x4a yv[8][4];
                // Y travels as 16-byte loads of whole 128-byte lines (a lane: rows 8 g + (lane >> 3), columns 4 (lane & 7) .. + 3 - the form the C
                // tiles leave in) and is turned INTO the accumulator layout through the wave's LDS scratch: dword loads in the accumulator
                // layout cost four times the address work per byte (128 load instructions per item and wave; measured 542 us, this form: r05_gemm.md)
                char* sc = lds + (w >> 1) * STAGE + 2 * PLA + (w & 1) * 5120;
                float* wr = reinterpret_cast<float*>(sc) + (4 * lh) * 40 + li;
                float4* rdp = reinterpret_cast<float4*>(sc + (lane >> 3) * 160 + (lane & 7) * 16);
                const uint32_t ylane = (uint32_t)(((lane >> 3) * (int)p.ldaux + 4 * (lane & 7)) * 4);
                const float* ybase = p.aux + (int64_t)cur.z * p.sAux + (int64_t)(cur.m0 + wm) * p.ldaux + cur.n0 + wn;
                auto yld = [](f32x4a& r, uint32_t off, const float* base) {        // (asm operands inside the generic lambdas below are not captured: a clang quirk)
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(off), "s"(base) : "memory");
                };
                auto ywait = [](f32x4a& r0, f32x4a& r1, f32x4a& r2, f32x4a& r3, auto NC) {
                    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "n"(decltype(NC)::value) : "memory");
                };
                auto yload = [&](auto TC) {
                    constexpr int t = decltype(TC)::value;
                    if constexpr (t < 8) {
                        const float* yb = ybase + (int64_t)(32 * (t >> 1)) * p.ldaux + 32 * (t & 1);
#pragma unroll
                        for (int g = 0; g < 4; ++g) yld(yv[t][g], ylane, yb + (int64_t)(8 * g) * p.ldaux);
                    }
                };
                float cs[2] = {0.f, 0.f}, cmx = 0.f;
                auto mul = [&](auto TC) {
                    constexpr int t = decltype(TC)::value;
                    constexpr int a = t >> 1, b = t & 1;
                    yload(std::integral_constant<int, t + PF>{});
                    ywait(yv[t][0], yv[t][1], yv[t][2], yv[t][3], std::integral_constant<int, 4 * ((t + PF < 7 ? t + PF : 7) - t)>{});
#pragma unroll
                    for (int g = 0; g < 4; ++g) rdp[g * 8 * 10] = make_float4(yv[t][g].x, yv[t][g].y, yv[t][g].z, yv[t][g].w);
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float y = wr[((e & 3) + 8 * (e >> 2)) * 40];
                        const float v = acc[a][b][e] * (y > 0.f ? 1.f : y + 1.f);
                        acc[a][b][e] = v;
                        cs[b] += v;
                        cmx = fmaxf(cmx, __builtin_fabsf(v));
                    }
                };
                yload(std::integral_constant<int, 0>{}); yload(std::integral_constant<int, 1>{}); yload(std::integral_constant<int, 2>{});
                static_assert(PF == 3, "the prologue issues PF tiles");
                mul(std::integral_constant<int, 0>{}); mul(std::integral_constant<int, 1>{}); mul(std::integral_constant<int, 2>{});
                mul(std::integral_constant<int, 3>{}); mul(std::integral_constant<int, 4>{}); mul(std::integral_constant<int, 5>{});
                mul(std::integral_constant<int, 6>{}); mul(std::integral_constant<int, 7>{});
                if (p.red) {
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const float c = cs[b] + __shfl_xor(cs[b], 32, 64);       // the two row halves of the accumulator layout
                        if (lh == 0) p.red[((int64_t)cur.z * p.redrows + 2 * (cur.m0 / BM) + (w >> 1)) * p.N + cur.n0 + wn + 32 * b + li] = c;
                    }
                }
                amax_publish_wave(cmx, p.amaxC);                         // one conditional atomic per wave and item (a maximum carried across the K loops was spilled)
                // phase B
#pragma unroll
                for (int b = 0; b < 2; ++b) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) wr[((e & 3) + 8 * (e >> 2)) * 40] = acc[a][b][e];
                        float* q = C + (int64_t)(cur.m0 + wm + 32 * a + (lane >> 3)) * p.ldc + cur.n0 + wn + 32 * b + 4 * (lane & 7);
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const float4 t = rdp[g * 8 * 10];
                            typedef float f4v __attribute__((ext_vector_type(4)));
                            __builtin_nontemporal_store(f4v{t.x, t.y, t.z, t.w}, reinterpret_cast<f4v*>(q + (int64_t)(8 * g) * p.ldc));
                        }
                    }
                }
                continue;
            }
            if (full_m && cur.n0 + BN <= p.N && act != 2) {
                // whole tiles: every 32 x 32 tile is turned through a wave-private 5 KB LDS scratch (rows of 40 words: the two half-waves of a
                // dword write hit disjoint banks) and leaves as FOUR 16-byte-per-lane stores of eight whole 128-byte lines each - 32 store
                // instructions per wave and tile instead of 128: a wave may have 63 vector-memory operations in flight, and with four
                // consumer waves 128 dword stores each stalled on that limit (160 of 415 us at 66 752 x 2048 x 384).
                // Scratch: the third A plane of the stages, which mode 2 does not use (waves 0, 1 in stage 0's, 2, 3 in stage 1's).
                char* sc = lds + (w >> 1) * STAGE + 2 * PLA + (w & 1) * 5120;
                float* wr = reinterpret_cast<float*>(sc) + (4 * lh) * 40 + li;
                const float4* rdp = reinterpret_cast<const float4*>(sc + (lane >> 3) * 160 + (lane & 7) * 16);
                // one 32 x 32 tile (row tile a, column block b).  In the transposed form a lane holds rows 8 g + (lane >> 3), columns 4 (lane & 7) .. + 3.
                // DACT: yv = the four float4 of Y at those places (requested before the tile is turned), cs += the lane's share of the column sums;
                // HEAD: w3 = the output layer's weights of the lane's four columns, qr[g] += the lane's share of the row dots
                auto tile = [&](int a, int b, const float4 (&yv)[4], float4& cs, const float4& w3, float (&qr)[4]) {
                    float v[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = acc[a][b][e] + bv[b];
                    if (act == 1) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) v[e] = elu1(v[e]);
                    }
                    if (act == 3) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) v[e] = softplus_nb(v[e]);
                    }
                    if (p.amaxC.slot && !DACT) {
#pragma unroll
                        for (int e = 0; e < 16; e += 2) cmax = fmaxf(cmax, fmaxf(__builtin_fabsf(v[e]), __builtin_fabsf(v[e + 1])));
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) wr[((e & 3) + 8 * (e >> 2)) * 40] = v[e];
                    float* q = C + (int64_t)(cur.m0 + wm + 32 * a + (lane >> 3)) * p.ldc + cur.n0 + wn + 32 * b + 4 * (lane & 7);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float4 t = rdp[g * 8 * 10];                         // rows 8 g + (lane >> 3): 8 rows x 160 bytes = 80 float4
                        if (DACT) {
                            t.x *= yv[g].x > 0.f ? 1.f : yv[g].x + 1.f; t.y *= yv[g].y > 0.f ? 1.f : yv[g].y + 1.f;
                            t.z *= yv[g].z > 0.f ? 1.f : yv[g].z + 1.f; t.w *= yv[g].w > 0.f ? 1.f : yv[g].w + 1.f;
                            cs.x += t.x; cs.y += t.y; cs.z += t.z; cs.w += t.w;
                            if (p.amaxC.slot) cmax = amax4(cmax, t);
                        }
                        if (HEAD) qr[g] += (t.x * w3.x + t.y * w3.y) + (t.z * w3.z + t.w * w3.w);
                        if (p.ntst) {                                       // streaming stores (RESEL_GEMM_NT): -4 % on the kernel alone at N = 2048
                            typedef float f4v __attribute__((ext_vector_type(4)));
                            __builtin_nontemporal_store(f4v{t.x, t.y, t.z, t.w}, reinterpret_cast<f4v*>(q + (int64_t)(8 * g) * p.ldc));
                        } else {
                            *reinterpret_cast<float4*>(q + (int64_t)(8 * g) * p.ldc) = t;
                        }
                    }
                };
                if constexpr (EPI == 0) {
                    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    const float4 nov[4] = {z4, z4, z4, z4};
                    float4 ncs = z4;
                    float nq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int a = 0; a < 4; ++a) tile(a, b, nov, ncs, z4, nq);
                } else {
                    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    {
                        // row tiles outermost: the row dots of a tile's 32 rows complete after its two column blocks (4 registers instead of 16)
                        const float4 nov[4] = {z4, z4, z4, z4};
                        float4 ncs = z4, w3[2];
#pragma unroll
                        for (int b = 0; b < 2; ++b) w3[b] = ld4(p.aux + (int64_t)cur.z * p.sAux + cur.n0 + wn + 32 * b + 4 * (lane & 7));
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
                            float qr[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int b = 0; b < 2; ++b) tile(a, b, nov, ncs, w3[b], qr);
                            // fold the eight column groups (lane bits 0..2): lanes with (lane & 7) == 0 hold a row's dot
                            float* qp = p.red + ((int64_t)cur.z * 2 * p.nt + 2 * (cur.n0 / BN) + (w & 1)) * p.M + cur.m0 + wm + 32 * a + (lane >> 3);
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                float r = qr[g];
                                r += __shfl_xor(r, 1, 64); r += __shfl_xor(r, 2, 64); r += __shfl_xor(r, 4, 64);
                                if ((lane & 7) == 0) qp[8 * g] = r;
                            }
                        }
                    }
                }
                continue;
            }
            // edge tiles (and the accumulating form): the accumulator layout goes out as it is - column n = li of block b, rows 4 lh + (e & 3) + 8 (e >> 2)
            float cs1[2] = {0.f, 0.f};
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                float qe[16];
                if (HEAD) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) qe[e] = 0.f;
                }
                const int mb = cur.m0 + wm + 32 * a + 4 * lh;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int n = cur.n0 + wn + 32 * b + li;
                    if (n >= p.N) continue;
                    float v[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = acc[a][b][e] + bv[b];
                    if (act == 1) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) v[e] = elu1(v[e]);
                    }
                    if (act == 3) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) v[e] = softplus_nb(v[e]);
                    }
                    if (DACT) {
                        const float* yrow = p.aux + (int64_t)cur.z * p.sAux + (int64_t)mb * p.ldaux + n;
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int dm = (e & 3) + 8 * (e >> 2);
                            float y = 0.f;
                            if (mb + dm < p.M) y = yrow[(int64_t)dm * p.ldaux]; else v[e] = 0.f;      // rows past M hold row 0's products: not part of the column sums
                            v[e] *= y > 0.f ? 1.f : y + 1.f;
                            cs1[b] += v[e];
                        }
                    }
                    if (HEAD) {
                        const float w3 = p.aux[(int64_t)cur.z * p.sAux + n];
#pragma unroll
                        for (int e = 0; e < 16; ++e) qe[e] += v[e] * w3;
                    }
                    float* crow = C + (int64_t)mb * p.ldc + n;
                    if (act == 2) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int dm = (e & 3) + 8 * (e >> 2);
                            if (mb + dm < p.M) v[e] += crow[(int64_t)dm * p.ldc];
                        }
                    }
                    if (full_m) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) crow[(int64_t)((e & 3) + 8 * (e >> 2)) * p.ldc] = v[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int dm = (e & 3) + 8 * (e >> 2);
                            if (mb + dm < p.M) crow[(int64_t)dm * p.ldc] = v[e];
                        }
                    }
                    if (p.amaxC.slot) {
#pragma unroll
                        for (int e = 0; e < 16; e += 2) cmax = fmaxf(cmax, fmaxf(__builtin_fabsf(v[e]), __builtin_fabsf(v[e + 1])));
                    }
                }
                if (HEAD) {                                                 // a row's 32 columns of a block sit in the 32 lanes of a half-wave
                    float* qp = p.red + ((int64_t)cur.z * 2 * p.nt + 2 * (cur.n0 / BN) + (w & 1)) * p.M;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float r = qe[e];
#pragma unroll
                        for (int o = 1; o < 32; o <<= 1) r += __shfl_xor(r, o, 64);
                        const int m = mb + (e & 3) + 8 * (e >> 2);
                        if (li == 0 && m < p.M) qp[m] = r;
                    }
                }
            }
            if (DACT && p.red) {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int n = cur.n0 + wn + 32 * b + li;
                    const float c = cs1[b] + __shfl_xor(cs1[b], 32, 64);     // the two row halves of the accumulator layout
                    if (lh == 0 && n < p.N) p.red[((int64_t)cur.z * 2 * p.mt + 2 * (cur.m0 / BM) + (w >> 1)) * p.N + n] = c;
                }
            }
        }
    }
    amax_publish_wave(cmax, p.amaxC);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Fourth edition (experimental, RESEL_GEMM_EDITION=4; mode 2, A [rows][K], whole K steps, one batch member, no K slices): a 256 x 256
// BLOCK TILE - a third fewer L2 -> CU bytes per product than 256 x 128 (profiles/r04_gemm.md: the K loop is byte-bound).  The 1 024
// accumulator registers per lane of such a tile do not fit the producer / consumer form (four consumer waves x 256 + fragments), so
// every wave does everything again (second edition's structure): 8 waves as 2 x 4, wave tile 128 x 64 = 128 accumulator registers,
// ONE fragment set (48 registers: the other wave of the SIMD covers the LDS round trips), 32 staging registers (each thread loads,
// splits and stores 1 / 512 of both tiles: four 16-byte pieces per operand and K step).  LDS: two planes per operand, 64 KB per stage,
// two stages; the epilogue turns its tiles through the stage that has just been consumed (a barrier per item keeps a fast wave's
// next plane stores out of a slow wave's scratch).
constexpr int BN8 = 256;
constexpr int PLB8 = BN8 * ROWB;
constexpr int STAGE8 = 2 * PLA + 2 * PLB8;          // 65 536 bytes
constexpr int W8_LDS = 2 * STAGE8;

__device__ __forceinline__ void tile_origin8(const Params& p, int t, int& m0, int& n0) {
    const int ntile = p.mt * p.nt;                   // p.nt counts 256-column tiles here
    const int q = ntile / 8, r = ntile % 8, x = t & 7, j = t >> 3;
    const int bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    m0 = (bid / p.nt) * BM;
    n0 = (bid % p.nt) * BN8;
}

template <bool BKC>
__global__ __launch_bounds__(512, 1) void gemm_w8_kernel(Params p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const float sa_ = f16_scale(amax_read(p.amaxA)), sb_ = f16_scale(amax_read(p.amaxB));
    const f32x2_t scA = {sa_, 2048.f * sa_}, scB = {sb_, 2048.f * sb_};
    const float unscale = (1.f / sa_) * (1.f / sb_);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = (w >> 2) * 128, wn = (w & 3) * 64;
    const int li = lane & 31, lh = lane >> 5;
    const int total = p.nfull, G = gridDim.x;
    if ((int)blockIdx.x >= total) return;

    Src<true, BM> sa;
    Src<BKC, BN8> sb;
    sa.init_lds(tid);
    sb.init_lds(tid);
    int p_item = blockIdx.x, p_k0 = 0;
    bool p_live = true;
    auto p_open = [&]() {
        int m0, n0;
        tile_origin8(p, p_item, m0, n0);
        sa.init(p.A, p.lda, p.M, m0, 0, tid);
        sb.init(p.B, p.ldb, p.N, n0, 0, tid);
        p_k0 = 0;
    };
    auto produce = [&]() {                          // global loads of the next K step in program order (also across items)
        if (!p_live) return;
        sa.load(p_k0, p.K);
        sb.load(p_k0, p.K);
        p_k0 += BK;
        if (p_k0 >= p.K) {
            p_item += G;
            if (p_item < total) p_open(); else p_live = false;
        }
    };
    auto stage_store = [&](int st) {
        sa.template store<PLA, 2, 1>(lds + st * STAGE8, scA);
        sb.template store<PLB8, 2, 2>(lds + st * STAGE8 + 2 * PLA, scB);
    };
    const char* fa[2];
    const char* fb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        fa[s] = lds + wm * ROWB + plane_off(li, 2 * s + lh);
        fb[s] = lds + 2 * PLA + wn * ROWB + plane_off(li, 2 * s + lh);
    }
    struct F4 { f16x8 a[2][4], b[2][2]; };
    auto rd = [&](F4& f, const char* pa, const char* pb) {
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
#pragma unroll
            for (int t = 0; t < 4; ++t) f.a[pi][t] = *reinterpret_cast<const f16x8*>(pa + pi * PLA + t * 32 * ROWB);
#pragma unroll
            for (int t = 0; t < 2; ++t) f.b[pi][t] = *reinterpret_cast<const f16x8*>(pb + pi * PLB8 + t * 32 * ROWB);
        }
    };
    f32x16 acc[4][2];
    auto mm = [&](const F4& f) {                    // a2 (2^-11 b1) + a1 b2 + a1 b1, the small terms first
        const f16x8 k = {(_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f,
                         (_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f};
        const f16x8 bs0 = f.b[0][0] * k, bs1 = f.b[0][1] * k;
#ifdef BF3_AB_NOMFMA
#pragma unroll
        for (int a = 0; a < 4; ++a) asm volatile("" :: "v"(f.a[0][a]), "v"(f.a[1][a]), "v"(bs0), "v"(bs1), "v"(f.b[0][0]), "v"(f.b[0][1]), "v"(f.b[1][0]), "v"(f.b[1][1]));
        return;
#endif
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[1][a], bs0, acc[a][0], 0, 0, 0);
            acc[a][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[1][a], bs1, acc[a][1], 0, 0, 0);
        }
#pragma unroll
        for (int q = 1; q >= 0; --q)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[0][a], f.b[q][0], acc[a][0], 0, 0, 0);
                acc[a][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[0][a], f.b[q][1], acc[a][1], 0, 0, 0);
            }
    };
    float cmax = 0.f;
    F4 f;
    p_open();
    produce();
    stage_store(0);
    produce();
    __syncthreads();
    int cur_st = 0;
    for (int c_item = blockIdx.x; c_item < total; c_item += G) {
        int m0, n0;
        tile_origin8(p, c_item, m0, n0);
        float zero = 0.f;
        asm volatile("" : "+v"(zero));
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] = zero;
        float bv[2] = {0.f, 0.f};
        for (int c_k0 = 0; c_k0 < p.K; c_k0 += BK) {
            const int so = cur_st * STAGE8;
            const bool fast = p_live && p_k0 + BK <= p.K;
            if (c_k0 + BK >= p.K && p.bias) {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int n = n0 + wn + 32 * b + li;
                    bv[b] = p.bias[n < p.N ? n : 0];
                }
            }
            BF3_FENCE();
            rd(f, fa[0] + so, fb[0] + so);
            mm(f);
            stage_store(cur_st ^ 1);                 // the tile of step s + 1 (its loads were issued a step ago)
            BF3_FENCE();
            rd(f, fa[1] + so, fb[1] + so);
            mm(f);
            sa.load_sched(fast);                     // the tile of step s + 2 into the registers the split has released
            sb.load_sched(fast);
            BF3_FENCE();
#ifdef BF3_AB_NOBAR
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // stage s + 1 is complete; every wave has read stage s
#endif
            BF3_FENCE();
            if (fast) {
                p_k0 += BK;
                if (p_k0 >= p.K) {
                    p_item += G;
                    if (p_item < total) p_open(); else p_live = false;
                }
            } else {
                produce();
            }
            BF3_FENCE();
            cur_st ^= 1;
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] *= unscale;
        float* C = p.C;
#ifdef BF3_AB_NOEPI
        if (lane == 0) C[(int64_t)m0 * p.ldc + n0 + w] = acc[0][0][0] + acc[1][1][1] + acc[2][0][2] + acc[3][1][3];
        __syncthreads();
        continue;
#endif
        const bool full_m = m0 + BM <= p.M;
        if (full_m && n0 + BN8 <= p.N) {
            // whole tiles leave through a wave-private 5 KB scratch in the stage that has just been consumed (cur_st now names the OTHER one,
            // which already holds the next item's first tile)
            char* sc = lds + (cur_st ^ 1) * STAGE8 + w * 5120;
            float* wr = reinterpret_cast<float*>(sc) + (4 * lh) * 40 + li;
            const float4* rdp = reinterpret_cast<const float4*>(sc + (lane >> 3) * 160 + (lane & 7) * 16);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    float v[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = acc[a][b][e] + bv[b];
                    if (p.act == 1) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) v[e] = elu1(v[e]);
                    }
                    if (p.act == 3) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) v[e] = softplus_nb(v[e]);
                    }
                    if (p.amaxC.slot) {
#pragma unroll
                        for (int e = 0; e < 16; e += 2) cmax = fmaxf(cmax, fmaxf(__builtin_fabsf(v[e]), __builtin_fabsf(v[e + 1])));
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) wr[((e & 3) + 8 * (e >> 2)) * 40] = v[e];
                    float* q = C + (int64_t)(m0 + wm + 32 * a + (lane >> 3)) * p.ldc + n0 + wn + 32 * b + 4 * (lane & 7);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 t = rdp[g * 8 * 10];
                        typedef float f4v __attribute__((ext_vector_type(4)));
                        __builtin_nontemporal_store(f4v{t.x, t.y, t.z, t.w}, reinterpret_cast<f4v*>(q + (int64_t)(8 * g) * p.ldc));
                    }
                }
            }
        } else {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int n = n0 + wn + 32 * b + li;
                if (n >= p.N) continue;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    float v[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = acc[a][b][e] + bv[b];
                    if (p.act == 1) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) v[e] = elu1(v[e]);
                    }
                    if (p.act == 3) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) v[e] = softplus_nb(v[e]);
                    }
                    const int mb = m0 + wm + 32 * a + 4 * lh;
                    float* crow = C + (int64_t)mb * p.ldc + n;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int dm = (e & 3) + 8 * (e >> 2);
                        if (mb + dm < p.M) crow[(int64_t)dm * p.ldc] = v[e];
                    }
                    if (p.amaxC.slot) {
#pragma unroll
                        for (int e = 0; e < 16; e += 2) cmax = fmaxf(cmax, fmaxf(__builtin_fabsf(v[e]), __builtin_fabsf(v[e + 1])));
                    }
                }
            }
        }
        __syncthreads();                             // the scratch sits in the stage the next item's first step stores into
    }
    amax_publish_wave(cmax, p.amaxC);
}

// C tile = epi(sum over the K slices of a split tile), fixed summation order: as gemm_fixup_kernel of gemm_f32.hip for 256 x 128 tiles
__global__ __launch_bounds__(256) void gemm_bf3_fixup_kernel(Params p) {
    __shared__ float4 part[3][64];
    const int tr = blockIdx.y, q = threadIdx.y;
    const int e = blockIdx.x * 64 + threadIdx.x, ml = e >> 5, nl = 4 * (e & 31);
    const float* s = p.slab + (int64_t)tr * p.nsl * TILE + ml * BN + nl;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    int i = q;
    for (; i + 12 < p.nsl; i += 16) {
        const float4 u0 = ld4(s + (int64_t)i * TILE), u1 = ld4(s + (int64_t)(i + 4) * TILE);
        const float4 u2 = ld4(s + (int64_t)(i + 8) * TILE), u3 = ld4(s + (int64_t)(i + 12) * TILE);
        v.x = (((v.x + u0.x) + u1.x) + u2.x) + u3.x; v.y = (((v.y + u0.y) + u1.y) + u2.y) + u3.y;
        v.z = (((v.z + u0.z) + u1.z) + u2.z) + u3.z; v.w = (((v.w + u0.w) + u1.w) + u2.w) + u3.w;
    }
    for (; i < p.nsl; i += 4) {
        const float4 u = ld4(s + (int64_t)i * TILE);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    if (q) part[q - 1][threadIdx.x] = v;
    __syncthreads();
    if (q) return;
    const float4 g1 = part[0][threadIdx.x], g2 = part[1][threadIdx.x], g3 = part[2][threadIdx.x];
    float o[4] = {(v.x + g1.x) + (g2.x + g3.x), (v.y + g1.y) + (g2.y + g3.y), (v.z + g1.z) + (g2.z + g3.z), (v.w + g1.w) + (g2.w + g3.w)};
    int z, m0, n0;
    tile_origin(p, p.nfull + tr, z, m0, n0);
    const int m = m0 + ml, n = n0 + nl;
    float cmax = 0.f;
    if (m < p.M && n < p.N) {
        float* c = p.C + (int64_t)z * p.sC + (int64_t)m * p.ldc + n;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (n + j >= p.N) break;
            float x = o[j] + (p.bias ? p.bias[(int64_t)z * p.sBias + n + j] : 0.f);
            if (p.act == 1) x = elu1(x);
            if (p.act == 3) x = softplus_nb(x);
            if (p.act == 2) x += c[j];
            c[j] = x;
            cmax = fmaxf(cmax, __builtin_fabsf(x));
        }
    }
    amax_publish_wave(cmax, p.amaxC);                // q == 0: one whole wave (threadIdx.y selects the wave)
}

struct Plan { int nfull, nsplit, nsl, kslice; };
// K slices for the tiles of a partly filled last round pay a fix-up launch (~6 us) and the slab round trip: only worth it when a whole
// tile's K loop is long
constexpr int g_split_min_ksteps = 4;   // thresholds 12 / 20 / 40 measured equal or slower on the whole update (profiles/r05_gemm.md)
inline Plan make_plan(int M, int N, int K, int batch) {
    const long nbt = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN) * batch;
    const int ksteps = (K + BK - 1) / BK;
    Plan pl{(int)nbt, 0, 1, ksteps * BK};
    const int r = (int)(nbt % GRID);
    if (r == 0 || r > GRID / 2 || ksteps < g_split_min_ksteps) return pl;
    int s = std::min(GRID / r, ksteps / 2);
    const int per = (ksteps + s - 1) / s;
    s = (ksteps + per - 1) / per;
    if (s < 2) return pl;
    pl.nfull = (int)(nbt - r); pl.nsplit = r; pl.nsl = s; pl.kslice = per * BK;
    return pl;
}

template <bool AKC, bool BKC, int SP>
int launch_one(const Params& p, dim3 grid, hipStream_t s) {
    // the attribute is per device (and the first call may come from any thread): one flag per device id, set after the call succeeds
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return RESEL_ELAUNCH;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)gemm_bf3_kernel<AKC, BKC, SP>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE) != hipSuccess)
            return RESEL_ELAUNCH;
        attr_set[dev].store(true, std::memory_order_release);
    }
    launch_timed(RESEL_PROF_GEMM, gemm_bf3_kernel<AKC, BKC, SP>, grid, dim3(NTH), (size_t)(2 * STAGE), s, p);
    return RESEL_OK;
}

template <bool AKC, bool BKC, int SP, int EPI = 0>
int launch_ws(const Params& p, dim3 grid, hipStream_t s) {
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return RESEL_ELAUNCH;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)gemm_ws_kernel<AKC, BKC, SP, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS) != hipSuccess)
            return RESEL_ELAUNCH;
        attr_set[dev].store(true, std::memory_order_release);
    }
    launch_timed(RESEL_PROF_GEMM, gemm_ws_kernel<AKC, BKC, SP, EPI>, grid, dim3(NTH_WS), (size_t)WS_LDS, s, p);
    return RESEL_OK;
}

// edition of the split GEMM: 3 = producer / consumer waves (gemm_ws_kernel), 2 = every wave does everything (gemm_bf3_kernel)
constexpr int g_nt = 1;               // non-temporal C stores of the third edition: 23.43 -> 23.30 ms per update, same box
int g_edition = [] { const char* e = getenv("RESEL_GEMM_EDITION"); return e ? atoi(e) : 3; }();

}  // namespace

#ifdef BF3_AB_CLOCK
extern "C" int resel_bf3_debug_clock(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_bf3_clock), sizeof(unsigned long long) * 2 * GRID);
}
#endif

namespace resel {

size_t gemm_bf3_workspace_bytes(int M, int N, int K, int batch) {
    const Plan pl = make_plan(M, N, K, batch);
    return (size_t)pl.nsplit * pl.nsl * TILE * sizeof(float);
}

// split in {3, 6, 9}, K >= 32; argument checks are the caller's (resel_gemm_f32)
// the fused epilogues (act 4 / 5) exist on the producer / consumer edition, for whole-K items: mode 2, K a multiple of 32, M > 128
bool gemm_bf3_fused_ok(int M, int N, int K, int64_t lda, int64_t ldb) {
    return g_edition == 3 && M > 128 && N >= 4 && K >= BK && K % BK == 0 && lda < (1 << 22) && ldb < (1 << 22);
}

int gemm_bf3_launch(const float* A, int64_t lda, int64_t strideA, int a_kcontig, const float* B, int64_t ldb, int64_t strideB, int b_kcontig,
                    const float* bias, int64_t strideBias, int act, float* C, int64_t ldc, int64_t strideC, void* workspace,
                    int M, int N, int K, int batch, int split, hipStream_t s, const float* amaxA, const float* amaxB,
                    unsigned long long* amax_c, unsigned amax_epoch, const float* aux, int64_t ldaux, int64_t strideAux, float* red, int redrows) {
    if (split == 2 && (!amaxA || !amaxB)) return RESEL_EINVAL;
    Plan pl = make_plan(M, N, K, batch);
    if (act >= 4) {                                  // every tile whole: the epilogue reductions are written per (tile, wave), no K slices
        if (split != 2 || !gemm_bf3_fused_ok(M, N, K, lda, ldb) || !aux) return RESEL_EINVAL;
        if (act == 4 && (M % BM || N % BN)) return RESEL_EINVAL;          // the act 4 epilogue has no edge form (resel_gemm_f32_dact splits the rows)
        const long nbt = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN) * batch;
        pl = Plan{(int)nbt, 0, 1, (K + BK - 1) / BK * BK};
    }
    if (pl.nsplit && (!workspace || !aligned16(workspace))) return RESEL_EINVAL;
    Params p{A, B, bias, C, (float*)workspace, lda, ldb, ldc, strideA, strideB, strideC, strideBias, M, N, K, act,
             (M + BM - 1) / BM, (N + BN - 1) / BN, pl.nfull, pl.nsplit, pl.nsl, pl.kslice, amaxA, amaxB, AmaxOut{amax_c, amax_epoch}, g_nt,
             aux, ldaux, strideAux, red, redrows};
    const int64_t total = (int64_t)pl.nfull + (int64_t)pl.nsplit * pl.nsl;
    dim3 grid((unsigned)std::min<int64_t>(total, GRID));
    int rc;
#define BF3_LAUNCH(SP) \
    do { if (a_kcontig && b_kcontig) rc = launch_one<true, true, SP>(p, grid, s); \
         else if (a_kcontig) rc = launch_one<true, false, SP>(p, grid, s); \
         else if (b_kcontig) rc = launch_one<false, true, SP>(p, grid, s); \
         else rc = launch_one<false, false, SP>(p, grid, s); } while (0)
#define WS_LAUNCH(SP) \
    do { if (a_kcontig && b_kcontig) rc = launch_ws<true, true, SP>(p, grid, s); \
         else if (a_kcontig) rc = launch_ws<true, false, SP>(p, grid, s); \
         else if (b_kcontig) rc = launch_ws<false, true, SP>(p, grid, s); \
         else rc = launch_ws<false, false, SP>(p, grid, s); } while (0)
    // third edition: mode 2, whole K steps (also per K slice); row strides within the 24-bit multiply of the piece offsets
    const bool ws = g_edition >= 3 && split == 2 && K % BK == 0 && pl.kslice % BK == 0 && lda < (1 << 22) && ldb < (1 << 22);
    // fourth edition (RESEL_GEMM_EDITION=4): tall products with A [rows][K], whole K steps, a whole number of 256-column tiles or a last
    // one that is more than half full, no bias stride (one batch member), epilogues 0 / 1 / 3
    if (g_edition == 4 && split == 2 && act != 2 && act < 4 && a_kcontig && batch == 1 && K % BK == 0 && K >= 2 * BK && M >= 2048 && N >= 192 &&
        ((N + BN8 - 1) / BN8 * BN8 - N) < 128 && lda < (1 << 22) && ldb < (1 << 22)) {
        Params q = p;
        q.nt = (N + BN8 - 1) / BN8;
        q.nfull = q.mt * q.nt;
        q.nsplit = 0;
        dim3 g8((unsigned)std::min<int64_t>(q.nfull, GRID));
        static std::atomic<bool> attr8[2][64];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return RESEL_ELAUNCH;
        if (!attr8[b_kcontig ? 1 : 0][dev].load(std::memory_order_acquire)) {
            const void* fn = b_kcontig ? (const void*)gemm_w8_kernel<true> : (const void*)gemm_w8_kernel<false>;
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, W8_LDS) != hipSuccess) return RESEL_ELAUNCH;
            attr8[b_kcontig ? 1 : 0][dev].store(true, std::memory_order_release);
        }
        if (b_kcontig) launch_timed(RESEL_PROF_GEMM, gemm_w8_kernel<true>, g8, dim3(512), (size_t)W8_LDS, s, q);
        else launch_timed(RESEL_PROF_GEMM, gemm_w8_kernel<false>, g8, dim3(512), (size_t)W8_LDS, s, q);
        return launch_status();
    }
    if (act >= 4) {                                   // fused epilogues: the layouts the trainer uses - A [rows][K]; B either way
        if (!a_kcontig) return RESEL_EINVAL;
        if (act == 4) rc = b_kcontig ? launch_ws<true, true, 2, 4>(p, grid, s) : launch_ws<true, false, 2, 4>(p, grid, s);
        else rc = b_kcontig ? launch_ws<true, true, 2, 5>(p, grid, s) : launch_ws<true, false, 2, 5>(p, grid, s);
    } else if (ws) WS_LAUNCH(2);
    else if (split == 3) BF3_LAUNCH(3); else if (split == 2) BF3_LAUNCH(2); else BF3_LAUNCH(6);
#undef WS_LAUNCH
#undef BF3_LAUNCH
    if (rc != RESEL_OK) return rc;
    if (pl.nsplit) hipLaunchKernelGGL(gemm_bf3_fixup_kernel, dim3(TILE / 4 / 64, pl.nsplit), dim3(64, 4), 0, s, p);
#ifdef BF3_AB_FIXUP2                    // ablation (right results): every fix-up launched TWICE - what the 86 fix-up launches of an update cost, measured
    if (pl.nsplit && act != 2) hipLaunchKernelGGL(gemm_bf3_fixup_kernel, dim3(TILE / 4 / 64, pl.nsplit), dim3(64, 4), 0, s, p);   // with the data left intact
#endif                                  // (skipping the fix-up instead leaves garbage tiles: the degenerate data lowers the chip's power draw, its clock
                                        //  rises and EVERY kernel of the update runs 5-13 % faster - profiles/r06_gemm.md)
    return launch_status();
}

}  // namespace resel
