// fp32 GEMM (fp32 in, fp32 accumulate, fp32 out) on the matrix cores of gfx950, with the tails of the fc / efc-E layers fused into
// the epilogue.  The products are formed either by the f32-input instruction (v_mfma_f32_32x32x2_f32: exact fp32 products and
// sums, 64 FLOP/clk/SIMD) or - 'SPLIT' modes, the default - from an exact three-way bf16 split of both operands on the bf16
// instruction (v_mfma_f32_32x32x16_bf16); see the SPLIT section below and DESIGN.md 4 'The GEMMs'.
//
//     C[b][m][n] = epi( sum_k A[b](m, k) * B[b](n, k) )            b = ensemble member (blockIdx.z), strides per operand
//
// Each operand is read either as a [rows][K] matrix (k contiguous: activations x, weights W[out][in]) or as a [K][rows]
// matrix (row index contiguous: the TRANSPOSED use of an activation matrix in a weight gradient, or a weight matrix in an
// input gradient), so one kernel serves
//     forward   y[M, out]  = x[M, in] W[out, in]^T          A = x  [rows][K],  B = W  [rows][K]   (+ bias, + ELU)
//     dgrad     dx[M, in]  = dy[M, out] W[out, in]          A = dy [rows][K],  B = W  [K][rows]
//     wgrad     dW[out,in] = dy[M, out]^T x[M, in]          A = dy [K][rows],  B = x  [K][rows]   (K = M = 66 752: split over blocks)
// (reference: torch.nn.Linear in models/rnn_base.py:101-105 and EnsembleLinear.forward, models/ensemble_linear_model.py:36-49).
//
// Tiling: 128 x 128 block tile, 4 waves as 2 x 2, wave tile 64 x 64 = 2 x 2 MFMA tiles (64 accumulator registers), K step 32.
// A 32x32x2 MFMA takes one A and one B value per lane: lane (i = l & 31, h = l >> 5) supplies A(i, k_h) and B(j = i, k_h).  A
// group of four MFMAs covers 8 consecutive k with the assignment k = 8 g + 4 h + e (e = MFMA in the group) - the sum over k is
// order-free per output, and with it a lane's four operand values are CONTIGUOUS in k: the LDS image of either operand is
// [row][k] (rows padded to 36 floats: conflict-free ds_read_b128, 256 B/clk) and one ds_read_b128 feeds four MFMAs.
// A [K][rows] operand is transposed on its way in: a thread loads a 4 (k) x 4 (rows) patch as four float4 along rows and
// stores its four COLUMNS as ds_write_b128 - the 4 x 4 transpose is register naming.
//
// Pipeline (one barrier per K step, LDS double-buffered, global -> registers -> LDS two steps ahead):
//     step s:  MFMA g0 | ds_write(step s+1) | MFMA g1 | global loads(step s+2) | MFMA g2 | barrier | ds_read frags(s+1) g0..g2
//              | MFMA g3 | ds_read frags(s+1) g3
// so the matrix pipe always has 16 MFMAs (1024 cycles) queued behind the barrier and the LDS round trip of the next step's
// fragments.  History (66 752 x 2048 x 384, PMC in DESIGN.md): fragments re-read next to each group of 8 MFMAs (the compiler's
// placement) 67 % pipe-busy, 102 TFLOP/s; all 16 reads hoisted above the 64 MFMAs 113 TFLOP/s; this pipeline: see DESIGN.md.
//
// Blocks are persistent: block b walks the items b, b + G, ... (G = 512 = two blocks per CU); the loads of an item's first
// steps and its epilogue stores overlap the neighbouring items' MFMAs.  An item is a whole output tile or - for the LAST,
// partly filled round of tiles, and for weight gradients (few tiles, K = 66 752) - a K slice of a tile whose partial sums go
// to a workspace slab; a fix-up kernel adds the slices in fixed order (deterministic, no atomics) and applies bias/activation.
// Without that the last round cost a full tile time at 4 % occupancy (1044 tiles on 512 slots: 68 % efficiency).
// Tile ids are XCD-aware: the n-tiles of one m-tile share an L2.
#include "resel_common.h"
#include <algorithm>
#include <cstdlib>

namespace resel {                      // gemm_bf3.hip: the split modes with the operands split once per block (second edition)
size_t gemm_bf3_workspace_bytes(int M, int N, int K, int batch);
int gemm_bf3_launch(const float* A, int64_t lda, int64_t strideA, int a_kcontig, const float* B, int64_t ldb, int64_t strideB, int b_kcontig,
                    const float* bias, int64_t strideBias, int act, float* C, int64_t ldc, int64_t strideC, void* workspace,
                    int M, int N, int K, int batch, int split, hipStream_t s, const float* amaxA, const float* amaxB,
                    unsigned long long* amax_c, unsigned amax_epoch, const float* aux = nullptr, int64_t ldaux = 0, int64_t strideAux = 0,
                    float* red = nullptr, int redrows = 0);
bool gemm_bf3_fused_ok(int M, int N, int K, int64_t lda, int64_t ldb);
}

namespace {
using namespace resel;

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 128, BN = 128, BK = 32, NG = BK / 8;
constexpr int LDK = BK + 4;           // LDS image [row][k]: 36 floats per row
constexpr int GRID = 512;             // persistent blocks: two per CU
constexpr int TILE = BM * BN;

struct GemmParams {
    const float *A, *B, *bias;
    float *C, *slab;
    int64_t lda, ldb, ldc, sA, sB, sC, sBias;    // leading dimensions (floats) and per-batch strides
    int M, N, K;
    int act;                                      // 0 none, 1 ELU, 2 C += product
    int mt, nt;                                   // tiles along m and n
    int nfull;                                    // items [0, nfull): whole tiles; then nsplit tiles x nsl K slices of kslice
    int nsplit, nsl, kslice;
    AmaxOut amaxC;                                // optional: publish max |C| of the stored values
};

__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : fast_exp(x) - 1.f; }

// tile t (member-major) -> member z, tile origin (m0, n0).  XCD-aware: the tiles of one XCD (ids congruent mod 8) walk the
// n-tiles of one m-tile after another.
__device__ __forceinline__ void tile_origin(const GemmParams& p, int t, int& z, int& m0, int& n0) {
    const int ntile = p.mt * p.nt;
    z = t / ntile;
    const int tt = t - z * ntile;
    const int q = ntile / 8, r = ntile % 8, x = tt & 7, j = tt >> 3;
    const int bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    m0 = (bid / p.nt) * BM;
    n0 = (bid % p.nt) * BN;
}

struct Item { int m0, n0, z, kbeg, kend, split; };     // split: index of the slab tile + 1, 0 for a whole tile
__device__ __forceinline__ Item decode(const GemmParams& p, int it) {
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

// One operand's share of a thread in a K step: four float4 along the operand's contiguous axis.
//   KC  (P[row][k]):  piece i = row (tid / 8 + 32 i), k = 4 (tid % 8) .. + 3
//   !KC (P[k][row]):  piece j = k (4 (tid % 8) + j), rows 4 (tid / 8) .. + 3
// Rows beyond the operand's extent are read from row 0 instead (their products land in output rows / columns that are never
// stored); k beyond the slice end must contribute zeros and is the only guarded case (last step of a K not a multiple of 32).
template <bool KC>
struct Src {
    const char* base;                              // uniform (SGPR pair): operand + member + tile origin + K position
    const char* base0;                             // ... at k = 0 of the operand (always a whole, valid K step: K >= 32 here)
    uint32_t off[4];                               // this thread's byte offsets of its four pieces from `base`
    int64_t step;                                  // bytes per K step
    int kofs;                                      // k of piece 0 inside the step
    __device__ __forceinline__ void init(const float* P, int64_t ld, int rows, int r0, int k0, int tid) {
        kofs = 4 * (tid & 7);
        if (KC) {
            base = (const char*)(P + (int64_t)r0 * ld + k0);
            base0 = (const char*)(P + (int64_t)r0 * ld);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rl = (tid >> 3) + 32 * i;
                off[i] = (uint32_t)(((r0 + rl < rows ? rl : 0) * ld + kofs) * 4);
            }
            step = BK * 4;
        } else {
            base = (const char*)(P + (int64_t)k0 * ld + r0);
            base0 = (const char*)(P + r0);
            const int rl = 4 * (tid >> 3);
#pragma unroll
            for (int j = 0; j < 4; ++j) off[j] = (uint32_t)(((kofs + j) * ld + (r0 + rl < rows ? rl : 0)) * 4);
            step = (int64_t)BK * ld * 4;
        }
    }
    // Branch-free load for the scheduled region of the SPLIT loop: `full` (uniform) = a whole K step is due - load it and
    // advance; otherwise (K tail, or nothing left to fetch) the same instructions re-read the tile's step at k = 0 - in bounds
    // for every item, also a K slice shorter than one step (the host sends K < 32 to the fp32 kernel) - and the caller's
    // guarded `load` then provides the real data.
    __device__ __forceinline__ void load_sched(float4 (&r)[4], bool full) {
        const char* b = full ? base : base0;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const float4*>(b + off[i]);
        if (full) base += step;
    }
    __device__ __forceinline__ void load(float4 (&r)[4], int k0, int kend) {
        if (k0 + BK <= kend) {
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const float4*>(base + off[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k0 + kofs + (KC ? 0 : i) < kend) r[i] = *reinterpret_cast<const float4*>(base + off[i]);
            }
        }
        base += step;
    }
};
template <bool KC>
__device__ __forceinline__ void tile_store(float* __restrict__ S, int tid, const float4 (&r)[4]) {
    if (KC) {
#pragma unroll
        for (int i = 0; i < 4; ++i) st4(S + ((tid >> 3) + 32 * i) * LDK + 4 * (tid & 7), r[i]);
    } else {
        float* q = S + (4 * (tid >> 3)) * LDK + 4 * (tid & 7);
        st4(q, make_float4(r[0].x, r[1].x, r[2].x, r[3].x));
        st4(q + LDK, make_float4(r[0].y, r[1].y, r[2].y, r[3].y));
        st4(q + 2 * LDK, make_float4(r[0].z, r[1].z, r[2].z, r[3].z));
        st4(q + 3 * LDK, make_float4(r[0].w, r[1].w, r[2].w, r[3].w));
    }
}

struct Frags { float4 a[NG][2], b[NG][2]; };
struct Planes;
// slot g of the lane's fragments.  fp32 MFMA: k = 8 g + 4 h .. + 3 (As / Bs point at the lane's row + 4 h).  SPLIT: the lane
// owns k = 16 s + 8 h .. + 7 of slab s in slots 2 s and 2 s + 1 (As / Bs point at the lane's row + 8 h).
template <int SPLIT>
__device__ __forceinline__ void read_frags(Frags& f, const float* __restrict__ As, const float* __restrict__ Bs, int g) {
    const int ko = SPLIT ? 16 * (g >> 1) + 4 * (g & 1) : 8 * g;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        f.a[g][t] = ld4(As + 32 * t * LDK + ko);
        f.b[g][t] = ld4(Bs + 32 * t * LDK + ko);
    }
}
__device__ __forceinline__ void read_a(Frags& f, const float* __restrict__ As, int g) {      // SPLIT-mode slot g of A / of B
#pragma unroll
    for (int t = 0; t < 2; ++t) f.a[g][t] = ld4(As + 32 * t * LDK + 16 * (g >> 1) + 4 * (g & 1));
}
__device__ __forceinline__ void read_b(Frags& f, const float* __restrict__ Bs, int g) {
#pragma unroll
    for (int t = 0; t < 2; ++t) f.b[g][t] = ld4(Bs + 32 * t * LDK + 16 * (g >> 1) + 4 * (g & 1));
}
// quarter q of a K step's matrix work
template <int SPLIT>
__device__ __forceinline__ void quarter(f32x16 (&acc)[2][2], const Frags& f, Planes (&pl)[2], int q);
__device__ __forceinline__ void mfma_group(f32x16 (&acc)[2][2], const Frags& f, int g) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[g][a].x, f.b[g][b].x, acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[g][a].y, f.b[g][b].y, acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[g][a].z, f.b[g][b].z, acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[g][a].w, f.b[g][b].w, acc[a][b], 0, 0, 0);
        }
}
// ---- SPLIT mode: fp32 operands as three bf16 planes, products on the bf16 matrix cores -------------------------------------
// x = x1 + x2 + x3 EXACTLY, each plane a bf16 (8 significant bits) obtained by truncation: x1 = top 16 bits of x, x2 = top 16
// bits of (x - x1) (exact difference), x3 = x - x1 - x2 (at most 8 significant bits left: exact).  A bf16 x bf16 product is
// exact in fp32, so  a b = sum_{p,q} a_p b_q  term by term; v_mfma_f32_32x32x16_bf16 accumulates the terms in fp32 like the
// fp32 MFMA accumulates a b itself.  SPLIT = 9 keeps all nine terms (the product is represented exactly: the same contract as
// the fp32 instruction, with nine accumulator roundings in place of one); SPLIT = 6 drops a2 b3, a3 b2, a3 b3 (each at most
// 2^-24 |a b|: the size of ONE fp32 rounding of the product).  8 k per lane and instruction at 32 cycles against 2 k at 64:
// 6 terms cost 192 cycles per 32 x 32 x 16 block where the fp32 instruction needs 512 (9 terms: 288).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct Planes { bf16x8 a[3][2], b[3][2]; };                        // [plane][tile]

__device__ __forceinline__ void split8(const float4& lo, const float4& hi, bf16x8& p1, bf16x8& p2, bf16x8& p3) {
#ifdef GEMM_AB_NOSPLIT                 // ablation (wrong results): no vector work, the matrix instructions and everything else stay
    p1 = __builtin_bit_cast(bf16x8, lo); p2 = __builtin_bit_cast(bf16x8, hi); p3 = p1;
    return;
#endif
    const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    uint32_t w1[4], w2[4], w3[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t u0 = __float_as_uint(x[2 * q]), u1 = __float_as_uint(x[2 * q + 1]);
        w1[q] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);        // {hi16(x1), hi16(x0)}: element 2q in the low half
        const float r0 = x[2 * q] - __uint_as_float(u0 & 0xffff0000u), r1 = x[2 * q + 1] - __uint_as_float(u1 & 0xffff0000u);
        const uint32_t v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
        w2[q] = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
        const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
        w3[q] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
    }
    p1 = __builtin_bit_cast(bf16x8, make_uint4(w1[0], w1[1], w1[2], w1[3]));
    p2 = __builtin_bit_cast(bf16x8, make_uint4(w2[0], w2[1], w2[2], w2[3]));
    p3 = __builtin_bit_cast(bf16x8, make_uint4(w3[0], w3[1], w3[2], w3[3]));
}
// planes of slab s (k = 16 s .. 16 s + 15; fragment slots 2 s and 2 s + 1 hold the lane's 8 consecutive k), one operand at a time
__device__ __forceinline__ void split_a(Planes& pl, const Frags& f, int s) {
#pragma unroll
    for (int t = 0; t < 2; ++t) split8(f.a[2 * s][t], f.a[2 * s + 1][t], pl.a[0][t], pl.a[1][t], pl.a[2][t]);
}
__device__ __forceinline__ void split_b(Planes& pl, const Frags& f, int s) {
#pragma unroll
    for (int t = 0; t < 2; ++t) split8(f.b[2 * s][t], f.b[2 * s + 1][t], pl.b[0][t], pl.b[1][t], pl.b[2][t]);
}
// Program order IS issue order within a wave: twelve MFMAs written back to back hold the wave's issue slot for 12 x 32 cycles
// and the vector work behind them then runs with the matrix pipe idle (PMC: MFMA busy 50 % + VALU issue 56 % = the whole
// kernel).  Ask the scheduler for MFMA, NV vector instructions, MFMA, ... (tools/micro/mfma_valu_overlap.hip: five to six
// vector instructions hide completely behind one 32-cycle MFMA, the rest cost 4.75 cycles each).
// MEM: what else rides in the region - 2: eight LDS writes, eight global loads (each refills the register the write released),
// four LDS reads; 3: four LDS reads
template <int NM, int NV, int MEM = 0>
__device__ __forceinline__ void interleave_mfma_valu() {
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
        if (MEM == 2 && i < 8) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        if (MEM == 2 && i < 8) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        if (MEM != 0 && i >= 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
}
// one term a_P b_Q on the four 32 x 32 tiles of the wave (each accumulator is touched every fourth instruction)
template <int P, int Q>
__device__ __forceinline__ void mfma_term(f32x16 (&acc)[2][2], const Planes& pl) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#ifdef GEMM_AB_NOMFMA                  // ablation (wrong results): the operands are consumed, no matrix instruction is issued
            asm volatile("" :: "v"(pl.a[P][a]), "v"(pl.b[Q][b]));
#else
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pl.a[P][a], pl.b[Q][b], acc[a][b], 0, 0, 0);
#endif
        }
}
// half 0: the small terms (added first), half 1: the leading ones
template <int SPLIT>
__device__ __forceinline__ void mfma_terms(f32x16 (&acc)[2][2], const Planes& pl, int half) {
    if (half == 0) {
        if (SPLIT == 9) { mfma_term<2, 2>(acc, pl); mfma_term<2, 1>(acc, pl); mfma_term<1, 2>(acc, pl); }
        mfma_term<2, 0>(acc, pl); mfma_term<0, 2>(acc, pl); mfma_term<1, 1>(acc, pl);
    } else {
        mfma_term<1, 0>(acc, pl); mfma_term<0, 1>(acc, pl); mfma_term<0, 0>(acc, pl);
    }
}

// quarter q of a K step's matrix work.  fp32 MFMA: k group q.  SPLIT: slab q / 2, small terms (even q) or leading terms (odd q),
// interleaved with a quarter of the operand splitting the NEXT slab needs: B then A of slab 1 during q = 0, 1; A then B of the
// next step's slab 0 during q = 2, 3 (their fragments are read right after the barrier that precedes q = 2).
template <int SPLIT>
__device__ __forceinline__ void quarter(f32x16 (&acc)[2][2], const Frags& f, Planes (&pl)[2], int q) {
    mfma_group(acc, f, q);
}
// SPLIT-mode quarter q: slab q / 2, small terms (even q) or leading terms (odd q), interleaved with a quarter of the operand
// splitting the NEXT slab needs - B then A of slab 1 during q = 0, 1; A then B of the next step's slab 0 during q = 2, 3 -
// and with the memory instructions of the step (MEM, see interleave_mfma_valu), all in one scheduling region.
template <int SPLIT, int Q, int MEM>
__device__ __forceinline__ void split_quarter(f32x16 (&acc)[2][2], const Frags& f, Planes (&pl)[2]) {
    if (Q == 0) split_b(pl[1], f, 1);
    if (Q == 1) split_a(pl[1], f, 1);
    if (Q == 2) split_a(pl[0], f, 0);
    if (Q == 3) split_b(pl[0], f, 0);
    mfma_terms<SPLIT>(acc, pl[Q >> 1], Q & 1);
    // pin the planes to THIS region (the optimiser otherwise sinks their computation to the block of their first use)
    Planes& np = pl[Q < 2 ? 1 : 0];
#pragma unroll
    for (int pi = 0; pi < 3; ++pi)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (Q == 0 || Q == 3) asm volatile("" : "+v"(np.b[pi][t]));
            else asm volatile("" : "+v"(np.a[pi][t]));
        }
    if (SPLIT == 6 || (Q & 1)) interleave_mfma_valu<12, 8, MEM>();
    else interleave_mfma_valu<24, 4, MEM>();
}

#define RESEL_FENCE() __builtin_amdgcn_sched_barrier(0)
#ifdef GEMM_STAMP                      // issue-time stamps of one wave (block 0, wave 0) at the phase boundaries of its K steps
__device__ unsigned long long g_gemm_stamps[64 * 16];
#define STAMP(i) do { if (stamp_on && stamp_step < 64) g_gemm_stamps[stamp_step * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

template <bool AKC, bool BKC, int SPLIT>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmParams p) {
    constexpr int ASZ = BM * LDK, BSZ = BN * LDK;
    __shared__ __attribute__((aligned(16))) float lds[2 * (ASZ + BSZ)];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
    const int li = lane & 31, lh = lane >> 5;
    const int total = p.nfull + p.nsplit * p.nsl;
    const int G = gridDim.x;
    if ((int)blockIdx.x >= total) return;

    // producer: global -> registers, two steps ahead of the MFMAs
    Src<AKC> sa;
    Src<BKC> sb;
    float4 ra[4], rb[4];
    int p_item = blockIdx.x, p_k0, p_kend;
    bool p_live = true;
    auto p_open = [&]() {
        const Item it = decode(p, p_item);
        sa.init(p.A + (int64_t)it.z * p.sA, p.lda, p.M, it.m0, it.kbeg, tid);
        sb.init(p.B + (int64_t)it.z * p.sB, p.ldb, p.N, it.n0, it.kbeg, tid);
        p_k0 = it.kbeg; p_kend = it.kend;
    };
    auto produce = [&]() {
        if (!p_live) return;
        sa.load(ra, p_k0, p_kend);
        sb.load(rb, p_k0, p_kend);
        p_k0 += BK;
        if (p_k0 >= p_kend) {
            p_item += G;
            if (p_item < total) p_open(); else p_live = false;
        }
    };
    // consumer
    int c_item = blockIdx.x;
    float* const lA = lds + (wm + li) * LDK + (SPLIT ? 8 : 4) * lh;              // this lane's fragment rows in buffer 0
    float* const lB = lds + ASZ + (wn + li) * LDK + (SPLIT ? 8 : 4) * lh;
    Frags f;
    Planes pl[2];
    p_open();
    produce();
    tile_store<AKC>(lds, tid, ra);
    tile_store<BKC>(lds + ASZ, tid, rb);
    produce();
    if (SPLIT) {                                    // SPLIT runs one step further ahead: both LDS buffers filled, a third tile in registers
        tile_store<AKC>(lds + ASZ + BSZ, tid, ra);
        tile_store<BKC>(lds + ASZ + BSZ + ASZ, tid, rb);
        produce();
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < NG; ++g) read_frags<SPLIT>(f, lA, lB, g);
    if (SPLIT) { split_a(pl[0], f, 0); split_b(pl[0], f, 0); }     // the loop keeps slab 0 split and slab 1's B fragments read
    int nb = ASZ + BSZ;                                             // offset of the buffer the NEXT step goes to
    float cmax = 0.f;                                               // max |C| of this lane's stores (published at the end)
#ifdef GEMM_STAMP
    const bool stamp_on = blockIdx.x == 0 && tid == 0;
    int stamp_step = 0;
#endif
    for (; c_item < total; c_item += G) {
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
            if (SPLIT) {
                // A: q0 | barrier | B: q1 + LDS write of the tile for step s + 2 (into the buffer step s was read from, free
                // once every wave has passed the barrier) + global loads for step s + 3 into the same registers | C: q2 | D: q3.
                // Every quarter also reads the 16 fragment registers the NEXT quarter splits (just in time: 32 raw registers
                // live instead of 64).  Nothing but the barrier is issued outside the shadow of a matrix instruction
                // (issue-time stamps of the version with the memory phases between the quarters: 4 x 620 cycles of quarters
                // + 1600 of ds_write / loads / reads / barrier).
                const int cb = ASZ + BSZ - nb;                          // offset of the buffer this step's tile is in
                RESEL_FENCE();
                STAMP(0);
                read_a(f, lA + cb, 2); read_a(f, lA + cb, 3);           // this step's slab 1, A
                split_quarter<SPLIT, 0, 3>(acc, f, pl);
                RESEL_FENCE();
                STAMP(1);
                __syncthreads();
                STAMP(2);
                if (c_k0 + BK >= cur.kend && p.bias && !cur.split) {    // requested most of a step before the epilogue needs them
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const int n = cur.n0 + wn + 32 * b + li;
                        bv[b] = p.bias[(int64_t)cur.z * p.sBias + (n < p.N ? n : 0)];
                    }
                }
                const float* nA = lA + nb;
                const float* nB = lB + nb;
                const bool fast = p_live && p_k0 + BK <= p_kend;
                RESEL_FENCE();
                tile_store<AKC>(lds + cb, tid, ra);
                tile_store<BKC>(lds + cb + ASZ, tid, rb);
                sa.load_sched(ra, fast);
                sb.load_sched(rb, fast);
                read_a(f, nA, 0); read_a(f, nA, 1);
                split_quarter<SPLIT, 1, 2>(acc, f, pl);
                RESEL_FENCE();
                if (fast) {
                    p_k0 += BK;
                    if (p_k0 >= p_kend) {
                        p_item += G;
                        if (p_item < total) p_open(); else p_live = false;
                    }
                } else {
                    produce();
                }
                RESEL_FENCE();
                STAMP(3);
                read_b(f, nB, 0); read_b(f, nB, 1);
                split_quarter<SPLIT, 2, 3>(acc, f, pl);
                RESEL_FENCE();
                STAMP(4);
                read_b(f, nB, 2); read_b(f, nB, 3);
                split_quarter<SPLIT, 3, 3>(acc, f, pl);
                RESEL_FENCE();
                STAMP(5);
#ifdef GEMM_STAMP
                ++stamp_step;
#endif
            } else {
                RESEL_FENCE();
                quarter<SPLIT>(acc, f, pl, 0);
                RESEL_FENCE();
                tile_store<AKC>(lds + nb, tid, ra);
                tile_store<BKC>(lds + nb + ASZ, tid, rb);
                RESEL_FENCE();
                quarter<SPLIT>(acc, f, pl, 1);
                RESEL_FENCE();
                if (c_k0 + BK >= cur.kend && p.bias && !cur.split) {    // requested most of a step before the epilogue needs them
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const int n = cur.n0 + wn + 32 * b + li;
                        bv[b] = p.bias[(int64_t)cur.z * p.sBias + (n < p.N ? n : 0)];
                    }
                }
                produce();
                RESEL_FENCE();
                quarter<SPLIT>(acc, f, pl, 2);
                RESEL_FENCE();
                __syncthreads();
                read_frags<SPLIT>(f, lA + nb, lB + nb, 0);
                read_frags<SPLIT>(f, lA + nb, lB + nb, 1);
                read_frags<SPLIT>(f, lA + nb, lB + nb, 2);
                RESEL_FENCE();
                quarter<SPLIT>(acc, f, pl, 3);
                RESEL_FENCE();
                read_frags<SPLIT>(f, lA + nb, lB + nb, 3);
                RESEL_FENCE();
            }
            nb = ASZ + BSZ - nb;
        }
        // epilogue: D layout col = lane & 31 (n), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (m)
        if (cur.split) {                                            // partial sums: dense 128 x 128 slab tile
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
                    if (p.amaxC.slot) {
#pragma unroll
                        for (int e = 0; e < 16; e += 2) cmax = fmaxf(cmax, fmaxf(__builtin_fabsf(v[e]), __builtin_fabsf(v[e + 1])));
                    }
                }
            }
        }
    }
    amax_publish_wave(cmax, p.amaxC);
}

// C tile = epi(sum over the K slices of a split tile).  Fixed summation order (deterministic): four interleaved slice groups
// (threadIdx.y) accumulate slices q, q + 4, ... each, then ((g0 + g1) + (g2 + g3)).  grid (TILE / 4 / 64, split tiles), block (64, 4).
__global__ __launch_bounds__(256) void gemm_fixup_kernel(GemmParams p) {
    __shared__ float4 part[3][64];
    const int tr = blockIdx.y, q = threadIdx.y;
    const int e = blockIdx.x * 64 + threadIdx.x, ml = e >> 5, nl = 4 * (e & 31);
    const float* s = p.slab + (int64_t)tr * p.nsl * TILE + ml * BN + nl;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    int i = q;
    for (; i + 12 < p.nsl; i += 16) {                     // four loads in flight per thread
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
    amax_publish_wave(cmax, p.amaxC);
}

// How the output tiles become items: whole tiles for the full rounds of the 512 block slots; the remaining r tiles are cut
// into K slices so that they fill the slots once more (at least two K steps per slice), also when r is everything (weight
// gradients: 6 tiles, K = 66 752).  r > 256 tiles are left whole (a split could not even double the blocks).
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
    const int per = (ksteps + s - 1) / s;           // K steps per slice
    s = (ksteps + per - 1) / per;                   // no empty slices
    if (s < 2) return pl;
    pl.nfull = (int)(nbt - r); pl.nsplit = r; pl.nsl = s; pl.kslice = per * BK;
    return pl;
}

}  // namespace

#ifdef GEMM_STAMP
extern "C" int resel_gemm_debug_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_gemm_stamps), sizeof(unsigned long long) * 64 * 16);
}
#endif

extern "C" size_t resel_gemm_f32_workspace_bytes(int M, int N, int K, int batch) {
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return 0;
    const Plan pl = make_plan(M, N, K, batch);
    return std::max(std::max((size_t)pl.nsplit * pl.nsl * TILE * sizeof(float), gemm_bf3_workspace_bytes(M, N, K, batch)),
                    gemm_any_workspace_bytes(M, N, K, batch));
}

extern "C" int resel_gemm_f32(const float* A, int64_t lda, int64_t strideA, int a_kcontig,
                              const float* B, int64_t ldb, int64_t strideB, int b_kcontig,
                              const float* bias, int64_t strideBias, int act,
                              float* C, int64_t ldc, int64_t strideC, void* workspace,
                              int M, int N, int K, int batch, int split, resel_stream_t stream) {
    if (split == 2) return RESEL_EINVAL;           // mode 2 needs the operand magnitudes: resel_gemm_f32x
    return resel_gemm_f32x(A, lda, strideA, a_kcontig, B, ldb, strideB, b_kcontig, bias, strideBias, act, C, ldc, strideC, workspace,
                           M, N, K, batch, split, nullptr, nullptr, nullptr, 0u, stream);
}

extern "C" int resel_gemm_f32x(const float* A, int64_t lda, int64_t strideA, int a_kcontig,
                               const float* B, int64_t ldb, int64_t strideB, int b_kcontig,
                               const float* bias, int64_t strideBias, int act,
                               float* C, int64_t ldc, int64_t strideC, void* workspace,
                               int M, int N, int K, int batch, int split, const float* amax_a, const float* amax_b,
                               void* amax_c, unsigned amax_epoch, resel_stream_t stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || batch <= 0 || act < 0 || act > 3) return RESEL_EINVAL;
    if (split != 0 && split != 2 && split != 3 && split != 6) return RESEL_EINVAL;
    if (split == 2 && (!amax_a || !amax_b)) return RESEL_EINVAL;
    if (amax_c && (reinterpret_cast<uintptr_t>(amax_c) & 7u)) return RESEL_EINVAL;
    if (split == 2 && M <= 128) split = 6;         // narrow shapes stay on the first edition's fp32-accurate bf16 split
    if (K < BK) split = 0;                         // the split kernels' scheduled loads assume one whole K step per item
    if (lda <= 0 || ldb <= 0 || ldc <= 0) return RESEL_EINVAL;
    // What the matrix-core editions need: float4 loads along each operand's contiguous axis (16-byte aligned rows, that extent a
    // multiple of 4), 32-bit piece offsets inside a tile (128 rows or 32 k of the leading dimension).  Everything else - the 6-wide
    // heads and their gradients, rank-2 projections, odd action counts - and the M <= 8 rows of a rollout step against a whole weight
    // matrix go to gemm_any.hip (exact fp32 FMAs, same epilogues, same magnitude publication): no shape is refused, none is left to a
    // vendor library.
    const bool mfma_ok = !(lda % 4 || ldb % 4 || strideA % 4 || strideB % 4 || !aligned16(A) || !aligned16(B) || (a_kcontig ? K : M) % 4 ||
                           (b_kcontig ? K : N) % 4 || lda >= (int64_t)1 << 22 || ldb >= (int64_t)1 << 22);
    if (!mfma_ok || gemm_any_rows_ok(A, lda, strideA, a_kcontig, 0, B, ldb, strideB, b_kcontig, M, K, act))
        return gemm_any_launch(A, lda, strideA, a_kcontig, B, ldb, strideB, b_kcontig, bias, strideBias, act, C, ldc, strideC, workspace, M, N, K,
                               batch, 0, (unsigned long long*)amax_c, amax_epoch, (hipStream_t)stream);
    // second edition (256 x 128 tiles) unless half of its tile rows would be padding: M <= 128 (narrow weight gradients) runs
    // 1.2-1.4x faster on the first edition's 128 x 128 tiles (66 752-token weight gradients [128, 256]: 48 vs 59 us, [80, 512]: 67 vs 95)
    if ((split == 2 || split == 3 || split == 6) && M > 128)
        return gemm_bf3_launch(A, lda, strideA, a_kcontig, B, ldb, strideB, b_kcontig, bias, strideBias, act, C, ldc, strideC, workspace,
                               M, N, K, batch, split, (hipStream_t)stream, amax_a, amax_b, (unsigned long long*)amax_c, amax_epoch);
    if (split == 3) split = 6;                     // the two-plane mode exists on the second-edition kernel only: narrow shapes keep mode 6
    const Plan pl = make_plan(M, N, K, batch);
    if (pl.nsplit && (!workspace || !aligned16(workspace))) return RESEL_EINVAL;
    GemmParams p{A, B, bias, C, (float*)workspace, lda, ldb, ldc, strideA, strideB, strideC, strideBias, M, N, K, act,
                 (M + BM - 1) / BM, (N + BN - 1) / BN, pl.nfull, pl.nsplit, pl.nsl, pl.kslice, AmaxOut{(unsigned long long*)amax_c, amax_epoch}};
    const int64_t total = (int64_t)pl.nfull + (int64_t)pl.nsplit * pl.nsl;
    dim3 grid((unsigned)std::min<int64_t>(total, GRID));
    hipStream_t s = (hipStream_t)stream;
#define RESEL_GEMM_LAUNCH(SP) \
    do { if (a_kcontig && b_kcontig) launch_timed(RESEL_PROF_GEMM, gemm_f32_kernel<true, true, SP>, grid, dim3(256), 0, s, p); \
         else if (a_kcontig) launch_timed(RESEL_PROF_GEMM, gemm_f32_kernel<true, false, SP>, grid, dim3(256), 0, s, p); \
         else if (b_kcontig) launch_timed(RESEL_PROF_GEMM, gemm_f32_kernel<false, true, SP>, grid, dim3(256), 0, s, p); \
         else launch_timed(RESEL_PROF_GEMM, gemm_f32_kernel<false, false, SP>, grid, dim3(256), 0, s, p); } while (0)
    if (split == 6) RESEL_GEMM_LAUNCH(6);
    else RESEL_GEMM_LAUNCH(0);
#undef RESEL_GEMM_LAUNCH
    if (pl.nsplit) hipLaunchKernelGGL(gemm_fixup_kernel, dim3(TILE / 4 / 64, pl.nsplit), dim3(64, 4), 0, s, p);
    return launch_status();
}

// ---- fused epilogues of the producer / consumer edition (ABI 7; product mode 2 only) ----------------------------------------------
namespace {
// q[z][m] = b3[z] + sum_{j < parts} part[z][j][m]   (fixed order)
__global__ __launch_bounds__(256) void head_fold_kernel(const float* __restrict__ part, int parts, int M, const float* __restrict__ b3, float* __restrict__ q) {
    const int z = blockIdx.y;
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float* pz = part + (int64_t)z * parts * M + m;
    float acc = b3 ? b3[z] : 0.f;
    for (int j = 0; j < parts; ++j) acc += pz[(int64_t)j * M];
    q[(int64_t)z * M + m] = acc;
}
bool fused_args_ok(const float* A, int64_t lda, int64_t strideA, int a_kcontig, const float* B, int64_t ldb, int64_t strideB, int b_kcontig,
                   const float* C, int64_t ldc, int M, int N, int K, int batch, const float* amax_a, const float* amax_b, const void* amax_c,
                   const void* workspace) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || batch <= 0 || !amax_a || !amax_b || !workspace || !aligned16(workspace)) return false;
    if (amax_c && (reinterpret_cast<uintptr_t>(amax_c) & 7u)) return false;
    if (lda % 4 || ldb % 4 || ldc % 4 || strideA % 4 || strideB % 4 || !aligned16(A) || !aligned16(B) || !aligned16(C)) return false;
    if ((a_kcontig ? K : M) % 4 || (b_kcontig ? K : N) % 4 || N % 4) return false;
    if (lda <= 0 || ldb <= 0 || ldc <= 0) return false;
    return gemm_bf3_fused_ok(M, N, K, lda, ldb);
}
}  // namespace

extern "C" int resel_gemm_f32_fused_supported(int kind, int M, int N, int K, int64_t lda, int64_t ldb) {
    if (kind == 4) return (M >= 256 && N % 128 == 0 && gemm_bf3_fused_ok(M, N, K, lda, ldb)) ? 1 : 0;
    if (kind == 5) return gemm_bf3_fused_ok(M, N, K, lda, ldb) ? 1 : 0;
    return 0;
}

extern "C" size_t resel_gemm_f32_fused_workspace_bytes(int M, int N, int K, int batch, int kind) {
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return 0;
    const size_t mt = M / 256, nt = (N + 127) / 128;
    const size_t own = (kind == 5 ? 2 * nt * (size_t)M : (2 * mt + 1) * (size_t)N) * batch * sizeof(float);
    // kind 4: + the split-K slabs of the plain product over the rows past the last whole 256-row tile
    return own + (kind == 4 && M % 256 ? resel_gemm_f32_workspace_bytes(M % 256, N, K, batch) + 256 : 0);
}

namespace {
// rows past the last whole tile of resel_gemm_f32_dact: C *= elu'(Y) in place, column sums into one partial row.  grid (N / 64, batch), 256 threads
__global__ __launch_bounds__(256) void dact_tail_kernel(float* __restrict__ C, int64_t ldc, int64_t sC, const float* __restrict__ Y, int64_t ldy, int64_t sY,
                                                        int rows, int N, float* __restrict__ part, int64_t part_stride, AmaxOut amax) {
    __shared__ float s_sum[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6, z = blockIdx.y;
    float acc = 0.f, mx = 0.f;
    if (c < N) {
        float* Cz = C + (int64_t)z * sC + c;
        const float* Yz = Y + (int64_t)z * sY + c;
        for (int r = rg; r < rows; r += 4) {
            const float y = Yz[(int64_t)r * ldy];
            const float v = Cz[(int64_t)r * ldc] * (y > 0.f ? 1.f : y + 1.f);
            Cz[(int64_t)r * ldc] = v;
            acc += v;
            mx = fmaxf(mx, __builtin_fabsf(v));
        }
    }
    s_sum[rg][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rg == 0 && c < N && part) part[(int64_t)z * part_stride + c] = (s_sum[0][threadIdx.x] + s_sum[1][threadIdx.x]) + (s_sum[2][threadIdx.x] + s_sum[3][threadIdx.x]);
    amax_publish_wave(mx, amax);
}
}  // namespace

extern "C" int resel_gemm_f32_dact(const float* A, int64_t lda, int64_t strideA, int a_kcontig,
                                   const float* B, int64_t ldb, int64_t strideB, int b_kcontig,
                                   const float* Y, int64_t ldy, int64_t strideY,
                                   float* C, int64_t ldc, int64_t strideC, float* dbias, void* workspace,
                                   int M, int N, int K, int batch, const float* amax_a, const float* amax_b,
                                   void* amax_c, unsigned amax_epoch, resel_stream_t stream) {
    if (!fused_args_ok(A, lda, strideA, a_kcontig, B, ldb, strideB, b_kcontig, C, ldc, M, N, K, batch, amax_a, amax_b, amax_c, workspace)) return RESEL_EINVAL;
    if (!Y || ldy % 4 || strideY % 4 || ldy <= 0 || !aligned16(Y) || !a_kcontig || M < 256 || N % 128) return RESEL_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    // the fused epilogue takes whole 256-row tiles; the (at most 255) rows behind them: plain product, then one small in-place pass
    const int Mm = M / 256 * 256, tail = M - Mm, rows = 2 * (Mm / 256) + (tail ? 1 : 0);
    float* part = dbias ? (float*)workspace : nullptr;                     // [batch][rows][N]
    int rc = gemm_bf3_launch(A, lda, strideA, a_kcontig, B, ldb, strideB, b_kcontig, nullptr, 0, 4, C, ldc, strideC, nullptr, Mm, N, K, batch, 2,
                             s, amax_a, amax_b, (unsigned long long*)amax_c, amax_epoch, Y, ldy, strideY, part, rows);
    if (rc != RESEL_OK) return rc;
    if (tail) {
        char* ws2 = (char*)workspace + ((size_t)rows * N * batch * sizeof(float) + 255) / 256 * 256;
        rc = resel_gemm_f32x(A + (int64_t)Mm * lda, lda, strideA, 1, B, ldb, strideB, b_kcontig, nullptr, 0, 0, C + (int64_t)Mm * ldc, ldc, strideC, ws2,
                             tail, N, K, batch, 2, amax_a, amax_b, nullptr, 0u, stream);
        if (rc != RESEL_OK) return rc;
        hipLaunchKernelGGL(dact_tail_kernel, dim3((N + 63) / 64, batch), dim3(256), 0, s, C + (int64_t)Mm * ldc, ldc, strideC, Y + (int64_t)Mm * ldy, ldy, strideY,
                           tail, N, part ? part + (int64_t)(rows - 1) * N : nullptr, (int64_t)rows * N, AmaxOut{(unsigned long long*)amax_c, amax_epoch});
    }
    if (dbias) launch_colsum(part, N, rows, N, dbias, s, 1, 0, batch);
    return launch_status();
}

extern "C" int resel_gemm_f32_head(const float* A, int64_t lda, int64_t strideA, int a_kcontig,
                                   const float* B, int64_t ldb, int64_t strideB, int b_kcontig,
                                   const float* bias, int64_t strideBias, const float* w3, int64_t strideW3, const float* b3,
                                   float* C, int64_t ldc, int64_t strideC, float* q, void* workspace,
                                   int M, int N, int K, int batch, const float* amax_a, const float* amax_b,
                                   void* amax_c, unsigned amax_epoch, resel_stream_t stream) {
    if (!fused_args_ok(A, lda, strideA, a_kcontig, B, ldb, strideB, b_kcontig, C, ldc, M, N, K, batch, amax_a, amax_b, amax_c, workspace)) return RESEL_EINVAL;
    if (!w3 || !q || strideW3 % 4 || !aligned16(w3)) return RESEL_EINVAL;
    const int rc = gemm_bf3_launch(A, lda, strideA, a_kcontig, B, ldb, strideB, b_kcontig, bias, strideBias, 5, C, ldc, strideC, nullptr, M, N, K, batch, 2,
                                   (hipStream_t)stream, amax_a, amax_b, (unsigned long long*)amax_c, amax_epoch, w3, 0, strideW3, (float*)workspace);
    if (rc != RESEL_OK) return rc;
    hipLaunchKernelGGL(head_fold_kernel, dim3((M + 255) / 256, batch), dim3(256), 0, (hipStream_t)stream, (const float*)workspace,
                       2 * ((N + 127) / 128), M, b3, q);
    return launch_status();
}
