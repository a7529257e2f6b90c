// Shared device helpers for the gfx950 kernels (CDNA4: wave64, 160 KiB LDS/CU, 256 CUs in 8 XCDs).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include "../../include/resel_hip.h"

#define RESEL_LOG2E 1.4426950408889634f
#define RESEL_LN2 0.6931471805599453f

namespace resel {

// Scalar (SMEM) loads: a wave-uniform address in the constant address space becomes s_load_dword*,
// so per-step coefficients shared by all 64 lanes cost no VALU/LDS bandwidth and feed v_fma directly
// as SGPR operands.
typedef const float __attribute__((address_space(4)))* cfloat_p;
__device__ __forceinline__ cfloat_p as_uniform(const float* p) {
    return (cfloat_p)(uintptr_t)p;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * RESEL_LOG2E); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

__device__ __forceinline__ float sigmoidf_(float x) { return fast_rcp(1.0f + fast_exp(-x)); }
__device__ __forceinline__ float siluf_(float x) { return x * fast_rcp(1.0f + fast_exp(-x)); }
// d/dx [x * sigmoid(x)] = s * (1 + x * (1 - s))
__device__ __forceinline__ float dsiluf_(float x) {
    float s = sigmoidf_(x);
    return s * (1.0f + x * (1.0f - s));
}
// torch.nn.functional.softplus (beta = 1, threshold = 20) as max(x, 0) + log1p(exp(-|x|)) on v_exp/v_log:
// for e = exp(-|x|) < 1e-4 the series e - e^2/2 replaces log(1 + e) (keeps relative accuracy of tiny deltas);
// at x > 20 the correction term is < 2.1e-9 and rounds away, matching torch's threshold branch.
__device__ __forceinline__ float softplusf_(float x) {
    const float e = fast_exp(-fabsf(x));
    const float l = e < 1e-4f ? e - 0.5f * e * e : __logf(1.0f + e);
    return fmaxf(x, 0.0f) + l;
}

// branch-free form for the hot tile passes of the scan kernels (the ?: above compiles to a divergent branch around v_log):
// both sides are computed, v_log_f32 is used raw (its argument 1 + e is in [1, 2]: no denormal pre-scaling needed)
__device__ __forceinline__ float softplus_nb(float x) {
    const float e = __builtin_amdgcn_exp2f(-RESEL_LOG2E * __builtin_fabsf(x));
    const float lg = __builtin_amdgcn_logf(1.0f + e) * RESEL_LN2;
    const float se = __builtin_fmaf(-0.5f * e, e, e);
    return fmaxf(x, 0.0f) + (e < 1e-4f ? se : lg);
}
__device__ __forceinline__ float silu_nb(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-RESEL_LOG2E * x)); }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// full-wave (64 lane) sum, result in every lane
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Column sums of row-major partial slabs in a fixed order (bitwise reproducible): out[o(c)] = sum_{k<K} part[k * ld + c],
// c < C.  With taps == 0 the output index is c; otherwise the columns are [ch][padded taps KT] and only the last `taps`
// of every KT go out, as out[ch * taps + tap].  256 threads = 16 columns x 16 row groups; grid = ceil(C / 16) blocks.
// blockIdx.y = batch: slab b starts at part + b * K * ld, its sums go to out + b * C (batched form: taps == 0 only).
static __global__ void colsum_kernel(const float* __restrict__ part, int64_t ld, int K, int C, float* __restrict__ out,
                                     int KT, int taps) {
    __shared__ float s_acc[16][17];
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    part += (int64_t)blockIdx.y * K * ld;
    out += (int64_t)blockIdx.y * C;
    float acc = 0.f;
    if (c < C) {                                   // four loads in flight per thread, fixed summation order
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int k = rg;
        for (; k + 48 < K; k += 64) {
            a0 += part[(int64_t)k * ld + c];
            a1 += part[(int64_t)(k + 16) * ld + c];
            a2 += part[(int64_t)(k + 32) * ld + c];
            a3 += part[(int64_t)(k + 48) * ld + c];
        }
        for (; k < K; k += 16) a0 += part[(int64_t)k * ld + c];
        acc = (a0 + a1) + (a2 + a3);
    }
    s_acc[rg][cl] = acc;
    __syncthreads();
    if (rg == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += s_acc[r][cl];
        if (taps == 0) out[c] = t;
        else {
            const int tap = c % KT - (KT - taps);
            if (tap >= 0) out[(c / KT) * taps + tap] = t;
        }
    }
}
inline void launch_colsum(const float* part, int64_t ld, int K, int C, float* out, hipStream_t s, int KT = 1, int taps = 0,
                          int batch = 1) {
    hipLaunchKernelGGL(colsum_kernel, dim3((C + 15) / 16, batch), dim3(256), 0, s, part, ld, K, C, out, KT, taps);
}

// ---- per-dispatch timing (bench.py's roofline measurement) ---------------------------------------------------------
// With profiling on, a kernel launched through launch_timed() is dispatched with hipExtLaunchKernelGGL and a (start, stop)
// HIP event pair bound to THAT dispatch on the stream it runs on (a plain hipEventRecord pair around a launch also counts
// queued predecessors on ROCm).  Slots are the RESEL_PROF_* ids of resel_hip.h; the registry lives in misc.hip.
struct ProfEvents;
void prof_push(int slot, hipEvent_t a, hipEvent_t b);
bool prof_on();
template <typename K, typename... Args>
inline void launch_timed(int slot, K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t s, Args... args) {
    if (!prof_on()) {
        hipLaunchKernelGGL(kernel, grid, block, lds, s, args...);
        return;
    }
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    hipExtLaunchKernelGGL(kernel, grid, block, lds, s, a, b, 0, args...);
    prof_push(slot, a, b);
}

// ---- producer-side operand magnitudes (mode 2 of resel_gemm_f32x) ----------------------------------------------------
// A kernel that WRITES a tensor a later GEMM reads can publish max |x| of what it wrote (include/resel_hip.h "magnitude handles").
// A handle is RESEL_AMAX_SUBSLOTS 8-byte words, 128 bytes apart: word j = {epoch : high 32 | float bits : low 32}.  A publisher
// raises the word its workgroup id selects with ONE 64-bit atomicMax per wave that has something to raise (thousands of waves
// finishing together would serialise on a single word: 88 atomics / us; eight lines, one per XCD under round-robin placement,
// also keep the plain read in front of the atomic fresh - an atomic drops the line from the XCD's own L2 only).  A larger epoch
// outranks any older content, so handles are never zeroed; kernels filling parts of one tensor share a handle and an epoch.
// A reader takes the newest epoch among the eight words and the largest magnitude carrying it.
#define RESEL_AMAX_SUBSLOTS 8
#define RESEL_AMAX_STRIDE 16                     /* 8-byte words between sub-slots (128 bytes) */
struct AmaxOut { unsigned long long* slot; unsigned epoch; };
__device__ __forceinline__ float amax4(float m, float4 v) {
    return fmaxf(fmaxf(m, fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y))), fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w)));
}
__device__ __forceinline__ void amax_publish_wave(float m, AmaxOut o) {      // m >= 0 per lane; all 64 lanes of the wave call
    if (o.slot == nullptr) return;
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
    if (__lane_id() == 0) {
        unsigned long long* w = o.slot + RESEL_AMAX_STRIDE * ((blockIdx.x + blockIdx.y + blockIdx.z) & (RESEL_AMAX_SUBSLOTS - 1));
        const unsigned long long mine = ((unsigned long long)o.epoch << 32) | __float_as_uint(m);
        if (*(volatile unsigned long long*)w < mine) atomicMax(w, mine);
    }
}
// the magnitude a handle holds (all lanes of the calling wave get it)
__device__ __forceinline__ float amax_read(const float* handle) {
    const unsigned long long* h = reinterpret_cast<const unsigned long long*>(handle);
    const int j = __lane_id() & (RESEL_AMAX_SUBSLOTS - 1);
    unsigned long long v = h[RESEL_AMAX_STRIDE * j];
#pragma unroll
    for (int s = RESEL_AMAX_SUBSLOTS / 2; s > 0; s >>= 1) {
        const unsigned long long o = __shfl_xor(v, s, 64);
        v = o > v ? o : v;                       // newest epoch first, then the larger magnitude
    }
    return __uint_as_float((unsigned)(v & 0xffffffffu));
}

inline int launch_status() { return hipGetLastError() == hipSuccess ? RESEL_OK : RESEL_ELAUNCH; }
// device word added to the `offset` of every counter-keyed dropout mask (resel_dropout_offset_base; nullptr: none)
const unsigned long long* dropout_offset_base();
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// gemm_any.hip: the shapes the matrix-core GEMM editions do not take (unaligned rows, tiny reductions) and the M <= 8 rollout rows
// rnd bits: 1 round A to bf16, 2 round B (and the bias) to bf16, 4 round the fp32 result to a bf16 value, 8 store bf16, 16 A is bf16 (rows form)
size_t gemm_any_workspace_bytes(int M, int N, int K, int batch);
bool gemm_any_rows_ok(const void* A, int64_t lda, int64_t strideA, int a_kcontig, int a_bf16, const float* B, int64_t ldb, int64_t strideB,
                      int b_kcontig, int M, int K, int act);
int gemm_any_launch(const void* A, int64_t lda, int64_t strideA, int a_kcontig, const float* B, int64_t ldb, int64_t strideB, int b_kcontig,
                    const float* bias, int64_t strideBias, int act, void* C, int64_t ldc, int64_t strideC, void* workspace,
                    int M, int N, int K, int batch, int rnd, unsigned long long* amax_c, unsigned amax_epoch, hipStream_t s);

}  // namespace resel
