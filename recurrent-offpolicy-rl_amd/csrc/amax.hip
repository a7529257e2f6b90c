// Largest magnitude of a GEMM operand, on the device, written into a magnitude handle (resel_common.h): the scale source of
// resel_gemm_f32x's mode 2 (fp16 planes of the SCALED operand) for operands whose producer published nothing.  HBM-bound single pass: float4 loads along the contiguous axis, one partial per block, and the block whose
// ticket is the last one folds the partials in a fixed order and resets the ticket - the result does not depend on block
// timing, needs no pre-zeroed output and no host synchronisation; `state` ([0] ticket, [1..] partials) must be zero before
// its first use and is left zeroed.  NaNs in the operand are ignored here (fmaxf); they reach the product through the planes.
#include "resel_common.h"
#include <algorithm>

namespace {
using namespace resel;
constexpr int AMAX_BLOCKS = 1024;

struct AmaxParams {
    const float* x;
    int64_t ld, stride;
    int rows, cols, batch;
    unsigned long long* out;
    unsigned epoch;
    unsigned* ticket;
    float* partial;
};

__global__ __launch_bounds__(256) void amax_kernel(AmaxParams p) {
    __shared__ float s_m[4];
    __shared__ int s_last;
    const int c4 = p.cols >> 2;                                  // float4 per row (cols % 4 == 0)
    const int64_t per_b = (int64_t)p.rows * c4, total = per_b * p.batch;
    float m0 = 0.f, m1 = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / per_b, r = (i - b * per_b) / c4, c = i - b * per_b - r * c4;
        const float4 v = ld4(p.x + b * p.stride + r * p.ld + 4 * c);
        m0 = fmaxf(m0, fmaxf(fabsf(v.x), fabsf(v.y)));
        m1 = fmaxf(m1, fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    float m = fmaxf(m0, m1);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
        __hip_atomic_store(&p.partial[blockIdx.x], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == gridDim.x - 1);
    }
    __syncthreads();
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    m = 0.f;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) m = fmaxf(m, __hip_atomic_load(&p.partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        // sub-slot 0 of the handle, stamped with this call's epoch (the other seven keep older epochs: readers ignore them)
        p.out[0] = ((unsigned long long)p.epoch << 32) | __float_as_uint(fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3])));
        __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
}  // namespace

// ---- magnitudes of ALL tensors of a flat parameter buffer in one launch: segment g = flat[begin[g], begin[g] + len[g]) publishes into the
// handle at handles + g * RESEL_AMAX_STRIDE * RESEL_AMAX_SUBSLOTS words.  grid (64, nseg): block (c, g) covers every 64th 256-element piece of g
// (first version: 8 blocks per segment, one load in flight per thread - 73 us per call on the 3 MB ensemble weights).
namespace {
__global__ __launch_bounds__(256) void amax_segments_kernel(const float* __restrict__ flat, const int64_t* __restrict__ begin,
                                                            const int64_t* __restrict__ len, unsigned long long* handles, unsigned epoch) {
    const int g = blockIdx.y;
    const int64_t b = begin[g], n = len[g];
    if ((int64_t)blockIdx.x * 256 >= n) return;                        // short segments (biases): one block does it
    const float* x = flat + b;
    const int64_t st = (int64_t)gridDim.x * 256;
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;                       // four loads in flight per thread
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * st < n; i += 4 * st) {
        m0 = fmaxf(m0, fabsf(x[i])); m1 = fmaxf(m1, fabsf(x[i + st])); m2 = fmaxf(m2, fabsf(x[i + 2 * st])); m3 = fmaxf(m3, fabsf(x[i + 3 * st]));
    }
    for (; i < n; i += st) m0 = fmaxf(m0, fabsf(x[i]));
    amax_publish_wave(fmaxf(fmaxf(m0, m1), fmaxf(m2, m3)), AmaxOut{handles + (int64_t)g * RESEL_AMAX_STRIDE * RESEL_AMAX_SUBSLOTS, epoch});
}
}  // namespace

extern "C" int resel_amax_segments(const float* flat, const int64_t* begin, const int64_t* len, int nseg, void* handles, unsigned epoch,
                                   resel_stream_t stream) {
    if (!flat || !begin || !len || nseg <= 0 || !handles || (reinterpret_cast<uintptr_t>(handles) & 7u)) return RESEL_EINVAL;
    hipLaunchKernelGGL(amax_segments_kernel, dim3(64, nseg), dim3(256), 0, (hipStream_t)stream, flat, begin, len, (unsigned long long*)handles, epoch);
    return launch_status();
}

extern "C" size_t resel_amax_state_bytes(void) { return (size_t)(AMAX_BLOCKS + 4) * sizeof(float); }

extern "C" int resel_amax(const float* x, int64_t ld, int64_t stride, int rows, int cols, int batch, void* out, unsigned epoch, void* state,
                          resel_stream_t stream) {
    if (!x || !out || !state || rows <= 0 || cols <= 0 || batch <= 0 || (cols & 3) || (ld & 3) || (stride & 3) || !aligned16(x) || !aligned16(state)
        || (reinterpret_cast<uintptr_t>(out) & 7u))
        return RESEL_EINVAL;
    const int64_t total = (int64_t)rows * (cols >> 2) * batch;
    const int blocks = (int)std::min<int64_t>(AMAX_BLOCKS, (total + 1023) / 1024 > 0 ? (total + 1023) / 1024 : 1);
    AmaxParams p{x, ld, stride, rows, cols, batch, (unsigned long long*)out, epoch, (unsigned*)state, (float*)state + 4};
    hipLaunchKernelGGL(amax_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    return launch_status();
}
