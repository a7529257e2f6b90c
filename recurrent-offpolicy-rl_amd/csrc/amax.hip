// Largest magnitude of a GEMM operand, on the device, written into a magnitude handle (resel_common.h): the scale source of
// resel_gemm_f32x's mode 2 (fp16 planes of the SCALED operand) for operands whose producer published nothing.  HBM-bound single pass:
// float4 loads along the contiguous axis, one maximum per wave, published like every producer publishes - one conditional 64-bit
// atomicMax of {epoch | float bits} on the sub-slot the block id selects.  A maximum does not depend on the order of its operands, so the
// result is independent of block timing; the call keeps NO state between launches (the first version folded per-block partials behind a
// ticket in a shared state buffer: two pre-passes in flight on different streams then mixed their partials - ADVICE r04).  NaNs in the
// operand are ignored here (fmaxf); they reach the product through the planes.
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
};

__global__ __launch_bounds__(256) void amax_kernel(AmaxParams p) {
    const int c4 = p.cols >> 2;                                  // float4 per row (cols % 4 == 0)
    const int64_t per_b = (int64_t)p.rows * c4, total = per_b * p.batch;
    float m0 = 0.f, m1 = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / per_b, r = (i - b * per_b) / c4, c = i - b * per_b - r * c4;
        const float4 v = ld4(p.x + b * p.stride + r * p.ld + 4 * c);
        m0 = fmaxf(m0, fmaxf(fabsf(v.x), fabsf(v.y)));
        m1 = fmaxf(m1, fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    amax_publish_wave(fmaxf(m0, m1), AmaxOut{p.out, p.epoch});
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

extern "C" size_t resel_amax_state_bytes(void) { return 0; }     // ABI 7: resel_amax keeps no state (the argument is ignored)

extern "C" int resel_amax(const float* x, int64_t ld, int64_t stride, int rows, int cols, int batch, void* out, unsigned epoch, void* state,
                          resel_stream_t stream) {
    (void)state;
    if (!x || !out || rows <= 0 || cols <= 0 || batch <= 0 || (cols & 3) || (ld & 3) || (stride & 3) || !aligned16(x)
        || (reinterpret_cast<uintptr_t>(out) & 7u))
        return RESEL_EINVAL;
    const int64_t total = (int64_t)rows * (cols >> 2) * batch;
    const int blocks = (int)std::min<int64_t>(AMAX_BLOCKS, (total + 1023) / 1024 > 0 ? (total + 1023) / 1024 : 1);
    AmaxParams p{x, ld, stride, rows, cols, batch, (unsigned long long*)out, epoch};
    hipLaunchKernelGGL(amax_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    return launch_status();
}

// ---- verify mode (RESEL_AMAX_VERIFY=1 on the Python side): is the magnitude a handle holds really an upper bound of max |x|?  One pass
// over the operand box of resel_amax; a wave whose maximum exceeds the handle's value reports into `err` (int32 [4], zero = clean):
// [0] number of reporting waves (atomicAdd), [1] float bits of the largest violating magnitude (atomicMax), [2] `tag` of the first
// report (atomicCAS from 0; the caller numbers its GEMM operands from 1), [3] float bits of the bound that first report saw.
namespace {
__global__ __launch_bounds__(256) void amax_check_kernel(AmaxParams p, const float* handle, int* err, int tag) {
    const int c4 = p.cols >> 2;
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
    const float bound = amax_read(handle);
    if (__lane_id() == 0 && m > bound) {
        atomicAdd(&err[0], 1);
        atomicMax(&err[1], (int)__float_as_uint(m));
        if (atomicCAS(&err[2], 0, tag) == 0) err[3] = (int)__float_as_uint(bound);
    }
}
}  // namespace

extern "C" int resel_amax_check(const float* x, int64_t ld, int64_t stride, int rows, int cols, int batch, const void* handle, int* err, int tag,
                                resel_stream_t stream) {
    if (!x || !handle || !err || rows <= 0 || cols <= 0 || batch <= 0 || (cols & 3) || (ld & 3) || (stride & 3) || !aligned16(x)
        || (reinterpret_cast<uintptr_t>(handle) & 7u) || tag == 0)
        return RESEL_EINVAL;
    const int64_t total = (int64_t)rows * (cols >> 2) * batch;
    const int blocks = (int)std::min<int64_t>(AMAX_BLOCKS, (total + 1023) / 1024 > 0 ? (total + 1023) / 1024 : 1);
    AmaxParams p{x, ld, stride, rows, cols, batch, nullptr, 0u};
    hipLaunchKernelGGL(amax_check_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, (const float*)handle, err, tag);
    return launch_status();
}
