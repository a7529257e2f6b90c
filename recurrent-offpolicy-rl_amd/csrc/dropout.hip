// Element-wise dropout with a counter-keyed keep mask (HBM-bound, one pass; the backward is the same pass over dy).
// Used for the residual / FFN dropouts of the cgpt decoder block (reference TransformerFlashAttention.py:46-53,72,84-85:
// nn.Dropout) so that EVERY random mask of a training-mode cgpt pass is a pure function of (seed, offset, element index):
// the CPU oracle restates the same function (oracle/kernels.py `dropout_keep`) and p > 0 passes compare element for element.
//     word(i >> 1) = mix32(((i >> 1) * 0x9E3779B1) ^ key(seed, offset));   keep(i) = 16-bit half (i & 1) of it < thr16,
//     thr16 = round((1 - p) * 65536);  y = keep ? x / (1 - p) : 0.
#include "resel_common.h"

namespace {
using namespace resel;

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t stream_key(uint64_t seed, uint64_t offset) {
    uint32_t x = mix32(0xC2B2AE3Du ^ (uint32_t)(offset >> 32));
    x = mix32(x ^ (uint32_t)offset);
    x = mix32(x ^ (uint32_t)(seed >> 32));
    return mix32(x ^ (uint32_t)seed);
}

__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, uint32_t thr16,
                                                      float rp, uint64_t seed, uint64_t offset) {
    const uint32_t key = stream_key(seed, offset);
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t w0 = mix32(((uint32_t)(2 * i) * 0x9E3779B1u) ^ key), w1 = mix32(((uint32_t)(2 * i + 1) * 0x9E3779B1u) ^ key);
        float4 v = ld4(x + 4 * i);
        v.x = (w0 & 0xffffu) < thr16 ? v.x * rp : 0.f;
        v.y = (w0 >> 16) < thr16 ? v.y * rp : 0.f;
        v.z = (w1 & 0xffffu) < thr16 ? v.z * rp : 0.f;
        v.w = (w1 >> 16) < thr16 ? v.w * rp : 0.f;
        st4(y + 4 * i, v);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {                 // ragged tail
        const int64_t i = (n4 << 2) + threadIdx.x;
        const uint32_t w = mix32(((uint32_t)(i >> 1) * 0x9E3779B1u) ^ key);
        y[i] = ((i & 1) ? (w >> 16) : (w & 0xffffu)) < thr16 ? x[i] * rp : 0.f;
    }
}

}  // namespace

extern "C" int resel_dropout(const float* x, float* y, int64_t n, float p_drop, uint64_t seed, uint64_t offset, resel_stream_t stream) {
    if (!x || !y || n < 0 || !(p_drop >= 0.f && p_drop < 1.f) || !aligned16(x) || !aligned16(y)) return RESEL_EINVAL;
    if (n == 0) return RESEL_OK;
    const uint32_t thr16 = (uint32_t)lrintf((1.f - p_drop) * 65536.f);
    const int64_t n4 = (n + 3) >> 2;
    const int blocks = (int)(n4 + 255) / 256 < 2048 ? (int)((n4 + 255) / 256) : 2048;
    hipLaunchKernelGGL(dropout_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, n, thr16, 1.f / (1.f - p_drop), seed, offset);
    return launch_status();
}
