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
                                                      float rp, uint64_t seed, uint64_t offset, const unsigned long long* obase) {
    const uint32_t key = stream_key(seed, offset + (obase ? *obase : 0ull));
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

// FFN hidden of the cgpt block: y = dropout(gelu(x)) in ONE pass (reference TransformerFlashAttention.py:46-53: `nn.GELU()` then
// `nn.Dropout`; erf form), and its backward dx = dy * keep / (1 - p) * gelu'(x) from the saved pre-activation x - the mask is
// regenerated from the counter, never stored.  BWD = false: (x) -> y;  BWD = true: (x, dy) -> dx.  thr16 = 65536 keeps everything.
template <bool BWD>
__device__ __forceinline__ float gelu_or_grad(float x, float g) {
    const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
    if (!BWD) return x * cdf;
    const float pdf = __expf(-0.5f * x * x) * 0.39894228040143267794f;
    return g * (cdf + x * pdf);
}
template <bool BWD>
__global__ __launch_bounds__(256) void gelu_dropout_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ out,
                                                           int64_t n, uint32_t thr16, float rp, uint64_t seed, uint64_t offset, const unsigned long long* obase, AmaxOut amax) {
    const uint32_t key = stream_key(seed, offset + (obase ? *obase : 0ull));
    const int64_t n4 = n >> 2;
    float omax = 0.f;                                // max |out| of this thread's stores (the fc2 / fc1-gradient GEMMs scale their operand with it)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t w0 = mix32(((uint32_t)(2 * i) * 0x9E3779B1u) ^ key), w1 = mix32(((uint32_t)(2 * i + 1) * 0x9E3779B1u) ^ key);
        const float4 v = ld4(x + 4 * i);
        float4 g = make_float4(1.f, 1.f, 1.f, 1.f);
        if (BWD) g = ld4(dy + 4 * i);
        float4 o;
        o.x = (w0 & 0xffffu) < thr16 ? gelu_or_grad<BWD>(v.x, g.x) * rp : 0.f;
        o.y = (w0 >> 16) < thr16 ? gelu_or_grad<BWD>(v.y, g.y) * rp : 0.f;
        o.z = (w1 & 0xffffu) < thr16 ? gelu_or_grad<BWD>(v.z, g.z) * rp : 0.f;
        o.w = (w1 >> 16) < thr16 ? gelu_or_grad<BWD>(v.w, g.w) * rp : 0.f;
        st4(out + 4 * i, o);
        omax = amax4(omax, o);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {                 // ragged tail
        const int64_t i = (n4 << 2) + threadIdx.x;
        const uint32_t w = mix32(((uint32_t)(i >> 1) * 0x9E3779B1u) ^ key);
        out[i] = ((i & 1) ? (w >> 16) : (w & 0xffffu)) < thr16 ? gelu_or_grad<BWD>(x[i], BWD ? dy[i] : 1.f) * rp : 0.f;
        omax = fmaxf(omax, __builtin_fabsf(out[i]));
    }
    amax_publish_wave(omax, amax);
}

}  // namespace

extern "C" int resel_gelu_dropout_fwd(const float* x, float* y, int64_t n, float p_drop, uint64_t seed, uint64_t offset,
                                      void* amax_y, unsigned amax_epoch, resel_stream_t stream) {
    if (!x || !y || n < 0 || !(p_drop >= 0.f && p_drop < 1.f) || !aligned16(x) || !aligned16(y) || (reinterpret_cast<uintptr_t>(amax_y) & 7u)) return RESEL_EINVAL;
    if (n == 0) return RESEL_OK;
    const uint32_t thr16 = (uint32_t)lrintf((1.f - p_drop) * 65536.f);
    const int64_t n4 = (n + 3) >> 2;
    const int blocks = (int)(n4 + 255) / 256 < 2048 ? (int)((n4 + 255) / 256) : 2048;
    hipLaunchKernelGGL(gelu_dropout_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (const float*)nullptr, y, n, thr16,
                       1.f / (1.f - p_drop), seed, offset, resel::dropout_offset_base(), AmaxOut{(unsigned long long*)amax_y, amax_epoch});
    return launch_status();
}

extern "C" int resel_gelu_dropout_bwd(const float* x, const float* dy, float* dx, int64_t n, float p_drop, uint64_t seed, uint64_t offset,
                                      void* amax_dx, unsigned amax_epoch, resel_stream_t stream) {
    if (!x || !dy || !dx || n < 0 || !(p_drop >= 0.f && p_drop < 1.f) || !aligned16(x) || !aligned16(dy) || !aligned16(dx)
        || (reinterpret_cast<uintptr_t>(amax_dx) & 7u)) return RESEL_EINVAL;
    if (n == 0) return RESEL_OK;
    const uint32_t thr16 = (uint32_t)lrintf((1.f - p_drop) * 65536.f);
    const int64_t n4 = (n + 3) >> 2;
    const int blocks = (int)(n4 + 255) / 256 < 2048 ? (int)((n4 + 255) / 256) : 2048;
    hipLaunchKernelGGL(gelu_dropout_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, dy, dx, n, thr16, 1.f / (1.f - p_drop), seed,
                       offset, resel::dropout_offset_base(), AmaxOut{(unsigned long long*)amax_dx, amax_epoch});
    return launch_status();
}

extern "C" int resel_dropout(const float* x, float* y, int64_t n, float p_drop, uint64_t seed, uint64_t offset, resel_stream_t stream) {
    if (!x || !y || n < 0 || !(p_drop >= 0.f && p_drop < 1.f) || !aligned16(x) || !aligned16(y)) return RESEL_EINVAL;
    if (n == 0) return RESEL_OK;
    const uint32_t thr16 = (uint32_t)lrintf((1.f - p_drop) * 65536.f);
    const int64_t n4 = (n + 3) >> 2;
    const int blocks = (int)(n4 + 255) / 256 < 2048 ? (int)((n4 + 255) / 256) : 2048;
    hipLaunchKernelGGL(dropout_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, n, thr16, 1.f / (1.f - p_drop), seed, offset, resel::dropout_offset_base());
    return launch_status();
}
