// Packed full-trajectory batch assembled ON THE DEVICE from a device-resident replay ring.
//
// The host keeps the trajectory bookkeeping and the sampling RNG (reference nested_replay_memory.py:103-185 - the numpy
// call order is contractual) and ships only a PLAN: one int4 per sampled trajectory = (batch row, first slot, length
// including the `skip` leading slots, first transition index in the ring).  Three HBM-bound passes then build exactly
// the array the host path builds (bit-exact: it is all copies and flag writes):
//   init     out[r, t, :] = 0, start = 1                      (padding is "start" everywhere)
//   segments slots [pos, pos+skip-1): start = 1; slot pos+skip-1 (PRE-STEP): next_state <- s_0, reward <- r_in0,
//            state <- last_state_0 (column pairs passed by the host), start = 1; slots [pos+skip, pos+n): the transitions,
//            validity column W <- their mask
//   flags    column W+1 = validity extended one slot earlier, column W+2 = start with the last 1 before data cleared
//            (the target pass's flags, reference sac_full_length_rnn_ensembleQ.py:338-342), done <- 0 where timeout > 0
// One float per thread along the contiguous column axis: every wave access is a fully used row segment.
#include "resel_common.h"

namespace {
using namespace resel;

__global__ void gather_init_kernel(float* __restrict__ out, int64_t ntok, int WO, int c_start) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ntok * WO) return;
    out[i] = (int)(i % WO) == c_start ? 1.f : 0.f;
}

// grid = (ceil(maxlen * W / 256), nseg)
__global__ void gather_segments_kernel(const float* __restrict__ buffer, int W, int64_t capacity, const int* __restrict__ seg, int skip,
                                       int rows, int Tp, int c_mask, int c_start, const int* __restrict__ pre_pairs, int npairs,
                                       float* __restrict__ out) {
    const int4 sg = reinterpret_cast<const int4*>(seg)[blockIdx.y];       // row, pos, n, first transition
    // a plan entry that does not fit the output / the ring is dropped (its slots stay padding) instead of written out of bounds
    if (sg.x < 0 || sg.x >= rows || sg.y < 0 || sg.z < skip || (int64_t)sg.y + sg.z > Tp || sg.w < 0 ||
        (int64_t)sg.w + (sg.z - skip) > capacity || (sg.z > skip ? false : sg.w >= capacity)) return;
    const int WO = W + 3;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= sg.z * W) return;
    const int i = e / W, col = e % W;
    float* dst = out + ((int64_t)sg.x * Tp + sg.y + i) * WO;
    if (i >= skip) {
        const float v = buffer[(int64_t)(sg.w + i - skip) * W + col];
        dst[col] = v;
        if (col == c_mask) dst[W] = v;
    } else if (i == skip - 1) {
        float v = col == c_start ? 1.f : 0.f;
        const float* first = buffer + (int64_t)sg.w * W;
        for (int k = 0; k < npairs; ++k)
            if (pre_pairs[2 * k] == col) v = first[pre_pairs[2 * k + 1]];
        dst[col] = v;
    } else if (col == c_start) {
        dst[col] = 1.f;
    }
}

__global__ void gather_flags_kernel(float* __restrict__ out, int rows, int Tp, int W, int c_start, int c_done, int c_timeout) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * Tp) return;
    const int t = (int)(i % Tp);
    const int WO = W + 3;
    float* p = out + i * WO;
    const float v = p[W], st = p[c_start];
    float tv = v, ts = st;
    if (t + 1 < Tp) {
        if (p[WO + W] - v == 1.f) tv = 1.f;
        if (p[WO + c_start] - st == -1.f) ts = 0.f;
    }
    p[W + 1] = tv;
    p[W + 2] = ts;
    if (c_timeout >= 0 && p[c_timeout] > 0.f) p[c_done] = 0.f;
}

}  // namespace

extern "C" int resel_gather_trajs(const float* buffer, int W, int64_t capacity, const int* segments, int nseg, int max_len, int skip, int rows, int Tp,
                                  int c_mask, int c_start, int c_done, int c_timeout, const int* pre_pairs, int npairs,
                                  float* out, resel_stream_t stream) {
    if (!buffer || !segments || !out || W <= 0 || capacity <= 0 || nseg <= 0 || max_len <= 0 || skip < 1 || rows <= 0 || Tp <= 0) return RESEL_EINVAL;
    if (c_mask < 0 || c_mask >= W || c_start < 0 || c_start >= W || c_done < 0 || c_done >= W || c_timeout >= W || (npairs > 0 && !pre_pairs))
        return RESEL_EINVAL;
    if (!aligned16(segments)) return RESEL_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int WO = W + 3;
    const int64_t ntok = (int64_t)rows * Tp;
    hipLaunchKernelGGL(gather_init_kernel, dim3((unsigned)((ntok * WO + 255) / 256)), dim3(256), 0, s, out, ntok, WO, c_start);
    hipLaunchKernelGGL(gather_segments_kernel, dim3((unsigned)(((int64_t)max_len * W + 255) / 256), nseg), dim3(256), 0, s, buffer, W,
                       capacity, segments, skip, rows, Tp, c_mask, c_start, pre_pairs, npairs, out);
    hipLaunchKernelGGL(gather_flags_kernel, dim3((unsigned)((ntok + 255) / 256)), dim3(256), 0, s, out, rows, Tp, W, c_start, c_done,
                       c_timeout);
    return launch_status();
}
