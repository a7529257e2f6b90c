// Library identification.
#include "resel_common.h"
#include <cstdint>

extern "C" int resel_abi_version(void) { return 8; }

// ---- dropout offset base: a device word that every counter-keyed mask kernel (resel_dropout, resel_gelu_dropout_*, resel_attn_varlen_*
// with p_drop > 0) adds to its `offset` argument when it RUNS.  A captured update bakes the host-drawn offsets into its kernel nodes; with
// the base advanced by a node of the same graph every replay draws fresh masks (forward and backward of one replay read the same value).
namespace { const unsigned long long* g_drop_base = nullptr; }
namespace resel { const unsigned long long* dropout_offset_base() { return g_drop_base; } }
extern "C" int resel_dropout_offset_base(const void* base) {
    if (base && ((uintptr_t)base & 7)) return RESEL_EINVAL;
    g_drop_base = (const unsigned long long*)base;
    return RESEL_OK;
}
extern "C" const char* resel_build_info(void) { return "resel_hip gfx950 (CDNA4, wave64) built " __DATE__ " " __TIME__; }

// ---- per-dispatch timing registry (see resel_common.h launch_timed) ----
#include <utility>
#include <vector>
namespace resel {
namespace {
bool g_prof_enabled = false;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_ev[RESEL_PROF_NSLOTS];
}
bool prof_on() { return g_prof_enabled; }
void prof_push(int slot, hipEvent_t a, hipEvent_t b) {
    if (slot >= 0 && slot < RESEL_PROF_NSLOTS) g_prof_ev[slot].emplace_back(a, b);
}
}  // namespace resel

extern "C" int resel_profile_enable(int on) {
    resel::g_prof_enabled = on != 0;
    return RESEL_OK;
}

extern "C" int resel_profile_collect(int kernel_id, double* total_us, int* launches) {
    if (kernel_id < 0 || kernel_id >= RESEL_PROF_NSLOTS || !total_us || !launches) return RESEL_EINVAL;
    double tot = 0.0;
    int n = 0;
    for (auto& pr : resel::g_prof_ev[kernel_id]) {
        float ms = 0.f;
        if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
            tot += 1e3 * ms;
            ++n;
        }
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    resel::g_prof_ev[kernel_id].clear();
    *total_us = tot;
    *launches = n;
    return RESEL_OK;
}

// ---- resel_place_blocks: out [rows, cols] = zeros with up to 8 source blocks copied in at (r0, c0) - ONE launch for what the input encoders of
// the policy / value networks assembled with block_diag + cat + pad (the block-diagonal weight of the merged encoder GEMM, its bias, the
// concatenated zero-padded input rows: ~10 ATen launches per call, six calls per update).
namespace {
struct PlaceBlock { const float* src; long long ld; int r0, nr, c0, nc; };
struct PlaceArgs { PlaceBlock b[8]; int n; };
__global__ __launch_bounds__(256) void place_blocks_kernel(float* __restrict__ out, long long ld_out, long long total, int cols, PlaceArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (k < a.n) {
            const int rr = r - a.b[k].r0, cc = c - a.b[k].c0;
            if ((unsigned)rr < (unsigned)a.b[k].nr && (unsigned)cc < (unsigned)a.b[k].nc) v = a.b[k].src[(long long)rr * a.b[k].ld + cc];
        }
    }
    out[(long long)r * ld_out + c] = v;
}
}  // namespace

extern "C" int resel_place_blocks(float* out, int64_t ld_out, int rows, int cols, int nblk, const float* const* src, const int64_t* ld_src,
                                  const int* r0, const int* nr, const int* c0, const int* nc, resel_stream_t stream) {
    if (!out || rows <= 0 || cols <= 0 || ld_out < cols || nblk < 0 || nblk > 8 || (nblk && (!src || !ld_src || !r0 || !nr || !c0 || !nc))) return RESEL_EINVAL;
    PlaceArgs a{};
    a.n = nblk;
    for (int k = 0; k < nblk; ++k) {
        if (!src[k] || nr[k] <= 0 || nc[k] <= 0 || r0[k] < 0 || c0[k] < 0 || r0[k] + nr[k] > rows || c0[k] + nc[k] > cols || ld_src[k] < nc[k]) return RESEL_EINVAL;
        a.b[k] = PlaceBlock{src[k], (long long)ld_src[k], r0[k], nr[k], c0[k], nc[k]};
    }
    const long long total = (long long)rows * cols;
    hipLaunchKernelGGL(place_blocks_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, (long long)ld_out, total, cols, a);
    return resel::launch_status();
}
