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
