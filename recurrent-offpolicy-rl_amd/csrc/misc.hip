// Library identification.
#include "resel_common.h"

extern "C" int resel_abi_version(void) { return 5; }
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
