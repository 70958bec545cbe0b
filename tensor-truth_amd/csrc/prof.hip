// Optional per-kernel timing with HIP events recorded on the launch stream (thread-local).
#include <vector>

#include "common.h"

namespace {
struct Rec { int id; hipEvent_t e0, e1; };
struct EvPair { hipEvent_t first, second; };     // (a type of this translation unit: its vector instantiations are not dynamic symbols)
struct ProfState {
    bool on = false;
    unsigned mask = ~0u;     // bit id set: kernel id is timed
    std::vector<Rec> recs;
    std::vector<EvPair> pool;
};
thread_local ProfState g_prof;
constexpr size_t kMaxRecs = 1 << 16;
}  // namespace

bool tt_prof_on() { return g_prof.on && g_prof.recs.size() < kMaxRecs; }

void tt_prof_begin(int id, hipStream_t st) {
    Rec r{id, nullptr, nullptr};
    if (!((g_prof.mask >> id) & 1u)) { g_prof.recs.push_back(r); return; }   // placeholder: tt_prof_end pairs with it
    if (!g_prof.pool.empty()) {
        r.e0 = g_prof.pool.back().first;
        r.e1 = g_prof.pool.back().second;
        g_prof.pool.pop_back();
    } else {
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
    }
    (void)hipEventRecord(r.e0, st);
    g_prof.recs.push_back(r);
}

void tt_prof_end(hipStream_t st) {
    if (!g_prof.recs.empty() && g_prof.recs.back().e1) (void)hipEventRecord(g_prof.recs.back().e1, st);
    if (!g_prof.recs.empty() && !g_prof.recs.back().e0) g_prof.recs.pop_back();
}

extern "C" int tt_prof_enable(int on) {
    for (auto& r : g_prof.recs) g_prof.pool.push_back(EvPair{r.e0, r.e1});
    g_prof.recs.clear();
    g_prof.on = on != 0;
    g_prof.mask = (on == 0 || on == 1) ? ~0u : (unsigned)on;   // on > 1: bit mask of kernel ids (1 << id)
    return TT_OK;
}

extern "C" int tt_prof_read(int which, double* total_ms_host, int* launches_host) {
    double tot = 0.0;
    int n = 0;
    for (auto& r : g_prof.recs) {
        if (r.id != which) continue;
        if (hipEventSynchronize(r.e1) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            tot += ms;
            ++n;
        }
    }
    if (total_ms_host) *total_ms_host = tot;
    if (launches_host) *launches_host = n;
    return TT_OK;
}
