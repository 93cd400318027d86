// core.hip -- version / error plumbing of the C ABI.
#include "common.h"
#include <string.h>
#include <atomic>
#include <mutex>

static thread_local char g_err[512] = "";

void tdgp_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int tdgp_cu_count() {
    static std::atomic<int> cache[256];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::atomic<int>& c = cache[dev & 255];
    int v = c.load(std::memory_order_relaxed);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) v = 256;
        c.store(v, std::memory_order_relaxed);
    }
    return v;
}

int* tdgp_fault_word() {
    static std::atomic<int*> word{nullptr};
    static std::mutex mu;
    int* w = word.load(std::memory_order_acquire);
    if (w) return w;
    std::lock_guard<std::mutex> lk(mu);
    w = word.load(std::memory_order_acquire);
    if (!w) {
        void* hp = nullptr;
        // 64 bytes of pinned, mapped, coherent, PORTABLE (every device of the process sees it) host memory: the one allocation the library makes (not device memory; never freed)
        if (hipHostMalloc(&hp, 64, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess || !hp) { (void)hipGetLastError(); return nullptr; }
        *(volatile int*)hp = 0;
        w = (int*)hp;
        word.store(w, std::memory_order_release);
    }
    return w;
}

TDGP_API int tdgp_device_fault(int clear) {
    int* w = tdgp_fault_word();
    if (!w) return 0;
    return clear ? __atomic_exchange_n(w, 0, __ATOMIC_RELAXED) : __atomic_load_n(w, __ATOMIC_RELAXED);
}

TDGP_API int tdgp_version(void) { return 100; }
TDGP_API const char* tdgp_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------------------------------------
// per-kernel timing
// ---------------------------------------------------------------------------------------------------------
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {
struct ProfEvent { const char* name; hipEvent_t a, b; };
std::mutex g_mu;
std::vector<ProfEvent> g_events;
std::vector<hipEvent_t> g_pool;         // events created ahead of the profiled launches (hipEventCreate per dispatch starved the GPU: ~190 launches per forward)
std::atomic<int> g_on{0};
}  // namespace

bool tdgp_prof_on() { return g_on.load(std::memory_order_relaxed) != 0; }

// A fresh (start, stop) event pair for one launch, registered under `name`; the caller hands both to hipExtLaunchKernelGGL.
void tdgp_prof_events(const char* name, hipEvent_t* a, hipEvent_t* b) {
    ProfEvent e;
    e.name = name;
    std::lock_guard<std::mutex> lk(g_mu);
    auto take = [&]() {
        hipEvent_t ev = nullptr;
        if (!g_pool.empty()) { ev = g_pool.back(); g_pool.pop_back(); }
        else (void)hipEventCreate(&ev);
        return ev;
    };
    e.a = take(); e.b = take();
    *a = e.a; *b = e.b;
    g_events.push_back(e);
}

static void prof_clear() {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& e : g_events) { g_pool.push_back(e.a); g_pool.push_back(e.b); }        // recycled, not destroyed
    g_events.clear();
}

TDGP_API int tdgp_profile_enable(int on) {
    prof_clear();
    if (on) {                               // a pool for ~2 forwards' worth of launches, created before anything is timed
        std::lock_guard<std::mutex> lk(g_mu);
        while (g_pool.size() < 1024) { hipEvent_t ev = nullptr; if (hipEventCreate(&ev) != hipSuccess) break; g_pool.push_back(ev); }
    }
    g_on.store(on ? 1 : 0);
    return TDGP_OK;
}

// Writes one line per kernel: "<name> <launches> <total_ms> <min_ms> <max_ms>\n".  Blocks until the recorded events
// have completed (the only entry point that synchronises).  Returns the number of bytes needed (incl. NUL).
TDGP_API int64_t tdgp_profile_report(char* buf, int64_t cap) {
    struct Acc { int64_t n = 0; double tot = 0, mn = 1e30, mx = 0; };
    std::map<std::string, Acc> acc;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (auto& e : g_events) {
            (void)hipEventSynchronize(e.b);
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, e.a, e.b) != hipSuccess) continue;
            Acc& a = acc[e.name];
            a.n++; a.tot += ms; a.mn = ms < a.mn ? ms : a.mn; a.mx = ms > a.mx ? ms : a.mx;
        }
    }
    std::string out;
    char line[256];
    for (auto& kv : acc) {
        snprintf(line, sizeof(line), "%s %lld %.6f %.6f %.6f\n", kv.first.c_str(), (long long)kv.second.n, kv.second.tot, kv.second.mn, kv.second.mx);
        out += line;
    }
    if (buf && cap > 0) {
        const int64_t n = (int64_t)out.size() < cap - 1 ? (int64_t)out.size() : cap - 1;
        memcpy(buf, out.data(), (size_t)n);
        buf[n] = 0;
    }
    return (int64_t)out.size() + 1;
}
