// vaa_capi.hip — error plumbing and device probing for libvaa_hip.so (include/vaa.h).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "vaa_common.h"

namespace vaa {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- asynchronous device-side failures (vaa_async_error) ----
// One word of pinned, device-mapped host memory per process: a kernel that has to give up (today: the one-launch K3's grid-wide hand-over
// running out of polls) ORs a bit into it through its device alias; the host reads it without synchronising — in every check_launch() (peek) and in
// vaa_async_error() (report + clear) — so the failure surfaces as VAA_E_LAUNCH on every library call after the kernel ran, never as a silent NaN.
static std::atomic<unsigned*> g_async_word{nullptr};
static std::mutex g_async_mu;

unsigned* async_error_word() {
    unsigned* w = g_async_word.load(std::memory_order_acquire);
    if (w) return w;
    std::lock_guard<std::mutex> lk(g_async_mu);
    w = g_async_word.load(std::memory_order_acquire);
    if (w) return w;
    void* p = nullptr;
    if (hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable) != hipSuccess || !p) {
        (void)hipGetLastError();
        return nullptr;
    }
    memset(p, 0, 64);
    g_async_word.store(reinterpret_cast<unsigned*>(p), std::memory_order_release);
    return reinterpret_cast<unsigned*>(p);
}

// VAA_OK, or VAA_E_LAUNCH (+ message) while a failure is recorded. The word is STICKY: every check_launch() of every library call reads it
// without clearing (so the call whose outputs were poisoned, and every call after it, fails — on whichever stream or thread it runs); only the
// explicit poll vaa_async_error() (clear = true) reports AND clears it, so the attack loops' once-per-outer-iteration check always sees it.
int async_error_poll(const char* what, bool clear) {
    unsigned* w = g_async_word.load(std::memory_order_acquire);
    if (!w) return VAA_OK;
    const unsigned bits = clear ? __atomic_exchange_n(w, 0u, __ATOMIC_ACQ_REL) : __atomic_load_n(w, __ATOMIC_ACQUIRE);
    if (bits == 0u) return VAA_OK;
    set_error("%s: a kernel of this process reported a device-side failure (bits 0x%x%s): its outputs are NaN-poisoned%s", what, bits,
              (bits & VAA_ASYNC_K3_HANDOVER_TIMEOUT) ? ": the one-launch K3 hand-over timed out — unset VAA_K3_ONE_PASS" : "",
              clear ? "" : "; library calls keep failing until vaa_async_error() is polled");
    return VAA_E_LAUNCH;
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return VAA_E_LAUNCH;
    }
    return async_error_poll(what, false);
}

// ---- per-dispatch profiler (vaa_prof_*) ----
struct ProfRec {
    hipEvent_t start, stop;
    const char* name;
};
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof;   // pool of event pairs; the first g_prof_used are recorded
static int g_prof_used = 0;
static int g_prof_cap = 0;             // the capacity of the CURRENT arming (the pool itself only grows)
static std::atomic<bool> g_prof_armed{false};  // read without the lock on every launch

bool prof_next(const char* name, hipEvent_t* start, hipEvent_t* stop) {
    if (!g_prof_armed.load(std::memory_order_acquire)) return false;  // fast path of every un-profiled launch
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_armed || g_prof_used >= g_prof_cap) return false;  // capacity reached: later dispatches run unprofiled
    ProfRec& r = g_prof[g_prof_used++];
    r.name = name;
    *start = r.start;
    *stop = r.stop;
    return true;
}

}  // namespace vaa

extern "C" {

int vaa_prof_start(int capacity) {
    using namespace vaa;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (capacity < 0) { set_error("vaa_prof_start: negative capacity"); return VAA_E_INVALID; }
    while ((int)g_prof.size() < capacity) {
        ProfRec r;
        r.name = "";
        if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) {
            set_error("vaa_prof_start: hipEventCreate failed");
            (void)hipGetLastError();
            return VAA_E_LAUNCH;
        }
        g_prof.push_back(r);
    }
    g_prof_used = 0;
    g_prof_cap = capacity;
    g_prof_armed = capacity > 0;
    return VAA_OK;
}

int vaa_prof_stop(void) {
    std::lock_guard<std::mutex> lk(vaa::g_prof_mu);
    vaa::g_prof_armed = false;
    return vaa::g_prof_used;
}

int vaa_prof_get(int i, const char** name, float* usec) {
    using namespace vaa;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (i < 0 || i >= g_prof_used || !usec) { set_error("vaa_prof_get: record %d out of range (%d recorded)", i, g_prof_used); return VAA_E_INVALID; }
    float ms = 0.0f;
    hipError_t e = hipEventSynchronize(g_prof[i].stop);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, g_prof[i].start, g_prof[i].stop);
    if (e != hipSuccess) {
        set_error("vaa_prof_get: %s", hipGetErrorString(e));
        (void)hipGetLastError();
        return VAA_E_LAUNCH;
    }
    if (name) *name = g_prof[i].name;
    *usec = ms * 1000.0f;
    return VAA_OK;
}

const char* vaa_last_error(void) { return vaa::g_err; }

int vaa_async_error(void) { return vaa::async_error_poll("vaa_async_error", true); }

int vaa_version(void) { return 100; /* 0.1.0 */ }

int vaa_device_check(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        vaa::set_error("no HIP device visible");
        (void)hipGetLastError();
        return VAA_E_NODEVICE;
    }
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
        vaa::set_error("hipGetDeviceProperties failed");
        return VAA_E_NODEVICE;
    }
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
        vaa::set_error("device %d is %s; libvaa_hip.so is built for gfx950 only", dev, p.gcnArchName);
        return VAA_E_NODEVICE;
    }
    return VAA_OK;
}

}  // extern "C"
