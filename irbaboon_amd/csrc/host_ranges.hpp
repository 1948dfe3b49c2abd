// host_ranges.hpp -- how the one-shot functions (fdl_engine.hip, deconv_engine.hip) reach the caller's planar host buffers from the device.
//
// Round 5 rule (profiles/NOTES.md, "Two wrong one-shot results"): PAGEABLE caller memory is never registered by default.  The zero-copy
// ("direct") form of a one-shot call is taken only when every buffer it would read or write already IS HIP host memory -- pages the
// runtime allocated and pinned (hipHostMalloc, irb_host_alloc) or that the caller registered itself -- or when the caller opts in
// (IRB_OFFLINE_DIRECT=1, documented with its exposure in INTEGRATION.md).  Everything else goes through the library's own pinned staging.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

namespace {

// IRB_OFFLINE_DIRECT: unset / "auto" = direct only for HIP host memory; 1 = also register pageable buffers for the call (opt-in); 0 = never direct
enum class DirectMode { kNever = 0, kRegister = 1, kPinnedOnly = 2 };
inline DirectMode direct_mode_from_env()
{
    const char *v = std::getenv("IRB_OFFLINE_DIRECT");
    if (!v || !*v || !std::strcmp(v, "auto")) return DirectMode::kPinnedOnly;
    return std::atoi(v) != 0 ? DirectMode::kRegister : DirectMode::kNever;
}

// [p, p + bytes) lies in ONE HIP host allocation (pinned by hipHostMalloc, or registered by the caller): its device address.
// A pageable pointer makes the runtime's look-up fail -- that error is consumed here, not left for the next launch check.
inline bool hip_host_span(const void *p, size_t bytes, char **dev)
{
    if (!p || bytes == 0) return false;
    hipPointerAttribute_t a0, a1;
    if (hipPointerGetAttributes(&a0, p) != hipSuccess) { (void) hipGetLastError(); return false; }
    if (a0.type != hipMemoryTypeHost || !a0.devicePointer) return false;
    const char *last = static_cast<const char *>(p) + bytes - 1;
    if (hipPointerGetAttributes(&a1, last) != hipSuccess) { (void) hipGetLastError(); return false; }
    if (a1.type != hipMemoryTypeHost || !a1.devicePointer) return false;
    if (static_cast<char *>(a1.devicePointer) - static_cast<char *>(a0.devicePointer) != (ptrdiff_t) (bytes - 1)) return false;   // two allocations that happen to be neighbours
    *dev = static_cast<char *>(a0.devicePointer);
    return true;
}

// The caller's planar buffers made addressable by the GPU for the duration of one call.  Spans that are HIP host memory already are
// used as they are; the others are hipHostRegister'ed ONLY if `allow_register` (1-2 us per range, tools/pin_probe.hip): channels that
// share pages (one allocation, as in a juce::AudioBuffer or a 2-D numpy array) as one range.
struct HostRanges {
    struct Range { const char *host; size_t len; char *dev; bool registered; };
    struct Span { const char *p; size_t bytes; };
    std::vector<Span> spans;                                       // collected by add()
    std::vector<Range> r;                                          // what commit() resolved / registered
    int n_registered = 0;
    void add(const float *const *chan, int nch, size_t floats)
    {
        for (int c = 0; c < nch; c++)
            if (floats > 0) spans.push_back({reinterpret_cast<const char *>(chan[c]), floats * sizeof(float)});
    }
    // true: every span has a device address.  Without allow_register the first pageable span ends the attempt (one runtime look-up
    // for the usual pageable caller).
    bool commit(bool allow_register)
    {
        constexpr uintptr_t PG = 4096;
        std::vector<std::pair<uintptr_t, uintptr_t>> hull;         // page hulls of the spans that need a registration
        for (const Span &s : spans) {
            char *d = nullptr;
            if (hip_host_span(s.p, s.bytes, &d)) { r.push_back({s.p, s.bytes, d, false}); continue; }
            if (!allow_register) return false;
            const uintptr_t b = reinterpret_cast<uintptr_t>(s.p);
            hull.emplace_back(b / PG * PG, (b + s.bytes + PG - 1) / PG * PG);
        }
        // hulls that touch or overlap (channels of one allocation; an audio buffer and a result buffer that share a page) become ONE
        // registration: the runtime refuses ranges that overlap a registered one
        std::sort(hull.begin(), hull.end());
        for (size_t i = 0; i < hull.size();) {
            uintptr_t lo = hull[i].first, hi = hull[i].second;
            size_t k = i + 1;
            while (k < hull.size() && hull[k].first <= hi) { hi = std::max(hi, hull[k].second); k++; }
            void *d = nullptr;
            if (hipHostRegister(reinterpret_cast<void *>(lo), hi - lo, hipHostRegisterMapped) != hipSuccess) { (void) hipGetLastError(); return false; }
            r.push_back({reinterpret_cast<const char *>(lo), (size_t) (hi - lo), nullptr, true});
            n_registered++;
            if (hipHostGetDevicePointer(&d, reinterpret_cast<void *>(lo), 0) != hipSuccess) { (void) hipGetLastError(); return false; }
            r.back().dev = static_cast<char *>(d);
            i = k;
        }
        return true;
    }
    float *dev(const float *p) const
    {
        const char *c = reinterpret_cast<const char *>(p);
        for (const Range &q : r) if (c >= q.host && c < q.host + q.len) return reinterpret_cast<float *>(q.dev + (c - q.host));
        return nullptr;
    }
    void release()
    {
        for (const Range &q : r)
            if (q.registered && hipHostUnregister(const_cast<char *>(q.host)) != hipSuccess) (void) hipGetLastError();   // (a refused call must not leave its error for the next launch check)
        r.clear();
        spans.clear();
        n_registered = 0;
    }
    ~HostRanges() { release(); }
};

}  // namespace
