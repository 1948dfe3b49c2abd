// host_ranges.hpp -- the caller's pageable buffers registered with the HIP runtime for the duration of one call (shared by the one-shot
// functions of fdl_engine.hip and deconv_engine.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <utility>
#include <vector>

namespace {

// The caller's pageable planar buffers made addressable by the GPU for the duration of one call (hipHostRegister: 1-2 us per range
// on this system, tools/pin_probe.hip -- against 70-190 us for copying 4 MiB through pinned staging on one host thread).  Channels
// that share pages (one allocation, as in a juce::AudioBuffer or a 2-D numpy array) are registered as one range.
struct HostRanges {
    struct Range { char *host; size_t len; char *dev; };
    std::vector<std::pair<uintptr_t, uintptr_t>> hull;            // page hulls collected by add()
    std::vector<Range> r;                                          // what commit() registered
    void add(const float *const *chan, int nch, size_t floats)
    {
        constexpr uintptr_t PG = 4096;
        for (int c = 0; c < nch; c++) {
            const uintptr_t b = reinterpret_cast<uintptr_t>(chan[c]);
            if (floats > 0) hull.emplace_back(b / PG * PG, (b + floats * sizeof(float) + PG - 1) / PG * PG);
        }
    }
    // hulls that touch or overlap (channels of one allocation; an audio buffer and a result buffer that share a page) become ONE
    // registration: the runtime refuses ranges that overlap a registered one
    bool commit()
    {
        std::sort(hull.begin(), hull.end());
        for (size_t i = 0; i < hull.size();) {
            uintptr_t lo = hull[i].first, hi = hull[i].second;
            size_t k = i + 1;
            while (k < hull.size() && hull[k].first <= hi) { hi = std::max(hi, hull[k].second); k++; }
            void *d = nullptr;
            if (hipHostRegister(reinterpret_cast<void *>(lo), hi - lo, hipHostRegisterMapped) != hipSuccess) { (void) hipGetLastError(); return false; }
            r.push_back({reinterpret_cast<char *>(lo), (size_t) (hi - lo), nullptr});
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
        for (const Range &q : r) if (hipHostUnregister(q.host) != hipSuccess) (void) hipGetLastError();   // (a refused call must not leave its error for the next launch check)
        r.clear();
        hull.clear();
    }
    ~HostRanges() { release(); }
};

}  // namespace
