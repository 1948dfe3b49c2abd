// tiled_stamps.hip -- DIAGNOSTIC build of the time-tiled MAC (k_cmac_tiled_pk) with per-workgroup stamps: when each
// workgroup starts and ends on the 100 MHz clock, how long its partition loop takes in shader cycles, and which CU it ran on.
// Answers "where do the microseconds between the waves' own work and the kernel's duration go" (launch ramp, rounds,
// stragglers, tail).  Not part of libirbfdl.so.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -DIRB_STAMPS -o tools/tiled_stamps tools/tiled_stamps.hip
//   tools/tiled_stamps T splits
#define IRB_STAMPS 1
#include "../irbaboon_amd/csrc/fdl_kernels.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int T = argc > 1 ? std::atoi(argv[1]) : 3750, splits = argc > 2 ? std::atoi(argv[2]) : 2;
    constexpr int N = 512, M = N / 2, B = 256, P = 375, TT = 16;
    const int Ppad = 384, R = Ppad + T + 8, ntiles = (T + TT - 1) / TT, grid = ntiles * splits;
    float2 *dH, *dX, *dpart;
    int2 *dinfo;
    unsigned long long *dst;
    CK(hipMalloc(&dH, (size_t) M * (Ppad + 8) * sizeof(float2)));
    CK(hipMalloc(&dX, (size_t) M * R * sizeof(float2)));
    CK(hipMalloc(&dpart, (size_t) T * splits * M * sizeof(float2)));
    CK(hipMalloc(&dinfo, sizeof(int2)));
    CK(hipMalloc(&dst, (size_t) grid * 16 * sizeof(unsigned long long)));
    {
        std::vector<float2> h((size_t) M * std::max(Ppad + 8, R));
        unsigned long long z = 12345;
        for (auto &v : h) { z = z * 6364136223846793005ull + 1442695040888963407ull; v.x = (float) ((long long) (z >> 40) - (1 << 23)) / (float) (1 << 23); v.y = -v.x * 0.5f; }
        CK(hipMemcpy(dH, h.data(), (size_t) M * (Ppad + 8) * sizeof(float2), hipMemcpyHostToDevice));
        CK(hipMemcpy(dX, h.data(), (size_t) M * R * sizeof(float2), hipMemcpyHostToDevice));
        const int2 info = make_int2(0, P);
        CK(hipMemcpy(dinfo, &info, sizeof info, hipMemcpyHostToDevice));
    }
    irb::StepArgs a;
    std::memset(&a, 0, sizeof a);
    a.stamps = dst; a.H = dH; a.X = dX; a.stream_info = dinfo; a.part = dpart;
    a.B = B; a.P = Ppad; a.R = R; a.head = Ppad; a.splits = splits; a.T = T; a.nq = splits;
    std::vector<unsigned long long> st((size_t) grid * 16);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int it = 0; it < 6; it++) {
        CK(hipMemset(dst, 0, (size_t) grid * 16 * sizeof(unsigned long long)));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        irb::k_cmac_tiled_pk<N, 16, 8, 2><<<dim3(grid), dim3(irb::kThreads)>>>(a);
        CK(hipEventRecord(e1));
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    CK(hipMemcpy(st.data(), dst, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int g = 0; g < grid; g++) { t0 = std::min(t0, st[g * 16 + 11]); t1 = std::max(t1, st[g * 16 + 12]); }
    std::vector<double> start, dur, loop;
    std::map<unsigned, int> per_cu;
    std::map<unsigned, std::vector<std::pair<double, double>>> cu_iv;
    for (int g = 0; g < grid; g++) {
        const unsigned long long *t = &st[(size_t) g * 16];
        start.push_back((t[11] - t0) / 100.0);
        dur.push_back((t[12] - t[11]) / 100.0);
        loop.push_back((double) (t[1] - t[0]));
        const unsigned hw = (unsigned) t[14], cu = ((unsigned) t[15] & 15u) << 16 | (hw & 0xFF00u) | ((hw >> 13) & 0x7u) << 4;   // XCC, CU_ID[11:8], SE_ID[15:13]
        per_cu[cu]++;
        cu_iv[cu].push_back({(t[11] - t0) / 100.0, (t[12] - t0) / 100.0});
    }
    auto pct = [](std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v[(size_t) (p * (v.size() - 1))]; };
    int maxper = 0, overlap2 = 0;
    for (auto &kv : cu_iv) {
        maxper = std::max(maxper, (int) kv.second.size());
        auto &v = kv.second;
        for (size_t i = 0; i < v.size(); i++) for (size_t j = i + 1; j < v.size(); j++)
            if (std::min(v[i].second, v[j].second) - std::max(v[i].first, v[j].first) > 0.5 * std::min(v[i].second - v[i].first, v[j].second - v[j].first)) overlap2++;
    }
    std::printf("T=%d splits=%d grid=%d: event time %.1f us; first start -> last end %.1f us (100 MHz clock)\n", T, splits, grid, ms * 1e3, (t1 - t0) / 100.0);
    std::printf("  workgroup start (us after the first): p10 %.1f  p50 %.1f  p90 %.1f  max %.1f\n", pct(start, .1), pct(start, .5), pct(start, .9), pct(start, 1));
    std::printf("  workgroup duration (us):               p10 %.1f  p50 %.1f  p90 %.1f  max %.1f\n", pct(dur, .1), pct(dur, .5), pct(dur, .9), pct(dur, 1));
    std::printf("  partition loop (shader cycles):        p10 %.0f  p50 %.0f  p90 %.0f  max %.0f\n", pct(loop, .1), pct(loop, .5), pct(loop, .9), pct(loop, 1));
    std::printf("  distinct CUs used %zu, most workgroups on one CU %d, workgroup pairs on one CU that overlap in time: %d\n", per_cu.size(), maxper, overlap2);
    return 0;
}
