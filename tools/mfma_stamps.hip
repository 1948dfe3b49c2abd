// mfma_stamps.hip -- DIAGNOSTIC build of the matrix-core tiled MAC (k_cmac_mfma) with per-workgroup stamps: start / end on the
// 100 MHz clock, the multiply loop in shader cycles (so: the clock the chip holds under this load, and cycles per MFMA), which CU.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -DIRB_STAMPS -o tools/mfma_stamps tools/mfma_stamps.hip
//   tools/mfma_stamps T splits P      (synthetic operands in the kernel's own layouts; timing only)
#define IRB_STAMPS 1
#include "../irbaboon_amd/csrc/fdl_kernels.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(1024) void k_null(unsigned long long *stamps)
{
    __shared__ float2 l_o[256 * 17];                   // the same LDS and thread footprint as k_cmac_mfma
    if (threadIdx.x == 4000) l_o[0] = make_float2(0.f, 0.f);
    unsigned long long r;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) :: "memory");
    if (threadIdx.x == 0) stamps[blockIdx.x] = r;
}

int main(int argc, char **argv)
{
    const int T = argc > 1 ? std::atoi(argv[1]) : 3750, splits = argc > 2 ? std::atoi(argv[2]) : 1, P = argc > 3 ? std::atoi(argv[3]) : 375;
    constexpr int N = 512, M = N / 2, B = 256;
    const int ntiles = (T + irb::kMfmaTile - 1) / irb::kMfmaTile, grid = ntiles * splits * (M / irb::kMfmaBins);
    const int xt_hist = irb::mfma_hist(P), xt_len = xt_hist + ntiles * irb::kMfmaTile, ht_len = irb::mfma_taps(P);
    float2 *dHt, *dXt, *dpart;
    int2 *dinfo;
    unsigned long long *dst;
    CK(hipMalloc(&dHt, (size_t) M * ht_len * sizeof(float2)));
    CK(hipMalloc(&dXt, (size_t) M * xt_len * sizeof(float2)));
    CK(hipMalloc(&dpart, (size_t) T * splits * M * sizeof(float2)));
    CK(hipMalloc(&dinfo, sizeof(int2)));
    CK(hipMalloc(&dst, (size_t) (grid + 16) * 16 * sizeof(unsigned long long)));
    {
        std::vector<float2> h((size_t) M * std::max(ht_len, xt_len));
        unsigned long long z = 12345;
        for (auto &v : h) { z = z * 6364136223846793005ull + 1442695040888963407ull; v.x = (float) ((long long) (z >> 40) - (1 << 23)) / (float) (1 << 23); v.y = -v.x * 0.5f; }
        CK(hipMemcpy(dHt, h.data(), (size_t) M * ht_len * sizeof(float2), hipMemcpyHostToDevice));
        CK(hipMemcpy(dXt, h.data(), (size_t) M * xt_len * sizeof(float2), hipMemcpyHostToDevice));
        const int2 info = make_int2(0, P);
        CK(hipMemcpy(dinfo, &info, sizeof info, hipMemcpyHostToDevice));
    }
    irb::StepArgs a;
    std::memset(&a, 0, sizeof a);
    a.stamps = dst; a.Xt = dXt; a.Ht = dHt; a.xt_len = xt_len; a.xt_hist = xt_hist; a.ht_len = ht_len; a.stream_info = dinfo; a.part = dpart;
    a.B = B; a.splits = splits; a.T = T; a.nq = splits; a.tile_first = 0; a.tile_count = ntiles; a.t_base = 0; a.part_T = T;
    std::vector<unsigned long long> st((size_t) (grid + 16) * 16);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int it = 0; it < 6; it++) {
        CK(hipMemset(dst, 0, (size_t) (grid + 16) * 16 * sizeof(unsigned long long)));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        irb::k_cmac_mfma<N><<<dim3(grid), dim3(irb::kMfmaThreads)>>>(a);
        CK(hipEventRecord(e1));
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    {   // what a launch of this shape costs with nothing in it: event to event, and first stamp to last
        float nms = 0, best = 1e9f;
        unsigned long long *dn;
        CK(hipMalloc(&dn, (size_t) grid * sizeof(unsigned long long)));
        for (int it = 0; it < 6; it++) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            k_null<<<dim3(grid), dim3(1024)>>>(dn);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&nms, e0, e1));
            best = std::min(best, nms);
        }
        std::vector<unsigned long long> ns(grid);
        CK(hipMemcpy(ns.data(), dn, ns.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::printf("empty kernel of the same grid / block / LDS: event time %.1f us (best of 6; last %.1f), first stamp -> last stamp %.1f us\n", best * 1e3, nms * 1e3,
                    (*std::max_element(ns.begin(), ns.end()) - *std::min_element(ns.begin(), ns.end())) / 100.0);
        CK(hipFree(dn));
    }
    CK(hipMemcpy(st.data(), dst, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int g = 0; g < grid; g++) { t0 = std::min(t0, st[g * 16 + 11]); t1 = std::max(t1, st[g * 16 + 12]); }
    std::vector<double> start, dur, loop, store, clk, ph[5];
    std::map<unsigned, int> per_cu;
    for (int g = 0; g < grid; g++) {
        const unsigned long long *t = &st[(size_t) g * 16];
        start.push_back((t[11] - t0) / 100.0);
        dur.push_back((t[12] - t[11]) / 100.0);
        loop.push_back((double) (t[2] - t[0]));
        ph[0].push_back((double) (t[1] - t[0])); ph[1].push_back((double) (t[3] - t[1])); ph[2].push_back((double) (t[2] - t[3])); ph[3].push_back((double) (t[4] - t[2])); ph[4].push_back((double) (t[7] - t[4]));
        store.push_back((double) (t[7] - t[2]));
        clk.push_back((double) (t[7] - t[0]) / (double) (t[12] - t[11]) * 0.1);
        const unsigned hw = (unsigned) t[14], cu = ((unsigned) t[15] & 15u) << 16 | (hw & 0xFF00u) | ((hw >> 13) & 0x7u) << 4;
        per_cu[cu]++;
    }
    auto pct = [](std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v[(size_t) (p * (v.size() - 1))]; };
    int maxper = 0;
    for (auto &kv : per_cu) maxper = std::max(maxper, kv.second);
    const int J = irb::mfma_chunks(P);
    const double mfma_per_wave = (double) ((J + splits - 1) / splits) * 16;      // chunks of a wave x 16 v_mfma_f32_16x16x4 (32 cycles each; four waves share a SIMD)
    const double flops = 8.0 * P * M * T;
    std::printf("T=%d splits=%d P=%d grid=%d: event time %.1f us = %.1f TFLOP/s useful (%.3f of 157.3); first start -> last end %.1f us (100 MHz clock)\n",
                T, splits, P, grid, ms * 1e3, flops / ms / 1e9, flops / ms / 1e9 / 157.3, (t1 - t0) / 100.0);
    std::printf("  workgroup start (us after the first): p10 %.1f  p50 %.1f  p90 %.1f  max %.1f\n", pct(start, .1), pct(start, .5), pct(start, .9), pct(start, 1));
    std::printf("  workgroup duration (us):               p10 %.1f  p50 %.1f  p90 %.1f  max %.1f\n", pct(dur, .1), pct(dur, .5), pct(dur, .9), pct(dur, 1));
    std::printf("  first loads + chunk loop (wave 0, shader cycles): p10 %.0f  p50 %.0f  p90 %.0f  max %.0f   (<= %.0f MFMAs per wave: %.1f cycles each at p50; 128 = the matrix pipe's own time with four waves per SIMD)\n",
                pct(loop, .1), pct(loop, .5), pct(loop, .9), pct(loop, 1), mfma_per_wave, pct(loop, .5) / mfma_per_wave);
    std::printf("  phases of wave 0 at p50 (shader cycles): arguments + indices %.0f | first operands %.0f | chunk loop %.0f | wait for the other waves %.0f | stores %.0f\n",
                pct(ph[0], .5), pct(ph[1], .5), pct(ph[2], .5), pct(ph[3], .5), pct(ph[4], .5));
    std::printf("  transpose + stores (shader cycles):    p10 %.0f  p50 %.0f  p90 %.0f  max %.0f\n", pct(store, .1), pct(store, .5), pct(store, .9), pct(store, 1));
    std::printf("  shader clock (GHz):                    p10 %.2f  p50 %.2f  p90 %.2f\n", pct(clk, .1), pct(clk, .5), pct(clk, .9));
    std::printf("  distinct CUs used %zu, most workgroups on one CU %d\n", per_cu.size(), maxper);
    return 0;
}
