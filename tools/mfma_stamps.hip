// mfma_stamps.hip -- DIAGNOSTIC build of the matrix-core tiled MAC (k_cmac_mfma) with per-workgroup stamps: start / end on the
// 100 MHz clock, the multiply loop in shader cycles (so: the clock the chip holds under this load, and cycles per MFMA), which CU.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -DIRB_STAMPS -o tools/mfma_stamps tools/mfma_stamps.hip
//   tools/mfma_stamps T splits P
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
    const int T = argc > 1 ? std::atoi(argv[1]) : 3750, splits = argc > 2 ? std::atoi(argv[2]) : 2, P = argc > 3 ? std::atoi(argv[3]) : 375;
    constexpr int N = 512, M = N / 2, B = 256;
    const int Ppad = (P + 23) / 24 * 24, R = Ppad + T + 8, ntiles = (T + irb::kMfmaTile - 1) / irb::kMfmaTile, grid = ntiles * splits * (M / irb::kMfmaBins);
    float2 *dH, *dX, *dpart;
    int2 *dinfo;
    unsigned long long *dst;
    CK(hipMalloc(&dH, (size_t) M * (Ppad + 8) * sizeof(float2)));
    CK(hipMalloc(&dX, (size_t) M * R * sizeof(float2)));
    CK(hipMalloc(&dpart, (size_t) T * splits * M * sizeof(float2)));
    CK(hipMalloc(&dinfo, sizeof(int2)));
    CK(hipMalloc(&dst, (size_t) (grid + 16) * 16 * sizeof(unsigned long long)));
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
    std::vector<unsigned long long> st((size_t) (grid + 16) * 16);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&irb::k_cmac_mfma<N>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) irb::kMfmaLds));
    for (int it = 0; it < 6; it++) {
        CK(hipMemset(dst, 0, (size_t) (grid + 16) * 16 * sizeof(unsigned long long)));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        irb::k_cmac_mfma<N><<<dim3(grid), dim3(irb::kMfmaThreads), irb::kMfmaLds>>>(a);
        CK(hipEventRecord(e1));
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    CK(hipMemcpy(st.data(), dst, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int g = 0; g < grid; g++) { t0 = std::min(t0, st[g * 16 + 11]); t1 = std::max(t1, st[g * 16 + 12]); }
    std::vector<double> start, dur, stage, loop, store, clk, tst, tmu;
    std::map<unsigned, std::vector<std::pair<double, double>>> cu_iv;
    for (int g = 0; g < grid; g++) {
        const unsigned long long *t = &st[(size_t) g * 16];
        start.push_back((t[11] - t0) / 100.0);
        dur.push_back((t[12] - t[11]) / 100.0);
        stage.push_back((double) (t[1] - t[0]));
        loop.push_back((double) (t[2] - t[1]));
        store.push_back((double) (t[7] - t[2]));
        tst.push_back((double) t[3]);
        tmu.push_back((double) t[4]);
        clk.push_back((double) (t[7] - t[0]) / (double) (t[12] - t[11]) * 0.1);
        const unsigned hw = (unsigned) t[14], cu = ((unsigned) t[15] & 15u) << 16 | (hw & 0xFF00u) | ((hw >> 13) & 0x7u) << 4;
        cu_iv[cu].push_back({(t[11] - t0) / 100.0, (t[12] - t0) / 100.0});
    }
    auto pct = [](std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v[(size_t) (p * (v.size() - 1))]; };
    int maxper = 0;
    for (auto &kv : cu_iv) maxper = std::max(maxper, (int) kv.second.size());
    const int J = irb::mfma_chunks(P);
    const double mfma_per_wave = (double) ((J + splits - 1) / splits) * 16;      // chunks x bins of a wave x 16 v_mfma_f32_16x16x4 (32 cycles each)
    std::printf("T=%d splits=%d P=%d grid=%d: event time %.1f us; first start -> last end %.1f us (100 MHz clock)\n", T, splits, P, grid, ms * 1e3, (t1 - t0) / 100.0);
    std::printf("  workgroup start (us after the first): p10 %.1f  p50 %.1f  p90 %.1f  max %.1f\n", pct(start, .1), pct(start, .5), pct(start, .9), pct(start, 1));
    std::printf("  workgroup duration (us):               p10 %.1f  p50 %.1f  p90 %.1f  max %.1f\n", pct(dur, .1), pct(dur, .5), pct(dur, .9), pct(dur, 1));
    std::printf("  first stage (shader cycles):           p10 %.0f  p50 %.0f  p90 %.0f  max %.0f\n", pct(stage, .1), pct(stage, .5), pct(stage, .9), pct(stage, 1));
    std::printf("  multiply loop (shader cycles):         p10 %.0f  p50 %.0f  p90 %.0f  max %.0f   (<= %.0f MFMAs per wave: %.1f cycles each at p50)\n",
                pct(loop, .1), pct(loop, .5), pct(loop, .9), pct(loop, 1), mfma_per_wave, pct(loop, .5) / mfma_per_wave);
    std::printf("    of it inside the stages' MFMA code:  p10 %.0f  p50 %.0f  p90 %.0f  max %.0f   (wave 0; the rest: barriers)\n", pct(tmu, .1), pct(tmu, .5), pct(tmu, .9), pct(tmu, 1));
    std::printf("  stores (shader cycles):                p10 %.0f  p50 %.0f  p90 %.0f  max %.0f\n", pct(store, .1), pct(store, .5), pct(store, .9), pct(store, 1));
    std::printf("  shader clock (GHz):                    p10 %.2f  p50 %.2f  p90 %.2f\n", pct(clk, .1), pct(clk, .5), pct(clk, .9));
    std::printf("  distinct CUs used %zu, most workgroups on one CU %d\n", cu_iv.size(), maxper);
    if (argc > 4) {                                   // one workgroup's waves, stage by stage: cycles after that stage's barrier at which a wave starts / ends its MFMA code
        const unsigned long long *w = &st[(size_t) grid * 16];
        for (int stg = 0; stg < 4; stg++) {
            unsigned long long b = ~0ull;
            for (int wv = 0; wv < 16; wv++) if (w[wv * 16 + stg * 4]) b = std::min(b, w[wv * 16 + stg * 4]);
            if (b == ~0ull) continue;
            std::printf("  stage %d of workgroup %d (wave: SIMD, MFMA code from .. to, cycles after the barrier):", stg, grid / 2);
            for (int wv = 0; wv < 16; wv++) {
                const unsigned long long *e = w + wv * 16 + stg * 4;
                if (wv % 4 == 0) std::printf("\n   ");
                std::printf(" w%-2d s%llu %5llu..%-6llu", wv, (e[3] >> 4) & 3, e[1] - b, e[2] - b);
            }
            std::printf("\n");
        }
    }
    return 0;
}
