// clock_under_load.hip -- the shader clock the chip holds under the two HBM-bound kernels of the bench, and their streaming rate,
// on the same buffers: k_step_fused<512,2,true> (one block per stream-step) and k_tile8_fused<512,true> (eight blocks per call), 2048
// streams x 375 partitions.  Thread 0 of every workgroup stamps s_memtime (shader clock) and s_memrealtime (100 MHz) at its start
// and end: GHz = d(memtime) / d(memrealtime) / 10.  Diagnostic build (-DIRB_STAMPS -DIRB_STAMPS_ENDS_ONLY), not part of libirbfdl.so.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -DIRB_STAMPS -DIRB_STAMPS_ENDS_ONLY -o tools/clock_under_load tools/clock_under_load.hip
#define IRB_STAMPS 1
#define IRB_STAMPS_ENDS_ONLY 1
#include "../irbaboon_amd/csrc/fdl_kernels.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int S = argc > 1 ? std::atoi(argv[1]) : 2048;
    constexpr int N = 512, M = N / 2, B = 256, P = 375, Ppad = 384, T = 8, R = Ppad + T + 8;
    const size_t SM = (size_t) S * M;
    float2 *dH, *dX, *dtw;
    float *dovl, *din, *dout;
    int2 *dinfo;
    unsigned long long *dst;
    CK(hipMalloc(&dH, (SM * Ppad + 8 * M) * sizeof(float2)));
    CK(hipMalloc(&dX, SM * R * sizeof(float2)));
    CK(hipMalloc(&dtw, N * sizeof(float2)));
    CK(hipMalloc(&dovl, (size_t) 2 * S * B * sizeof(float)));
    CK(hipMalloc(&din, (size_t) S * T * B * sizeof(float)));
    CK(hipMalloc(&dout, (size_t) S * T * B * sizeof(float)));
    CK(hipMalloc(&dinfo, S * sizeof(int2)));
    CK(hipMalloc(&dst, (size_t) S * 16 * sizeof(unsigned long long)));
    {
        std::vector<float2> h((size_t) M * R * 64), tw(N);                // 64 streams' worth of values, repeated
        unsigned long long z = 12345;
        for (auto &v : h) { z = z * 6364136223846793005ull + 1442695040888963407ull; v.x = (float) ((long long) (z >> 40) - (1 << 23)) / (float) (1 << 23); v.y = -v.x * 0.5f; }
        for (size_t off = 0; off < SM * Ppad; off += h.size()) CK(hipMemcpy(dH + off, h.data(), std::min(h.size(), SM * Ppad - off) * sizeof(float2), hipMemcpyHostToDevice));
        for (size_t off = 0; off < SM * R; off += h.size()) CK(hipMemcpy(dX + off, h.data(), std::min(h.size(), SM * R - off) * sizeof(float2), hipMemcpyHostToDevice));
        for (int j = 0; j < N; j++) { const double ph = -2.0 * 3.14159265358979323846 * j / N; tw[j] = make_float2((float) std::cos(ph), (float) std::sin(ph)); }
        CK(hipMemcpy(dtw, tw.data(), N * sizeof(float2), hipMemcpyHostToDevice));
        std::vector<float> x((size_t) S * T * B, 0.25f);
        CK(hipMemcpy(din, x.data(), x.size() * sizeof(float), hipMemcpyHostToDevice));
        std::vector<int2> info(S);
        for (int s = 0; s < S; s++) info[s] = make_int2(s, P);
        CK(hipMemcpy(dinfo, info.data(), S * sizeof(int2), hipMemcpyHostToDevice));
        CK(hipMemset(dovl, 0, (size_t) 2 * S * B * sizeof(float)));
    }
    irb::StepArgs a;
    std::memset(&a, 0, sizeof a);
    a.stamps = dst;
    a.H = dH; a.X = dX; a.ovl = dovl; a.ovl_next = dovl + (size_t) S * B; a.tw = dtw; a.stream_info = dinfo;
    a.in = din; a.out = dout; a.in_stride = (long) T * B; a.out_stride = (long) T * B;
    a.B = B; a.P = Ppad; a.R = R; a.splits = 1; a.nq = 1; a.in_len = a.out_len = (long) 1 << 62;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<unsigned long long> st((size_t) S * 16);
    auto pct = [](std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v[(size_t) (p * (v.size() - 1))]; };
    for (int which = 0; which < 2; which++) {
        std::vector<double> ms_all, clk_all;
        for (int it = 0; it < 12; it++) {
            a.head = (8 * it) % R; a.T = which ? T : 1;
            CK(hipMemset(dst, 0, (size_t) S * 16 * sizeof(unsigned long long)));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            if (which) irb::k_tile8_fused<N, true><<<dim3(S), dim3(irb::kThreads)>>>(a);
            else       irb::k_step_fused<N, 2, true><<<dim3(S), dim3(irb::kThreads)>>>(a);
            CK(hipEventRecord(e1));
            CK(hipGetLastError());
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (it < 3) continue;
            ms_all.push_back(ms);
            CK(hipMemcpy(st.data(), dst, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            for (int g = 0; g < S; g++) {
                const unsigned long long *t = &st[(size_t) g * 16];
                if (t[13] > t[11]) clk_all.push_back((double) (t[10] - t[0]) / (double) (t[13] - t[11]) * 0.1);
            }
        }
        const double rows = which ? (double) S * (2.0 * P + T) : (double) S * 2.0 * P;      // H + FDL window, 2 KiB each
        const double ms = pct(ms_all, 0.5);
        std::printf("%-28s %4d streams: %.3f ms per launch, %.2f TB/s of H + FDL rows; shader clock over a workgroup's lifetime p10 %.2f  p50 %.2f  p90 %.2f GHz\n",
                    which ? "k_tile8_fused<512,true>" : "k_step_fused<512,2,true>", S, ms, rows * 2048.0 / ms / 1e9, pct(clk_all, .1), pct(clk_all, .5), pct(clk_all, .9));
    }
    return 0;
}
