// step_stamps.hip -- DIAGNOSTIC build of the one-block step kernels with s_memtime stamps between their phases
// (cdna_hip_programming.md section 7, "In-kernel stamps").  Not part of libirbfdl.so: this file compiles
// irbaboon_amd/csrc/fdl_kernels.hpp with -DIRB_STAMPS, launches a kernel directly on synthetic buffers and prints
// where a step spends its cycles.  Read SHARES, not totals: the stamps' fences forbid overlaps the product build has.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -DIRB_STAMPS -o tools/step_stamps tools/step_stamps.hip
//   tools/step_stamps fused|cols  B  P  S  [W PS U]
#define IRB_STAMPS 1
#include "../irbaboon_amd/csrc/fdl_kernels.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static const char *kFused[] = {"start", "older partitions streamed", "small loads in LDS (+barrier)", "forward FFT", "split + ring store (+barrier)",
                               "partition 0 (+barrier)", "stripe reduction", "inverse split", "inverse FFT", "overlap-add issued", "stores landed"};
static const char *kCols[] = {"start", "all loads issued", "small loads in LDS (+barrier)", "forward FFT + split (+barrier)", "partition loads consumed",
                              "stripes reduced, slice stored", "slice drained (+barrier)", "ticket returned (+barrier)", "slices summed (last arriver)",
                              "inverse split + FFT", "overlap-add landed"};

template <int N>
static void launch(const std::string &mode, irb::StepArgs a, unsigned *tickets, int S, int W, int PS, int *grid_out, int U = 8)
{
    if (mode == "fused") {
        *grid_out = S;
        irb::k_step_fused<N, 2, false><<<dim3(S), dim3(irb::kThreads)>>>(a);
    } else if (W == 4) {
        *grid_out = S * irb::ColsGeom<N, 4>::CS * PS;
        irb::k_step_cols<N, 4, 0><<<dim3(*grid_out), dim3(irb::kThreads)>>>(a, tickets);
    } else if (W == 16) {
        *grid_out = S * irb::ColsGeom<N, 16>::CS * PS;
        irb::k_step_cols<N, 16, 0><<<dim3(*grid_out), dim3(irb::kThreads)>>>(a, tickets);
    } else {
        *grid_out = S * irb::ColsGeom<N, 64>::CS * PS;
        irb::k_step_cols<N, 64, 0><<<dim3(*grid_out), dim3(irb::kThreads)>>>(a, tickets);
    }
}

int main(int argc, char **argv)
{
    if (argc < 5) { std::fprintf(stderr, "usage: %s fused|cols B P S [W PS]\n", argv[0]); return 2; }
    const std::string mode = argv[1];
    const int B = std::atoi(argv[2]), P = std::atoi(argv[3]), S = std::atoi(argv[4]);
    const int W = argc > 5 ? std::atoi(argv[5]) : 4, PS = argc > 6 ? std::atoi(argv[6]) : 1, U = argc > 7 ? std::atoi(argv[7]) : 8;
    int N = 32;
    while (N < 2 * B - 1) N *= 2;
    if (N != 256 && N != 512 && N != 1024) { std::fprintf(stderr, "B must give N in {256,512,1024}\n"); return 2; }
    const int M = N / 2, Ppad = (P + 7) / 8 * 8, R = P;
    const size_t SM = (size_t) S * M;

    float2 *dH, *dX, *dtw, *dpart;
    float *dovl, *din, *dout;
    int2 *dinfo;
    unsigned *dtick;
    unsigned long long *dst;
    const int maxgrid = S * (M / 2) * std::max(PS, 1) + 16;
    CK(hipMalloc(&dH, SM * Ppad * sizeof(float2)));
    CK(hipMalloc(&dX, SM * R * sizeof(float2)));
    CK(hipMalloc(&dtw, N * sizeof(float2)));
    CK(hipMalloc(&dpart, SM * std::max(PS, 1) * sizeof(float2)));
    CK(hipMalloc(&dovl, (size_t) S * B * sizeof(float)));
    CK(hipMalloc(&din, (size_t) S * B * sizeof(float)));
    CK(hipMalloc(&dout, (size_t) S * B * sizeof(float)));
    CK(hipMalloc(&dinfo, S * sizeof(int2)));
    CK(hipMalloc(&dtick, S * sizeof(unsigned)));
    CK(hipMalloc(&dst, (size_t) maxgrid * 16 * sizeof(unsigned long long)));
    {
        std::vector<float2> h(SM * Ppad), tw(N);
        unsigned long long z = 12345;
        for (auto &v : h) { z = z * 6364136223846793005ull + 1442695040888963407ull; v.x = (float) ((long long) (z >> 40) - (1 << 23)) / (float) (1 << 23); v.y = -v.x * 0.5f; }
        CK(hipMemcpy(dH, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice));
        h.resize(SM * R);
        CK(hipMemcpy(dX, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice));
        for (int j = 0; j < N; j++) { const double ph = -2.0 * 3.14159265358979323846 * j / N; tw[j] = make_float2((float) std::cos(ph), (float) std::sin(ph)); }
        CK(hipMemcpy(dtw, tw.data(), N * sizeof(float2), hipMemcpyHostToDevice));
        std::vector<float> x((size_t) S * B, 0.25f);
        CK(hipMemcpy(din, x.data(), x.size() * sizeof(float), hipMemcpyHostToDevice));
        std::vector<int2> info(S);
        for (int s = 0; s < S; s++) info[s] = make_int2(s, P);
        CK(hipMemcpy(dinfo, info.data(), S * sizeof(int2), hipMemcpyHostToDevice));
        CK(hipMemset(dovl, 0, (size_t) S * B * sizeof(float)));
        CK(hipMemset(dtick, 0, S * sizeof(unsigned)));
    }
    irb::StepArgs a;
    std::memset(&a, 0, sizeof a);
    a.stamps = dst;
    a.H = dH; a.X = dX; a.ovl = dovl; a.ovl_next = dovl; a.tw = dtw; a.stream_info = dinfo;
    a.in = din; a.out = dout; a.part = dpart; a.in_stride = B; a.out_stride = B;
    a.B = B; a.P = Ppad; a.R = R; a.splits = PS; a.T = 1; a.nq = 1;

    int grid = 0;
    const int reps = 200;
    std::vector<double> sum(16, 0.0), sumlast(16, 0.0);
    std::vector<unsigned long long> st((size_t) maxgrid * 16);
    int nlast = 0, nall = 0;
    std::vector<double> mx(16, 0.0);                    // per phase: the slowest workgroup, averaged over launches
    double span_ticket = 0, span_total = 0, span_first = 0;   // 100 MHz clock: first start -> last ticket / -> end; first start -> last start
    int nspan = 0;
    for (int it = 0; it < reps; it++) {
        a.head = it % R;
        CK(hipMemset(dst, 0, (size_t) maxgrid * 16 * sizeof(unsigned long long)));
        switch (N) {
        case 256: launch<256>(mode, a, dtick, S, W, PS, &grid, U); break;
        case 512: launch<512>(mode, a, dtick, S, W, PS, &grid, U); break;
        default: launch<1024>(mode, a, dtick, S, W, PS, &grid, U); break;
        }
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        if (it < 20) continue;                                   // warm-up
        CK(hipMemcpy(st.data(), dst, (size_t) grid * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        {
            std::vector<double> m(16, 0.0);
            unsigned long long r0min = ~0ull, r0max = 0, r7max = 0, r10max = 0;
            for (int g = 0; g < grid; g++) {
                const unsigned long long *t = &st[(size_t) g * 16];
                for (int i = 1; i <= 7; i++) if (t[i] && t[i - 1]) m[i] = std::max(m[i], (double) (t[i] - t[i - 1]));
                if (t[11]) { r0min = std::min(r0min, t[11]); r0max = std::max(r0max, t[11]); }
                r7max = std::max(r7max, t[12]);
                r10max = std::max(r10max, t[13]);
            }
            for (int i = 1; i <= 7; i++) mx[i] += m[i];
            if (r10max && r0min != ~0ull) { span_ticket += (double) (r7max - r0min); span_total += (double) (r10max - r0min); span_first += (double) (r0max - r0min); nspan++; }
        }
        for (int g = 0; g < grid; g++) {
            const unsigned long long *t = &st[(size_t) g * 16];
            const bool last = t[10] != 0;
            for (int i = 1; i <= 10; i++) {
                if (!t[i] || !t[i - 1]) continue;
                const double d = (double) (t[i] - t[i - 1]);
                if (i <= 7 || mode == "fused") sum[i] += d;
                if (last) sumlast[i] += d;
            }
            nall++;
            if (last) nlast++;
        }
    }
    std::printf("%s B=%d N=%d P=%d S=%d W=%d PS=%d U=%d grid=%d  (cycles of s_memtime = shader clock; ~40 per stamp not subtracted)\n",
                mode.c_str(), B, N, P, S, W, PS, U, grid);
    const char **names = mode == "fused" ? kFused : kCols;
    double tot = 0;
    for (int i = 1; i <= 10; i++) {
        const bool tail = mode != "fused" && i >= 8;
        const double v = tail ? (nlast ? sumlast[i] / nlast : 0.0) : (mode == "fused" ? sum[i] / std::max(nall, 1) : sum[i] / std::max(nall, 1));
        tot += v;
        if (i <= 7 && mode != "fused")
            std::printf("  %2d %-40s %8.0f cyc   slowest workgroup %8.0f\n", i, names[i], v, mx[i] / std::max(reps - 20, 1));
        else
            std::printf("  %2d %-40s %8.0f cyc%s\n", i, names[i], v, tail ? "  (last arriver only)" : "");
    }
    if (nspan)
        std::printf("  100 MHz clock: first workgroup start -> last start %.2f us, -> last ticket %.2f us, -> end of last arriver %.2f us\n",
                    span_first / nspan / 100.0, span_ticket / nspan / 100.0, span_total / nspan / 100.0);
    std::printf("  sum %.0f cycles = %.2f us at 2.4 GHz\n", tot, tot / 2400.0);
    return 0;
}
