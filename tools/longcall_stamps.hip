// longcall_stamps.hip -- DIAGNOSTIC build of the one-grid long call (k_long_call) with per-workgroup stamps on the 100 MHz clock:
// when the workgroups of each region start, when their wait for the earlier region ends, when they end.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -DIRB_STAMPS -o tools/longcall_stamps tools/longcall_stamps.hip
//   tools/longcall_stamps T splits P [three]     (one stream, N = 512, synthetic operands; timing only.  three = the three kernels apart)
#define IRB_STAMPS 1
#include "../irbaboon_amd/csrc/fdl_kernels.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int T = argc > 1 ? std::atoi(argv[1]) : 3750, splits = argc > 2 ? std::atoi(argv[2]) : 1, P = argc > 3 ? std::atoi(argv[3]) : 375;
    const bool three = argc > 4 && !std::strcmp(argv[4], "three");
    constexpr int N = 512, M = N / 2, B = 256;
    const int ntiles = (T + irb::kMfmaTile - 1) / irb::kMfmaTile, nB = ntiles * splits * (M / irb::kMfmaBins);
    const int groups = (T + 15) / 16, nA = groups, nC = (T + irb::kLcOlaBlocks - 1) / irb::kLcOlaBlocks, nC3 = (T + 3) / 4;
    const int grid = nA + nB + std::max(nC, nC3);
    const int xt_hist = irb::mfma_hist(P), xt_len = xt_hist + ntiles * irb::kMfmaTile, ht_len = irb::mfma_taps(P), R = P + T + 32;
    float2 *dHt, *dXt, *dpart, *dX, *dtw;
    float *din, *dout, *dovl;
    int2 *dinfo;
    unsigned *dflags, *dstatus;
    unsigned long long *dst;
    CK(hipMalloc(&dHt, (size_t) M * ht_len * sizeof(float2)));
    CK(hipMalloc(&dXt, (size_t) M * xt_len * sizeof(float2)));
    CK(hipMalloc(&dX, (size_t) M * R * sizeof(float2)));
    CK(hipMalloc(&dtw, (size_t) N * sizeof(float2)));
    CK(hipMalloc(&dpart, (size_t) T * splits * M * sizeof(float2)));
    CK(hipMalloc(&din, (size_t) T * B * sizeof(float)));
    CK(hipMalloc(&dout, (size_t) T * B * sizeof(float)));
    CK(hipMalloc(&dovl, (size_t) 2 * B * sizeof(float)));
    CK(hipMalloc(&dinfo, sizeof(int2)));
    CK(hipMalloc(&dflags, (size_t) (nA + nB) * sizeof(unsigned)));
    CK(hipMalloc(&dstatus, sizeof(unsigned)));
    CK(hipMalloc(&dst, (size_t) (grid + 16) * 16 * sizeof(unsigned long long)));
    CK(hipMemset(dflags, 0, (size_t) (nA + nB) * sizeof(unsigned)));
    CK(hipMemset(dstatus, 0, sizeof(unsigned)));
    CK(hipMemset(dovl, 0, (size_t) 2 * B * sizeof(float)));
    {
        std::vector<float2> h((size_t) M * std::max(std::max(ht_len, xt_len), R));
        unsigned long long z = 12345;
        for (auto &v : h) { z = z * 6364136223846793005ull + 1442695040888963407ull; v.x = (float) ((long long) (z >> 40) - (1 << 23)) / (float) (1 << 23); v.y = -v.x * 0.5f; }
        CK(hipMemcpy(dHt, h.data(), (size_t) M * ht_len * sizeof(float2), hipMemcpyHostToDevice));
        CK(hipMemcpy(dXt, h.data(), (size_t) M * xt_len * sizeof(float2), hipMemcpyHostToDevice));
        CK(hipMemcpy(dX, h.data(), (size_t) M * R * sizeof(float2), hipMemcpyHostToDevice));
        CK(hipMemcpy(dtw, h.data(), (size_t) N * sizeof(float2), hipMemcpyHostToDevice));
        CK(hipMemcpy(din, h.data(), (size_t) T * B * sizeof(float), hipMemcpyHostToDevice));
        const int2 info = make_int2(0, P);
        CK(hipMemcpy(dinfo, &info, sizeof info, hipMemcpyHostToDevice));
    }
    irb::StepArgs a;
    std::memset(&a, 0, sizeof a);
    a.stamps = dst; a.X = dX; a.ovl = dovl; a.ovl_next = dovl + B; a.tw = dtw; a.stream_info = dinfo; a.in = din; a.out = dout; a.part = dpart;
    a.status = dstatus; a.in_stride = a.out_stride = (long) T * B; a.B = B; a.P = P; a.R = R; a.head = 0;
    a.Xt = dXt; a.Ht = dHt; a.xt_len = xt_len; a.xt_hist = xt_hist; a.ht_len = ht_len; a.hist_first = -1;
    a.splits = splits; a.T = T; a.nq = splits; a.tile_first = 0; a.tile_count = ntiles; a.t_base = 0; a.part_T = T;
    a.t_first = 0; a.t_count = T; a.reg_t0[0] = 0; a.reg_t0[1] = T; a.reg_t0[2] = T; a.reg_nq[0] = splits; a.reg_nq[1] = 1;
    a.in_len = a.out_len = (long) 1 << 62; a.pk_blocks = 16; a.ck_blocks = 1;
    a.lc_flags = dflags; a.lc_nA = nA; a.lc_nB = nB;
    std::vector<unsigned long long> st((size_t) (grid + 16) * 16);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0, best = 1e9f;
    for (int it = 0; it < 8; it++) {
        CK(hipMemset(dst, 0, (size_t) (grid + 16) * 16 * sizeof(unsigned long long)));
        CK(hipDeviceSynchronize());
        a.lc_seq = it + 1;
        CK(hipEventRecord(e0));
        if (three) {
            irb::k_push_fft_batch_t<N><<<dim3(nA), dim3(1024)>>>(a, 1);
            a.stamps = dst + (size_t) nA * 16;
            irb::k_cmac_mfma<N><<<dim3(nB), dim3(irb::kMfmaThreads)>>>(a);
            a.stamps = dst + (size_t) (nA + nB) * 16;
            irb::k_ifft_ola_run<N, 4><<<dim3(nC3), dim3(320)>>>(a, 1);
            a.stamps = dst;
        } else {
            irb::k_long_call<N><<<dim3(nA + nB + nC), dim3(irb::kMfmaThreads)>>>(a, 1);
        }
        CK(hipEventRecord(e1));
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    unsigned status = 0;
    CK(hipMemcpy(&status, dstatus, sizeof status, hipMemcpyDeviceToHost));
    CK(hipMemcpy(st.data(), dst, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::printf("T=%d splits=%d P=%d %s: regions %d + %d + %d workgroups; event time %.1f us (best of 8: %.1f); status word %u\n", T, splits, P,
                three ? "THREE LAUNCHES" : "one grid", nA, nB, three ? nC3 : nC, ms * 1e3, best * 1e3, status);
    auto pct = [](std::vector<double> v, double p) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[(size_t) (p * (v.size() - 1))]; };
    unsigned long long t0 = ~0ull;
    for (int g = 0; g < nA; g++) if (st[(size_t) g * 16 + 11]) t0 = std::min(t0, st[(size_t) g * 16 + 11]);
    const char *names[3] = {"forward FFTs", "multiply    ", "inverse FFTs"};
    const int lo[4] = {0, nA, nA + nB, nA + nB + (three ? nC3 : nC)};
    for (int r = 0; r < 3; r++) {
        std::vector<double> start, ready, end;
        for (int g = lo[r]; g < lo[r + 1]; g++) {
            const unsigned long long *t = &st[(size_t) g * 16];
            if (!t[11]) continue;
            start.push_back((t[11] - t0) / 100.0);
            if (t[8]) ready.push_back((t[8] - t0) / 100.0);
            if (t[12]) end.push_back((t[12] - t0) / 100.0);
        }
        std::printf("  %s  start min %.1f p50 %.1f max %.1f | %s min %.1f p50 %.1f max %.1f | end min %.1f p50 %.1f max %.1f   (us after the grid's first workgroup)\n",
                    names[r], pct(start, 0), pct(start, .5), pct(start, 1), r == 0 ? "Xt written " : "wait over  ", pct(ready, 0), pct(ready, .5), pct(ready, 1),
                    pct(end, 0), pct(end, .5), pct(end, 1));
    }
    return 0;
}
