// fdl_longcall.hpp -- a long call (>= 256 blocks per stream, multiplied on the matrix cores) as ONE grid.  Included by fdl_kernels.hpp.
//
// irb_fdl_process with many blocks per call ran three kernels back to back: k_push_fft_batch_t (forward FFTs into the ring and into
// Xt), k_cmac_mfma (the Toeplitz-block multiply), k_ifft_ola_run (sum of the partial spectra, inverse FFT, overlap-add;
// fp/convolution.cpp:128-145, 171-202, 206-213 for all blocks of the call).  At configs[1] (one stream, 3750 blocks) the two
// kernel boundaries and the three kernels' own fill and drain were ~17 of the call's 46 us (profiles/r04_mfma_kernel_trace.txt).
// Here the three are REGIONS of one grid, in dispatch order:
//     workgroups [0, lc_nA)                  forward FFTs of 16 blocks each (+ the workgroups that rebuild Xt's history columns)
//                [lc_nA, lc_nA + lc_nB)      one (tile of 256 blocks) x (16 bins) x (range of Toeplitz chunks) each, the call's plan
//                [lc_nA + lc_nB, grid)       inverse FFT + overlap-add of 15 blocks each (16 waves: the run's predecessor + 15)
// every workgroup the code of its stand-alone kernel (the same device functions, FUSED = true), 1024 threads, one per CU.  A
// workgroup of a later region waits for exactly the workgroups of the earlier region whose bytes it addresses -- the hand-off is
// described at lc_publish / lc_wait in fdl_mfma.hpp; its index arithmetic is restated in tests/kernel_model.py
// (long_call_handoffs) and checked there against what each workgroup reads.
// N <= 512 (a forward-FFT workgroup = one whole segment of 16 blocks = 16 waves); longer transforms keep the three launches.
#pragma once

namespace irb {

constexpr int kLcOlaBlocks = 15;                       // output blocks per workgroup of the third region

template <int N> struct LongCallLds {
    static constexpr size_t kPush = sizeof(float2) * (TwLds<N>::SIZE + 16 * (N / 2 + 1));
    static constexpr size_t kMac = sizeof(float2) * kMfmaTile * kMfmaOS;
    static constexpr size_t kOla = sizeof(float2) * (TwLds<N>::SIZE + (kLcOlaBlocks + 1) * (N / 2));
    static constexpr size_t kMax = kPush > kMac ? (kPush > kOla ? kPush : kOla) : (kMac > kOla ? kMac : kOla);
    static constexpr size_t kBytes = (kMax + 15) / 16 * 16;
};

template <int N>
__global__ __launch_bounds__(kMfmaThreads) void k_long_call(StepArgs a, int S)
{
    static_assert(PushTGeom<N>::NBW * kWave == kMfmaThreads && (kLcOlaBlocks + 1) * kWave == kMfmaThreads, "one block size for the three regions");
    __shared__ __attribute__((aligned(16))) char l_raw[LongCallLds<N>::kBytes];
    float2 *const l_f2 = reinterpret_cast<float2 *>(l_raw);
    const int bid = blockIdx.x;
    if (bid < a.lc_nA) {
        push_fft_batch_t_body<N, true>(a, S, bid, l_f2, reinterpret_cast<float2 (*)[N / 2 + 1]>(l_f2 + TwLds<N>::SIZE));
    } else if (bid < a.lc_nA + a.lc_nB) {
        cmac_mfma_body<N, true>(a, S, bid - a.lc_nA, a.lc_nB, l_f2);
    } else {
        ifft_ola_run_body<N, kLcOlaBlocks, true>(a, S, bid - a.lc_nA - a.lc_nB, l_f2, reinterpret_cast<float2 (*)[N / 2]>(l_f2 + TwLds<N>::SIZE));
    }
}

}  // namespace irb
