// fdl_tiled_pk.hpp -- the time-tiled multiply-accumulate of calls that carry T > 1 blocks per stream, second
// generation (round 3), and the inverse-FFT / overlap-add kernel that follows it.  Included by fdl_kernels.hpp.
//
// What round 2's tile (k_cmac_tiled<N,8>, removed in round 5) was bound by (tools/valu_peak.hip + its own byte count): a lane multiplies
// 8 partitions x 8 outputs x 2 bins per step = 512 v_fma_f32 from 16 row loads of 16 B, i.e. 4 flop per byte
// through the CU's vector L1 (64 B/clk): at the fp32 vector peak the L1 would have to deliver exactly its own
// peak, so the kernel sat where two equally loaded pipes without perfect overlap sit (0.36 of 157 TFLOP/s).
// This kernel
//   * multiplies with v_pk_fma_f32 (two FMAs per lane per instruction, operand halves picked by op_sel /
//     neg_lo): one complex multiply-accumulate = 2 instructions instead of 4, and the measured packed rate
//     needs half the resident waves (130 TFLOP/s at 2 waves per SIMD, 145 at 4; v_fma_f32: 115 / 125);
//   * keeps a 16-output tile: per U partitions 2U row loads feed 16 U complex products per bin (8 flop per
//     L1 byte), the X window lives in a register ring whose indices are compile-time constants (the loop is
//     unrolled over the ring period), so the window never moves between registers.
// The arithmetic per accumulator is the same FMA sequence as that kernel's: results were bit-identical to it (tested until its removal).
#pragma once

namespace irb {

// H rows per IR set are a multiple of this (zero rows past the IR's partitions): one trip around the register ring of the
// tile below (3 steps of 8 partitions), so a worker never needs a remainder loop
constexpr int kTilePeriod = 24;

// acc += x * h for one complex bin, as two packed FMAs:
//   acc.xy += ( x.x, x.x) * (f.x, f.y)
//   acc.xy += (-x.y, x.y) * (g.y, g.x)
// f == g == h for an ordinary bin; for packed bin 0 (DC and Nyquist: two real products) f = (h.x, 0), g = (h.y, 0).
__device__ __forceinline__ void cmac_pk(v2f_t &acc, const v2f_t x, const v2f_t f, const v2f_t g)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(x), "v"(f));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(acc) : "v"(x), "v"(g));
}

template <int TT, int U>
struct TilePk {
    static constexpr int W = (TT + U - 1 + U - 1) / U * U;      // register ring: >= TT+U-1 rows, a multiple of U
    static constexpr int PERIOD = W / U;                         // steps after which the ring indices repeat
    // logical window row i (time j0 - k0 - (U-1) + i) of step S lives in ring register (i - S*U) mod W
    static constexpr int phys(int i, int S) { return ((i - S * U) % W + W) % W; }
};

// one step of the tile: U partitions [k0, k0+U) against the TT outputs of the tile
template <int N, int TT, int U, int S, bool NT, bool PACKED>
__device__ __forceinline__ void tile_step(const float4 *__restrict__ H4, const float4 *__restrict__ X4, int col, int k0, int j0,
                                          int head, int R, float4 (&xr)[TilePk<TT, U>::W], v2f_t (&acc)[TT][2], bool packed)
{
    using G = MacGeom<N>;
    using TP = TilePk<TT, U>;
    float4 h[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        h[u] = ld16<NT>(H4 + (size_t) (k0 + u) * G::L + col);
        xr[TP::phys(U - 1 - u, S)] = ld16<NT>(X4 + (size_t) ring_slot(head, j0 - k0 - u, R) * G::L + col);
    }
    __builtin_amdgcn_sched_barrier(0);         // loads first
#pragma unroll
    for (int u = 0; u < U; u++) {
        const v2f_t hz = {h[u].z, h[u].w};
        v2f_t f = {h[u].x, h[u].y}, g = f;
        if constexpr (PACKED) {                // only the wave that holds column 0 runs this variant
            f = v2f_t{h[u].x, packed ? 0.0f : h[u].y};
            g = v2f_t{packed ? h[u].y : h[u].x, packed ? 0.0f : h[u].y};
        }
#pragma unroll
        for (int j = 0; j < TT; j++) {
            const float4 x = xr[TP::phys(j - u + U - 1, S)];
            cmac_pk(acc[j][0], v2f_t{x.x, x.y}, f, g);
            cmac_pk(acc[j][1], v2f_t{x.z, x.w}, hz, hz);
        }
    }
}

// the PERIOD steps of one trip around the register ring, S = 0 .. PERIOD-1, in one basic block
template <int N, int TT, int U, int S, bool NT, bool PACKED>
__device__ __forceinline__ void tile_period(const float4 *__restrict__ H4, const float4 *__restrict__ X4, int col, int k0, int j0,
                                            int head, int R, float4 (&xr)[TilePk<TT, U>::W], v2f_t (&acc)[TT][2], bool packed)
{
    if constexpr (S < TilePk<TT, U>::PERIOD) {
        tile_step<N, TT, U, S, NT, PACKED>(H4, X4, col, k0 + S * U, j0, head, R, xr, acc, packed);
        tile_period<N, TT, U, S + 1, NT, PACKED>(H4, X4, col, k0, j0, head, R, xr, acc, packed);
    }
}

// partitions [sb, se) of one worker; se - sb is a multiple of kTilePeriod (whole trips around the ring: every register
// index is a constant and nothing ever moves between registers)
template <int N, int TT, int U, bool NT, bool PACKED>
__device__ __forceinline__ void tile_run(const float4 *__restrict__ H4, const float4 *__restrict__ X4, int col, int sb, int se, int j0,
                                         int head, int R, v2f_t (&acc)[TT][2], bool packed)
{
    using G = MacGeom<N>;
    using TP = TilePk<TT, U>;
    static_assert(kTilePeriod % (U * TP::PERIOD) == 0, "worker ranges are multiples of kTilePeriod partitions");
    float4 xr[TP::W];
#pragma unroll
    for (int i = U; i < TT + U - 1; i++)
        xr[TP::phys(i, 0)] = ld16<NT>(X4 + (size_t) ring_slot(head, j0 - sb - (U - 1) + i, R) * G::L + col);
#pragma unroll 1
    for (int k0 = sb; k0 < se; k0 += U * TP::PERIOD)
        tile_period<N, TT, U, 0, NT, PACKED>(H4, X4, col, k0, j0, head, R, xr, acc, packed);
}

// grid = S * ceil(T/TT) * splits.  The stripes of a workgroup (partition sub-ranges of the same outputs)
// are added through LDS before anything leaves the CU, so a workgroup writes ONE partial spectrum per output:
// part[s][j][split][M] (round 2 wrote one per stripe: at one stream x 3750 blocks that was 61 MB of partials written,
// flushed at the kernel boundary and read back by the inverse-FFT kernel -- more bytes than H and the FDL together).
// H rows are zero padded to multiples of kTilePeriod by the host (and the allocation ends with spare rows for the last
// read-ahead of a diagnostic build); a worker's partition range is a whole number of periods.
// Tried and dropped (profiles/r03_tiled_stamps.txt): a 16 x 4 tile with the NEXT step's rows already in flight while the
// current step is multiplied (window ring of TT + 2U - 1 rows, H double buffered, 255 VGPRs) -- same time within 1 %: the
// loop is bound by how fast one or two waves per SIMD issue v_pk_fma_f32 (5.4-6.5 cycles each at the ~1.7 GHz the chip
// holds under this load), not by the latency of its loads.
template <int N, int TT, int U, int MINW, bool NT = false>
__global__ __launch_bounds__(kThreads, MINW) void k_cmac_tiled_pk(StepArgs a)
{
    using G = MacGeom<N>;
    constexpr int M = N / 2;
    __shared__ float4 l_red[G::STRIPES > 1 ? G::STRIPES - 1 : 1][TT][G::LW];
    const int ntiles = (a.T + TT - 1) / TT;
    const int wg = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int split = wg % a.splits, j0 = ((wg / a.splits) % ntiles) * TT, s = wg / (a.splits * ntiles);
    const int2 si = a.stream_info[s];
    const int set = si.x, np = si.y;
    const int col0 = threadIdx.x % G::LW;
    const int stripe = G::LW >= kWave ? __builtin_amdgcn_readfirstlane(threadIdx.x / G::LW) : (int) (threadIdx.x / G::LW);
    const int q = split * G::STRIPES + stripe;
    const int nG = (np + kTilePeriod - 1) / kTilePeriod, nW = a.splits * G::STRIPES;
    const int sb = (int) ((long) nG * q / nW) * kTilePeriod, se = (int) ((long) nG * (q + 1) / nW) * kTilePeriod;
    const float4 *H4 = reinterpret_cast<const float4 *>(a.H + (size_t) set * a.P * M);
    const float4 *X4 = reinterpret_cast<const float4 *>(a.X + (size_t) s * a.R * M);
    IRB_STAMP(a, 0);
#ifdef IRB_STAMPS
    if (threadIdx.x == 0) {                            // where this workgroup runs: HW_ID (wave / SIMD / CU / SE) and the XCD
        a.stamps[(size_t) blockIdx.x * 16 + 14] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID, 32 bits
        a.stamps[(size_t) blockIdx.x * 16 + 15] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));      // HW_REG_XCC_ID[3:0]
    }
#endif

#pragma unroll 1
    for (int c = 0; c < G::CPT; c++) {
        const int col = col0 + c * G::LW;
        v2f_t acc[TT][2];
#pragma unroll
        for (int j = 0; j < TT; j++) { acc[j][0] = v2f_t{0.f, 0.f}; acc[j][1] = v2f_t{0.f, 0.f}; }
        if (sb < se) {
            // column 0 (packed bin 0) sits in lane 0 of ONE wave per stripe: only that wave pays for the operand selects
            const bool wave_has_col0 = G::LW >= kWave ? __builtin_amdgcn_readfirstlane(col) == 0 : true;
            if (wave_has_col0) tile_run<N, TT, U, NT, true>(H4, X4, col, sb, se, j0, a.head, a.R, acc, col == 0);
            else               tile_run<N, TT, U, NT, false>(H4, X4, col, sb, se, j0, a.head, a.R, acc, false);
        }
        IRB_STAMP(a, 1);
        if constexpr (G::STRIPES > 1) {
            if (c > 0) __syncthreads();                // the previous column block's sums have been read
            if (stripe > 0) {
#pragma unroll
                for (int j = 0; j < TT; j++) l_red[stripe - 1][j][col0] = make_float4(acc[j][0].x, acc[j][0].y, acc[j][1].x, acc[j][1].y);
            }
            __syncthreads();
            if (stripe == 0) {
#pragma unroll
                for (int j = 0; j < TT; j++) {
#pragma unroll
                    for (int t = 0; t < G::STRIPES - 1; t++) {
                        const float4 v = l_red[t][j][col0];
                        acc[j][0].x += v.x; acc[j][0].y += v.y; acc[j][1].x += v.z; acc[j][1].y += v.w;
                    }
                }
            }
        }
        if (stripe == 0) {
#pragma unroll
            for (int j = 0; j < TT; j++) {
                if (j0 + j < a.T) {
                    float4 *dst = reinterpret_cast<float4 *>(a.part + (((size_t) s * a.T + j0 + j) * a.nq + split) * M);
                    dst[col] = make_float4(acc[j][0].x, acc[j][0].y, acc[j][1].x, acc[j][1].y);
                }
            }
        }
    }
    IRB_STAMP(a, 7);
}

// ---------------------------------------------------------------------------------------------
// k_tile8_fused: a call of T <= 8 blocks per stream for batches that give every stream ONE workgroup (HBM-sized batches of
// short calls: a DAW host block of 2048 at B = 256) as a single launch -- k_push_fft_batch, k_cmac_tiled_pk<N, 8, 4> and
// k_ifft_ola_run in one kernel, like k_step_fused for one block:
//   A  the four waves transform the call's blocks (two each) into the FDL ring (fp/convolution.cpp:128-145); barrier
//   B  the 8-output x 4-partition register tile over ALL partitions of the stream (tile_run, fp/convolution.cpp:171-202): every
//      H row and every FDL row is read once for the eight outputs
//   C  the stripes' sums meet in LDS: eight spectra
//   D  the four waves invert them (two each, :206) and add the overlaps (:209-213): out[j] = y_j[0:B] + y_(j-1)[B:2B], the first from
//      the overlap the previous call saved, the last one's second half saved for the next
// Nothing but the FDL rows themselves goes through HBM twice: the three-launch form wrote the eight new spectra and eight partial
// spectra per stream, flushed them at two kernel boundaries and read them back (profiles/r04_tiled8_pmc.txt: 18 + 24 us of
// 572 and 67 MB of 3.3 GB at 2048 streams).  grid = S, block = 256; 256 <= N <= 1024.
// ---------------------------------------------------------------------------------------------
// phase A of k_tile8_fused*: the four waves transform the call's blocks (two each) into the FDL ring; ends with a workgroup barrier
// that also drains the stores (the new rows are in the L2 before any wave of this workgroup loads them)
template <int N, int TT>
__device__ __forceinline__ void tile8_push(const StepArgs &a, int s, float2 (&l_buf)[TT][N / 2], float2 *l_tw, int wave, int lane)
{
    constexpr int M = N / 2, WAVES = kThreads / kWave, PER = TT / WAVES;
    TwiddleStage<N, kThreads> tws;
    tws.load(a.tw);
#pragma unroll
    for (int r = 0; r < PER; r++) {
        const int j = wave + WAVES * r;
        if (j < a.T) wave_load_block<M>(a.in + (long) s * a.in_stride + (long) j * a.B, a.B, l_buf[j], lane);
    }
    tws.store(l_tw);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PER; r++) {
        const int j = wave + WAVES * r;
        if (j < a.T) {
            wave_fft<M, false>(l_buf[j], l_tw, lane);
            float2 *dst = a.X + ((size_t) s * a.R + (a.head + j) % a.R) * M;
            wave_split_forward<M>(l_buf[j], l_tw, lane, [&](int b, float2 v) { dst[b] = v; });
        }
    }
    __syncthreads();
}

// phase D: the eight summed spectra in l_buf -> inverse FFTs (two per wave, fp/convolution.cpp:206), overlap-add (:209-213):
// out[j] = y_j[0:B] + y_(j-1)[B:2B], block 0 adds the overlap the previous call saved, the last block's second half is saved
template <int N, int TT>
__device__ __forceinline__ void tile8_ifft_ola(const StepArgs &a, int s, float2 (&l_buf)[TT][N / 2], const float2 *l_tw, int wave, int lane)
{
    constexpr int M = N / 2, WAVES = kThreads / kWave, PER = TT / WAVES;
    constexpr int VPL = M > kWave ? M / kWave : 1;     // samples per lane of a half block (B <= M)
    const float scale = 1.0f / (float) N;
#pragma unroll
    for (int r = 0; r < PER; r++) {
        const int j = wave + WAVES * r;
        if (j < a.T) {
            wave_split_inverse<M>(l_buf[j], l_tw, lane);
            wave_fft<M, true>(l_buf[j], l_tw, lane);
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PER; r++) {
        const int j = wave + WAVES * r;
        if (j < a.T) {
            const float *y = reinterpret_cast<const float *>(l_buf[j]);
            const float *yp = reinterpret_cast<const float *>(l_buf[j > 0 ? j - 1 : 0]);
            const float *o = a.ovl + (size_t) s * a.B;           // (kept out of registers until here: the tile needs them all)
            float *out = a.out + (long) s * a.out_stride + (long) j * a.B;
#pragma unroll
            for (int i = 0; i < VPL; i++) {
                const int n = lane + i * kWave;
                if (n < a.B) out[n] = y[n] * scale + (j > 0 ? yp[a.B + n] * scale : o[n]);
            }
            if (j == a.T - 1) {
                float *on = a.ovl_next + (size_t) s * a.B;
                for (int n = lane; n < a.B; n += kWave) on[n] = y[a.B + n] * scale;
            }
        }
    }
}

template <int N, bool NT>
__global__ __launch_bounds__(kThreads, 4) void k_tile8_fused(StepArgs a)
{
    using G = MacGeom<N>;
    constexpr int M = N / 2, TT = 8, U = 4;
    static_assert(G::CPT == 1 && N >= 256, "one column block per thread");
    __shared__ float2 l_tw[TwLds<N>::SIZE];
    __shared__ float2 l_buf[TT][M];
    __shared__ float4 l_red[G::STRIPES > 1 ? G::STRIPES - 1 : 1][TT][G::LW];
    const int s = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave), lane = threadIdx.x % kWave;
    const int2 si = a.stream_info[s];
    const int set = si.x, np = si.y;
    IRB_STAMP(a, 0);
    tile8_push<N, TT>(a, s, l_buf, l_tw, wave, lane);                                  // ---- A
    // ---- B: the register tile over the stream's partitions, stripes = partition sub-ranges of whole ring periods
    const int col = threadIdx.x % G::LW;
    const int stripe = G::LW >= kWave ? __builtin_amdgcn_readfirstlane(threadIdx.x / G::LW) : (int) (threadIdx.x / G::LW);
    const int nG = (np + kTilePeriod - 1) / kTilePeriod;
    const int sb = (int) ((long) nG * stripe / G::STRIPES) * kTilePeriod, se = (int) ((long) nG * (stripe + 1) / G::STRIPES) * kTilePeriod;
    const float4 *H4 = reinterpret_cast<const float4 *>(a.H + (size_t) set * a.P * M);
    const float4 *X4 = reinterpret_cast<const float4 *>(a.X + (size_t) s * a.R * M);
    v2f_t acc[TT][2];
#pragma unroll
    for (int j = 0; j < TT; j++) { acc[j][0] = v2f_t{0.f, 0.f}; acc[j][1] = v2f_t{0.f, 0.f}; }
    if (sb < se) {
        const bool wave_has_col0 = G::LW >= kWave ? __builtin_amdgcn_readfirstlane(col) == 0 : true;
        if (wave_has_col0) tile_run<N, TT, U, NT, true>(H4, X4, col, sb, se, 0, a.head, a.R, acc, col == 0);
        else               tile_run<N, TT, U, NT, false>(H4, X4, col, sb, se, 0, a.head, a.R, acc, false);
    }
    // ---- C: stripes -> eight spectra in l_buf
    if constexpr (G::STRIPES > 1) {
        if (stripe > 0) {
#pragma unroll
            for (int j = 0; j < TT; j++) l_red[stripe - 1][j][col] = make_float4(acc[j][0].x, acc[j][0].y, acc[j][1].x, acc[j][1].y);
        }
        __syncthreads();
    }
    if (stripe == 0) {
#pragma unroll
        for (int j = 0; j < TT; j++) {
            float4 v = make_float4(acc[j][0].x, acc[j][0].y, acc[j][1].x, acc[j][1].y);
#pragma unroll
            for (int t = 0; t < G::STRIPES - 1; t++) {
                const float4 w = l_red[t][j][col];
                v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
            }
            reinterpret_cast<float4 *>(l_buf[j])[col] = v;
        }
    }
    __syncthreads();
    tile8_ifft_ola<N, TT>(a, s, l_buf, l_tw, wave, lane);                              // ---- D
    IRB_STAMP(a, 10);
}

// ---------------------------------------------------------------------------------------------
// k_ifft_ola_run: inverse FFT + overlap-add of the time-tiled path without the duplicate transform.
// A workgroup owns NB consecutive output blocks of one stream and runs NB + 1 waves: wave w transforms block
// jb - 1 + w (wave 0: the predecessor of the run, needed only for its tail; for jb = 0 it fetches the saved
// overlap instead); after one barrier wave w >= 1 adds its own first half to the second half its neighbour left
// in LDS (reference fp/convolution.cpp:209-213).  NB + 1 transforms per NB blocks (round 2: 2 NB), critical
// path one transform.  grid = S * ceil(T / NB).
// ---------------------------------------------------------------------------------------------
// blocks per workgroup: LDS = (NB + 1) * N * 4 + 3 N * 2 bytes must stay under the 64 KiB static limit
template <int N> struct OlaGeom { static constexpr int NB = N <= 1024 ? 4 : N == 2048 ? 2 : 1; };

template <int N, int NB>
__global__ __launch_bounds__(kWave * (NB + 1)) void k_ifft_ola_run(StepArgs a, int S)
{
    constexpr int M = N / 2;
    constexpr int VPL = M > kWave ? M / kWave : 1;     // samples per lane of a half block (B <= M)
    __shared__ float2 l_tw[TwLds<N>::SIZE];
    __shared__ float2 l_buf[NB + 1][M];
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    // this launch covers blocks [t_first, t_first + t_count) of the call (t_first a multiple of NB)
    const int runs = (a.t_count + NB - 1) / NB;
    const int s = blockIdx.x / runs, jb = a.t_first + (blockIdx.x % runs) * NB;
    const int j = jb - 1 + wave;                       // this wave's block
    const bool live = j < a.t_first + a.t_count && (j >= 0);
    const float scale = 1.0f / (float) N;
    float2 *buf = l_buf[wave];
    float *yw = reinterpret_cast<float *>(buf);
    TwiddleStage<N, kWave * (NB + 1)> tws;
    tws.load(a.tw);
    float ov[VPL];
    if (j < 0) {                                       // wave 0 of the first run: the overlap saved by the previous call
        const float *o = a.ovl + (size_t) s * a.B;
#pragma unroll
        for (int i = 0; i < VPL; i++) ov[i] = o[min(lane + i * kWave, a.B - 1)];
    }
    if (live) {
        const int r = j >= a.reg_t0[1] ? 1 : 0;                                              // the region of block j (wave-uniform)
        const int nq = a.reg_nq[r];
        wave_sum_rows<N>(a.part + a.reg_off[r] + ((size_t) s * (a.reg_t0[r + 1] - a.reg_t0[r]) + (j - a.reg_t0[r])) * nq * M, nq, buf, lane);
    }
    tws.store(l_tw);
    __syncthreads();
    if (live) {
        wave_split_inverse<M>(buf, l_tw, lane);
        wave_fft<M, true>(buf, l_tw, lane);
    } else if (j < 0) {
#pragma unroll
        for (int i = 0; i < VPL; i++) {
            const int n = lane + i * kWave;
            if (n < a.B) yw[a.B + n] = ov[i] * (float) N;     // as an unscaled tail, like the transformed ones
        }
    }
    __syncthreads();
    if (!live) return;
    const float *y = reinterpret_cast<const float *>(buf);
    if (wave >= 1) {
        const float *yp = reinterpret_cast<const float *>(l_buf[wave - 1]);
        float *out = a.out + (long) s * a.out_stride + (long) j * a.B;
        const long room = a.out_len - (long) j * a.B;              // samples of this block that may be written (a result buffer's ragged end)
#pragma unroll
        for (int i = 0; i < VPL; i++) {
            const int n = lane + i * kWave;
            if (n < a.B && n < room) {
                const float v = y[n] * scale + yp[a.B + n] * scale;
                // (staged one-shot call: written THROUGH to the host's staging -- an ordinary store may stay in the L2 until the
                // kernel ends, long after the chunk's word below has told the host to read)
                if (a.ck_flag) __hip_atomic_store(out + n, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                else out[n] = v;
            }
        }
    }
    if (j == a.T - 1) {
        float *on = a.ovl_next + (size_t) s * a.B;
        for (int n = lane; n < a.B; n += kWave) on[n] = y[a.B + n] * scale;
    }
    if (a.ck_flag && wave >= 1) {
        // staged one-shot call: block j of stream s has left for the host's staging.  Its samples were stored write-through (system
        // scope), so "left" = acknowledged = vmcnt(0); a release FENCE here would write the whole L2 back, per wave (measured: this
        // kernel 173 us instead of 86).  Then the count (relaxed: ordering comes from the wait in
        // front of it and from the data dependence on its result); the count's last arriver clears it for the next piece and
        // raises the chunk's word -- issued after every other wave's samples were acknowledged, so behind them on the link.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            const int ck = j / a.ck_blocks;
            const unsigned expect = (unsigned) S * (unsigned) min(a.ck_blocks, a.T - ck * a.ck_blocks);
            const unsigned old = __hip_atomic_fetch_add(a.ck_cnt + ck, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1u == expect) {
                __hip_atomic_store(a.ck_cnt + ck, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.ck_flag + 16 * ck, a.ck_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

}  // namespace irb
