// fdl_mfma.hpp -- the time-tiled multiply-accumulate of long calls on the matrix cores.  Included by fdl_kernels.hpp.
//
// Along the time axis a bin of the frequency-domain delay line is a 1-D convolution,
//     Y[t] = sum_{m < np} H[m] * X[t - m]                     (fp/convolution.cpp:185-205, one complex bin, all blocks of a call)
// and a 1-D convolution cut into segments of 16 is a sum of dense 16 x 16 matrix products: with O[n][i] = Y[16 n + i],
//     O[n, :] = sum_j  Xc[n - j, :] * M_j,     M_j[k][i] = H[16 j + i - k]   (Toeplitz; H = 0 outside [0, np)),
//                                              Xc[c][k]  = X[16 c + k]       (the FDL's time axis folded into rows of 16).
// Every M_j is dense (one triangle comes from chunk j of the IR spectra, the other from chunk j - 1), so the products carry no
// structural zeros apart from the padding of np to whole chunks.  This is genuinely GEMM-shaped fp32 work -- per bin a
// [T/16 x 16 J] x [16 J x 16] product -- on v_mfma_f32_16x16x4_f32 (exact fp32, the vector ALU's 157.3 TFLOP/s peak; measured with
// tools/mfma_peak.hip: 0.89-0.97 of it from one wave per SIMD).
//
// Round 4: NO LDS staging and NO barrier in the loop.  Round 3's kernel brought the FDL rows [slot][bin] and the IR rows
// [partition][bin] through a double-buffered LDS image, one barrier per stage of 9 chunks: 0.50 of the peak at configs[1], the
// rest went to the first stage's burst, ~700 cycles per barrier round, a window re-loaded per stage and LDS operand reads
// between the MFMAs.  Here the two operands are kept a SECOND time in the layout the instruction wants, written once by the
// kernels that produce them:
//     Xt[stream][bin][xt_hist + t]   the FDL with time contiguous per bin (k_push_fft_batch_t; the blocks before the call's
//                                    first one sit in [0, xt_hist): k_xt_history rebuilds them from the ring when there are any)
//     Ht[set][bin][16 + m]           the IR spectra with the tap index contiguous per bin, 16 zero taps in front, zeros behind
//                                    (k_rows_to_columns at irb_fdl_set_ir*)
// A wave owns ONE bin of one tile of 256 output blocks and one range of chunks, and nothing it reads is shared with another
// wave through LDS: per chunk it issues 16 MFMAs and
//   * 4 buffer loads of 8 B per lane for the H operand: lane (i, kq), k group g reads tap 16 j + i - 4 g - kq -- 31 consecutive taps
//     of its bin, 2-3 cache lines, the same for every tile (L2 hits); negative and too large taps are the layout's zeros;
//   * 4 buffer loads for the ONE new row of 16 blocks the X operand needs: operand lane (n, kq) of chunk j + 1 holds
//     Xc[s0 + n - j - 1][4 g + kq], which is what lane (n - 1, kq) held for chunk j -- a DPP row shift (v_mov_b32 row_shr:1) moves the
//     window along, and the lanes n = 0, which have no left neighbour, keep the freshly loaded value (bound_ctrl off).
// Loads are issued one chunk of MFMAs ahead of their use into the register set the previous chunk has just released (two sets swap
// roles, the loop is unrolled by two: no copies); the row step sits in an SGPR and the k group in the instruction's immediate
// offset, so a chunk costs 8 VMEM + 8 DPP + a handful of scalar instructions next to its 16 MFMAs, in ~50 VGPRs.  The only LDS use
// and the only barrier are at the end: the 256 x 16 results of the sixteen waves (bins) of a workgroup are transposed so that each
// partial-spectrum row leaves as one whole 128-byte line (part[s][t][split][M], summed over splits by k_ifft_ola_run).
//
// Not bit-identical to the vector kernels (another summation order); parity with the oracle at the same tolerance (tests).
// The index arithmetic (tiles, splits, operand positions, the shifting window, the transposed layouts) is restated in
// tests/kernel_model.py (mfma_time_axis) and checked against the direct sum on a machine without a GPU.
// Non-finite input: 0 * X products exist for X blocks next to the ones an output really depends on (zero taps in front of
// and behind the IR, up to 255 blocks later / 31 blocks earlier), so an Inf or NaN sample can spread further along the time
// axis than in the reference; finite input is unaffected.  Xt is zeroed at create and only ever holds spectra, so the blocks
// a tile reads beyond the end of a call are finite.
#pragma once

namespace irb {

typedef float v4f_t __attribute__((ext_vector_type(4)));

constexpr int kMfmaSeg = 16;                           // segment length = MFMA tile edge
constexpr int kMfmaTile = kMfmaSeg * kMfmaSeg;         // output blocks per workgroup tile
constexpr int kMfmaBins = 16;                          // bins per workgroup = waves per workgroup
constexpr int kMfmaThreads = kMfmaBins * kWave;        // 1024
constexpr int kMfmaOS = kMfmaBins + 1;                 // row stride of the output tile in LDS

// chunks of the Toeplitz sum for np partitions: j = 0 .. J-1 covers every m = 16 j + i - k in [0, np)
__host__ __device__ constexpr int mfma_chunks(int np) { return np > 0 ? (np + kMfmaSeg - 2) / kMfmaSeg + 1 : 0; }
// blocks of history in front of a call's first block in Xt: chunk J - 1 of tile 0 reads back to -16 (J - 1), the loads that run
// ahead of the last chunk to -16 (J + 1) -- everything the kernel addresses exists, nothing depends on buffer range checks
__host__ __device__ constexpr int mfma_hist(int P) { return kMfmaSeg * (mfma_chunks(P) + 2); }
// taps per bin row of Ht: 16 zeros, the taps, zeros up to the last index the run-ahead loads read (16 + 16 (J + 1) + 15)
__host__ __device__ constexpr int mfma_taps(int P) { return kMfmaSeg * (mfma_chunks(P) + 3); }

// ---------------------------------------------------------------------------------------------
// k_rows_to_columns: rows [r0, r0 + nrows) of a [row][M] spectrum array (ring addressed: physical row (first + r) mod R) become
// columns [c0 + r] of a [M][ldc] array; rows at or beyond `valid` are written as zeros.  Used for Ht (IR spectra -> tap-contiguous,
// at set-up) and for Xt's history (the FDL ring's last blocks -> time-contiguous, before a matrix-core call that continues a stream).
// A workgroup transposes 16 rows x 64 bins through LDS: 512-byte pieces in, whole 128-byte lines out.
// grid = (ceil(nrows / 16), M / 64 or 1, sets), block = 256.
// ---------------------------------------------------------------------------------------------
template <int M>
__global__ __launch_bounds__(256) void k_rows_to_columns(const float2 *__restrict__ src, long src_set_stride, int first, int R, int valid,
                                                          float2 *__restrict__ dst, long dst_set_stride, int ldc, int c0, int nrows)
{
    constexpr int BW = M < 64 ? M : 64;                // bins per workgroup
    __shared__ float2 l_t[16][BW + 1];
    const int r0 = blockIdx.x * 16, b0 = blockIdx.y * BW, set = blockIdx.z;
    const float2 *s = src + (size_t) set * src_set_stride;
    float2 *d = dst + (size_t) set * dst_set_stride;
    for (int e = threadIdx.x; e < 16 * BW; e += 256) {
        const int r = e / BW, b = e % BW;
        float2 v = make_float2(0.f, 0.f);
        if (r0 + r < nrows && r0 + r < valid) {
            int pr = first + r0 + r;
            if (R > 0) { pr %= R; if (pr < 0) pr += R; }
            v = s[(size_t) pr * M + b0 + b];
        }
        l_t[r][b] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 16 * BW; e += 256) {
        const int b = e / 16, r = e % 16;
        if (r0 + r < nrows) d[(size_t) (b0 + b) * ldc + c0 + r0 + r] = l_t[r][b];
    }
}

// ---------------------------------------------------------------------------------------------
// k_push_fft_batch_t: the forward FFTs of blocks [t_first, t_first + t_count) of a call (k_push_fft_batch's work) for engines
// whose long calls multiply on the matrix cores: every spectrum is written twice -- into the FDL ring (what every other kernel
// reads and what the next call's history is rebuilt from) and, through LDS, into Xt with time contiguous per bin.
// A workgroup = NBW consecutive blocks of one stream, one wave each; Xt leaves in pieces of NBW * 8 bytes (whole 128-byte lines
// for N <= 512).  grid = S * ceil(t_count / NBW); t_first is a multiple of NBW.
// ---------------------------------------------------------------------------------------------
template <int N> struct PushTGeom { static constexpr int NBW = N <= 512 ? 16 : N == 1024 ? 8 : N == 2048 ? 4 : 2; };

template <int N>
__global__ __launch_bounds__(kWave * PushTGeom<N>::NBW) void k_push_fft_batch_t(StepArgs a, int S)
{
    constexpr int M = N / 2, NBW = PushTGeom<N>::NBW;
    __shared__ float2 l_tw[TwLds<N>::SIZE];
    __shared__ float2 l_buf[NBW][M + 1];               // (+1: the transposed read below walks the block index at a fixed bin)
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const int groups = (a.t_count + NBW - 1) / NBW;
    const int s = blockIdx.x / groups, j0 = a.t_first + (blockIdx.x % groups) * NBW;
    const int j = j0 + wave;
    const bool live = j < a.t_first + a.t_count;
    TwiddleStage<N, kWave * NBW> tws;
    tws.load(a.tw);
    if (live) wave_load_block_upto<M>(a.in + (long) s * a.in_stride + (long) j * a.B, a.B, a.in_len - (long) j * a.B, l_buf[wave], lane);
    tws.store(l_tw);
    __syncthreads();
    if (live) {
        wave_fft<M, false>(l_buf[wave], l_tw, lane);
        int slot = (a.head + j) % a.R;
        float2 *dst = a.X + ((size_t) s * a.R + slot) * M;
        // the split reads bins b and M - b of the transform and writes the same two positions: in place, after all reads of the wave
        float2 *lb = l_buf[wave];
        wave_split_forward<M>(lb, l_tw, lane, [&](int b, float2 v) { dst[b] = v; lb[b] = v; });
    }
    __syncthreads();
    // Xt[s][bin][xt_hist + j0 .. + NBW): thread -> (bin, piece of 4 blocks)
    constexpr int Q = NBW >= 4 ? NBW / 4 : 1, E = NBW >= 4 ? 4 : NBW;
    float2 *xt = a.Xt + (size_t) s * M * a.xt_len + a.xt_hist + j0;
    for (int e = threadIdx.x; e < M * Q; e += kWave * NBW) {
        const int bin = e / Q, q = e % Q;
#pragma unroll
        for (int i = 0; i < E; i++)
            if (j0 + E * q + i < a.t_first + a.t_count) xt[(size_t) bin * a.xt_len + E * q + i] = l_buf[E * q + i][bin];
    }
}

// 8-byte buffer load: per-lane byte offset in a VGPR, row step in an SGPR, k group in the immediate; out of range reads 0
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
template <int IMM>
__device__ __forceinline__ float2 buf_ld8(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const v2u_t v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff + IMM, 0);
    return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}

// one component of the X operand moves one segment along: lane (n, kq) takes lane (n - 1, kq)'s value, lanes n = 0 keep `fresh`
__device__ __forceinline__ float shift_in(float fresh, float window)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fresh), __float_as_int(window), 0x111 /* row_shr:1 */, 0xf, 0xf, false));
}

// The chunks [jb, je) of one wave: its bin of one tile.  PACKED = bin 0 (DC and Nyquist: two real products).
//   x operand of chunk j, k group g, lane (n, kq):  Xt[xt_hist + 256 tile + 16 (n - j) + 4 g + kq]
//   h operand                       lane (i, kq):  Ht[16 + 16 j + i - 4 g - kq]
template <bool PACKED>
__device__ __forceinline__ void mfma_chunk_run(__amdgpu_buffer_rsrc_t rx, __amdgpu_buffer_rsrc_t rh, int lane, int x_elem0, int jb, int je,
                                               v4f_t &acc_r, v4f_t &acc_n, v4f_t &acc_i)
{
    const int n = lane % kMfmaSeg, kq = lane / kMfmaSeg;
    const unsigned vx_win = (unsigned) (kMfmaSeg * n + kq) * 8u, vx_new = (unsigned) kq * 8u, vh = (unsigned) (n - kq + 3) * 8u;
    // scalar byte offsets: the window of chunk jb, the new row of chunk j (element x_elem0 - 16 j), the taps of chunk j
    auto sx = [&](int j) { return (unsigned) (x_elem0 - kMfmaSeg * j) * 8u; };
    auto sh = [&](int j) { return (unsigned) (kMfmaSeg * j) * 8u; };
    // Two register sets that swap roles every chunk (the loop is unrolled by two, so nothing is ever copied): while set A is
    // multiplied, set B holds the next chunk's taps and -- in its lanes n = 0 -- the next chunk's new row of blocks; the DPP shift
    // then completes B's window from A's in place, and A, now free, receives the loads of the chunk after that.  A load is issued
    // one chunk of MFMAs (512 cycles alone, ~2000 with the SIMD's other waves) before its first use.
    float2 xa[4], ha[4], xb[4], hb[4];
    auto load_h = [&](float2 (&h)[4], int j) {
        h[0] = buf_ld8<(13 - 0) * 8>(rh, vh, sh(j)); h[1] = buf_ld8<(13 - 4) * 8>(rh, vh, sh(j));
        h[2] = buf_ld8<(13 - 8) * 8>(rh, vh, sh(j)); h[3] = buf_ld8<(13 - 12) * 8>(rh, vh, sh(j));
    };
    auto load_x = [&](float2 (&x)[4], unsigned v, int j) {
        x[0] = buf_ld8<0>(rx, v, sx(j)); x[1] = buf_ld8<32>(rx, v, sx(j)); x[2] = buf_ld8<64>(rx, v, sx(j)); x[3] = buf_ld8<96>(rx, v, sx(j));
    };
    auto multiply = [&](const float2 (&x)[4], const float2 (&h)[4]) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
            if constexpr (PACKED) {
                acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(x[g].x, h[g].x, acc_r, 0, 0, 0);
                acc_i = __builtin_amdgcn_mfma_f32_16x16x4f32(x[g].y, h[g].y, acc_i, 0, 0, 0);
            } else {
                acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(x[g].x, h[g].x, acc_r, 0, 0, 0);
                acc_i = __builtin_amdgcn_mfma_f32_16x16x4f32(x[g].y, h[g].x, acc_i, 0, 0, 0);
                acc_n = __builtin_amdgcn_mfma_f32_16x16x4f32(x[g].y, h[g].y, acc_n, 0, 0, 0);      // subtracted at the end: no sign flip in the loop
                acc_i = __builtin_amdgcn_mfma_f32_16x16x4f32(x[g].x, h[g].y, acc_i, 0, 0, 0);
            }
        }
    };
    // chunk j is multiplied out of (xc, hc); (xn, hn) -- loaded for chunk j + 1 -- becomes its window; then (xc, hc) is reloaded for chunk j + 2
#ifndef IRB_MFMA_EXP
#define IRB_MFMA_EXP 0                                 // diagnostic builds (tools/mfma_stamps.hip): 1 no window shift, 2 no loads in the loop -- timing only, wrong sums
#endif
    auto step = [&](int j, float2 (&xc)[4], float2 (&hc)[4], float2 (&xn)[4]) {
        __builtin_amdgcn_sched_barrier(0);
        multiply(xc, hc);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(IRB_MFMA_EXP & 1)) {
#pragma unroll
            for (int g = 0; g < 4; g++) xn[g] = make_float2(shift_in(xn[g].x, xc[g].x), shift_in(xn[g].y, xc[g].y));
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(IRB_MFMA_EXP & 2)) {
            load_x(xc, vx_new, j + 2);                 // (a chunk past the range reads blocks / taps that exist in the padded layouts: never used)
            load_h(hc, j + 2);
        }
    };
    load_x(xa, vx_win, jb);
    load_h(ha, jb);
    load_x(xb, vx_new, jb + 1);
    load_h(hb, jb + 1);
    int j = jb;
#pragma unroll 1
    for (; j + 1 < je; j += 2) {
        step(j, xa, ha, xb);
        step(j + 1, xb, hb, xa);
    }
    if (j < je) { __builtin_amdgcn_sched_barrier(0); multiply(xa, ha); }
}

// grid = S * tile_count * splits * (M / 16); block = 1024.  Partial spectra: part[s][t - t_base][split][M], part_T blocks per
// stream (a call may be multiplied in two launches -- whole rounds of workgroups, then the remaining tiles with more partition
// splits -- each with its own partial-spectrum region; see launch_tiled).
template <int N>
__global__ __launch_bounds__(kMfmaThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_cmac_mfma(StepArgs a)
{
    constexpr int M = N / 2;
    static_assert(M % kMfmaBins == 0, "bins per workgroup");
    constexpr int GROUPS = M / kMfmaBins;
    __shared__ float2 l_o[kMfmaTile * kMfmaOS];
    const int wg = xcd_contiguous_id(blockIdx.x, gridDim.x);       // the bin groups and splits of one tile share an XCD's L2 (the same FDL lines)
    const int bg = wg % GROUPS, split = (wg / GROUPS) % a.splits, tile = a.tile_first + (wg / (GROUPS * a.splits)) % a.tile_count,
              s = wg / (GROUPS * a.splits * a.tile_count);
    const int2 si = a.stream_info[s];
    const int set = si.x, np = si.y;
    const int J = mfma_chunks(np);
    const int jb = (int) ((long) J * split / a.splits), je = (int) ((long) J * (split + 1) / a.splits);
    const int b0 = bg * kMfmaBins;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid / kWave), lane = tid % kWave;
    IRB_STAMP(a, 0);
#ifdef IRB_STAMPS
    if (threadIdx.x == 0) {                            // where this workgroup runs: HW_ID (wave / SIMD / CU / SE) and the XCD
        a.stamps[(size_t) blockIdx.x * 16 + 14] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
        a.stamps[(size_t) blockIdx.x * 16 + 15] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));
    }
#endif
    const int bin = b0 + wave;                                     // wave-uniform: the descriptors below live in scalar registers
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(a.Xt + ((size_t) s * M + bin) * a.xt_len, 0, a.xt_len * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(a.Ht) + ((size_t) set * M + bin) * a.ht_len, 0, a.ht_len * 8, 0x00020000);
    v4f_t acc_r, acc_n, acc_i;                                     // O[n = 4 kq + v][i = lane % 16]: re = acc_r - acc_n (Xr Hr and Xi Hi apart), im
#pragma unroll
    for (int v = 0; v < 4; v++) { acc_r[v] = 0.f; acc_n[v] = 0.f; acc_i[v] = 0.f; }
    const int x_elem0 = a.xt_hist + kMfmaTile * tile;              // Xt element of the tile's first block
    if (jb < je) {
        if (bin == 0) mfma_chunk_run<true>(rx, rh, lane, x_elem0, jb, je, acc_r, acc_n, acc_i);
        else          mfma_chunk_run<false>(rx, rh, lane, x_elem0, jb, je, acc_r, acc_n, acc_i);
    }
    IRB_STAMP(a, 2);
    // ---- store: accumulator register v of lane (i, kq) is O[n = 4 kq + v][i], i.e. block 256 tile + 16 n + i of the call; through
    // LDS, so that each output row leaves as one whole 128-byte line
    const int i = lane % kMfmaSeg, kq = lane / kMfmaSeg;
#pragma unroll
    for (int v = 0; v < 4; v++) l_o[(kMfmaSeg * (4 * kq + v) + i) * kMfmaOS + wave] = make_float2(acc_r[v] - acc_n[v], acc_i[v]);
    __syncthreads();
    const int sbin = tid % kMfmaBins, srow = tid / kMfmaBins;
#pragma unroll
    for (int it = 0; it < kMfmaTile / (kMfmaThreads / kMfmaBins); it++) {
        const int tl = srow + (kMfmaThreads / kMfmaBins) * it, t = kMfmaTile * tile + tl;
        if (t < a.T) a.part[(((size_t) s * a.part_T + (t - a.t_base)) * a.nq + split) * M + b0 + sbin] = l_o[tl * kMfmaOS + sbin];
    }
    IRB_STAMP(a, 7);
}

}  // namespace irb
