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
//     Xt[stream][bin][xt_hist + t]   the FDL with time contiguous per bin, re / im planar inside every segment of 16 blocks (mfma_xt_float)
//                                    (k_push_fft_batch_t; the blocks before the call's first one sit in [0, xt_hist): rebuilt from
//                                    the ring by extra workgroups of that launch when there are any)
//     Ht[set][bin][16 + m]           the IR spectra with the tap index contiguous per bin, 16 zero taps in front, zeros behind
//                                    (k_rows_to_columns at irb_fdl_set_ir*)
// A wave owns ONE bin of one tile of 256 output blocks and one range of chunks, and nothing it reads is shared with another
// wave through LDS memory.  Per chunk it issues 12 MFMAs (a complex product as three real ones), THREE small loads (one of the chunk's
// 31 taps by half the lanes, two of 16 bytes by the four lanes that start a new segment of the X window), 8 ds_bpermute that hand
// the taps to the lanes that multiply with them, 8 DPP moves that shift the X window by one segment and 4 packed adds -- see
// mfma_chunk_run.  The row step of a load sits in an SGPR; ~110 VGPRs (one workgroup of 16 waves per CU).  The only LDS memory and
// the only barrier are at the end: the
// 256 x 16 results of the sixteen waves (bins) of a workgroup are transposed so that each partial-spectrum row leaves as one whole
// 128-byte line (part[s][t][split][M], summed over splits by k_ifft_ola_run).
//
// Not bit-identical to the vector kernels (another summation order); parity with the oracle at the same tolerance (tests).
// The index arithmetic (tiles, splits, operand positions, the shifting window, the transposed layouts) is restated in
// tests/kernel_model.py (mfma_time_axis) and checked against the direct sum on a machine without a GPU.
// Non-finite input: 0 * X products exist for X blocks next to the ones an output really depends on (zero taps in front of
// and behind the IR, up to 255 blocks later / 31 blocks earlier), so an Inf or NaN sample can spread further along the time
// axis than in the reference WITHIN a call; finite input is unaffected.  Between calls nothing leaks: the only blocks beyond a call's
// last one that its kept outputs touch are the rest of the last segment of 16, and k_push_fft_batch_t writes those as zeros.
#pragma once

#ifndef IRB_MFMA_EXP
#define IRB_MFMA_EXP 0                                 // diagnostic builds (tools/mfma_stamps.hip): 2 = no loads in the loop -- timing only, wrong sums
#endif

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
__host__ __device__ constexpr int mfma_hist(int P) { return kMfmaSeg * (mfma_chunks(P) + 3); }
// taps per bin row of Ht: 16 zeros, the taps, zeros up to the last index the run-ahead loads read (16 + 16 (J + 1) + 15)
__host__ __device__ constexpr int mfma_taps(int P) { return kMfmaSeg * (mfma_chunks(P) + 4); }

// Xt inside every segment of 16 blocks (32 floats): component c (0 re, 1 im) of block 4 g + kq is float 8 kq + 4 c + g, so that the
// four k groups an operand lane (n, kq) needs -- blocks 4 g + kq, g = 0 .. 3, of one segment -- are 16 contiguous bytes of real
// parts followed by 16 of imaginary parts (two dwordx4 loads; re + im of two k groups is then one v_pk_add_f32)
__host__ __device__ constexpr int mfma_xt_float(int t, int c) { return ((t & ~15) << 1) | ((t & 3) << 3) | (c << 2) | ((t >> 2) & 3); }
// the 8-byte slot p < 16 of a segment therefore holds component (p >> 1) & 1 of the blocks mfma_xt_slot_block(p) and that + 4
__host__ __device__ constexpr int mfma_xt_slot_block(int p) { return ((p & 1) << 3) | (p >> 2); }
__device__ __forceinline__ float mfma_component(const float2 &v, int c) { return c ? v.y : v.x; }
// 8 bytes written through (non-temporal): rows the NEXT kernel reads once -- they leave during the kernel, not in its end-of-kernel write-back
__device__ __forceinline__ void nt_store2(float2 *dst, float2 v)
{
    v2f_t q = {v.x, v.y};
    __builtin_nontemporal_store(q, reinterpret_cast<v2f_t *>(dst));
}

// ---------------------------------------------------------------------------------------------
// k_rows_to_columns: rows [0, nrows) of a [row][M] spectrum array become columns [c0 + r] of a [M][ldc] array; rows at or beyond
// `valid` are written as zeros.  Builds Ht (IR spectra -> tap-contiguous) at set-up.  A workgroup transposes 16 rows x 64 bins
// through LDS: 512-byte pieces in, whole 128-byte lines out.  grid = (ceil(nrows / 16), M / 64 or 1, sets), block = 256.
// (Xt's history columns are rebuilt from the FDL ring the same way by extra workgroups of k_push_fft_batch_t.)
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
// for N <= 512).  grid = S * ceil(t_count / NBW) (+ S * (xt_hist / 16) * max(1, M / 64) workgroups that rebuild Xt's history columns
// when the call continues a stream); t_first is a multiple of NBW.
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
    if ((int) blockIdx.x >= S * groups) {
        // extra workgroups of a call that continues a stream (a.hist_first >= 0): the xt_hist blocks before the call's first one go
        // from the FDL ring into Xt's history columns (k_rows_to_columns' transpose, plus the segment permutation), 16 ring
        // rows x 64 bins per workgroup through LDS.
        // (xt_hist = 16 (J + 3) reaches further back than the ring keeps apart from this call's own rows when the call is within
        // xt_hist - Ppad - 8 blocks of the handle's longest: the oldest history rows would share ring slots with blocks the other
        // workgroups of this launch are writing.  Those rows are more than R - T >= Ppad + 8 > np blocks old -- they only ever meet
        // the zero taps behind the IR -- so they are not read at all: zeros go into their columns, and nothing of this call's own
        // input, finite or not, can turn up in front of it.)
        constexpr int BW = M < 64 ? M : 64, BY = M / BW;
        float2 (*l_t)[BW + 1] = reinterpret_cast<float2 (*)[BW + 1]>(&l_buf[0][0]);
        const int w = blockIdx.x - S * groups, r0 = (w % (a.xt_hist / 16)) * 16, b0 = ((w / (a.xt_hist / 16)) % BY) * BW, s = w / ((a.xt_hist / 16) * BY);
        const float2 *src = a.X + (size_t) s * a.R * M;
        float2 *dst = a.Xt + (size_t) s * M * a.xt_len;
        for (int e = threadIdx.x; e < 16 * BW; e += kWave * NBW) {
            const int r = e / BW, b = e % BW;
            const bool own_row = a.xt_hist - (r0 + r) + a.T > a.R;          // its ring slot holds a block of THIS call
            l_t[r][b] = own_row ? make_float2(0.f, 0.f) : src[(size_t) ((a.hist_first + r0 + r) % a.R) * M + b0 + b];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 16 * BW; e += kWave * NBW) {
            const int b = e / 16, q = e % 16, c = (q >> 1) & 1, t = mfma_xt_slot_block(q);
            dst[(size_t) (b0 + b) * a.xt_len + r0 + q] = make_float2(mfma_component(l_t[t][b], c), mfma_component(l_t[t + 4][b], c));
        }
        return;
    }
    const int s = blockIdx.x / groups, j0 = a.t_first + (blockIdx.x % groups) * NBW;
    const int j = j0 + wave;
    const bool live = j < a.t_first + a.t_count;
    TwiddleStage<N, kWave * NBW> tws;
    tws.load(a.tw);
    if (a.pk_flag && (long) j0 * a.B < a.in_len) {
        // staged one-shot call: the host may still be copying this workgroup's samples into the staging.  One lane polls the item's
        // word over the link (~2 us a look) for at most 2 s of the 100 MHz clock -- a host that died must not leave the grid behind;
        // what a timed-out workgroup then transforms is wrong and the status word says so.  No cache invalidate behind the wait (an
        // acquire fence per wave made this kernel twice as slow): the host only uses this form when every block is whole 128-byte
        // lines, so no line of this workgroup's samples was fetched by anyone before the word was seen (the kernel's start dropped
        // whatever earlier kernels left).
        if (threadIdx.x == 0) {
            const unsigned *w = a.pk_flag + 16 * (size_t) (j0 / a.pk_blocks);
            unsigned long long t0, t1;
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
            while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != a.pk_seq) {
                __builtin_amdgcn_s_sleep(8);
                asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
                if (t1 - t0 > 200000000ull) { __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            }
        }
        __syncthreads();
    }
    if (live) wave_load_block_upto<M>(a.in + (long) s * a.in_stride + (long) j * a.B, a.B, a.in_len - (long) j * a.B, l_buf[wave], lane);
    if (live && a.dbg_sums) wave_block_checksum<N>(l_buf[wave], lane, a.dbg_sums + (size_t) s * a.T + j);
    tws.store(l_tw);
    __syncthreads();
    if (live) {
        wave_fft<M, false>(l_buf[wave], l_tw, lane);
        int slot = (a.head + j) % a.R;
        float2 *dst = a.X + ((size_t) s * a.R + slot) * M;
        // the split reads bins b and M - b of the transform and writes the same two positions: in place, after all reads of the wave
        float2 *lb = l_buf[wave];
        wave_split_forward<M>(lb, l_tw, lane, [&](int b, float2 v) { nt_store2(dst + b, v); lb[b] = v; });
    }
    __syncthreads();
    // Xt[s][bin][xt_hist + j0 .. + NBW), inside every segment of 16 blocks in mfma_xt_float order: thread -> (bin, slot p of the
    // workgroup's blocks); consecutive threads write consecutive slots (whole 128-byte lines for NBW = 16).  The blocks between the
    // call's last one and the end of ITS segment of 16 are written as zeros: k_cmac_mfma multiplies them (by zero taps) into the last
    // outputs of the call, and what an earlier, longer call left there may be non-finite (0 * NaN; a cached one-shot engine, a
    // streaming handle after a burst -- the reference's calls are independent).
    float2 *xt = a.Xt + (size_t) s * M * a.xt_len + a.xt_hist;
    const int t_end = a.t_first + a.t_count;
    for (int e = threadIdx.x; e < M * NBW; e += kWave * NBW) {
        const int bin = e / NBW, p = e % NBW;
        if constexpr (NBW == kMfmaSeg) {               // the workgroup holds a whole segment: slot p takes one component of two of its blocks
            const int c = (p >> 1) & 1, t = mfma_xt_slot_block(p);
            float2 *dst = xt + (size_t) bin * a.xt_len + j0 + p;
            nt_store2(dst, make_float2(j0 + t < t_end ? mfma_component(l_buf[t][bin], c) : 0.f, j0 + t + 4 < t_end ? mfma_component(l_buf[t + 4][bin], c) : 0.f));
        } else {                                       // part of a segment: the two components of block j0 + p go to their (scattered) floats
            float *xf = reinterpret_cast<float *>(xt + (size_t) bin * a.xt_len);
            const bool have = j0 + p < t_end;
            xf[mfma_xt_float(j0 + p, 0)] = have ? l_buf[p][bin].x : 0.f;
            xf[mfma_xt_float(j0 + p, 1)] = have ? l_buf[p][bin].y : 0.f;
        }
    }
    if constexpr (NBW < kMfmaSeg) {
        // (the workgroup that holds the call's last block also clears the blocks of that segment behind its own)
        const int seg_end = (t_end + kMfmaSeg - 1) / kMfmaSeg * kMfmaSeg;
        if (j0 < t_end && j0 + NBW >= t_end && j0 + NBW < seg_end) {
            const int extra = seg_end - (j0 + NBW);
            for (int e = threadIdx.x; e < M * extra; e += kWave * NBW) {
                const int bin = e / extra, t = j0 + NBW + e % extra;
                float *xf = reinterpret_cast<float *>(xt + (size_t) bin * a.xt_len);
                xf[mfma_xt_float(t, 0)] = 0.f;
                xf[mfma_xt_float(t, 1)] = 0.f;
            }
        }
    }
}

// buffer loads: per-lane byte offset in a VGPR, row step in an SGPR, a constant in the immediate; everything the kernel addresses
// exists in the padded layouts
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
template <int IMM>
__device__ __forceinline__ float2 buf_ld8(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const v2u_t v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff + IMM, 0);
    return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}
template <int IMM>
__device__ __forceinline__ void buf_ld16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float2 &lo, float2 &hi)
{
    const v4u_t v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff + IMM, 0);
    lo = make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
    hi = make_float2(__uint_as_float(v.z), __uint_as_float(v.w));
}

// one component of the X operand moves one segment along: lane (n, kq) takes lane (n - 1, kq)'s value, lanes n = 0 keep `fresh`
__device__ __forceinline__ float shift_in(float fresh, float window)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fresh), __float_as_int(window), 0x111 /* row_shr:1 */, 0xf, 0xf, false));
}

// every lane fetches the value lane `byte_addr / 4` holds (the LDS crossbar: no memory, no vector-ALU slot)
__device__ __forceinline__ float2 from_lane(int byte_addr, float2 v)
{
    return make_float2(__int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v.x))),
                       __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v.y))));
}

// The chunks [jb, je) of one wave: its bin of one tile.  PACKED = bin 0 (DC and Nyquist: two real products).
//   x operand of chunk j, k group g, lane (n, kq):  block 256 tile + 16 (n - j) + 4 g + kq = Xt floats of segment x_elem0 / 16 + n - j at 8 kq + g (re), + 4 (im)
//   h operand                       lane (i, kq):  tap 16 j + i - 4 g - kq = Ht element 16 + that
// A complex product is THREE real ones (Xr Hr, Xi Hi, (Xr + Xi)(Hr + Hi); re = first - second, im = third - first - second, formed
// once after the loop from three accumulators): 12 MFMAs per chunk instead of 16.  The sums cost four v_pk_add_f32 per chunk.
// (Rounding: the imaginary part now carries the error of three sums instead of two -- measured against the oracle in the tests,
// the same tolerance.  A non-finite sample can turn into NaN where the four-product form gives Inf: include/irb_fdl.h at irb_fdl_process.)
// What a chunk fetches is what is NEW, once per wave, and the lanes share it among themselves -- the texture path (one per CU, 64 bytes
// a clock) is the resource sixteen independent waves run out of first: with every lane loading its own operands (8 loads of 512
// bytes per chunk, most of them the same few addresses) the kernel took 37 us at configs[1], with a window re-loaded from the L1
// every chunk 44 us, against 30 us with no loads at all (tools/mfma_stamps.hip, profiles/r04_mfma_stamps.txt):
//   * taps: the 31 a chunk touches (16 j - 15 .. 16 j + 15) are ONE 8-byte load by lanes 0 .. 30; lane (i, kq) then takes tap
//     i - kq - 4 g + 15 from that lane through ds_bpermute, four times (g) -- lane addresses that never change;
//   * blocks: operand lane (n, kq) of chunk j + 1 holds what lane (n - 1, kq) held for chunk j, so a DPP row shift moves the window
//     along and only the lanes n = 0 load: four lanes, 32 contiguous bytes each (mfma_xt_float keeps a lane's four k groups together).
// The lanes that have nothing to load carry a byte offset beyond the buffer (the descriptor's range check returns zeros without a
// memory access): no exec-mask branches in the loop.  Every non-MFMA instruction takes matrix-pipe time from the other three waves
// of the SIMD (ablation in profiles/NOTES.md), so the loop is exactly 12 MFMAs + 8 ds_bpermute + 8 DPP moves + 4 packed adds + 3 loads.
template <bool PACKED>
__device__ __forceinline__ void mfma_chunk_run(__amdgpu_buffer_rsrc_t rx, __amdgpu_buffer_rsrc_t rh, int lane, int x_elem0, int jb, int je,
                                               v4f_t &acc_r, v4f_t &acc_n, v4f_t &acc_i, const StepArgs &a)
{
    const int n = lane % kMfmaSeg, kq = lane / kMfmaSeg;
    constexpr unsigned kNoLoad = 0x7ffffff0u;                              // a byte offset beyond every buffer: the lane's load returns zeros and fetches nothing
    const unsigned vx = (unsigned) (kMfmaSeg * n + 4 * kq) * 8u;           // the lane's four k groups in the window's first chunk
    const unsigned vx_new = n == 0 ? (unsigned) (4 * kq) * 8u : kNoLoad;   // ... in a chunk's new segment (lanes n = 0)
    const unsigned vh = lane < 31 ? (unsigned) (lane + 1) * 8u : kNoLoad;  // lane l < 31 loads tap 16 j - 15 + l (Ht element 16 j + 1 + l)
    int from[4];
#pragma unroll
    for (int g = 0; g < 4; g++) from[g] = (15 + n - kq - 4 * g) * 4;       // whose tap this lane multiplies with in k group g
    // scalar byte offsets of chunk j: its window starts 16 j blocks before the tile, its taps at 16 j
    auto sx = [&](int j) { return (unsigned) (x_elem0 - kMfmaSeg * j) * 8u; };
    auto sh = [&](int j) { return (unsigned) (kMfmaSeg * j) * 8u; };
    // operands of one chunk: k groups 0 .. 3 of the blocks (re, im, re + im) and of the taps; `taps` = what the lane loaded
    struct Set { v4f_t xr, xi; v2f_t xs[2], hr[2], hi[2], hs[2]; float2 taps; };
    Set A, B;
    auto load_taps = [&](Set &s, int j) { s.taps = buf_ld8<0>(rh, vh, sh(j)); };
    auto load_x = [&](Set &s, unsigned v, int j) {                         // (a chunk past the range reads blocks / taps that exist in the padded layouts: never used)
        const v4u_t re = __builtin_amdgcn_raw_buffer_load_b128(rx, v, sx(j), 0), im = __builtin_amdgcn_raw_buffer_load_b128(rx, v, sx(j) + 16, 0);
#pragma unroll
        for (int g = 0; g < 4; g++) { s.xr[g] = __uint_as_float(re[g]); s.xi[g] = __uint_as_float(im[g]); }
    };
    auto spread_taps = [&](Set &s) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const float2 t = from_lane(from[g], s.taps);
            s.hr[g / 2][g % 2] = t.x; s.hi[g / 2][g % 2] = t.y;
        }
    };
    auto sums = [&](Set &s) {                                              // four v_pk_add_f32
        if constexpr (!PACKED) {
            s.xs[0] = s.xr.lo + s.xi.lo; s.xs[1] = s.xr.hi + s.xi.hi;
            s.hs[0] = s.hr[0] + s.hi[0]; s.hs[1] = s.hr[1] + s.hi[1];
        }
    };
    auto multiply = [&](const Set &s) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
            acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(s.xr[g], s.hr[g / 2][g % 2], acc_r, 0, 0, 0);
            if constexpr (PACKED) {
                acc_i = __builtin_amdgcn_mfma_f32_16x16x4f32(s.xi[g], s.hi[g / 2][g % 2], acc_i, 0, 0, 0);
            } else {                                   // three real products per complex one: Xr Hr, Xi Hi, (Xr + Xi)(Hr + Hi); combined after the loop
                acc_n = __builtin_amdgcn_mfma_f32_16x16x4f32(s.xi[g], s.hi[g / 2][g % 2], acc_n, 0, 0, 0);
                acc_i = __builtin_amdgcn_mfma_f32_16x16x4f32(s.xs[g / 2][g % 2], s.hs[g / 2][g % 2], acc_i, 0, 0, 0);
            }
        }
    };
    // Three register sets rotate (the loop is unrolled by three, nothing is copied): chunk j is multiplied out of `cur`; before
    // that `nxt` (taps and new blocks loaded two chunks ago) starts to become chunk j + 1's operands -- the lane exchanges run
    // under the MFMAs; `far` holds chunk j + 2's loads in flight; `cur` is then reloaded for chunk j + 3
    auto step = [&](int j, Set &cur, Set &nxt) {
        __builtin_amdgcn_sched_barrier(0);
        spread_taps(nxt);
#pragma unroll
        for (int g = 0; g < 4; g++) { nxt.xr[g] = shift_in(nxt.xr[g], cur.xr[g]); nxt.xi[g] = shift_in(nxt.xi[g], cur.xi[g]); }
        __builtin_amdgcn_sched_barrier(0);
        multiply(cur);
        __builtin_amdgcn_sched_barrier(0);
        sums(nxt);
        if constexpr (!(IRB_MFMA_EXP & 2)) { load_x(cur, vx_new, j + 3); load_taps(cur, j + 3); }
    };
    Set C;
    load_x(A, vx, jb);                                                     // the whole window of the first chunk: every lane its own 32 bytes
    load_taps(A, jb);
    load_x(B, vx_new, jb + 1);
    load_taps(B, jb + 1);
    load_x(C, vx_new, jb + 2);
    load_taps(C, jb + 2);
    spread_taps(A);
    sums(A);
    IRB_STAMP(a, 3);
    int j = jb;
#pragma unroll 1
    for (; j + 3 <= je; j += 3) {
        step(j, A, B);
        step(j + 1, B, C);
        step(j + 2, C, A);
    }
    if (j < je) {                                                          // one or two chunks left
        step(j, A, B);
        if (j + 1 < je) { __builtin_amdgcn_sched_barrier(0); multiply(B); }
    }
}

// grid = S * tile_count * splits * (M / 16); block = 1024.  Partial spectra: part[s][t - t_base][split][M], part_T blocks per
// stream (a call may be multiplied in two launches -- whole rounds of workgroups, then the remaining tiles with more partition
// splits -- each with its own partial-spectrum region; see launch_tiled).
template <int N>
__global__ __launch_bounds__(kMfmaThreads) void k_cmac_mfma(StepArgs a)
{
    constexpr int M = N / 2;
    static_assert(M % kMfmaBins == 0, "bins per workgroup");
    constexpr int GROUPS = M / kMfmaBins;
    __shared__ float2 l_o[kMfmaTile * kMfmaOS];
    // one grid may carry both launches of a call's plan (whole rounds of workgroups, then the remaining tiles with their partitions
    // split): the second region's workgroups are the LAST in dispatch order, short, and start as the first region's CUs free up
    const bool second = a.mf_wgs0 > 0 && (int) blockIdx.x >= a.mf_wgs0;
    const int tile_first = second ? a.mf2_tile_first : a.tile_first, tile_count = second ? a.mf2_tile_count : a.tile_count,
              splits = second ? a.mf2_splits : a.splits, t_base = second ? a.mf2_t_base : a.t_base, part_T = second ? a.mf2_part_T : a.part_T;
    float2 *const part = second ? a.part + a.mf2_part_off : a.part;
    const int first_wgs = a.mf_wgs0 > 0 ? min(a.mf_wgs0, (int) gridDim.x) : (int) gridDim.x;
    // the bin groups and splits of one tile share an XCD's L2 (the same FDL lines)
    const int wg = second ? xcd_contiguous_id(blockIdx.x - a.mf_wgs0, gridDim.x - a.mf_wgs0) : xcd_contiguous_id(blockIdx.x, first_wgs);
    const int bg = wg % GROUPS, split = (wg / GROUPS) % splits, tile = tile_first + (wg / (GROUPS * splits)) % tile_count,
              s = wg / (GROUPS * splits * tile_count);
    const int2 si = a.stream_info[s];
    const int set = si.x, np = si.y;
    const int J = mfma_chunks(np);
    const int jb = (int) ((long) J * split / splits), je = (int) ((long) J * (split + 1) / splits);
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
    IRB_STAMP(a, 1);
    if (jb < je) {
        if (bin == 0) mfma_chunk_run<true>(rx, rh, lane, x_elem0, jb, je, acc_r, acc_n, acc_i, a);
        else          mfma_chunk_run<false>(rx, rh, lane, x_elem0, jb, je, acc_r, acc_n, acc_i, a);
    }
    IRB_STAMP(a, 2);
    // ---- store: accumulator register v of lane (i, kq) is O[n = 4 kq + v][i], i.e. block 256 tile + 16 n + i of the call; through
    // LDS, so that each output row leaves as one whole 128-byte line
    const int i = lane % kMfmaSeg, kq = lane / kMfmaSeg;
#pragma unroll
    for (int v = 0; v < 4; v++) l_o[(kMfmaSeg * (4 * kq + v) + i) * kMfmaOS + wave] = make_float2(acc_r[v] - acc_n[v], bin == 0 ? acc_i[v] : acc_i[v] - acc_r[v] - acc_n[v]);
    __syncthreads();
    IRB_STAMP(a, 4);
    const int sbin = tid % kMfmaBins, srow = tid / kMfmaBins;
#pragma unroll
    for (int it = 0; it < kMfmaTile / (kMfmaThreads / kMfmaBins); it++) {
        const int tl = srow + (kMfmaThreads / kMfmaBins) * it, t = kMfmaTile * tile + tl;
        // non-temporal: the rows are read once, by the next kernel -- written through now instead of by the end-of-kernel write-back (-0.8 us)
        if (t < a.T) {
            const float2 val = l_o[tl * kMfmaOS + sbin];
            v2f_t q = {val.x, val.y};
            __builtin_nontemporal_store(q, reinterpret_cast<v2f_t *>(part + (((size_t) s * part_T + (t - t_base)) * splits + split) * M + b0 + sbin));
        }
    }
    IRB_STAMP(a, 7);
}

}  // namespace irb
