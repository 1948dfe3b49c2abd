// fdl_kernels.hpp -- gfx950 (MI355X, CDNA4) device code of the partitioned-convolution engine.
//
// Data layout in HBM (all fp32):
//   H   [ir_set][P][M]   float2   IR partition spectra, M = N/2 packed bins per partition:
//                                 bin 0 = (DC.re, Nyquist.re), bins 1..M-1 ordinary complex
//                                 (packing precedent: reference fp/ir.cpp:123-124).  A row is N*4 bytes.
//   X   [stream][R][M]   float2   frequency-domain delay line (FDL) ring, same packing; slot of block t
//                                 is t mod R; partition k pairs with slot (head-k) mod R
//                                 (reference fp/convolution.cpp:129,166-167,198-201).
//   ovl [stream][B]      float    overlap saved from the previous block (fp/convolution.cpp:209-213).
//   tw  [N]              float2   W_N^j = exp(-2*pi*i*j/N), computed in double on the host.
//
// One wave = 64 lanes; workgroups are 256 threads (4 waves).  In the multiply-accumulate a lane
// owns two adjacent complex bins (one 16-byte load per spectrum row), a row is covered by
// LW = min(M/2, 256) lanes, and the 256/LW "stripes" of a workgroup walk interleaved partitions;
// stripes are combined through LDS.  No MFMA: 8 flop per 16 streamed bytes, HBM-bound.
#pragma once
#include <hip/hip_runtime.h>

#include "fft_wave.hpp"


namespace irb {

constexpr int kThreads = 256;

// transforms per workgroup in the FFT-only kernels (one wave each); LDS = NF*M*8 + N*4 bytes
template <int N> struct FftGeom { static constexpr int NF = N <= 2048 ? 4 : 2; };

// Diagnostic builds only (tools/step_stamps.hip compiles this header with -DIRB_STAMPS; libirbfdl.so never does):
// s_memtime stamps of the step kernels' phases, written to a buffer of their own by thread 0 of every workgroup.
#ifdef IRB_STAMPS
#define IRB_STAMP(a, i) irb::stamp_((a).stamps, (i))
#else
#define IRB_STAMP(a, i) do { } while (0)
#endif

struct StepArgs {
#ifdef IRB_STAMPS
    unsigned long long *stamps; // [grid][16]
#endif
    const float2 *H;            // [n_sets][P][M]  (P here = allocated rows per set, a multiple of 8)
    float2       *X;            // [S][R][M]
    float        *ovl;          // [S][B]
    const float2 *tw;           // [N]
    const int2   *stream_info;  // [S]: x = IR set the stream multiplies with, y = partitions in use (<= P) of that set;
                                // ONE 8-byte record so a workgroup learns both with a single (scalar) load before its first row
    const float  *in;           // block j of stream s at in + s*in_stride + j*B
    float        *out;
    float2       *part;         // [S][splits][M] partial sums (split path)
    unsigned long long *gran;   // [S][splits][2M] tagged granules {epoch : value} of the granule hand-off (k_step_cols<.., 2>)
    unsigned     *status;       // host-visible word: set to 1 if a granule wait gave up (bounded spin)
    unsigned      epoch;        // tag of this launch's granules: step counter + 1, never 0
    long          in_stride, out_stride;
    int           B, P, R, head, splits;
    // time-tiled path (several blocks per call)
    float        *ovl_next;     // overlap written by the last block of the call (double buffered)
    int           T, nq;        // blocks in this call; partial rows per output (splits * stripes)
    // real-time callback in one launch (k_rt_pair): rolling IR-partition refresh + completion word
    const float  *rf_src;       // time-domain IR [P*B] (host-mapped): block j re-derives partition (rf_part + j) mod P from it
    int           rf_part;
    unsigned     *done;         // host-mapped word that receives done_seq after the last output store (nullptr: none)
    unsigned      done_seq;
    int           debug_drop;   // test hook (IRB_DEBUG_DROP_GRANULES): the last workgroup of the grid withholds its granules in this launch
    // matrix-core path of long calls (fdl_mfma.hpp): the FDL and the IR spectra once more, time / tap contiguous per bin
    float2       *Xt;           // [S][M][xt_len]: element xt_hist + t = block t of the call, [0, xt_hist) = the blocks before it
    const float2 *Ht;           // [n_sets][M][ht_len]: element 16 + m = tap m (zeros in front and behind)
    int           xt_len, xt_hist, ht_len;
    int           hist_first;   // >= 0: k_push_fft_batch_t also rebuilds Xt's history columns from the ring, starting at this ring row
    int           tile_first, tile_count;   // the 256-block tiles this launch multiplies (a call may take two launches) ...
    int           t_base, part_T;           // ... and where its partial spectra go: part[s][t - t_base][nq][M], part_T blocks per stream
    int           mf_wgs0;                  // > 0: ONE k_cmac_mfma grid carries both launches of a call's plan -- workgroups [0, mf_wgs0) are the
                                            // first (tile_first, tile_count, splits, t_base, part_T, part), the rest the second:
    int           mf2_tile_first, mf2_tile_count, mf2_splits, mf2_t_base, mf2_part_T;
    long          mf2_part_off;             // its partial-spectrum region starts at part + mf2_part_off
    // batched FFT kernels: the blocks [t_first, t_first + t_count) of the call (irb_convolve_periodic moves a call's blocks in and out in chunks)
    int           t_first, t_count;
    // k_ifft_ola_run: where the partial spectra of a block are.  A matrix-core call's multiply may run as two launches (whole rounds of
    // workgroups, then the remaining tiles with more partition splits), each with its own region of `part`: region r holds blocks
    // [reg_t0[r], reg_t0[r + 1]) as part + reg_off[r] + ((s * blocks of the region + block - reg_t0[r]) * reg_nq[r]) * M
    int           reg_t0[3];    // (one region: {0, T, T})
    int           reg_nq[2];
    long          reg_off[2];   // in float2
    // ... and the samples that exist behind `in` / may be written behind `out`, per stream (irb_convolve_periodic lets these kernels read
    // the caller's audio and write the caller's result buffer in place: blocks past the end read as zeros / are cut off)
    long          in_len, out_len;
    // k_ifft_ola_run for the staged one-shot call: ONE launch writes all blocks of a piece into pinned host memory and tells the host
    // chunk by chunk -- a wave that has written block j counts it in ck_cnt[j / ck_blocks] (device memory), the wave that completes a
    // chunk stores ck_seq into the chunk's word in mapped host memory (ck_flag + 16 * chunk: one cache line each)
    unsigned     *ck_cnt, *ck_flag;
    int           ck_blocks;
    unsigned      ck_seq;
    // k_push_fft_batch_t for the staged one-shot call: ONE launch transforms all blocks of a piece out of pinned host memory that the
    // host is still filling -- a workgroup waits (bounded) until the word of its item (pk_blocks blocks, a multiple of the workgroup's 16)
    // in mapped host memory (pk_flag + 16 * item) holds pk_seq, which the packing thread stores behind the item's samples
    const unsigned *pk_flag;
    int           pk_blocks;
    unsigned      pk_seq;
    // diagnostic (IRB_DEBUG=input_checksums=1, tools/fuzz_paths.py): the batched forward-FFT kernels leave the sum of the bit patterns of
    // every block's samples AS THEY READ THEM at dbg_sums[s * T + j] -- what a wrong result is compared with first (did the device see
    // the bytes the host sees?)
    unsigned     *dbg_sums;
};

#ifdef IRB_STAMPS
__device__ __forceinline__ void stamp_(unsigned long long *buf, int i)
{
#ifdef IRB_STAMPS_ENDS_ONLY
    if (i != 0 && i != 10) return;                   // (tools/clock_under_load.hip: a kernel's first and last stamp only -- the phase stamps' fences stay out of it)
#endif
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (threadIdx.x == 0) buf[(size_t) blockIdx.x * 16 + i] = t;
    if (i == 0 || i == 7 || i == 10) {               // the 100 MHz reference clock too: comparable ACROSS workgroups / XCDs
        unsigned long long r;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) :: "memory");
        if (threadIdx.x == 0) buf[(size_t) blockIdx.x * 16 + (i == 0 ? 11 : i == 7 ? 12 : 13)] = r;
    }
}
#endif

// diagnostic (StepArgs::dbg_sums): the sum of the bit patterns of the zero-padded block a wave has just put into z (every lane adds the
// floats it stored itself)
template <int N>
__device__ __forceinline__ void wave_block_checksum(const float2 *z, int lane, unsigned *dst)
{
    const unsigned *zu = reinterpret_cast<const unsigned *>(z);
    unsigned acc = 0;
    for (int n = lane; n < N; n += kWave) acc += zu[n];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, kWave);
    if (lane == 0) *dst = acc;
}

// Overlap-add (reference fp/convolution.cpp:209-213) by one wave: y = N real samples, unscaled, in LDS.
__device__ __forceinline__ void wave_overlap_add(const float *y, float scale, int B, float *__restrict__ ovl,
                                                 float *__restrict__ out, int lane)
{
    for (int i = lane; i < B; i += kWave) {
        const float o = y[i] * scale + ovl[i];
        ovl[i] = y[B + i] * scale;
        out[i] = o;
    }
}

// One wave's share of a block step's audio I/O, kept in registers across the multiply-accumulate: XL samples per
// lane of the new block (sample lane + 64 i) and of the overlap saved by the previous step.  Loads are clamped,
// not predicated (a predicated load makes hipcc wait for it at the join); the padding is applied on use.
template <int N>
struct BlockIo {
    static constexpr int M = N / 2;
    static constexpr int XL = M >= kWave ? M / kWave : 1;        // B <= M
    float x[XL], ov[XL];

    __device__ __forceinline__ void load(const float *__restrict__ xin, const float *__restrict__ ovl, int B, int lane)
    {
#pragma unroll
        for (int i = 0; i < XL; i++) {
            const int n = min(lane + i * kWave, B - 1);
            x[i] = xin[n];
            ov[i] = ovl[n];
        }
    }
    // only the samples (a later block of the same launch: its overlap is already in `ov`, see overlap_add_keep)
    __device__ __forceinline__ void load_x(const float *__restrict__ xin, int B, int lane)
    {
#pragma unroll
        for (int i = 0; i < XL; i++) x[i] = xin[min(lane + i * kWave, B - 1)];
    }
    // the block, zero padded to N reals, as the M-point complex input z[n] = x[2n] + i x[2n+1] (same memory)
    __device__ __forceinline__ void block_to_lds(float2 *buf, int B, int lane) const
    {
        float *zf = reinterpret_cast<float *>(buf);
#pragma unroll
        for (int i = 0; i < N / kWave; i++) {
            const int n = lane + i * kWave;
            if (i < XL) zf[n] = n < B ? x[i] : 0.0f;
            else zf[n] = 0.0f;
        }
        if constexpr (N < kWave) { if (lane < N) zf[lane] = lane < B ? x[0] : 0.0f; }
    }
    // overlap-add (reference fp/convolution.cpp:209-213): y = N real samples, unscaled, in LDS
    __device__ __forceinline__ void overlap_add(const float2 *buf, int B, float *__restrict__ ovl, float *__restrict__ out, int lane) const
    {
        const float *y = reinterpret_cast<const float *>(buf);
        const float scale = 1.0f / (float) N;
#pragma unroll
        for (int i = 0; i < XL; i++) {
            const int n = lane + i * kWave;
            if (n < B) {
                out[n] = y[n] * scale + ov[i];
                ovl[n] = y[B + n] * scale;
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------
// Multiply-accumulate over partitions [kb, ke) of one stream (reference fp/convolution.cpp:171-202):
// acc += X[(head-k) mod R] * H[k].  Each lane: 16-byte loads of two complex bins from both rows.
// dc accumulates the two real products of packed bin 0 (used by the lane that owns column 0).
// ---------------------------------------------------------------------------------------------
// the same, and the new overlap also stays in `ov` for the next block of this launch
template <int N>
__device__ __forceinline__ void overlap_add_keep(BlockIo<N> &io, const float2 *buf, int B, float *__restrict__ ovl, float *__restrict__ out, int lane)
{
    const float *y = reinterpret_cast<const float *>(buf);
    const float scale = 1.0f / (float) N;
#pragma unroll
    for (int i = 0; i < BlockIo<N>::XL; i++) {
        const int n = min(lane + i * kWave, B - 1);
        const float o = y[n] * scale + io.ov[i], t = y[B + n] * scale;
        io.ov[i] = t;
        if (lane + i * kWave < B) { out[n] = o; ovl[n] = t; }
    }
}

template <int N>
struct MacGeom {
    static constexpr int M = N / 2;
    static constexpr int L = M / 2;                               // float4 per row
    static constexpr int LW = L < kThreads ? L : kThreads;        // lanes across a row
    static constexpr int CPT = L / LW;                            // float4 columns per thread
    static constexpr int STRIPES = kThreads / LW;                 // partitions in flight per pass
};

typedef float v4f_t __attribute__((ext_vector_type(4)));
// 16-byte global load; NT = non-temporal hint for rows that are streamed once per step
template <bool NT>
__device__ __forceinline__ float4 ld16(const float4 *p)
{
    if (NT) {
        const v4f_t v = __builtin_nontemporal_load(reinterpret_cast<const v4f_t *>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *p;
}

template <int N, int U, bool NT = false>
__device__ __forceinline__ void mac_range(const float4 *__restrict__ H4, const float4 *__restrict__ X4,
                                          int kb, int ke, int head, int R,
                                          float4 (&acc)[MacGeom<N>::CPT], float2 &dc)
{
    using G = MacGeom<N>;
    const int col = threadIdx.x % G::LW;
    const int stripe = threadIdx.x / G::LW;
    int k = kb + stripe;
    // full groups of U partitions per stripe: all loads issued before the first use
    for (; k + (U - 1) * G::STRIPES < ke; k += U * G::STRIPES) {
        float4 h[U][G::CPT], x[U][G::CPT];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int kk = k + u * G::STRIPES;
            int slot = head - kk;
            if (slot < 0) slot += R;
#pragma unroll
            for (int c = 0; c < G::CPT; c++) {
                h[u][c] = ld16<NT>(H4 + (size_t) kk * G::L + col + c * G::LW);
                x[u][c] = ld16<NT>(X4 + (size_t) slot * G::L + col + c * G::LW);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            dc.x += x[u][0].x * h[u][0].x;
            dc.y += x[u][0].y * h[u][0].y;
#pragma unroll
            for (int c = 0; c < G::CPT; c++) {
                acc[c].x += x[u][c].x * h[u][c].x - x[u][c].y * h[u][c].y;
                acc[c].y += x[u][c].x * h[u][c].y + x[u][c].y * h[u][c].x;
                acc[c].z += x[u][c].z * h[u][c].z - x[u][c].w * h[u][c].w;
                acc[c].w += x[u][c].z * h[u][c].w + x[u][c].w * h[u][c].z;
            }
        }
    }
    for (; k < ke; k += G::STRIPES) {
        int slot = head - k;
        if (slot < 0) slot += R;
#pragma unroll
        for (int c = 0; c < G::CPT; c++) {
            const float4 h = ld16<NT>(H4 + (size_t) k * G::L + col + c * G::LW);
            const float4 x = ld16<NT>(X4 + (size_t) slot * G::L + col + c * G::LW);
            if (c == 0) { dc.x += x.x * h.x; dc.y += x.y * h.y; }
            acc[c].x += x.x * h.x - x.y * h.y;
            acc[c].y += x.x * h.y + x.y * h.x;
            acc[c].z += x.z * h.z - x.w * h.w;
            acc[c].w += x.z * h.w + x.w * h.z;
        }
    }
}

__device__ __forceinline__ void lds_barrier();     // defined below: workgroup barrier that waits for LDS traffic only

// Combine the stripes of a workgroup: part is LDS [STRIPES][M]; result Y[M] (packed) in `Y`.
// LDS_ONLY: the two barriers do not drain the wave's global loads (k_rt_pair keeps the next block's samples in flight).
template <int N, bool LDS_ONLY = false>
__device__ __forceinline__ void reduce_stripes(const float4 (&acc)[MacGeom<N>::CPT], float2 dc, float2 *part, float2 *Y)
{
    using G = MacGeom<N>;
    const int col = threadIdx.x % G::LW;
    const int stripe = threadIdx.x / G::LW;
#pragma unroll
    for (int c = 0; c < G::CPT; c++) {
        const int b = 2 * (col + c * G::LW);
        float2 lo = make_float2(acc[c].x, acc[c].y);
        if (b == 0) lo = dc;                       // packed bin 0: two real products, not a complex one
        part[stripe * G::M + b] = lo;
        part[stripe * G::M + b + 1] = make_float2(acc[c].z, acc[c].w);
    }
    if (LDS_ONLY) lds_barrier(); else __syncthreads();
    for (int b = threadIdx.x; b < G::M; b += kThreads) {
        float2 s = part[b];
#pragma unroll
        for (int t = 1; t < G::STRIPES; t++) s = cadd(s, part[t * G::M + b]);
        Y[b] = s;
    }
    if (LDS_ONLY) lds_barrier(); else __syncthreads();
}

template <int N>
struct Lds {
    static constexpr int M = N / 2;
    float2 tw[TwLds<N>::SIZE];                     // twiddle table W_N^j, j < 3N/4
    float2 buf[M];                                 // wave 0's FFT buffer; later the summed spectrum Y
    float2 scratch[MacGeom<N>::STRIPES * M];       // first the new block's spectrum, then the stripe partials
};

// ---------------------------------------------------------------------------------------------
// k_step_fused: one workgroup owns one stream for one block step:
//   input block -> FFT -> ring slot `head`; MAC over all partitions; IFFT; overlap-add; output.
// grid = S.  The dominant kernel when many streams are batched (HBM-bound, 8*P*N + 16*B bytes).
// Wave 0 runs the two small FFTs (fft_wave.hpp) while the other waves already stream partitions.
// ---------------------------------------------------------------------------------------------
template <int N, int U = 2, bool NT = false>
__global__ __launch_bounds__(kThreads) void k_step_fused(StepArgs a)
{
    using G = MacGeom<N>;
    constexpr int M = N / 2;
    __shared__ Lds<N> lds;
    const int s = blockIdx.x;
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const int2 si = a.stream_info[s];
    const int set = si.x, np = si.y;
    const float4 *H4 = reinterpret_cast<const float4 *>(a.H + (size_t) set * a.P * M);
    float2 *Xs = a.X + (size_t) s * a.R * M;
    const float4 *X4 = reinterpret_cast<const float4 *>(Xs);

    IRB_STAMP(a, 0);
    // issue the small loads now (twiddle half table, and wave 0: the new block) so that their latency
    // hides under the partition stream; they are parked in registers until the MAC loop is done
    constexpr int TWPT = (N / 2 + kThreads - 1) / kThreads;
    float2 twr[TWPT];
#pragma unroll
    for (int i = 0; i < TWPT; i++) twr[i] = a.tw[min((int) threadIdx.x + i * kThreads, N / 2 - 1)];
    BlockIo<N> io;
    if (wave == 0) io.load(a.in + (size_t) s * a.in_stride, a.ovl + (size_t) s * a.B, a.B, lane);

    // older partitions: their rows do not depend on the new block
    float4 acc[G::CPT];
#pragma unroll
    for (int c = 0; c < G::CPT; c++) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 dc = make_float2(0.f, 0.f);
    mac_range<N, U, NT>(H4, X4, 1, np, a.head, a.R, acc, dc);
    IRB_STAMP(a, 1);

#pragma unroll
    for (int i = 0; i < TWPT; i++) {
        const int idx = threadIdx.x + i * kThreads;
        if (idx < N / 2) put_twiddle<N>(lds.tw, idx, twr[i]);
    }
    if (wave == 0) io.block_to_lds(lds.buf, a.B, lane);
    __syncthreads();
    IRB_STAMP(a, 2);
    if (wave == 0) {      // new block: FFT, packed spectrum to LDS (for partition 0) and into the ring
        wave_fft<M, false>(lds.buf, lds.tw, lane);
        IRB_STAMP(a, 3);
        float2 *ring = Xs + (size_t) a.head * M;
        wave_split_forward<M>(lds.buf, lds.tw, lane, [&](int b, float2 v) { lds.scratch[b] = v; ring[b] = v; });
    }
    __syncthreads();
    IRB_STAMP(a, 4);

    if (np > 0 && threadIdx.x < G::LW) {           // partition 0 with the new block's spectrum
        const int col = threadIdx.x;
#pragma unroll
        for (int c = 0; c < G::CPT; c++) {
            const float4 h = H4[col + c * G::LW];
            const float4 x = reinterpret_cast<const float4 *>(lds.scratch)[col + c * G::LW];
            if (c == 0) { dc.x += x.x * h.x; dc.y += x.y * h.y; }
            acc[c].x += x.x * h.x - x.y * h.y;
            acc[c].y += x.x * h.y + x.y * h.x;
            acc[c].z += x.z * h.z - x.w * h.w;
            acc[c].w += x.z * h.w + x.w * h.z;
        }
    }
    __syncthreads();
    IRB_STAMP(a, 5);

    reduce_stripes<N>(acc, dc, lds.scratch, lds.buf);
    IRB_STAMP(a, 6);
    if (wave == 0) {
        wave_split_inverse<M>(lds.buf, lds.tw, lane);
        IRB_STAMP(a, 7);
        wave_fft<M, true>(lds.buf, lds.tw, lane);
        IRB_STAMP(a, 8);
        io.overlap_add(lds.buf, a.B, a.ovl + (size_t) s * a.B, a.out + (size_t) s * a.out_stride, lane);
        IRB_STAMP(a, 9);
#ifdef IRB_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        IRB_STAMP(a, 10);
#endif
    }
}

// ---------------------------------------------------------------------------------------------
// k_step_fused_pair: the fused step for TWO streams that multiply with the SAME IR spectra -- stereo audio
// with a mono IR (IRMonoAudioStereo, reference fp/convolution.cpp:32-33,176-181) and the real-time path,
// which feeds IR channel 0 to both channels (Source/PluginProcessor.cpp:485).  One workgroup per instance:
// every H row is loaded once and used for both channels, so the step streams 3 rows per partition instead
// of 4 (-25 % HBM bytes).  Waves 0 and 1 run the two channels' FFTs side by side.  grid = S/2.
// ---------------------------------------------------------------------------------------------
template <int N>
struct LdsPair {
    static constexpr int M = N / 2;
    float2 tw[TwLds<N>::SIZE];
    float2 buf[2][M];
    float2 scratch[2][MacGeom<N>::STRIPES * M];
};

__device__ __forceinline__ void cmac_pair(float4 &acc, float2 &dc, const float4 x, const float4 h, bool first)
{
    if (first) { dc.x += x.x * h.x; dc.y += x.y * h.y; }
    acc.x += x.x * h.x - x.y * h.y;
    acc.y += x.x * h.y + x.y * h.x;
    acc.z += x.z * h.z - x.w * h.w;
    acc.w += x.z * h.w + x.w * h.z;
}

template <int N, bool NT>
__global__ __launch_bounds__(kThreads) void k_step_fused_pair(StepArgs a)
{
    using G = MacGeom<N>;
    constexpr int M = N / 2, U = 2;
    __shared__ LdsPair<N> lds;
    const int s0 = 2 * blockIdx.x;
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const int2 si = a.stream_info[s0];
    const int set = si.x, np = si.y;
    const float4 *H4 = reinterpret_cast<const float4 *>(a.H + (size_t) set * a.P * M);
    float2 *Xs[2] = {a.X + (size_t) s0 * a.R * M, a.X + (size_t) (s0 + 1) * a.R * M};
    const float4 *X4[2] = {reinterpret_cast<const float4 *>(Xs[0]), reinterpret_cast<const float4 *>(Xs[1])};
    const int col = threadIdx.x % G::LW, stripe = threadIdx.x / G::LW;

    constexpr int TWPT = (N / 2 + kThreads - 1) / kThreads;
    float2 twr[TWPT];
#pragma unroll
    for (int i = 0; i < TWPT; i++) twr[i] = a.tw[min((int) threadIdx.x + i * kThreads, N / 2 - 1)];
    BlockIo<N> io;
    if (wave < 2) io.load(a.in + (size_t) (s0 + wave) * a.in_stride, a.ovl + (size_t) (s0 + wave) * a.B, a.B, lane);

    float4 acc[2][G::CPT];
    float2 dc[2] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
#pragma unroll
    for (int c = 0; c < G::CPT; c++) { acc[0][c] = make_float4(0.f, 0.f, 0.f, 0.f); acc[1][c] = acc[0][c]; }
    int k = 1 + stripe;
    for (; k + (U - 1) * G::STRIPES < np; k += U * G::STRIPES) {
        float4 h[U][G::CPT], x0[U][G::CPT], x1[U][G::CPT];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int kk = k + u * G::STRIPES;
            int slot = a.head - kk;
            if (slot < 0) slot += a.R;
#pragma unroll
            for (int c = 0; c < G::CPT; c++) {
                h[u][c] = ld16<NT>(H4 + (size_t) kk * G::L + col + c * G::LW);
                x0[u][c] = ld16<NT>(X4[0] + (size_t) slot * G::L + col + c * G::LW);
                x1[u][c] = ld16<NT>(X4[1] + (size_t) slot * G::L + col + c * G::LW);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int c = 0; c < G::CPT; c++) {
                cmac_pair(acc[0][c], dc[0], x0[u][c], h[u][c], c == 0);
                cmac_pair(acc[1][c], dc[1], x1[u][c], h[u][c], c == 0);
            }
    }
    for (; k < np; k += G::STRIPES) {
        int slot = a.head - k;
        if (slot < 0) slot += a.R;
#pragma unroll
        for (int c = 0; c < G::CPT; c++) {
            const float4 h = ld16<NT>(H4 + (size_t) k * G::L + col + c * G::LW);
            cmac_pair(acc[0][c], dc[0], ld16<NT>(X4[0] + (size_t) slot * G::L + col + c * G::LW), h, c == 0);
            cmac_pair(acc[1][c], dc[1], ld16<NT>(X4[1] + (size_t) slot * G::L + col + c * G::LW), h, c == 0);
        }
    }

#pragma unroll
    for (int i = 0; i < TWPT; i++) {
        const int idx = threadIdx.x + i * kThreads;
        if (idx < N / 2) put_twiddle<N>(lds.tw, idx, twr[i]);
    }
    if (wave < 2) io.block_to_lds(lds.buf[wave], a.B, lane);
    __syncthreads();
    if (wave < 2) {       // new blocks: FFT, spectrum to LDS (for partition 0) and into the ring
        wave_fft<M, false>(lds.buf[wave], lds.tw, lane);
        float2 *ring = Xs[wave] + (size_t) a.head * M;
        float2 *sc = lds.scratch[wave];
        wave_split_forward<M>(lds.buf[wave], lds.tw, lane, [&](int b, float2 v) { sc[b] = v; ring[b] = v; });
    }
    __syncthreads();
    if (np > 0 && threadIdx.x < G::LW) {
#pragma unroll
        for (int c = 0; c < G::CPT; c++) {
            const float4 h = H4[col + c * G::LW];
            cmac_pair(acc[0][c], dc[0], reinterpret_cast<const float4 *>(lds.scratch[0])[col + c * G::LW], h, c == 0);
            cmac_pair(acc[1][c], dc[1], reinterpret_cast<const float4 *>(lds.scratch[1])[col + c * G::LW], h, c == 0);
        }
    }
    __syncthreads();
    reduce_stripes<N>(acc[0], dc[0], lds.scratch[0], lds.buf[0]);
    reduce_stripes<N>(acc[1], dc[1], lds.scratch[1], lds.buf[1]);
    if (wave < 2) {
        wave_split_inverse<M>(lds.buf[wave], lds.tw, lane);
        wave_fft<M, true>(lds.buf[wave], lds.tw, lane);
        io.overlap_add(lds.buf[wave], a.B, a.ovl + (size_t) (s0 + wave) * a.B, a.out + (size_t) (s0 + wave) * a.out_stride, lane);
    }
}

// ---------------------------------------------------------------------------------------------
// k_step_cols: the one-block step for FEW streams (latency path) and for mid-size batches, in ONE launch.
// A workgroup owns a COLUMN slice of the spectrum (W float4 = 2W complex bins) and a partition range:
//   grid = S * CS * PS,  CS = (N/4)/W column slices,  PS = partition splits (a.splits).
// Thread (col, stripe) = (tid % W, tid / W) loads its column of up to 8 partitions (H row and FDL row, 16 B each)
// in ONE round of loads; the 256/W stripes are combined by DPP row rotations + one LDS pass.  Slicing by columns
// instead of by partitions shrinks the inter-workgroup hand-off from one whole spectrum row per workgroup
// (the partition-split kernels of rounds 1-4: splits * N*4 bytes per stream, read serially by the last arriver) to PS rows per stream.
// The new block's forward FFT is done REDUNDANTLY by every workgroup of partition range 0 (the CUs are idle
// anyway on this path) while its partition loads are in flight, so nothing waits for another workgroup's FFT;
// each of them writes only its own column slice of the spectrum into the ring.
// Hand-off (cdna_hip_programming.md Guideline 16 / MI355X_MICROARCH.md "Valid forms", first table row): every
// byte of the Y slices is stored with an sc1 (write-through) 8-byte store, the storing wave drains vmcnt,
// workgroup barrier, ONE lane adds to the stream's unsharded agent-scope ticket counter, the workgroup whose
// add came last loads every slice with sc1 8-byte loads after a barrier that wave joins.  No release or
// acquire fence (their ~1.7 us each is what this path is made of), no spinning, no co-residency assumption.
// ---------------------------------------------------------------------------------------------
// XCD-aware workgroup id: hardware deals consecutive workgroups round-robin over the 8 XCDs (each with
// its own 4 MiB L2); this bijection gives every XCD one contiguous range of ids instead, so workgroups
// that share X/H rows (neighbouring time tiles) also share an L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_contiguous_id(int bid, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

template <int CTRL>
__device__ __forceinline__ float dpp_row(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// LDS-only workgroup barrier: waits for this wave's LDS traffic, NOT for global loads in flight (a plain
// __syncthreads() drains vmcnt too and would serialise the FFT behind the partition loads)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

typedef unsigned long long u64_t;
__device__ __forceinline__ void st_sc1(float2 *p, float2 v)
{
    u64_t bits;
    __builtin_memcpy(&bits, &v, 8);
    __hip_atomic_store(reinterpret_cast<u64_t *>(p), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float2 ld_sc1(const float2 *p)
{
    const u64_t bits = __hip_atomic_load(reinterpret_cast<const u64_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float2 v;
    __builtin_memcpy(&v, &bits, 8);
    return v;
}

// lane-wise sum over the four 16-lane rows of a wave (v_permlane16_swap / v_permlane32_swap, gfx950): every lane
// ends up with the sum of the lanes that share its position in the row
__device__ __forceinline__ float rows_sum(float v)
{
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

template <int N, int W_>
struct ColsGeom {
    static constexpr int M = N / 2, L = M / 2;
    static constexpr int W = W_ < L ? W_ : L;                   // float4 columns per slice
    static constexpr int CS = L / W;                            // column slices per row
    static constexpr int TWL = N / 2 >= kWave ? N / 2 / kWave : 1;   // host twiddle entries per lane of ONE wave
};

// Roles inside a workgroup of k_step_cols:
//   * wave 0 of a workgroup of partition range 0 ("FFT wave"): small loads only (twiddles, the new block, H row 0,
//     the saved overlap), forward FFT + split of the new block, partition 0 of its columns, the slice of the new
//     spectrum into the ring.  It issues no partition loads, so its FFT starts as soon as its 2-3 KiB have landed
//     and runs while the other waves' loads are in flight (profiles/r02_step_stamps.txt: with every wave issuing
//     partition loads first, the texture addresser kept the FFT waiting ~0.8 us).
//   * every other wave ("loader"): its stripe of the partition range, U rows of H and of the FDL per thread per
//     round, unconditional 16-byte loads; wave 0 of the other workgroups also stages the twiddles and prefetches
//     the overlap, because ANY workgroup can turn out to be the last arriver that runs the inverse FFT.
// Hand-off forms of k_step_cols (HO):
//   0  write-through (sc1) slices + drained wait + ONE ticket per workgroup; the last arriver finishes the stream
//      (form 1 of rounds 2-4 -- plain slices + agent-scope release / acquire fences around the ticket, the textbook form, ~3.4 us
//      slower per step -- was the fallback for devices other than gfx950; the library is built for gfx950 only and no longer carries it)
//   2  tagged granules (cdna_hip_programming.md Guideline 16, R2: "the data is the flag"): every float of a slice is
//      ONE aligned 8-byte sc1 store {tag = epoch, value}; a producer exits right after issuing them -- no drain, no
//      barrier, no ticket.  The workgroup (slice 0, range 0) of each stream is the reducer: it re-reads the stream's
//      granules with sc1 loads until every tag equals this launch's epoch (bounded spin: on a 50 ms timeout it raises
//      a.status and carries on, so a lost producer cannot hang the grid).  Two memory round trips (the producers'
//      store, the reducer's sweep) instead of four (store drain, ticket, slice loads + their barriers).  All
//      workgroups of the grid must be able to run while the reducer spins: the host uses it for grids <= 256 only.
template <int N, int W_, int HO, int U = 8>
__global__ __launch_bounds__(kThreads) void k_step_cols(StepArgs a, unsigned *__restrict__ tickets)
{
    static_assert(HO == 0 || HO == 2, "hand-off forms: 0 write-through slices + ticket, 2 tagged granules");
    using C = ColsGeom<N, W_>;
    constexpr int M = C::M, W = C::W, CS = C::CS, TWL = C::TWL, NWAVES = kThreads / kWave;
    __shared__ float2 l_tw[TwLds<N>::SIZE];
    __shared__ float2 l_buf[M];                        // FFT work buffer; later the summed spectrum
    __shared__ float2 l_spec[M];                       // packed spectrum of the new block
    __shared__ float4 l_red[NWAVES][W];
    __shared__ float2 l_dc[NWAVES];
    __shared__ unsigned l_ticket;
    const int PS = a.splits;
    const int wg = xcd_contiguous_id(blockIdx.x, gridDim.x);       // neighbouring column slices share an XCD's L2
    const int cs = wg % CS, ps = (wg / CS) % PS, s = wg / (CS * PS);
    const int tid = threadIdx.x, wave = tid / kWave, lane = tid % kWave;
    const int2 si = a.stream_info[s];
    const int set = si.x, np = si.y;
    // partition 0 pairs with the block transformed in this launch and belongs to range 0; the older partitions
    // [1, np) are divided in proportion to the loader waves: 3 in a workgroup of range 0, 4 in the others
    const int nold = np > 0 ? np - 1 : 0;
    const int wtot = 4 * PS - 1, w0 = ps == 0 ? 0 : 4 * ps - 1, w1 = 4 * ps + 3;
    const int kb = 1 + (int) ((long) nold * w0 / wtot), ke = 1 + (int) ((long) nold * w1 / wtot);
    const float4 *H4 = reinterpret_cast<const float4 *>(a.H + (size_t) set * a.P * M);
    float2 *Xs = a.X + (size_t) s * a.R * M;
    const float4 *X4 = reinterpret_cast<const float4 *>(Xs);
    const bool first = (ps == 0);                      // this workgroup also owns partition 0 of its columns
    const bool fft_wave = first && wave == 0;
    IRB_STAMP(a, 0);

    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 dc = make_float2(0.f, 0.f);
    BlockIo<N> io;                                     // wave 0: new block + saved overlap (prefetched: any workgroup may run the tail)
    float2 twr[TWL];
    const bool may_finish = HO != 2 || (cs == 0 && ps == 0);   // granule form: the reducer is known in advance
    if (wave == 0 && (first || may_finish)) {          // loads are clamped, never lane-predicated (see BlockIo)
#pragma unroll
        for (int i = 0; i < TWL; i++) twr[i] = a.tw[min(lane + i * kWave, N / 2 - 1)];
        io.load(a.in + (size_t) s * a.in_stride, a.ovl + (size_t) s * a.B, a.B, lane);
    }

    if (fft_wave) {
        const int gcol = cs * W + (lane < W ? lane : 0);
        const float4 h0 = H4[gcol];                    // row 0 always exists (used only when np > 0)
        IRB_STAMP(a, 1);
#pragma unroll
        for (int i = 0; i < TWL; i++) {
            const int idx = lane + i * kWave;
            if (idx < N / 2) put_twiddle<N>(l_tw, idx, twr[i]);
        }
        io.block_to_lds(l_buf, a.B, lane);
        wave_sync();
        IRB_STAMP(a, 2);
        wave_fft<M, false>(l_buf, l_tw, lane);
        wave_split_forward<M>(l_buf, l_tw, lane, [&](int b, float2 v) { l_spec[b] = v; });
        wave_sync();
        IRB_STAMP(a, 3);
        if (lane < W) {                                // partition 0 of this slice; the slice also goes into the ring
            const float4 x0 = reinterpret_cast<const float4 *>(l_spec)[gcol];
            reinterpret_cast<float4 *>(Xs + (size_t) a.head * M)[gcol] = x0;
            if (np > 0) {
                dc = make_float2(x0.x * h0.x, x0.y * h0.y);
                acc.x = x0.x * h0.x - x0.y * h0.y;
                acc.y = x0.x * h0.y + x0.y * h0.x;
                acc.z = x0.z * h0.z - x0.w * h0.w;
                acc.w = x0.z * h0.w + x0.w * h0.z;
            }
        }
        IRB_STAMP(a, 4);
    } else {
        // loader waves: (first workgroup: waves 1..3, else all four) x (W columns) -> NS stripes
        const int ltid = first ? tid - kWave : tid;
        const int NS = (first ? kThreads - kWave : kThreads) / W;
        const int col = ltid % W, stripe = ltid / W, gcol = cs * W + col;
        float4 h[U], x[U];
        auto issue = [&](int k0) {
            const int kdup = k0 < ke ? k0 : 0;         // a slot past the range re-reads a row just asked for; masked on use
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int kk = k0 + u * NS < ke ? k0 + u * NS : kdup;
                int slot = a.head - kk;
                if (slot < 0) slot += a.R;
                h[u] = H4[(size_t) kk * C::L + gcol];
                x[u] = X4[(size_t) slot * C::L + gcol];
            }
        };
        auto consume = [&](int k0) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const bool live = k0 + u * NS < ke;
                const float4 xv = live ? x[u] : make_float4(0.f, 0.f, 0.f, 0.f);
                dc.x += xv.x * h[u].x;
                dc.y += xv.y * h[u].y;
                acc.x += xv.x * h[u].x - xv.y * h[u].y;
                acc.y += xv.x * h[u].y + xv.y * h[u].x;
                acc.z += xv.z * h[u].z - xv.w * h[u].w;
                acc.w += xv.z * h[u].w + xv.w * h[u].z;
            }
        };
        int k = kb + stripe;
        issue(k);
        __builtin_amdgcn_sched_barrier(0);
        IRB_STAMP(a, 1);
        if (wave == 0 && may_finish) {                 // twiddles for a possible inverse FFT later (this wave's own use)
#pragma unroll
            for (int i = 0; i < TWL; i++) {
                const int idx = lane + i * kWave;
                if (idx < N / 2) put_twiddle<N>(l_tw, idx, twr[i]);
            }
        }
        consume(k);
        for (k += NS * U; k < ke; k += NS * U) {       // ranges longer than one round (few splits)
            issue(k);
            consume(k);
        }
        IRB_STAMP(a, 4);
        // combine the stripes inside the wave by lane permutes: DPP rotations within rows of 16 lanes, permlane
        // swaps across the rows; lanes < W end up with the wave's column sums
        if constexpr (W <= 8) {
            acc.x += dpp_row<0x128>(acc.x); acc.y += dpp_row<0x128>(acc.y); acc.z += dpp_row<0x128>(acc.z); acc.w += dpp_row<0x128>(acc.w);
            dc.x += dpp_row<0x128>(dc.x); dc.y += dpp_row<0x128>(dc.y);
        }
        if constexpr (W <= 4) {
            acc.x += dpp_row<0x124>(acc.x); acc.y += dpp_row<0x124>(acc.y); acc.z += dpp_row<0x124>(acc.z); acc.w += dpp_row<0x124>(acc.w);
            dc.x += dpp_row<0x124>(dc.x); dc.y += dpp_row<0x124>(dc.y);
        }
        if constexpr (W <= 16) {
            acc.x = rows_sum(acc.x); acc.y = rows_sum(acc.y); acc.z = rows_sum(acc.z); acc.w = rows_sum(acc.w);
            dc.x = rows_sum(dc.x); dc.y = rows_sum(dc.y);
        } else if constexpr (W == 32) {
            const auto sw = [](float v) {
                const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                return __uint_as_float(b[0]) + __uint_as_float(b[1]);
            };
            acc.x = sw(acc.x); acc.y = sw(acc.y); acc.z = sw(acc.z); acc.w = sw(acc.w); dc.x = sw(dc.x); dc.y = sw(dc.y);
        }
    }
    if (lane < W) {                                    // W <= 64: lane == column of the slice
        l_red[wave][lane] = acc;
        if (lane == 0) l_dc[wave] = dc;
    }
    lds_barrier();
    float2 *slab = a.part + ((size_t) s * PS + ps) * M;
    if (tid < W) {
        float4 t = l_red[0][tid];
        float2 d = l_dc[0];
#pragma unroll
        for (int g = 1; g < NWAVES; g++) {
            const float4 v = l_red[g][tid];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            d.x += l_dc[g].x; d.y += l_dc[g].y;
        }
        const int b = 2 * (cs * W + tid);
        const float2 lo = b == 0 ? d : make_float2(t.x, t.y);       // packed bin 0: two real products
        const float2 hi = make_float2(t.z, t.w);
        if constexpr (HO == 2) {
            if (a.debug_drop && blockIdx.x == gridDim.x - 1 && !(cs == 0 && ps == 0)) return;   // test hook: a producer that never publishes
            unsigned long long *g = a.gran + ((size_t) s * PS + ps) * (2 * M) + 2 * b;     // one granule per float
            const unsigned long long tag = (unsigned long long) a.epoch << 32;
            __hip_atomic_store(g + 0, tag | __float_as_uint(lo.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(g + 1, tag | __float_as_uint(lo.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(g + 2, tag | __float_as_uint(hi.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(g + 3, tag | __float_as_uint(hi.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else { st_sc1(slab + b, lo); st_sc1(slab + b + 1, hi); }
    }

    if constexpr (HO == 2) {
        if (!may_finish) return;                       // producer: done (its granule stores complete on their own)
        // ---- reducer: sweep the stream's granules until every tag is this launch's epoch, summing the PS rows ----
        const unsigned long long *rows = a.gran + (size_t) s * PS * (2 * M);
        float *yf = reinterpret_cast<float *>(l_buf);
        const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();          // 100 MHz
        bool gave_up = false;
        // this thread's floats f = tid + 256 i, i < FPT, of every row r < PS: items q = r * FPT + i, swept 8 at a time
        // (all loads of a sweep in flight together; a sweep repeats until its 8 tags match)
        constexpr int FPT = N >= kThreads ? N / kThreads : 1;
        float v[FPT];
#pragma unroll
        for (int i = 0; i < FPT; i++) v[i] = 0.0f;
        const int items = FPT * PS;
        for (int q0 = 0; q0 < items; q0 += 8) {
            unsigned long long g[8];
            for (;;) {
                bool ready = true;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int q = q0 + j, f = tid + (q % FPT) * kThreads;
                    if (q < items && f < 2 * M) {
                        g[j] = __hip_atomic_load(rows + (size_t) (q / FPT) * (2 * M) + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ready = ready && (unsigned) (g[j] >> 32) == a.epoch;
                    } else {
                        g[j] = 0;                          // absent item: adds +0
                    }
                }
                if (ready) break;
                if (__builtin_amdgcn_s_memrealtime() - t_start > 5000000ull) { gave_up = true; break; }   // 50 ms
                __builtin_amdgcn_s_sleep(2);
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int i = (q0 + j) % FPT;
#pragma unroll
                for (int ii = 0; ii < FPT; ii++)           // static register index
                    if (ii == i) v[ii] += __uint_as_float((unsigned) g[j]);
            }
        }
#pragma unroll
        for (int i = 0; i < FPT; i++) {
            const int f = tid + i * kThreads;
            if (f < 2 * M) yf[f] = v[i];
        }
        if (gave_up) __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // a stream whose sum is incomplete writes neither output nor overlap: the host reports IRB_ERR_STATE for this step
        // and the caller resets the handle (the ring slot of this block has been written, the overlap is one block stale)
        if (__syncthreads_or(gave_up)) return;
        if (wave == 0) {
            wave_split_inverse<M>(l_buf, l_tw, lane);
            wave_fft<M, true>(l_buf, l_tw, lane);
            io.overlap_add(l_buf, a.B, a.ovl + (size_t) s * a.B, a.out + (size_t) s * a.out_stride, lane);
        }
        return;
    }

    // ---- publish: drained stores, barrier, one ticket ----
    const unsigned total = (unsigned) (CS * PS);
    IRB_STAMP(a, 5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    IRB_STAMP(a, 6);
    if (tid == 0) {
        l_ticket = __hip_atomic_fetch_add(&tickets[s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    IRB_STAMP(a, 7);
    if (l_ticket != total - 1) return;

    // ---- last arriver of this stream: sum the PS slabs, inverse FFT, overlap-add ----
    if (tid == 0) __hip_atomic_store(&tickets[s], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float2 *rows = a.part + (size_t) s * PS * M;
    for (int b = tid; b < M; b += kThreads) {
        float2 v = make_float2(0.f, 0.f);
        int r = 0;
        for (; r + 3 < PS; r += 4) {                   // four loads in flight
            float2 w[4];
#pragma unroll
            for (int i = 0; i < 4; i++)
                w[i] = ld_sc1(rows + (size_t) (r + i) * M + b);
            v.x += (w[0].x + w[1].x) + (w[2].x + w[3].x);
            v.y += (w[0].y + w[1].y) + (w[2].y + w[3].y);
        }
        for (; r < PS; r++) v = cadd(v, ld_sc1(rows + (size_t) r * M + b));
        l_buf[b] = v;
    }
    __syncthreads();
    IRB_STAMP(a, 8);
    if (wave == 0) {
        wave_split_inverse<M>(l_buf, l_tw, lane);
        wave_fft<M, true>(l_buf, l_tw, lane);
        IRB_STAMP(a, 9);
        io.overlap_add(l_buf, a.B, a.ovl + (size_t) s * a.B, a.out + (size_t) s * a.out_stride, lane);   // overlap prefetched at the top
#ifdef IRB_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        IRB_STAMP(a, 10);
#endif
    }
}

// ---------------------------------------------------------------------------------------------
// Time-tiled path: a call that carries T > 1 new blocks per stream (host blocks larger than B, or the
// whole signal in the offline function).  All T input spectra are pushed first, then every output
//   Y_{j}[b] = sum_k X_{t0+j-k}[b] * H_k[b],  j = 0..T-1
// is independent, and TT consecutive outputs share their H loads: a lane keeps a sliding window of
// 2*TT-1 X rows and TT accumulators in registers (per outer step: TT new X rows + TT H rows feed
// TT*TT complex multiply-adds per bin).  HBM/L2 traffic per output drops by ~TT versus streaming.
//   k_push_fft_batch   grid = S*t_count (block index fastest): blocks [t_first, t_first + t_count) of the call
//   k_cmac_tiled_pk    grid = S*ceil(T/TT)*splits: stripes own contiguous partition sub-ranges and
//                      write their partial spectra straight from registers: part[s][j][q][M]
//   k_ifft_ola_run     grid = S*ceil(T/NB): sums the partials of its blocks, inverse FFTs, overlap-add (fdl_tiled_pk.hpp)
// ---------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(kWave * FftGeom<N>::NF) void k_push_fft_batch(StepArgs a, int S)
{
    constexpr int M = N / 2, NF = FftGeom<N>::NF;
    __shared__ float2 l_tw[TwLds<N>::SIZE];
    __shared__ float2 l_buf[NF][M];
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const int item = blockIdx.x * NF + wave;
    const bool live = item < S * a.t_count;            // this launch: blocks [t_first, t_first + t_count) of the call
    const int j = a.t_first + item % a.t_count, s = item / a.t_count;
    TwiddleStage<N, kWave * NF> tws;
    tws.load(a.tw);
    if (live) wave_load_block_upto<M>(a.in + (long) s * a.in_stride + (long) j * a.B, a.B, a.in_len - (long) j * a.B, l_buf[wave], lane);
    if (live && a.dbg_sums) wave_block_checksum<N>(l_buf[wave], lane, a.dbg_sums + (size_t) s * a.T + j);
    tws.store(l_tw);
    __syncthreads();
    if (!live) return;
    wave_fft<M, false>(l_buf[wave], l_tw, lane);
    const int slot = (a.head + j) % a.R;
    float2 *dst = a.X + ((size_t) s * a.R + slot) * M;
    wave_split_forward<M>(l_buf[wave], l_tw, lane, [&](int b, float2 v) { dst[b] = v; });
}

__device__ __forceinline__ int ring_slot(int head, int d, int R)
{
    int sl = head + d;
    if (sl < 0) sl += R;
    if (sl >= R) sl -= R;
    return sl;
}

// Variant with ONE complex bin per lane and TT = 16 outputs per tile: half the bytes per FMA of round 2's two-bins-per-lane tile
// (per 16 partitions: 32 eight-byte loads feed 256 complex MACs), at the price of 8-byte loads and ~140 VGPRs.
template <int N, int TT>
__global__ __launch_bounds__(kThreads) void k_cmac_tiled1(StepArgs a)
{
    constexpr int M = N / 2;
    constexpr int LW = M < kThreads ? M : kThreads;          // lanes across a row
    constexpr int CPT = M / LW, STRIPES = kThreads / LW;
    const int ntiles = (a.T + TT - 1) / TT;
    const int wg = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int split = wg % a.splits, j0 = ((wg / a.splits) % ntiles) * TT, s = wg / (a.splits * ntiles);
    const int2 si = a.stream_info[s];
    const int set = si.x, np = si.y;
    const int col0 = threadIdx.x % LW;
    const int stripe = LW >= kWave ? __builtin_amdgcn_readfirstlane(threadIdx.x / LW) : (int) (threadIdx.x / LW);
    const int q = split * STRIPES + stripe;
    const int nG = (np + 7) / 8, nW = a.splits * STRIPES;      // H rows are zero-padded to multiples of 8
    const int sb = (int) ((long) nG * q / nW) * 8, se = (int) ((long) nG * (q + 1) / nW) * 8;
    const float2 *H2 = a.H + (size_t) set * a.P * M;
    const float2 *X2 = a.X + (size_t) s * a.R * M;
#pragma unroll 1
    for (int c = 0; c < CPT; c++) {
        const int col = col0 + c * LW;
        const bool packed = (col == 0);
        float2 acc[TT];
#pragma unroll
        for (int j = 0; j < TT; j++) acc[j] = make_float2(0.f, 0.f);
        int k0 = sb;
        if (k0 < se) {
            float2 xr[TT + 7];                                 // window for 8 partitions x TT outputs
#pragma unroll
            for (int i = 8; i < TT + 7; i++) xr[i] = X2[(size_t) ring_slot(a.head, j0 - k0 - 7 + i, a.R) * M + col];
            for (; k0 < se; k0 += 8) {
                float2 h[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    h[u] = H2[(size_t) (k0 + u) * M + col];
                    xr[7 - u] = X2[(size_t) ring_slot(a.head, j0 - k0 - u, a.R) * M + col];
                }
                __builtin_amdgcn_sched_barrier(0);                                    // loads stay in front
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const float p = packed ? 0.0f : h[u].y;
                    const float qq = packed ? h[u].y : h[u].x;
#pragma unroll
                    for (int j = 0; j < TT; j++) {
                        const float2 x = xr[j - u + 7];
                        acc[j].x = fmaf(x.x, h[u].x, acc[j].x);  acc[j].x = fmaf(-x.y, p, acc[j].x);
                        acc[j].y = fmaf(x.x, p, acc[j].y);       acc[j].y = fmaf(x.y, qq, acc[j].y);
                    }
                }
#pragma unroll
                for (int i = TT + 6; i >= 8; i--) xr[i] = xr[i - 8];
            }
        }
#pragma unroll
        for (int j = 0; j < TT; j++)
            if (j0 + j < a.T) a.part[(((size_t) s * a.T + j0 + j) * a.nq + q) * M + col] = acc[j];
    }
}

// one wave: sum the nq partial rows of one output block into buf (LDS)
template <int N>
__device__ __forceinline__ void wave_sum_rows(const float2 *__restrict__ src, int nq, float2 *buf, int lane)
{
    constexpr int M = N / 2, L = M / 2, CPL = L >= kWave ? L / kWave : 1;
    const float4 *src4 = reinterpret_cast<const float4 *>(src);
    float4 v[CPL];
#pragma unroll
    for (int c = 0; c < CPL; c++) v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    // four rows x all of this lane's columns per round, every load of the round issued before the first add (clamped
    // rows / columns instead of predicated loads; a `load; add` loop waits for each load in turn)
    for (int q0 = 0; q0 < nq; q0 += 4) {
        float4 w[4][CPL];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int c = 0; c < CPL; c++)
                w[i][c] = src4[(size_t) min(q0 + i, nq - 1) * L + min(lane + c * kWave, L - 1)];
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (q0 + i < nq) {
#pragma unroll
                for (int c = 0; c < CPL; c++) { v[c].x += w[i][c].x; v[c].y += w[i][c].y; v[c].z += w[i][c].z; v[c].w += w[i][c].w; }
            }
    }
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const int col = lane + c * kWave;
        if (L >= kWave || col < L) reinterpret_cast<float4 *>(buf)[col] = v[c];
    }
    wave_sync();
}

// ---------------------------------------------------------------------------------------------
// k_reset: what a fresh convolvePeriodic call starts from (fp/convolution.cpp:59,67,81,87) in ONE launch: the FDL ring, both overlap
// buffers, the history columns of the matrix-core path's transposed FDL, the granule tags of the column-sliced step.  Four runtime
// memsets cost four launches of ~4-5 us each in front of every one-shot call; the bytes themselves are a few microseconds.
// Segments of 16 bytes, grid-stride.
// ---------------------------------------------------------------------------------------------
struct ResetArgs {
    float4 *p[3];                // linear regions (ring, overlap, granules): pointers (16-byte aligned: hipMalloc) ...
    size_t n16[3];               // ... and lengths in 16-byte units (0: none)
    float4 *xt;                  // Xt rows: `rows` rows of `pitch16` units, of which the first `hist16` are cleared
    size_t rows, pitch16, hist16;
};

__global__ __launch_bounds__(256) void k_reset(ResetArgs r)
{
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t tid = (size_t) blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t) gridDim.x * blockDim.x;
    for (int k = 0; k < 3; k++)
        for (size_t i = tid; i < r.n16[k]; i += nth) r.p[k][i] = z;
    for (size_t i = tid; i < r.rows * r.hist16; i += nth) r.xt[(i / r.hist16) * r.pitch16 + i % r.hist16] = z;
}

// ---------------------------------------------------------------------------------------------
// k_ir_fft: IR partitions -> spectra (reference fp/convolution.cpp:106-125).  one wave per partition:
// row r of `ir` (time domain, [n_rows][P*B], zero padded) -> H set first_set + r.
// ---------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(kWave * FftGeom<N>::NF) void k_ir_fft(const float *__restrict__ ir, float2 *__restrict__ H,
                                                                   const float2 *__restrict__ tw, int B, int P, int first_set,
                                                                   int first_partition, int nparts, int nrows)
{
    constexpr int M = N / 2, NF = FftGeom<N>::NF;
    __shared__ float2 l_tw[TwLds<N>::SIZE];
    __shared__ float2 l_buf[NF][M];
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const int item = blockIdx.x * NF + wave;
    const bool live = item < nparts * nrows;
    const int k = item % nparts, r = item / nparts;
    TwiddleStage<N, kWave * NF> tws;
    tws.load(tw);
    if (live) wave_load_block<M>(ir + ((size_t) r * nparts + k) * B, B, l_buf[wave], lane);
    tws.store(l_tw);
    __syncthreads();
    if (!live) return;
    wave_fft<M, false>(l_buf[wave], l_tw, lane);
    float2 *dst = H + ((size_t) (first_set + r) * P + first_partition + k) * M;
    wave_split_forward<M>(l_buf[wave], l_tw, lane, [&](int b, float2 v) { dst[b] = v; });
}

// ---------------------------------------------------------------------------------------------
// k_ir_fft_direct: the same transform straight out of the CALLER's impulse response (registered host memory or device memory),
// for the one-shot function: no staging copy on the host, no copy command, and what irb_fdl_set_ir does around its k_ir_fft
// in separate device operations folded in -- the zero rows up to the tile period (rows np .. np_pad), the stream records.
// Row r of the spectra = source r, or (average = 1) (src[0] + src[1]) / 2 sample by sample, the host's order of operations
// (tools::sumToMono, fp/tools.cpp:25-28); samples at and beyond `len` read as zeros (fp/convolution.cpp:106-118).
// one wave per row of H: grid = ceil(nrows * np_pad / NF).
// ---------------------------------------------------------------------------------------------
struct IrDirectArgs {
    const float *src[2];
    float2 *H;                   // [set][P][M]
    const float2 *tw;
    int2 *stream_info;           // records [0, ninfo) are written by the launch (nullptr: none)
    int2 info[2];
    int ninfo, average, nrows, len, B, P, first_set, np, np_pad;
};

template <int N>
__global__ __launch_bounds__(kWave * FftGeom<N>::NF) void k_ir_fft_direct(IrDirectArgs a)
{
    constexpr int M = N / 2, NF = FftGeom<N>::NF;
    __shared__ float2 l_tw[TwLds<N>::SIZE];
    __shared__ float2 l_buf[NF][M];
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    if (blockIdx.x == 0 && threadIdx.x < a.ninfo && a.stream_info) a.stream_info[threadIdx.x] = a.info[threadIdx.x];
    const int item = blockIdx.x * NF + wave;
    const bool live = item < a.nrows * a.np_pad;
    const int k = live ? item % a.np_pad : 0, r = live ? item / a.np_pad : 0;
    float2 *dst = a.H + ((size_t) (a.first_set + r) * a.P + k) * M;
    if (live && k >= a.np) {                           // a zero row behind the IR's partitions (wave-uniform)
        for (int b = lane; b < M; b += kWave) dst[b] = make_float2(0.f, 0.f);
    }
    const bool fft = live && k < a.np;
    TwiddleStage<N, kWave * NF> tws;
    tws.load(a.tw);
    if (fft) {
        float *zf = reinterpret_cast<float *>(l_buf[wave]);
        const float *s0 = a.src[a.average ? 0 : r], *s1 = a.src[1];
        const long base = (long) k * a.B;
        for (int n = lane; n < N; n += kWave) {
            const long idx = base + n;
            float v = 0.0f;
            if (n < a.B && idx < a.len) {
                v = s0[idx];
                if (a.average) { v += s1[idx]; v /= 2.0f; }
            }
            zf[n] = v;
        }
        wave_sync();
    }
    tws.store(l_tw);
    __syncthreads();
    if (!fft) return;
    wave_fft<M, false>(l_buf[wave], l_tw, lane);
    wave_split_forward<M>(l_buf[wave], l_tw, lane, [&](int b, float2 v) { dst[b] = v; });
}

}  // namespace irb

#include "fdl_tiled_pk.hpp"
#include "fdl_mfma.hpp"
#include "fdl_rt_kernel.hpp"
