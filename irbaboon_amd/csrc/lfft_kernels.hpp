// lfft_kernels.hpp -- large real FFT for the config-5 deconvolution chain (gfx950).
//
// tools::fftTransform / fftInvTransform (reference fp/tools.cpp:321-369) transform ONE long buffer
// (2^16 in the plug-in, 2^20 in BASELINE config 5).  An N-point real transform is an M = N/2 point
// complex transform plus a split pass; for M > 2048 the complex transform is the classic four-step
// factorisation M = M1*M2 built from the one-wave LDS transform of fft_wave.hpp:
//   k_lfft_cols<M1>   one wave per column n2: gather z[n1*M2+n2], M1-point FFT, multiply by W_M^{n2*k1},
//                     store T[k1*M2+n2]
//   k_lfft_rows<M2>   one wave per row k1: M2-point FFT of T[k1*M2 + .], store Z[k1 + M1*k2]
// Twiddles: per-size half tables for the wave FFT; W_N^e for the inter-pass factor and the real split
// comes from two small tables, W_N^e = A[e / E] * B[e % E] (both computed in double on the host).
// Spectrum layout at the interface is the reference's: 2N floats, interleaved (re, im) for bins 0..N/2
// (Nyquist at float index N), the rest zero.
#pragma once
#include <hip/hip_runtime.h>

#include "fft_wave.hpp"

namespace irb {

struct LfftArgs {
    const float2 *tw1, *tw2;      // half tables for the M1- and M2-point wave FFTs (M1, M2 entries)
    const float2 *tabA, *tabB;    // W_N^(i*E), i < N/E ; W_N^i, i < E
    int           logE;           // E = 1 << logE
    int           N, M, M1, M2, batch;
    float         scale;          // applied by the last pass of an inverse transform (1/N), 1 otherwise
    int           zero_fill;      // forward: write the zeros above the Nyquist bin (0: the caller's buffer is known to hold them)
};

__device__ __forceinline__ float2 big_twiddle(const LfftArgs &a, unsigned e)    // W_N^e, e < N
{
    return cmul(a.tabA[e >> a.logE], a.tabB[e & ((1u << a.logE) - 1u)]);
}

template <int M1> struct WaveGeom { static constexpr int NF = M1 <= 1024 ? 4 : 2; };

// one wave copies M complex values src[n * stride], n < M, into LDS: its global loads go out in batches of up to 8
// before the LDS stores of the batch (a `load; store to LDS` loop makes hipcc wait for every load in turn)
template <int M>
__device__ __forceinline__ void wave_gather(const float2 *__restrict__ src, size_t stride, float2 *l_dst, int lane)
{
    constexpr int PL = M >= kWave ? M / kWave : 1, CH = PL < 8 ? PL : 8;
#pragma unroll 1
    for (int c0 = 0; c0 < PL; c0 += CH) {
        float2 v[CH];
#pragma unroll
        for (int i = 0; i < CH; i++) v[i] = src[(size_t) min(lane + (c0 + i) * kWave, M - 1) * stride];
#pragma unroll
        for (int i = 0; i < CH; i++) {
            const int n = lane + (c0 + i) * kWave;
            if (M >= kWave || n < M) l_dst[n] = v[i];
        }
    }
}

// ---- pass 1: columns ------------------------------------------------------------------------------
template <int M1, bool INV>
__global__ __launch_bounds__(kWave * WaveGeom<M1>::NF) void k_lfft_cols(LfftArgs a, const float2 *__restrict__ in, float2 *__restrict__ out)
{
    constexpr int NF = WaveGeom<M1>::NF;
    __shared__ float2 l_tw[TwLds<2 * M1>::SIZE];
    __shared__ float2 l_buf[NF][M1];
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const long item = (long) blockIdx.x * NF + wave;
    const bool live = item < (long) a.batch * a.M2;
    const int n2 = (int) (item % a.M2);
    const long b = item / a.M2;
    for (int i = threadIdx.x; i < M1; i += blockDim.x) put_twiddle<2 * M1>(l_tw, i, a.tw1[i]);
    if (live) {
        const float2 *src = in + b * a.M + n2;
        wave_gather<M1>(src, (size_t) a.M2, l_buf[wave], lane);
    }
    __syncthreads();
    if (!live) return;
    wave_fft<M1, INV>(l_buf[wave], l_tw, lane);
    float2 *dst = out + b * a.M + n2;
    for (int k1 = lane; k1 < M1; k1 += kWave) {
        const float2 w = big_twiddle(a, 2u * (unsigned) n2 * (unsigned) k1);     // W_M^(n2*k1) = W_N^(2*n2*k1)
        const float2 v = l_buf[wave][k1];
        dst[(size_t) k1 * a.M2] = INV ? cmulc(v, w) : cmul(v, w);
    }
}

// ---- pass 2: rows ---------------------------------------------------------------------------------
template <int M2, bool INV>
__global__ __launch_bounds__(kWave * WaveGeom<M2>::NF) void k_lfft_rows(LfftArgs a, const float2 *__restrict__ in, float2 *__restrict__ out)
{
    constexpr int NF = WaveGeom<M2>::NF;
    __shared__ float2 l_tw[TwLds<2 * M2>::SIZE];
    __shared__ float2 l_buf[NF][M2];
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const long item = (long) blockIdx.x * NF + wave;
    const bool live = item < (long) a.batch * a.M1;
    const int k1 = (int) (item % a.M1);
    const long b = item / a.M1;
    for (int i = threadIdx.x; i < M2; i += blockDim.x) put_twiddle<2 * M2>(l_tw, i, a.tw2[i]);
    if (live) {
        const float2 *src = in + b * a.M + (size_t) k1 * M2;
        wave_gather<M2>(src, 1, l_buf[wave], lane);
    }
    __syncthreads();
    if (!live) return;
    wave_fft<M2, INV>(l_buf[wave], l_tw, lane);
    float2 *dst = out + b * a.M + k1;
    for (int k2 = lane; k2 < M2; k2 += kWave) {
        const float2 v = l_buf[wave][k2];
        dst[(size_t) k2 * a.M1] = make_float2(v.x * a.scale, v.y * a.scale);
    }
}

// ---- tiled passes: R adjacent columns (pass 1) / rows (pass 2) per workgroup, one wave each, with the global
// traffic staged through a padded LDS tile so that every global access is an R*8-byte contiguous piece
// (the one-wave-per-item kernels above touch 8 bytes per 4-KiB stride in one of their two phases).
// Dynamic LDS: R*(MX+1) + 3*MX/2 float2 (tile + the wave FFT's twiddle table).
template <int MX, int R, bool INV, bool COLS>
__global__ __launch_bounds__(kWave * R) void k_lfft_tile(LfftArgs a, const float2 *__restrict__ in, float2 *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char l_raw[];
    constexpr int LD = MX + 1;                               // padded row: transposed accesses spread over banks
    constexpr int NT = kWave * R;
    float2 *tile = reinterpret_cast<float2 *>(l_raw);        // [R][LD]
    float2 *l_tw = tile + R * LD;                            // [TwLds<2*MX>::SIZE]
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const int other = COLS ? a.M2 : a.M1;                    // length of the dimension the groups run along
    const int groups = other / R;
    const int g = blockIdx.x % groups;
    const long b = blockIdx.x / groups;
    const int i0 = g * R;                                    // first column n2 (COLS) / first row k1 (rows)
    const float2 *src = in + b * a.M;
    float2 *dst = out + b * a.M;
    // every global load of this thread is issued before the first LDS store: one memory latency for the tile (and the
    // twiddles) instead of one per element -- a `load; store to LDS` loop makes hipcc wait for each load in turn
    constexpr int EPT = MX >= kWave ? MX / kWave : 1;        // tile elements per thread: MX*R / (64*R)
    constexpr int TPT = (MX + NT - 1) / NT;                  // twiddle entries per thread
    float2 twv[TPT], v[EPT];
    const float2 *twsrc = COLS ? a.tw1 : a.tw2;
    // the inter-pass factor's table entries (columns pass) are asked for now, with everything else this thread loads: they
    // are needed only after the column transform, and a load issued there is a round trip nothing overlaps
    constexpr int KPL = MX >= kWave ? MX / kWave : 1;
    float2 wa[COLS ? KPL : 1], wb[COLS ? KPL : 1];
    if (COLS) {
        const unsigned n2 = (unsigned) (i0 + wave);
#pragma unroll
        for (int i = 0; i < KPL; i++) {
            const unsigned e = 2u * n2 * (unsigned) min(lane + i * kWave, MX - 1);
            wa[i] = a.tabA[e >> a.logE];
            wb[i] = a.tabB[e & ((1u << a.logE) - 1u)];
        }
    }
#pragma unroll
    for (int i = 0; i < TPT; i++) twv[i] = twsrc[min((int) threadIdx.x + i * NT, MX - 1)];
#pragma unroll
    for (int i = 0; i < EPT; i++) {
        const int idx = min((int) threadIdx.x + i * NT, MX * R - 1);   // clamped, not predicated (MX < 64 only)
        if (COLS) v[i] = src[(size_t) (idx / R) * a.M2 + i0 + idx % R];          // x[n1*M2 + n2]: R contiguous n2 per n1
        else v[i] = src[(size_t) (i0 + idx / MX) * MX + idx % MX];               // T[k1*M2 + n2]: whole rows
    }
#pragma unroll
    for (int i = 0; i < TPT; i++) {
        const int idx = threadIdx.x + i * NT;
        if (idx < MX) put_twiddle<2 * MX>(l_tw, idx, twv[i]);
    }
#pragma unroll
    for (int i = 0; i < EPT; i++) {
        const int idx = threadIdx.x + i * NT;
        if (MX >= kWave || idx < MX * R) {
            if (COLS) tile[(idx % R) * LD + idx / R] = v[i];
            else tile[(idx / MX) * LD + idx % MX] = v[i];
        }
    }
    __syncthreads();
    float2 *mine = tile + wave * LD;
    wave_fft<MX, INV>(mine, l_tw, lane);
    if (COLS) {
        // inter-pass factor W_M^(n2*k1) = A[e / E] * B[e % E] (loaded at the top of the kernel, see wa / wb)
#pragma unroll
        for (int i = 0; i < KPL; i++) {
            const int k1 = lane + i * kWave;
            if (MX >= kWave || k1 < MX) {
                const float2 w = cmul(wa[i], wb[i]);
                mine[k1] = INV ? cmulc(mine[k1], w) : cmul(mine[k1], w);
            }
        }
    }
    __syncthreads();
    if (COLS) {
        for (int idx = threadIdx.x; idx < MX * R; idx += NT) {          // T[k1*M2 + n2]
            const int k1 = idx / R, r = idx % R;
            dst[(size_t) k1 * a.M2 + i0 + r] = tile[r * LD + k1];
        }
    } else {
        for (int idx = threadIdx.x; idx < MX * R; idx += NT) {          // Z[k1 + M1*k2]: R contiguous k1 per k2
            const int k2 = idx / R, r = idx % R;
            const float2 v = tile[r * LD + k2];
            dst[(size_t) k2 * a.M1 + i0 + r] = make_float2(v.x * a.scale, v.y * a.scale);
        }
    }
}

// ---- the real <-> complex split fused into the tiled passes (single launches saved: forward = cols pass + this rows
// pass, inverse = this cols pass + rows pass) ---------------------------------------------------------------------------
//
// forward, second pass: a workgroup takes R rows k1 of T that are closed under the mirror e -> M - e of the split
// (e = k1 + M1*k2  <->  M - e = (M1 - k1) + M1*(M2 - 1 - k2)): R/2 pairs (p, M1 - p); pair 0 is the two self-mirrored rows
// (0, M1/2).  Tile slots: r < R/2 -> row P0 + r, r >= R/2 -> the partner of slot R-1-r, so that partner(r) = R-1-r and both
// halves are ascending runs of adjacent rows (adjacent threads write adjacent bins).  After the row FFTs every bin
// X[e] = E + W_N^e * O is formed from Z[e] and Z[M-e], both in LDS, and written straight into the spectrum buffer
// (2N floats per item, bins 0..M, zeros above: the workgroups share the zero fill).
template <int MX, int R>
__global__ __launch_bounds__(kWave * R) void k_lfft_rows_split(LfftArgs a, const float2 *__restrict__ in, float *__restrict__ spec)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char l_raw[];
    constexpr int LD = MX + 1, NT = kWave * R, H = R / 2;
    constexpr int EPT = MX >= kWave ? MX / kWave : 1, TPT = (MX + NT - 1) / NT;
    float2 *tile = reinterpret_cast<float2 *>(l_raw);        // [R][LD]
    float2 *l_tw = tile + R * LD;
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const int M1 = a.M1, M = a.M;
    const int groups = M1 / R;                               // (M1/2 pairs) / (R/2 pairs per workgroup)
    const int g = blockIdx.x % groups;
    const long b = blockIdx.x / groups;
    const int P0 = g * H;
    auto row_of = [&](int r) {
        const int p = r < H ? P0 + r : P0 + (R - 1 - r);
        return r < H ? p : (p == 0 ? M1 / 2 : M1 - p);
    };
    const float2 *src = in + b * M;
    float2 *X = reinterpret_cast<float2 *>(spec + b * 2 * a.N);

    __shared__ float2 l_w1[R];                               // W_N^k1 of the tile's rows
    float2 twv[TPT], v[EPT];
    float2 w1a = make_float2(1.f, 0.f), w1b = w1a;
    if (threadIdx.x < R) {
        const unsigned k1 = (unsigned) row_of(threadIdx.x);
        w1a = a.tabA[k1 >> a.logE];
        w1b = a.tabB[k1 & ((1u << a.logE) - 1u)];
    }
#pragma unroll
    for (int i = 0; i < TPT; i++) twv[i] = a.tw2[min((int) threadIdx.x + i * NT, MX - 1)];
#pragma unroll
    for (int i = 0; i < EPT; i++) {
        const int idx = min((int) threadIdx.x + i * NT, MX * R - 1);
        v[i] = src[(size_t) row_of(idx / MX) * MX + idx % MX];                       // T[k1*M2 + n2]: whole rows
    }
    if (threadIdx.x < R) l_w1[threadIdx.x] = cmul(w1a, w1b);
#pragma unroll
    for (int i = 0; i < TPT; i++) {
        const int idx = threadIdx.x + i * NT;
        if (idx < MX) put_twiddle<2 * MX>(l_tw, idx, twv[i]);
    }
#pragma unroll
    for (int i = 0; i < EPT; i++) {
        const int idx = threadIdx.x + i * NT;
        if (MX >= kWave || idx < MX * R) tile[(idx / MX) * LD + idx % MX] = v[i];
    }
    __syncthreads();
    wave_fft<MX, false>(tile + wave * LD, l_tw, lane);
    __syncthreads();

    // split: thread -> (k2 = idx / R, slot r = idx % R).  W_N^e with e = k1 + M1*k2 factors into W_N^k1 (one value per tile
    // slot: R table look-ups per workgroup, issued at the top) times W_N^(M1*k2) = W_(2*M2)^k2 -- the row transform's own
    // table, already in LDS.  (Round 2 fetched two table entries from global memory per output bin.)
#pragma unroll
    for (int i = 0; i < EPT; i++) {
        const int idx = threadIdx.x + i * NT;
        if (MX >= kWave || idx < MX * R) {
            const int r = idx % R, k2 = idx / R, k1 = row_of(r);
            const int e = k1 + M1 * k2;
            const bool self = (P0 == 0) && (r == 0 || r == R - 1);       // rows 0 and M1/2 mirror onto themselves
            const int mr = self ? r : R - 1 - r;
            const int mc = (self && r == 0) ? (MX - k2) & (MX - 1) : MX - 1 - k2;
            const float2 p = tile[r * LD + k2], q = tile[mr * LD + mc];
            if (e == 0) {
                X[0] = make_float2(p.x + p.y, 0.0f);
                X[M] = make_float2(p.x - p.y, 0.0f);
            } else {
                const float2 ev = make_float2(0.5f * (p.x + q.x), 0.5f * (p.y - q.y));      // (p + conj q)/2
                const float2 od = make_float2(0.5f * (p.y + q.y), -0.5f * (p.x - q.x));     // (p - conj q)/(2i)
                X[e] = cadd(ev, cmul(cmul(l_w1[r], l_tw[k2]), od));
            }
        }
    }
    if (a.zero_fill) {
        // zeros above the Nyquist bin (float2 indices M+1 .. N-1), shared over the workgroups of this item; skipped for the
        // plan's own spectrum buffer, whose upper half nothing ever writes after the plan zeroed it
        const int per = (M - 1 + groups - 1) / groups;
        for (int i = threadIdx.x; i < per; i += NT) {
            const int z = g * per + i;
            if (z < M - 1) X[M + 1 + z] = make_float2(0.f, 0.f);
        }
    }
}

// inverse, first pass: the columns pass reading the SPECTRUM: z[e] = E' + i*O' from X[e] and X[M-e] (plain global
// loads of the mirrored bin: no pairing constraint), then the column FFTs and the inter-pass factor as k_lfft_tile.
template <int MX, int R>
__global__ __launch_bounds__(kWave * R) void k_lfft_cols_unsplit(LfftArgs a, const float *__restrict__ spec, float2 *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char l_raw[];
    constexpr int LD = MX + 1, NT = kWave * R;
    constexpr int EPT = MX >= kWave ? MX / kWave : 1, TPT = (MX + NT - 1) / NT;
    constexpr int CH = EPT < 8 ? EPT : 8;                    // elements per thread per round (4 loads each in flight)
    float2 *tile = reinterpret_cast<float2 *>(l_raw);
    float2 *l_tw = tile + R * LD;
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const int M = a.M, M2 = a.M2;
    const int groups = M2 / R;
    const int g = blockIdx.x % groups;
    const long b = blockIdx.x / groups;
    const int i0 = g * R;
    const float2 *X = reinterpret_cast<const float2 *>(spec + b * 2 * a.N);
    float2 *dst = out + b * M;

    __shared__ float2 l_w2[R];                               // W_N^n2 of the tile's columns
    float2 twv[TPT];
    // inter-pass factor of this wave's column (needed after the column transform): asked for at the top
    constexpr int KPL = MX >= kWave ? MX / kWave : 1;
    float2 wa2[KPL], wb2[KPL];
    {
        const unsigned n2 = (unsigned) (i0 + wave);
#pragma unroll
        for (int i = 0; i < KPL; i++) {
            const unsigned e = 2u * n2 * (unsigned) min(lane + i * kWave, MX - 1);
            wa2[i] = a.tabA[e >> a.logE];
            wb2[i] = a.tabB[e & ((1u << a.logE) - 1u)];
        }
    }
    float2 c2a = make_float2(1.f, 0.f), c2b = c2a;
    if (threadIdx.x < R) {
        const unsigned n2 = (unsigned) (i0 + threadIdx.x);
        c2a = a.tabA[n2 >> a.logE];
        c2b = a.tabB[n2 & ((1u << a.logE) - 1u)];
    }
#pragma unroll
    for (int i = 0; i < TPT; i++) twv[i] = a.tw1[min((int) threadIdx.x + i * NT, MX - 1)];
#pragma unroll 1
    for (int c0 = 0; c0 < EPT; c0 += CH) {
        float2 p[CH], q[CH];
#pragma unroll
        for (int i = 0; i < CH; i++) {
            const int idx = min((int) threadIdx.x + (c0 + i) * NT, MX * R - 1);
            const int e = (idx / R) * M2 + i0 + idx % R;                            // x[n1*M2 + n2]: R contiguous n2 per n1
            p[i] = X[e];
            q[i] = X[M - e];                                                         // e = 0 reads the Nyquist bin X[M]
        }
        if (c0 == 0) {
            // the split twiddle W_N^e, e = n1*M2 + n2, is W_N^n2 (one value per tile column) times W_N^(n1*M2) = W_(2*M1)^n1 -- the
            // column transform's own table, staged in LDS while the first bins are in flight (round 2 fetched two table entries
            // from global memory per element)
#pragma unroll
            for (int i = 0; i < TPT; i++) {
                const int idx = threadIdx.x + i * NT;
                if (idx < MX) put_twiddle<2 * MX>(l_tw, idx, twv[i]);
            }
            if (threadIdx.x < R) l_w2[threadIdx.x] = cmul(c2a, c2b);
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < CH; i++) {
            const int idx = threadIdx.x + (c0 + i) * NT;
            if (MX >= kWave || idx < MX * R) {
                const int e = (idx / R) * M2 + i0 + idx % R;
                float2 z;
                if (e == 0) {
                    // performRealOnlyInverseTransform rebuilds the negative half from bins 0..N/2 and takes the real
                    // part of the result: only Re of DC and Nyquist survive
                    z = make_float2(p[i].x + q[i].x, p[i].x - q[i].x);
                } else {
                    const float2 ev = make_float2(p[i].x + q[i].x, p[i].y - q[i].y);        // p + conj q
                    const float2 d = make_float2(p[i].x - q[i].x, p[i].y + q[i].y);         // p - conj q
                    const float2 o = cmulc(d, cmul(l_w2[idx % R], l_tw[idx / R]));          // * W_N^{-e}
                    z = make_float2(ev.x - o.y, ev.y + o.x);
                }
                tile[(idx % R) * LD + idx / R] = z;
            }
        }
    }
    __syncthreads();
    float2 *mine = tile + wave * LD;
    wave_fft<MX, true>(mine, l_tw, lane);
#pragma unroll
    for (int i = 0; i < KPL; i++) {
        const int k1 = lane + i * kWave;
        if (MX >= kWave || k1 < MX) mine[k1] = cmulc(mine[k1], cmul(wa2[i], wb2[i]));
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < MX * R; idx += NT) {              // T[k1*M2 + n2]
        const int k1 = idx / R, r = idx % R;
        dst[(size_t) k1 * M2 + i0 + r] = tile[r * LD + k1];
    }
}

// ---- M <= 2048: the whole complex transform in one wave ----------------------------------------------
template <int M, bool INV>
__global__ __launch_bounds__(kWave * WaveGeom<M>::NF) void k_lfft_small(LfftArgs a, const float2 *__restrict__ in, float2 *__restrict__ out)
{
    constexpr int NF = WaveGeom<M>::NF;
    __shared__ float2 l_tw[TwLds<2 * M>::SIZE];
    __shared__ float2 l_buf[NF][M];
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const long b = (long) blockIdx.x * NF + wave;
    const bool live = b < a.batch;
    for (int i = threadIdx.x; i < M; i += blockDim.x) put_twiddle<2 * M>(l_tw, i, a.tw1[i]);
    if (live) wave_gather<M>(in + b * M, 1, l_buf[wave], lane);
    __syncthreads();
    if (!live) return;
    wave_fft<M, INV>(l_buf[wave], l_tw, lane);
    for (int n = lane; n < M; n += kWave) {
        const float2 v = l_buf[wave][n];
        out[b * M + n] = make_float2(v.x * a.scale, v.y * a.scale);
    }
}

// ---- real <-> complex split passes (element-wise over bin pairs b, M-b) ---------------------------------
// forward: Z = FFT_M(z) -> spec: 2N floats per batch item, bins 0..M interleaved, zeros above
// (performRealOnlyForwardTransform's layout; unnormalised)
__global__ __launch_bounds__(256) void k_rsplit_fwd(LfftArgs a, const float2 *__restrict__ Z, float *__restrict__ spec)
{
    const int M = a.M;
    const long idx = (long) blockIdx.x * blockDim.x + threadIdx.x;
    const long per = M / 2 + 1;
    if (idx >= per * a.batch) return;
    const int b = (int) (idx % per);
    const long bt = idx / per;
    const float2 *z = Z + bt * M;
    float2 *X = reinterpret_cast<float2 *>(spec + bt * 2 * a.N);
    if (b == 0) {
        X[0] = make_float2(z[0].x + z[0].y, 0.0f);
        X[M] = make_float2(z[0].x - z[0].y, 0.0f);
    } else {
        const float2 p = z[b], q = z[M - b];
        const float2 e = make_float2(0.5f * (p.x + q.x), 0.5f * (p.y - q.y));
        const float2 o = make_float2(0.5f * (p.y + q.y), -0.5f * (p.x - q.x));
        const float2 t = cmul(big_twiddle(a, (unsigned) b), o);
        X[b] = cadd(e, t);
        if (b != M - b) X[M - b] = make_float2(e.x - t.x, t.y - e.y);
    }
    // clear the unused upper half (float indices N+2 .. 2N): bins M+1 .. N-1
    if (a.zero_fill)
        for (long i = idx + M + 1 - bt * per; i < a.N; i += per) X[i] = make_float2(0.f, 0.f);
}

// inverse: spec (bins 0..M) -> Z with IFFT_M(Z)[n] = N*(y[2n] + i*y[2n+1])
__global__ __launch_bounds__(256) void k_rsplit_inv(LfftArgs a, const float *__restrict__ spec, float2 *__restrict__ Z)
{
    const int M = a.M;
    const long idx = (long) blockIdx.x * blockDim.x + threadIdx.x;
    const long per = M / 2 + 1;
    if (idx >= per * a.batch) return;
    const int b = (int) (idx % per);
    const long bt = idx / per;
    const float2 *X = reinterpret_cast<const float2 *>(spec + bt * 2 * a.N);
    float2 *z = Z + bt * M;
    if (b == 0) {
        // performRealOnlyInverseTransform rebuilds the negative half from bins 0..N/2 and takes the real
        // part of the result: only Re of DC and Nyquist survive
        z[0] = make_float2(X[0].x + X[M].x, X[0].x - X[M].x);
    } else {
        const float2 p = X[b], q = X[M - b];
        const float2 e = make_float2(p.x + q.x, p.y - q.y);
        const float2 d = make_float2(p.x - q.x, p.y + q.y);
        const float2 o = cmulc(d, big_twiddle(a, (unsigned) b));
        z[b] = make_float2(e.x - o.y, e.y + o.x);
        if (b != M - b) z[M - b] = make_float2(e.x + o.y, o.x - e.y);
    }
}

// z (M complex per item, = N reals interleaved) * 1/N -> N real samples
__global__ __launch_bounds__(256) void k_scale_real(const float *__restrict__ z, float *__restrict__ y, long n, float scale)
{
    const long i = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = z[i] * scale;
}

// tools::complexDivCartesian over bins 0..N/2 (reference fp/convolution.cpp:373-378, fp/tools.cpp:72-86):
// num /= den in place, a zero denominator leaves the numerator as it is
__global__ __launch_bounds__(256) void k_cdiv(float2 *__restrict__ num, const float2 *__restrict__ den, int nbins)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbins) return;
    const float a = num[i].x, b = num[i].y, c = den[i].x, d = den[i].y;
    if (c == 0.0f && d == 0.0f) return;
    const float den2 = c * c + d * d;
    num[i] = make_float2((a * c + b * d) / den2, (b * c - a * d) / den2);
}

// ---- averagingFilter, element-wise phases (reference fp/convolution.cpp:429-436 and :531-543) -------------
// phase A: log of each bin's amplitude, amplitude first clamped to +-1e-16 (fp/tools.cpp:213-231)
__global__ __launch_bounds__(256) void k_smooth_log_ampl(const float2 *__restrict__ spec, float *__restrict__ logA, int nbins)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbins) return;
    const double re = spec[i].x, im = spec[i].y;
    float a = (float) sqrt(re * re + im * im);            // binAmpl: sqrt(pow(re,2) + pow(im,2)) in double
    if (a < 1e-16f) a = 1e-16f;                           // amplitudes are never negative
    logA[i] = logf(a);
}

// phase C: new amplitude exp(avg), old phase, back to (re, im); include flags as in the reference
__global__ __launch_bounds__(256) void k_smooth_apply(float2 *__restrict__ spec, const float *__restrict__ avg, int nbins,
                                                      int log_avg, int include_phase, int include_amplitude)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbins) return;
    float na = avg[i];
    if (log_avg) na = expf(na);
    if (!signbit(na) && na < 1e-16f) na = 1e-16f;         // roundTo1TenQuadrillionth
    if (signbit(na) && na > -1e-16f) na = -1e-16f;
    float re = spec[i].x, im = spec[i].y;
    if (!signbit(re) && re < 1e-11f) re = 0.0f;           // roundToZero(.., 1e-11)
    if (signbit(re) && re > -1e-11f) re = 0.0f;
    if (!signbit(im) && im < 1e-11f) im = 0.0f;
    if (signbit(im) && im > -1e-11f) im = 0.0f;
    float phase = atan2f(im, re);
    if (!include_amplitude) na = 1.0f;
    if (!include_phase) phase = 0.0f;
    spec[i] = make_float2(na * cosf(phase), na * sinf(phase));
}

// tools::fftTransform(buffer, formatAmplPhase = true), reference fp/tools.cpp:337-342: every bin 0..N/2 becomes
// (binAmpl, binPhase).  The reference stores the amplitude into the real slot FIRST and only then calls binPhase on the same
// two floats, so the phase it keeps is atan2(im, amplitude), not atan2(im, re): restated as written.
__global__ __launch_bounds__(256) void k_ampl_phase(float2 *__restrict__ spec, int nbins)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbins) return;
    const double re = spec[i].x, im = spec[i].y;
    const float ampl = (float) sqrt(re * re + im * im);       // binAmpl: sqrt(pow(re, 2.0) + pow(im, 2.0)) in double (fp/tools.cpp:222-225)
    spec[i] = make_float2(ampl, atan2f(spec[i].y, ampl));     // binPhase on (amplitude, im) (fp/tools.cpp:228-231, :339-340)
}

// tools::complexMul over bins 0..N/2 (reference fp/convolution.cpp:326-331, fp/tools.cpp:44-52): a *= h
__global__ __launch_bounds__(256) void k_cmul(float2 *__restrict__ a, const float2 *__restrict__ h, int nbins)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbins) return;
    const float2 x = a[i], c = h[i];
    a[i] = make_float2(x.x * c.x - x.y * c.y, x.y * c.x + x.x * c.y);
}

}  // namespace irb
