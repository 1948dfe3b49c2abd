// fft_wave.hpp -- one-wave (64 lanes) complex FFT in LDS for gfx950.
//
// An M-point transform (M = N/2 of the real FFT, 16..2048) is owned by ONE wavefront: no workgroup
// barrier between stages, so a 256-thread workgroup runs four independent transforms.  Stockham
// autosort, radix-4 stages plus one radix-2 stage when log2 M is odd, done IN PLACE: a wave executes in
// lockstep and the LDS serves one wave's instructions in order, so when every lane first reads all the
// points of its butterflies and only then writes, no point is overwritten before it was read.  A
// wavefront-scope fence + wave_barrier between the phases stops the compiler from moving a lane's own
// LDS accesses across them.
//
// The stages are unrolled at compile time (strides, masks and twiddle steps are immediates), and a stage
// issues its twiddle reads together with its data reads, so one LDS round trip per stage is exposed
// instead of two.  A lone wave issues one vector instruction per ~4 cycles, which makes instruction count
// the other cost of this code (measured with tools/step_stamps: profiles/r02_step_stamps.txt); twiddles
// therefore come from a table of 3N/4 entries W_N^j staged once per workgroup in LDS (the host table holds
// j < N/2; W_N^{j+N/2} = -W_N^j is written at staging time), which covers every index a radix-4 stage asks
// for without sign fix-ups.
// Index arithmetic is the same as in tests/kernel_model.py.
#pragma once
#include <hip/hip_runtime.h>

namespace irb {

constexpr int kWave = 64;

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) { /* a * conj(b) */ return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// orders one wave's LDS traffic: everything before is complete and visible to the other lanes
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// LDS twiddle entries of an N-point real transform (M = N/2 point complex FFT + split): W_N^j, j < 3N/4
template <int N> struct TwLds { static constexpr int SIZE = 3 * N / 4; };

// one entry of the host's half table (W_N^i, i < N/2) into the LDS table: entry i, and for i < N/4 also i + N/2
template <int N>
__device__ __forceinline__ void put_twiddle(float2 *l_tw, int i, float2 w)
{
    l_tw[i] = w;
    if (i < N / 4) l_tw[i + N / 2] = make_float2(-w.x, -w.y);
}

// the whole workgroup copies the table to LDS; caller synchronises before first use
template <int N>
__device__ __forceinline__ void stage_twiddles(const float2 *__restrict__ g_tw, float2 *l_tw)
{
    for (int i = threadIdx.x; i < N / 2; i += blockDim.x) put_twiddle<N>(l_tw, i, g_tw[i]);
}

// the same in two halves for a workgroup of NTHREADS threads, so that a kernel can issue its other global loads between
// them: load() only issues the table loads, store() writes LDS (the first instruction that waits for them)
template <int N, int NTHREADS>
struct TwiddleStage {
    static constexpr int PER = (N / 2 + NTHREADS - 1) / NTHREADS;
    float2 r[PER];
    __device__ __forceinline__ void load(const float2 *__restrict__ g_tw)
    {
#pragma unroll
        for (int i = 0; i < PER; i++) r[i] = g_tw[min((int) threadIdx.x + i * NTHREADS, N / 2 - 1)];
    }
    __device__ __forceinline__ void store(float2 *l_tw) const
    {
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int idx = threadIdx.x + i * NTHREADS;
            if (idx < N / 2) put_twiddle<N>(l_tw, idx, r[i]);
        }
    }
};

template <int N>
__device__ __forceinline__ float2 twiddle(const float2 *l_tw, int i)   // W_N^i, 0 <= i < N
{
    const float2 w = l_tw[i & (N / 2 - 1)];
    return (i & (N / 2)) ? make_float2(-w.x, -w.y) : w;
}

// one radix-4 Stockham stage with butterfly span Ns (1, 4, 16, ...)
template <int M, int Ns, bool INV>
__device__ __forceinline__ void radix4_stage(float2 *buf, const float2 *l_tw, int lane)
{
    constexpr int N = 2 * M, B4 = M / 4, BPL = B4 > kWave ? B4 / kWave : 1, tstep = N / (4 * Ns);
    float2 v[BPL][4], w[BPL][3];
#pragma unroll
    for (int i = 0; i < BPL; i++) {
        const int j = lane + i * kWave;
        if (B4 >= kWave || j < B4) {
#pragma unroll
            for (int r = 0; r < 4; r++) v[i][r] = buf[j + r * B4];
            if constexpr (Ns > 1) {
                const int k = j & (Ns - 1);
                w[i][0] = l_tw[k * tstep];
                w[i][1] = l_tw[2 * k * tstep];
                w[i][2] = l_tw[3 * k * tstep];            // < 3N/4: inside the extended table
            }
        }
    }
    wave_sync();
#pragma unroll
    for (int i = 0; i < BPL; i++) {
        const int j = lane + i * kWave;
        if (B4 >= kWave || j < B4) {
            const int k = j & (Ns - 1);
            float2 v0 = v[i][0], v1 = v[i][1], v2 = v[i][2], v3 = v[i][3];
            if constexpr (Ns > 1) {
                if (INV) { v1 = cmulc(v1, w[i][0]); v2 = cmulc(v2, w[i][1]); v3 = cmulc(v3, w[i][2]); }
                else     { v1 = cmul(v1, w[i][0]);  v2 = cmul(v2, w[i][1]);  v3 = cmul(v3, w[i][2]); }
            }
            const float2 s0 = cadd(v0, v2), d0 = csub(v0, v2), s1 = cadd(v1, v3), e = csub(v1, v3);
            const float2 d1 = INV ? make_float2(-e.y, e.x) : make_float2(e.y, -e.x);   // (+i or -i) * e
            const int j0 = 4 * j - 3 * k;
            buf[j0]          = cadd(s0, s1);
            buf[j0 + Ns]     = cadd(d0, d1);
            buf[j0 + 2 * Ns] = csub(s0, s1);
            buf[j0 + 3 * Ns] = csub(d0, d1);
        }
    }
    wave_sync();
}

// the closing radix-2 stage of a transform whose log2 M is odd (Ns == M/2)
template <int M, bool INV>
__device__ __forceinline__ void radix2_stage(float2 *buf, const float2 *l_tw, int lane)
{
    constexpr int B2 = M / 2, BPL = B2 > kWave ? B2 / kWave : 1;
    float2 v[BPL][2], w[BPL];
#pragma unroll
    for (int i = 0; i < BPL; i++) {
        const int j = lane + i * kWave;
        if (B2 >= kWave || j < B2) {
            v[i][0] = buf[j];
            v[i][1] = buf[j + B2];
            w[i] = l_tw[2 * j];                           // W_N^(2j) = W_M^j, j < M/2
        }
    }
    wave_sync();
#pragma unroll
    for (int i = 0; i < BPL; i++) {
        const int j = lane + i * kWave;
        if (B2 >= kWave || j < B2) {
            const float2 v1 = INV ? cmulc(v[i][1], w[i]) : cmul(v[i][1], w[i]);
            buf[j]      = cadd(v[i][0], v1);
            buf[j + B2] = csub(v[i][0], v1);
        }
    }
    wave_sync();
}

template <int M, int Ns, bool INV>
__device__ __forceinline__ void fft_stages(float2 *buf, const float2 *l_tw, int lane)
{
    if constexpr (Ns * 4 <= M) {
        radix4_stage<M, Ns, INV>(buf, l_tw, lane);
        fft_stages<M, Ns * 4, INV>(buf, l_tw, lane);
    } else if constexpr (Ns < M) {
        radix2_stage<M, INV>(buf, l_tw, lane);
    }
}

// In-place M-point complex FFT of buf[0..M) by the calling wave; l_tw = LDS table of the 2M-point real
// transform (TwLds<2M>::SIZE entries).  The caller must have wave_sync()ed (or crossed a workgroup barrier)
// after filling buf; on return the result is complete and visible to the wave.
template <int M, bool INV>
__device__ __forceinline__ void wave_fft(float2 *buf, const float2 *l_tw, int lane)
{
    fft_stages<M, 1, INV>(buf, l_tw, lane);
}

// B real samples (zero padded to N) -> z[n] = x[2n] + i*x[2n+1], n < M, by one wave.
// All loads (coalesced, clamped instead of predicated) are issued before the first LDS store: one memory latency.
template <int M>
__device__ __forceinline__ void wave_load_block(const float *__restrict__ x, int B, float2 *z, int lane)
{
    constexpr int N = 2 * M, XL = M >= kWave ? M / kWave : 1;   // B <= M samples: XL per lane
    float v[XL];
#pragma unroll
    for (int i = 0; i < XL; i++) v[i] = x[min(lane + i * kWave, B - 1)];
    float *zf = reinterpret_cast<float *>(z);                   // z[n] = (x[2n], x[2n+1]): the block as N floats
#pragma unroll
    for (int i = 0; i < N / kWave; i++) {
        const int n = lane + i * kWave;
        if (i < XL) zf[n] = n < B ? v[i] : 0.0f;
        else zf[n] = 0.0f;
    }
    if constexpr (N < kWave) { if (lane < N) zf[lane] = lane < B ? v[0] : 0.0f; }
    wave_sync();
}

// the same for a block of which only the first `have` samples exist behind x (a signal's last, ragged block; have <= 0: none --
// nothing is read then)
template <int M>
__device__ __forceinline__ void wave_load_block_upto(const float *__restrict__ x, int B, long have, float2 *z, int lane)
{
    if (have >= B) { wave_load_block<M>(x, B, z, lane); return; }
    if (have > 0) { wave_load_block<M>(x, (int) have, z, lane); return; }
    constexpr int N = 2 * M;
    float *zf = reinterpret_cast<float *>(z);
    for (int n = lane; n < N; n += kWave) zf[n] = 0.0f;
    wave_sync();
}

// Forward split of Z = FFT_M(z) into the packed spectrum of the N real samples (unnormalised, as
// performRealOnlyForwardTransform leaves it, reference fp/convolution.cpp:123,144), written through
// `store(bin, value)` (global and/or LDS).  Bin 0 = (DC, Nyquist).  A lane produces bins b and M-b together,
// b = lane + 64 i < M/2; the lane of b = 0 also emits the self-paired bin M/2.  All LDS reads are issued
// before the first store.
template <int M, typename Store>
__device__ __forceinline__ void wave_split_forward(const float2 *Z, const float2 *l_tw, int lane, Store store)
{
    constexpr int H = M / 2, I = H > kWave ? H / kWave : 1;
    float2 p[I], q[I], w[I];
#pragma unroll
    for (int i = 0; i < I; i++) {
        const int b = lane + i * kWave;
        if (H >= kWave || b < H) {
            p[i] = Z[b];
            q[i] = Z[(M - b) & (M - 1)];
            w[i] = l_tw[b];
        }
    }
    const float2 zh = Z[H];
    wave_sync();
#pragma unroll
    for (int i = 0; i < I; i++) {
        const int b = lane + i * kWave;
        if (H >= kWave || b < H) {
            if (b == 0) {
                store(0, make_float2(p[i].x + p[i].y, p[i].x - p[i].y));
                store(H, make_float2(zh.x, -zh.y));
            } else {
                const float2 e = make_float2(0.5f * (p[i].x + q[i].x), 0.5f * (p[i].y - q[i].y));      // (p + conj q)/2
                const float2 o = make_float2(0.5f * (p[i].y + q[i].y), -0.5f * (p[i].x - q[i].x));     // (p - conj q)/(2i)
                const float2 t = cmul(w[i], o);
                store(b, cadd(e, t));
                store(M - b, make_float2(e.x - t.x, t.y - e.y));                                       // conj(e - t)
            }
        }
    }
}

// Inverse split, in place: packed spectrum Y[0..M) -> Z with IFFT_M(Z)[n] = N*(y[2n] + i*y[2n+1]).
// Each lane reads its pairs (b, M-b) and rewrites exactly those (lane 0 also the self-paired bin M/2).
template <int M>
__device__ __forceinline__ void wave_split_inverse(float2 *Y, const float2 *l_tw, int lane)
{
    constexpr int H = M / 2, I = H > kWave ? H / kWave : 1;
    float2 p[I], q[I], w[I];
#pragma unroll
    for (int i = 0; i < I; i++) {
        const int b = lane + i * kWave;
        if (H >= kWave || b < H) {
            p[i] = Y[b];
            q[i] = Y[(M - b) & (M - 1)];
            w[i] = l_tw[b];
        }
    }
    const float2 zh = Y[H];
    wave_sync();
#pragma unroll
    for (int i = 0; i < I; i++) {
        const int b = lane + i * kWave;
        if (H >= kWave || b < H) {
            if (b == 0) {
                Y[0] = make_float2(p[i].x + p[i].y, p[i].x - p[i].y);
                Y[H] = make_float2(2.0f * zh.x, -2.0f * zh.y);
            } else {
                const float2 e = make_float2(p[i].x + q[i].x, p[i].y - q[i].y);        // p + conj q
                const float2 d = make_float2(p[i].x - q[i].x, p[i].y + q[i].y);        // p - conj q
                const float2 o = cmulc(d, w[i]);                                       // * W_N^{-b}
                Y[b] = make_float2(e.x - o.y, e.y + o.x);                              // e + i*o
                Y[M - b] = make_float2(e.x + o.y, o.x - e.y);                          // conj(e) + i*conj(o)
            }
        }
    }
    wave_sync();
}

}  // namespace irb
