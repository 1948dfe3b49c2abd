// fft_wave.hpp -- one-wave (64 lanes) complex FFT in LDS for gfx950.
//
// An M-point transform (M = N/2 of the real FFT, 16..2048) is owned by ONE wavefront: no workgroup
// barrier between stages, so a 256-thread workgroup runs four independent transforms.  Stockham
// autosort, radix-4 stages plus one radix-2 stage when log2 M is odd, done IN PLACE: a wave executes in
// lockstep and the LDS serves one wave's instructions in order, so when every lane first reads all the
// points of its butterflies and only then writes, no point is overwritten before it was read.  A
// wavefront-scope fence + wave_barrier between the phases stops the compiler from moving a lane's own
// LDS accesses across them.  Twiddles come from a half table W_N^j, j < N/2, staged once per workgroup
// in LDS (W_N^{j+N/2} = -W_N^j covers the 3N/4 range the radix-4 stages need).
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

// the whole workgroup copies the half twiddle table to LDS; caller synchronises before first use
template <int N>
__device__ __forceinline__ void stage_twiddles(const float2 *__restrict__ g_tw, float2 *l_tw)
{
    for (int i = threadIdx.x; i < N / 2; i += blockDim.x) l_tw[i] = g_tw[i];
}

template <int N>
__device__ __forceinline__ float2 twiddle(const float2 *l_tw, int i)   // W_N^i, 0 <= i < N
{
    const float2 w = l_tw[i & (N / 2 - 1)];
    return (i & (N / 2)) ? make_float2(-w.x, -w.y) : w;
}

// In-place M-point complex FFT of buf[0..M) by the calling wave.  The caller must have wave_sync()ed
// after filling buf; on return the result is complete and visible to the wave.
template <int M, bool INV>
__device__ __forceinline__ void wave_fft(float2 *buf, const float2 *l_tw, int lane)
{
    constexpr int N = 2 * M;
    constexpr int B4 = M / 4;                                  // radix-4 butterflies per stage
    constexpr int BPL4 = B4 > kWave ? B4 / kWave : 1;          // per lane
    int Ns = 1;
#pragma unroll 1
    for (; Ns * 4 <= M; Ns *= 4) {
        const int tstep = N / (Ns * 4);
        float2 v[BPL4][4];
#pragma unroll
        for (int i = 0; i < BPL4; i++) {
            const int j = lane + i * kWave;
            if (j < B4) {
#pragma unroll
                for (int r = 0; r < 4; r++) v[i][r] = buf[j + r * B4];
            }
        }
        wave_sync();
#pragma unroll
        for (int i = 0; i < BPL4; i++) {
            const int j = lane + i * kWave;
            if (j < B4) {
                const int k = j & (Ns - 1);
                float2 v0 = v[i][0], v1 = v[i][1], v2 = v[i][2], v3 = v[i][3];
                if (Ns > 1) {
                    const float2 w1 = twiddle<N>(l_tw, k * tstep), w2 = twiddle<N>(l_tw, 2 * k * tstep),
                                 w3 = twiddle<N>(l_tw, 3 * k * tstep);
                    if (INV) { v1 = cmulc(v1, w1); v2 = cmulc(v2, w2); v3 = cmulc(v3, w3); }
                    else     { v1 = cmul(v1, w1);  v2 = cmul(v2, w2);  v3 = cmul(v3, w3); }
                }
                const float2 s0 = cadd(v0, v2), d0 = csub(v0, v2), s1 = cadd(v1, v3), e = csub(v1, v3);
                const float2 d1 = INV ? make_float2(-e.y, e.x) : make_float2(e.y, -e.x);   // (+i or -i) * e
                const int j0 = ((j - k) << 2) + k;
                buf[j0]          = cadd(s0, s1);
                buf[j0 + Ns]     = cadd(d0, d1);
                buf[j0 + 2 * Ns] = csub(s0, s1);
                buf[j0 + 3 * Ns] = csub(d0, d1);
            }
        }
        wave_sync();
    }
    if (Ns < M) {   // one radix-2 stage, Ns == M/2
        constexpr int B2 = M / 2;
        constexpr int BPL2 = B2 > kWave ? B2 / kWave : 1;
        const int tstep = N / (Ns * 2);
        float2 v[BPL2][2];
#pragma unroll
        for (int i = 0; i < BPL2; i++) {
            const int j = lane + i * kWave;
            if (j < B2) { v[i][0] = buf[j]; v[i][1] = buf[j + B2]; }
        }
        wave_sync();
#pragma unroll
        for (int i = 0; i < BPL2; i++) {
            const int j = lane + i * kWave;
            if (j < B2) {
                const int k = j & (Ns - 1);
                const float2 w = twiddle<N>(l_tw, k * tstep);
                const float2 v1 = INV ? cmulc(v[i][1], w) : cmul(v[i][1], w);
                const int j0 = ((j - k) << 1) + k;
                buf[j0]      = cadd(v[i][0], v1);
                buf[j0 + Ns] = csub(v[i][0], v1);
            }
        }
        wave_sync();
    }
}

// B real samples (zero padded to N) -> z[n] = x[2n] + i*x[2n+1], n < M, by one wave.
template <int M>
__device__ __forceinline__ void wave_load_block(const float *__restrict__ x, int B, float2 *z, int lane)
{
    for (int n = lane; n < M; n += kWave) {
        const int i = 2 * n;
        z[n] = make_float2(i < B ? x[i] : 0.0f, i + 1 < B ? x[i + 1] : 0.0f);
    }
    wave_sync();
}

// Forward split of Z = FFT_M(z) into the packed spectrum of the N real samples (unnormalised, as
// performRealOnlyForwardTransform leaves it, reference fp/convolution.cpp:123,144), written to `dst`
// (global or LDS).  Bin 0 = (DC, Nyquist).  A lane produces bins b and M-b together.
template <int M, typename Store>
__device__ __forceinline__ void wave_split_forward(const float2 *Z, const float2 *l_tw, int lane, Store store)
{
    constexpr int N = 2 * M;
    for (int b = lane; b <= M / 2; b += kWave) {
        if (b == 0) {
            store(0, make_float2(Z[0].x + Z[0].y, Z[0].x - Z[0].y));
        } else {
            const float2 p = Z[b], q = Z[M - b];
            const float2 e = make_float2(0.5f * (p.x + q.x), 0.5f * (p.y - q.y));      // (p + conj q)/2
            const float2 o = make_float2(0.5f * (p.y + q.y), -0.5f * (p.x - q.x));     // (p - conj q)/(2i)
            const float2 t = cmul(twiddle<N>(l_tw, b), o);
            store(b, cadd(e, t));
            if (b != M - b) store(M - b, make_float2(e.x - t.x, t.y - e.y));           // conj(e - t)
        }
    }
}

// Inverse split, in place: packed spectrum Y[0..M) -> Z with IFFT_M(Z)[n] = N*(y[2n] + i*y[2n+1]).
// Each lane reads its pair (b, M-b) and rewrites exactly that pair, so no cross-lane hazard.
template <int M>
__device__ __forceinline__ void wave_split_inverse(float2 *Y, const float2 *l_tw, int lane)
{
    constexpr int N = 2 * M;
    for (int b = lane; b <= M / 2; b += kWave) {
        if (b == 0) {
            const float2 y0 = Y[0];
            Y[0] = make_float2(y0.x + y0.y, y0.x - y0.y);
        } else {
            const float2 p = Y[b], q = Y[M - b];
            const float2 e = make_float2(p.x + q.x, p.y - q.y);        // p + conj q
            const float2 d = make_float2(p.x - q.x, p.y + q.y);        // p - conj q
            const float2 o = cmulc(d, twiddle<N>(l_tw, b));            // * W_N^{-b}
            Y[b] = make_float2(e.x - o.y, e.y + o.x);                  // e + i*o
            if (b != M - b) Y[M - b] = make_float2(e.x + o.y, o.x - e.y);   // conj(e) + i*conj(o)
        }
    }
    wave_sync();
}

}  // namespace irb
