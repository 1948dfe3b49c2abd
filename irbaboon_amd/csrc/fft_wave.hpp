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

// Packed fp32 (v_pk_*_f32: both halves of a 64-bit register pair in one instruction, the swizzles and sign flips of complex
// arithmetic in the op_sel / neg modifiers): a lone wave issues one vector instruction per ~4-5 cycles whatever it does, so the
// butterflies below are written in HALF the instructions of the scalar form (a complex product 2 instead of 4, a complex
// add 1 instead of 2, a multiplication by +-i none at all).  Same roundings as the scalar fma sequence.
typedef float v2f_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f_t pk(float2 a) { v2f_t r = {a.x, a.y}; return r; }
__device__ __forceinline__ float2 unpk(v2f_t a) { return make_float2(a.x, a.y); }
__device__ __forceinline__ v2f_t lds_ld(const float2 *p) { return *reinterpret_cast<const v2f_t *>(p); }
__device__ __forceinline__ void lds_st(float2 *p, v2f_t v) { *reinterpret_cast<v2f_t *>(p) = v; }
// a * w
__device__ __forceinline__ v2f_t pk_cmul(v2f_t a, v2f_t w)
{
    v2f_t t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));                                   // (a.x w.x,  a.x w.y)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(t) : "v"(a), "v"(w));            // (-a.y w.y, a.y w.x) +
    return t;
}
// a * conj(w)
__device__ __forceinline__ v2f_t pk_cmulc(v2f_t a, v2f_t w)
{
    v2f_t t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));                      // (a.x w.x, -a.x w.y)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "+v"(t) : "v"(a), "v"(w));                           // (a.y w.y,  a.y w.x) +
    return t;
}
// a + (e.y, -e.x) = a - i e          and          a + (-e.y, e.x) = a + i e
__device__ __forceinline__ v2f_t pk_add_mi(v2f_t a, v2f_t e)
{
    v2f_t r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(e));
    return r;
}
__device__ __forceinline__ v2f_t pk_add_pi(v2f_t a, v2f_t e)
{
    v2f_t r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(e));
    return r;
}
// a + conj(b), a - conj(b), conj(a - b)
__device__ __forceinline__ v2f_t pk_add_conj(v2f_t a, v2f_t b)
{
    v2f_t r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f_t pk_sub_conj(v2f_t a, v2f_t b)
{
    v2f_t r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f_t pk_conj_sub(v2f_t a, v2f_t b)
{
    v2f_t r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));                                      // (a.x - b.x, b.y - a.y)
    return r;
}

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
    v2f_t v[BPL][4], w[BPL][3];
#pragma unroll
    for (int i = 0; i < BPL; i++) {
        const int j = lane + i * kWave;
        if (B4 >= kWave || j < B4) {
#pragma unroll
            for (int r = 0; r < 4; r++) v[i][r] = lds_ld(buf + j + r * B4);
            if constexpr (Ns > 1) {
                const int k = j & (Ns - 1);
                w[i][0] = lds_ld(l_tw + k * tstep);
                w[i][1] = lds_ld(l_tw + 2 * k * tstep);
                w[i][2] = lds_ld(l_tw + 3 * k * tstep);   // < 3N/4: inside the extended table
            }
        }
    }
    wave_sync();
#pragma unroll
    for (int i = 0; i < BPL; i++) {
        const int j = lane + i * kWave;
        if (B4 >= kWave || j < B4) {
            const int k = j & (Ns - 1);
            v2f_t v0 = v[i][0], v1 = v[i][1], v2 = v[i][2], v3 = v[i][3];
            if constexpr (Ns > 1) {
                if (INV) { v1 = pk_cmulc(v1, w[i][0]); v2 = pk_cmulc(v2, w[i][1]); v3 = pk_cmulc(v3, w[i][2]); }
                else     { v1 = pk_cmul(v1, w[i][0]);  v2 = pk_cmul(v2, w[i][1]);  v3 = pk_cmul(v3, w[i][2]); }
            }
            const v2f_t s0 = v0 + v2, d0 = v0 - v2, s1 = v1 + v3, e = v1 - v3;
            const int j0 = 4 * j - 3 * k;
            lds_st(buf + j0, s0 + s1);
            lds_st(buf + j0 + Ns, INV ? pk_add_pi(d0, e) : pk_add_mi(d0, e));         // d0 + (+i or -i) e
            lds_st(buf + j0 + 2 * Ns, s0 - s1);
            lds_st(buf + j0 + 3 * Ns, INV ? pk_add_mi(d0, e) : pk_add_pi(d0, e));     // d0 - (+i or -i) e
        }
    }
    wave_sync();
}

// the closing radix-2 stage of a transform whose log2 M is odd (Ns == M/2)
template <int M, bool INV>
__device__ __forceinline__ void radix2_stage(float2 *buf, const float2 *l_tw, int lane)
{
    constexpr int B2 = M / 2, BPL = B2 > kWave ? B2 / kWave : 1;
    v2f_t v[BPL][2], w[BPL];
#pragma unroll
    for (int i = 0; i < BPL; i++) {
        const int j = lane + i * kWave;
        if (B2 >= kWave || j < B2) {
            v[i][0] = lds_ld(buf + j);
            v[i][1] = lds_ld(buf + j + B2);
            w[i] = lds_ld(l_tw + 2 * j);                  // W_N^(2j) = W_M^j, j < M/2
        }
    }
    wave_sync();
#pragma unroll
    for (int i = 0; i < BPL; i++) {
        const int j = lane + i * kWave;
        if (B2 >= kWave || j < B2) {
            const v2f_t v1 = INV ? pk_cmulc(v[i][1], w[i]) : pk_cmul(v[i][1], w[i]);
            lds_st(buf + j, v[i][0] + v1);
            lds_st(buf + j + B2, v[i][0] - v1);
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
    v2f_t p[I], q[I], w[I];
#pragma unroll
    for (int i = 0; i < I; i++) {
        const int b = lane + i * kWave;
        if (H >= kWave || b < H) {
            p[i] = lds_ld(Z + b);
            q[i] = lds_ld(Z + ((M - b) & (M - 1)));
            w[i] = lds_ld(l_tw + b);
        }
    }
    const float2 zh = Z[H];
    wave_sync();
    const v2f_t half = {0.5f, 0.5f};
#pragma unroll
    for (int i = 0; i < I; i++) {
        const int b = lane + i * kWave;
        if (H >= kWave || b < H) {
            if (b == 0) {
                store(0, make_float2(p[i].x + p[i].y, p[i].x - p[i].y));
                store(H, make_float2(zh.x, -zh.y));
            } else {
                const v2f_t e = pk_add_conj(p[i], q[i]) * half;                // (p + conj q) / 2
                const v2f_t d = pk_sub_conj(p[i], q[i]) * half;                // (p - conj q) / 2;  o = -i d
                v2f_t t;                                                       // t = w o = (w.x d.y + w.y d.x,  w.y d.y - w.x d.x)
                asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(t) : "v"(w[i]), "v"(d));       // (w.x d.y, -w.x d.x)
                asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(t) : "v"(w[i]), "v"(d));            // (w.y d.x,  w.y d.y) +
                store(b, unpk(e + t));
                store(M - b, unpk(pk_conj_sub(e, t)));                         // conj(e - t)
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
    v2f_t p[I], q[I], w[I];
#pragma unroll
    for (int i = 0; i < I; i++) {
        const int b = lane + i * kWave;
        if (H >= kWave || b < H) {
            p[i] = lds_ld(Y + b);
            q[i] = lds_ld(Y + ((M - b) & (M - 1)));
            w[i] = lds_ld(l_tw + b);
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
                const v2f_t e = pk_add_conj(p[i], q[i]);                       // p + conj q
                const v2f_t d = pk_sub_conj(p[i], q[i]);                       // p - conj q
                const v2f_t o = pk_cmulc(d, w[i]);                             // * W_N^{-b}
                lds_st(Y + b, pk_add_pi(e, o));                                // e + i o
                v2f_t r;                                                       // conj(e) + i conj(o) = (e.x + o.y, o.x - e.y)
                asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[1,0]" : "=v"(r) : "v"(e), "v"(o));
                lds_st(Y + (M - b), r);
            }
        }
    }
    wave_sync();
}

}  // namespace irb
