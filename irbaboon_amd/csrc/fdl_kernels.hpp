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

namespace irb {

constexpr int kThreads = 256;

struct StepArgs {
    const float2 *H;            // [n_sets][P][M]  (P here = allocated rows per set, a multiple of 8)
    float2       *X;            // [S][R][M]
    float        *ovl;          // [S][B]
    const float2 *tw;           // [N]
    const int    *set_of_stream;// [S]
    const int    *np_of_set;    // [n_sets]: partitions in use (<= P)
    const float  *in;           // block j of stream s at in + s*in_stride + j*B
    float        *out;
    float2       *part;         // [S][splits][M] partial sums (split path)
    long          in_stride, out_stride;
    int           B, P, R, head, splits;
    // time-tiled path (several blocks per call)
    float        *ovl_next;     // overlap written by the last block of the call (double buffered)
    int           T, nq;        // blocks in this call; partial rows per output (splits * stripes)
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) { /* a * conj(b) */ return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// ---------------------------------------------------------------------------------------------
// M-point complex FFT in LDS: autosort Stockham, radix-4 stages then one radix-2 stage if log2 M
// is odd.  One thread per butterfly, ping-pong between a and b; returns the buffer holding the
// result.  Callers must __syncthreads() after filling `a`; the function ends with a barrier.
// (Index arithmetic is modelled and checked on the CPU by tests/kernel_model.py.)
// ---------------------------------------------------------------------------------------------
template <int M, bool INV>
__device__ __forceinline__ float2 *fft_lds(float2 *a, float2 *b, const float2 *__restrict__ tw)
{
    constexpr int N = 2 * M;
    int Ns = 1;
#pragma unroll 1
    for (; Ns * 4 <= M; Ns *= 4) {
        const int stride = M / 4;
        const int tstep = N / (Ns * 4);
        for (int j = threadIdx.x; j < stride; j += kThreads) {
            const int k = j & (Ns - 1);
            float2 v0 = a[j], v1 = a[j + stride], v2 = a[j + 2 * stride], v3 = a[j + 3 * stride];
            if (Ns > 1) {
                const float2 w1 = tw[k * tstep], w2 = tw[2 * k * tstep], w3 = tw[3 * k * tstep];
                if (INV) { v1 = cmulc(v1, w1); v2 = cmulc(v2, w2); v3 = cmulc(v3, w3); }
                else     { v1 = cmul(v1, w1);  v2 = cmul(v2, w2);  v3 = cmul(v3, w3); }
            }
            const float2 s0 = cadd(v0, v2), d0 = csub(v0, v2), s1 = cadd(v1, v3), e = csub(v1, v3);
            const float2 d1 = INV ? make_float2(-e.y, e.x) : make_float2(e.y, -e.x);   // (+i or -i) * e
            const int j0 = ((j - k) << 2) + k;
            b[j0]          = cadd(s0, s1);
            b[j0 + Ns]     = cadd(d0, d1);
            b[j0 + 2 * Ns] = csub(s0, s1);
            b[j0 + 3 * Ns] = csub(d0, d1);
        }
        __syncthreads();
        float2 *t = a; a = b; b = t;
    }
    if (Ns < M) {   // one radix-2 stage, Ns == M/2
        const int stride = M / 2;
        const int tstep = N / (Ns * 2);
        for (int j = threadIdx.x; j < stride; j += kThreads) {
            const int k = j & (Ns - 1);
            float2 v0 = a[j], v1 = a[j + stride];
            const float2 w = tw[k * tstep];
            v1 = INV ? cmulc(v1, w) : cmul(v1, w);
            const int j0 = ((j - k) << 1) + k;
            b[j0]      = cadd(v0, v1);
            b[j0 + Ns] = csub(v0, v1);
        }
        __syncthreads();
        float2 *t = a; a = b; b = t;
    }
    return a;
}

// B real samples (zero padded to N) -> z[n] = x[2n] + i*x[2n+1], n < M.
template <int M>
__device__ __forceinline__ void load_block(const float *__restrict__ x, int B, float2 *z)
{
    for (int n = threadIdx.x; n < M; n += kThreads) {
        const int i = 2 * n;
        z[n] = make_float2(i < B ? x[i] : 0.0f, i + 1 < B ? x[i + 1] : 0.0f);
    }
}

// Forward split: Z = FFT_M(z) -> packed spectrum of the N real samples (unnormalised, as
// performRealOnlyForwardTransform leaves it; reference fp/convolution.cpp:123,144).
template <int M>
__device__ __forceinline__ void split_forward(const float2 *Z, float2 *Xp, const float2 *__restrict__ tw)
{
    for (int b = threadIdx.x; b < M; b += kThreads) {
        if (b == 0) {
            Xp[0] = make_float2(Z[0].x + Z[0].y, Z[0].x - Z[0].y);
        } else {
            const float2 p = Z[b], q = Z[M - b];
            const float2 e = make_float2(0.5f * (p.x + q.x), 0.5f * (p.y - q.y));      // (p + conj q)/2
            const float2 o = make_float2(0.5f * (p.y + q.y), -0.5f * (p.x - q.x));     // (p - conj q)/(2i)
            Xp[b] = cadd(e, cmul(tw[b], o));
        }
    }
}

// Inverse split: packed spectrum Y -> Z with IFFT_M(Z)[n] = N * (y[2n] + i*y[2n+1]).
template <int M>
__device__ __forceinline__ void split_inverse(const float2 *Y, float2 *Z, const float2 *__restrict__ tw)
{
    for (int b = threadIdx.x; b < M; b += kThreads) {
        if (b == 0) {
            Z[0] = make_float2(Y[0].x + Y[0].y, Y[0].x - Y[0].y);
        } else {
            const float2 p = Y[b], q = Y[M - b];
            const float2 e = make_float2(p.x + q.x, p.y - q.y);        // p + conj q
            const float2 d = make_float2(p.x - q.x, p.y + q.y);        // p - conj q
            const float2 o = cmulc(d, tw[b]);                          // * W_N^{-b}
            Z[b] = make_float2(e.x - o.y, e.y + o.x);                  // e + i*o
        }
    }
}

// Overlap-add (reference fp/convolution.cpp:209-213): y = N real samples, unscaled, in LDS.
__device__ __forceinline__ void overlap_add(const float *y, float scale, int B, float *__restrict__ ovl, float *__restrict__ out)
{
    for (int i = threadIdx.x; i < B; i += kThreads) {
        const float o = y[i] * scale + ovl[i];
        ovl[i] = y[B + i] * scale;
        out[i] = o;
    }
}

// ---------------------------------------------------------------------------------------------
// Multiply-accumulate over partitions [kb, ke) of one stream (reference fp/convolution.cpp:171-202):
// acc += X[(head-k) mod R] * H[k].  Each lane: 16-byte loads of two complex bins from both rows.
// dc accumulates the two real products of packed bin 0 (used by the lane that owns column 0).
// ---------------------------------------------------------------------------------------------
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

// Combine the stripes of a workgroup: part is LDS [STRIPES][M]; result Y[M] (packed) in `Y`.
template <int N>
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
    __syncthreads();
    for (int b = threadIdx.x; b < G::M; b += kThreads) {
        float2 s = part[b];
#pragma unroll
        for (int t = 1; t < G::STRIPES; t++) s = cadd(s, part[t * G::M + b]);
        Y[b] = s;
    }
    __syncthreads();
}

template <int N>
struct Lds {
    static constexpr int M = N / 2;
    float2 a[M];
    float2 b[M];
    float2 part[MacGeom<N>::STRIPES * M];
};

// ---------------------------------------------------------------------------------------------
// k_step_fused: one workgroup owns one stream for one block step:
//   input block -> FFT -> ring slot `head`; MAC over all partitions; IFFT; overlap-add; output.
// grid = S.  The dominant kernel when many streams are batched (HBM-bound, 8*P*N + 16*B bytes).
// ---------------------------------------------------------------------------------------------
template <int N, int U = 4, bool NT = false>
__global__ __launch_bounds__(kThreads) void k_step_fused(StepArgs a)
{
    using G = MacGeom<N>;
    constexpr int M = N / 2;
    __shared__ Lds<N> lds;
    const int s = blockIdx.x;
    const int set = a.set_of_stream[s];
    const int np = a.np_of_set[set];
    const float4 *H4 = reinterpret_cast<const float4 *>(a.H + (size_t) set * a.P * M);
    float2 *Xs = a.X + (size_t) s * a.R * M;
    const float4 *X4 = reinterpret_cast<const float4 *>(Xs);

    // older partitions first: their loads do not depend on the new block
    float4 acc[G::CPT];
#pragma unroll
    for (int c = 0; c < G::CPT; c++) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 dc = make_float2(0.f, 0.f);
    mac_range<N, U, NT>(H4, X4, 1, np, a.head, a.R, acc, dc);

    // new block: FFT, store into the ring, and its product with partition 0
    load_block<M>(a.in + (size_t) s * a.in_stride, a.B, lds.a);
    __syncthreads();
    float2 *Z = fft_lds<M, false>(lds.a, lds.b, a.tw);
    float2 *Xn = (Z == lds.a) ? lds.b : lds.a;
    split_forward<M>(Z, Xn, a.tw);
    __syncthreads();
    for (int b = threadIdx.x; b < M; b += kThreads) Xs[(size_t) a.head * M + b] = Xn[b];
    if (np > 0 && threadIdx.x < G::LW) {
        const int col = threadIdx.x;
#pragma unroll
        for (int c = 0; c < G::CPT; c++) {
            const float4 h = H4[col + c * G::LW];
            const float4 x = reinterpret_cast<const float4 *>(Xn)[col + c * G::LW];
            if (c == 0) { dc.x += x.x * h.x; dc.y += x.y * h.y; }
            acc[c].x += x.x * h.x - x.y * h.y;
            acc[c].y += x.x * h.y + x.y * h.x;
            acc[c].z += x.z * h.z - x.w * h.w;
            acc[c].w += x.z * h.w + x.w * h.z;
        }
    }
    __syncthreads();

    float2 *Y = lds.a;
    reduce_stripes<N>(acc, dc, lds.part, Y);
    split_inverse<M>(Y, lds.b, a.tw);
    __syncthreads();
    float2 *z = fft_lds<M, true>(lds.b, lds.a, a.tw);
    overlap_add(reinterpret_cast<const float *>(z), 1.0f / (float) N, a.B,
                a.ovl + (size_t) s * a.B, a.out + (size_t) s * a.out_stride);
}

// ---------------------------------------------------------------------------------------------
// Split path, for few streams with long IRs (one workgroup per stream cannot fill the chip):
//   k_push_fft        grid = S            new block -> FFT -> ring slot `head`
//   k_cmac_partial    grid = S*splits     partial sums over a contiguous partition range
//   k_reduce_ifft_ola grid = S            sum of the partials -> IFFT -> overlap-add
// ---------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(kThreads) void k_push_fft(StepArgs a)
{
    constexpr int M = N / 2;
    __shared__ float2 la[M], lb[M];
    const int s = blockIdx.x;
    load_block<M>(a.in + (size_t) s * a.in_stride, a.B, la);
    __syncthreads();
    float2 *Z = fft_lds<M, false>(la, lb, a.tw);
    float2 *Xn = (Z == la) ? lb : la;
    split_forward<M>(Z, Xn, a.tw);
    __syncthreads();
    float2 *dst = a.X + ((size_t) s * a.R + a.head) * M;
    for (int b = threadIdx.x; b < M; b += kThreads) dst[b] = Xn[b];
}

template <int N>
__global__ __launch_bounds__(kThreads) void k_cmac_partial(StepArgs a)
{
    using G = MacGeom<N>;
    constexpr int M = N / 2;
    __shared__ float2 part[G::STRIPES * M];
    __shared__ float2 Y[M];
    const int j = blockIdx.x % a.splits, s = blockIdx.x / a.splits;
    const int set = a.set_of_stream[s];
    const int np = a.np_of_set[set];
    const int kb = (int) ((long) np * j / a.splits), ke = (int) ((long) np * (j + 1) / a.splits);
    const float4 *H4 = reinterpret_cast<const float4 *>(a.H + (size_t) set * a.P * M);
    const float4 *X4 = reinterpret_cast<const float4 *>(a.X + (size_t) s * a.R * M);
    float4 acc[G::CPT];
#pragma unroll
    for (int c = 0; c < G::CPT; c++) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 dc = make_float2(0.f, 0.f);
    mac_range<N, 4>(H4, X4, kb, ke, a.head, a.R, acc, dc);
    reduce_stripes<N>(acc, dc, part, Y);
    float2 *dst = a.part + ((size_t) s * a.splits + j) * M;
    for (int b = threadIdx.x; b < M; b += kThreads) dst[b] = Y[b];
}

template <int N>
__global__ __launch_bounds__(kThreads) void k_reduce_ifft_ola(StepArgs a)
{
    constexpr int M = N / 2;
    __shared__ float2 la[M], lb[M];
    const int s = blockIdx.x;
    const float2 *src = a.part + (size_t) s * a.splits * M;
    for (int b = threadIdx.x; b < M; b += kThreads) {
        float2 v = src[b];
        for (int j = 1; j < a.splits; j++) v = cadd(v, src[(size_t) j * M + b]);
        la[b] = v;
    }
    __syncthreads();
    split_inverse<M>(la, lb, a.tw);
    __syncthreads();
    float2 *z = fft_lds<M, true>(lb, la, a.tw);
    overlap_add(reinterpret_cast<const float *>(z), 1.0f / (float) N, a.B,
                a.ovl + (size_t) s * a.B, a.out + (size_t) s * a.out_stride);
}


// ---------------------------------------------------------------------------------------------
// Time-tiled path: a call that carries T > 1 new blocks per stream (host blocks larger than B, or the
// whole signal in the offline function).  All T input spectra are pushed first, then every output
//   Y_{j}[b] = sum_k X_{t0+j-k}[b] * H_k[b],  j = 0..T-1
// is independent, and TT consecutive outputs share their H loads: a lane keeps a sliding window of
// 2*TT-1 X rows and TT accumulators in registers (per outer step: TT new X rows + TT H rows feed
// TT*TT complex multiply-adds per bin).  HBM/L2 traffic per output drops by ~TT versus streaming.
//   k_push_fft_batch   grid = S*T (block index fastest)
//   k_cmac_tiled       grid = S*ceil(T/TT)*splits: stripes own contiguous partition sub-ranges and
//                      write their partial spectra straight from registers: part[s][j][q][M]
//   k_ifft_ola_batch   grid = S*T: sums the nq partials of block j and of block j-1 (for its tail)
// ---------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(kThreads) void k_push_fft_batch(StepArgs a)
{
    constexpr int M = N / 2;
    __shared__ float2 la[M], lb[M];
    const int j = blockIdx.x % a.T, s = blockIdx.x / a.T;
    load_block<M>(a.in + (size_t) s * a.in_stride + (size_t) j * a.B, a.B, la);
    __syncthreads();
    float2 *Z = fft_lds<M, false>(la, lb, a.tw);
    float2 *Xn = (Z == la) ? lb : la;
    split_forward<M>(Z, Xn, a.tw);
    __syncthreads();
    int slot = a.head + j;
    if (slot >= a.R) slot -= a.R;
    float2 *dst = a.X + ((size_t) s * a.R + slot) * M;
    for (int b = threadIdx.x; b < M; b += kThreads) dst[b] = Xn[b];
}

// XCD-aware workgroup id: hardware deals consecutive workgroups round-robin over the 8 XCDs (each with
// its own 4 MiB L2); this bijection gives every XCD one contiguous range of ids instead, so workgroups
// that share X/H rows (neighbouring time tiles) also share an L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_contiguous_id(int bid, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

__device__ __forceinline__ int ring_slot(int head, int d, int R)
{
    int sl = head + d;
    if (sl < 0) sl += R;
    if (sl >= R) sl -= R;
    return sl;
}

// one complex-pair multiply-accumulate; `first_is_packed` lanes treat the first complex of the float4
// as packed bin 0 = two real bins: p = 0, q = h.y there (see k_cmac_tiled)
__device__ __forceinline__ void cmac2(float4 &acc, const float4 x, const float4 h, float p, float q)
{
    acc.x = fmaf(x.x, h.x, acc.x);  acc.x = fmaf(-x.y, p, acc.x);
    acc.y = fmaf(x.x, p, acc.y);    acc.y = fmaf(x.y, q, acc.y);
    acc.z = fmaf(x.z, h.z, acc.z);  acc.z = fmaf(-x.w, h.w, acc.z);
    acc.w = fmaf(x.z, h.w, acc.w);  acc.w = fmaf(x.w, h.z, acc.w);
}

template <int N, int TT>
__global__ __launch_bounds__(kThreads) void k_cmac_tiled(StepArgs a)
{
    using G = MacGeom<N>;
    constexpr int M = N / 2;
    const int ntiles = (a.T + TT - 1) / TT;
    const int wg = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int split = wg % a.splits, j0 = ((wg / a.splits) % ntiles) * TT, s = wg / (a.splits * ntiles);
    const int set = a.set_of_stream[s];
    const int np = a.np_of_set[set];
    const int col0 = threadIdx.x % G::LW, stripe = threadIdx.x / G::LW;
    const int q = split * G::STRIPES + stripe;
    // whole groups of TT partitions per worker (worker = one stripe of one split).  The H rows between
    // np and the next multiple of TT are kept at zero by the host, so there is no remainder loop.
    const int nG = (np + TT - 1) / TT, nW = a.splits * G::STRIPES;
    const int sb = (int) ((long) nG * q / nW) * TT, se = (int) ((long) nG * (q + 1) / nW) * TT;
    const float4 *H4 = reinterpret_cast<const float4 *>(a.H + (size_t) set * a.P * M);
    const float4 *X4 = reinterpret_cast<const float4 *>(a.X + (size_t) s * a.R * M);

#pragma unroll 1
    for (int c = 0; c < G::CPT; c++) {
        const int col = col0 + c * G::LW;
        const bool packed = (col == 0);
        float4 acc[TT];
#pragma unroll
        for (int j = 0; j < TT; j++) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);

        int k0 = sb;
        if (k0 < se) {
            // window xr[i] = X_{t0 + base + i}, base = j0 - k0 - (TT-1)
            float4 xr[2 * TT - 1];
#pragma unroll
            for (int i = TT; i < 2 * TT - 1; i++)
                xr[i] = X4[(size_t) ring_slot(a.head, j0 - k0 - (TT - 1) + i, a.R) * G::L + col];
            for (; k0 < se; k0 += TT) {
                float4 h[TT];
#pragma unroll
                for (int u = 0; u < TT; u++) h[u] = H4[(size_t) (k0 + u) * G::L + col];
#pragma unroll
                for (int i = 0; i < TT; i++)
                    xr[i] = X4[(size_t) ring_slot(a.head, j0 - k0 - (TT - 1) + i, a.R) * G::L + col];
#pragma unroll
                for (int u = 0; u < TT; u++) {
                    const float p = packed ? 0.0f : h[u].y;
                    const float qq = packed ? h[u].y : h[u].x;
#pragma unroll
                    for (int j = 0; j < TT; j++) cmac2(acc[j], xr[j - u + TT - 1], h[u], p, qq);
                }
#pragma unroll
                for (int i = 2 * TT - 2; i >= TT; i--) xr[i] = xr[i - TT];   // next step: base -= TT
            }
        }
#pragma unroll
        for (int j = 0; j < TT; j++) {
            if (j0 + j < a.T) {
                float4 *dst = reinterpret_cast<float4 *>(a.part + (((size_t) s * a.T + j0 + j) * a.nq + q) * M);
                dst[col] = acc[j];
            }
        }
    }
}

template <int N>
__global__ __launch_bounds__(kThreads) void k_ifft_ola_batch(StepArgs a)
{
    constexpr int M = N / 2;
    __shared__ float2 la[M], lb[M];
    __shared__ float tail[M];             // B <= N/2 = M
    const int j = blockIdx.x % a.T, s = blockIdx.x / a.T;
    const float scale = 1.0f / (float) N;
    // tail of the previous block of this call, or the saved overlap for the first block
    if (j > 0) {
        const float2 *src = a.part + ((size_t) s * a.T + (j - 1)) * a.nq * M;
        for (int b = threadIdx.x; b < M; b += kThreads) {
            float2 v = src[b];
            for (int q = 1; q < a.nq; q++) v = cadd(v, src[(size_t) q * M + b]);
            la[b] = v;
        }
        __syncthreads();
        split_inverse<M>(la, lb, a.tw);
        __syncthreads();
        const float *y = reinterpret_cast<const float *>(fft_lds<M, true>(lb, la, a.tw));
        for (int i = threadIdx.x; i < a.B; i += kThreads) tail[i] = y[a.B + i] * scale;
    } else {
        const float *ov = a.ovl + (size_t) s * a.B;
        for (int i = threadIdx.x; i < a.B; i += kThreads) tail[i] = ov[i];
    }
    __syncthreads();
    const float2 *src = a.part + ((size_t) s * a.T + j) * a.nq * M;
    for (int b = threadIdx.x; b < M; b += kThreads) {
        float2 v = src[b];
        for (int q = 1; q < a.nq; q++) v = cadd(v, src[(size_t) q * M + b]);
        la[b] = v;
    }
    __syncthreads();
    split_inverse<M>(la, lb, a.tw);
    __syncthreads();
    const float *y = reinterpret_cast<const float *>(fft_lds<M, true>(lb, la, a.tw));
    float *out = a.out + (size_t) s * a.out_stride + (size_t) j * a.B;
    for (int i = threadIdx.x; i < a.B; i += kThreads) out[i] = y[i] * scale + tail[i];
    if (j == a.T - 1) {
        float *on = a.ovl_next + (size_t) s * a.B;
        for (int i = threadIdx.x; i < a.B; i += kThreads) on[i] = y[a.B + i] * scale;
    }
}

// ---------------------------------------------------------------------------------------------
// k_ir_fft: IR partitions -> spectra (reference fp/convolution.cpp:106-125).  grid = (P, n_rows):
// row r of `ir` (time domain, [n_rows][P*B], zero padded) -> H set first_set + r.
// ---------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(kThreads) void k_ir_fft(const float *__restrict__ ir, float2 *__restrict__ H,
                                                     const float2 *__restrict__ tw, int B, int P, int first_set,
                                                     int first_partition)
{
    constexpr int M = N / 2;
    __shared__ float2 la[M], lb[M];
    const int k = blockIdx.x, r = blockIdx.y;
    load_block<M>(ir + ((size_t) r * gridDim.x + k) * B, B, la);
    __syncthreads();
    float2 *Z = fft_lds<M, false>(la, lb, tw);
    float2 *Xn = (Z == la) ? lb : la;
    split_forward<M>(Z, Xn, tw);
    __syncthreads();
    float2 *dst = H + ((size_t) (first_set + r) * P + first_partition + k) * M;
    for (int b = threadIdx.x; b < M; b += kThreads) dst[b] = Xn[b];
}

}  // namespace irb
