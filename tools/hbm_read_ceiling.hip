// hbm_read_ceiling.hip -- what a pure streaming READ of the bench's footprint reaches on this MI355X:
// the practical ceiling for k_step_fused (which is >99.7 % reads).  Each workgroup reads two contiguous
// 768 KB regions (like one stream's H and FDL) with 16-byte loads, sums them, writes one float.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/hbm_read_ceiling tools/hbm_read_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
template <bool NT, int U>
__global__ __launch_bounds__(256) void k_read(const v4f *__restrict__ a, const v4f *__restrict__ b, float *out, int rows /*float4 per region*/)
{
    const v4f *pa = a + (size_t) blockIdx.x * rows, *pb = b + (size_t) blockIdx.x * rows;
    v4f acc = {0, 0, 0, 0};
    for (int i = threadIdx.x; i + (U - 1) * 256 < rows; i += U * 256) {
        v4f x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            x[u] = NT ? __builtin_nontemporal_load(pa + i + u * 256) : pa[i + u * 256];
            y[u] = NT ? __builtin_nontemporal_load(pb + i + u * 256) : pb[i + u * 256];
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc += x[u] * y[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;
}
// same two-region pattern, but fewer concurrent regions: TPB threads per workgroup, each workgroup walks
// `per` streams one after the other
template <int TPB>
__global__ __launch_bounds__(TPB) void k_read_seq(const v4f *__restrict__ a, const v4f *__restrict__ b, float *out, int rows, int per)
{
    v4f acc = {0, 0, 0, 0};
    for (int q = 0; q < per; q++) {
        const size_t sidx = (size_t) blockIdx.x * per + q;
        const v4f *pa = a + sidx * rows, *pb = b + sidx * rows;
        for (int i = threadIdx.x; i + TPB < rows; i += 2 * TPB) {
            v4f x0 = __builtin_nontemporal_load(pa + i), x1 = __builtin_nontemporal_load(pa + i + TPB);
            v4f y0 = __builtin_nontemporal_load(pb + i), y1 = __builtin_nontemporal_load(pb + i + TPB);
            acc += x0 * y0 + x1 * y1;
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;
}
// two regions, rows of 2 KB (128 float4): region a ascending, region b walked DESCENDING row by row (what the FDL
// ring does: partition k pairs with slot head-k), 2 stripes of 128 lanes like k_step_fused<512>
template <bool BACKWARD>
__global__ __launch_bounds__(256) void k_read_rows(const v4f *__restrict__ a, const v4f *__restrict__ b, float *out, int nrows)
{
    const v4f *pa = a + (size_t) blockIdx.x * nrows * 128, *pb = b + (size_t) blockIdx.x * nrows * 128;
    const int col = threadIdx.x & 127, stripe = threadIdx.x >> 7;
    v4f acc = {0, 0, 0, 0};
    for (int k = stripe; k + 2 < nrows; k += 4) {
        const int s0 = BACKWARD ? nrows - 1 - k : k, s1 = BACKWARD ? nrows - 1 - (k + 2) : k + 2;
        v4f x0 = __builtin_nontemporal_load(pa + (size_t) k * 128 + col), x1 = __builtin_nontemporal_load(pa + (size_t) (k + 2) * 128 + col);
        v4f y0 = __builtin_nontemporal_load(pb + (size_t) s0 * 128 + col), y1 = __builtin_nontemporal_load(pb + (size_t) s1 * 128 + col);
        acc += x0 * y0 + x1 * y1;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;
}
// linear grid-stride stream of the same bytes: the friendliest possible pattern
template <bool NT>
__global__ __launch_bounds__(256) void k_linear(const v4f *__restrict__ a, float *out, size_t n)
{
    v4f acc = {0, 0, 0, 0};
    const size_t stride = (size_t) gridDim.x * 256 * 4;
    for (size_t i = (size_t) blockIdx.x * 256 * 4 + threadIdx.x; i + 3 * 256 < n; i += stride) {
        v4f x0 = NT ? __builtin_nontemporal_load(a + i) : a[i];
        v4f x1 = NT ? __builtin_nontemporal_load(a + i + 256) : a[i + 256];
        v4f x2 = NT ? __builtin_nontemporal_load(a + i + 512) : a[i + 512];
        v4f x3 = NT ? __builtin_nontemporal_load(a + i + 768) : a[i + 768];
        acc += x0 + x1 + x2 + x3;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;
}
// fills a buffer with pseudo-random floats in [-1, 1): DRAM power, and with it the clocks the chip sustains, depend on
// the DATA (MI355X_MICROARCH.md, DVFS: zero-filled inputs ran 19 % faster in a GEMM), so the ceiling must be measured
// on data like the spectra the real kernel streams, not on zeros
__global__ void k_fill(float *p, size_t n, unsigned seed)
{
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        unsigned long long z = (i + seed) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        p[i] = (float) ((long long) (z >> 40) - (1 << 23)) / (float) (1 << 23);
    }
}

int main(int argc, char **argv)
{
    const bool zeros = argc > 1 && argv[1][0] == 'z';
    const int S = 2048, rows = 375 * 128;            // 375 rows of 2 KB = 128 float4
    const size_t n = (size_t) S * rows;
    v4f *a, *b; float *o;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&o, S * 4);
    if (zeros) { hipMemset(a, 0, n * 16); hipMemset(b, 0, n * 16); }
    else { k_fill<<<4096, 256>>>((float *) a, n * 4, 1u); k_fill<<<4096, 256>>>((float *) b, n * 4, 77u); }
    hipDeviceSynchronize();
    printf("data: %s\n", zeros ? "all zeros" : "pseudo-random floats in [-1, 1)");
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, const char *name) {
        for (int i = 0; i < 5; i++) kern<<<S, 256>>>(a, b, o, rows);
        hipEventRecord(e0);
        for (int i = 0; i < 50; i++) kern<<<S, 256>>>(a, b, o, rows);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-24s %.1f GB/s (%.4f ms per pass of %.3f GB)\n", name, 2.0 * n * 16 / (ms / 50 * 1e-3) / 1e9, ms / 50, 2.0 * n * 16 / 1e9);
    };
    run(k_read<false, 2>, "default loads, U=2");
    run(k_read<true, 2>, "non-temporal, U=2");
    run(k_read<true, 4>, "non-temporal, U=4");
    run(k_read<false, 4>, "default loads, U=4");
    run(k_read<true, 2>, "non-temporal, U=2");
    auto runs = [&](auto kern, int tpb, int per, const char *name) {
        const int grid = S / per;
        for (int i = 0; i < 5; i++) kern<<<grid, tpb>>>(a, b, o, rows, per);
        hipEventRecord(e0);
        for (int i = 0; i < 50; i++) kern<<<grid, tpb>>>(a, b, o, rows, per);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s tpb %4d per %d: %.1f GB/s\n", name, tpb, per, 2.0 * n * 16 / (ms / 50 * 1e-3) / 1e9);
    };
    runs(k_read_seq<256>, 256, 1, "two regions nt");
    runs(k_read_seq<256>, 256, 2, "two regions nt");
    runs(k_read_seq<256>, 256, 4, "two regions nt");
    runs(k_read_seq<512>, 512, 1, "two regions nt");
    runs(k_read_seq<1024>, 1024, 1, "two regions nt");
    runs(k_read_seq<512>, 512, 2, "two regions nt");
    auto runr = [&](auto kern, const char *name) {
        for (int i = 0; i < 5; i++) kern<<<S, 256>>>(a, b, o, 375);
        hipEventRecord(e0);
        for (int i = 0; i < 50; i++) kern<<<S, 256>>>(a, b, o, 375);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %.1f GB/s\n", name, 2.0 * n * 16 / (ms / 50 * 1e-3) / 1e9);
    };
    runr(k_read_rows<false>, "rows: both ascending");
    runr(k_read_rows<true>, "rows: b descending");
    runr(k_read_rows<false>, "rows: both ascending");
    runr(k_read_rows<true>, "rows: b descending");
    auto runl = [&](auto kern, int grid, const char *name) {
        for (int i = 0; i < 5; i++) kern<<<grid, 256>>>(a, o, n);
        hipEventRecord(e0);
        for (int i = 0; i < 50; i++) kern<<<grid, 256>>>(a, o, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-24s grid %5d: %.1f GB/s\n", name, grid, 1.0 * n * 16 / (ms / 50 * 1e-3) / 1e9);
    };
    runl(k_linear<true>, 2048, "linear nt");
    runl(k_linear<false>, 2048, "linear default");
    runl(k_linear<true>, 4096, "linear nt");
    runl(k_linear<true>, 1024, "linear nt");
    runl(k_linear<true>, 8192, "linear nt");
    return 0;
}
