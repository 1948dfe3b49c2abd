// mfma_peak.hip -- what the exact-fp32 matrix instruction of gfx950 sustains (v_mfma_f32_32x32x2_f32: 4096 flops per wave per
// 64 cycles = the same 157.3 TFLOP/s peak as the vector ALU), register-only, at 1 / 2 / 4 waves per SIMD, with 2 or 4 independent
// accumulator tiles per wave and shader-clock stamps.  Question behind it: can a Toeplitz-block form of the time-tiled MAC get
// past the VALU kernel's 0.43 of peak?  Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v16f __attribute__((ext_vector_type(16)));

__device__ unsigned long long g_stamp[2 * 4096];

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(float *out, int iters, float a, float b)
{
    v16f acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++)
#pragma unroll
        for (int j = 0; j < 16; j++) acc[i][j] = (float) (threadIdx.x + i + j);
    float va = a + threadIdx.x * 1e-6f, vb = b - threadIdx.x * 1e-6f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(va, vb, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { g_stamp[2 * blockIdx.x] = t1 - t0; g_stamp[2 * blockIdx.x + 1] = r1 - r0; }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; i++)
#pragma unroll
        for (int j = 0; j < 16; j++) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef float v4f __attribute__((ext_vector_type(4)));

// v_mfma_f32_16x16x4_f32: 2048 flops per wave, 8 passes (32 cycles)
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma16(float *out, int iters, float a, float b)
{
    v4f acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (float) (threadIdx.x + i + j);
    float va = a + threadIdx.x * 1e-6f, vb = b - threadIdx.x * 1e-6f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(va, vb, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { g_stamp[2 * blockIdx.x] = t1 - t0; g_stamp[2 * blockIdx.x + 1] = r1 - r0; }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the tiled MAC's inner loop without its memory traffic: 16 MFMAs on three accumulators (r, i, n, i), eight operand registers,
// a short loop (unroll 1), THREADS threads per workgroup
template <int THREADS, int PATTERN>
__global__ __launch_bounds__(THREADS) void k_mfma16_loop(float *out, int iters, float a, float b)
{
    v4f acc_r = {0.f, 0.f, 0.f, 0.f}, acc_i = acc_r, acc_n = acc_r;
    float x[8], h[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { x[i] = a + threadIdx.x * 1e-6f * (i + 1); h[i] = b - threadIdx.x * 1e-6f * (i + 1); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
            if (PATTERN == 0) {             // r, i, n, i
                acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(x[2 * g], h[2 * g], acc_r, 0, 0, 0);
                acc_i = __builtin_amdgcn_mfma_f32_16x16x4f32(x[2 * g + 1], h[2 * g], acc_i, 0, 0, 0);
                acc_n = __builtin_amdgcn_mfma_f32_16x16x4f32(x[2 * g + 1], h[2 * g + 1], acc_n, 0, 0, 0);
                acc_i = __builtin_amdgcn_mfma_f32_16x16x4f32(x[2 * g], h[2 * g + 1], acc_i, 0, 0, 0);
            } else {                        // four accumulators: no MFMA depends on one of the three before it
                acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(x[2 * g], h[2 * g], acc_r, 0, 0, 0);
                acc_i = __builtin_amdgcn_mfma_f32_16x16x4f32(x[2 * g + 1], h[2 * g], acc_i, 0, 0, 0);
                acc_n = __builtin_amdgcn_mfma_f32_16x16x4f32(x[2 * g + 1], h[2 * g + 1], acc_n, 0, 0, 0);
                acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(x[2 * g], h[2 * g + 1], acc_r, 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { g_stamp[2 * blockIdx.x] = t1 - t0; g_stamp[2 * blockIdx.x + 1] = r1 - r0; }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) s += acc_r[j] + acc_i[j] + acc_n[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static void run_loop(const char *name, K kern, int threads, int iters, float *d_out)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int grid = 256;
    kern<<<grid, threads>>>(d_out, iters, 1.0001f, 0.9999f);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0));
        kern<<<grid, threads>>>(d_out, iters, 1.0001f, 0.9999f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    std::vector<unsigned long long> st(2 * grid);
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 2 * grid));
    std::vector<double> clk, cyc;
    for (int g = 0; g < grid; g++) { clk.push_back((double) st[2 * g] / (double) st[2 * g + 1] * 0.1); cyc.push_back((double) st[2 * g]); }
    std::sort(clk.begin(), clk.end());
    std::sort(cyc.begin(), cyc.end());
    const double n_mfma = (double) iters * 16, waves_per_simd = threads / 256.0;
    printf("%-44s %10.0f %8.3f %9.1f %11.2f %22.2f\n", name, waves_per_simd, best, 256.0 * threads / 64 * n_mfma * 2048.0 / best / 1e9, clk[clk.size() / 2],
           cyc[cyc.size() / 2] / (n_mfma * waves_per_simd));
}

template <typename K>
static void run(const char *name, K kern, int nacc, int wgs_per_cu, int iters, float *d_out, int reps_in = 4, double flops_each = 4096.0)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int grid = 256 * wgs_per_cu;
    kern<<<grid, 256>>>(d_out, iters, 1.0001f, 0.9999f);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0));
        kern<<<grid, 256>>>(d_out, iters, 1.0001f, 0.9999f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    std::vector<unsigned long long> st(2 * grid);
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 2 * grid));
    std::vector<double> clk, cyc;
    for (int g = 0; g < grid; g++) { clk.push_back((double) st[2 * g] / (double) st[2 * g + 1] * 0.1); cyc.push_back((double) st[2 * g]); }
    std::sort(clk.begin(), clk.end());
    std::sort(cyc.begin(), cyc.end());
    const double n_mfma = (double) iters * reps_in * nacc;
    const double flops = 256.0 * wgs_per_cu * 4 * n_mfma * flops_each;
    printf("%-28s %10d %8.3f %9.1f %11.2f %22.2f\n", name, wgs_per_cu, best, flops / best / 1e9, clk[clk.size() / 2], cyc[cyc.size() / 2] / (n_mfma * wgs_per_cu));
}

int main()
{
    float *d_out;
    CK(hipMalloc(&d_out, sizeof(float) * 256 * 8 * 256));
    printf("%-28s %10s %8s %9s %11s %22s\n", "loop", "waves/SIMD", "ms", "TFLOP/s", "clock GHz", "cycles/mfma per SIMD");
    for (int w : {1, 2, 4}) {
        run("mfma_f32_32x32x2, 2 tiles", k_mfma<2>, 2, w, 2000, d_out);
        run("mfma_f32_32x32x2, 4 tiles", k_mfma<4>, 4, w, 1000, d_out);
        run("mfma_f32_16x16x4, 2 tiles", k_mfma16<2>, 2, w, 2000, d_out, 8, 2048.0);
        run("mfma_f32_16x16x4, 4 tiles", k_mfma16<4>, 4, w, 1000, d_out, 8, 2048.0);
    }
    printf("%-44s %10s %8s %9s %11s %22s\n", "16-MFMA loop body, one workgroup per CU", "waves/SIMD", "ms", "TFLOP/s", "clock GHz", "cycles/mfma per SIMD");
    run_loop("r,i,n,i  256 threads", k_mfma16_loop<256, 0>, 256, 4000, d_out);
    run_loop("r,i,n,i  512 threads", k_mfma16_loop<512, 0>, 512, 2000, d_out);
    run_loop("r,i,n,i 1024 threads", k_mfma16_loop<1024, 0>, 1024, 1000, d_out);
    run_loop("r,i,n,r 1024 threads", k_mfma16_loop<1024, 1>, 1024, 1000, d_out);
    return 0;
}
