// valu_peak.hip -- what fp32 rate can the vector ALU of gfx950 actually sustain, and with which instruction?
// The time-tiled MAC (k_cmac_tiled) is priced against the 157.3 TFLOP/s fp32 vector peak; this probe measures, on the
// box, register-only loops of
//   A  v_fma_f32                      (one FMA per lane per instruction)
//   B  v_pk_fma_f32                   (two FMAs per lane per instruction)
//   C  the complex multiply-accumulate written as two v_pk_fma_f32 with op_sel / neg_lo modifiers (checked on the host)
// at 1, 2, 4 and 8 waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 tools/valu_peak.hip -o tools/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int NACC = 16;

// shader-clock and 100 MHz stamps around the loop (one lane per workgroup; a diagnostic tool: the stamps are the point)
__device__ unsigned long long g_stamp[4 * 2048];
struct Stamp {
    unsigned long long t0, r0;
    __device__ __forceinline__ void begin() { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    __device__ __forceinline__ void end() const
    {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) { g_stamp[2 * blockIdx.x] = t1 - t0; g_stamp[2 * blockIdx.x + 1] = r1 - r0; }
    }
};

__global__ __launch_bounds__(256) void k_fma(float *out, int iters, float a, float b)
{
    float acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = (float) (threadIdx.x + i);
    Stamp st;
    st.begin();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < NACC; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    st.end();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_pk_fma(float *out, int iters, float a, float b)
{
    v2f acc[NACC];
    const v2f va = {a, a * 0.5f}, vb = {b, b * 0.25f};
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = v2f{(float) (threadIdx.x + i), (float) i};
    Stamp st;
    st.begin();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < NACC; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(va), "v"(vb));
    }
    st.end();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// acc += x * h (complex) as two packed FMAs:
//   acc.xy += (x.x, x.x) * (h.x, h.y)          op_sel:[0,0,0] op_sel_hi:[0,1,1]
//   acc.xy += (-x.y, x.y) * (h.y, h.x)         op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]
__device__ __forceinline__ void cmac_pk(v2f &acc, const v2f x, const v2f h)
{
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                 "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
                 : "+v"(acc) : "v"(x), "v"(h));
}

__global__ __launch_bounds__(256) void k_cmac_pk(float *out, int iters, float a, float b)
{
    v2f acc[NACC];
    const v2f x = {a, a * 0.5f}, h = {b, b * 0.25f};
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = v2f{0.f, 0.f};
    Stamp st;
    st.begin();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int i = 0; i < NACC; i++) cmac_pk(acc[i], x, h);
    }
    st.end();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_cmac_check(float *out, const float *in)
{
    v2f acc = {in[0], in[1]};
    const v2f x = {in[2], in[3]}, h = {in[4], in[5]};
    cmac_pk(acc, x, h);
    out[0] = acc.x;
    out[1] = acc.y;
}

static double g_clock_ghz = 0, g_cycles = 0;      // of the last run(): median over workgroups

template <typename K>
static double run(K kern, int wgs_per_cu, int iters, float *d_out, int reps = 5)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int grid = 256 * wgs_per_cu;
    kern<<<grid, 256>>>(d_out, iters, 1.0001f, 0.9999f);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0));
        kern<<<grid, 256>>>(d_out, iters, 1.0001f, 0.9999f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    std::vector<unsigned long long> st(2 * grid);
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 2 * grid));
    std::vector<double> clk, cyc;
    for (int g = 0; g < grid; g++) { clk.push_back((double) st[2 * g] / (double) st[2 * g + 1] * 0.1); cyc.push_back((double) st[2 * g]); }
    std::sort(clk.begin(), clk.end());
    std::sort(cyc.begin(), cyc.end());
    g_clock_ghz = clk[clk.size() / 2];
    g_cycles = cyc[cyc.size() / 2];
    return best;
}

int main()
{
    float *d_out, *d_in;
    CK(hipMalloc(&d_out, sizeof(float) * 256 * 8 * 256));
    CK(hipMalloc(&d_in, sizeof(float) * 8));
    const float hin[6] = {0.5f, -0.25f, 1.5f, -2.0f, 0.75f, 3.0f};
    CK(hipMemcpy(d_in, hin, sizeof hin, hipMemcpyHostToDevice));
    k_cmac_check<<<1, 1>>>(d_out, d_in);
    float got[2];
    CK(hipMemcpy(got, d_out, sizeof got, hipMemcpyDeviceToHost));
    const float er = hin[0] + hin[2] * hin[4] - hin[3] * hin[5], ei = hin[1] + hin[2] * hin[5] + hin[3] * hin[4];
    printf("cmac_pk check: got (%g, %g) expected (%g, %g) %s\n", got[0], got[1], er, ei, (got[0] == er && got[1] == ei) ? "OK" : "MISMATCH");

    const int iters = 4000;
    printf("%-34s %10s %8s %9s %11s %22s\n", "loop", "waves/SIMD", "ms", "TFLOP/s", "clock GHz", "cycles/instr per SIMD");
    for (int w : {1, 2, 4, 8}) {
        const double lanes = 256.0 * w * 256;
        double ms = run(k_fma, w, iters, d_out);
        printf("%-34s %10d %8.3f %9.1f %11.2f %22.2f\n", "v_fma_f32", w, ms, lanes * iters * 4 * NACC * 2 / ms / 1e9, g_clock_ghz, g_cycles / (iters * 4.0 * NACC * w));
        ms = run(k_pk_fma, w, iters, d_out);
        printf("%-34s %10d %8.3f %9.1f %11.2f %22.2f\n", "v_pk_fma_f32", w, ms, lanes * iters * 4 * NACC * 4 / ms / 1e9, g_clock_ghz, g_cycles / (iters * 4.0 * NACC * w));
        ms = run(k_cmac_pk, w, iters, d_out);
        printf("%-34s %10d %8.3f %9.1f %11.2f %22.2f\n", "cmac = 2 x v_pk_fma_f32 (op_sel)", w, ms, lanes * iters * 2 * NACC * 8 / ms / 1e9, g_clock_ghz, g_cycles / (iters * 2.0 * NACC * 2 * w));
    }
    return 0;
}
