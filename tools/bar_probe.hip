// bar_probe.hip -- can the audio thread store its input blocks straight into device memory (large BAR) instead of leaving them in
// pinned host memory for the kernel to fetch over PCIe?  Fine-grained device allocation written by the CPU, read by a kernel that
// answers through mapped host memory; 2000 rounds with fresh values (a stale GPU cache line would show), timed against the same
// round with the input in mapped pinned host memory.
//   hipcc -O2 --offload-arch=gfx950 tools/bar_probe.hip -o tools/bar_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k(const float *in, float *out, unsigned *done, unsigned seq, int n)
{
    for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = in[i] * 2.0f;
    __syncthreads();
    if (threadIdx.x == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); __hip_atomic_store(done, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
}

int main()
{
    int large_bar = -1;
    (void) hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, 0);
    printf("hipDeviceAttributeIsLargeBar = %d\n", large_bar);
    const int n = 768;                                  // 3 KB: two 256-sample blocks + one IR partition
    float *d_in = nullptr, *h_in = nullptr, *h_out = nullptr, *dh_in, *dh_out;
    unsigned *h_done, *dh_done;
    CK(hipExtMallocWithFlags((void **) &d_in, 1 << 16, hipDeviceMallocFinegrained));
    CK(hipHostMalloc((void **) &h_in, 1 << 16, hipHostMallocMapped));
    CK(hipHostMalloc((void **) &h_out, 1 << 16, hipHostMallocMapped));
    CK(hipHostMalloc((void **) &h_done, 64, hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void **) &dh_in, h_in, 0));
    CK(hipHostGetDevicePointer((void **) &dh_out, h_out, 0));
    CK(hipHostGetDevicePointer((void **) &dh_done, h_done, 0));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    std::vector<float> src(n);
    *h_done = 0;
    unsigned seq = 0;
    for (int mode = 0; mode < 2; mode++) {              // 0: input in pinned host memory, 1: input stored into device memory through the BAR
        std::vector<double> us;
        int bad = 0;
        for (int r = 0; r < 2200; r++) {
            for (int i = 0; i < n; i++) src[i] = (float) (r * 7 + i);
            auto t0 = std::chrono::steady_clock::now();
            if (mode == 0) std::memcpy(h_in, src.data(), sizeof(float) * n);
            else { std::memcpy(d_in, src.data(), sizeof(float) * n); __builtin_ia32_sfence(); }
            k<<<1, 256, 0, st>>>(mode == 0 ? dh_in : d_in, dh_out, dh_done, ++seq, n);
            while (*(volatile unsigned *) h_done != seq) __builtin_ia32_pause();
            auto t1 = std::chrono::steady_clock::now();
            for (int i = 0; i < n; i++) bad += h_out[i] != src[i] * 2.0f;
            if (r >= 200) us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
        }
        std::sort(us.begin(), us.end());
        printf("%s: median %.2f us, p99 %.2f us per round (copy 3 KB + launch + kernel + completion word), wrong values %d\n",
               mode == 0 ? "input in mapped pinned host memory " : "input stored through the BAR into HBM", us[us.size() / 2], us[us.size() * 99 / 100], bad);
    }
    return 0;
}
