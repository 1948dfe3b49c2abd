// pin_probe.hip -- what registering a caller's pageable buffer costs on this system (hipHostRegister / hipHostUnregister), next to the
// single-thread memcpy it would replace in irb_convolve_periodic.  hipcc -O2 --offload-arch=gfx950 -o tools/pin_probe tools/pin_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_touch(float *p, size_t n) { for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) p[i] = (float) i; }
int main()
{
    CK(hipSetDevice(0));
    for (size_t mb : {1, 4, 16}) {
        const size_t bytes = mb << 20;
        float *buf = (float *) aligned_alloc(4096, bytes), *pin = nullptr;
        std::memset(buf, 1, bytes);
        CK(hipHostMalloc((void **) &pin, bytes));
        std::memset(pin, 2, bytes);
        std::vector<double> reg, unreg, cpy, kern;
        for (int it = 0; it < 15; it++) {
            double t0 = now_us();
            CK(hipHostRegister(buf, bytes, hipHostRegisterMapped));
            double t1 = now_us();
            float *d = nullptr;
            CK(hipHostGetDevicePointer((void **) &d, buf, 0));
            k_touch<<<256, 256>>>(d, bytes / 4);
            CK(hipDeviceSynchronize());
            double t2 = now_us();
            CK(hipHostUnregister(buf));
            double t3 = now_us();
            std::memcpy(buf, pin, bytes);
            double t4 = now_us();
            reg.push_back(t1 - t0); kern.push_back(t2 - t1); unreg.push_back(t3 - t2); cpy.push_back(t4 - t3);
        }
        auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        std::printf("%zu MiB: hipHostRegister %.1f us, kernel writing it over PCIe + sync %.1f us, hipHostUnregister %.1f us; memcpy pinned -> pageable %.1f us\n",
                    mb, med(reg), med(kern), med(unreg), med(cpy));
        std::free(buf);
        CK(hipHostFree(pin));
    }
    return 0;
}
