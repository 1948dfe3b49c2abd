// cu_mask_probe.hip -- (1) does hipExtStreamCreateWithCUMask confine a kernel's workgroups to the CUs of its mask on this system,
// (2) do two kernels on two streams run at the same time when one of them is long-lived and was launched first -- the situation of the
// one-shot pipeline's forward-FFT kernel (link-bound, every CU taken) and the matrix-core multiply behind it.
// hipcc -O2 --offload-arch=gfx950 -o tools/cu_mask_probe tools/cu_mask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// every workgroup: where it runs (XCC_ID, HW_ID) and the 100 MHz clock at start / end; spins `ticks` of that clock
__global__ __launch_bounds__(1024) void k_spin(unsigned long long *rec, unsigned ticks)
{
    unsigned long long t0, t1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    do { asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory"); } while (t1 - t0 < ticks);
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)), xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));
        rec[blockIdx.x * 4 + 0] = ((unsigned long long) xcc << 32) | hw;
        rec[blockIdx.x * 4 + 1] = t0;
        rec[blockIdx.x * 4 + 2] = t1;
    }
}

static int distinct_cus(const std::vector<unsigned long long> &r, int n)
{
    std::set<unsigned long long> s;
    for (int i = 0; i < n; i++) {
        const unsigned hw = (unsigned) r[i * 4], xcc = (unsigned) (r[i * 4] >> 32) & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;      // HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
        s.insert(((unsigned long long) xcc << 16) | (se << 8) | (sh << 4) | cu);
    }
    return (int) s.size();
}

int main()
{
    CK(hipSetDevice(0));
    uint32_t io[8], mac[8];
    for (int w = 0; w < 8; w++) { io[w] = 0x01010101u; mac[w] = ~io[w]; }
    hipStream_t s_io, s_mac, s_a, s_b;
    CK(hipExtStreamCreateWithCUMask(&s_io, 8, io));
    CK(hipExtStreamCreateWithCUMask(&s_mac, 8, mac));
    CK(hipStreamCreateWithFlags(&s_a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s_b, hipStreamNonBlocking));
    const int NA = 1024, NB = 224;
    unsigned long long *ra, *rb;
    CK(hipHostMalloc((void **) &ra, sizeof(unsigned long long) * 4 * NA));
    CK(hipHostMalloc((void **) &rb, sizeof(unsigned long long) * 4 * NA));
    auto run = [&](hipStream_t sa, hipStream_t sb, const char *what) -> int {
        // A: 1024 workgroups of 1024 threads, 20 us each (a long-lived kernel that wants every CU), launched FIRST; B: 224 workgroups, 50 us each
        for (int rep = 0; rep < 3; rep++) {
            k_spin<<<NA, 1024, 0, sa>>>(ra, 2000);
            k_spin<<<NB, 1024, 0, sb>>>(rb, 5000);
            CK(hipStreamSynchronize(sa));
            CK(hipStreamSynchronize(sb));
        }
        std::vector<unsigned long long> va(ra, ra + 4 * NA), vb(rb, rb + 4 * NA);
        unsigned long long a0 = ~0ull, a1 = 0, b0 = ~0ull, b1 = 0;
        for (int i = 0; i < NA; i++) { a0 = std::min(a0, va[i * 4 + 1]); a1 = std::max(a1, va[i * 4 + 2]); }
        for (int i = 0; i < NB; i++) { b0 = std::min(b0, vb[i * 4 + 1]); b1 = std::max(b1, vb[i * 4 + 2]); }
        std::printf("%s: A (1024 wgs x 20 us) ran %.1f .. %.1f us on %d distinct CUs; B (224 wgs x 50 us) ran %.1f .. %.1f us on %d distinct CUs\n", what,
                    0.0, (a1 - a0) / 100.0, distinct_cus(va, NA), ((double) b0 - (double) a0) / 100.0, ((double) b1 - (double) a0) / 100.0, distinct_cus(vb, NB));
        return 0;
    };
    if (run(s_a, s_b, "plain streams ")) return 1;
    if (run(s_io, s_mac, "masked streams")) return 1;
    if (run(s_io, s_b, "A masked, B plain")) return 1;
    return 0;
}
