// deconv_engine.hip -- host side of the config-5 chain: large real FFT plans (irb_lfft_*), the one-shot
// mirrors of tools::fftTransform / fftInvTransform (reference fp/tools.cpp:321-369), convolution::deconvolve
// (fp/convolution.cpp:351-403), ir::invertFilter (fp/ir.cpp:13-18), and the host-resident IR preparation the
// north star keeps on the CPU: convolution::averagingFilter (fp/convolution.cpp:406-546) and ir::shifteroo
// (fp/ir.cpp:85-103).  The transforms and the spectral division run on the GPU; there is no CPU fallback
// for them.
#include "../../include/irb_deconv.h"
#include "lfft_kernels.hpp"
#include "host_ranges.hpp"
#include "irb_env.hpp"
#include "../../include/irb_fdl.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

extern "C" int irb_set_error_(int code, const char *msg);   // fdl_engine.hip

namespace {

int failf(int code, const char *fmt, const char *what, const char *detail)
{
    char msg[400];
    std::snprintf(msg, sizeof msg, fmt, what, detail);
    return irb_set_error_(code, msg);
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return failf(IRB_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

bool is_pow2(int x) { return x != 0 && (x & (x - 1)) == 0; }
int ilog2(int x) { int l = 0; while ((1 << (l + 1)) <= x) l++; return l; }

std::vector<float2> root_table(int root, int count, int stride)      // W_root^(i*stride), i < count, in double
{
    std::vector<float2> t((size_t) std::max(count, 1));
    for (int i = 0; i < count; i++) {
        const double ph = -2.0 * 3.14159265358979323846 * (double) i * (double) stride / (double) root;
        t[i] = make_float2((float) std::cos(ph), (float) std::sin(ph));
    }
    return t;
}

}  // namespace

struct SmoothWindows;
void free_windows(SmoothWindows *w);

struct irb_lfft {
    int device = 0, N = 0, M = 0, M1 = 0, M2 = 0, max_batch = 1, logE = 0;
    int tile_rows = 0;                           // four-step passes: rows/columns per workgroup (0 = one wave per item)
    int fuse_split = 1;                          // real <-> complex split inside the tiled passes (2 launches instead of 3): bit 0 = inverse, bit 1 = forward
    hipStream_t stream = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    float2 *d_tw1 = nullptr, *d_tw2 = nullptr, *d_tabA = nullptr, *d_tabB = nullptr;
    float2 *d_w0 = nullptr, *d_w1 = nullptr;     // complex work buffers [max_batch][M]
    float *d_time = nullptr, *d_spec = nullptr;  // [max_batch][N] samples, [max_batch][2N] spectrum (one-shot API)
    bool spec_refill = false;                    // irb_lfft_invalidate_spec: the next forward transform into d_spec restores the zeros and the flag
    bool spec_upper_zero = false;                // floats above the Nyquist bin of d_spec are zero (set at create, kept by every
                                                 // kernel of this library: they write bins 0..N/2 only): forward into d_spec skips the fill
    // smoothing (averagingFilter) workspace, created by the first smoothing call on this plan: log amplitudes and window
    // means live in MAPPED pinned host memory (the element-wise kernels write / read it directly), and the window bounds
    // of the last (bins, Hz per bin, octave fraction) are kept
    float *h_log = nullptr, *h_mean = nullptr, *d_hlog = nullptr, *d_hmean = nullptr;
    SmoothWindows *win = nullptr;
    // convolveNonPeriodic's pinned staging [max_batch][N] (created by its first staged call on this plan) and the events behind its
    // download chunks
    float *h_stage = nullptr;
    hipEvent_t ev_chunk[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

namespace {

#define FOR_M(M_, ...)                                                       \
    switch (M_) {                                                            \
    case 16:   { constexpr int MM = 16;   __VA_ARGS__; } break;              \
    case 32:   { constexpr int MM = 32;   __VA_ARGS__; } break;              \
    case 64:   { constexpr int MM = 64;   __VA_ARGS__; } break;              \
    case 128:  { constexpr int MM = 128;  __VA_ARGS__; } break;              \
    case 256:  { constexpr int MM = 256;  __VA_ARGS__; } break;              \
    case 512:  { constexpr int MM = 512;  __VA_ARGS__; } break;              \
    case 1024: { constexpr int MM = 1024; __VA_ARGS__; } break;              \
    case 2048: { constexpr int MM = 2048; __VA_ARGS__; } break;              \
    default: return irb_set_error_(IRB_ERR_ARG, "unsupported FFT factor");   \
    }

irb::LfftArgs args_of(const irb_lfft *p, int batch)
{
    irb::LfftArgs a;
    a.tw1 = p->d_tw1; a.tw2 = p->d_tw2; a.tabA = p->d_tabA; a.tabB = p->d_tabB; a.logE = p->logE;
    a.N = p->N; a.M = p->M; a.M1 = p->M1; a.M2 = p->M2; a.batch = batch; a.scale = 1.0f; a.zero_fill = 1;
    return a;
}

// one tiled pass; R is fixed per plan (8 or 16 rows per workgroup; LDS = R*(MX+1)*8 + MX*8 bytes, up to 144 KiB).
// SETUP = true only raises the kernel's dynamic-LDS limit (done once per plan, at create).
template <int MX, bool INV, bool COLS, bool SETUP = false>
int launch_tile(irb_lfft *p, const irb::LfftArgs &a, const float2 *src, float2 *dst, int nwg)
{
    const int R = p->tile_rows;
    const size_t lds = sizeof(float2) * ((size_t) R * (MX + 1) + irb::TwLds<2 * MX>::SIZE);
    if (R == 16 && MX <= 1024) {
        auto k = irb::k_lfft_tile<MX, 16, INV, COLS>;
        if (SETUP) { HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); }
        else k<<<dim3(nwg), dim3(irb::kWave * 16), lds, p->stream>>>(a, src, dst);
    } else if (R == 8) {
        auto k = irb::k_lfft_tile<MX, 8, INV, COLS>;
        if (SETUP) { HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); }
        else k<<<dim3(nwg), dim3(irb::kWave * 8), lds, p->stream>>>(a, src, dst);
    } else if (R == 4) {
        auto k = irb::k_lfft_tile<MX, 4, INV, COLS>;
        if (SETUP) { HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); }
        else k<<<dim3(nwg), dim3(irb::kWave * 4), lds, p->stream>>>(a, src, dst);
    } else if (R == 2) {
        auto k = irb::k_lfft_tile<MX, 2, INV, COLS>;
        if (SETUP) { HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); }
        else k<<<dim3(nwg), dim3(irb::kWave * 2), lds, p->stream>>>(a, src, dst);
    } else {
        return irb_set_error_(IRB_ERR_ARG, "unsupported tile height");
    }
    return IRB_OK;
}

// the tiled passes with the real <-> complex split fused in (k_lfft_rows_split / k_lfft_cols_unsplit): FWD = second pass of
// a forward transform (T -> spectrum), !FWD = first pass of an inverse transform (spectrum -> T)
template <int MX, bool FWD, bool SETUP = false>
int launch_fused(irb_lfft *p, const irb::LfftArgs &a, const void *src, void *dst, int nwg)
{
    const int R = p->tile_rows;
    const size_t lds = sizeof(float2) * ((size_t) R * (MX + 1) + irb::TwLds<2 * MX>::SIZE);
#define IRB_FUSED_CASE(R_)                                                                                                        \
    if (R == R_) {                                                                                                                \
        if constexpr (FWD) {                                                                                                      \
            auto k = irb::k_lfft_rows_split<MX, R_>;                                                                              \
            if (SETUP) { HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); } \
            else k<<<dim3(nwg), dim3(irb::kWave * R_), lds, p->stream>>>(a, static_cast<const float2 *>(src), static_cast<float *>(dst));           \
        } else {                                                                                                                  \
            auto k = irb::k_lfft_cols_unsplit<MX, R_>;                                                                            \
            if (SETUP) { HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); } \
            else k<<<dim3(nwg), dim3(irb::kWave * R_), lds, p->stream>>>(a, static_cast<const float *>(src), static_cast<float2 *>(dst));           \
        }                                                                                                                         \
        return IRB_OK;                                                                                                            \
    }
    IRB_FUSED_CASE(4)
    IRB_FUSED_CASE(8)
#undef IRB_FUSED_CASE
    return irb_set_error_(IRB_ERR_ARG, "unsupported tile height for the fused split passes");
}

bool fused_split(const irb_lfft *p, bool forward)
{
    return (p->fuse_split & (forward ? 2 : 1)) && p->M2 > 1 && (p->tile_rows == 4 || p->tile_rows == 8) && p->M1 >= 2 * p->tile_rows;
}

int setup_tiles(irb_lfft *p)
{
    if (p->tile_rows <= 0) return IRB_OK;
    const irb::LfftArgs a = {};
    if (p->M2 > 1 && (p->tile_rows == 4 || p->tile_rows == 8)) {
        FOR_M(p->M2, { int rc = launch_fused<MM, true, true>(p, a, nullptr, nullptr, 0); if (rc != IRB_OK) return rc; });
        FOR_M(p->M1, { int rc = launch_fused<MM, false, true>(p, a, nullptr, nullptr, 0); if (rc != IRB_OK) return rc; });
    }
    FOR_M(p->M1, { int rc = launch_tile<MM, false, true, true>(p, a, nullptr, nullptr, 0); if (rc == IRB_OK) rc = launch_tile<MM, true, true, true>(p, a, nullptr, nullptr, 0); if (rc != IRB_OK) return rc; });
    FOR_M(p->M2, { int rc = launch_tile<MM, false, false, true>(p, a, nullptr, nullptr, 0); if (rc == IRB_OK) rc = launch_tile<MM, true, false, true>(p, a, nullptr, nullptr, 0); if (rc != IRB_OK) return rc; });
    return IRB_OK;
}

// complex M-point transform of `batch` items: src -> dst (dst may equal neither work buffer's alias of src)
template <bool INV>
int complex_fft(irb_lfft *p, const float2 *src, float2 *tmp, float2 *dst, int batch)
{
    irb::LfftArgs a = args_of(p, batch);
    if (INV) a.scale = 1.0f / (float) p->N;             // the last pass of an inverse transform applies 1/N
    if (p->M2 == 1) {
        FOR_M(p->M1, {
            constexpr int NF = irb::WaveGeom<MM>::NF;
            irb::k_lfft_small<MM, INV><<<dim3((batch + NF - 1) / NF), dim3(irb::kWave * NF), 0, p->stream>>>(a, src, dst);
        });
    } else if (p->tile_rows > 0) {
        // tiled passes (coalesced global traffic through LDS); R waves per workgroup
        FOR_M(p->M1, { int rc = launch_tile<MM, INV, true>(p, a, src, tmp, batch * (p->M2 / p->tile_rows)); if (rc != IRB_OK) return rc; });
        FOR_M(p->M2, { int rc = launch_tile<MM, INV, false>(p, a, tmp, dst, batch * (p->M1 / p->tile_rows)); if (rc != IRB_OK) return rc; });
    } else {
        FOR_M(p->M1, {
            constexpr int NF = irb::WaveGeom<MM>::NF;
            const long items = (long) batch * p->M2;
            irb::k_lfft_cols<MM, INV><<<dim3((unsigned) ((items + NF - 1) / NF)), dim3(irb::kWave * NF), 0, p->stream>>>(a, src, tmp);
        });
        FOR_M(p->M2, {
            constexpr int NF = irb::WaveGeom<MM>::NF;
            const long items = (long) batch * p->M1;
            irb::k_lfft_rows<MM, INV><<<dim3((unsigned) ((items + NF - 1) / NF)), dim3(irb::kWave * NF), 0, p->stream>>>(a, tmp, dst);
        });
    }
    HIP_TRY(hipGetLastError());
    return IRB_OK;
}

}  // namespace

// One-shot entry points (fftTransform, deconvolve, ...) reuse the most recent plan of the calling thread when
// device, length and batch capacity fit: plan creation (stream, tables, five hipMallocs) costs far more than a
// 2^20-point transform.  The cached plan is released by irb_release_cached_plans(), or by the cache's destructor when
// the calling thread exits (thread-local objects of the main thread are destroyed at exit() before the HIP runtime's
// static teardown; a pool thread of a plug-in host frees its plan when it ends).
extern "C" void irb_lfft_destroy(irb_lfft *p);
struct PlanCache {
    irb_lfft *plan = nullptr;
    ~PlanCache() { if (plan) { irb_lfft_destroy(plan); plan = nullptr; } }
};
thread_local PlanCache g_cache;

extern "C" int irb_lfft_create(int device, int fft_len, int max_batch, irb_lfft **out);

int cached_plan(int device, int N, int batch, irb_lfft **out)
{
    irb_lfft *p = g_cache.plan;
    if (p && p->device == device && p->N == N && p->max_batch >= batch) { *out = p; return IRB_OK; }
    if (p) { irb_lfft_destroy(p); g_cache.plan = nullptr; }
    int rc = irb_lfft_create(device, N, batch, &p);
    if (rc != IRB_OK) return rc;
    g_cache.plan = p;
    *out = p;
    return IRB_OK;
}

extern "C" {

void irb_fdl_release_cached_engine_(void);     // fdl_engine.hip

void irb_release_cached_plans(void)
{
    irb_fdl_release_cached_engine_();
    if (g_cache.plan) { irb_lfft_destroy(g_cache.plan); g_cache.plan = nullptr; }
}

int irb_next_power_of_two(int x)                       // tools::nextPowerOfTwo, fp/tools.cpp:180-196
{
    if (is_pow2(x)) return x;
    int r = 1;
    while (r <= x) r *= 2;
    return r;
}

void irb_lfft_destroy(irb_lfft *p)
{
    if (!p) return;
    (void) hipSetDevice(p->device);
    if (p->stream) (void) hipStreamSynchronize(p->stream);
    (void) irb_device_quiet_begin(p->device);          // the frees wait for every stream of the device: resident real-time servers step aside
    (void) hipFree(p->d_tw1); (void) hipFree(p->d_tw2); (void) hipFree(p->d_tabA); (void) hipFree(p->d_tabB);
    (void) hipFree(p->d_w0); (void) hipFree(p->d_w1); (void) hipFree(p->d_time); (void) hipFree(p->d_spec);
    if (p->h_log) (void) hipHostFree(p->h_log);
    if (p->h_mean) (void) hipHostFree(p->h_mean);
    if (p->h_stage) (void) hipHostFree(p->h_stage);
    for (auto &e : p->ev_chunk) if (e) (void) hipEventDestroy(e);
    free_windows(p->win);
    for (auto &e : p->ev) if (e) (void) hipEventDestroy(e);
    if (p->stream) (void) hipStreamDestroy(p->stream);
    (void) irb_device_quiet_end(p->device);
    delete p;
}

int irb_lfft_create(int device, int fft_len, int max_batch, irb_lfft **out)
{
    if (!out) return irb_set_error_(IRB_ERR_ARG, "null argument");
    *out = nullptr;
    if (!is_pow2(fft_len) || fft_len < 32 || fft_len > (1 << 23))
        return irb_set_error_(IRB_ERR_ARG, "fft_len must be a power of two in [32, 2^23]");
    if (max_batch < 1) return irb_set_error_(IRB_ERR_ARG, "max_batch must be >= 1");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return irb_set_error_(IRB_ERR_HIP, "HIP device not present");
    HIP_TRY(hipSetDevice(device));
    irb_lfft *p = new (std::nothrow) irb_lfft;
    if (!p) return irb_set_error_(IRB_ERR_ARG, "out of host memory");
    p->device = device; p->N = fft_len; p->M = fft_len / 2; p->max_batch = max_batch;
    const int m = ilog2(p->M);
    if (p->M <= 2048) { p->M1 = p->M; p->M2 = 1; }
    else {
        p->M1 = 1 << (m / 2); p->M2 = p->M / p->M1;
        const int l1 = dbg_int("lfft_log_m1", 0);                         // tuning (IRB_DEBUG, irb_env.hpp): another split of M = M1 * M2 (both factors 16..2048)
        if (l1 >= 4 && l1 <= 11 && m - l1 >= 4 && m - l1 <= 11) { p->M1 = 1 << l1; p->M2 = p->M / p->M1; }
    }
    p->logE = std::min(10, ilog2(p->N));
    if (p->M2 > 1) {
        // rows / columns per workgroup: 8 gives 64-byte global pieces; a single 2^20 transform then has only 64-128
        // workgroups for 256 CUs, so small batches take 4 (32-byte pieces, twice the workgroups: latency-bound anyway)
        p->tile_rows = (long) max_batch * std::min(p->M1, p->M2) / 8 >= 256 ? 8 : 4;
        p->tile_rows = dbg_int("lfft_tile", p->tile_rows);                // tuning: 0, 4, 8, 16
        // fused split passes (2 launches per transform instead of 3) pay while the transform is launch-bound
        // (profiles/r02_lfft_fuse_ab.txt, r03_lfft_ab.txt): both directions up to 2^20 points (round 3: the forward pass takes its
        // split twiddles from the row transform's LDS table and skips the zero fill of the plan's own spectrum buffer)
        p->fuse_split = (p->N <= (1 << 20) ? 1 : 0) | (p->N <= (1 << 20) ? 2 : 0);
        p->fuse_split = dbg_int("lfft_fuse_split", p->fuse_split);
        if (p->tile_rows == 16 && std::max(p->M1, p->M2) > 1024) p->tile_rows = 8;
    }
    const int E = 1 << p->logE;
    int rc = IRB_OK;
    do {
        hipError_t e;
        if ((e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking)) != hipSuccess ||
            (e = hipEventCreate(&p->ev[0])) != hipSuccess || (e = hipEventCreate(&p->ev[1])) != hipSuccess) {
            rc = failf(IRB_ERR_HIP, "%s: %s", "stream/event create", hipGetErrorString(e)); break;
        }
        const auto t1 = root_table(2 * p->M1, p->M1, 1), t2 = root_table(2 * p->M2, p->M2, 1);
        const auto tA = root_table(p->N, p->N / E, E), tB = root_table(p->N, E, 1);
        const size_t cw = sizeof(float2) * (size_t) max_batch * p->M;
        if ((e = hipMalloc(&p->d_tw1, sizeof(float2) * t1.size())) != hipSuccess ||
            (e = hipMalloc(&p->d_tw2, sizeof(float2) * t2.size())) != hipSuccess ||
            (e = hipMalloc(&p->d_tabA, sizeof(float2) * tA.size())) != hipSuccess ||
            (e = hipMalloc(&p->d_tabB, sizeof(float2) * tB.size())) != hipSuccess ||
            (e = hipMalloc(&p->d_w0, cw)) != hipSuccess || (e = hipMalloc(&p->d_w1, cw)) != hipSuccess ||
            (e = hipMalloc(&p->d_time, sizeof(float) * (size_t) max_batch * p->N)) != hipSuccess ||
            (e = hipMalloc(&p->d_spec, sizeof(float) * (size_t) max_batch * 2 * p->N)) != hipSuccess ||
            (e = hipMemcpy(p->d_tw1, t1.data(), sizeof(float2) * t1.size(), hipMemcpyHostToDevice)) != hipSuccess ||
            (e = hipMemcpy(p->d_tw2, t2.data(), sizeof(float2) * t2.size(), hipMemcpyHostToDevice)) != hipSuccess ||
            (e = hipMemcpy(p->d_tabA, tA.data(), sizeof(float2) * tA.size(), hipMemcpyHostToDevice)) != hipSuccess ||
            (e = hipMemcpy(p->d_tabB, tB.data(), sizeof(float2) * tB.size(), hipMemcpyHostToDevice)) != hipSuccess) {
            rc = failf(IRB_ERR_HIP, "%s: %s", "plan allocation", hipGetErrorString(e)); break;
        }
        if ((e = hipMemset(p->d_spec, 0, sizeof(float) * (size_t) max_batch * 2 * p->N)) != hipSuccess) {
            rc = failf(IRB_ERR_HIP, "%s: %s", "plan buffers", hipGetErrorString(e)); break;
        }
        p->spec_upper_zero = true;
        rc = setup_tiles(p);
    } while (0);
    if (rc != IRB_OK) { irb_lfft_destroy(p); return rc; }
    *out = p;
    return IRB_OK;
}

// d_in: [batch][N] real samples (already zero padded); d_spec: [batch][2N] floats, reference layout
int irb_lfft_forward_device(irb_lfft *p, const float *d_in, float *d_spec, int batch)
{
    if (!p || !d_in || !d_spec) return irb_set_error_(IRB_ERR_ARG, "null argument");
    if (batch < 1 || batch > p->max_batch) return irb_set_error_(IRB_ERR_ARG, "batch out of range");
    HIP_TRY(hipSetDevice(p->device));
    const bool fill = !(d_spec == p->d_spec && p->spec_upper_zero);  // 4 MB of zeros per 2^20-point item unless they are known to be there
    if (fill && d_spec == p->d_spec && p->spec_refill && batch == p->max_batch) { p->spec_upper_zero = true; p->spec_refill = false; }   // this call rewrites every float of the buffer
    if (fused_split(p, true)) {                                     // columns pass, then rows pass + split in one launch
        irb::LfftArgs a = args_of(p, batch);
        a.zero_fill = fill;
        FOR_M(p->M1, { int rc = launch_tile<MM, false, true>(p, a, reinterpret_cast<const float2 *>(d_in), p->d_w0, batch * (p->M2 / p->tile_rows)); if (rc != IRB_OK) return rc; });
        FOR_M(p->M2, { int rc = launch_fused<MM, true>(p, a, p->d_w0, d_spec, batch * (p->M1 / p->tile_rows)); if (rc != IRB_OK) return rc; });
        HIP_TRY(hipGetLastError());
        return IRB_OK;
    }
    int rc = complex_fft<false>(p, reinterpret_cast<const float2 *>(d_in), p->d_w0, p->d_w1, batch);
    if (rc != IRB_OK) return rc;
    const long work = (long) batch * (p->M / 2 + 1);
    irb::LfftArgs as = args_of(p, batch);
    as.zero_fill = fill;
    irb::k_rsplit_fwd<<<dim3((unsigned) ((work + 255) / 256)), dim3(256), 0, p->stream>>>(as, p->d_w1, d_spec);
    HIP_TRY(hipGetLastError());
    return IRB_OK;
}

// d_spec: [batch][2N] floats (bins 0..N/2 used); d_out: [batch][N] samples, scaled 1/N
int irb_lfft_inverse_device(irb_lfft *p, const float *d_spec, float *d_out, int batch)
{
    if (!p || !d_spec || !d_out) return irb_set_error_(IRB_ERR_ARG, "null argument");
    if (batch < 1 || batch > p->max_batch) return irb_set_error_(IRB_ERR_ARG, "batch out of range");
    HIP_TRY(hipSetDevice(p->device));
    if (fused_split(p, false)) {                                    // split + columns pass in one launch, then the rows pass (scaled 1/N)
        irb::LfftArgs a = args_of(p, batch);
        a.scale = 1.0f / (float) p->N;
        FOR_M(p->M1, { int rc = launch_fused<MM, false>(p, a, d_spec, p->d_w1, batch * (p->M2 / p->tile_rows)); if (rc != IRB_OK) return rc; });
        FOR_M(p->M2, { int rc = launch_tile<MM, true, false>(p, a, p->d_w1, reinterpret_cast<float2 *>(d_out), batch * (p->M1 / p->tile_rows)); if (rc != IRB_OK) return rc; });
        HIP_TRY(hipGetLastError());
        return IRB_OK;
    }
    const long work = (long) batch * (p->M / 2 + 1);
    irb::k_rsplit_inv<<<dim3((unsigned) ((work + 255) / 256)), dim3(256), 0, p->stream>>>(args_of(p, batch), d_spec, p->d_w0);
    HIP_TRY(hipGetLastError());
    // the last pass writes the N real samples (as M interleaved pairs), already scaled, straight to d_out
    int rc = complex_fft<true>(p, p->d_w0, p->d_w1, reinterpret_cast<float2 *>(d_out), batch);
    if (rc != IRB_OK) return rc;
    HIP_TRY(hipGetLastError());
    return IRB_OK;
}

int irb_lfft_synchronize(irb_lfft *p)
{
    if (!p) return irb_set_error_(IRB_ERR_ARG, "null handle");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return IRB_OK;
}

int irb_lfft_mark(irb_lfft *p, int slot)
{
    if (!p || slot < 0 || slot > 1) return irb_set_error_(IRB_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipEventRecord(p->ev[slot], p->stream));
    return IRB_OK;
}

int irb_lfft_elapsed_ms(irb_lfft *p, float *ms)
{
    if (!p || !ms) return irb_set_error_(IRB_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipEventSynchronize(p->ev[1]));
    HIP_TRY(hipEventElapsedTime(ms, p->ev[0], p->ev[1]));
    return IRB_OK;
}

int irb_lfft_invalidate_spec(irb_lfft *p)
{
    if (!p) return irb_set_error_(IRB_ERR_ARG, "null handle");
    p->spec_upper_zero = false;                        // the next forward transform into the plan's spectrum buffer writes all 2N floats again ...
    p->spec_refill = true;                             // ... and re-arms the short form behind it
    return IRB_OK;
}

int irb_lfft_buffers(irb_lfft *p, float **d_time, float **d_spec)
{
    if (!p || !d_time || !d_spec) return irb_set_error_(IRB_ERR_ARG, "null argument");
    *d_time = p->d_time; *d_spec = p->d_spec;
    return IRB_OK;
}

// ---- tools::fftTransform (fp/tools.cpp:321-346) ---------------------------------------------------------
static int fft_transform_impl(int device, const float *in, int len, float *out, bool ampl_phase)
{
    if (!in || !out || len < 1) return irb_set_error_(IRB_ERR_ARG, "bad argument");
    const int N = std::max(32, irb_next_power_of_two(len));
    if (N != irb_next_power_of_two(len)) return irb_set_error_(IRB_ERR_ARG, "buffers shorter than 17 samples are not supported");
    irb_lfft *p = nullptr;
    int rc = cached_plan(device, N, 1, &p);
    if (rc != IRB_OK) return rc;
    hipError_t e = hipMemsetAsync(p->d_time, 0, sizeof(float) * N, p->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(p->d_time, in, sizeof(float) * len, hipMemcpyHostToDevice, p->stream);
    if (e == hipSuccess) { rc = irb_lfft_forward_device(p, p->d_time, p->d_spec, 1); }
    if (e == hipSuccess && rc == IRB_OK && ampl_phase) {            // :337-342, bins 0..N/2 (`bin <= N; bin += 2`)
        const int nbins = N / 2 + 1;
        irb::k_ampl_phase<<<dim3((nbins + 255) / 256), dim3(256), 0, p->stream>>>(reinterpret_cast<float2 *>(p->d_spec), nbins);
        e = hipGetLastError();
    }
    if (e == hipSuccess && rc == IRB_OK) e = hipMemcpyAsync(out, p->d_spec, sizeof(float) * 2 * N, hipMemcpyDeviceToHost, p->stream);
    if (e == hipSuccess && rc == IRB_OK) e = hipStreamSynchronize(p->stream);
    if (e != hipSuccess) rc = failf(IRB_ERR_HIP, "%s: %s", "irb_fft_transform", hipGetErrorString(e));
    return rc;
}

int irb_fft_transform(int device, const float *in, int len, float *out) { return fft_transform_impl(device, in, len, out, false); }
int irb_fft_transform_ampl_phase(int device, const float *in, int len, float *out) { return fft_transform_impl(device, in, len, out, true); }

// ---- tools::fftInvTransform (fp/tools.cpp:351-369): fft_size = 2N floats in, N samples out --------------
int irb_fft_inv_transform(int device, const float *in, int fft_size, float *out)
{
    if (!in || !out || fft_size < 64 || !is_pow2(fft_size)) return irb_set_error_(IRB_ERR_ARG, "fft_size must be a power of two >= 64");
    const int N = fft_size / 2;
    irb_lfft *p = nullptr;
    int rc = cached_plan(device, N, 1, &p);
    if (rc != IRB_OK) return rc;
    // only bins 0..N/2 are read by the inverse transform (performRealOnlyInverseTransform ignores the rest): copy just those,
    // which also keeps the upper half of the plan's spectrum buffer at zero
    hipError_t e = hipMemcpyAsync(p->d_spec, in, sizeof(float) * (size_t) (N + 2), hipMemcpyHostToDevice, p->stream);
    if (e == hipSuccess) rc = irb_lfft_inverse_device(p, p->d_spec, p->d_time, 1);
    if (e == hipSuccess && rc == IRB_OK) e = hipMemcpyAsync(out, p->d_time, sizeof(float) * N, hipMemcpyDeviceToHost, p->stream);
    if (e == hipSuccess && rc == IRB_OK) e = hipStreamSynchronize(p->stream);
    if (e != hipSuccess) rc = failf(IRB_ERR_HIP, "%s: %s", "irb_fft_inv_transform", hipGetErrorString(e));
    return rc;
}

// ---- host-resident IR preparation ----------------------------------------------------------------------
//
// Fractional-octave smoothing (convolution::averagingFilter, fp/convolution.cpp:406-546) in three steps over the
// complex bins c = 0..N/2 of an N-point real transform:
//   1. every bin's window [c / r, c * r], r = 2^(fraction/2), rounded to whole bins (octave_windows);
//   2. the mean over the window of log|X| (or of |X|), the window running past Nyquist where the spectrum is 0;
//   3. the spectrum rebuilt from the new amplitude and the old phase (rebuild_bin).
// Results must equal the reference's float for float, and step 2 is where that bites: the reference keeps ONE
// float accumulator that slides along the spectrum (entering bins added, leaving bins subtracted, in that order),
// so its rounding depends on the whole history.  sliding_mean reproduces exactly that accumulation order; it is
// the reason this stays a sequential host loop (a parallel prefix sum in double differs from the float history
// by up to 1e-3 at 2^20 points: tests/test_host_code.py::test_sliding_float_sum_drift).

}  // extern "C" (the helpers below are C++)

namespace smoothing {

constexpr float kAmplFloor = 1e-16f;                                // tools::roundTo1TenQuadrillionth, fp/tools.cpp:213-218
constexpr float kZeroBelow = 1e-11f;                                // tools::roundToZero threshold, fp/convolution.cpp:534-536

inline float floor_magnitude(float v)                               // |v| < 1e-16 -> +-1e-16, sign kept (+0 counts as positive)
{
    // the reference compares the float with the DOUBLE constant 1e-16 and stores the float nearest to it
    if (std::signbit(v)) return (double) v > -1e-16 ? -kAmplFloor : v;
    return (double) v < 1e-16 ? kAmplFloor : v;
}

inline float flush_small(float v)                                   // |v| < 1e-11 -> +0
{
    if (std::signbit(v)) return v > -kZeroBelow ? 0.0f : v;
    return v < kZeroBelow ? 0.0f : v;
}

struct Windows {
    std::vector<int> first, last;                                   // per bin: the window's bounds, in bins (last may exceed N/2)
};

Windows octave_windows(int nbins, double hz_per_bin, double octave_fraction)      // fp/convolution.cpp:451-458
{
    Windows w;
    w.first.resize((size_t) nbins);
    w.last.resize((size_t) nbins);
    const double ratio = std::pow(2.0, octave_fraction / 2.0);      // half of the fraction to either side
    for (int c = 0; c < nbins; c++) {
        const double centre_hz = c * hz_per_bin;
        w.first[(size_t) c] = (int) std::round((centre_hz / ratio) / hz_per_bin);
        w.last[(size_t) c] = (int) std::round((centre_hz * ratio) / hz_per_bin);
    }
    return w;
}

// mean[c] = (sum of value(j), j in window c) / window length, with the reference's accumulation order (:482-505,519):
// one float accumulator; for each bin first the values that left the window are subtracted (ascending), then the
// values that entered are added (ascending).  value(j) = v[j] for j < have, `pad` for have <= j < limit (bins between
// Nyquist and the end of the reference's 2N-float buffer hold amplitude 0), and bins >= limit are not added at all
// although they count in the window length (:497).
void sliding_mean(const float *v, int have, int limit, float pad, const Windows &w, int nbins, float *mean)
{
    float acc = 0.0f;
    int left = 0, right = -1;                                       // the window the accumulator currently covers: [left, right]
    for (int c = 0; c < nbins; c++) {
        const int first = w.first[(size_t) c], last = w.last[(size_t) c];
        for (; left < first; left++) acc -= left < have ? v[left] : pad;
        for (int j = right + 1; j <= last; j++)
            if (j < limit) acc += j < have ? v[j] : pad;
        right = last;
        mean[c] = (float) (acc / ((double) (last - first) + 1.0));
    }
}

// the linear-average variant restarts its sum for every bin (:507-516)
void restarted_mean(const float *v, int limit, const Windows &w, int nbins, float *mean)
{
    for (int c = 0; c < nbins; c++) {
        const int first = w.first[(size_t) c], last = w.last[(size_t) c];
        float acc = 0.0f;
        for (int j = first; j <= last && j < limit; j++) acc += v[j];
        mean[c] = (float) (acc / ((double) (last - first) + 1.0));
    }
}

}  // namespace smoothing

// window bounds kept with a plan between smoothing calls (see smooth_on_device)
struct SmoothWindows {
    int nbins = 0;
    double hz_per_bin = 0.0, octave_fraction = 0.0;
    smoothing::Windows w;
};

void free_windows(SmoothWindows *w) { delete w; }

extern "C" {

// convolution::averagingFilter (fp/convolution.cpp:406-546), one channel, in place, HOST code.
// buffer: fft_size = 2N floats, interleaved (re, im) for bins 0..N/2.
void irb_averaging_filter(float *buffer, int fft_size, double octave_fraction, double sample_rate, int log_avg,
                          int include_phase, int include_amplitude)
{
    if (!buffer || !is_pow2(fft_size)) return;                      // :412-415 (DBG + return)
    const int n = fft_size / 2, nbins = n / 2 + 1;                  // transform length, bins 0..N/2
    const smoothing::Windows win = smoothing::octave_windows(nbins, (sample_rate / 2) / (double) (n / 2), octave_fraction);
    float2 *bins = reinterpret_cast<float2 *>(buffer);
    std::vector<float> ampl((size_t) n, 0.0f), mean((size_t) nbins);     // amplitudes of bins 0..N-1: zero above Nyquist
    for (int c = 0; c < nbins; c++)                                 // tools::binAmpl: double arithmetic, float result
        ampl[(size_t) c] = (float) std::sqrt(std::pow((double) bins[c].x, 2.0) + std::pow((double) bins[c].y, 2.0));
    if (log_avg) {
        for (int c = 0; c < nbins; c++) ampl[(size_t) c] = std::log(smoothing::floor_magnitude(ampl[(size_t) c]));
        smoothing::sliding_mean(ampl.data(), nbins, n, std::log(smoothing::kAmplFloor), win, nbins, mean.data());
        for (int c = 0; c < nbins; c++) mean[(size_t) c] = std::exp(mean[(size_t) c]);
    } else {
        smoothing::restarted_mean(ampl.data(), n, win, nbins, mean.data());
    }
    for (int c = 0; c < nbins; c++) {                               // :531-543
        const float magnitude = include_amplitude ? smoothing::floor_magnitude(mean[(size_t) c]) : 1.0f;
        const float re = smoothing::flush_small(bins[c].x), im = smoothing::flush_small(bins[c].y);
        const float phase = include_phase ? std::atan2(im, re) : 0.0f;
        bins[c] = make_float2(magnitude * std::cos(phase), magnitude * std::sin(phase));
    }
}

void irb_shifteroo(float *buf, int n)                               // fp/ir.cpp:85-103
{
    if (!buf || n < 2) return;
    const int second = n / 2, first = second + (n % 2 != 0 ? 1 : 0);
    std::vector<float> t((size_t) n);
    std::memcpy(t.data(), buf + first, sizeof(float) * (size_t) second);
    std::memcpy(t.data() + second, buf, sizeof(float) * (size_t) first);
    std::memcpy(buf, t.data(), sizeof(float) * (size_t) n);
}

// averagingFilter with its element-wise phases on the GPU (log amplitude; exp / atan2 / cos / sin) and only
// the order-sensitive float running sum on the host, on a spectrum that stays in device memory.  `passes`
// repetitions (deconvolve uses 3).  Same operation order as irb_averaging_filter; the transcendental
// functions are the device's, so results agree to rounding (1e-6), not bit for bit.
// Per pass: k_smooth_log_ampl writes the log amplitudes straight into mapped pinned host memory, ONE stream wait, the
// sequential sliding sum on the host (pinned in, pinned out), k_smooth_apply reads the means from the mapped buffer --
// no copy commands and no staging through pageable memory (round 2: two copies through pageable vectors and two waits
// per pass).  The window bounds (two divisions and two roundings in double per bin, ~1.5 ms at 2^19 bins) are computed
// once per (bins, Hz per bin, octave fraction) and kept with the plan.
static int smooth_on_device(irb_lfft *p, float *d_spec, int N, double octave_fraction, double sample_rate,
                            int passes, int include_phase, int include_amplitude)
{
    const int nbins = N / 2 + 1;
    const double hz_per_bin = (sample_rate / 2) / (double) (N / 2);
    hipError_t e = hipSuccess;
    if (!p->h_log) {
        if ((e = hipHostMalloc(reinterpret_cast<void **>(&p->h_log), sizeof(float) * (size_t) (p->N / 2 + 1), hipHostMallocMapped)) != hipSuccess ||
            (e = hipHostMalloc(reinterpret_cast<void **>(&p->h_mean), sizeof(float) * (size_t) (p->N / 2 + 1), hipHostMallocMapped)) != hipSuccess ||
            (e = hipHostGetDevicePointer(reinterpret_cast<void **>(&p->d_hlog), p->h_log, 0)) != hipSuccess ||
            (e = hipHostGetDevicePointer(reinterpret_cast<void **>(&p->d_hmean), p->h_mean, 0)) != hipSuccess)
            return failf(IRB_ERR_HIP, "%s: %s", "smoothing workspace", hipGetErrorString(e));
    }
    if (!p->win || p->win->nbins != nbins || p->win->hz_per_bin != hz_per_bin || p->win->octave_fraction != octave_fraction) {
        free_windows(p->win);
        p->win = new (std::nothrow) SmoothWindows;
        if (!p->win) return irb_set_error_(IRB_ERR_ARG, "out of host memory");
        p->win->nbins = nbins; p->win->hz_per_bin = hz_per_bin; p->win->octave_fraction = octave_fraction;
        p->win->w = smoothing::octave_windows(nbins, hz_per_bin, octave_fraction);
    }
    const smoothing::Windows &win = p->win->w;
    const float pad = std::log(smoothing::kAmplFloor);              // bins above Nyquist count as amplitude 0 -> 1e-16
    for (int pass = 0; pass < passes; pass++) {
        irb::k_smooth_log_ampl<<<dim3((nbins + 255) / 256), dim3(256), 0, p->stream>>>(reinterpret_cast<const float2 *>(d_spec), p->d_hlog, nbins);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
        if (e != hipSuccess) return failf(IRB_ERR_HIP, "%s: %s", "smoothing (log amplitudes)", hipGetErrorString(e));
        smoothing::sliding_mean(p->h_log, nbins, N, pad, win, nbins, p->h_mean);              // the order-sensitive float history
        irb::k_smooth_apply<<<dim3((nbins + 255) / 256), dim3(256), 0, p->stream>>>(reinterpret_cast<float2 *>(d_spec), p->d_hmean, nbins, 1,
                                                                                      include_phase, include_amplitude);
        if ((e = hipGetLastError()) != hipSuccess) return failf(IRB_ERR_HIP, "%s: %s", "k_smooth_apply", hipGetErrorString(e));
    }
    // h_mean is rewritten by the next pass only after that pass's own wait, which is stream-ordered behind this apply kernel;
    // the caller's next wait (download of the result) covers the last one
    return IRB_OK;
}

// ---- convolution::deconvolve (fp/convolution.cpp:351-403) ---------------------------------------------
int irb_deconvolve(int device, const float *num, int len_num, const float *den, int len_den, double sample_rate,
                   int smoothing, int include_phase, int include_amplitude, float *out, int out_len)
{
    if (!num || !den || !out || len_num < 1 || len_den < 1) return irb_set_error_(IRB_ERR_ARG, "bad argument");
    const int len = std::max(len_num, len_den);                     // :359-362 the shorter one is zero extended
    const int N = irb_next_power_of_two(len);                       // fp/tools.cpp:323
    if (N < 32) return irb_set_error_(IRB_ERR_ARG, "buffers shorter than 17 samples are not supported");
    if (out_len < N) return irb_set_error_(IRB_ERR_ARG, "out must hold nextPowerOfTwo(max(len)) samples");
    irb_lfft *p = nullptr;
    int rc = cached_plan(device, N, 2, &p);
    if (rc != IRB_OK) return rc;
    do {
        hipError_t e = hipMemsetAsync(p->d_time, 0, sizeof(float) * 2 * (size_t) N, p->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(p->d_time, num, sizeof(float) * len_num, hipMemcpyHostToDevice, p->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(p->d_time + N, den, sizeof(float) * len_den, hipMemcpyHostToDevice, p->stream);
        if (e != hipSuccess) { rc = failf(IRB_ERR_HIP, "%s: %s", "upload", hipGetErrorString(e)); break; }
        rc = irb_lfft_forward_device(p, p->d_time, p->d_spec, 2);   // :365-366 both transforms in one batch
        if (rc != IRB_OK) break;
        const int nbins = N / 2 + 1;                                // :373-378 bins 0..N/2
        irb::k_cdiv<<<dim3((nbins + 255) / 256), dim3(256), 0, p->stream>>>(
            reinterpret_cast<float2 *>(p->d_spec), reinterpret_cast<const float2 *>(p->d_spec + 2 * (size_t) N), nbins);
        if ((e = hipGetLastError()) != hipSuccess) { rc = failf(IRB_ERR_HIP, "%s: %s", "k_cdiv", hipGetErrorString(e)); break; }
        if (smoothing) {                                            // :389-394: three passes of averagingFilter(1/13 octave)
            const float smoothPerAvg = 1.0 / 13.0;
            rc = smooth_on_device(p, p->d_spec, N, smoothPerAvg, sample_rate, 3, include_phase, include_amplitude);
            if (rc != IRB_OK) break;
        }
        rc = irb_lfft_inverse_device(p, p->d_spec, p->d_time, 1);   // :397
        if (rc != IRB_OK) break;
        e = hipMemcpyAsync(out, p->d_time, sizeof(float) * N, hipMemcpyDeviceToHost, p->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
        if (e != hipSuccess) { rc = failf(IRB_ERR_HIP, "%s: %s", "download", hipGetErrorString(e)); break; }
        if (!include_phase) irb_shifteroo(out, N);                  // :400
    } while (0);
    return rc;
}

// ---- ir::invertFilter (fp/ir.cpp:13-18): deconvolve(generatePulse(n), ir) with the default flags -------
int irb_invert_filter(int device, const float *ir, int len, int sample_rate, float *out, int out_len)
{
    if (!ir || len < 1) return irb_set_error_(IRB_ERR_ARG, "bad argument");
    std::vector<float> pulse((size_t) len, 0.0f);
    pulse[0] = 1.0f;                                                // tools::generatePulse(n, 0), fp/tools.cpp:235-241
    return irb_deconvolve(device, pulse.data(), len, ir, len, (double) sample_rate, 1, 1, 1, out, out_len);
}

// ---- fp::ExpSineSweep (host) ---------------------------------------------------------------------------
// Farina's exponential sine sweep x[i] = sin(K (e^(i/L) - 1)) over T = rate * duration samples between the angular
// frequencies w_lo and w_hi (radians per sample): L = T / ln(w_hi / w_lo) is the e-folding time of the instantaneous
// frequency and K = w_lo L its phase scale.  The frequency doubles every T / log2(w_hi / w_lo) samples, which is all
// the index <-> frequency maps use.  Arithmetic follows fp/ExpSineSweep.cpp operation for operation (doubles), so
// that generated sweeps are the reference's sample for sample.
namespace {

struct FarinaSweep {
    double rate, samples, w_lo, w_hi;

    static bool valid(double duration, double rate, double f_lo, double f_hi) { return duration > 0 && rate > 0 && f_lo > 0 && f_hi > f_lo; }

    FarinaSweep(double duration, double rate_, double f_lo, double f_hi)                          // fp/ExpSineSweep.cpp:212-220
        : rate(rate_), samples(rate_ * duration), w_lo(f_lo / rate_ * 2 * M_PI), w_hi(f_hi / rate_ * 2 * M_PI) {}

    int length() const { return (int) samples; }
    double efold() const { return samples / std::log(w_hi / w_lo); }                              // L
    double phase_scale() const { return samples * w_lo / std::log(w_hi / w_lo); }                 // K
    double octaves() const { return std::log(w_hi / w_lo) / std::log(2); }
    double start_hz() const { return w_lo * rate / (2 * M_PI); }

    void render(double gain, double *out) const                                                   // :26-41
    {
        const double K = phase_scale(), L = efold();
        for (int i = 0, n = length(); i < n; i++) out[i] = gain * std::sin(K * (std::exp((double) i / L) - 1.0));
    }
    // time-reversed sweep with a -6 dB/octave envelope: the matched inverse filter (:59-79)
    void render_inverse(double gain, double *out) const
    {
        const int n = length();
        render(gain, out);
        std::reverse(out, out + n);
        const double per_sample = std::pow(10.0, (-6.0 * std::log2(w_hi / w_lo)) / 20.0 / samples);
        double envelope = per_sample;
        for (int i = 0; i < n; i++) { out[i] *= envelope; envelope *= per_sample; }
    }
    int index_of(double hz) const                                                                 // :243-258; -1: not in this sweep
    {
        const int i = (int) std::round(std::log(hz / start_hz()) / std::log(2) / octaves() * samples);
        return (i < 0 || i >= samples) ? -1 : i;
    }
    double hz_at(int index) const { return start_hz() * std::pow(2.0, (double) index / samples * octaves()); }   // :225-240
};

}  // namespace

int irb_sweep_generate(double duration_secs, double sample_rate, double low_freq, double high_freq, double dB_gain,
                       double *out, int out_len)
{
    if (!out || !FarinaSweep::valid(duration_secs, sample_rate, low_freq, high_freq)) return irb_set_error_(IRB_ERR_ARG, "bad sweep parameters");
    const FarinaSweep sw(duration_secs, sample_rate, low_freq, high_freq);
    if (out_len < sw.length()) return irb_set_error_(IRB_ERR_ARG, "out too small for the sweep");
    sw.render(std::pow(10.0, dB_gain / 20.0), out);                                               // tools::dBToLin
    return sw.length();
}

int irb_sweep_generate_inv(double duration_secs, double sample_rate, double low_freq, double high_freq, double dB_gain,
                           double *out, int out_len)
{
    if (!out || !FarinaSweep::valid(duration_secs, sample_rate, low_freq, high_freq)) return irb_set_error_(IRB_ERR_ARG, "bad sweep parameters");
    const FarinaSweep sw(duration_secs, sample_rate, low_freq, high_freq);
    if (out_len < sw.length()) return irb_set_error_(IRB_ERR_ARG, "out too small for the sweep");
    sw.render_inverse(std::pow(10.0, dB_gain / 20.0), out);
    return sw.length();
}

int irb_sweep_sample_index_at_freq(double freq, double duration_secs, double sample_rate, double low_freq, double high_freq)
{
    if (!FarinaSweep::valid(duration_secs, sample_rate, low_freq, high_freq) || !(freq > 0)) return -1;
    return FarinaSweep(duration_secs, sample_rate, low_freq, high_freq).index_of(freq);
}

double irb_sweep_freq_at_sample_index(int index, double duration_secs, double sample_rate, double low_freq, double high_freq)
{
    if (!FarinaSweep::valid(duration_secs, sample_rate, low_freq, high_freq)) return -1;
    const FarinaSweep sw(duration_secs, sample_rate, low_freq, high_freq);
    return (index < 0 || index >= sw.length()) ? -1 : sw.hz_at(index);
}

// ExpSineSweep::linFadeout (:147-164): linear ramp to zero from sample `index` to the end
int irb_sweep_lin_fadeout(double *sweep, int n, int index)
{
    if (!sweep || n < 0) return irb_set_error_(IRB_ERR_ARG, "bad argument");
    if (n == 0 || index < 0 || index >= n) return IRB_OK;                                         // nothing to fade
    const int span = n - index;
    for (int i = index; i < n; i++) sweep[i] = sweep[i] * (span - (i - index)) / span;
    return IRB_OK;
}

// ---- convolution::convolveNonPeriodic (fp/convolution.cpp:246-347) ------------------------------------
void irb_copy_stream_(void *dst, const void *src, size_t bytes);   // host_copy.cpp

int irb_convolve_nonperiodic(int device, const float *const *audio, int audio_ch, int La,
                             const float *const *ir, int ir_ch, int Lh, float *const *out)
{
    if (!audio || !ir || !out || La < 0 || Lh < 0 || audio_ch < 1) return irb_set_error_(IRB_ERR_ARG, "bad argument");
    const long Lout = (long) La + Lh - 1;                             // :251
    for (int c = 0; c < audio_ch; c++) if (!out[c]) return irb_set_error_(IRB_ERR_ARG, "null output channel");
    auto clear_out = [&]() { for (int c = 0; c < audio_ch; c++) if (Lout > 0) std::memset(out[c], 0, sizeof(float) * (size_t) Lout); };
    if (!((ir_ch == 1 || ir_ch == 2) && (audio_ch == 1 || audio_ch == 2))) {   // :262-275: a cleared buffer + the debug line
        clear_out();
        return irb_set_error_(IRB_ERR_LAYOUT, "Either buffer1 or buffer2 is not mono nor stereo. Abort abort");
    }
    if (Lout <= 0 || La == 0 || Lh == 0) { clear_out(); return IRB_OK; }
    if (Lout > (1 << 23)) { clear_out(); return irb_set_error_(IRB_ERR_ARG, "result longer than 2^23 samples"); }
    int N = 1;
    while (N < Lout) N *= 2;                                          // :278-281
    N = std::max(N, 32);                                              // longer zero padding: same linear convolution
    const int ir_rows = (ir_ch == 2 && audio_ch == 2) ? 2 : 1;        // :296-304
    const int batch = audio_ch + ir_rows;
    irb_lfft *p = nullptr;
    int rc = cached_plan(device, N, batch, &p);
    if (rc != IRB_OK) { clear_out(); return rc; }
    // the device side, the same for both forms: zero padding = one device memset in front of the uploads, then transform, multiply, inverse
    const int nbins = N / 2 + 1;
    auto multiply_and_invert = [&]() -> int {
        int r = irb_lfft_forward_device(p, p->d_time, p->d_spec, batch);
        for (int c = 0; c < audio_ch && r == IRB_OK; c++) {
            const int irc = (ir_rows == 2) ? c : 0;                   // :318-323
            irb::k_cmul<<<dim3((nbins + 255) / 256), dim3(256), 0, p->stream>>>(
                reinterpret_cast<float2 *>(p->d_spec + (size_t) c * 2 * N),
                reinterpret_cast<const float2 *>(p->d_spec + (size_t) (audio_ch + irc) * 2 * N), nbins);
        }
        hipError_t e;
        if (r == IRB_OK && (e = hipGetLastError()) != hipSuccess) r = failf(IRB_ERR_HIP, "%s: %s", "k_cmul", hipGetErrorString(e));
        if (r == IRB_OK) r = irb_lfft_inverse_device(p, p->d_spec, p->d_time, audio_ch);
        return r;
    };
    // ---- direct form (like irb_convolve_periodic's): the copy engine moves the samples straight between the caller's buffers and the
    // device, no host copy at all.  ONLY for buffers that are HIP host memory already (irb_host_alloc, hipHostMalloc, registered by
    // the caller), or with IRB_OFFLINE_DIRECT=1, which registers pageable buffers for the call (exposure: INTEGRATION.md section 1).
    // Not for the stereo-IR-to-mono-audio layout (its average is taken on the host below).
    const DirectMode mode = direct_mode_from_env();
    if (mode != DirectMode::kNever && !(ir_ch == 2 && audio_ch == 1)) {
        struct Quiet { int d; bool on; ~Quiet() { if (on) (void) irb_device_quiet_end(d); } } quiet{device, false};
        if (mode == DirectMode::kRegister) { (void) irb_device_quiet_begin(device); quiet.on = true; }   // registration is a device-wide operation for the runtime (irb_fdl.h)
        HostRanges ranges;
        ranges.add(audio, audio_ch, (size_t) La);
        ranges.add(ir, ir_rows, (size_t) Lh);
        ranges.add(out, audio_ch, (size_t) Lout);
        if (ranges.commit(mode == DirectMode::kRegister)) {
            hipError_t e = hipMemsetAsync(p->d_time, 0, sizeof(float) * (size_t) batch * N, p->stream);
            for (int c = 0; c < audio_ch && e == hipSuccess; c++)
                e = hipMemcpyAsync(p->d_time + (size_t) c * N, audio[c], sizeof(float) * (size_t) La, hipMemcpyHostToDevice, p->stream);
            for (int r = 0; r < ir_rows && e == hipSuccess; r++)
                e = hipMemcpyAsync(p->d_time + (size_t) (audio_ch + r) * N, ir[r], sizeof(float) * (size_t) Lh, hipMemcpyHostToDevice, p->stream);
            if (e != hipSuccess) rc = failf(IRB_ERR_HIP, "%s: %s", "upload", hipGetErrorString(e));
            if (rc == IRB_OK) rc = multiply_and_invert();
            for (int c = 0; c < audio_ch && rc == IRB_OK; c++)        // :340-344: the first La + Lh - 1 samples
                if ((e = hipMemcpyAsync(out[c], p->d_time + (size_t) c * N, sizeof(float) * (size_t) Lout, hipMemcpyDeviceToHost, p->stream)) != hipSuccess)
                    rc = failf(IRB_ERR_HIP, "%s: %s", "download", hipGetErrorString(e));
            e = hipStreamSynchronize(p->stream);                      // also on failure: nothing may still touch the caller's memory
            ranges.release();
            if (rc == IRB_OK && e != hipSuccess) rc = failf(IRB_ERR_HIP, "%s: %s", "convolveNonPeriodic", hipGetErrorString(e));
            if (rc != IRB_OK) clear_out();
            return rc;
        }
        ranges.release();
    }
    // ---- staged form (the default for pageable buffers): through the plan's pinned staging.  Every row is copied there and its
    // upload enqueued at once, so the copy engine moves row k while the host copies row k + 1; the result comes back in chunks of
    // ~1 MiB, each copied out to the caller's buffer (streaming stores) while the next is still on the link.
    do {
        hipError_t e = hipSuccess;
        if (!p->h_stage) {
            e = hipHostMalloc(reinterpret_cast<void **>(&p->h_stage), sizeof(float) * (size_t) p->max_batch * N);
            for (auto &ev : p->ev_chunk) if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e != hipSuccess) { rc = failf(IRB_ERR_HIP, "%s: %s", "pinned staging", hipGetErrorString(e)); break; }
        }
        e = hipMemsetAsync(p->d_time, 0, sizeof(float) * (size_t) batch * N, p->stream);
        for (int c = 0; c < audio_ch && e == hipSuccess; c++) {
            float *row = p->h_stage + (size_t) c * N;
            std::memcpy(row, audio[c], sizeof(float) * (size_t) La);
            e = hipMemcpyAsync(p->d_time + (size_t) c * N, row, sizeof(float) * (size_t) La, hipMemcpyHostToDevice, p->stream);
        }
        for (int r = 0; r < ir_rows && e == hipSuccess; r++) {
            float *row = p->h_stage + (size_t) (audio_ch + r) * N;
            std::memcpy(row, ir[r], sizeof(float) * (size_t) Lh);
            if (ir_ch == 2 && audio_ch == 1)                          // tools::sumToMono, fp/tools.cpp:25-28
                for (int i = 0; i < Lh; i++) { row[i] += ir[1][i]; row[i] /= 2.0f; }
            e = hipMemcpyAsync(p->d_time + (size_t) (audio_ch + r) * N, row, sizeof(float) * (size_t) Lh, hipMemcpyHostToDevice, p->stream);
        }
        if (e != hipSuccess) { rc = failf(IRB_ERR_HIP, "%s: %s", "upload", hipGetErrorString(e)); break; }
        rc = multiply_and_invert();
        if (rc != IRB_OK) break;
        // :340-344: the first La + Lh - 1 samples of every channel, in at most 8 chunks over all channels
        const int per_ch = std::max(1, 8 / audio_ch);
        const long chunk = std::max<long>(((Lout + per_ch - 1) / per_ch + 63) / 64 * 64, 1 << 16);
        struct Piece { int c; long off, n; };
        Piece pieces[8];
        int np = 0;
        for (int c = 0; c < audio_ch && e == hipSuccess; c++)
            for (long off = 0; off < Lout && e == hipSuccess && np < 8; off += chunk) {
                const long n = std::min<long>(chunk, Lout - off);
                e = hipMemcpyAsync(p->h_stage + (size_t) c * N + off, p->d_time + (size_t) c * N + off, sizeof(float) * (size_t) n, hipMemcpyDeviceToHost, p->stream);
                if (e == hipSuccess) e = hipEventRecord(p->ev_chunk[np], p->stream);
                pieces[np++] = {c, off, n};
            }
        for (int k = 0; k < np && e == hipSuccess; k++) {
            e = hipEventSynchronize(p->ev_chunk[k]);
            if (e == hipSuccess) irb_copy_stream_(out[pieces[k].c] + pieces[k].off, p->h_stage + (size_t) pieces[k].c * N + pieces[k].off, sizeof(float) * (size_t) pieces[k].n);
        }
        if (e != hipSuccess) { (void) hipStreamSynchronize(p->stream); rc = failf(IRB_ERR_HIP, "%s: %s", "download", hipGetErrorString(e)); break; }
    } while (0);
    if (rc != IRB_OK) clear_out();
    return rc;
}

}  // extern "C"
