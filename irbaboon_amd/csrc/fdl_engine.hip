// fdl_engine.hip -- host side of libirbfdl.so: the C ABI of include/irb_fdl.h on top of the
// gfx950 kernels in fdl_kernels.hpp.  Pure HIP runtime + C++17; no torch, no JUCE.
//
// There is deliberately NO CPU fallback in this library: every entry point that computes
// needs a HIP device and reports IRB_ERR_HIP when there is none.
#include "../../include/irb_fdl.h"
#include "fdl_kernels.hpp"
#include "host_ranges.hpp"
#include "irb_env.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(IRB_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

int fft_len_for_block(int B)
{
    int N = 1;
    while (N < 2 * B - 1) N *= 2;            // fp/convolution.cpp:45-48
    return std::max(N, 32);                  // any power of two >= 2B-1 gives the same linear convolution
}

}  // namespace

struct irb_fdl {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev[2] = {nullptr, nullptr};
    int I = 0, C = 0, S = 0, B = 0, N = 0, M = 0, P = 0, R = 0, Tmax = 1, splits = 1;
    int Ppad = 0;                            // H rows allocated per set: P rounded up to the time tile
    bool tiled1 = false;                     // one-bin-per-lane tiled MAC for HBM-sized batches (override: IRB_TILED1)
    bool tile8_fused = true;                 // calls of <= 8 blocks with one workgroup per stream: one launch (k_tile8_fused; IRB_TILED_FUSED8=0: the three-launch form)
    int mfma_min_T = 256;                    // calls of at least this many blocks multiply on the matrix cores (k_cmac_mfma); 0 = never (IRB_TILED_MFMA_MIN_T)
    int mfma_min_chunks = 6;                 // Toeplitz chunks (16 partitions each) a split keeps at least (IRB_TILED_MFMA_MIN_CHUNKS)
    int mfma_wgs = 256;                      // workgroups its partition splits aim for (IRB_TILED_MFMA_WGS)
    int tiled_chip_wgs = 768;                // workgroups the partition splits aim for (IRB_TILED_WGS)
    bool pair = false;                       // both channels of every instance share their IR set: paired kernel
    bool nt = false;                         // non-temporal row loads in the fused kernel
    int fused_u = 2;                         // partitions per stripe whose rows a thread of k_step_fused keeps in flight (2 or 4)
    int colW = 0;                            // > 0: one-block steps use k_step_cols with slices of colW float4 columns,
                                             // `splits` then counts partition splits (workgroups per stream = CS * splits)
    int cols_handoff = 0;                    // its hand-off (k_step_cols HO): 0 write-through slices + ticket, 1 fences + ticket, 2 tagged granules
    unsigned long long *d_gran = nullptr;    // [S][splits][N] granules of hand-off form 2
    unsigned *h_status = nullptr, *d_status = nullptr;   // host-mapped word a kernel raises when a bounded spin gives up
    unsigned *h_done = nullptr, *d_done = nullptr;       // host-mapped completion word of the one-launch real-time callback (k_rt_pair)
    // IRB_FDL_RESIDENT: the real-time callback served by a resident workgroup (k_rt_server) instead of a launch per callback
    bool resident = false, server_live = false;
    irb::RtMailbox *bar_mb = nullptr;                    // request mailbox, in the BAR-mapped allocation below
    hipStream_t rt_stream = nullptr;                     // the server's own stream: nothing else is ever enqueued behind it
    unsigned server_gen = 0, idle_ticks = 0, life_ticks = 0;   // generation of the running server; its timeouts (100 MHz ticks)
    float *bar_in = nullptr, *bar_ir = nullptr;          // IRB_FDL_HOST_IO on a large-BAR system: fine-grained DEVICE memory the audio thread stores its
                                                         // blocks [S][Tmax*B] and IR partitions [P*B] into directly (posted writes over PCIe), so that
                                                         // k_rt_pair's first loads are local instead of PCIe round trips (tools/bar_probe.hip: -2 us per callback)
    unsigned done_seq = 0;
    bool stepwise = false;                   // IRB_FDL_STEPWISE: a call of n blocks = n one-block steps, ring of P slots
    long long cols_launches = 0, debug_drop_at = 0;   // IRB_DEBUG_DROP_GRANULES=n (tests): the n-th column-sliced launch loses one producer's granules
    unsigned long long n_launches = 0, n_waits = 0, n_copies = 0;   // what the host asked of the runtime since create (irb_fdl_get_counters)
    int ovl_cur = 0;                         // which half of d_ovl holds the live overlap
    size_t part_rows = 0;                    // capacity of d_part in spectrum rows
    unsigned flags = 0;
    long long t = 0;                         // block steps done since reset
    size_t bytes = 0;

    // matrix-core path of long calls (fdl_mfma.hpp): the FDL and the IR spectra a second time, time / tap contiguous per bin
    bool mfma_capable = false;               // this handle's longest call multiplies on the matrix cores: d_Xt / d_Ht exist
    float2 *d_Xt = nullptr, *d_Ht = nullptr; // [S][M][xt_len], [S][M][ht_len]
    int xt_len = 0, xt_hist = 0, ht_len = 0;
    float2 *d_H = nullptr, *d_X = nullptr, *d_tw = nullptr, *d_part = nullptr;
    float *d_ovl = nullptr, *d_in = nullptr, *d_out = nullptr, *d_irtime = nullptr;
    int2 *d_stream_info = nullptr;          // [S] {IR set, partitions in use}: what the kernels read
    unsigned *d_tickets = nullptr;           // [S] arrival counters of k_step_split
    float *h_io = nullptr;                   // pinned: [S][Tmax*B]  (IRB_FDL_HOST_IO: [2][S][Tmax*B], input half + output half)
    float *h_ir = nullptr;                   // pinned: [2][C][P*B]: two staging slots, so that irb_fdl_set_ir never waits for the stream
    hipEvent_t ev_ir[2] = {nullptr, nullptr}; // recorded behind the work that reads a slot
    bool ir_used[2] = {false, false};
    int ir_slot = 0;
    int2 *h_info_slot = nullptr;             // pinned: [2][C] stream_info records of the instance a slot carries
    bool part_busy = false;                  // a partition refresh enqueued since the last wait may still read slot 0 of h_ir
    std::vector<char> part_slot_busy;        // which partition slots those refreshes used
    // IRB_FDL_HOST_IO: the kernels read blocks / IR partitions from, and write output blocks to, MAPPED pinned host memory
    // (zero copy): a blocking process call is pack + launch + wait, with no copy-engine round trips
    bool host_io = false;
    float *d_hio = nullptr, *d_hir = nullptr;   // device addresses of h_io / h_ir
    int *h_set_of_stream = nullptr, *h_np_of_set = nullptr;   // host: IR set of every stream, partitions in use of every set
    int2 *h_info = nullptr;                  // pinned staging of d_stream_info
    // staged one-shot call (IRB_FDL_HOST_IO engines that take long calls): chunk-completion counters in device memory and the
    // chunks' words in mapped host memory (one cache line each) that k_ifft_ola_run raises -- see StepArgs::ck_*
    unsigned *d_ck = nullptr, *h_ck = nullptr, *d_hck = nullptr;   // two sets of kCkMax each: consecutive pieces of a call alternate
    unsigned *h_pk = nullptr, *d_hpk = nullptr;                     // the other direction: words the packing threads raise per input item, polled by k_push_fft_batch_t (StepArgs::pk_*)
    unsigned ck_seq = 0;
    // ... and its pipeline over the pieces of a long signal: the forward FFTs of piece k + 1 run on a second stream while piece k's
    // inverse FFTs leave on the first; events order them around the multiplies
    hipStream_t stream_in = nullptr;
    hipEvent_t ev_pipe[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // forward FFTs of piece k done [k % 2], multiply of piece k done [2 + k % 2], FDL reset [4]
};

namespace {

// every wait on / copy through the runtime is counted (irb_fdl_get_counters: launches, waits and copies per callback are
// part of what bench.py reports for the real-time path)
inline hipError_t irb_sync(irb_fdl *h) { h->n_waits++; h->part_busy = false; return hipStreamSynchronize(h->stream); }
#define COPY_ASYNC(h_, ...)   ((h_)->n_copies++, hipMemcpyAsync(__VA_ARGS__))
#define COPY2D_ASYNC(h_, ...) ((h_)->n_copies++, hipMemcpy2DAsync(__VA_ARGS__))

template <typename T>
int dev_alloc(irb_fdl *h, T **p, size_t n)
{
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(p), std::max<size_t>(n, 1) * sizeof(T)));
    h->bytes += n * sizeof(T);
    return IRB_OK;
}

#define FOR_N(N_, ...)                                                     \
    switch (N_) {                                                          \
    case 32:   { constexpr int NN = 32;   __VA_ARGS__; } break;                   \
    case 64:   { constexpr int NN = 64;   __VA_ARGS__; } break;                   \
    case 128:  { constexpr int NN = 128;  __VA_ARGS__; } break;                   \
    case 256:  { constexpr int NN = 256;  __VA_ARGS__; } break;                   \
    case 512:  { constexpr int NN = 512;  __VA_ARGS__; } break;                   \
    case 1024: { constexpr int NN = 1024; __VA_ARGS__; } break;                   \
    case 2048: { constexpr int NN = 2048; __VA_ARGS__; } break;                   \
    case 4096: { constexpr int NN = 4096; __VA_ARGS__; } break;                   \
    default: return fail(IRB_ERR_ARG, "unsupported FFT length %d", N_);    \
    }

// the instantiations of the one-launch real-time callback (B <= 512: its LDS image fits the static 64 KiB)
#define FOR_N_RT(N_, ...)                                                  \
    switch (N_) {                                                          \
    case 32:   { constexpr int NN = 32;   __VA_ARGS__; } break;                   \
    case 64:   { constexpr int NN = 64;   __VA_ARGS__; } break;                   \
    case 128:  { constexpr int NN = 128;  __VA_ARGS__; } break;                   \
    case 256:  { constexpr int NN = 256;  __VA_ARGS__; } break;                   \
    case 512:  { constexpr int NN = 512;  __VA_ARGS__; } break;                   \
    case 1024: { constexpr int NN = 1024; __VA_ARGS__; } break;                   \
    default: return fail(IRB_ERR_ARG, "unsupported FFT length %d", N_);    \
    }

#define FOR_N_T8(N_, ...)                                                  \
    switch (N_) {                                                          \
    case 256:  { constexpr int NN = 256;  __VA_ARGS__; } break;                   \
    case 512:  { constexpr int NN = 512;  __VA_ARGS__; } break;                   \
    case 1024: { constexpr int NN = 1024; __VA_ARGS__; } break;                   \
    default: return fail(IRB_ERR_ARG, "unsupported FFT length %d", N_);    \
    }

// One-block steps: which kernel family, and how a stream is spread over workgroups.
//   fused  (colW = 0)   one workgroup per stream: HBM-sized batches (no hand-off at all)
//   cols   (colW > 0)   column slices x partition splits, single launch (k_step_cols): few streams (latency) and mid-size
//                       batches that one workgroup per stream cannot fill the chip with
// (rounds 1-4 also carried partition-split kernels with a whole-row hand-off behind IRB_FDL_FORCE_SPLIT; no heuristic chose them)
void choose_path(irb_fdl *h)
{
    const int L = h->M / 2;
    h->colW = 0; h->splits = 1; h->cols_handoff = 0;
    if (h->flags & IRB_FDL_FORCE_FUSED) return;
    const bool force_cols = (h->flags & IRB_FDL_FORCE_COLS) != 0;
    // Measured on MI355X (profiles/r02_latency_sweep.txt, r02_batch_sweep.txt, r02_step_stamps.txt):
    //  * from 224 streams on, one workgroup per stream (nearly) fills the chip and needs no hand-off: fused kernel
    //    (profiles/r03_batch_sweep.txt: 192 streams cols 1042 / fused 1002 Msamples/s, 256 streams cols 949 / fused 1060);
    //  * short IRs: the hand-off (~2 us: drained write-through store, ticket, slice loads) costs more than streaming
    //    the few rows in one workgroup (fused ~ 5.7 us + 0.05 us per 2-KiB partition row);
    //  * otherwise column slices x partition splits with <= 256 workgroups (a second workgroup per CU only queues
    //    behind the first on this latency-bound path), two rounds of loads per thread for long IRs (more splits
    //    shorten the loads but every workgroup adds ~13 ns to the stream's ticket counter and a row to the final sum).
    if (!force_cols && (h->S >= dbg_int("fused_from", 224) || (long) h->P * h->N <= dbg_int("fused_rows", 48) * 512L)) return;
    int W = 64;
    for (int w : {4, 16, 64}) {
        const int we = std::min(w, L);
        if ((long) h->S * (L / we) <= 64) { W = w; break; }
    }
    W = dbg_int("colw", W);
    if (W != 4 && W != 16 && W != 64) W = 64;
    const int we = std::min(W, L), CS = L / we;
    const long base = (long) h->S * CS;
    // two rounds of loads per thread (8 rows of H and 8 of the FDL each): more splits shorten the loads, but every
    // workgroup adds ~13 ns to the stream's ticket counter and a row to the last arriver's sum
    const int per_two_rounds = 2 * (irb::kThreads / we) * 8;
    int PS = (h->P + per_two_rounds - 1) / per_two_rounds;
    PS = (int) std::min<long>(PS, std::max<long>(1, 256 / base));
    PS = std::max(1, std::min(PS, 16));
    PS = std::max(1, dbg_int("psplits", PS));
    h->colW = W;
    h->splits = PS;
    // hand-off form: tagged granules (2) while the whole grid is certainly resident (<= 256 workgroups: the reducer spins),
    // else write-through slices + ticket (0)
    // (measured, profiles/r02_handoff_ab.txt: granules win on narrow slices -- 1 to 8 streams: 5.41 -> 5.15, 7.29 -> 6.39,
    // 7.18 -> 6.65 us -- and lose from 32 streams on, where one reducer per stream sweeps 2048+ granules).  Both lean on how gfx950
    // acknowledges write-through stores: measured there; the library is built for that chip only.
    h->cols_handoff = dbg_int("cols_handoff", (base * PS <= 256 && W < 64) ? 2 : 0) == 2 ? 2 : 0;
}

constexpr int kCkMax = 1024;                  // output chunks of a staged one-shot piece (irb_fdl::d_ck / h_ck)
constexpr int kRingSpare = 8;                 // ring slots beyond P history + the call's blocks (a tile's overhang)
constexpr int kPadP = irb::kTilePeriod;       // H rows per set: P rounded up to this; the extra rows stay zero

// outputs per time tile for a call of T blocks: 16, or 8 when the call has no more than 8 blocks (a 16-output tile would multiply
// eight outputs nobody keeps: on HBM-sized batches of 8-block calls the MAC is then as expensive as the row traffic)
int tile_outputs(const irb_fdl *, int T) { return T <= 8 ? 8 : 16; }

// partition splits for a call of T blocks: as many as still fit the resident slots in ONE round (every split adds a partial spectrum
// per output that is written, flushed and read back), every worker at least one ring period of partitions
int tiled_splits(const irb_fdl *h, int T)
{
    const int L = h->M / 2, LW = std::min(L, irb::kThreads), stripes = irb::kThreads / LW;
    const int tt = tile_outputs(h, T);
    const int ntiles = (T + tt - 1) / tt;
    const int by_work = std::max(1, (h->Ppad / kPadP) / stripes);
    const long wgs = (long) h->S * ntiles;
    const int by_chip = (int) std::max<long>(1, h->tiled_chip_wgs / wgs);
    return std::max(1, std::min(by_work, by_chip));
}

// long calls: the Toeplitz-block MAC on the matrix cores (fdl_mfma.hpp)
bool use_mfma(const irb_fdl *h, int T)
{
    // (mfma_capable: gfx950, not stepwise, the handle's longest call qualifies; set at create.)  The kernel multiplies whole tiles of 256
    // blocks: a call that would pad by more than half (257 .. 341 blocks) stays on the packed-FMA kernel (profiles/r04_mfma_ab.txt)
    const long padded = (long) (T + irb::kMfmaTile - 1) / irb::kMfmaTile * irb::kMfmaTile;
    return h->mfma_capable && T >= h->mfma_min_T && (2 * padded <= 3 * (long) T || h->mfma_min_T < irb::kMfmaTile);   // (a threshold below one tile = a test forcing the path)
}

// How a call of T blocks is multiplied there.  A workgroup = one 256-block tile x 16 bins x one range of Toeplitz chunks and there
// is one workgroup per CU at a time, so a launch should be whole rounds of (at most) mfma_wgs workgroups: a run of tiles that
// does not fill a whole number of rounds is multiplied in TWO launches -- the full rounds unsplit, then the remaining tiles with
// their partitions split over the idle CUs -- each with its own region of the partial spectra (its own rows per block).
// (configs[1] one-shot: 4125 blocks = 17 tiles x 16 bin groups = 272 workgroups on 256 CUs would take two rounds for 6 % more
// work than one; as 16 + 1 tiles the second launch is 64 workgroups of a quarter of the partitions each.)
// (Tried and dropped, profiles/r04_oneshot.txt: the one-shot call's signal in two halves on two queues, so that the second half
// arrives while the first is multiplied and sent back -- 0.304 ms against 0.291 at configs[1]: the halves' kernels share the CUs.)
struct MfmaLaunch { int tile_first, tile_count, splits; };
struct MfmaPlan {
    int n = 0;
    MfmaLaunch l[2];
    int t0(int i, int T) const { return i < n ? l[i].tile_first * irb::kMfmaTile : T; }      // first block of launch i's region
    size_t region_rows(int i, int S, int T) const { return (size_t) S * (size_t) (t0(i + 1, T) - t0(i, T)) * l[i].splits; }
    size_t part_rows(int S, int T) const { size_t r = 0; for (int i = 0; i < n; i++) r += region_rows(i, S, T); return r; }
};

MfmaPlan mfma_plan(const irb_fdl *h, int T)
{
    MfmaPlan p;
    const long units = (long) h->S * (h->M / irb::kMfmaBins);                                // workgroups per tile at one split
    const int by_work = std::max(1, irb::mfma_chunks(h->P) / h->mfma_min_chunks);           // every split keeps a few chunks
    const int target = h->mfma_wgs;
    const long slots = target;                                                               // workgroups resident at once: one per CU (two start late and multiply slower, profiles/r04_mfma_stamps.txt)
    auto splits_for = [&](long wgs) { return (int) std::max<long>(1, std::min<long>(by_work, slots / std::max<long>(1, wgs))); };
    const long per_round = target / units;                                                   // tiles that give every CU one workgroup
    auto run = [&](int first, int ntiles) {                                                  // tiles [first, first + ntiles)
        if (ntiles <= 0) return;
        const int full = per_round > 0 ? (int) (ntiles / per_round * per_round) : ntiles;
        if (full == 0) {
            p.l[p.n++] = {first, ntiles, ntiles * units <= target ? splits_for(ntiles * units) : 1};
            return;
        }
        p.l[p.n++] = {first, full, 1};
        // (the remainder's workgroups are few and short: down to three chunks per split)
        if (ntiles > full) p.l[p.n++] = {first + full, ntiles - full,
                                          (int) std::max<long>(1, std::min<long>(std::max(1, irb::mfma_chunks(h->P) / 3), slots / std::max<long>(1, (ntiles - full) * units)))};
    };
    run(0, (T + irb::kMfmaTile - 1) / irb::kMfmaTile);
    return p;
}

irb::StepArgs step_args(const irb_fdl *h, const float *d_in, float *d_out, long stride)
{
    irb::StepArgs a;
    a.H = h->d_H; a.X = h->d_X; a.ovl = h->d_ovl + (size_t) h->ovl_cur * h->S * h->B; a.tw = h->d_tw;
    a.ovl_next = a.ovl; a.T = 1; a.nq = 1;
    a.Xt = h->d_Xt; a.Ht = h->d_Ht; a.xt_len = h->xt_len; a.xt_hist = h->xt_hist; a.ht_len = h->ht_len; a.hist_first = -1;
    a.tile_first = 0; a.tile_count = 1; a.t_base = 0; a.part_T = 1;
    a.mf_wgs0 = 0; a.mf2_tile_first = 0; a.mf2_tile_count = 1; a.mf2_splits = 1; a.mf2_t_base = 0; a.mf2_part_T = 1; a.mf2_part_off = 0;
    a.t_first = 0; a.t_count = 1;
    for (int r = 0; r < 3; r++) a.reg_t0[r] = r == 0 ? 0 : 1;
    for (int r = 0; r < 2; r++) { a.reg_nq[r] = 1; a.reg_off[r] = 0; }
    a.in_len = a.out_len = (long) 1 << 62;
    a.rf_src = nullptr; a.rf_part = 0; a.done = nullptr; a.done_seq = 0; a.debug_drop = 0;
    a.ck_cnt = nullptr; a.ck_flag = nullptr; a.ck_blocks = 1; a.ck_seq = 0;
    a.pk_flag = nullptr; a.pk_blocks = 16; a.pk_seq = 0; a.dbg_sums = nullptr;
    a.stream_info = h->d_stream_info;
    a.in = d_in; a.out = d_out; a.part = h->d_part;
    a.in_stride = stride; a.out_stride = stride;
    a.B = h->B; a.P = h->Ppad; a.R = h->R; a.head = (int) (h->t % h->R); a.splits = h->splits;
    a.gran = h->d_gran; a.status = h->d_status; a.epoch = (unsigned) (h->t % 0xFFFFFFFFull) + 1u;
    return a;
}

#define LAUNCH_COLS(W_, HO_)                                                                                     \
    FOR_N(h->N, irb::k_step_cols<NN, W_, HO_><<<dim3(h->S * irb::ColsGeom<NN, W_>::CS * h->splits), dim3(irb::kThreads), 0, h->stream>>>(a, h->d_tickets))
#define LAUNCH_COLS_W(HO_)                                                                                        \
    if (h->colW == 4) { LAUNCH_COLS(4, HO_); } else if (h->colW == 16) { LAUNCH_COLS(16, HO_); } else { LAUNCH_COLS(64, HO_); }

int launch_cols(irb_fdl *h, irb::StepArgs a)
{
    h->n_launches++;
    a.debug_drop = (++h->cols_launches == h->debug_drop_at) ? 1 : 0;
    if (h->cols_handoff == 2) { LAUNCH_COLS_W(2) }
    else { LAUNCH_COLS_W(0) }
    HIP_TRY(hipGetLastError());
    return IRB_OK;
}

// a kernel's bounded spin gave up (hand-off form 2): report it once, at the next point where the host has waited
int check_status(irb_fdl *h)
{
    if (h->h_status && *h->h_status) {
        // the host has waited for the stream (every caller does before it asks): nothing enqueued still reads the staging.
        // No allocation, no stream, no pointer changes here -- this runs on the audio thread (include/irb_fdl.h: "after create
        // no entry point allocates device memory")
        *h->h_status = 0;
        *h->h_done = 0;
        std::fill(h->part_slot_busy.begin(), h->part_slot_busy.end(), 0);
        return fail(IRB_ERR_STATE, "in-kernel hand-off timed out (a producer workgroup never published its slice)");
    }
    return IRB_OK;
}

int launch_step(irb_fdl *h, const irb::StepArgs &a)
{
    if (h->colW > 0) return launch_cols(h, a);
    h->n_launches++;
    // rows streamed once per step from a working set beyond the 256 MiB Infinity Cache: non-temporal
    // loads (+10 % HBM rate, profiles/r01_fused_variants.txt); cache-resident sets keep the default policy
    if (h->pair && h->N <= 2048) {
        if (h->nt) { FOR_N(h->N, irb::k_step_fused_pair<NN, true><<<dim3(h->S / 2), dim3(irb::kThreads), 0, h->stream>>>(a)); }
        else       { FOR_N(h->N, irb::k_step_fused_pair<NN, false><<<dim3(h->S / 2), dim3(irb::kThreads), 0, h->stream>>>(a)); }
    }
    else if (h->fused_u == 4 && h->nt) { FOR_N(h->N, irb::k_step_fused<NN, 4, true><<<dim3(h->S), dim3(irb::kThreads), 0, h->stream>>>(a)); }
    else if (h->fused_u == 4)          { FOR_N(h->N, irb::k_step_fused<NN, 4, false><<<dim3(h->S), dim3(irb::kThreads), 0, h->stream>>>(a)); }
    else if (h->nt) { FOR_N(h->N, irb::k_step_fused<NN, 2, true><<<dim3(h->S), dim3(irb::kThreads), 0, h->stream>>>(a)); }
    else       { FOR_N(h->N, irb::k_step_fused<NN, 2, false><<<dim3(h->S), dim3(irb::kThreads), 0, h->stream>>>(a)); }
    HIP_TRY(hipGetLastError());
    return IRB_OK;
}

// ---- T > 1 blocks of every stream (see fdl_kernels.hpp, time-tiled path; fdl_mfma.hpp) -------------------------------------
// Three phases, each able to work on a sub-range of the call's blocks so that irb_convolve_periodic can move a signal in and out
// in chunks around ONE multiply over all of its blocks: forward FFTs of blocks [first, first + count), the multiply-accumulate
// of all T outputs, inverse FFTs + overlap-add of blocks [first, first + count).
struct TiledCall {
    irb::StepArgs a;
    int T = 0, tt = 0, ntiles = 0;
    bool mfma = false, one_bin = false;
    MfmaPlan plan;
};

int tiled_begin(irb_fdl *h, irb::StepArgs a, int T, TiledCall *c)
{
    const int L = h->M / 2, LW = std::min(L, irb::kThreads), stripes = irb::kThreads / LW;
    a.T = T;
    a.splits = tiled_splits(h, T);
    a.nq = a.splits * stripes;
    a.ovl_next = h->d_ovl + (size_t) (h->ovl_cur ^ 1) * h->S * h->B;
    a.t_first = 0; a.t_count = T; a.part_T = T; a.t_base = 0;
    c->T = T;
    // one complex bin per lane, 16 outputs per tile: fewer bytes per FMA.  Pays when the rows come from HBM (working
    // set beyond the Infinity Cache) and the call is long enough; cache-resident calls are faster with k_cmac_tiled
    c->mfma = use_mfma(h, T);
    c->one_bin = !c->mfma && h->tiled1 && T >= 16 && h->N >= 128 && h->N <= 2048;
    const int stripes1 = irb::kThreads / std::min(h->M, irb::kThreads);
    if (c->mfma) {
        c->plan = mfma_plan(h, T);
        if (c->plan.part_rows(h->S, T) > h->part_rows) return fail(IRB_ERR_STATE, "partial workspace too small (internal)");
        long off = 0;
        for (int r = 0; r < 3; r++) a.reg_t0[r] = c->plan.t0(r, T);
        for (int r = 0; r < c->plan.n; r++) {
            a.reg_nq[r] = c->plan.l[r].splits;
            a.reg_off[r] = off;
            off += (long) c->plan.region_rows(r, h->S, T) * h->M;
        }
    }
    else if (c->one_bin) a.nq = a.splits * stripes1;
    else a.nq = a.splits;                                    // k_cmac_tiled_pk adds its stripes before it stores
    if (!c->mfma) {
        if ((size_t) h->S * T * a.nq > h->part_rows) return fail(IRB_ERR_STATE, "partial workspace too small (internal)");
        for (int r = 0; r < 3; r++) a.reg_t0[r] = r == 0 ? 0 : T;                            // one region: part[s][t][nq][M]
        a.reg_nq[0] = a.nq; a.reg_off[0] = 0;
    }
    c->tt = tile_outputs(h, T);
    c->ntiles = c->one_bin ? (T + 15) / 16 : (T + c->tt - 1) / c->tt;
    c->a = a;
    c->a.hist_first = -1;
    if (c->mfma && h->t > 0) {
        // the call continues a stream: the blocks before its first one go from the ring into Xt's history columns (extra workgroups
        // of the first forward-FFT launch)
        int first = (int) ((h->t - h->xt_hist) % h->R);
        if (first < 0) first += h->R;
        c->a.hist_first = first;
    }
    return IRB_OK;
}

int tiled_push(irb_fdl *h, const TiledCall &c, int first, int count, hipStream_t stream = nullptr)
{
    if (!stream) stream = h->stream;
    irb::StepArgs a = c.a;
    a.t_first = first; a.t_count = count;
    h->n_launches++;
    if (c.mfma) {
        if (first != 0) a.hist_first = -1;                              // (the call's first launch carries the history)
        const int hist_wgs = a.hist_first >= 0 ? h->S * (h->xt_hist / 16) * std::max(1, h->M / 64) : 0;
        FOR_N(h->N, irb::k_push_fft_batch_t<NN><<<dim3(h->S * ((count + irb::PushTGeom<NN>::NBW - 1) / irb::PushTGeom<NN>::NBW) + hist_wgs),
                                                  dim3(irb::kWave * irb::PushTGeom<NN>::NBW), 0, stream>>>(a, h->S));
    } else {
        FOR_N(h->N, irb::k_push_fft_batch<NN><<<dim3((h->S * count + irb::FftGeom<NN>::NF - 1) / irb::FftGeom<NN>::NF), dim3(irb::kWave * irb::FftGeom<NN>::NF), 0, stream>>>(a, h->S));
    }
    HIP_TRY(hipGetLastError());
    return IRB_OK;
}

int tiled_mac(irb_fdl *h, const TiledCall &c)
{
    irb::StepArgs a = c.a;
    const dim3 mac_grid(h->S * c.ntiles * a.splits), mac_block(irb::kThreads);
    // HBM-sized batches with a single time tile per stream read every row once: non-temporal like the fused kernel
    const bool nt1 = h->nt && c.ntiles == 1;
    h->n_launches++;
    if (c.mfma) {
        const int groups = h->M / irb::kMfmaBins;
        float2 *const part0 = a.part;
        h->n_launches--;
        auto wgs = [&](const MfmaLaunch &l) { return h->S * l.tile_count * l.splits * groups; };
        if (c.plan.n == 2) {
            // both regions of the plan in ONE grid: the second's workgroups (few, short) come last in dispatch order and run on
            // the CUs the first round frees -- one launch's fixed cost instead of two
            const MfmaLaunch &l0 = c.plan.l[0], &l1 = c.plan.l[1];
            h->n_launches++;
            a.tile_first = l0.tile_first; a.tile_count = l0.tile_count; a.splits = l0.splits; a.nq = l0.splits;
            a.t_base = a.reg_t0[0]; a.part_T = a.reg_t0[1] - a.reg_t0[0]; a.part = part0 + a.reg_off[0];
            a.mf_wgs0 = wgs(l0);
            a.mf2_tile_first = l1.tile_first; a.mf2_tile_count = l1.tile_count; a.mf2_splits = l1.splits;
            a.mf2_t_base = a.reg_t0[1]; a.mf2_part_T = a.reg_t0[2] - a.reg_t0[1]; a.mf2_part_off = (long) a.reg_off[1] - (long) a.reg_off[0];
            FOR_N(h->N, irb::k_cmac_mfma<NN><<<dim3(wgs(l0) + wgs(l1)), dim3(irb::kMfmaThreads), 0, h->stream>>>(a));
        } else {
            const MfmaLaunch &l = c.plan.l[0];
            h->n_launches++;
            a.tile_first = l.tile_first; a.tile_count = l.tile_count; a.splits = l.splits; a.nq = l.splits;
            a.t_base = a.reg_t0[0]; a.part_T = a.reg_t0[1] - a.reg_t0[0]; a.part = part0 + a.reg_off[0];
            a.mf_wgs0 = 0;
            FOR_N(h->N, irb::k_cmac_mfma<NN><<<dim3(wgs(l)), dim3(irb::kMfmaThreads), 0, h->stream>>>(a));
        }
    }
    else if (c.one_bin) { FOR_N(h->N, irb::k_cmac_tiled1<NN, 16><<<mac_grid, mac_block, 0, h->stream>>>(a)); }
    else if (c.tt == 8 && nt1) { FOR_N(h->N, irb::k_cmac_tiled_pk<NN, 8, 4, 4, true><<<mac_grid, mac_block, 0, h->stream>>>(a)); }
    else if (c.tt == 8)        { FOR_N(h->N, irb::k_cmac_tiled_pk<NN, 8, 4, 4, false><<<mac_grid, mac_block, 0, h->stream>>>(a)); }
    else if (nt1) { FOR_N(h->N, irb::k_cmac_tiled_pk<NN, 16, 8, 2, true><<<mac_grid, mac_block, 0, h->stream>>>(a)); }
    else          { FOR_N(h->N, irb::k_cmac_tiled_pk<NN, 16, 8, 2, false><<<mac_grid, mac_block, 0, h->stream>>>(a)); }
    HIP_TRY(hipGetLastError());
    return IRB_OK;
}

int tiled_ifft(irb_fdl *h, const TiledCall &c, int first, int count, hipStream_t stream = nullptr)      // first: a multiple of OlaGeom::NB
{
    if (!stream) stream = h->stream;
    irb::StepArgs a = c.a;
    a.t_first = first; a.t_count = count;
    h->n_launches++;
    FOR_N(h->N, irb::k_ifft_ola_run<NN, irb::OlaGeom<NN>::NB><<<dim3(h->S * ((count + irb::OlaGeom<NN>::NB - 1) / irb::OlaGeom<NN>::NB)),
                                                                 dim3(irb::kWave * (irb::OlaGeom<NN>::NB + 1)), 0, stream>>>(a, h->S));
    HIP_TRY(hipGetLastError());
    return IRB_OK;
}

void tiled_end(irb_fdl *h, const TiledCall &c)
{
    h->ovl_cur ^= 1;
    h->t += c.T;
}

// calls of at most 8 blocks on batches whose streams get one workgroup each: push, multiply and inverse in ONE launch (k_tile8_fused)
bool use_tile8_fused(const irb_fdl *h, int T)
{
    return h->tile8_fused && h->N >= 256 && h->N <= 1024 && tile_outputs(h, T) == 8 && tiled_splits(h, T) == 1 && !use_mfma(h, T);
}

int launch_tiled(irb_fdl *h, irb::StepArgs a, int T)
{
    if (use_tile8_fused(h, T)) {
        a.T = T;
        a.ovl_next = h->d_ovl + (size_t) (h->ovl_cur ^ 1) * h->S * h->B;
        h->n_launches++;
        if (h->nt) { FOR_N_T8(h->N, irb::k_tile8_fused<NN, true><<<dim3(h->S), dim3(irb::kThreads), 0, h->stream>>>(a)); }
        else       { FOR_N_T8(h->N, irb::k_tile8_fused<NN, false><<<dim3(h->S), dim3(irb::kThreads), 0, h->stream>>>(a)); }
        HIP_TRY(hipGetLastError());
        h->ovl_cur ^= 1;
        return IRB_OK;
    }
    TiledCall c;
    int rc = tiled_begin(h, a, T, &c);
    if (rc == IRB_OK) rc = tiled_push(h, c, 0, T);
    if (rc == IRB_OK) rc = tiled_mac(h, c);
    if (rc == IRB_OK) rc = tiled_ifft(h, c, 0, T);
    if (rc == IRB_OK) h->ovl_cur ^= 1;                    // (h->t is advanced by the caller, as for one-block steps)
    return rc;
}

int launch_ir_fft(irb_fdl *h, const float *d_ir, int nparts, int nrows, int first_set, int first_partition)
{
    h->n_launches++;
    FOR_N(h->N, irb::k_ir_fft<NN><<<dim3((nparts * nrows + irb::FftGeom<NN>::NF - 1) / irb::FftGeom<NN>::NF),
                                     dim3(irb::kWave * irb::FftGeom<NN>::NF), 0, h->stream>>>(
                    d_ir, h->d_H, h->d_tw, h->B, h->Ppad, first_set, first_partition, nparts, nrows));
    HIP_TRY(hipGetLastError());
    return IRB_OK;
}

// the IR sets [first_set, first_set + nsets) once more with the tap index contiguous per bin (fdl_mfma.hpp); np rows of each are
// taps, everything behind them zeros
int launch_ht(irb_fdl *h, int first_set, int nsets, int np)
{
    if (!h->d_Ht || nsets < 1) return IRB_OK;
    h->n_launches++;
    const int nrows = h->ht_len - 16;
    const dim3 grid(nrows / 16, h->M >= 64 ? h->M / 64 : 1, nsets);
    FOR_N(h->N, irb::k_rows_to_columns<NN / 2><<<grid, dim3(256), 0, h->stream>>>(h->d_H + (size_t) first_set * h->Ppad * h->M, (long) h->Ppad * h->M, 0, 0, np,
                                                                                   h->d_Ht + (size_t) first_set * h->M * h->ht_len, (long) h->M * h->ht_len, h->ht_len, 16, nrows));
    HIP_TRY(hipGetLastError());
    return IRB_OK;
}

// both channels of every instance share their IR set: the paired step kernels apply
void update_pairing(irb_fdl *h)
{
    h->pair = (h->C == 2) && !(h->flags & IRB_FDL_NO_PAIRING);
    for (int i = 0; i < h->I && h->pair; i++) h->pair = h->h_set_of_stream[2 * i] == h->h_set_of_stream[2 * i + 1];
}

// ---- resident real-time servers and the rest of the process -----------------------------------------------------------
// A live k_rt_server occupies its stream for as long as callbacks keep arriving.  hipFree, hipHostFree and
// hipDeviceSynchronize wait for EVERY stream of the device, so another plug-in instance's prepareToPlay / releaseResources,
// a one-shot irb_convolve_periodic that swaps its cached engine, or a thread-exit guard would wait for that server's
// lifetime.  A "quiet section" (DeviceQuiet, irb_device_quiet_begin / _end) brackets every such call in this library: it
// raises a per-device count and posts the yield word of every resident-capable handle on the device.  A server leaves
// after the request in hand; while the count is raised an audio thread serves its callbacks with one k_rt_pair launch each
// (an ordinary kernel of a few microseconds) instead of starting a server, and resumes the resident form afterwards.
constexpr int kMaxDevices = 64;
std::mutex g_reg_mu;
std::vector<irb_fdl *> g_resident;                    // handles that own a mailbox (created with IRB_FDL_RESIDENT on a large-BAR device)
std::atomic<int> g_quiet[kMaxDevices];                // threads inside a quiet section, per device
std::atomic<bool> g_quiet_disabled{false};

bool device_quiet(int dev) { return dev >= 0 && dev < kMaxDevices && g_quiet[dev].load(std::memory_order_seq_cst) > 0; }

void quiet_begin(int dev)
{
    if (dev < 0 || dev >= kMaxDevices) return;
    static const bool off = dbg_int("device_quiet", 1) == 0;        // diagnostic: what the library did before round 4 (profiles/r04_resident_vs_process.txt)
    if (off) { g_quiet_disabled.store(true); return; }
    g_quiet[dev].fetch_add(1, std::memory_order_seq_cst);
    std::lock_guard<std::mutex> lk(g_reg_mu);
    for (irb_fdl *o : g_resident)
        if (o->device == dev && o->bar_mb) o->bar_mb->yield = 1;       // through the BAR; the owner clears it when it next starts a server
    __builtin_ia32_sfence();
}

void quiet_end(int dev)
{
    if (g_quiet_disabled.load()) return;
    if (dev >= 0 && dev < kMaxDevices) g_quiet[dev].fetch_sub(1, std::memory_order_seq_cst);
}

struct DeviceQuiet {
    int dev;
    explicit DeviceQuiet(int d) : dev(d) { quiet_begin(dev); }
    ~DeviceQuiet() { quiet_end(dev); }
    DeviceQuiet(const DeviceQuiet &) = delete;
    DeviceQuiet &operator=(const DeviceQuiet &) = delete;
};

void registry_add(irb_fdl *h) { std::lock_guard<std::mutex> lk(g_reg_mu); g_resident.push_back(h); }
void registry_remove(irb_fdl *h)
{
    std::lock_guard<std::mutex> lk(g_reg_mu);
    g_resident.erase(std::remove(g_resident.begin(), g_resident.end(), h), g_resident.end());
}

// ---- resident real-time server (k_rt_server) --------------------------------------------------------------------------
// the generation word the server stores last lives next to the completion word (h_done[1])
inline volatile unsigned *exited_word(irb_fdl *h) { return h->h_done + 1; }

int server_launch(irb_fdl *h, unsigned first_seq)
{
    // everything enqueued on the handle's own stream (IR tables, stream_info) must be in place before the server reads it
    HIP_TRY(irb_sync(h));
    irb::StepArgs a = step_args(h, h->bar_in, h->d_hio + (size_t) h->S * h->Tmax * h->B, (long) h->Tmax * h->B);
    a.rf_src = h->bar_ir; a.done = h->d_done;
    if (++h->server_gen == 0) ++h->server_gen;           // 0 is the exited word's initial value
    h->bar_mb->yield = 0;                                // whatever an earlier quiet section posted
    __builtin_ia32_sfence();
    h->n_launches++;
    if (h->C == 2) {
        FOR_N_RT(h->N, irb::k_rt_server<NN, 2><<<dim3(h->I), dim3(irb::kThreads), 0, h->rt_stream>>>(
                           a, h->bar_mb, h->d_done + 1, h->server_gen, first_seq, h->idle_ticks, h->life_ticks));
    } else {
        FOR_N_RT(h->N, irb::k_rt_server<NN, 1><<<dim3(h->I), dim3(irb::kThreads), 0, h->rt_stream>>>(
                           a, h->bar_mb, h->d_done + 1, h->server_gen, first_seq, h->idle_ticks, h->life_ticks));
    }
    HIP_TRY(hipGetLastError());
    h->server_live = true;
    // a quiet section that began between the caller's check and the store above had its yield word overwritten: post it again
    // (a section that begins after this line posts it itself)
    if (device_quiet(h->device)) { h->bar_mb->yield = 1; __builtin_ia32_sfence(); }
    return IRB_OK;
}

// every entry point other than the real-time callback stops the server first: it owns the engine's state while it runs
int server_stop(irb_fdl *h)
{
    if (!h->server_live) return IRB_OK;
    h->bar_mb->quit = 1;
    __builtin_ia32_sfence();
    h->n_waits++;
    hipError_t e = hipStreamSynchronize(h->rt_stream);   // at the latest the idle timeout ends it
    h->bar_mb->quit = 0;
    __builtin_ia32_sfence();
    h->server_live = false;
    if (e != hipSuccess) return fail(IRB_ERR_HIP, "real-time server: %s", hipGetErrorString(e));
    return IRB_OK;
}

#define STOP_SERVER(h_) do { if ((h_)->server_live) { int rc_ = server_stop(h_); if (rc_ != IRB_OK) return rc_; } } while (0)

int upload_maps(irb_fdl *h)
{
    update_pairing(h);
    HIP_TRY(irb_sync(h));          // the pinned staging may still feed an earlier upload
    for (int s = 0; s < h->S; s++) h->h_info[s] = make_int2(h->h_set_of_stream[s], h->h_np_of_set[h->h_set_of_stream[s]]);
    HIP_TRY(COPY_ASYNC(h, h->d_stream_info, h->h_info, sizeof(int2) * h->S, hipMemcpyHostToDevice, h->stream));
    return IRB_OK;
}

}  // namespace

// irb_convolve_periodic's cached engine and chunk pipeline, one per calling thread.  The guard frees everything (HBM, pinned
// staging, streams, the unpacking thread) when the thread exits -- a plug-in host's pool thread that made one-shot calls
// must not leak an engine; irb_release_cached_plans() is the explicit form.  (For the main thread this runs at exit(),
// before the HIP runtime's own static teardown.)
extern "C" void irb_fdl_destroy(irb_fdl *h);
namespace {

double now_ms()
{
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

// ---- the staged one-shot call's host pipeline ------------------------------------------------------------------------------
// The caller's pageable audio and result never become device-visible (host_ranges.hpp): their samples pass through the engine's
// own pinned staging.  Per piece (<= kOfflineMaxBlocks blocks of the signal):
//   in   the audio goes into the staging in chunks of growing size (the first is small, so that the device starts pulling samples
//        over PCIe a few microseconds into the call); every thread of the call claims chunks from one counter, the CALLING thread
//        launches a chunk's forward FFTs as soon as it and all earlier chunks are packed; the IR's samples are one more item of
//        that queue (copied into the pinned IR staging, then one launch transforms them there);
//   mul  one multiply over all blocks of the piece;
//   out  ONE launch of the inverse FFTs + overlap-add for the whole piece writes the output staging and raises a word in
//        mapped host memory per chunk of ~128 KiB (StepArgs::ck_*); every thread claims chunks from a second counter, spins on
//        the chunk's word and copies the chunk to the caller's buffer with streaming stores (host_copy.cpp).
// The helpers are plain C++ threads that make no HIP call at all; they sleep on a condition variable between calls.  A piece's
// state is reference-counted: a helper that wakes up after the piece is over finds nothing to claim and touches nothing of the call.
extern "C" void irb_copy_stream_(void *dst, const void *src, size_t bytes);   // host_copy.cpp

constexpr int kCrewMax = 8;                            // threads of a staged call, the caller included
constexpr long kSpinLimitNs = 4000000000L;             // a chunk word that has not come after 4 s: the call fails (a kernel that never finished)

struct StagedPiece {
    // geometry
    int C = 0, B = 0;
    long stride = 0;                                   // floats between the channels of the staging
    const float *audio[2] = {nullptr, nullptr};
    float *out[2] = {nullptr, nullptr};
    float *host_in = nullptr, *host_out = nullptr;     // this piece's half of the staging
    long p0 = 0, La = 0, lim = 0;                      // first block of the piece; samples of audio / of result the call has
    // the IR's samples (first piece only): rows of ir_len floats each; whoever gets to it first copies them -- the helpers look
    // here before anything else, the calling thread only once no audio is left to pack
    const float *ir[2] = {nullptr, nullptr};
    float *ir_stage[2] = {nullptr, nullptr};
    long ir_len = 0;
    int ir_rows = 0;
    std::atomic<unsigned char> ir_claimed{1}, ir_done{1};          // (1 / 1: this piece carries no IR)
    // in: the audio is PACKED in items of ~128 KiB per channel, claimed in order from one counter, and LAUNCHED in chunks of
    // whole items (few, growing: every launch of the forward-FFT kernel pays a few microseconds of ramp-up on the PCIe link)
    std::vector<int> item_first, item_count, item_chunk;          // blocks; the launch chunk an item belongs to
    std::vector<int> in_first, in_count;                           // launch chunks, in blocks
    std::unique_ptr<std::atomic<int>[]> chunk_left;                // items of the chunk not yet packed
    std::atomic<int> pack_next{0};
    // (matrix-core pieces: ONE forward-FFT launch for the whole piece, launched before anything is packed; its workgroups wait for the
    // word of their item -- pk_words[16 i] = seq, stored behind the item's samples.  All items then have pk_blocks blocks but the last)
    volatile unsigned *pk_words = nullptr;
    int pk_blocks = 0;
    // out
    int T = 0, ck_blocks = 1, nout = 0;
    volatile unsigned *words = nullptr;                // chunk k's word at words[16 k]
    unsigned seq = 0;
    std::atomic<int> unpack_next{0}, unpacked{0};
    // per-thread time accounting (index 0 = the calling thread)
    double pack_ms[kCrewMax] = {0}, unpack_ms[kCrewMax] = {0}, wait_ms[kCrewMax] = {0};

    bool copy_ir(int who)                              // false: someone else has it
    {
        if (ir_claimed.load(std::memory_order_acquire) || ir_claimed.exchange(1, std::memory_order_acq_rel)) return false;
        const double t0 = now_ms();
        for (int r = 0; r < ir_rows; r++) std::memcpy(ir_stage[r], ir[r], sizeof(float) * (size_t) ir_len);
        ir_done.store(1, std::memory_order_release);
        pack_ms[who] += now_ms() - t0;
        return true;
    }
    bool pack_one(int who)                             // false: nothing left to claim
    {
        const int n = (int) item_first.size();
        if (pack_next.load(std::memory_order_acquire) >= n) return false;
        const int i = pack_next.fetch_add(1, std::memory_order_acq_rel);
        if (i >= n) return false;
        const double t0 = now_ms();
        const long s0 = (p0 + item_first[(size_t) i]) * B, cnt = (long) item_count[(size_t) i] * B;
        const long have = std::max<long>(0, std::min<long>(cnt, La - s0));   // the rest of the item is the flush: the kernel reads zeros for it (StepArgs::in_len)
        for (int ch = 0; ch < C; ch++)
            if (have > 0) std::memcpy(host_in + (size_t) ch * stride + (size_t) item_first[(size_t) i] * B, audio[ch] + s0, sizeof(float) * (size_t) have);
        if (pk_words) __atomic_store_n(pk_words + 16 * (size_t) i, seq, __ATOMIC_RELEASE);
        chunk_left[(size_t) item_chunk[(size_t) i]].fetch_sub(1, std::memory_order_acq_rel);
        pack_ms[who] += now_ms() - t0;
        return true;
    }
    bool all_claimed() const { return pack_next.load(std::memory_order_acquire) >= (int) item_first.size(); }
    int claim_chunk()                                  // an output chunk nobody has yet, -1: none left
    {
        if (unpack_next.load(std::memory_order_acquire) >= nout) return -1;
        const int k = unpack_next.fetch_add(1, std::memory_order_acq_rel);
        return k < nout ? k : -1;
    }
    bool chunk_ready(int k) const { return __atomic_load_n(words + 16 * (size_t) k, __ATOMIC_ACQUIRE) == seq; }
    void copy_chunk(int k, int who)
    {
        const double t1 = now_ms();
        const int first = k * ck_blocks, cnt = std::min(ck_blocks, T - first);
        const long s0 = (p0 + first) * B, n = (long) cnt * B;
        const long take = std::max<long>(0, std::min<long>(n, lim - s0));
        for (int ch = 0; ch < C; ch++)
            if (take > 0) irb_copy_stream_(out[ch] + s0, host_out + (size_t) ch * stride + (size_t) first * B, sizeof(float) * (size_t) take);
        unpack_ms[who] += now_ms() - t1;
        unpacked.fetch_add(1, std::memory_order_acq_rel);
    }
    bool all_unpacked() const { return unpacked.load(std::memory_order_acquire) >= nout; }
};

// One staged call = its pieces + what the threads share.  Reference-counted: a helper that wakes up after the call is over finds
// `finished` set and touches nothing of the call.
struct StagedCall {
    std::vector<std::unique_ptr<StagedPiece>> piece;
    std::atomic<int> open{0};                          // pieces [0, open) may be worked on (their halves of the staging are free)
    // the call ends (finished) or fails (gave_up: set, then `active` -- helpers inside work() -- is waited for before the failing call
    // returns: nothing may touch the caller's buffers afterwards).  seq_cst: a helper that enters late sees the flag, one that entered
    // earlier is waited for.
    std::atomic<int> finished{0}, gave_up{0}, active{0};

    bool stop() const { return finished.load() || gave_up.load(); }
    // something to pack in the open pieces?  (the older one first: the device wants it first)
    bool pack_some(int who)
    {
        const int n = open.load(std::memory_order_acquire);
        for (int k = std::max(0, n - 2); k < n; k++)
            if (piece[(size_t) k]->copy_ir(who) || piece[(size_t) k]->pack_one(who)) return true;
        return false;
    }
    // the oldest output chunk nobody has: wait for its word (packing whatever turns up meanwhile: the device's input must not starve
    // behind a thread that waits for its output), copy it out.  false: no chunk left in the open pieces.
    // (only_ready: take a chunk only if its word is there already -- the calling thread between its launches)
    bool unpack_some(int who, int &oldest, bool only_ready = false)
    {
        const int n = open.load(std::memory_order_acquire);
        while (oldest < n && piece[(size_t) oldest]->unpack_next.load(std::memory_order_acquire) >= piece[(size_t) oldest]->nout) oldest++;
        for (int k = oldest; k < n; k++) {
            StagedPiece &p = *piece[(size_t) k];
            if (only_ready) {
                const int nx = p.unpack_next.load(std::memory_order_acquire);
                if (nx >= p.nout || !p.chunk_ready(nx)) continue;
            }
            const int c = p.claim_chunk();
            if (c < 0) continue;
            const double t0 = now_ms();
            timespec ts0;
            clock_gettime(CLOCK_MONOTONIC, &ts0);
            for (unsigned spins = 0; !p.chunk_ready(c); spins++) {
                if (gave_up.load(std::memory_order_relaxed)) return true;
                if ((spins & 63u) == 63u && pack_some(who)) continue;
                __builtin_ia32_pause();
                if ((spins & 4095u) == 4095u) {
                    timespec ts1;
                    clock_gettime(CLOCK_MONOTONIC, &ts1);
                    if ((ts1.tv_sec - ts0.tv_sec) * 1000000000L + (ts1.tv_nsec - ts0.tv_nsec) > kSpinLimitNs) { gave_up.store(1); return true; }
                }
            }
            p.wait_ms[who] += now_ms() - t0;
            p.copy_chunk(c, who);
            return true;
        }
        return false;
    }
    void work(int who)                                 // a helper thread's life inside one call
    {
        active.fetch_add(1);
        int oldest = 0;
        while (!stop()) {
            if (pack_some(who)) continue;
            if (unpack_some(who, oldest)) continue;
            __builtin_ia32_pause();
        }
        active.fetch_sub(1);
    }
    void abandon()                                     // the call has failed: no helper is inside it when this returns
    {
        gave_up.store(1);
        timespec t0;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (unsigned spins = 0; active.load() > 0; spins++) {
            __builtin_ia32_pause();
            if ((spins & 4095u) == 4095u) {
                timespec t1;
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if ((t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) > 2 * kSpinLimitNs) break;
            }
        }
    }
};

struct Crew {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv;
    std::shared_ptr<StagedCall> cur;
    std::atomic<unsigned long long> gen{0};
    bool quit = false;
    long linger_ns = 150000;                           // a helper that has finished a call keeps looking for the next one this long before it
                                                       // sleeps: calls in a row (a batch render) do not pay a thread wake-up each (~40 us)
    void run(int who)
    {
        unsigned long long seen = 0;
        for (;;) {
            std::shared_ptr<StagedCall> p;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return quit || gen.load(std::memory_order_acquire) != seen; });
                if (quit) return;
                seen = gen.load(std::memory_order_acquire);
                p = cur;
            }
            if (p) p->work(who);
            p.reset();
            timespec t0;
            clock_gettime(CLOCK_MONOTONIC, &t0);
            for (unsigned spins = 0; gen.load(std::memory_order_acquire) == seen; spins++) {
                __builtin_ia32_pause();
                if ((spins & 255u) == 255u) {
                    timespec t1;
                    clock_gettime(CLOCK_MONOTONIC, &t1);
                    if ((t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) > linger_ns) break;
                }
            }
        }
    }
    int ensure(int n)                                  // helpers that exist afterwards (<= n)
    {
        n = std::min(n, kCrewMax - 1);
        try {
            while ((int) th.size() < n) { const int who = (int) th.size() + 1; th.emplace_back([this, who] { run(who); }); }
        } catch (...) { }                              // no more threads to be had: the caller works with the ones there are
        return (int) th.size();
    }
    void post(std::shared_ptr<StagedCall> p)
    {
        { std::lock_guard<std::mutex> lk(mu); cur = std::move(p); gen.fetch_add(1, std::memory_order_acq_rel); }
        cv.notify_all();
    }
    void stop()
    {
        if (th.empty()) return;
        { std::lock_guard<std::mutex> lk(mu); quit = true; if (cur) cur->finished.store(1); cur.reset(); }
        cv.notify_all();
        for (auto &t : th) t.join();
        th.clear();
        quit = false;
    }
};

struct OfflineEngineGuard {
    irb_fdl *eng = nullptr;                            // created with IRB_FDL_HOST_IO: its mapped pinned staging [input | output][C][Tmax*B] is what the kernels read and write
    std::vector<hipEvent_t> ev;                        // phase boundaries of the pieces (timing; created on first use): 4 per piece
    Crew crew;
    irb_convolve_profile prof;
    bool have_prof = false;
    // diagnostic (IRB_DEBUG=input_checksums=1): per-block checksums of the samples the forward-FFT kernels read in the last call, [C][steps]
    unsigned *d_sums = nullptr;
    size_t d_sums_cap = 0;
    std::vector<unsigned> sums;
    int sums_streams = 0, sums_blocks = 0;
    void release()
    {
        crew.stop();
        if (d_sums) { (void) hipFree(d_sums); d_sums = nullptr; d_sums_cap = 0; }
        if (!eng) return;
        DeviceQuiet quiet(eng->device);                // frees wait for every stream of the device: resident servers step aside
        (void) hipSetDevice(eng->device);
        (void) irb_sync(eng);
        irb_fdl_destroy(eng);
        eng = nullptr;
        for (auto &e : ev) if (e) (void) hipEventDestroy(e);
        ev.clear();
    }
    ~OfflineEngineGuard() { release(); }
};
thread_local OfflineEngineGuard g_offline;
std::atomic<bool> g_offline_profile_events{false};     // irb_convolve_profile_events(): read the chunks' device events into the profile
constexpr int kOfflineMaxBlocks = 8192;                   // blocks per piece at most (two pieces' staging exist side by side)
constexpr int kOfflineTimedPieces = 16;                   // pieces whose phases the device-event profile follows one by one
}

extern "C" {

const char *irb_last_error(void) { return g_err; }
int irb_set_error_(int code, const char *msg) { return fail(code, "%s", msg); }   // for rt_adapter.hip
const char *irb_version(void) { return "irbaboon_amd 0.1 (gfx950)"; }

void irb_fdl_destroy(irb_fdl *h)
{
    if (!h) return;
    (void) hipSetDevice(h->device);
    (void) server_stop(h);
    registry_remove(h);                                  // (no other thread posts into this handle's mailbox from here on)
    if (h->stream) (void) irb_sync(h);
    DeviceQuiet quiet(h->device);                        // the frees below wait for every stream of the device: other instances' servers step aside
    (void) hipFree(h->d_H); (void) hipFree(h->d_X); (void) hipFree(h->d_tw); (void) hipFree(h->d_part);
    (void) hipFree(h->d_Xt); (void) hipFree(h->d_Ht);
    (void) hipFree(h->d_ovl); (void) hipFree(h->d_in); (void) hipFree(h->d_out); (void) hipFree(h->d_irtime);
    (void) hipFree(h->d_stream_info); (void) hipFree(h->d_tickets); (void) hipFree(h->d_gran);
    if (h->h_status) (void) hipHostFree(h->h_status);
    if (h->h_done) (void) hipHostFree(h->h_done);
    if (h->h_ck) (void) hipHostFree(h->h_ck);
    if (h->h_pk) (void) hipHostFree(h->h_pk);
    (void) hipFree(h->d_ck);
    if (h->stream_in) { (void) hipStreamSynchronize(h->stream_in); (void) hipStreamDestroy(h->stream_in); }
    for (auto &e : h->ev_pipe) if (e) (void) hipEventDestroy(e);
    if (h->bar_in) (void) hipFree(h->bar_in);
    if (h->rt_stream) (void) hipStreamDestroy(h->rt_stream);
    if (h->h_io) (void) hipHostFree(h->h_io);
    if (h->h_ir) (void) hipHostFree(h->h_ir);
    if (h->h_info_slot) (void) hipHostFree(h->h_info_slot);
    for (auto &e : h->ev_ir) if (e) (void) hipEventDestroy(e);
    if (h->h_set_of_stream) (void) hipHostFree(h->h_set_of_stream);
    if (h->h_np_of_set) (void) hipHostFree(h->h_np_of_set);
    if (h->h_info) (void) hipHostFree(h->h_info);
    for (auto &e : h->ev) if (e) (void) hipEventDestroy(e);
    if (h->own_stream && h->stream) (void) hipStreamDestroy(h->stream);
    delete h;
}

int irb_fdl_create(const irb_fdl_config *cfg, irb_fdl **out)
{
    if (!cfg || !out) return fail(IRB_ERR_ARG, "null argument");
    *out = nullptr;
    if (cfg->n_instances < 1 || cfg->channels < 1 || cfg->block < 1 || cfg->max_ir_len < 1 || cfg->max_blocks_per_call < 1)
        return fail(IRB_ERR_ARG, "n_instances, channels, block, max_ir_len and max_blocks_per_call must be >= 1");
    if (cfg->block > 2048) return fail(IRB_ERR_ARG, "block %d > 2048 not supported", cfg->block);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev) return fail(IRB_ERR_HIP, "device %d not present (%d HIP devices)", cfg->device, ndev);
    HIP_TRY(hipSetDevice(cfg->device));

    irb_fdl *h = new (std::nothrow) irb_fdl;
    if (!h) return fail(IRB_ERR_ARG, "out of host memory");
    h->device = cfg->device;
    h->I = cfg->n_instances; h->C = cfg->channels; h->S = h->I * h->C;
    h->B = cfg->block; h->N = fft_len_for_block(h->B); h->M = h->N / 2;
    h->P = (cfg->max_ir_len + h->B - 1) / h->B;
    h->Ppad = (h->P + kPadP - 1) / kPadP * kPadP;
    h->Tmax = cfg->max_blocks_per_call;
    h->flags = cfg->flags;
    h->stepwise = (h->flags & IRB_FDL_STEPWISE) != 0 || h->Tmax == 1;
    h->R = h->stepwise ? h->P : h->Ppad + h->Tmax + kRingSpare;     // tiled path: P past blocks + the call's blocks + tile overhang
    choose_path(h);
    h->nt = (size_t) h->S * h->M * sizeof(float2) * (size_t) (h->Ppad + h->R) > ((size_t) 256 << 20);
    { bool set = false; const int v = dbg_int("fused_nt", 0, &set); if (set) h->nt = v != 0; }     // tuning override
    h->tiled1 = h->nt;
    { bool set = false; const int v = dbg_int("tiled1", 0, &set); if (set) h->tiled1 = v != 0; }
    h->debug_drop_at = dbg_int("drop_granules", 0);
    // at most one workgroup per CU (no second workgroup to overlap its loads with): twice the rows in flight per thread
    // (profiles/r03_batch_sweep.txt: 256 streams 945 -> 1060 Msamples/s; as soon as some CUs hold two workgroups U = 2 is faster)
    h->fused_u = dbg_int("fused_u", (h->colW == 0 && h->S <= 256) ? 4 : 2);
    if (h->fused_u != 4) h->fused_u = 2;
    h->tile8_fused = dbg_int("tiled_fused8", 1) != 0;
    h->mfma_min_T = std::max(0, dbg_int("mfma_min_t", 256));
    h->mfma_wgs = std::max(1, dbg_int("mfma_wgs", 256));
    h->mfma_min_chunks = std::max(1, dbg_int("mfma_min_chunks", 6));
    // gfx950 only (what the library is built for): the instruction timings and the DPP / buffer-load forms the kernel is written in are that chip's
    h->mfma_capable = !h->stepwise && h->mfma_min_T > 0 && h->Tmax >= h->mfma_min_T && h->M % irb::kMfmaBins == 0;
    if (h->mfma_capable) {
        h->xt_hist = irb::mfma_hist(h->P);
        h->xt_len = h->xt_hist + (h->Tmax + irb::kMfmaTile - 1) / irb::kMfmaTile * irb::kMfmaTile;
        h->ht_len = irb::mfma_taps(h->P);
    }
    h->tiled_chip_wgs = 512;     // resident workgroups of k_cmac_tiled_pk: 256 CUs x the two per CU its registers allow

    int rc = IRB_OK;
    auto guard = [&](int r) { if (r != IRB_OK && rc == IRB_OK) rc = r; return r == IRB_OK; };
    do {
        if (cfg->stream) { h->stream = static_cast<hipStream_t>(cfg->stream); }
        else {
            hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
            if (e != hipSuccess) { rc = fail(IRB_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); break; }
            h->own_stream = true;
        }
        for (auto &e : h->ev) if (hipEventCreate(&e) != hipSuccess) { rc = fail(IRB_ERR_HIP, "hipEventCreate failed"); break; }
        if (rc != IRB_OK) break;
        const size_t SM = (size_t) h->S * h->M;
        if (!guard(dev_alloc(h, &h->d_H, SM * h->Ppad + (size_t) 8 * h->M))) break;     // + spare rows: the tiled MAC's last prefetch reads past the last set
        if (!guard(dev_alloc(h, &h->d_X, SM * h->R))) break;
        if (!guard(dev_alloc(h, &h->d_tw, (size_t) h->N))) break;
        if (h->mfma_capable && !guard(dev_alloc(h, &h->d_Xt, SM * h->xt_len))) break;
        if (h->mfma_capable && !guard(dev_alloc(h, &h->d_Ht, SM * h->ht_len))) break;
        {
            const int L = h->M / 2, LW = std::min(L, irb::kThreads), stripes = irb::kThreads / LW;
            size_t rows = (size_t) h->S * std::max(h->splits, 1);
            for (int T = 2; T <= h->Tmax && !h->stepwise; T++) {
                if (use_mfma(h, T)) rows = std::max(rows, mfma_plan(h, T).part_rows(h->S, T));
                else rows = std::max(rows, (size_t) h->S * T * tiled_splits(h, T) * stripes);
            }
            h->part_rows = rows;
        }
        if (!guard(dev_alloc(h, &h->d_part, h->part_rows * h->M))) break;
        if (!guard(dev_alloc(h, &h->d_ovl, (size_t) 2 * h->S * h->B))) break;
        if (!guard(dev_alloc(h, &h->d_in, (size_t) h->S * h->Tmax * h->B))) break;
        if (!guard(dev_alloc(h, &h->d_out, (size_t) h->S * h->Tmax * h->B))) break;
        if (!guard(dev_alloc(h, &h->d_irtime, (size_t) h->C * h->P * h->B))) break;
        if (!guard(dev_alloc(h, &h->d_stream_info, (size_t) h->S))) break;
        if (!guard(dev_alloc(h, &h->d_tickets, (size_t) h->S))) break;
        if (h->colW > 0 && !guard(dev_alloc(h, &h->d_gran, (size_t) h->S * h->splits * h->N))) break;
        hipError_t e;
        h->host_io = (h->flags & IRB_FDL_HOST_IO) != 0;
        const unsigned hflags = h->host_io ? hipHostMallocMapped : hipHostMallocDefault;
        if ((e = hipHostMalloc(reinterpret_cast<void **>(&h->h_io), sizeof(float) * (size_t) (h->host_io ? 2 : 1) * h->S * h->Tmax * h->B, hflags)) != hipSuccess ||
            (e = hipHostMalloc(reinterpret_cast<void **>(&h->h_ir), sizeof(float) * (size_t) 2 * h->C * h->P * h->B, hflags)) != hipSuccess ||
            (e = hipHostMalloc(reinterpret_cast<void **>(&h->h_info_slot), sizeof(int2) * (size_t) 2 * h->C)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&h->ev_ir[0], hipEventDisableTiming)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&h->ev_ir[1], hipEventDisableTiming)) != hipSuccess ||
            (h->host_io && (e = hipHostGetDevicePointer(reinterpret_cast<void **>(&h->d_hio), h->h_io, 0)) != hipSuccess) ||
            (h->host_io && (e = hipHostGetDevicePointer(reinterpret_cast<void **>(&h->d_hir), h->h_ir, 0)) != hipSuccess) ||
            (e = hipHostMalloc(reinterpret_cast<void **>(&h->h_set_of_stream), sizeof(int) * h->S)) != hipSuccess ||
            (e = hipHostMalloc(reinterpret_cast<void **>(&h->h_np_of_set), sizeof(int) * h->S)) != hipSuccess ||
            (e = hipHostMalloc(reinterpret_cast<void **>(&h->h_info), sizeof(int2) * h->S)) != hipSuccess ||
            (e = hipHostMalloc(reinterpret_cast<void **>(&h->h_status), sizeof(unsigned), hipHostMallocMapped)) != hipSuccess ||
            (e = hipHostGetDevicePointer(reinterpret_cast<void **>(&h->d_status), h->h_status, 0)) != hipSuccess ||
            (e = hipHostMalloc(reinterpret_cast<void **>(&h->h_done), 64, hipHostMallocMapped)) != hipSuccess ||
            (e = hipHostGetDevicePointer(reinterpret_cast<void **>(&h->d_done), h->h_done, 0)) != hipSuccess) {
            rc = fail(IRB_ERR_HIP, "hipHostMalloc: %s", hipGetErrorString(e));
            break;
        }
        *h->h_status = 0;
        *h->h_done = 0;
        if (h->host_io && !h->stepwise) {                 // the one-shot function's engine
            if ((e = hipHostMalloc(reinterpret_cast<void **>(&h->h_ck), sizeof(unsigned) * 16 * 2 * kCkMax, hipHostMallocMapped)) != hipSuccess ||
                (e = hipHostGetDevicePointer(reinterpret_cast<void **>(&h->d_hck), h->h_ck, 0)) != hipSuccess ||
                (e = hipHostMalloc(reinterpret_cast<void **>(&h->h_pk), sizeof(unsigned) * 16 * 2 * kCkMax, hipHostMallocMapped)) != hipSuccess ||
                (e = hipHostGetDevicePointer(reinterpret_cast<void **>(&h->d_hpk), h->h_pk, 0)) != hipSuccess ||
                (e = hipMalloc(reinterpret_cast<void **>(&h->d_ck), sizeof(unsigned) * 2 * kCkMax)) != hipSuccess ||
                (e = hipMemsetAsync(h->d_ck, 0, sizeof(unsigned) * 2 * kCkMax, h->stream)) != hipSuccess ||
                (e = hipStreamCreateWithFlags(&h->stream_in, hipStreamNonBlocking)) != hipSuccess) {
                rc = fail(IRB_ERR_HIP, "chunk words: %s", hipGetErrorString(e));
                break;
            }
            for (auto &ev : h->ev_pipe) if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e != hipSuccess) { rc = fail(IRB_ERR_HIP, "pipeline events: %s", hipGetErrorString(e)); break; }
            std::memset(h->h_ck, 0, sizeof(unsigned) * 16 * 2 * kCkMax);
            std::memset(h->h_pk, 0, sizeof(unsigned) * 16 * 2 * kCkMax);
        }
        if (h->host_io && h->stepwise && dbg_int("rt_bar", 1)) {     // (the real-time callback's staging; irb_convolve_periodic's engine is host_io too but never stepwise)
            int large_bar = 0;
            if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, h->device) == hipSuccess && large_bar) {
                const size_t nin = ((size_t) h->S * h->Tmax * h->B + 15) / 16 * 16, nir = ((size_t) h->P * h->B + 15) / 16 * 16;
                if (hipExtMallocWithFlags(reinterpret_cast<void **>(&h->bar_in), sizeof(float) * (nin + nir) + sizeof(irb::RtMailbox), hipDeviceMallocFinegrained) == hipSuccess) {
                    h->bar_ir = h->bar_in + nin;
                    h->bar_mb = reinterpret_cast<irb::RtMailbox *>(h->bar_ir + nir);
                    h->bytes += sizeof(float) * (nin + nir) + sizeof(irb::RtMailbox);
                    std::memset(h->bar_mb, 0, sizeof(irb::RtMailbox));          // through the BAR, like everything the host puts there
                    __builtin_ia32_sfence();
                    if ((h->flags & IRB_FDL_RESIDENT) && hipStreamCreateWithFlags(&h->rt_stream, hipStreamNonBlocking) == hipSuccess) {
                        h->resident = true;
                        h->idle_ticks = (unsigned) std::max(1, env_int("IRB_RT_SERVER_IDLE_US", 20000)) * 100u;
                        // lifetime cap: the longest a device-wide wait made OUTSIDE this library (a host's own hipFree or
                        // hipDeviceSynchronize) can be held up by continuous callbacks; a generation change costs one launch
                        h->life_ticks = (unsigned) std::max(1, env_int("IRB_RT_SERVER_LIFE_MS", 250)) * 100000u;
                        registry_add(h);
                    }
                } else {
                    (void) hipGetLastError();                    // no such memory here: the pinned staging serves
                    h->bar_in = nullptr;
                }
            }
        }
        h->part_slot_busy.assign((size_t) h->P, 0);
        // twiddles: W_N^j in double, rounded to float
        std::vector<float2> tw(h->N);
        for (int j = 0; j < h->N; j++) {
            const double ph = -2.0 * 3.14159265358979323846 * (double) j / (double) h->N;
            tw[j] = make_float2((float) std::cos(ph), (float) std::sin(ph));
        }
        if ((e = hipMemcpy(h->d_tw, tw.data(), sizeof(float2) * h->N, hipMemcpyHostToDevice)) != hipSuccess ||
            (e = hipMemsetAsync(h->d_H, 0, sizeof(float2) * (SM * h->Ppad + (size_t) 8 * h->M), h->stream)) != hipSuccess ||
            (e = hipMemsetAsync(h->d_X, 0, sizeof(float2) * SM * h->R, h->stream)) != hipSuccess ||
            (h->d_Xt && (e = hipMemsetAsync(h->d_Xt, 0, sizeof(float2) * SM * h->xt_len, h->stream)) != hipSuccess) ||
            (h->d_Ht && (e = hipMemsetAsync(h->d_Ht, 0, sizeof(float2) * SM * h->ht_len, h->stream)) != hipSuccess) ||
            (e = hipMemsetAsync(h->d_ovl, 0, sizeof(float) * (size_t) 2 * h->S * h->B, h->stream)) != hipSuccess ||
            (e = hipMemsetAsync(h->d_tickets, 0, sizeof(unsigned) * (size_t) h->S, h->stream)) != hipSuccess ||
            (h->d_gran && (e = hipMemsetAsync(h->d_gran, 0, sizeof(unsigned long long) * (size_t) h->S * h->splits * h->N, h->stream)) != hipSuccess) ||
            (e = hipMemsetAsync(h->d_in, 0, sizeof(float) * (size_t) h->S * h->Tmax * h->B, h->stream)) != hipSuccess ||
            (e = hipMemsetAsync(h->d_out, 0, sizeof(float) * (size_t) h->S * h->Tmax * h->B, h->stream)) != hipSuccess) {
            rc = fail(IRB_ERR_HIP, "device init: %s", hipGetErrorString(e));
            break;
        }
        for (int s = 0; s < h->S; s++) { h->h_set_of_stream[s] = s; h->h_np_of_set[s] = 0; }   // no IR yet: output is silence
        if (!guard(upload_maps(h))) break;
        if ((e = irb_sync(h)) != hipSuccess) { rc = fail(IRB_ERR_HIP, "sync: %s", hipGetErrorString(e)); break; }
    } while (0);
    if (rc != IRB_OK) { char keep[sizeof g_err]; std::memcpy(keep, g_err, sizeof keep); irb_fdl_destroy(h); std::memcpy(g_err, keep, sizeof keep); return rc; }
    *out = h;
    return IRB_OK;
}

int irb_fdl_get_info(const irb_fdl *h, irb_fdl_info *info)
{
    if (!h || !info) return fail(IRB_ERR_ARG, "null argument");
    info->fft_len = h->N; info->partitions = h->P; info->ring_slots = h->R; info->streams = h->S;
    info->splits = h->colW > 0 ? (h->M / 2 / std::min(h->colW, h->M / 2)) * h->splits : h->splits;
    info->device_bytes = h->bytes;
    info->algorithmic_bytes_per_block_step = 8.0 * h->P * h->N + 16.0 * h->B;
    return IRB_OK;
}

const char *irb_fdl_mac_kernel_name(const irb_fdl *h, int n_blocks)
{
    if (!h || n_blocks < 1) return "";
    const int T = h->stepwise ? 1 : std::min(n_blocks, h->Tmax);
    if (T == 1) {
        if (h->colW > 0) return "k_step_cols";
        return (h->pair && h->N <= 2048) ? "k_step_fused_pair" : "k_step_fused";
    }
    if (use_tile8_fused(h, T)) return "k_tile8_fused";
    if (use_mfma(h, T)) return "k_cmac_mfma";
    if (h->tiled1 && T >= 16 && h->N >= 128 && h->N <= 2048) return "k_cmac_tiled1<16>";
    return tile_outputs(h, T) == 8 ? "k_cmac_tiled_pk<8,4>" : "k_cmac_tiled_pk<16,8>";
}

int irb_fdl_set_ir(irb_fdl *h, int instance, const float *const *ir, int ir_channels, int len)
{
    if (!h || !ir) return fail(IRB_ERR_ARG, "null argument");
    if (instance < 0 || instance >= h->I) return fail(IRB_ERR_ARG, "instance %d out of range [0,%d)", instance, h->I);
    if (len < 0 || len > h->P * h->B) return fail(IRB_ERR_ARG, "IR length %d exceeds max_ir_len capacity %d", len, h->P * h->B);
    // fp/convolution.cpp:28-42: only mono/stereo IR x mono/stereo audio are known layouts
    if (!((ir_channels == 1 || ir_channels == 2) && (h->C == 1 || h->C == 2)))
        return fail(IRB_ERR_LAYOUT, "Either audio (%d ch) or IR (%d ch) is not mono nor stereo", h->C, ir_channels);
    for (int c = 0; c < ir_channels; c++) if (!ir[c] && len > 0) return fail(IRB_ERR_ARG, "ir[%d] is null", c);
    HIP_TRY(hipSetDevice(h->device));
    STOP_SERVER(h);
    // Asynchronous: the IR goes through one of two pinned staging slots, and the only wait is for the work that read
    // THIS slot two calls ago (an event, not the stream), so a host thread loading thousands of IRs keeps the copy
    // engine and the FFT kernel busy instead of waiting three times per instance.
    const int slot = h->ir_slot;
    if (slot == 0 && h->part_busy) HIP_TRY(irb_sync(h));             // a rolling partition refresh may still read slot 0
    if (h->ir_used[slot]) { h->n_waits++; HIP_TRY(hipEventSynchronize(h->ev_ir[slot])); }
    float *stage = h->h_ir + (size_t) slot * h->C * h->P * h->B;
    const int np = (len + h->B - 1) / h->B;
    const int nrows = (ir_channels == 2 && h->C == 2) ? 2 : 1;        // fp/convolution.cpp:94-101
    const size_t row = (size_t) np * h->B;
    for (int r = 0; r < nrows; r++) {
        float *dst = stage + r * row;
        std::memcpy(dst, ir[r], sizeof(float) * (size_t) len);
        std::fill(dst + len, dst + row, 0.0f);
    }
    if (ir_channels == 2 && h->C == 1) {                              // tools::sumToMono, fp/tools.cpp:25-28
        for (int i = 0; i < len; i++) { stage[i] += ir[1][i]; stage[i] /= 2.0f; }
    }
    const int first_set = instance * h->C;
    if (np > 0) {
        int rc;
        if (h->host_io && h->d_hir) rc = launch_ir_fft(h, h->d_hir + (stage - h->h_ir), np, nrows, first_set, 0);   // the kernel reads the pinned staging itself: no copy command in front of it
        else {
            HIP_TRY(COPY_ASYNC(h, h->d_irtime, stage, sizeof(float) * row * nrows, hipMemcpyHostToDevice, h->stream));
            rc = launch_ir_fft(h, h->d_irtime, np, nrows, first_set, 0);
        }
        if (rc != IRB_OK) return rc;
    }
    // the time-tiled kernels walk partitions in groups of kPadP: rows up to the next multiple stay zero
    const int np_pad = std::min((np + kPadP - 1) / kPadP * kPadP, h->Ppad);
    for (int r = 0; r < nrows && np_pad > np; r++)
        HIP_TRY(hipMemsetAsync(h->d_H + ((size_t) (first_set + r) * h->Ppad + np) * h->M, 0,
                               sizeof(float2) * (size_t) (np_pad - np) * h->M, h->stream));
    { int rc = launch_ht(h, first_set, nrows, np); if (rc != IRB_OK) return rc; }
    int2 *info = h->h_info_slot + (size_t) slot * h->C;
    for (int c = 0; c < h->C; c++) {
        h->h_set_of_stream[first_set + c] = first_set + (nrows == 2 ? c : 0);   // fp/convolution.cpp:176-181
        h->h_np_of_set[first_set + c] = np;
    }
    for (int c = 0; c < h->C; c++) info[c] = make_int2(h->h_set_of_stream[first_set + c], h->h_np_of_set[h->h_set_of_stream[first_set + c]]);
    update_pairing(h);
    HIP_TRY(COPY_ASYNC(h, h->d_stream_info + first_set, info, sizeof(int2) * h->C, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipEventRecord(h->ev_ir[slot], h->stream));
    h->ir_used[slot] = true;
    h->ir_slot ^= 1;
    return IRB_OK;
}

// irb_fdl_set_ir's work for instance 0 of the one-shot function's engine when the caller's impulse response is device-readable
// (registered for the call): ONE launch transforms it where it lies (k_ir_fft_direct: no pinned staging copy on the host, no copy
// command, zero rows and stream records written by the same launch) + the tap-contiguous copy for the matrix cores.  d_ir: device
// pointers of the caller's channels; they must stay valid until the stream has been waited for.
static int set_ir_direct(irb_fdl *h, const float *const *d_ir, int ir_channels, int len)
{
    if (len < 0 || len > h->P * h->B) return fail(IRB_ERR_ARG, "IR length %d exceeds max_ir_len capacity %d", len, h->P * h->B);
    if (!((ir_channels == 1 || ir_channels == 2) && (h->C == 1 || h->C == 2)))
        return fail(IRB_ERR_LAYOUT, "Either audio (%d ch) or IR (%d ch) is not mono nor stereo", h->C, ir_channels);
    const int np = (len + h->B - 1) / h->B;
    const int nrows = (ir_channels == 2 && h->C == 2) ? 2 : 1;        // fp/convolution.cpp:94-101
    const int np_pad = std::min((np + kPadP - 1) / kPadP * kPadP, h->Ppad);
    for (int c = 0; c < h->C; c++) {
        h->h_set_of_stream[c] = nrows == 2 ? c : 0;                   // fp/convolution.cpp:176-181
        h->h_np_of_set[c] = np;
    }
    update_pairing(h);
    irb::IrDirectArgs a;
    std::memset(&a, 0, sizeof a);
    a.src[0] = d_ir[0]; a.src[1] = ir_channels == 2 ? d_ir[1] : d_ir[0];
    a.H = h->d_H; a.tw = h->d_tw; a.stream_info = h->d_stream_info;
    a.ninfo = h->C;
    for (int c = 0; c < h->C; c++) a.info[c] = make_int2(h->h_set_of_stream[c], h->h_np_of_set[h->h_set_of_stream[c]]);
    a.average = (ir_channels == 2 && h->C == 1) ? 1 : 0;              // tools::sumToMono, fp/tools.cpp:25-28
    a.nrows = nrows; a.len = len; a.B = h->B; a.P = h->Ppad; a.first_set = 0; a.np = np; a.np_pad = np_pad;
    h->n_launches++;
    const int items = std::max(1, nrows * np_pad);
    FOR_N(h->N, irb::k_ir_fft_direct<NN><<<dim3((items + irb::FftGeom<NN>::NF - 1) / irb::FftGeom<NN>::NF), dim3(irb::kWave * irb::FftGeom<NN>::NF), 0, h->stream>>>(a));
    HIP_TRY(hipGetLastError());
    return launch_ht(h, 0, nrows, np);
}

int irb_fdl_set_ir_partition(irb_fdl *h, int instance, int partition, const float *samples)
{
    if (!h || !samples) return fail(IRB_ERR_ARG, "null argument");
    if (instance < 0 || instance >= h->I) return fail(IRB_ERR_ARG, "instance %d out of range", instance);
    if (partition < 0 || partition >= h->P) return fail(IRB_ERR_ARG, "partition %d out of range [0,%d)", partition, h->P);
    HIP_TRY(hipSetDevice(h->device));
    STOP_SERVER(h);
    const int first_set = instance * h->C;
    // the staging slot of this partition: a call may be enqueued for P distinct partitions before anything is synchronised;
    // a second refresh of the SAME partition slot (another instance, or the same one again) first waits for the stream,
    // because the kernel enqueued earlier may not have read the slot yet
    if (!h->part_busy) std::fill(h->part_slot_busy.begin(), h->part_slot_busy.end(), 0);
    if (h->part_slot_busy[partition] || h->ir_used[0]) {
        HIP_TRY(irb_sync(h));
        std::fill(h->part_slot_busy.begin(), h->part_slot_busy.end(), 0);
        h->ir_used[0] = false;                                       // whatever irb_fdl_set_ir staged in slot 0 has been consumed
    }
    h->part_slot_busy[partition] = 1;
    h->part_busy = true;
    float *hs = h->h_ir + (size_t) partition * h->B;
    float *ds = h->d_irtime + (size_t) partition * h->B;
    std::memcpy(hs, samples, sizeof(float) * h->B);
    int rc;
    if (h->host_io) {
        rc = launch_ir_fft(h, h->d_hir + (size_t) partition * h->B, 1, 1, first_set, partition);     // the kernel reads the pinned staging itself
    } else {
        HIP_TRY(COPY_ASYNC(h, ds, hs, sizeof(float) * h->B, hipMemcpyHostToDevice, h->stream));
        rc = launch_ir_fft(h, ds, 1, 1, first_set, partition);
    }
    if (rc != IRB_OK) return rc;
    rc = launch_ht(h, first_set, 1, h->P);                           // (engines that take long calls only: the plug-in's are stepwise)
    if (rc != IRB_OK) return rc;
    bool changed = false;
    for (int c = 0; c < h->C; c++) {
        if (h->h_set_of_stream[first_set + c] != first_set || h->h_np_of_set[first_set + c] != h->P) changed = true;
        h->h_set_of_stream[first_set + c] = first_set;               // Source/PluginProcessor.cpp:485
        h->h_np_of_set[first_set + c] = h->P;
    }
    if (changed) return upload_maps(h);
    return IRB_OK;
}

int irb_fdl_get_ir_spectra(irb_fdl *h, int instance, int channel, float *spectra, int n_partitions)
{
    if (!h || !spectra) return fail(IRB_ERR_ARG, "null argument");
    if (instance < 0 || instance >= h->I || channel < 0 || channel >= h->C) return fail(IRB_ERR_ARG, "instance/channel out of range");
    const int set = h->h_set_of_stream[instance * h->C + channel];
    if (n_partitions < 0 || n_partitions > h->h_np_of_set[set])
        return fail(IRB_ERR_ARG, "n_partitions %d exceeds the %d partitions held", n_partitions, h->h_np_of_set[set]);
    HIP_TRY(hipSetDevice(h->device));
    STOP_SERVER(h);
    HIP_TRY(COPY_ASYNC(h, spectra, h->d_H + (size_t) set * h->Ppad * h->M, sizeof(float2) * (size_t) n_partitions * h->M,
                           hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(irb_sync(h));
    return IRB_OK;
}

int irb_fdl_set_ir_spectra(irb_fdl *h, int instance, const float *const *spectra, int ir_channels, int n_partitions)
{
    if (!h || !spectra) return fail(IRB_ERR_ARG, "null argument");
    if (instance < 0 || instance >= h->I) return fail(IRB_ERR_ARG, "instance %d out of range", instance);
    if (n_partitions < 0 || n_partitions > h->P) return fail(IRB_ERR_ARG, "n_partitions %d exceeds capacity %d", n_partitions, h->P);
    if (!((ir_channels == 1 || (ir_channels == 2 && h->C == 2)) && (h->C == 1 || h->C == 2)))
        return fail(IRB_ERR_LAYOUT, "spectra for %d IR channels cannot feed %d audio channels", ir_channels, h->C);
    HIP_TRY(hipSetDevice(h->device));
    STOP_SERVER(h);
    const int first_set = instance * h->C, np = n_partitions;
    for (int r = 0; r < ir_channels; r++) {
        if (!spectra[r] && np > 0) return fail(IRB_ERR_ARG, "spectra[%d] is null", r);
        float2 *dst = h->d_H + (size_t) (first_set + r) * h->Ppad * h->M;
        if (np > 0) HIP_TRY(COPY_ASYNC(h, dst, spectra[r], sizeof(float2) * (size_t) np * h->M, hipMemcpyHostToDevice, h->stream));
        const int np_pad = std::min((np + kPadP - 1) / kPadP * kPadP, h->Ppad);
        if (np_pad > np) HIP_TRY(hipMemsetAsync(dst + (size_t) np * h->M, 0, sizeof(float2) * (size_t) (np_pad - np) * h->M, h->stream));
    }
    { int rc = launch_ht(h, first_set, ir_channels, np); if (rc != IRB_OK) return rc; }
    for (int c = 0; c < h->C; c++) {
        h->h_set_of_stream[first_set + c] = first_set + (ir_channels == 2 ? c : 0);
        h->h_np_of_set[first_set + c] = np;
    }
    int rc = upload_maps(h);
    if (rc != IRB_OK) return rc;
    HIP_TRY(irb_sync(h));
    return IRB_OK;
}

int irb_fdl_reset(irb_fdl *h)
{
    if (!h) return fail(IRB_ERR_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    STOP_SERVER(h);
    // one launch: the ring, both overlaps, (matrix-core engines) the blocks "before" the next call in Xt -- the columns of the calls
    // themselves are rewritten by every call --, (column-sliced engines) the granule tags: the step counter restarts
    irb::ResetArgs r;
    std::memset(&r, 0, sizeof r);
    const size_t ovl_bytes = sizeof(float) * (size_t) 2 * h->S * h->B;
    r.p[0] = reinterpret_cast<float4 *>(h->d_X);   r.n16[0] = sizeof(float2) * (size_t) h->S * h->M * h->R / 16;        // (M >= 16: rows are multiples of 16 bytes)
    r.p[1] = reinterpret_cast<float4 *>(h->d_ovl); r.n16[1] = ovl_bytes / 16;
    if (h->d_gran) { r.p[2] = reinterpret_cast<float4 *>(h->d_gran); r.n16[2] = sizeof(unsigned long long) * (size_t) h->S * h->splits * h->N / 16; }
    if (h->d_Xt) { r.xt = reinterpret_cast<float4 *>(h->d_Xt); r.rows = (size_t) h->S * h->M; r.pitch16 = (size_t) h->xt_len / 2; r.hist16 = (size_t) h->xt_hist / 2; }
    h->n_launches++;
    const size_t work = r.n16[0] + r.n16[1] + r.n16[2] + r.rows * r.hist16;
    irb::k_reset<<<dim3((unsigned) std::min<size_t>(2048, (work + 255) / 256 + 1)), dim3(256), 0, h->stream>>>(r);
    HIP_TRY(hipGetLastError());
    if (ovl_bytes % 16) HIP_TRY(hipMemsetAsync(reinterpret_cast<char *>(h->d_ovl) + ovl_bytes / 16 * 16, 0, ovl_bytes % 16, h->stream));   // (2 S B floats not a multiple of 4: the last few bytes)
    h->t = 0;
    h->ovl_cur = 0;
    return IRB_OK;
}

int irb_fdl_process_device(irb_fdl *h, const float *d_in, float *d_out, long stride, int n_blocks)
{
    if (!h || !d_in || !d_out) return fail(IRB_ERR_ARG, "null argument");
    if (n_blocks < 0) return fail(IRB_ERR_ARG, "n_blocks < 0");
    if (stride < (long) n_blocks * h->B && h->S > 1) return fail(IRB_ERR_ARG, "stride %ld smaller than n_blocks*block", stride);
    HIP_TRY(hipSetDevice(h->device));
    STOP_SERVER(h);
    for (int j = 0; j < n_blocks;) {
        const int T = h->stepwise ? 1 : std::min(n_blocks - j, h->Tmax);
        irb::StepArgs a = step_args(h, d_in + (size_t) j * h->B, d_out + (size_t) j * h->B, stride);
        int rc = (T > 1) ? launch_tiled(h, a, T) : launch_step(h, a);
        if (rc != IRB_OK) return rc;
        h->t += T;
        j += T;
    }
    return IRB_OK;
}

int irb_fdl_synchronize(irb_fdl *h)
{
    if (!h) return fail(IRB_ERR_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    STOP_SERVER(h);
    HIP_TRY(irb_sync(h));
    return check_status(h);
}

// one chunk (nb <= Tmax blocks starting at block `done`) of the host-pointer call, split so that several handles
// can be driven concurrently from one thread: begin = pack + H2D + kernels + D2H (all enqueued), end = wait + unpack
static int process_chunk_begin(irb_fdl *h, const float *const *in, int done, int nb)
{
    HIP_TRY(hipSetDevice(h->device));
    STOP_SERVER(h);
    const long stride = (long) h->Tmax * h->B;
    const size_t n = (size_t) nb * h->B;
    for (int s = 0; s < h->S; s++) std::memcpy(h->h_io + (size_t) s * stride, in[s] + (size_t) done * h->B, sizeof(float) * n);
    if (h->host_io)                                   // zero copy: input half -> kernels -> output half, both in mapped host memory
        return irb_fdl_process_device(h, h->d_hio, h->d_hio + (size_t) h->S * stride, stride, nb);
    HIP_TRY(COPY_ASYNC(h, h->d_in, h->h_io, sizeof(float) * (size_t) h->S * stride, hipMemcpyHostToDevice, h->stream));
    int rc = irb_fdl_process_device(h, h->d_in, h->d_out, stride, nb);
    if (rc != IRB_OK) return rc;
    HIP_TRY(COPY_ASYNC(h, h->h_io, h->d_out, sizeof(float) * (size_t) h->S * stride, hipMemcpyDeviceToHost, h->stream));
    return IRB_OK;
}

static int process_chunk_end(irb_fdl *h, float *const *out, int done, int nb)
{
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(irb_sync(h));      // (polling hipStreamQuery instead was measured: no gain)
    const long stride = (long) h->Tmax * h->B;
    const size_t n = (size_t) nb * h->B;
    const float *res = h->h_io + (h->host_io ? (size_t) h->S * stride : 0);
    for (int s = 0; s < h->S; s++) std::memcpy(out[s] + (size_t) done * h->B, res + (size_t) s * stride, sizeof(float) * n);
    return check_status(h);
}

int irb_fdl_process(irb_fdl *h, const float *const *in, float *const *out, int n_blocks)
{
    if (!h || !in || !out) return fail(IRB_ERR_ARG, "null argument");
    if (n_blocks < 0) return fail(IRB_ERR_ARG, "n_blocks < 0");
    for (int s = 0; s < h->S; s++) if (!in[s] || !out[s]) return fail(IRB_ERR_ARG, "in/out[%d] is null", s);
    for (int done = 0; done < n_blocks; done += h->Tmax) {
        const int nb = std::min(h->Tmax, n_blocks - done);
        int rc = process_chunk_begin(h, in, done, nb);
        if (rc == IRB_OK) rc = process_chunk_end(h, out, done, nb);
        if (rc != IRB_OK) return rc;
    }
    return IRB_OK;
}

// Wait for the completion word of a k_rt_pair launch: the audio thread spins on host memory (the kernel's last store
// lands there ~1 us after its last output sample) instead of entering the runtime's wait.  Bounded: after 2 ms the
// runtime's own wait takes over, which also surfaces a failed launch.
static int wait_done_word(irb_fdl *h, unsigned seq)
{
    volatile unsigned *w = h->h_done;
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (unsigned spins = 0; *w != seq; spins++) {
        __builtin_ia32_pause();
        if ((spins & 1023u) == 1023u) {
            timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) > 2000000L) {
                HIP_TRY(irb_sync(h));
                if (*w != seq) return fail(IRB_ERR_STATE, "real-time step finished without its completion word");
                break;
            }
        }
    }
    // let the runtime retire its finished commands now and then (nothing else on this path ever asks it to)
    if ((seq & 255u) == 0) (void) hipStreamQuery(h->stream);
    return IRB_OK;
}

int irb_fdl_process_refresh(irb_fdl *h, int instance, const float *const *in, float *const *out, int n_blocks,
                            const float *ir, int first_partition)
{
    if (!h || !in || !out || !ir) return fail(IRB_ERR_ARG, "null argument");
    if (instance < 0 || instance >= h->I) return fail(IRB_ERR_ARG, "instance %d out of range", instance);
    if (n_blocks < 0 || n_blocks > h->Tmax) return fail(IRB_ERR_ARG, "n_blocks %d outside [0, max_blocks_per_call = %d]", n_blocks, h->Tmax);
    if (first_partition < 0 || first_partition >= h->P) return fail(IRB_ERR_ARG, "partition %d out of range [0,%d)", first_partition, h->P);
    if (!h->stepwise) return fail(IRB_ERR_STATE, "irb_fdl_process_refresh needs IRB_FDL_STEPWISE (or max_blocks_per_call = 1)");
    for (int s = 0; s < h->S; s++) if (!in[s] || !out[s]) return fail(IRB_ERR_ARG, "in/out[%d] is null", s);
    if (n_blocks == 0) return IRB_OK;
    HIP_TRY(hipSetDevice(h->device));
    const int first_set = instance * h->C;
    bool changed = false;                                // Source/PluginProcessor.cpp:485: IR channel 0 for every channel, all P rows in use
    for (int c = 0; c < h->C; c++) {
        if (h->h_set_of_stream[first_set + c] != first_set || h->h_np_of_set[first_set + c] != h->P) changed = true;
        h->h_set_of_stream[first_set + c] = first_set;
        h->h_np_of_set[first_set + c] = h->P;
    }
    if (changed) { int rc = upload_maps(h); if (rc != IRB_OK) return rc; }

    const long stride = (long) h->Tmax * h->B;
    const size_t n = (size_t) n_blocks * h->B;
    // one launch for the whole call: a single stereo or mono instance whose step runs in one workgroup, blocks and IR
    // samples in mapped host memory (what the plug-in adapter creates)
    const bool one_launch = h->host_io && h->I == 1 && (h->pair || h->C == 1) && h->colW == 0 && h->splits <= 1 && h->N <= 1024 &&
                            !(h->flags & IRB_FDL_NO_RT_KERNEL);
    if (h->ir_used[0]) { h->n_waits++; HIP_TRY(hipEventSynchronize(h->ev_ir[0])); h->ir_used[0] = false; }   // slot 0 of the IR staging is about to be rewritten
    // (not while a quiet section is open on this device -- another thread of the process is inside a device-wide runtime wait
    // that a live server would hold up: the callback then takes the one-launch form below)
    if (one_launch && h->resident && !(device_quiet(h->device) && (!h->server_live || *exited_word(h) == h->server_gen))) {
        // the resident server: post the request into the BAR mailbox (samples and parameters, a store fence, the request
        // number), make sure a server generation is alive, spin on the completion word
        for (int s = 0; s < h->S; s++) std::memcpy(h->bar_in + (size_t) s * stride, in[s], sizeof(float) * n);
        for (int j = 0; j < n_blocks; j++) {
            const int p = (first_partition + j) % h->P;
            std::memcpy(h->bar_ir + (size_t) p * h->B, ir + (size_t) p * h->B, sizeof(float) * h->B);
        }
        unsigned seq = ++h->done_seq;
        if (seq == 0) seq = ++h->done_seq;
        h->bar_mb->n_blocks = n_blocks; h->bar_mb->first_partition = first_partition; h->bar_mb->head = (int) (h->t % h->R);
        __builtin_ia32_sfence();
        h->bar_mb->seq = seq;
        __builtin_ia32_sfence();
        volatile unsigned *done = h->h_done, *exited = exited_word(h);
        bool served = true;
        if (!h->server_live || *exited == h->server_gen) { int rc = server_launch(h, seq); if (rc != IRB_OK) return rc; }
        h->n_waits++;
        timespec t0;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (unsigned spins = 0; *done != seq; spins++) {
            __builtin_ia32_pause();
            if ((spins & 255u) != 255u) continue;
            if (*exited == h->server_gen && *done != seq) {          // that generation left (idle / lifetime / yield) without this request
                if (device_quiet(h->device)) { served = false; break; }     // ... because the device is wanted quiet: one launch serves this callback
                int rc = server_launch(h, seq);                      // ... otherwise the next generation takes it
                if (rc != IRB_OK) return rc;
                continue;
            }
            timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) > 200000000L) {
                (void) server_stop(h);
                return fail(IRB_ERR_STATE, "real-time server did not answer within 200 ms");
            }
        }
        if (served) {
            h->t += n_blocks;
            const float *res = h->h_io + (size_t) h->S * stride;
            for (int s = 0; s < h->S; s++) std::memcpy(out[s], res + (size_t) s * stride, sizeof(float) * n);
            return IRB_OK;
        }
        // the posted request number stays behind unanswered: later generations start from their own first request and never match it
    }
    STOP_SERVER(h);
    if (one_launch) {
        // blocks and IR partitions: stored by this thread straight into device memory where the BAR allows it (write-only,
        // write-combined; the fence below drains the combining buffers before the launch's doorbell), else left in mapped
        // pinned host memory for the kernel to fetch
        float *in_stage = h->bar_in ? h->bar_in : h->h_io, *ir_stage = h->bar_in ? h->bar_ir : h->h_ir;
        for (int s = 0; s < h->S; s++) std::memcpy(in_stage + (size_t) s * stride, in[s], sizeof(float) * n);
        for (int j = 0; j < n_blocks; j++) {
            const int p = (first_partition + j) % h->P;
            std::memcpy(ir_stage + (size_t) p * h->B, ir + (size_t) p * h->B, sizeof(float) * h->B);
        }
        if (h->bar_in) __builtin_ia32_sfence();
        irb::StepArgs a = step_args(h, h->bar_in ? h->bar_in : h->d_hio, h->d_hio + (size_t) h->S * stride, stride);
        a.T = n_blocks; a.rf_src = h->bar_in ? h->bar_ir : h->d_hir; a.rf_part = first_partition;
        a.done = h->d_done; a.done_seq = ++h->done_seq;
        if (a.done_seq == 0) a.done_seq = ++h->done_seq;            // 0 is the word's initial value
        h->n_launches++;
        if (h->C == 2) { FOR_N_RT(h->N, irb::k_rt_pair<NN, 2><<<dim3(h->I), dim3(irb::kThreads), 0, h->stream>>>(a)); }
        else           { FOR_N_RT(h->N, irb::k_rt_pair<NN, 1><<<dim3(h->I), dim3(irb::kThreads), 0, h->stream>>>(a)); }
        HIP_TRY(hipGetLastError());
        h->t += n_blocks;
        h->n_waits++;
        int rc = wait_done_word(h, a.done_seq);
        if (rc != IRB_OK) return rc;
        const float *res = h->h_io + (size_t) h->S * stride;
        for (int s = 0; s < h->S; s++) std::memcpy(out[s], res + (size_t) s * stride, sizeof(float) * n);
        return IRB_OK;
    }
    // any other engine shape: refresh launch + step launch per block, all enqueued before the single wait
    for (int s = 0; s < h->S; s++) std::memcpy(h->h_io + (size_t) s * stride, in[s], sizeof(float) * n);
    const float *d_src = h->host_io ? h->d_hio : h->d_in;
    float *d_dst = h->host_io ? h->d_hio + (size_t) h->S * stride : h->d_out;
    if (!h->host_io) HIP_TRY(COPY_ASYNC(h, h->d_in, h->h_io, sizeof(float) * (size_t) h->S * stride, hipMemcpyHostToDevice, h->stream));
    for (int j = 0; j < n_blocks; j++) {
        const int p = (first_partition + j) % h->P;
        int rc = irb_fdl_set_ir_partition(h, instance, p, ir + (size_t) p * h->B);
        if (rc != IRB_OK) return rc;
        rc = irb_fdl_process_device(h, d_src + (size_t) j * h->B, d_dst + (size_t) j * h->B, stride, 1);
        if (rc != IRB_OK) return rc;
    }
    if (!h->host_io) HIP_TRY(COPY_ASYNC(h, h->h_io, h->d_out, sizeof(float) * (size_t) h->S * stride, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(irb_sync(h));
    const float *res = h->h_io + (h->host_io ? (size_t) h->S * stride : 0);
    for (int s = 0; s < h->S; s++) std::memcpy(out[s], res + (size_t) s * stride, sizeof(float) * n);
    return check_status(h);
}

int irb_fdl_get_counters(const irb_fdl *h, unsigned long long *launches, unsigned long long *waits, unsigned long long *copies)
{
    if (!h) return fail(IRB_ERR_ARG, "null handle");
    if (launches) *launches = h->n_launches;
    if (waits) *waits = h->n_waits;
    if (copies) *copies = h->n_copies;
    return IRB_OK;
}

int irb_fdl_process_packed(irb_fdl *h, const float *in, float *out, long stride, int n_blocks)
{
    if (!h || !in || !out) return fail(IRB_ERR_ARG, "null argument");
    if (n_blocks < 0) return fail(IRB_ERR_ARG, "n_blocks < 0");
    if (stride < (long) n_blocks * h->B) return fail(IRB_ERR_ARG, "stride %ld smaller than n_blocks*block", stride);
    HIP_TRY(hipSetDevice(h->device));
    STOP_SERVER(h);
    const long dstride = (long) h->Tmax * h->B;
    for (int done = 0; done < n_blocks; done += h->Tmax) {
        const int nb = std::min(h->Tmax, n_blocks - done);
        const size_t width = sizeof(float) * (size_t) nb * h->B;
        HIP_TRY(COPY2D_ASYNC(h, h->d_in, sizeof(float) * dstride, in + (size_t) done * h->B, sizeof(float) * stride, width, h->S,
                                 hipMemcpyHostToDevice, h->stream));
        int rc = irb_fdl_process_device(h, h->d_in, h->d_out, dstride, nb);
        if (rc != IRB_OK) return rc;
        HIP_TRY(COPY2D_ASYNC(h, out + (size_t) done * h->B, sizeof(float) * stride, h->d_out, sizeof(float) * dstride, width, h->S,
                                 hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(irb_sync(h));
        rc = check_status(h);
        if (rc != IRB_OK) return rc;
    }
    return IRB_OK;
}

void *irb_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1) != hipSuccess) { fail(IRB_ERR_HIP, "hipHostMalloc(%zu) failed", bytes); return nullptr; }
    return p;
}

void irb_host_free(void *p)
{
    if (!p) return;
    int dev = 0;
    (void) hipGetDevice(&dev);
    DeviceQuiet quiet(dev);
    (void) hipHostFree(p);
}

int irb_fdl_mark(irb_fdl *h, int slot)
{
    if (!h || slot < 0 || slot > 1) return fail(IRB_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipEventRecord(h->ev[slot], h->stream));
    return IRB_OK;
}

int irb_fdl_elapsed_ms(irb_fdl *h, float *ms)
{
    if (!h || !ms) return fail(IRB_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipEventSynchronize(h->ev[1]));
    HIP_TRY(hipEventElapsedTime(ms, h->ev[0], h->ev[1]));
    return IRB_OK;
}

int irb_fdl_device_io(irb_fdl *h, float **d_in, float **d_out, long *stride)
{
    if (!h || !d_in || !d_out || !stride) return fail(IRB_ERR_ARG, "null argument");
    *d_in = h->d_in; *d_out = h->d_out; *stride = (long) h->Tmax * h->B;
    return IRB_OK;
}

int irb_fdl_upload(irb_fdl *h, float *d_dst, const float *src, size_t n)
{
    if (!h || !d_dst || !src) return fail(IRB_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    STOP_SERVER(h);
    HIP_TRY(COPY_ASYNC(h, d_dst, src, sizeof(float) * n, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(irb_sync(h));
    return IRB_OK;
}

int irb_fdl_download(irb_fdl *h, float *dst, const float *d_src, size_t n)
{
    if (!h || !dst || !d_src) return fail(IRB_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    STOP_SERVER(h);
    HIP_TRY(COPY_ASYNC(h, dst, d_src, sizeof(float) * n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(irb_sync(h));
    return IRB_OK;
}

// ---------------------------------------------------------------------------------------------
// fp::convolution::convolvePeriodic (fp/convolution.cpp:14-242) as one call.
// ---------------------------------------------------------------------------------------------
int irb_convolve_periodic(int device, const float *const *audio, int audio_ch, int La,
                          const float *const *ir, int ir_ch, int Lh, int B, int keep_reference_tail,
                          float *const *out)
{
    if (!audio || !ir || !out) return fail(IRB_ERR_ARG, "null argument");
    if (La < 0 || Lh < 0 || B < 1 || audio_ch < 1) return fail(IRB_ERR_ARG, "negative length or block < 1");
    const long Lout = (long) La + Lh - 1;
    for (int c = 0; c < audio_ch; c++) if (!out[c]) return fail(IRB_ERR_ARG, "out[%d] is null", c);
    auto clear_out = [&](long from) {                                               // :20-21 (what is computed below is not cleared first)
        for (int c = 0; c < audio_ch; c++) if (Lout > from) std::memset(out[c] + from, 0, sizeof(float) * (size_t) (Lout - from));
    };
    if (!((ir_ch == 1 || ir_ch == 2) && (audio_ch == 1 || audio_ch == 2))) {        // :28-42
        clear_out(0);
        return fail(IRB_ERR_LAYOUT, "Either buffer1 or buffer2 is not mono nor stereo. Abort abort");
    }
    for (int c = 0; c < audio_ch; c++)                                              // (the reference returns a NEW buffer: result and audio never share samples)
        for (int d = 0; d < audio_ch; d++)
            if (audio[d] && La > 0 && Lout > 0 && out[c] < audio[d] + La && audio[d] < out[c] + Lout)
                return fail(IRB_ERR_ARG, "out[%d] overlaps audio[%d]: the result buffer must not alias the audio", c, d);
    if (Lh == 0 || La == 0 || Lout <= 0) { clear_out(0); return IRB_OK; }           // nothing to convolve: the cleared output is the answer
    // :52 -- the reference takes ceil of a float ratio; samples beyond P*B are never read (:106-118)
    const int P = (int) std::ceil((float) Lh / (float) B);
    const int Lh_eff = (int) std::min<long>(Lh, (long) P * B);
    const int T_last = La / B;                        // last block that holds audio (possibly empty), :128
    long steps = (long) T_last + P;                   // :233
    const long written = steps * B;
    if (!keep_reference_tail && Lout > written) steps += 1;   // flush the overlap the reference drops (:236-238)
    const long lim = keep_reference_tail ? std::min<long>(Lout, written) : Lout;    // samples this call produces (:225-227)
    clear_out(lim);                                                                 // the dropped tail stays 0 (:236-238)

    // ---- the pipeline ---------------------------------------------------------------------------------------------------
    // The signal goes through in PIECES: per piece the forward FFTs of its blocks ("in": the samples come over PCIe), ONE
    // multiply-accumulate over all of its blocks (the shape the matrix-core kernel is best at), the inverse FFTs + overlap-add ("out":
    // the result leaves over PCIe); the FDL (ring, overlap, the matrix cores' history columns) carries from piece to piece.  Where there
    // are several pieces, piece k + 1's "in" runs on a second stream while piece k's "out" still leaves on the first:
    //     stream_in :  in(0)            in(1)            in(2)      ...        in(k + 1) waits for mul(k): the multiply reads what in() writes
    //     stream    :        mul(0) out(0)   mul(1) out(1)   mul(2) out(2)      mul(k) waits for in(k)
    // Where the samples are: in the caller's own buffers when those are HIP host memory ("direct form"), else in the engine's pinned
    // staging, two pieces side by side, filled and emptied by the threads of StagedCall ("staged form", the default for pageable memory).
    const double t_begin = now_ms();
    irb_convolve_profile prof;
    std::memset(&prof, 0, sizeof prof);
    // ---- direct or staged?  Direct ONLY for buffers that are HIP host memory already (irb_host_alloc / hipHostMalloc, or registered by
    // the caller), or when the caller has opted in to the per-call registration of pageable memory (IRB_OFFLINE_DIRECT=1; exposure:
    // INTEGRATION.md section 1, profiles/NOTES.md "Two wrong one-shot results").
    const DirectMode mode = direct_mode_from_env();
    std::unique_ptr<DeviceQuiet> quiet;
    HostRanges ranges, ir_ranges;
    bool direct = false, ir_direct = false;
    float *d_audio[2] = {nullptr, nullptr}, *d_res[2] = {nullptr, nullptr};
    const float *d_irp[2] = {nullptr, nullptr};
    // resolve the caller's buffers; register == true only AFTER the engine exists (see below)
    auto resolve_direct = [&](bool register_pageable) {
        const double t_reg = now_ms();
        if (register_pageable) {
            // (registering and unregistering host memory are device-wide operations for the runtime: with a resident real-time server
            // alive on the device the call waited until that server's callbacks stopped, so the registering form sits in a quiet section)
            quiet.reset(new DeviceQuiet(device));
            ranges.add(audio, audio_ch, (size_t) La);
            ranges.add(out, audio_ch, (size_t) Lout);
            ranges.add(ir, ir_ch, (size_t) Lh_eff);
            direct = ir_direct = ranges.commit(true);
            if (!direct) {                                           // (a read-only mapping of the IR is refused: the staged IR set-up serves then)
                ranges.release();
                ranges.add(audio, audio_ch, (size_t) La);
                ranges.add(out, audio_ch, (size_t) Lout);
                direct = ranges.commit(true);
            }
        } else {
            ranges.add(audio, audio_ch, (size_t) La);
            ranges.add(out, audio_ch, (size_t) Lout);
            direct = ranges.commit(false);
            if (direct) { ir_ranges.add(ir, ir_ch, (size_t) Lh_eff); ir_direct = ir_ranges.commit(false); }
        }
        for (int c = 0; c < audio_ch && direct; c++) {
            d_audio[c] = ranges.dev(audio[c]);
            d_res[c] = ranges.dev(out[c]);
            direct = d_audio[c] && d_res[c];
        }
        for (int c = 0; c < ir_ch && direct && ir_direct; c++) {
            d_irp[c] = (register_pageable ? ranges : ir_ranges).dev(ir[c]);
            if (!d_irp[c]) ir_direct = false;
        }
        if (!direct) { ranges.release(); ir_ranges.release(); quiet.reset(); ir_direct = false; }   // not HIP host memory (or a registration refused): the staged form
        else { prof.register_ms = now_ms() - t_reg; prof.registered_ranges = ranges.n_registered; }
    };
    if (mode == DirectMode::kPinnedOnly && steps >= 2) resolve_direct(false);     // (look-ups only: nothing of the process's memory changes its status)
    auto release_ranges = [&]() { ranges.release(); ir_ranges.release(); quiet.reset(); direct = ir_direct = false; };

    // ---- how the signal is cut into pieces.  One piece wherever it fits: a piece costs the multiply a whole round of workgroups and the
    // pipeline two cross-stream waits, and the link carries only ~1.4x as much both ways as one way (profiles/r05_oneshot_pieces.txt).
    std::vector<int> piece_T;
    auto plan_pieces = [&]() {
        piece_T.clear();
        // (staged: two pieces' staging exist side by side; override: tests run many pieces on small signals)
        const long cap = std::max(2, dbg_int("offline_max_blocks", direct ? 2 * kOfflineMaxBlocks : kOfflineMaxBlocks));
        const long piece_min = std::max<long>(256, (1L << 18) / B);                        // a piece worth its launches: a MiB of samples per channel
        const long n = std::max<long>(1, std::min<long>(dbg_int("offline_pieces", 1), steps / piece_min));   // (tools: the piece-count sweeps of profiles/r05_oneshot_pieces.txt)
        long Tp = (steps + n - 1) / n;
        if (Tp >= 256) Tp = (Tp + 255) / 256 * 256;                                        // whole tiles of the matrix-core kernel
        Tp = std::min(Tp, cap);
        for (long done = 0; done < steps; done += Tp) piece_T.push_back((int) std::min(Tp, steps - done));
        if (piece_T.size() >= 2 && piece_T.back() == 1 && Tp >= 3) { piece_T[piece_T.size() - 2] -= 1; piece_T.back() = 2; }   // (a one-block piece is a plain step: not in the pipeline)
    };
    plan_pieces();
    {
        bool one_block_piece = false;
        for (int T : piece_T) one_block_piece |= T == 1;
        if (direct && one_block_piece) { release_ranges(); plan_pieces(); }              // (tests that force two-block pieces: a plain step reads whole blocks)
    }
    const int n_pieces = (int) piece_T.size(), Tp = piece_T[0];
    const int Tneed = std::max(2, n_pieces > 1 ? 2 * Tp : Tp);         // blocks of staging per channel (>= 2: a one-block engine would be a stepwise one, without the pipeline's streams)
    // the calling thread's previous engine is reused when it is large enough: creating one (device and pinned
    // allocations) costs ~100x the kernels of a typical call.  irb_release_cached_plans() frees it.
    irb_fdl *h = g_offline.eng;
    if (!(h && h->device == device && h->C == audio_ch && h->B == B && h->P >= P && h->Tmax >= Tneed && h->Tmax <= 4 * Tneed + 1024)) {
        g_offline.release();
        irb_fdl_config cfg;
        std::memset(&cfg, 0, sizeof cfg);
        cfg.device = device; cfg.n_instances = 1; cfg.channels = audio_ch; cfg.block = B; cfg.max_ir_len = P * B;
        cfg.max_blocks_per_call = Tneed < 512 ? Tneed : (Tneed + 1023) / 1024 * 1024;   // (a later call of this block size and IR length that is a little longer still fits)
        cfg.flags = IRB_FDL_HOST_IO;
        int rc0 = irb_fdl_create(&cfg, &h);
        if (rc0 != IRB_OK) { clear_out(0); return rc0; }
        g_offline.eng = h;
        prof.engine_created = 1;
    }
    OfflineEngineGuard &g = g_offline;
    // every error return from here on: nothing of the call may still be reading the staging or writing the result, and the result
    // is the cleared buffer the reference returns when it gives up (:39-42) -- never a half-written one
    std::shared_ptr<StagedCall> sc;                                     // what the helper threads may be working on
    auto bail = [&](int code) {
        char keep[sizeof g_err];
        std::memcpy(keep, g_err, sizeof keep);
        if (sc) sc->abandon();
        (void) hipStreamSynchronize(h->stream_in);
        (void) irb_sync(h);
        (void) hipMemsetAsync(h->d_ck, 0, sizeof(unsigned) * 2 * kCkMax, h->stream);   // (a piece that died half-way leaves counts behind)
        (void) irb_sync(h);
        (void) hipGetLastError();
        clear_out(0);
        std::memcpy(g_err, keep, sizeof keep);
        return code;
    };
    int rc = IRB_OK;
    // The opt-in registering form, only NOW: a registration covers whole pages, so small heap objects next to the caller's buffers become
    // "registered host memory" too for as long as it lasts -- the runtime refused the engine's own set-up copies out of such a neighbour
    // ("invalid argument": tools/fuzz_paths.py, round 5) when the registration came first.  From here to the release only kernels
    // are launched.  (The signal was cut as for the staged form, so a refused registration needs no second plan.)
    if (mode == DirectMode::kRegister && steps >= 2) {
        bool one_block_piece = false;
        for (int T : piece_T) one_block_piece |= T == 1;
        if (!one_block_piece) resolve_direct(true);
    }
    const bool device_times = g_offline_profile_events;
    if (device_times) {
        const size_t want = (size_t) 5 * std::min(n_pieces, kOfflineTimedPieces);
        while (g.ev.size() < want) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) return bail(fail(IRB_ERR_HIP, "offline events"));
            g.ev.push_back(e);
        }
    }
    // The IR's set-up is ONE launch that transforms the samples where they lie (k_ir_fft_direct: partition FFTs, zero rows, stream
    // records) + the tap-contiguous copy for the matrix cores, on the multiply's stream in front of the first multiply.  "Where they
    // lie" is the engine's pinned IR staging unless the caller's IR is device-visible itself.
    bool ir_started = false;
    float *ir_stage[2] = {h->h_ir, h->h_ir + (size_t) h->P * h->B};             // (h_ir holds 2 C P B floats: two rows of any IR this engine takes)
    const float *ir_stage_dev[2] = {h->d_hir, h->d_hir + (size_t) h->P * h->B};
    auto launch_ir_from_staging = [&]() -> int { ir_started = true; return set_ir_direct(h, ir_stage_dev, ir_ch, Lh_eff); };
    auto ir_now = [&]() -> int {                                                   // copy + launch on the calling thread
        if (ir_started) return IRB_OK;
        for (int r = 0; r < ir_ch; r++) std::memcpy(ir_stage[r], ir[r], sizeof(float) * (size_t) Lh_eff);
        return launch_ir_from_staging();
    };
    // ---- staged form: the call's shared state, built and handed to the helpers BEFORE anything is asked of the device (they wake
    // up -- or, lingering from the call before, start at once -- while this thread resets the FDL)
    const long stride = (long) h->Tmax * h->B;                          // floats between the channels of the staging
    float *const host_in = h->h_io, *const host_out = h->h_io + (size_t) audio_ch * stride;
    float *const dev_in = h->d_hio, *const dev_out = h->d_hio + (size_t) audio_ch * stride;
    int helpers = 0;
    if (!direct) {
        // default: 5 with the caller where the host has the cores (profiles/NOTES.md, round 5: configs[1] 0.311 / 0.301 / 0.295 / 0.289 / 0.292 ms
        // with 2 / 3 / 4 / 5 / 6 threads, medians of five alternating passes on one box), never more than half of them
        const int hw = (int) std::thread::hardware_concurrency();
        const int dflt_threads = std::max(1, std::min(5, hw > 0 ? hw / 2 : 3));
        const int want_threads = std::max(1, std::min(kCrewMax, env_int("IRB_OFFLINE_THREADS", dflt_threads)));
        const long call_bytes = (long) sizeof(float) * ((long) La + Lout) * audio_ch;
        // (a helper needs tens of microseconds to wake up: calls that move less than a megabyte are over before it would have copied anything)
        helpers = (want_threads > 1 && call_bytes >= dbg_int("offline_helpers_from", 1 << 20)) ? g.crew.ensure(want_threads - 1) : 0;
        const int fixed_chunk = dbg_int("offline_chunk", 0);           // (tests: every item, launch chunk and output chunk this many blocks)
        g.crew.linger_ns = 1000L * std::max(0, dbg_int("offline_linger_us", 150));
        int ola_nb = 4;
        FOR_N(h->N, ola_nb = irb::OlaGeom<NN>::NB);
        sc = std::make_shared<StagedCall>();
        long p0 = 0;
        for (int k = 0; k < n_pieces; p0 += piece_T[(size_t) k], k++) {
            const int T = piece_T[(size_t) k];
            std::unique_ptr<StagedPiece> pc(new StagedPiece);
            const size_t half = n_pieces > 1 ? (size_t) (k & 1) * (size_t) Tp * B : 0;
            pc->C = audio_ch; pc->B = B; pc->stride = stride; pc->p0 = p0; pc->La = La; pc->lim = lim; pc->T = T;
            for (int c = 0; c < audio_ch; c++) { pc->audio[c] = audio[c]; pc->out[c] = out[c]; }
            pc->host_in = host_in + half; pc->host_out = host_out + half;
            // launch chunks: ~256 KiB per channel first, then four times the one before; a remainder shorter than twice the next chunk joins it
            // (multiples of 16 blocks: the forward-FFT kernels' workgroups).  Items: ~128 KiB per channel.
            const int c0 = fixed_chunk > 0 ? (fixed_chunk + 15) / 16 * 16 : std::max(16, (65536 / B + 15) / 16 * 16);
            int it = fixed_chunk > 0 ? c0 : std::max(16, (32768 / B + 15) / 16 * 16);
            // (the kernel that waits for the items' words is the matrix-core path's; blocks of whole 128-byte lines: see there)
            const bool one_launch = T > 1 && use_mfma(h, T) && B % 32 == 0 && dbg_int("offline_in_one_launch", 1) != 0;
            if (one_launch) {
                while ((T + it - 1) / it > kCkMax) it *= 2;
                pc->in_first.push_back(0); pc->in_count.push_back(T);
                pc->pk_words = h->h_pk + (size_t) 16 * kCkMax * (size_t) (k & 1);
                pc->pk_blocks = it;
            }
            for (int first = 0, c = c0; first < T && T > 1 && !one_launch; c = fixed_chunk > 0 ? c : 4 * c) {      // (a one-block piece is a plain step on the calling thread: nothing of it is for the helpers)
                const int cnt = (T - first < 2 * c && fixed_chunk == 0) ? T - first : std::min(c, T - first);
                pc->in_first.push_back(first); pc->in_count.push_back(cnt);
                first += cnt;
            }
            const int nin = (int) pc->in_first.size();
            pc->chunk_left.reset(new std::atomic<int>[(size_t) nin]);
            for (int q = 0; q < nin; q++) {
                int n_items = 0;
                for (int f = pc->in_first[(size_t) q], e = f + pc->in_count[(size_t) q]; f < e; f += it, n_items++) {
                    pc->item_first.push_back(f); pc->item_count.push_back(std::min(it, e - f)); pc->item_chunk.push_back(q);
                }
                pc->chunk_left[(size_t) q].store(n_items, std::memory_order_relaxed);
            }
            if (k == 0 && T > 1) {                                        // the IR's samples travel with the first piece
                pc->ir_rows = ir_ch; pc->ir_len = Lh_eff;
                for (int r = 0; r < ir_ch; r++) { pc->ir[r] = ir[r]; pc->ir_stage[r] = ir_stage[r]; }
                pc->ir_done.store(0, std::memory_order_relaxed);
                pc->ir_claimed.store(0, std::memory_order_release);
            }
            // output chunks: ~128 KiB per channel (multiples of the inverse-FFT kernel's blocks per workgroup), at most kCkMax
            int ck = fixed_chunk > 0 ? fixed_chunk : std::max(ola_nb, 32768 / B);
            ck = (ck + ola_nb - 1) / ola_nb * ola_nb;
            while ((T + ck - 1) / ck > kCkMax) ck *= 2;
            pc->ck_blocks = ck; pc->nout = T > 1 ? (T + ck - 1) / ck : 0;   // (a one-block piece is copied out by the calling thread itself)
            pc->words = h->h_ck + (size_t) 16 * kCkMax * (size_t) (k & 1);
            if (++h->ck_seq == 0) ++h->ck_seq;                            // (0 is the words' initial value)
            pc->seq = h->ck_seq;
            sc->piece.push_back(std::move(pc));
        }
        sc->open.store(std::min(2, n_pieces), std::memory_order_release);   // (the first two pieces' halves of the staging are free from the start: piece 1 is packed while piece 0 is still on its way in)
        if (helpers > 0) g.crew.post(sc);
    }
    prof.setup_ms = now_ms() - t_begin - prof.register_ms;

    rc = irb_fdl_reset(h);
    if (rc != IRB_OK) return bail(rc);
    hipStream_t const s_in = h->stream_in, s_mac = h->stream, s_out = h->stream;
    if (hipEventRecord(h->ev_pipe[4], s_mac) != hipSuccess || hipStreamWaitEvent(s_in, h->ev_pipe[4], 0) != hipSuccess) return bail(fail(IRB_ERR_HIP, "pipeline events"));
    if (direct && ir_direct) {
        // the IR transformed where it lies, at once (the first forward FFTs pull audio over the link on the other stream meanwhile)
        rc = set_ir_direct(h, d_irp, ir_ch, Lh_eff);
        ir_started = true;
        if (rc != IRB_OK) return bail(rc);
    }
    const long in_stride = direct && audio_ch == 2 ? (long) (d_audio[1] - d_audio[0]) : stride, out_stride = direct && audio_ch == 2 ? (long) (d_res[1] - d_res[0]) : stride;
    auto spin_timed_out = [](const timespec &t0) {
        timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        return (t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) > kSpinLimitNs;
    };
    // the calling thread copies output chunks too whenever it would otherwise wait: until piece `upto` is out completely
    int my_oldest = 0;
    auto unpack_until = [&](int upto) -> int {
        timespec ts0;
        clock_gettime(CLOCK_MONOTONIC, &ts0);
        for (unsigned spins = 0; !sc->piece[(size_t) upto]->all_unpacked();) {
            if (sc->gave_up.load()) break;
            if (sc->unpack_some(0, my_oldest)) continue;
            __builtin_ia32_pause();                                      // the last chunks are in the helpers' hands
            if ((++spins & 4095u) == 0 && spin_timed_out(ts0)) sc->gave_up.store(1);
        }
        if (!sc->gave_up.load()) return IRB_OK;
        // a chunk's word never came: a kernel of this call did not finish.  Let the runtime say why, if it knows.
        const hipError_t e = hipStreamSynchronize(h->stream);
        return fail(e != hipSuccess ? IRB_ERR_HIP : IRB_ERR_STATE, "one-shot pipeline: an output chunk never arrived (%s)", e != hipSuccess ? hipGetErrorString(e) : "no device error reported");
    };
    int mac_launches = 1, nin_total = 0, nout_total = 0, last_ck = 0;
    const bool want_sums = dbg_int("input_checksums", 0) != 0;
    g.sums.clear(); g.sums_streams = g.sums_blocks = 0;
    if (want_sums) {
        const size_t need = (size_t) audio_ch * (size_t) steps;
        if (g.d_sums_cap < need) {
            if (g.d_sums) (void) hipFree(g.d_sums);
            g.d_sums = nullptr; g.d_sums_cap = 0;
            if (hipMalloc(reinterpret_cast<void **>(&g.d_sums), sizeof(unsigned) * need) != hipSuccess) return bail(fail(IRB_ERR_HIP, "checksum buffer"));
            g.d_sums_cap = need;
        }
        g.sums.assign(need, 0u);
    }
    std::vector<unsigned> piece_sums;
    long p0 = 0;
    for (int k = 0; k < n_pieces && rc == IRB_OK; p0 += piece_T[(size_t) k], k++) {
        const int T = piece_T[(size_t) k];
        const bool timed = device_times && k < kOfflineTimedPieces;
        hipEvent_t *tev = timed ? &g.ev[(size_t) 5 * k] : nullptr;
        StagedPiece *pc = direct ? nullptr : sc->piece[(size_t) k].get();
        const size_t half = (!direct && n_pieces > 1) ? (size_t) (k & 1) * (size_t) Tp * B : 0;
        // staged: the halves of the staging this piece uses held piece k - 2: that one must be out completely
        if (!direct && k >= 2) { rc = unpack_until(k - 2); if (rc != IRB_OK) break; }
        if (!direct && sc->open.load(std::memory_order_acquire) < k + 1) sc->open.store(k + 1, std::memory_order_release);
        irb::StepArgs a = direct ? step_args(h, d_audio[0] + p0 * B, d_res[0] + p0 * B, in_stride) : step_args(h, dev_in + half, dev_out + half, stride);
        a.out_stride = out_stride;
        a.in_len = std::max<long>(0, (long) La - p0 * B);               // blocks past the audio's end are the flush: the kernels read zeros, not memory
        a.out_len = std::max<long>(0, lim - p0 * B);                     // nothing is written past the samples this call produces
        if (T == 1) {
            // a one-block piece (a signal shorter than a block with a one-partition IR; tests that force tiny pieces): a plain step,
            // outside the pipeline -- everything before it is waited for, staged form only
            if (k >= 1) { rc = unpack_until(k - 1); if (rc != IRB_OK) break; }
            const long have = std::max<long>(0, std::min<long>(B, (long) La - p0 * B)), take = std::max<long>(0, std::min<long>(B, lim - p0 * B));
            for (int ch = 0; ch < audio_ch; ch++) {
                float *dst = pc->host_in + (size_t) ch * stride;
                if (have > 0) std::memcpy(dst, audio[ch] + p0 * B, sizeof(float) * (size_t) have);
                std::fill(dst + have, dst + B, 0.0f);
            }
            rc = ir_now();
            if (rc == IRB_OK && hipStreamSynchronize(s_in) != hipSuccess) rc = fail(IRB_ERR_HIP, "one-shot convolution: pipeline stream");
            if (rc == IRB_OK) rc = launch_step(h, a);
            if (rc != IRB_OK) break;
            h->t += 1;
            const hipError_t e = irb_sync(h);
            if (e != hipSuccess) { rc = fail(IRB_ERR_HIP, "one-shot convolution: %s", hipGetErrorString(e)); break; }
            for (int ch = 0; ch < audio_ch; ch++)
                if (take > 0) std::memcpy(out[ch] + p0 * B, pc->host_out + (size_t) ch * stride, sizeof(float) * (size_t) take);
            // the next piece's forward FFTs wait for "multiply k done": nothing is pending, the event says so
            if (hipEventRecord(h->ev_pipe[2 + (k & 1)], h->stream) != hipSuccess) { rc = fail(IRB_ERR_HIP, "hipEventRecord failed"); break; }
            continue;
        }
        if (want_sums) {
            // (diagnostic: a piece's sums are fetched before the next piece reuses the buffer -- these calls are not the fast ones)
            if (k >= 1) {
                (void) hipStreamSynchronize(s_in);
                piece_sums.resize((size_t) audio_ch * piece_T[(size_t) k - 1]);
                if (hipMemcpy(piece_sums.data(), g.d_sums, sizeof(unsigned) * piece_sums.size(), hipMemcpyDeviceToHost) == hipSuccess)
                    for (int c = 0; c < audio_ch; c++)
                        for (int j = 0; j < piece_T[(size_t) k - 1]; j++) g.sums[(size_t) c * steps + (size_t) (p0 - piece_T[(size_t) k - 1] + j)] = piece_sums[(size_t) c * piece_T[(size_t) k - 1] + j];
            }
            a.dbg_sums = g.d_sums;
        }
        TiledCall call;
        rc = tiled_begin(h, a, T, &call);
        if (rc != IRB_OK) break;
        mac_launches = call.mfma ? call.plan.n : 1;
        // ---- in, on stream_in: behind the multiply of the piece before (it reads the columns these kernels write)
        if (k >= 1 && hipStreamWaitEvent(s_in, h->ev_pipe[2 + ((k - 1) & 1)], 0) != hipSuccess) { rc = fail(IRB_ERR_HIP, "hipStreamWaitEvent failed"); break; }
        if (timed) (void) hipEventRecord(tev[0], s_in);
        if (direct) {
            rc = tiled_push(h, call, 0, T, s_in);
            if (rc == IRB_OK) rc = ir_now();                              // (an IR that is not device-visible: through the staging)
        } else {
            // launch every chunk's forward FFTs as soon as it and its predecessors are packed; pack when there is nothing to launch
            const int nin = (int) pc->in_first.size();
            nin_total += nin; nout_total += pc->nout; last_ck = pc->ck_blocks;
            timespec ts0;
            clock_gettime(CLOCK_MONOTONIC, &ts0);
            unsigned spins = 0;
            int launched = 0;
            if (pc->pk_words) {
                // one launch for the whole piece, NOW: its workgroups wait for the items' words; this thread packs with the others and is
                // through when every item has an owner and the IR's transform is on its way
                call.a.pk_flag = h->d_hpk + (size_t) 16 * kCkMax * (size_t) (k & 1); call.a.pk_blocks = pc->pk_blocks; call.a.pk_seq = pc->seq;
                rc = tiled_push(h, call, 0, T, s_in);
                launched = nin;
            }
            for (; rc == IRB_OK;) {
                while (launched < nin && rc == IRB_OK && pc->chunk_left[(size_t) launched].load(std::memory_order_acquire) == 0) {
                    rc = tiled_push(h, call, pc->in_first[(size_t) launched], pc->in_count[(size_t) launched], s_in);
                    launched++;
                }
                if (rc == IRB_OK && !ir_started && pc->ir_done.load(std::memory_order_acquire)) rc = launch_ir_from_staging();
                if (rc != IRB_OK || (launched == nin && ir_started && (!pc->pk_words || pc->all_claimed()))) break;
                if (pc->pack_one(0)) continue;
                if (pc->copy_ir(0)) continue;                              // (no audio left to claim: the IR, unless a helper has it)
                if (k >= 1 && sc->unpack_some(0, my_oldest, true)) continue;   // (... or an output chunk of the piece before that has arrived)
                __builtin_ia32_pause();                                    // a helper is still copying what comes next
                if ((++spins & 4095u) == 0 && spin_timed_out(ts0)) rc = fail(IRB_ERR_STATE, "one-shot pipeline: a packing thread never finished its chunk");
            }
        }
        if (rc != IRB_OK) break;
        if (timed) (void) hipEventRecord(tev[1], s_in);
        if (hipEventRecord(h->ev_pipe[k & 1], s_in) != hipSuccess || hipStreamWaitEvent(s_mac, h->ev_pipe[k & 1], 0) != hipSuccess) { rc = fail(IRB_ERR_HIP, "pipeline events"); break; }
        // ---- the multiply over all blocks of the piece, then out (staged: one launch that raises the chunks' words as it goes)
        if (timed) (void) hipEventRecord(tev[4], s_mac);
        rc = tiled_mac(h, call);
        if (rc != IRB_OK) break;
        if (hipEventRecord(h->ev_pipe[2 + (k & 1)], s_mac) != hipSuccess) { rc = fail(IRB_ERR_HIP, "hipEventRecord failed"); break; }
        if (timed) (void) hipEventRecord(tev[2], s_mac);
        if (!direct) {
            call.a.ck_cnt = h->d_ck + (size_t) kCkMax * (size_t) (k & 1); call.a.ck_flag = h->d_hck + (size_t) 16 * kCkMax * (size_t) (k & 1);
            call.a.ck_blocks = pc->ck_blocks; call.a.ck_seq = pc->seq;
        }
        rc = tiled_ifft(h, call, 0, T, s_out);
        if (rc != IRB_OK) break;
        if (timed) (void) hipEventRecord(tev[3], s_out);
        tiled_end(h, call);
    }
    // ---- the end of the call: direct -- one wait; staged -- the remaining output chunks, with the helpers
    const double t_wait = now_ms();
    if (direct) {
        const hipError_t e = irb_sync(h);                                // also on failure: nothing may still be reading or writing the caller's memory
        const hipError_t e2 = hipStreamSynchronize(s_in);
        if (rc == IRB_OK && (e != hipSuccess || e2 != hipSuccess)) rc = fail(IRB_ERR_HIP, "one-shot convolution: %s", hipGetErrorString(e != hipSuccess ? e : e2));
        ranges.release(); ir_ranges.release(); quiet.reset();            // (registered form: the caller's pages are the caller's again)
        prof.wait_ms = now_ms() - t_wait;
    } else if (rc == IRB_OK) {
        for (int k = std::max(0, n_pieces - 2); k < n_pieces && rc == IRB_OK; k++) rc = unpack_until(k);
        prof.stall_ms = now_ms() - t_wait;
    }
    if (rc != IRB_OK) return bail(rc);
    if (want_sums && piece_T.back() > 1) {
        const int Tl = piece_T.back();
        (void) hipStreamSynchronize(s_in);
        piece_sums.resize((size_t) audio_ch * Tl);
        if (hipMemcpy(piece_sums.data(), g.d_sums, sizeof(unsigned) * piece_sums.size(), hipMemcpyDeviceToHost) == hipSuccess)
            for (int c = 0; c < audio_ch; c++)
                for (int j = 0; j < Tl; j++) g.sums[(size_t) c * steps + (size_t) (steps - Tl + j)] = piece_sums[(size_t) c * Tl + j];
        g.sums_streams = audio_ch; g.sums_blocks = (int) steps;
    }
    if (sc) sc->finished.store(1);
    rc = check_status(h);
    if (rc != IRB_OK) return bail(rc);
    prof.total_ms = now_ms() - t_begin;
    prof.direct = direct ? 1 : 0;
    if (sc) {
        for (auto &pcp : sc->piece) {
            prof.pack_ms += pcp->pack_ms[0]; prof.unpack_ms += pcp->unpack_ms[0]; prof.wait_ms += pcp->wait_ms[0];
            for (int w = 1; w < kCrewMax; w++) { prof.pack_helper_ms += pcp->pack_ms[w]; prof.unpack_helper_ms += pcp->unpack_ms[w]; prof.wait_helper_ms += pcp->wait_ms[w]; }
        }
    }
    prof.fft_in_ms = prof.mac_ms = prof.fft_out_ms = prof.device_ms = -1.0;
    if (device_times) {
        // sums over the (first kOfflineTimedPieces) pieces; with several pieces the phases of neighbours overlap, so they add up to more than device_ms
        const int nt = std::min(n_pieces, kOfflineTimedPieces);
        float in = 0, mac = 0, outp = 0, all = 0, ms = 0;
        bool okt = hipEventSynchronize(g.ev[(size_t) 5 * (nt - 1) + 3]) == hipSuccess;
        for (int k = 0; k < nt && okt; k++) {
            if (piece_T[(size_t) k] == 1) continue;
            hipEvent_t *tev = &g.ev[(size_t) 5 * k];
            okt = hipEventElapsedTime(&ms, tev[0], tev[1]) == hipSuccess; in += ms;
            if (okt) { okt = hipEventElapsedTime(&ms, tev[4], tev[2]) == hipSuccess; mac += ms; }
            if (okt) { okt = hipEventElapsedTime(&ms, tev[2], tev[3]) == hipSuccess; outp += ms; }
        }
        if (okt && piece_T[0] != 1 && piece_T[(size_t) nt - 1] != 1) okt = hipEventElapsedTime(&all, g.ev[0], g.ev[(size_t) 5 * (nt - 1) + 3]) == hipSuccess;
        (void) hipGetLastError();
        if (okt) { prof.fft_in_ms = in; prof.mac_ms = mac; prof.fft_out_ms = outp; prof.device_ms = all; }
    }
    prof.chunks = direct ? n_pieces : nout_total; prof.in_chunks = direct ? n_pieces : nin_total; prof.blocks_per_chunk = direct ? Tp : last_ck;
    prof.pieces = n_pieces; prof.steps = (int) steps;
    prof.mac_on_matrix_cores = use_mfma(h, Tp) ? 1 : 0;
    prof.mac_launches = prof.mac_on_matrix_cores ? mac_launches : 1;
    prof.helper_thread = helpers > 0 ? 1 : 0;
    prof.threads = 1 + helpers;
    g.prof = prof; g.have_prof = true;
    return IRB_OK;
}

int irb_device_quiet_begin(int device)
{
    if (device < 0 || device >= kMaxDevices) return fail(IRB_ERR_ARG, "device %d out of range", device);
    quiet_begin(device);
    return IRB_OK;
}

int irb_device_quiet_end(int device)
{
    if (device < 0 || device >= kMaxDevices) return fail(IRB_ERR_ARG, "device %d out of range", device);
    if (!device_quiet(device) && !g_quiet_disabled.load()) return fail(IRB_ERR_STATE, "irb_device_quiet_end without a matching begin");
    quiet_end(device);
    return IRB_OK;
}

int irb_convolve_profile_events(int on)
{
    const int was = g_offline_profile_events ? 1 : 0;
    g_offline_profile_events = on != 0;
    return was;
}

// diagnostic, not part of the interface (tools/fuzz_paths.py): the calling thread's last irb_convolve_periodic call under
// IRB_DEBUG=input_checksums=1 -- per stream and block the sum of the bit patterns of the samples the forward-FFT kernel read
int irb_debug_input_checksums_(unsigned *dst, int capacity, int *streams, int *blocks)
{
    if (!dst || !streams || !blocks) return fail(IRB_ERR_ARG, "null argument");
    *streams = g_offline.sums_streams; *blocks = g_offline.sums_blocks;
    const size_t n = (size_t) g_offline.sums_streams * (size_t) g_offline.sums_blocks;
    if (n == 0 || g_offline.sums.size() < n) return fail(IRB_ERR_STATE, "no checksums: the last call did not run under IRB_DEBUG=input_checksums=1");
    if ((size_t) capacity < n) return fail(IRB_ERR_ARG, "buffer too small for %zu checksums", n);
    std::memcpy(dst, g_offline.sums.data(), sizeof(unsigned) * n);
    return IRB_OK;
}

int irb_convolve_last_profile(irb_convolve_profile *out)
{
    if (!out) return fail(IRB_ERR_ARG, "null argument");
    if (!g_offline.have_prof) return fail(IRB_ERR_STATE, "no irb_convolve_periodic call has completed on this thread");
    *out = g_offline.prof;
    return IRB_OK;
}

// ---- several GPUs from one host process ------------------------------------------------------------------
struct irb_fdl_multi {
    std::vector<irb_fdl *> shard;
    std::vector<int> first;           // first[d] = first global instance of shard d; first[D] = total
    int C = 0, Tmax = 1;
};

void irb_fdl_multi_destroy(irb_fdl_multi *m)
{
    if (!m) return;
    for (irb_fdl *h : m->shard) irb_fdl_destroy(h);
    delete m;
}

int irb_fdl_multi_create(const irb_fdl_config *cfg, const int *devices, int n_devices, irb_fdl_multi **out)
{
    if (!cfg || !devices || !out) return fail(IRB_ERR_ARG, "null argument");
    *out = nullptr;
    if (n_devices < 1 || cfg->n_instances < n_devices) return fail(IRB_ERR_ARG, "need 1 <= n_devices <= n_instances");
    irb_fdl_multi *m = new (std::nothrow) irb_fdl_multi;
    if (!m) return fail(IRB_ERR_ARG, "out of host memory");
    m->C = cfg->channels; m->Tmax = cfg->max_blocks_per_call;
    const int base = cfg->n_instances / n_devices, extra = cfg->n_instances % n_devices;
    int lo = 0;
    for (int d = 0; d < n_devices; d++) {
        irb_fdl_config c = *cfg;
        c.device = devices[d]; c.stream = nullptr; c.n_instances = base + (d < extra ? 1 : 0);
        irb_fdl *h = nullptr;
        int rc = irb_fdl_create(&c, &h);
        if (rc != IRB_OK) { irb_fdl_multi_destroy(m); return rc; }
        m->shard.push_back(h);
        m->first.push_back(lo);
        lo += c.n_instances;
    }
    m->first.push_back(lo);
    *out = m;
    return IRB_OK;
}

int irb_fdl_multi_owner(const irb_fdl_multi *m, int instance, int *shard, int *local_instance)
{
    if (!m) return fail(IRB_ERR_ARG, "null handle");
    if (instance < 0 || instance >= m->first.back()) return fail(IRB_ERR_ARG, "instance %d out of range", instance);
    int d = 0;
    while (instance >= m->first[d + 1]) d++;
    if (shard) *shard = d;
    if (local_instance) *local_instance = instance - m->first[d];
    return IRB_OK;
}

int irb_fdl_multi_set_ir(irb_fdl_multi *m, int instance, const float *const *ir, int ir_channels, int len)
{
    int d = 0, li = 0;
    int rc = irb_fdl_multi_owner(m, instance, &d, &li);
    if (rc != IRB_OK) return rc;
    return irb_fdl_set_ir(m->shard[d], li, ir, ir_channels, len);
}

int irb_fdl_multi_reset(irb_fdl_multi *m)
{
    if (!m) return fail(IRB_ERR_ARG, "null handle");
    for (irb_fdl *h : m->shard) { int rc = irb_fdl_reset(h); if (rc != IRB_OK) return rc; }
    return IRB_OK;
}

int irb_fdl_multi_process(irb_fdl_multi *m, const float *const *in, float *const *out, int n_blocks)
{
    if (!m || !in || !out) return fail(IRB_ERR_ARG, "null argument");
    if (n_blocks < 0) return fail(IRB_ERR_ARG, "n_blocks < 0");
    const int D = (int) m->shard.size();
    for (int s = 0; s < m->first.back() * m->C; s++) if (!in[s] || !out[s]) return fail(IRB_ERR_ARG, "in/out[%d] is null", s);
    for (int done = 0; done < n_blocks; done += m->Tmax) {
        const int nb = std::min(m->Tmax, n_blocks - done);
        for (int d = 0; d < D; d++) {                  // enqueue on every device first ...
            int rc = process_chunk_begin(m->shard[d], in + (size_t) m->first[d] * m->C, done, nb);
            if (rc != IRB_OK) return rc;
        }
        for (int d = 0; d < D; d++) {                  // ... then collect
            int rc = process_chunk_end(m->shard[d], out + (size_t) m->first[d] * m->C, done, nb);
            if (rc != IRB_OK) return rc;
        }
    }
    return IRB_OK;
}

// ---- ir::IRtoRealFFTRaw (fp/ir.cpp:106-147): an IR's partitions as packed spectra rows, the layout the reference's plug-in keeps
// its IR in.  The rows are exactly what k_ir_fft_direct writes (bin 0 = (DC, Nyquist)): one launch transforms the IR's samples
// where they lie and writes the rows where the host reads them.  "Where they lie" / "where the host reads them" are the caller's own
// buffers when those are HIP host memory (or with IRB_OFFLINE_DIRECT=1: registered for the call), otherwise -- the default for
// pageable buffers, host_ranges.hpp -- a pinned staging pair kept per thread: one host copy in, one out.
// (Until round 4 every call built and destroyed an engine: 4.9 ms for a 2-second IR; now ~0.1 ms.)
namespace {
struct RawPlan {
    int device = -1, N = 0;
    size_t rows = 0, in_floats = 0;
    float2 *d_H = nullptr, *d_tw = nullptr;
    float *h_in = nullptr, *h_out = nullptr;           // pinned, mapped: [in_floats], [rows * N]
    float *d_hin = nullptr, *d_hout = nullptr;         // their device addresses
    hipStream_t stream = nullptr;
    void release()
    {
        if (device < 0) return;
        DeviceQuiet quiet(device);                     // frees wait for every stream of the device: resident servers step aside
        (void) hipSetDevice(device);
        if (stream) { (void) hipStreamSynchronize(stream); (void) hipStreamDestroy(stream); }
        (void) hipFree(d_H); (void) hipFree(d_tw);
        if (h_in) (void) hipHostFree(h_in);
        if (h_out) (void) hipHostFree(h_out);
        d_H = d_tw = nullptr; h_in = h_out = d_hin = d_hout = nullptr; stream = nullptr; device = -1; N = 0; rows = 0; in_floats = 0;
    }
    ~RawPlan() { release(); }
};
thread_local RawPlan g_raw;
}

int irb_ir_to_real_fft_raw(int device, const float *ir, int len, int ir_part_size, float *out, int out_len, int *n_rows)
{
    if (!ir || !out || len < 1) return fail(IRB_ERR_ARG, "bad argument");
    if (ir_part_size < 16 || ir_part_size > 2048 || (ir_part_size & (ir_part_size - 1)))
        return fail(IRB_ERR_ARG, "irPartSize must be a power of two in [16, 2048]");
    const int bufcount = len / ir_part_size + 1;                      // :107
    const int N = ir_part_size * 2, M = ir_part_size;                 // :108
    if ((long) out_len < (long) bufcount * N) return fail(IRB_ERR_ARG, "out too small");
    const int np = (len + ir_part_size - 1) / ir_part_size;
    HIP_TRY(hipSetDevice(device));
    RawPlan &g = g_raw;
    if (g.device != device || g.N != N || g.rows < (size_t) np || g.in_floats < (size_t) len) {
        g.release();
        hipError_t e = hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&g.d_tw), sizeof(float2) * (size_t) N);
        const size_t rows = ((size_t) np + 63) / 64 * 64, in_floats = rows * (size_t) ir_part_size;
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&g.d_H), sizeof(float2) * rows * M);
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&g.h_in), sizeof(float) * in_floats, hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&g.h_out), sizeof(float2) * rows * M, hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void **>(&g.d_hin), g.h_in, 0);
        if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void **>(&g.d_hout), g.h_out, 0);
        std::vector<float2> tw((size_t) N);
        for (int j = 0; j < N; j++) {                                 // twiddles: W_N^j in double, rounded to float (as irb_fdl_create)
            const double ph = -2.0 * 3.14159265358979323846 * (double) j / (double) N;
            tw[(size_t) j] = make_float2((float) std::cos(ph), (float) std::sin(ph));
        }
        if (e == hipSuccess) e = hipMemcpy(g.d_tw, tw.data(), sizeof(float2) * (size_t) N, hipMemcpyHostToDevice);
        g.device = device;                                            // (so that release() frees whatever exists)
        if (e != hipSuccess) { g.release(); return fail(IRB_ERR_HIP, "IRtoRealFFTRaw plan: %s", hipGetErrorString(e)); }
        g.N = N; g.rows = rows; g.in_floats = in_floats;
    }
    auto transform = [&](const float *d_src, float2 *d_dst) -> int {              // rows [0, np) of the IR at d_src -> d_dst
        irb::IrDirectArgs a;
        std::memset(&a, 0, sizeof a);
        a.src[0] = a.src[1] = d_src;
        a.H = d_dst; a.tw = g.d_tw; a.stream_info = nullptr; a.ninfo = 0;
        a.average = 0; a.nrows = 1; a.len = len; a.B = ir_part_size; a.P = (int) g.rows; a.first_set = 0; a.np = np; a.np_pad = np;
        FOR_N(N, irb::k_ir_fft_direct<NN><<<dim3((np + irb::FftGeom<NN>::NF - 1) / irb::FftGeom<NN>::NF), dim3(irb::kWave * irb::FftGeom<NN>::NF), 0, g.stream>>>(a));
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? IRB_OK : fail(IRB_ERR_HIP, "IRtoRealFFTRaw: %s", hipGetErrorString(e));
    };
    auto finish = [&]() {
        if (bufcount > np) std::memset(out + (size_t) np * N, 0, sizeof(float) * (size_t) (bufcount - np) * N);   // an untouched last buffer stays zero
        if (n_rows) *n_rows = bufcount;
        return IRB_OK;
    };
    const DirectMode mode = direct_mode_from_env();
    if (mode != DirectMode::kNever) {
        std::unique_ptr<DeviceQuiet> quiet;
        if (mode == DirectMode::kRegister) quiet.reset(new DeviceQuiet(device));   // registration is a device-wide operation for the runtime
        HostRanges ranges;
        const float *irp[1] = {ir};
        float *outp[1] = {out};
        ranges.add(irp, 1, (size_t) len);
        ranges.add(outp, 1, (size_t) np * N);
        const float *d_ir = nullptr;
        float *d_out = nullptr;
        if (ranges.commit(mode == DirectMode::kRegister) && (d_ir = ranges.dev(ir)) != nullptr && (d_out = ranges.dev(out)) != nullptr) {
            const int rct = transform(d_ir, reinterpret_cast<float2 *>(d_out));
            const hipError_t es = hipStreamSynchronize(g.stream);     // also on failure: nothing may still touch the caller's memory
            ranges.release();
            if (rct != IRB_OK) return rct;
            if (es != hipSuccess) return fail(IRB_ERR_HIP, "IRtoRealFFTRaw: %s", hipGetErrorString(es));
            return finish();
        }
        ranges.release();
    }
    std::memcpy(g.h_in, ir, sizeof(float) * (size_t) len);
    const int rct = transform(g.d_hin, reinterpret_cast<float2 *>(g.d_hout));
    const hipError_t es = hipStreamSynchronize(g.stream);
    if (rct != IRB_OK) return rct;
    if (es != hipSuccess) return fail(IRB_ERR_HIP, "IRtoRealFFTRaw: %s", hipGetErrorString(es));
    std::memcpy(out, g.h_out, sizeof(float) * (size_t) np * N);
    return finish();
}

void irb_fdl_release_cached_engine_(void)
{
    g_offline.release();
    g_raw.release();
}

}  // extern "C"
