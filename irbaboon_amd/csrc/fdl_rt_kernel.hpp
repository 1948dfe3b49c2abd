// fdl_rt_kernel.hpp -- the real-time callback of the plug-in (reference Source/PluginProcessor.cpp:452-518) as ONE launch.
// Included by fdl_kernels.hpp.
//
// k_rt_pair<N, NCH>: one workgroup per instance -- stereo, both channels multiplying with the same IR spectra (the plug-in feeds
// IR channel 0 to both, :485; NCH = 2, "wave 2" below), or mono (NCH = 1: wave 1 takes the IR partition).  For each of the call's
// T blocks, in order:
//   * wave 2 re-derives ONE IR partition from the time-domain IR (the rolling refresh, :455-461): block j of the call
//     refreshes partition (rf_part + j) mod P -- forward FFT next to the two audio FFTs of waves 0 and 1, spectrum into
//     the H row for later launches and into LDS for this block;
//   * all four waves multiply-accumulate the older partitions straight from HBM/L2, skipping the row being refreshed;
//   * partition 0 pairs with the spectra the audio FFTs just produced, the refreshed partition with its LDS copy;
//   * inverse FFTs, overlap-add (:504-510), outputs to (host-mapped) memory.
// Blocks of one call are serialised by workgroup barriers (one workgroup owns the instance, so there is no hand-off
// between workgroups), and the last thing the kernel does is a system-scope store of `done_seq` to a host-mapped word:
// the audio thread spins on that word instead of entering the runtime's wait.  One launch and one wait per callback
// whatever the host block size (round 2: two launches and one runtime wait per 256-sample block).
// k_rt_server<N>: the same blocks served by a workgroup that stays resident between callbacks (no launch at all), below.
#pragma once

namespace irb {

template <int N, int NCH>
struct LdsRt {
    static constexpr int M = N / 2;
    float2 tw[TwLds<N>::SIZE];
    float2 buf[NCH + 1][M];                            // FFT work buffers: the audio channels, then the IR partition
    float2 hnew[M];                                    // packed spectrum of the partition refreshed with this block
    float2 scratch[NCH][MacGeom<N>::STRIPES * M];      // new audio spectra, then the stripe partials
};

// one sample the host may have stored through the BAR while a resident kernel was already running: system scope, past
// every cache of the device (a plain load could be served from a line the kernel fetched for the previous request)
template <bool SYS>
__device__ __forceinline__ float ld_sample(const float *p)
{
    if (SYS) return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    return *p;
}

template <int N, bool SYS>
__device__ __forceinline__ void rt_load_x(BlockIo<N> &io, const float *__restrict__ xin, int B, int lane)
{
#pragma unroll
    for (int i = 0; i < BlockIo<N>::XL; i++) io.x[i] = ld_sample<SYS>(xin + min(lane + i * kWave, B - 1));
}

// T blocks of one instance (workgroup blockIdx.x) with NCH = 1 or 2 audio channels that multiply with the same IR spectra, block j
// refreshing partition (rf_part + j) mod np first; waves 0 .. NCH-1 own the audio channels' transforms, wave NCH the IR partition's;
// the twiddle table is in lds.tw (staged by the caller, covered by the first barrier here)
template <int N, bool SYS, int NCH>
__device__ __forceinline__ void rt_blocks(const StepArgs &a, LdsRt<N, NCH> &lds, int T, int head0, int rf_part)
{
    using G = MacGeom<N>;
    constexpr int M = N / 2, U = 4;
    const int s0 = NCH * blockIdx.x;
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const int2 si = a.stream_info[s0];
    const int set = si.x, np = si.y;
    float2 *Hs = const_cast<float2 *>(a.H) + (size_t) set * a.P * M;
    const float4 *H4 = reinterpret_cast<const float4 *>(Hs);
    float2 *Xs[NCH];
    const float4 *X4[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) { Xs[ch] = a.X + (size_t) (s0 + ch) * a.R * M; X4[ch] = reinterpret_cast<const float4 *>(Xs[ch]); }
    const int col = threadIdx.x % G::LW, stripe = threadIdx.x / G::LW;
    const float4 *hnew4 = reinterpret_cast<const float4 *>(lds.hnew);

    // samples of the first block + the saved overlaps (audio waves), of the first IR partition (wave NCH).  Later blocks' samples
    // are asked for one block ahead (below); the overlap stays in registers.
    BlockIo<N> io;
    if (wave < NCH) {
        rt_load_x<N, SYS>(io, a.in + (size_t) (s0 + wave) * a.in_stride, a.B, lane);
        const float *ov = a.ovl + (size_t) (s0 + wave) * a.B;
#pragma unroll
        for (int i = 0; i < BlockIo<N>::XL; i++) io.ov[i] = ov[min(lane + i * kWave, a.B - 1)];
    } else if (wave == NCH) {
        rt_load_x<N, SYS>(io, a.rf_src + (size_t) (np > 0 ? rf_part % np : 0) * a.B, a.B, lane);
    }

    for (int j = 0; j < T; j++) {
        const int head = (head0 + j) % a.R;
        const int rp = np > 0 ? (rf_part + j) % np : 0;
        // rows that meet operands still in flight: H row 0 (partition 0 pairs with this block's spectra) and the FDL rows
        // the refreshed partition pairs with
        int slot_r = head - rp;
        if (slot_r < 0) slot_r += a.R;
        float4 h0[G::CPT], xr[NCH][G::CPT];
#pragma unroll
        for (int c = 0; c < G::CPT; c++) {
            h0[c] = H4[col + c * G::LW];
#pragma unroll
            for (int ch = 0; ch < NCH; ch++) xr[ch][c] = X4[ch][(size_t) slot_r * G::L + col + c * G::LW];
        }

        // older partitions 1 .. np-1 except the one being refreshed (its H row is rewritten by wave 2 in this block)
        float4 acc[NCH][G::CPT];
        float2 dc[NCH];
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            dc[ch] = make_float2(0.f, 0.f);
#pragma unroll
            for (int c = 0; c < G::CPT; c++) acc[ch][c] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int k = 1 + stripe; k < np; k += U * G::STRIPES) {
            float4 h[U][G::CPT], x[NCH][U][G::CPT];
#pragma unroll
            for (int u = 0; u < U; u++) {              // clamped, never predicated: all loads of the group in flight together
                const int kk = min(k + u * G::STRIPES, np - 1);
                int slot = head - kk;
                if (slot < 0) slot += a.R;
#pragma unroll
                for (int c = 0; c < G::CPT; c++) {
                    h[u][c] = H4[(size_t) kk * G::L + col + c * G::LW];
#pragma unroll
                    for (int ch = 0; ch < NCH; ch++) x[ch][u][c] = X4[ch][(size_t) slot * G::L + col + c * G::LW];
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int kk = k + u * G::STRIPES;
                const bool live = kk < np && kk != rp;
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int c = 0; c < G::CPT; c++)
#pragma unroll
                    for (int ch = 0; ch < NCH; ch++) cmac_pair(acc[ch][c], dc[ch], live ? x[ch][u][c] : z, h[u][c], c == 0);
            }
        }

        if (wave <= NCH) io.block_to_lds(lds.buf[wave], a.B, lane);
        if (j + 1 < T) {                               // the next block's samples land while this block is transformed
            if (wave < NCH) rt_load_x<N, SYS>(io, a.in + (size_t) (s0 + wave) * a.in_stride + (size_t) (j + 1) * a.B, a.B, lane);
            else if (wave == NCH) rt_load_x<N, SYS>(io, a.rf_src + (size_t) (np > 0 ? (rf_part + j + 1) % np : 0) * a.B, a.B, lane);
        }
        // barriers inside a block order LDS traffic only (lds_barrier: no vmcnt drain), so those loads stay in flight
        lds_barrier();
        if (wave < NCH) {                              // new audio blocks: FFT, spectrum to LDS (partition 0) and into the ring
            wave_fft<M, false>(lds.buf[wave], lds.tw, lane);
            float2 *ring = a.X + ((size_t) (s0 + wave) * a.R + head) * M;
            float2 *sc = lds.scratch[wave];
            wave_split_forward<M>(lds.buf[wave], lds.tw, lane, [&](int b, float2 v) { sc[b] = v; ring[b] = v; });
        } else if (wave == NCH && np > 0) {            // the refreshed IR partition: spectrum to LDS and into its H row
            wave_fft<M, false>(lds.buf[NCH], lds.tw, lane);
            float2 *row = Hs + (size_t) rp * M;
            wave_split_forward<M>(lds.buf[NCH], lds.tw, lane, [&](int b, float2 v) { lds.hnew[b] = v; row[b] = v; });
        }
        lds_barrier();
        if (np > 0 && threadIdx.x < G::LW) {
#pragma unroll
            for (int c = 0; c < G::CPT; c++) {
                const float4 hn = hnew4[col + c * G::LW];
                const float4 hp0 = rp == 0 ? hn : h0[c];
#pragma unroll
                for (int ch = 0; ch < NCH; ch++) {
                    cmac_pair(acc[ch][c], dc[ch], reinterpret_cast<const float4 *>(lds.scratch[ch])[col + c * G::LW], hp0, c == 0);
                    if (rp != 0) cmac_pair(acc[ch][c], dc[ch], xr[ch][c], hn, c == 0);
                }
            }
        }
        lds_barrier();
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) reduce_stripes<N, true>(acc[ch], dc[ch], lds.scratch[ch], lds.buf[ch]);
        if (wave < NCH) {
            wave_split_inverse<M>(lds.buf[wave], lds.tw, lane);
            wave_fft<M, true>(lds.buf[wave], lds.tw, lane);
            overlap_add_keep<N>(io, lds.buf[wave], a.B, a.ovl + (size_t) (s0 + wave) * a.B,
                                a.out + (size_t) (s0 + wave) * a.out_stride + (size_t) j * a.B, lane);
        }
        // end of the block: its ring slot, H row and overlap are complete (every wave has drained its stores) before
        // the next block of this call reads them; one workgroup, one CU: a workgroup barrier is the whole hand-off
        __syncthreads();
    }
}

template <int N, int NCH = 2>
__global__ __launch_bounds__(kThreads) void k_rt_pair(StepArgs a)
{
    __shared__ LdsRt<N, NCH> lds;
    stage_twiddles<N>(a.tw, lds.tw);                   // once for every block of the call (first barrier of rt_blocks covers it)
    rt_blocks<N, false, NCH>(a, lds, a.T, a.head, a.rf_part);
    if (a.done != nullptr && threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: the output blocks reach host memory before the word
        __hip_atomic_store(a.done, a.done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---------------------------------------------------------------------------------------------
// k_rt_server: the same callback WITHOUT a launch per callback (opt-in, IRB_FDL_RESIDENT).  A launch + dispatch + completion
// signal cost ~8 us on this system for an empty kernel (tools/bar_probe.hip) -- two thirds of the one-launch callback.  Here
// one workgroup stays resident and serves requests the audio thread posts into a mailbox in BAR-mapped device memory:
// inputs and IR partitions first, then {n_blocks, first_partition, head}, a store fence, then the request number.  Thread 0
// polls the request number with system-scope loads (s_sleep between polls); the workgroup runs rt_blocks and answers with
// the same completion word as k_rt_pair.  Every wave reaches an exit: a quit word (the host stops the server before any
// other call touches the engine), a yield word (another thread of the process is about to make a device-synchronising runtime
// call -- hipFree, hipDeviceSynchronize -- which would otherwise wait for this kernel: irb_device_quiet_begin), an idle timeout
// (no request for idle_ticks of the 100 MHz clock: the transport stopped), and a lifetime cap (250 ms: the longest a FOREIGN
// device-wide wait can be held up).  The kernel's last act is a system-scope store of its generation to `exited`: a host that posted a
// request while the server was on its way out sees that word with the request unanswered and launches the next
// generation, whose first request it is -- a request is therefore served exactly once.
// ---------------------------------------------------------------------------------------------
struct alignas(64) RtMailbox {
    unsigned seq;                                      // request number, written LAST by the host
    int      n_blocks, first_partition, head;
    union {
        struct {
            unsigned quit;                             // the handle's own thread: leave now (it then waits for the kernel)
            unsigned yield;                            // any thread of the process (irb_device_quiet_begin): leave after the request in hand
        };
        unsigned long long leave;                      // both, as the server polls them: one 8-byte load
    };
    unsigned pad[10];
};
static_assert(sizeof(RtMailbox) == 64 && offsetof(RtMailbox, leave) == 16, "one cache line; quit and yield share an aligned 8-byte word");

template <int N, int NCH = 2>
__global__ __launch_bounds__(kThreads) void k_rt_server(StepArgs a, const RtMailbox *mb, unsigned *exited, unsigned gen,
                                                         unsigned first_seq, unsigned idle_ticks, unsigned life_ticks)
{
    __shared__ LdsRt<N, NCH> lds;
    __shared__ int l_cmd[4];
    stage_twiddles<N>(a.tw, lds.tw);
    const unsigned long long born = __builtin_amdgcn_s_memrealtime();
    for (unsigned next = first_seq;; next = next + 1 == 0 ? 1 : next + 1) {
        if (threadIdx.x == 0) {
            int cmd = 0;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                if (__hip_atomic_load(&mb->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == next) { cmd = 1; break; }
                if (__hip_atomic_load(&mb->leave, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;   // quit | yield
                const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                if (now - t0 > idle_ticks || now - born > life_ticks) break;
                __builtin_amdgcn_s_sleep(4);
            }
            if (cmd) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
                l_cmd[1] = __hip_atomic_load(&mb->n_blocks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                l_cmd[2] = __hip_atomic_load(&mb->first_partition, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                l_cmd[3] = __hip_atomic_load(&mb->head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            l_cmd[0] = cmd;
        }
        __syncthreads();
        const int cmd = l_cmd[0], T = l_cmd[1], part = l_cmd[2], head = l_cmd[3];
        __syncthreads();                               // l_cmd may be rewritten by thread 0 as soon as it polls again
        if (!cmd) break;
        rt_blocks<N, true, NCH>(a, lds, T, head, part);   // ends with a barrier behind every wave's drained stores
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            __hip_atomic_store(a.done, next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        __hip_atomic_store(exited, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace irb
