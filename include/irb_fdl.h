/*
 * irb_fdl.h -- C ABI of the MI355X partitioned-convolution engine (libirbfdl.so).
 *
 * This is the drop-in boundary for the ONE hot path of Flixor/IRBaboon that this
 * repository accelerates: uniformly partitioned overlap-add FFT convolution with
 * a frequency-domain delay line (FDL).  The reference has no FFI for it -- the
 * path is a C++ free function and a block of code inlined in the plug-in's audio
 * callback -- so each entry point below names the reference code it replaces
 * (paths relative to the reference checkout).  Plain pointers and sizes only; no
 * C++/torch/JUCE types cross this boundary.
 *
 * Vocabulary (as in the reference): "block" = processBlockSize samples (B),
 * "partition" = one B-sample slice of the impulse response (IR), N = FFT length
 * (smallest power of two >= 2B-1, fp/convolution.cpp:45-48), P = ceil(Lh/B)
 * partitions (fp/convolution.cpp:52), "instance" = one independent plug-in /
 * IR x stream pair, "stream" = one (instance, channel) convolution.
 *
 * Threading: one caller thread per handle ("the audio thread").  After create
 * no entry point allocates device memory.  The *_device entry points enqueue
 * work on the handle's HIP stream and return without waiting.
 *
 * Errors: every int-returning function gives IRB_OK (0) or a negative code, and
 * irb_last_error() (thread local) describes the last failure.  The reference's
 * own convention -- DBG() and a cleared output buffer
 * (fp/convolution.cpp:39-42) -- is kept by irb_convolve_periodic, which clears
 * `out` before it reports IRB_ERR_LAYOUT.
 */
#ifndef IRB_FDL_H
#define IRB_FDL_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IRB_OK            0
#define IRB_ERR_ARG      (-1)   /* bad argument / size                              */
#define IRB_ERR_HIP      (-2)   /* HIP runtime failure (no GPU, OOM, launch error)  */
#define IRB_ERR_LAYOUT   (-3)   /* channel layout outside mono/stereo x mono/stereo */
#define IRB_ERR_STATE    (-4)   /* call not legal in this state                     */

/* irb_fdl_config.flags */
#define IRB_FDL_FORCE_FUSED   0x1u  /* always one workgroup per stream               */
/*      0x2u, 0x4u: the partition-split kernels of rounds 1-4 (IRB_FDL_FORCE_SPLIT, IRB_FDL_SPLIT_3LAUNCH); removed in round 5 -- no
 *      heuristic selected them; the values stay unused */
#define IRB_FDL_NO_PAIRING    0x8u  /* never share IR rows between the two channels of an instance */
#define IRB_FDL_FORCE_COLS    0x10u /* always the column-sliced single-launch step (k_step_cols)   */
#define IRB_FDL_HOST_IO       0x20u /* irb_fdl_process / set_ir_partition: kernels read and write mapped pinned host memory directly
                                       (zero copy; for small real-time engines: one launch + one wait per block, no copy round trips) */
#define IRB_FDL_STEPWISE      0x40u /* a call of n <= max_blocks_per_call blocks runs n one-block steps, never the time-tiled
                                       kernels; the FDL ring then holds exactly P slots (the real-time adapter's engine)   */
#define IRB_FDL_NO_RT_KERNEL  0x80u /* irb_fdl_process_refresh: always the refresh launch + step launch per block (A/B, tests) */
#define IRB_FDL_RESIDENT      0x100u /* irb_fdl_process_refresh (single stereo or mono instance, IRB_FDL_HOST_IO, large-BAR system): requests are served by
                                        a workgroup that stays resident on the GPU between callbacks (no kernel launch per callback); it exits
                                        by itself after 20 ms without a request (the next callback starts a new one), after 250 ms at the
                                        latest, whenever any other entry point of the handle is called, and when any thread opens a quiet
                                        section on the device (irb_device_quiet_begin).
                                        WHAT A LIVE SERVER MEANS FOR THE REST OF THE PROCESS: hipFree, hipHostFree, hipDeviceSynchronize and
                                        anything else that waits for every stream of the device wait for the server kernel too.  Every such
                                        call inside this library is bracketed by a quiet section, so prepareToPlay / releaseResources of
                                        another instance or a one-shot irb_convolve_periodic are never held up by it (tests:
                                        test_resident_server_does_not_stall_other_handles).  The host's OWN device-wide waits should be
                                        bracketed with irb_device_quiet_begin / _end; one that is not waits at most the 250 ms lifetime cap. */

typedef struct irb_fdl irb_fdl;          /* batched FDL convolver, device resident  */
typedef struct irb_rt irb_rt;            /* processBlock-shaped adapter on top      */

typedef struct irb_fdl_config {
    int   device;               /* HIP device ordinal                                 */
    int   n_instances;          /* independent instances batched on this device       */
    int   channels;             /* audio channels per instance (reference: 1 or 2)    */
    int   block;                /* B, processBlockSize (Source/PluginProcessor.h:197) */
    int   max_ir_len;           /* longest IR in samples; fixes P                     */
    int   max_blocks_per_call;  /* most blocks one process call may carry (>= 1)      */
    void *stream;               /* hipStream_t to enqueue on; NULL: own stream        */
    unsigned flags;
} irb_fdl_config;

typedef struct irb_fdl_info {
    int    fft_len;             /* N                                                  */
    int    partitions;          /* P                                                  */
    int    ring_slots;          /* FDL ring length (>= P)                             */
    int    streams;             /* n_instances * channels                             */
    int    splits;              /* workgroups per stream in the MAC kernel (1=fused)  */
    size_t device_bytes;        /* HBM held by this handle                            */
    double algorithmic_bytes_per_block_step;   /* per stream: 8*P*N + 16*B */
} irb_fdl_info;

/* Replaces: plug-in ctor + prepareToPlay ring/FFT set-up
 * (Source/PluginProcessor.cpp:57-82, 203-233) and the buffer allocation of
 * convolvePeriodic (fp/convolution.cpp:45-87).  FDL and overlap start cleared. */
int  irb_fdl_create(const irb_fdl_config *cfg, irb_fdl **out);

/* Replaces: releaseResources / dtor (Source/PluginProcessor.cpp:240-248, 89-96). */
void irb_fdl_destroy(irb_fdl *h);

/* Geometry the handle derived (N, P, ring length ...).  No reference counterpart: the reference keeps these as
 * locals (fp/convolution.cpp:45-52) / members (Source/PluginProcessor.h:194-198).    */
int  irb_fdl_get_info(const irb_fdl *h, irb_fdl_info *info);

/* Replaces: IR partitioning + forward FFT (fp/convolution.cpp:106-125) including
 * the channel-layout rules (fp/convolution.cpp:28-42, 94-101, 120-121, 176-181):
 *   ir_channels 1            -> every audio channel of `instance` uses ir[0]
 *   ir_channels 2, channels 2 -> audio channel c uses ir[c]
 *   ir_channels 2, channels 1 -> (ir[0]+ir[1])/2 (tools::sumToMono, fp/tools.cpp:14-30)
 * ir: host pointers, planar [ir_channels][len], len <= max_ir_len.  Takes effect
 * for all partitions at once (the whole-IR behaviour of convolvePeriodic).  The
 * samples are copied before the call returns; the transform is enqueued on the
 * handle's stream (ordered before every later process call), so loading the IRs of
 * thousands of instances does not wait for the GPU once per instance.             */
int  irb_fdl_set_ir(irb_fdl *h, int instance, const float *const *ir, int ir_channels, int len);

/* Replaces: the rolling one-partition-per-block refresh of the real-time path
 * (Source/PluginProcessor.cpp:455-461): re-derives partition `partition` of
 * every stream of `instance` from `block` host samples (IR channel 0 feeds both
 * audio channels there, :485).                                                   */
int  irb_fdl_set_ir_partition(irb_fdl *h, int instance, int partition, const float *samples);

/* IR partition spectra as an interchange format: n_partitions rows of N floats per IR channel, each row
 * the partition's N-point real FFT with im(0) replaced by re(N/2) -- the packing the reference itself
 * uses when it exports partitioned IR spectra (ir::IRtoRealFFTRaw, fp/ir.cpp:106-147, :123-124).  This is
 * exactly the engine's HBM layout, so set/get are plain copies.  `get` reads the set that audio channel
 * `channel` of `instance` multiplies with; `set` follows the channel-layout rules of irb_fdl_set_ir
 * (a stereo IR for mono audio must be summed to mono by the caller before its transform). */
int  irb_fdl_get_ir_spectra(irb_fdl *h, int instance, int channel, float *spectra, int n_partitions);
int  irb_fdl_set_ir_spectra(irb_fdl *h, int instance, const float *const *spectra, int ir_channels, int n_partitions);

/* Zero the FDL and the overlap (what a fresh convolvePeriodic call starts from,
 * fp/convolution.cpp:59,67,81,87).  IR spectra are kept.                         */
int  irb_fdl_reset(irb_fdl *h);

/* Replaces: one or more FDL steps -- forward FFT of the new block into the ring
 * (fp/convolution.cpp:128-145; RT Source/PluginProcessor.cpp:428-444), the
 * multiply-accumulate over partitions (fp/convolution.cpp:171-202; RT :480-501),
 * inverse FFT (:206; RT :504) and overlap-add (:209-213; RT :506-510).
 * in/out: host pointers, planar, one per stream (stream = instance*channels+ch),
 * n_blocks*block samples each; in[s]==out[s] is allowed.  Blocks until done.
 * Calls of several blocks multiply in tiles along the time axis (>= 256 blocks: on the
 * matrix cores) whose zero padding meets neighbouring blocks: for finite samples the
 * result is the reference's; an Inf or NaN sample can reach output blocks a tile's
 * width (<= 255 blocks) outside the np blocks it reaches in the reference, and where
 * the reference's four-product complex multiply gives Inf the matrix-core path (three
 * real products: Xr Hr, Xi Hi, (Xr+Xi)(Hr+Hi)) can give NaN.                          */
int  irb_fdl_process(irb_fdl *h, const float *const *in, float *const *out, int n_blocks);

/* Replaces: the body of the IRCAP_IDLE branch for the n_blocks complete blocks of one host callback
 * (Source/PluginProcessor.cpp:452-518), in ONE wait: before block j is stepped, partition (first_partition + j) mod P
 * of `instance` is re-derived from ir + that partition's offset (the rolling refresh, :455-461; IR channel 0 feeds every
 * audio channel, :485), then the block goes through the FDL step (:463-510).  `ir`: host pointer to P*block samples.
 * in/out as irb_fdl_process, n_blocks <= max_blocks_per_call; the handle must be IRB_FDL_STEPWISE (or one block per
 * call).  A single stereo or mono instance created with IRB_FDL_HOST_IO and a short IR ring (the plug-in adapter's engine) runs
 * the whole call as one kernel launch and waits on a host-mapped completion word; other shapes enqueue a refresh launch
 * and a step launch per block and wait once.  Blocks until done.                                                    */
int  irb_fdl_process_refresh(irb_fdl *h, int instance, const float *const *in, float *const *out, int n_blocks,
                             const float *ir, int first_partition);

/* What this handle has asked of the HIP runtime since create: kernel launches, blocking waits, copy commands.
 * Measurement support (bench.py reports launches and waits per real-time callback); no reference counterpart.   */
int  irb_fdl_get_counters(const irb_fdl *h, unsigned long long *launches, unsigned long long *waits,
                          unsigned long long *copies);

/* Same steps (fp/convolution.cpp:128-213; RT Source/PluginProcessor.cpp:428-510) with one packed host buffer per
 * direction: stream s, block j at in + s*stride + j*block (floats).
 * No per-stream packing on the host; with buffers from irb_host_alloc (pinned) the transfers run at PCIe
 * speed.  in == out is allowed.  Blocks until done.                                  */
int  irb_fdl_process_packed(irb_fdl *h, const float *in, float *out, long stride, int n_blocks);

/* Page-locked host memory for the packed API (hipHostMalloc / hipHostFree); no reference counterpart. */
void *irb_host_alloc(size_t bytes);
void  irb_host_free(void *p);

/* Same steps (fp/convolution.cpp:128-213) with device-resident audio: stream s, block j lives at
 * d_in + s*stride + j*block (floats); likewise d_out.  Asynchronous.            */
int  irb_fdl_process_device(irb_fdl *h, const float *d_in, float *d_out, long stride, int n_blocks);

/* Wait for everything enqueued on the handle's stream (no reference counterpart: the reference is synchronous).
 * Returns IRB_ERR_STATE if a step kernel's bounded in-kernel wait gave up since the last check (multi-workgroup steps
 * of few streams hand partial spectra over inside one launch; a wait is abandoned after 50 ms instead of hanging the
 * GPU; the affected stream then writes neither output nor overlap for that block).  The blocking process calls report
 * the same condition.  After IRB_ERR_STATE call irb_fdl_reset before relying on the handle's output again.           */
int  irb_fdl_synchronize(irb_fdl *h);

/* Stream-ordered timing on the handle's stream (hipEvent): `mark` records event
 * slot 0 or 1; `elapsed_ms` waits for both and returns slot1 - slot0.  Measurement
 * support for bench.py; no reference counterpart (the reference has no timing code). */
int  irb_fdl_mark(irb_fdl *h, int slot);
int  irb_fdl_elapsed_ms(irb_fdl *h, float *ms);

/* Device scratch owned by the handle (no reference counterpart), for callers that want device-resident
 * audio without another allocator: [streams][max_blocks_per_call*block] floats
 * each.  Pointers stay valid until destroy.                                      */
int  irb_fdl_device_io(irb_fdl *h, float **d_in, float **d_out, long *stride);
int  irb_fdl_upload(irb_fdl *h, float *d_dst, const float *src, size_t n_floats);
int  irb_fdl_download(irb_fdl *h, float *dst, const float *d_src, size_t n_floats);

/* Replaces: AudioBuffer<float> fp::convolution::convolvePeriodic(AudioBuffer<float>&
 * audio, AudioBuffer<float>& ir, int processBlockSize) (fp/convolution.hpp:31,
 * fp/convolution.cpp:14-242), whole signal in, whole signal out.
 * audio: planar [audio_ch][La]; ir: planar [ir_ch][Lh]; out: planar
 * [audio_ch][La+Lh-1].  keep_reference_tail != 0 reproduces the reference's
 * dropped tail (only (floor(La/B)+P)*B samples are written, the rest stay 0,
 * fp/convolution.cpp:233-238); 0 writes the complete linear convolution.
 * `out` must not overlap `audio` (the reference returns a new buffer; IRB_ERR_ARG otherwise).
 * How the samples reach the device.  Buffers in ordinary (pageable) memory -- a juce::AudioBuffer, a std::vector, a numpy array
 * -- are NEVER made device-visible: their samples pass through the library's own pinned staging ("staged form": chunks packed by
 * up to IRB_OFFLINE_THREADS host threads, default 5 with the caller, while the device already transforms the chunks before).
 * When audio AND out are HIP host memory already (irb_host_alloc, hipHostMalloc, or registered by the caller itself) the kernels
 * read and write them in place ("direct form", no host copy); the same for the impulse response on its own.
 * IRB_OFFLINE_DIRECT=1 opts in to registering pageable buffers with the runtime for the duration of the call (the round-4
 * default; faster by ~0.1 ms per call, but two calls in ~400 000 read a few stale pages of a registered pageable input --
 * INTEGRATION.md section 1); IRB_OFFLINE_DIRECT=0 forbids the direct form altogether.  irb_convolve_last_profile tells which
 * form a call took.  On any error the result buffer is cleared (the reference's give-up value, fp/convolution.cpp:39-42).
 * The calling thread's engine (device buffers, pinned staging, streams) is kept for the next call of a similar size;
 * it is freed by irb_release_cached_plans() (irb_deconv.h) or when the thread exits.        */
int  irb_convolve_periodic(int device,
                           const float *const *audio, int audio_ch, int La,
                           const float *const *ir, int ir_ch, int Lh,
                           int processBlockSize, int keep_reference_tail,
                           float *const *out);

/* Quiet section (no reference counterpart: the reference runs on the CPU and owns no device): between _begin and _end no
 * resident real-time server (IRB_FDL_RESIDENT) runs on `device` -- live ones leave after the request in hand, audio threads
 * serve their callbacks with one kernel launch each and resume the resident form afterwards -- so that a device-wide
 * runtime wait made in between (hipFree, hipHostFree, hipDeviceSynchronize, by the host or by another library) is not held
 * up by a server kernel.  Counted: sections may nest and overlap across threads.  Every such wait inside this library
 * already sits in one. */
int  irb_device_quiet_begin(int device);
int  irb_device_quiet_end(int device);

/* Where the time of the calling thread's last irb_convolve_periodic call went (measurement support for bench.py; no reference
 * counterpart: fp::convolution::convolvePeriodic, fp/convolution.cpp:14-242, has no timing code).  Host figures are wall-clock
 * milliseconds on the thread that did the work; they overlap each other and the device figures (the call is a pipeline), so
 * they do not add up to total_ms.  Device figures are sums over the chunks of hipEvent intervals on the queues the work ran
 * on, collected only while irb_convolve_profile_events(1) is in force (-1 otherwise: reading events costs host time). */
typedef struct irb_convolve_profile {
    double total_ms;            /* the whole call                                                           */
    double setup_ms;            /* engine look-up (or creation: engine_created), FDL reset, IR upload + transform enqueued */
    double register_ms;         /* direct form: resolving (opt-in: hipHostRegister-ing) the caller's audio, IR and result buffers */
    double pack_ms;             /* staged form, calling thread: caller's audio -> pinned staging             */
    double unpack_ms;           /* staged form, calling thread: pinned staging -> caller's out (its share of the chunks) */
    double unpack_helper_ms;    /* staged form, unpacking thread: the same for the other chunks              */
    double wait_ms;             /* calling thread blocked on the device (direct: the one wait of the call)   */
    double wait_helper_ms;      /* staged form: unpacking thread blocked on output chunks' events            */
    double stall_ms;            /* staged form: calling thread waiting for the unpacking thread at the end   */
    double fft_in_ms;           /* device: the IR's partition FFTs + forward FFTs of all blocks (direct: reading the caller's IR and audio
                                   over PCIe; staged: also the host's packing it waits for, and the multiply)  */
    double mac_ms;              /* device: the multiply-accumulate (direct form; -1 in the staged form)     */
    double fft_out_ms;          /* device: inverse FFTs + overlap-add of all blocks, written to the caller's buffer / the staging */
    double device_ms;           /* device: first forward-FFT launch -> last inverse-FFT launch done          */
    int direct;                 /* 1: the kernels read / wrote the caller's (HIP host, or opt-in registered) memory; 0: staged copies */
    int chunks, blocks_per_chunk, pieces, steps;   /* steps = floor(La/B) + P (+1 without the reference's dropped tail); pieces = multiplies */
    int helper_thread, engine_created, mac_on_matrix_cores, mac_launches;
    double pack_helper_ms;      /* (fields from here on: appended in round 5) staged form, helper threads: caller's audio / IR -> pinned staging (their sum) */
    int registered_ranges;      /* direct form: ranges of pageable memory registered for the call (0 unless IRB_OFFLINE_DIRECT=1) */
    int threads;                /* host threads that worked on the call (1 = the caller alone)                */
    int in_chunks;              /* staged form: input chunks (forward-FFT launches); `chunks` counts the output chunks */
} irb_convolve_profile;
int  irb_convolve_last_profile(irb_convolve_profile *out);
/* switch the device-event part of the profile on / off for the whole process; returns the previous setting (measurement
 * support; no reference counterpart) */
int  irb_convolve_profile_events(int on);

/* Which kernel a call of n_blocks blocks per stream multiplies with on this handle ("k_step_fused", "k_step_cols",
 * "k_cmac_tiled_pk<16,8>", "k_cmac_mfma" ...): what the handle's geometry, the device and the tuning overrides selected at
 * create, not a guess.  Measurement support; no reference counterpart (the reference has one loop, fp/convolution.cpp:171-202). */
const char *irb_fdl_mac_kernel_name(const irb_fdl *h, int n_blocks);

/* ---- several GPUs from one host process ------------------------------------------------------------------
 * No reference counterpart (the reference is one plug-in instance on one CPU thread); the per-instance work is
 * the same as irb_fdl_* above.  Instances never communicate, so a group of handles shards them by contiguous ranges: device d of D owns
 * instances [d*I/D, (d+1)*I/D) (balanced to within one).  No peer access, no collective.  irb_fdl_multi_process
 * enqueues pack/H2D/kernels/D2H on every device's stream from the calling thread and only then waits, so the
 * devices run concurrently.  `devices` may name a device more than once (tests on a one-GPU box do). */
typedef struct irb_fdl_multi irb_fdl_multi;
int  irb_fdl_multi_create(const irb_fdl_config *cfg /* n_instances = total over all devices; .device, .stream ignored */,
                          const int *devices, int n_devices, irb_fdl_multi **out);
void irb_fdl_multi_destroy(irb_fdl_multi *m);
/* which shard owns `instance`, and its index there (no reference counterpart) */
int  irb_fdl_multi_owner(const irb_fdl_multi *m, int instance, int *shard, int *local_instance);
int  irb_fdl_multi_set_ir(irb_fdl_multi *m, int instance, const float *const *ir, int ir_channels, int len);
int  irb_fdl_multi_reset(irb_fdl_multi *m);
/* the FDL steps of irb_fdl_process (fp/convolution.cpp:128-213) on every shard; in/out: one host pointer per
 * GLOBAL stream (instance*channels + ch), n_blocks*block samples each */
int  irb_fdl_multi_process(irb_fdl_multi *m, const float *const *in, float *const *out, int n_blocks);

/* ---- processBlock-shaped adapter ---------------------------------------------
 * Replaces the IRCAP_IDLE branch of IRBaboonAudioProcessor::processBlock
 * (Source/PluginProcessor.cpp:408-562) with its set-up in the ctor (:57-82) and
 * prepareToPlay (:164-234): host-block <-> 256 re-blocking (:421-445, :525-560),
 * latency max(hostBlock, block) (:169-174), IR channel 0 for every audio channel
 * (:485), irPartitions fixed by ir_ring_len (:225), and the rolling
 * one-partition-per-block IR refresh (:455-461).  Output gain and the limiter
 * (:567-574) stay in the plug-in.  Default IR: unit delta at sample 100 (:46).   */
int  irb_rt_create(int device, int channels, int block, int ir_ring_len, irb_rt **out);
void irb_rt_destroy(irb_rt *p);
int  irb_rt_prepare_to_play(irb_rt *p, int samples_per_block);
int  irb_rt_release_resources(irb_rt *p);
/* IRtoConvolve (:411-414): host samples, copied; must hold >= partitions*block.  */
int  irb_rt_set_ir(irb_rt *p, const float *ir, int len);
/* The IRCAP_IDLE branch itself (Source/PluginProcessor.cpp:408-562).  buffer: planar host [channels][n],
 * processed in place.                                                               */
int  irb_rt_process_block(irb_rt *p, float *const *buffer, int n);
/* Replaces processBlockBypassed (Source/PluginProcessor.cpp:584-590): host-only ring that keeps the plug-in's
 * block delay while bypassed.  n must equal the prepared samples_per_block.            */
int  irb_rt_process_block_bypassed(irb_rt *p, float *const *buffer, int n);
/* Measurement support for bench.py (no reference counterpart: the reference has no timing code): runs n_callbacks
 * irb_rt_process_block calls of n samples per channel on white noise and writes each callback's wall time in
 * microseconds to us_out[n_callbacks]; irb_rt_counters reports callbacks made and what they asked of the HIP
 * runtime (kernel launches, blocking waits, copy commands) since prepare_to_play.                               */
int  irb_rt_time_callbacks(irb_rt *p, int n, int n_callbacks, float *us_out);
int  irb_rt_counters(const irb_rt *p, unsigned long long *callbacks, unsigned long long *launches,
                     unsigned long long *waits, unsigned long long *copies);
int  irb_rt_latency(const irb_rt *p);
int  irb_rt_partitions(const irb_rt *p);

const char *irb_last_error(void);
const char *irb_version(void);

#ifdef __cplusplus
}
#endif
#endif
