/*
 * fp_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's uniformly partitioned convolution
 * path (Flixor/IRBaboon), used ONLY by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg as the checker / CPU baseline.  Nothing under
 * irbaboon_amd/ or include/ may include, link or call this.
 *
 * PARITY UNPINNED: the reference ships no golden vectors or tests
 *  and cannot be compiled in this image (it needs the
 * un-vendored JUCE 6.0.1 and Boost 1.69 headers).  The restatement is pinned
 * only by (a) the reference's own known-answer constant, IRpulse = unit
 * delta at sample 100 (Source/PluginProcessor.cpp:46) => pure delay, and
 * (b) float64 direct convolution, the mathematical definition of what
 * fp::convolution::convolvePeriodic computes.  The FFT butterflies live in
 * the third-party juce::dsp::FFT (module juce_dsp, JUCE 6.0.1, not in
 * /root/reference); fp_fft_* restates the published algorithm of its
 * portable fallback engine (kissfft-style radix-4/2 decimation in time,
 * twiddles computed in double and stored as float, inverse scaled by 1/N).
 *
 * Every function cites the reference file:line it follows
 * (paths relative to /root/reference).
 */
#ifndef FP_ORACLE_H
#define FP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- juce::dsp::FFT stand-in (call sites fp/convolution.cpp:75-77,123,144,206) ---- */
typedef struct fp_fft fp_fft;
fp_fft *fp_fft_create(int order);            /* N = 1 << order                       */
void    fp_fft_destroy(fp_fft *f);
int     fp_fft_size(const fp_fft *f);
/* d holds 2N floats; reads d[0..N) real, writes N interleaved complex bins (bins
 * above N/2 are the conjugate mirror, as the fallback engine leaves them). */
void    fp_fft_forward_real(const fp_fft *f, float *d);
/* consumes bins 0..N/2 of d, writes N real samples to d[0..N) scaled by 1/N.      */
void    fp_fft_inverse_real(const fp_fft *f, float *d);

/* ---- fp::tools (fp/tools.cpp:14-30, 44-52) ---- */
void fp_complexMul(float *a, float *b, float c, float d);
void fp_sumToMono(float *ch0, float *ch1, int n);

/* ---- fp::convolution::convolvePeriodic (fp/convolution.cpp:14-242) ----
 * audio: planar [audio_ch][La]; ir: planar [ir_ch][Lh]; out: planar
 * [audio_ch][La+Lh-1], overwritten.  Returns 0, or -1 for an unknown channel
 * layout (output cleared, as the reference does at :39-42).
 * Includes the dropped-tail quirk (:236-238). */
int fp_convolvePeriodic(const float *const *audio, int audio_ch, int La,
                        const float *const *ir, int ir_ch, int Lh,
                        int processBlockSize, float *const *out);

/* ---- fp::convolution::convolveNonPeriodic (fp/convolution.cpp:246-347) ---- */
int fp_convolveNonPeriodic(const float *const *audio, int audio_ch, int La,
                           const float *const *ir, int ir_ch, int Lh,
                           float *const *out);

/* ---- config-5 chain: tools::fftTransform / fftInvTransform (fp/tools.cpp:321-369),
 *      tools::complexDivCartesian (:72-86), convolution::averagingFilter (fp/convolution.cpp:406-546),
 *      convolution::deconvolve (:351-403), ir::shifteroo (fp/ir.cpp:85-103), ir::invertFilter (:13-18),
 *      tools::generatePulse (fp/tools.cpp:235-241).  Mono buffers, as deconvolve forces (:353-356). ---- */
int  fp_nextPowerOfTwo(int x);
/* in: len samples; out: 2*N floats, N = nextPowerOfTwo(len); returns N */
int  fp_fftTransform(const float *in, int len, float *out);
/* the same with formatAmplPhase = true (fp/tools.cpp:337-342): (amplitude, atan2(im, amplitude)) per bin 0..N/2 */
int  fp_fftTransformAmplPhase(const float *in, int len, float *out);
/* in: 2*N floats (fftSize = 2N); out: N samples */
void fp_fftInvTransform(const float *in, int fftSize, float *out);
void fp_complexDivCartesian(float *a, float *b, float c, float d);
/* buffer: fftSize floats (one channel), modified in place */
void fp_averagingFilter(float *buffer, int fftSize, double octaveFraction, double sampleRate, int logAvg,
                        int includePhase, int includeAmplitude);
/* out: N = nextPowerOfTwo(max(len_num, len_den)) samples; returns N */
int  fp_deconvolve(const float *num, int len_num, const float *den, int len_den, double sampleRate,
                   int smoothing, int includePhase, int includeAmplitude, float *out);
void fp_shifteroo(float *buf, int n);
int  fp_invertFilter(const float *ir, int len, int sampleRate, float *out);

/* ---- fp::ExpSineSweep (fp/ExpSineSweep.cpp:26-41, 59-79, 101-164, 212-260): the sweep that feeds config 5.
 *      All in double, as the reference.  generate writes (int)(SR*durationSecs) samples and returns that count. ---- */
int    fp_sweep_generate(double durationSecs, double sampleRate, double lowFreq, double highFreq, double dBGain, double *out);
int    fp_sweep_generateInv(double durationSecs, double sampleRate, double lowFreq, double highFreq, double dBGain, double *out);
int    fp_sweep_getSampleIndexAtFreq(double freq, double durationSecs, double sampleRate, double lowFreq, double highFreq);
double fp_sweep_getFreqAtSampleIndex(int index, double durationSecs, double sampleRate, double lowFreq, double highFreq);
void   fp_sweep_linFadeout(double *sweep, int n, int index);

/* ---- IRBaboonAudioProcessor convolution state (Source/PluginProcessor.cpp
 *      ctor :57-82, prepareToPlay :164-234, processBlock IDLE branch :408-562,
 *      rings per fp/CircularBufferArray.cpp:94-128) ----
 * channels = generalInputAudioChannels (2 in the plug-in), block = processBlockSize
 * (256), ir_ring_len = IRpulse length (2048) that fixes irPartitions (:225).
 * The IR fed to the rolling refresh (:455-461) is set with fp_rt_set_ir; it must
 * hold at least irPartitions*block samples (the reference does not check, :457). */
typedef struct fp_rt fp_rt;
fp_rt *fp_rt_create(int channels, int block, int ir_ring_len);
void   fp_rt_destroy(fp_rt *p);
void   fp_rt_prepareToPlay(fp_rt *p, int samplesPerBlock);
void   fp_rt_set_ir(fp_rt *p, const float *ir, int len);
/* in-place on planar buffer[channels][n]; convolution part only (no gain/limiter). */
void   fp_rt_processBlock(fp_rt *p, float *const *buffer, int n);
/* processBlockBypassed (Source/PluginProcessor.cpp:584-590): the bypass ring keeps the plug-in's block delay.
 * n must equal the prepared host block size. */
void   fp_rt_processBlockBypassed(fp_rt *p, float *const *buffer, int n);
int    fp_rt_latency(const fp_rt *p);
int    fp_rt_partitions(const fp_rt *p);

#ifdef __cplusplus
}
#endif
#endif
