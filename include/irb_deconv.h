/*
 * irb_deconv.h -- C ABI of the config-5 chain (libirbfdl.so): the exp-sine-sweep deconvolution path of
 * Flixor/IRBaboon.  Large real FFTs and the spectral division run on the MI355X; the fractional-octave
 * smoothing and the half swap are host code (the north star keeps fp::ir IR preparation on the host).
 * Same conventions as irb_fdl.h (error codes, irb_last_error()).  Paths are relative to the reference.
 *
 * Spectrum layout = the reference's: a buffer of 2N floats, interleaved (re, im) for bins 0..N/2
 * (Nyquist at float index N), unnormalised forward, inverse scaled by 1/N (fp/tools.cpp:321-369).
 */
#ifndef IRB_DECONV_H
#define IRB_DECONV_H

#include "irb_fdl.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct irb_lfft irb_lfft;     /* plan for one FFT length, device resident */

/* Replaces: tools::nextPowerOfTwo (fp/tools.cpp:180-196). */
int  irb_next_power_of_two(int x);

/* Replaces: tools::fftTransform(buffer) (fp/tools.cpp:321-346): len host samples -> out[2N],
 * N = nextPowerOfTwo(len) (channel 0 only, as the reference copies, :328).  len >= 17. */
int  irb_fft_transform(int device, const float *in, int len, float *out);
/* Replaces: tools::fftTransform(buffer, formatAmplPhase = true) (fp/tools.cpp:321-346, the loop at :337-342): bins
 * 0..N/2 of the same transform as (amplitude, phase) pairs -- tools::binAmpl / binPhase (fp/tools.cpp:222-231).  As in the
 * reference the amplitude overwrites the real part before binPhase reads it, so the stored phase is atan2(im, amplitude). */
int  irb_fft_transform_ampl_phase(int device, const float *in, int len, float *out);

/* Replaces: tools::fftInvTransform(buffer) (fp/tools.cpp:351-369): in[fft_size = 2N] -> out[N]. */
int  irb_fft_inv_transform(int device, const float *in, int fft_size, float *out);

/* Replaces: convolution::averagingFilter (fp/convolution.cpp:406-546), one channel, in place, HOST code. */
void irb_averaging_filter(float *buffer, int fft_size, double octave_fraction, double sample_rate,
                          int log_avg, int include_phase, int include_amplitude);

/* Replaces: ir::shifteroo (fp/ir.cpp:85-103), HOST code. */
void irb_shifteroo(float *buffer, int n);

/* Replaces: convolution::deconvolve(numerator, denominator, sampleRate, smoothing, includePhase,
 * includeAmplitude) (fp/convolution.cpp:351-403): mono, the shorter buffer zero extended, FFT length
 * N = nextPowerOfTwo(max length) with no further padding (so the division is circular), zero
 * denominators skipped (fp/tools.cpp:74-77), three 1/13-octave smoothing passes when `smoothing`,
 * half swap when !include_phase.  out must hold N samples (out_len >= N). */
int  irb_deconvolve(int device, const float *num, int len_num, const float *den, int len_den,
                    double sample_rate, int smoothing, int include_phase, int include_amplitude,
                    float *out, int out_len);

/* Replaces: ir::invertFilter(buffer, sampleRate) (fp/ir.cpp:13-18). */
int  irb_invert_filter(int device, const float *ir, int len, int sample_rate, float *out, int out_len);

/* Replaces: AudioBuffer<float> convolution::convolveNonPeriodic(audio, ir) (fp/convolution.cpp:246-347): one
 * FFT of length N = smallest power of two >= La+Lh-1 per signal, bin-wise product over bins 0..N/2
 * (tools::complexMul), inverse, first La+Lh-1 samples.  Same channel-layout rules and error behaviour as
 * irb_convolve_periodic (unknown layout: cleared output + IRB_ERR_LAYOUT).  out: [audio_ch][La+Lh-1]; every sample of it is
 * written (computed, or cleared on an error / empty input).  For the duration of the call the audio, IR and result buffers are
 * registered with the HIP runtime and the copy engine moves the samples straight between them and the device; the stereo-IR-to-
 * mono-audio layout and buffers the runtime refuses to register go through host staging instead. */
int  irb_convolve_nonperiodic(int device, const float *const *audio, int audio_ch, int La,
                              const float *const *ir, int ir_ch, int Lh, float *const *out);

/* Replaces: ir::IRtoRealFFTRaw(buffer, irPartSize) (fp/ir.cpp:106-147) without its hard-coded text dump:
 * the mono IR cut into len/irPartSize + 1 partitions, each transformed with N = 2*irPartSize and packed
 * (im(0) := re(N/2)), consolidated as rows of N floats.  irPartSize must be a power of two >= 16.
 * out_len >= (len/irPartSize + 1) * 2*irPartSize.  Returns the number of rows through *n_rows.  The IR and result buffers are
 * registered with the HIP runtime for the call (one launch transforms the IR where it lies, one copy brings the rows back; a small
 * device buffer is kept per calling thread and freed by irb_release_cached_plans()). */
int  irb_ir_to_real_fft_raw(int device, const float *ir, int len, int ir_part_size, float *out, int out_len, int *n_rows);

/* ---- fp::ExpSineSweep, HOST code (the sweep generator stays on the host: one-off O(N) work) --------------
 * Replaces: ExpSineSweep::generate / generateInv / getSampleIndexAtFreq / getFreqAtSampleIndex / linFadeout
 * (fp/ExpSineSweep.cpp:26-41, 59-79, 101-146, 147-164, 212-260).  Samples are double like the reference's
 * AudioBuffer<double>; `generate` writes (int)(sampleRate*durationSecs) samples (out_len must hold them) and
 * returns that count, or a negative error code.  linFadeout takes the start index (getSampleIndexAtFreq). */
int    irb_sweep_generate(double duration_secs, double sample_rate, double low_freq, double high_freq, double dB_gain,
                          double *out, int out_len);
int    irb_sweep_generate_inv(double duration_secs, double sample_rate, double low_freq, double high_freq, double dB_gain,
                              double *out, int out_len);
int    irb_sweep_sample_index_at_freq(double freq, double duration_secs, double sample_rate, double low_freq, double high_freq);
double irb_sweep_freq_at_sample_index(int index, double duration_secs, double sample_rate, double low_freq, double high_freq);
int    irb_sweep_lin_fadeout(double *sweep, int n, int index);

/* The one-shot entry points (here and irb_convolve_periodic) keep the calling thread's most recent FFT plan /
 * engine (device buffers, twiddle tables, stream) for reuse; this releases them now.  The cache is per thread and is
 * also freed when its thread exits (a thread-local guard object), so a host thread pool does not leak.  No reference counterpart
 * (the reference constructs a dsp::FFT per call, fp/tools.cpp:331,359). */
void irb_release_cached_plans(void);

/* ---- reusable plan with device-resident buffers (bench, repeated transforms) ----
 * The counterpart of constructing juce::dsp::FFT objects once and reusing them (fp/convolution.cpp:73-77);
 * forward/inverse are performRealOnlyForwardTransform / performRealOnlyInverseTransform (fp/tools.cpp:335,363)
 * on device buffers; buffers / synchronize / mark / elapsed_ms have no reference counterpart. */
int  irb_lfft_create(int device, int fft_len, int max_batch, irb_lfft **out);   /* fft_len: 2^5 .. 2^23 */
void irb_lfft_destroy(irb_lfft *p);
/* scratch owned by the plan (no reference counterpart): d_time [max_batch][N] floats, d_spec [max_batch][2N] floats.
 * The plan zeroes d_spec once and a forward transform INTO d_spec then writes bins 0..N/2 only (the reference layout's
 * zeros above the Nyquist bin are already there); a caller that stores into floats N+2 .. 2N-1 of d_spec itself must
 * clear them again before the next forward transform into it.  Forward transforms into any other buffer write all 2N floats. */
int  irb_lfft_buffers(irb_lfft *p, float **d_time, float **d_spec);
/* for a caller that has stored into floats N+2 .. 2N-1 of the plan's d_spec (through the pointer above): the next forward transform
 * into d_spec of the plan's full batch writes all 2N floats of every item again and the short form resumes behind it
 * (no reference counterpart: the reference's transform rewrites its whole work buffer every time, fp/tools.cpp:331-335) */
int  irb_lfft_invalidate_spec(irb_lfft *p);
/* performRealOnlyForwardTransform / performRealOnlyInverseTransform (fp/tools.cpp:335,363) on device buffers */
int  irb_lfft_forward_device(irb_lfft *p, const float *d_in, float *d_spec, int batch);   /* asynchronous */
int  irb_lfft_inverse_device(irb_lfft *p, const float *d_spec, float *d_out, int batch);  /* asynchronous */
/* stream wait and hipEvent timing for bench.py; no reference counterpart */
int  irb_lfft_synchronize(irb_lfft *p);
int  irb_lfft_mark(irb_lfft *p, int slot);
int  irb_lfft_elapsed_ms(irb_lfft *p, float *ms);

#ifdef __cplusplus
}
#endif
#endif
