/*
 * fp_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See fp_oracle.h for scope, pinning status ("parity unpinned") and the rule
 * that only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may use it.  All arithmetic is fp32 like the reference; loop structure
 * follows the cited reference lines so that the timed CPU baseline does the
 * same work per partition (copy, complexMul through pointers, 2N adds).
 */
#include "fp_oracle.h"

#define _USE_MATH_DEFINES
#define _GNU_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* juce::dsp::FFT fallback engine, restated (JUCE 6.0.1, juce_dsp; absent     */
/* from /root/reference).  Published algorithm: recursive mixed-radix          */
/* decimation in time (radix 4 first, then 2), one twiddle table of N entries */
/* computed in double precision and rounded to float, inverse = conjugate     */
/* table followed by a 1/N scale.  Real transforms run the complex transform  */
/* on (x, 0) / on the Hermitian-completed spectrum.                            */
/* ------------------------------------------------------------------------- */

typedef struct { float re, im; } cpx;

struct fp_fft {
    int n;
    int nfac;
    int radix[32], length[32];
    cpx *tw_fwd, *tw_inv;
    cpx *scratch;
};

static void make_twiddles(cpx *tw, int n, int inverse)
{
    const double f = (inverse ? 2.0 : -2.0) * 3.14159265358979323846 / (double) n;
    for (int i = 0; i < n; i++) {
        double ph = (double) i * f;
        tw[i].re = (float) cos(ph);
        tw[i].im = (float) sin(ph);
    }
}

fp_fft *fp_fft_create(int order)
{
    fp_fft *f = (fp_fft *) calloc(1, sizeof *f);
    f->n = 1 << order;
    f->tw_fwd = (cpx *) malloc(sizeof(cpx) * (size_t) f->n);
    f->tw_inv = (cpx *) malloc(sizeof(cpx) * (size_t) f->n);
    f->scratch = (cpx *) malloc(sizeof(cpx) * (size_t) f->n);
    make_twiddles(f->tw_fwd, f->n, 0);
    make_twiddles(f->tw_inv, f->n, 1);
    int n = f->n;
    while (n > 1) {
        int r = (n % 4 == 0) ? 4 : 2;
        n /= r;
        f->radix[f->nfac] = r;
        f->length[f->nfac] = n;
        f->nfac++;
    }
    return f;
}

void fp_fft_destroy(fp_fft *f)
{
    if (!f) return;
    free(f->tw_fwd); free(f->tw_inv); free(f->scratch); free(f);
}

int fp_fft_size(const fp_fft *f) { return f->n; }

static void bfly2(cpx *d, const cpx *tw, int stride, int len)
{
    for (int i = 0; i < len; i++) {
        cpx w = tw[i * stride], b = d[i + len], s;
        s.re = b.re * w.re - b.im * w.im;
        s.im = b.re * w.im + b.im * w.re;
        d[i + len].re = d[i].re - s.re;
        d[i + len].im = d[i].im - s.im;
        d[i].re += s.re;
        d[i].im += s.im;
    }
}

static void bfly4(cpx *d, const cpx *tw, int stride, int len, int inverse)
{
    for (int i = 0; i < len; i++) {
        cpx w1 = tw[i * stride], w2 = tw[2 * i * stride], w3 = tw[3 * i * stride];
        cpx a = d[i], b = d[i + len], c = d[i + 2 * len], e = d[i + 3 * len];
        cpx s0, s1, s2, s3, s4, s5;
        s0.re = b.re * w1.re - b.im * w1.im;  s0.im = b.re * w1.im + b.im * w1.re;
        s1.re = c.re * w2.re - c.im * w2.im;  s1.im = c.re * w2.im + c.im * w2.re;
        s2.re = e.re * w3.re - e.im * w3.im;  s2.im = e.re * w3.im + e.im * w3.re;
        s3.re = s0.re + s2.re;  s3.im = s0.im + s2.im;
        s4.re = s0.re - s2.re;  s4.im = s0.im - s2.im;
        s5.re = a.re - s1.re;   s5.im = a.im - s1.im;
        a.re += s1.re;          a.im += s1.im;
        d[i + 2 * len].re = a.re - s3.re;  d[i + 2 * len].im = a.im - s3.im;
        d[i].re = a.re + s3.re;            d[i].im = a.im + s3.im;
        if (inverse) {
            d[i + len].re = s5.re - s4.im;      d[i + len].im = s5.im + s4.re;
            d[i + 3 * len].re = s5.re + s4.im;  d[i + 3 * len].im = s5.im - s4.re;
        } else {
            d[i + len].re = s5.re + s4.im;      d[i + len].im = s5.im - s4.re;
            d[i + 3 * len].re = s5.re - s4.im;  d[i + 3 * len].im = s5.im + s4.re;
        }
    }
}

static void fft_rec(const fp_fft *f, const cpx *tw, int inverse, const cpx *in, cpx *out,
                    int stride, int level)
{
    const int r = f->radix[level], len = f->length[level];
    if (len == 1) {
        for (int i = 0; i < r; i++) out[i] = in[i * stride];
    } else {
        for (int i = 0; i < r; i++)
            fft_rec(f, tw, inverse, in + i * stride, out + i * len, stride * r, level + 1);
    }
    if (r == 4) bfly4(out, tw, stride, len, inverse);
    else        bfly2(out, tw, stride, len);
}

static void fft_complex(const fp_fft *f, const cpx *in, cpx *out, int inverse)
{
    if (f->n == 1) { out[0] = in[0]; return; }
    fft_rec(f, inverse ? f->tw_inv : f->tw_fwd, inverse, in, out, 1, 0);
    if (inverse) {
        const float s = 1.0f / (float) f->n;
        for (int i = 0; i < f->n; i++) { out[i].re *= s; out[i].im *= s; }
    }
}

void fp_fft_forward_real(const fp_fft *f, float *d)
{
    const int n = f->n;
    if (n == 1) return;                      /* the engine's own early-out for a 1-point transform */
    for (int i = 0; i < n; i++) { f->scratch[i].re = d[i]; f->scratch[i].im = 0.0f; }
    fft_complex(f, f->scratch, (cpx *) d, 0);
}

void fp_fft_inverse_real(const fp_fft *f, float *d)
{
    const int n = f->n;
    if (n == 1) return;
    cpx *in = (cpx *) d;
    for (int i = n >> 1; i < n; i++) { in[i].re = in[n - i].re; in[i].im = -in[n - i].im; }
    fft_complex(f, in, f->scratch, 1);
    for (int i = 0; i < n; i++) { d[i] = f->scratch[i].re; d[i + n] = f->scratch[i].im; }
}

/* ------------------------------------------------------------------------- */
/* fp::tools                                                                   */
/* ------------------------------------------------------------------------- */

/* fp/tools.cpp:44-52 */
void fp_complexMul(float *a, float *b, float c, float d)
{
    float re = (*a) * c - (*b) * d;
    float im = (*b) * c + (*a) * d;
    *a = re;
    *b = im;
}

/* fp/tools.cpp:14-30 */
void fp_sumToMono(float *p0, float *p1, int n)
{
    for (int i = 0; i < n; i++) {
        p0[i] += p1[i];
        p0[i] /= 2.0f;
        p1[i] = 0.0f;
    }
}

/* ------------------------------------------------------------------------- */
/* fp::convolution                                                             */
/* ------------------------------------------------------------------------- */

enum { L_UNKNOWN, L_IRMONO_AMONO, L_IRMONO_ASTEREO, L_IRSTEREO_AMONO, L_IRSTEREO_ASTEREO };

/* fp/convolution.cpp:28-37, 262-269 */
static int classify(int ir_ch, int audio_ch)
{
    if (ir_ch == 1 && audio_ch == 1) return L_IRMONO_AMONO;
    if (ir_ch == 1 && audio_ch == 2) return L_IRMONO_ASTEREO;
    if (ir_ch == 2 && audio_ch == 1) return L_IRSTEREO_AMONO;
    if (ir_ch == 2 && audio_ch == 2) return L_IRSTEREO_ASTEREO;
    return L_UNKNOWN;
}

static int highest_bit(int n) { int b = 0; while ((1 << (b + 1)) <= n) b++; return b; }

/* fp/convolution.cpp:14-242 */
int fp_convolvePeriodic(const float *const *audio, int audio_ch, int La,
                        const float *const *ir, int ir_ch, int Lh,
                        int B, float *const *out)
{
    const int Lout = La + Lh - 1;
    for (int c = 0; c < audio_ch; c++) memset(out[c], 0, sizeof(float) * (size_t) (Lout > 0 ? Lout : 0));   /* :20-21 */

    const int layout = classify(ir_ch, audio_ch);
    if (layout == L_UNKNOWN) return -1;                                   /* :39-42 */
    if (Lh <= 0) return 0;   /* reference: irPartitions == 0 -> 'p % 0' at :129 (undefined); zeros here */

    int N = 1;
    while (N < 2 * B - 1) N *= 2;                                         /* :45-48 */
    const int fftBlockSize = 2 * N;                                       /* :49 */
    const int P = (int) ceilf((float) Lh / (float) B);                    /* :52 */

    const size_t irStride = (size_t) ir_ch * fftBlockSize, auStride = (size_t) audio_ch * fftBlockSize;
    float *irFft = (float *) calloc((size_t) (P > 0 ? P : 1) * irStride, sizeof(float));      /* :55-60 */
    float *auFft = (float *) calloc((size_t) (P > 0 ? P : 1) * auStride, sizeof(float));      /* :63-68 */
    int irFftCount = 0, auIdx = 0;

    fp_fft *fft = fp_fft_create(highest_bit(N));                          /* :71-77 */

    float *convResult = (float *) calloc(auStride, sizeof(float));        /* :80-81 */
    float *inplace = (float *) calloc(auStride, sizeof(float));           /* :83-84 */
    float *overlap = (float *) calloc((size_t) audio_ch * B, sizeof(float)); /* :86-87 */

    int p = 0, endAudio = 0;
    const int irFwdChMax = (layout == L_IRSTEREO_ASTEREO) ? 2 : 1;        /* :94-101 */

    do {
        if (p < P) {                                                      /* :106-125 */
            int n = ((p + 1) * B <= Lh) ? B : Lh - p * B;
            for (int c = 0; c < ir_ch; c++)
                memcpy(irFft + (size_t) p * irStride + (size_t) c * fftBlockSize, ir[c] + (size_t) p * B, sizeof(float) * (size_t) n);
            if (layout == L_IRSTEREO_AMONO)
                fp_sumToMono(irFft + (size_t) p * irStride, irFft + (size_t) p * irStride + fftBlockSize, fftBlockSize);
            for (int c = 0; c < irFwdChMax; c++)
                fp_fft_forward_real(fft, irFft + (size_t) p * irStride + (size_t) c * fftBlockSize);
            irFftCount++;
        }

        if (p * B <= La) {                                                /* :128-149 */
            auIdx = p % P;
            memset(auFft + (size_t) auIdx * auStride, 0, sizeof(float) * auStride);
            int n = ((p + 1) * B <= La) ? B : La - p * B;
            for (int c = 0; c < audio_ch; c++) {
                float *dst = auFft + (size_t) auIdx * auStride + (size_t) c * fftBlockSize;
                if (n > 0) memcpy(dst, audio[c] + (size_t) p * B, sizeof(float) * (size_t) n);
                fp_fft_forward_real(fft, dst);
            }
            if ((p + 1) * B > La) endAudio++;
        } else {
            endAudio++;                                                   /* :151-153 */
        }

        memset(convResult, 0, sizeof(float) * auStride);                  /* :158 */

        for (int ch = 0; ch < audio_ch; ch++) {                           /* :160-215 */
            float *conv = convResult + (size_t) ch * fftBlockSize;
            float *ovl = overlap + (size_t) ch * B;
            int k = endAudio - 1 > 0 ? endAudio - 1 : 0;                  /* :166 */
            int s = auIdx;                                                /* :167 */
            while (k < irFftCount) {                                      /* :171-202 */
                memcpy(inplace, auFft + (size_t) s * auStride, sizeof(float) * auStride);   /* :173 makeCopyOf, all channels */
                float *ip = inplace + (size_t) ch * fftBlockSize;
                int irCh = (layout == L_IRMONO_AMONO || layout == L_IRSTEREO_ASTEREO) ? ch : 0;  /* :176-180 */
                const float *h = irFft + (size_t) k * irStride + (size_t) irCh * fftBlockSize;
                for (int i = 0; i <= N; i += 2)                           /* :184-189 */
                    fp_complexMul(ip + i, ip + i + 1, h[i], h[i + 1]);
                for (int i = 0; i < fftBlockSize; i++)                    /* :193-195 */
                    conv[i] += ip[i];
                k++;
                s--;
                if (s < 0) s = P - 1;                                     /* :198-201 */
            }
            fp_fft_inverse_real(fft, conv);                               /* :206 */
            for (int i = 0; i < B; i++) {                                 /* :210-213 */
                conv[i] += ovl[i];
                ovl[i] = conv[B + i];
            }
        }

        for (int ch = 0; ch < audio_ch; ch++) {                           /* :219-230 */
            int n = ((p + 1) * B <= Lout) ? B : Lout - p * B;
            if (n > 0)
                memcpy(out[ch] + (size_t) p * B, convResult + (size_t) ch * fftBlockSize, sizeof(float) * (size_t) n);
        }
        p++;
    } while (endAudio < P);                                               /* :233 */

    fp_fft_destroy(fft);
    free(irFft); free(auFft); free(convResult); free(inplace); free(overlap);
    return 0;
}

/* fp/convolution.cpp:246-347 */
int fp_convolveNonPeriodic(const float *const *audio, int audio_ch, int La,
                           const float *const *ir, int ir_ch, int Lh,
                           float *const *out)
{
    const int Lout = La + Lh - 1;                                         /* :251 */
    const int layout = classify(ir_ch, audio_ch);
    if (layout == L_UNKNOWN) {                                            /* :271-275 */
        for (int c = 0; c < audio_ch; c++) memset(out[c], 0, sizeof(float) * (size_t) (Lout > 0 ? Lout : 0));
        return -1;
    }
    int N = 1;
    while (N < Lout) N *= 2;                                              /* :278-281 */
    const int fftBlockSize = 2 * N;
    fp_fft *fft = fp_fft_create(highest_bit(N));

    /* setSize(..., keepExisting) is taken as zero-extension (SURVEY.md App. B hazard) */
    float *a = (float *) calloc((size_t) audio_ch * fftBlockSize, sizeof(float));   /* :291 */
    float *h = (float *) calloc((size_t) ir_ch * fftBlockSize, sizeof(float));      /* :293 */
    for (int c = 0; c < audio_ch; c++) memcpy(a + (size_t) c * fftBlockSize, audio[c], sizeof(float) * (size_t) La);
    for (int c = 0; c < ir_ch; c++) memcpy(h + (size_t) c * fftBlockSize, ir[c], sizeof(float) * (size_t) Lh);

    int irFwdChMax = 1;
    if (layout == L_IRSTEREO_ASTEREO) irFwdChMax = 2;
    if (layout == L_IRSTEREO_AMONO) fp_sumToMono(h, h + fftBlockSize, fftBlockSize);  /* :301 */
    for (int c = 0; c < irFwdChMax; c++) fp_fft_forward_real(fft, h + (size_t) c * fftBlockSize);  /* :307-308 */

    for (int ch = 0; ch < audio_ch; ch++) {                               /* :311-337 */
        float *ap = a + (size_t) ch * fftBlockSize;
        fp_fft_forward_real(fft, ap);
        int irCh = (layout == L_IRMONO_AMONO || layout == L_IRSTEREO_ASTEREO) ? ch : 0;
        const float *hp = h + (size_t) irCh * fftBlockSize;
        for (int i = 0; i <= N; i += 2)
            fp_complexMul(ap + i, ap + i + 1, hp[i], hp[i + 1]);
        fp_fft_inverse_real(fft, ap);
        memcpy(out[ch], ap, sizeof(float) * (size_t) Lout);               /* :340-344 */
    }
    fp_fft_destroy(fft);
    free(a); free(h);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* config-5 chain                                                              */
/* ------------------------------------------------------------------------- */

/* fp/tools.cpp:180-196 */
int fp_nextPowerOfTwo(int x)
{
    if (x != 0 && (x & (x - 1)) == 0) return x;
    int r = 1;
    while (r <= x) r *= 2;      /* recursion at :189-196: first power of two strictly above x */
    return r;
}

/* fp/tools.cpp:321-346 (formatAmplPhase = false) */
int fp_fftTransform(const float *in, int len, float *out)
{
    const int N = fp_nextPowerOfTwo(len);
    memset(out, 0, sizeof(float) * 2 * (size_t) N);
    memcpy(out, in, sizeof(float) * (size_t) len);
    fp_fft *f = fp_fft_create(highest_bit(N));
    fp_fft_forward_real(f, out);
    fp_fft_destroy(f);
    return N;
}

/* fp/tools.cpp:321-346 with formatAmplPhase = true: the loop at :337-342 over `bin <= N; bin += 2`.  binAmpl (:222-225) is
 * sqrt(pow(re, 2.0) + pow(im, 2.0)) in double, stored as float INTO THE REAL SLOT; binPhase (:228-231) then reads that slot:
 * atan2(im, amplitude). */
int fp_fftTransformAmplPhase(const float *in, int len, float *out)
{
    const int N = fp_fftTransform(in, len, out);
    for (int bin = 0; bin <= N; bin += 2) {
        out[bin] = (float) sqrt(pow((double) out[bin], 2.0) + pow((double) out[bin + 1], 2.0));
        out[bin + 1] = (float) atan2((double) out[bin + 1], (double) out[bin]);
    }
    return N;
}

/* fp/tools.cpp:351-369 */
void fp_fftInvTransform(const float *in, int fftSize, float *out)
{
    const int N = fftSize / 2;
    float *tmp = (float *) malloc(sizeof(float) * (size_t) fftSize);
    memcpy(tmp, in, sizeof(float) * (size_t) fftSize);
    fp_fft *f = fp_fft_create(highest_bit(N));
    fp_fft_inverse_real(f, tmp);
    fp_fft_destroy(f);
    memcpy(out, tmp, sizeof(float) * (size_t) N);      /* :366 result is in the first half */
    free(tmp);
}

/* fp/tools.cpp:72-86 */
void fp_complexDivCartesian(float *a, float *b, float c, float d)
{
    if (c == 0.0 && d == 0.0) return;
    float re = ((*a) * c + (*b) * d) / (c * c + d * d);
    float im = ((*b) * c - (*a) * d) / (c * c + d * d);
    *a = re;
    *b = im;
}

/* fp/tools.cpp:199-209 */
static void roundToZero(float *x, float threshold)
{
    if (threshold < 0.0f) return;
    if (!signbit(*x) && *x < threshold) *x = 0.0f;
    if (signbit(*x) && *x > -(threshold)) *x = 0.0f;
}

/* fp/tools.cpp:213-218 */
static void roundTo1TenQuadrillionth(float *x)
{
    if (!signbit(*x) && *x < 1e-16) *x = 1e-16;
    if (signbit(*x) && *x > -1e-16) *x = -1e-16;
}

/* fp/tools.cpp:222-231; float arguments pick the float overloads of atan2 in C++ */
static float binAmpl(const float *p) { return (float) sqrt(pow((double) p[0], 2.0) + pow((double) p[1], 2.0)); }
static float binPhase(const float *p) { return atan2f(p[1], p[0]); }

/* fp/convolution.cpp:406-546, one channel */
void fp_averagingFilter(float *buffer, int fftSize, double octaveFraction, double sampleRate, int logAvg,
                        int includePhase, int includeAmplitude)
{
    if (!(fftSize != 0 && (fftSize & (fftSize - 1)) == 0)) return;           /* :412-415 */
    float *oldAmpl = (float *) calloc((size_t) fftSize / 2, sizeof(float));   /* :417 real-frequency amplitudes */
    float *newAmpl = (float *) calloc((size_t) fftSize, sizeof(float));       /* :418 */
    const int N = fftSize / 2;
    const double fractPerSide = octaveFraction / 2.0;
    const double nyquist = sampleRate / 2;
    const double freqPerBin = nyquist / (double) (N / 2);                     /* :425 */

    for (int bin = 0; bin <= N; bin += 2) oldAmpl[bin / 2] = binAmpl(buffer + bin);   /* :429-436 */

    float runningSum = 0.0f;
    int prevLowerBin = 0, prevUpperBin = -2;
    for (int bin = 0; bin <= N; bin += 2) {                                   /* :449 */
        double binFreq = (bin / 2) * freqPerBin;
        double lowerFreq = binFreq / pow(2.0, fractPerSide);
        int lowerBin = 2 * (int) round(lowerFreq / freqPerBin);
        double upperFreq = binFreq * pow(2.0, fractPerSide);
        int upperBin = 2 * (int) round(upperFreq / freqPerBin);
        double binRangeLength = (double) (upperBin - lowerBin) / 2.0 + 1.0;
        if (logAvg) {                                                         /* :482-505 */
            int subtractBin = prevLowerBin;
            while (subtractBin < lowerBin) {
                float v = oldAmpl[subtractBin / 2];
                roundTo1TenQuadrillionth(&v);
                v = logf(v);
                runningSum -= v;
                subtractBin += 2;
            }
            int addBin = prevUpperBin + 2;
            while (addBin <= upperBin) {
                if (addBin < fftSize) {
                    float v = oldAmpl[addBin / 2];
                    roundTo1TenQuadrillionth(&v);
                    v = logf(v);
                    runningSum += v;
                }
                addBin += 2;
            }
        } else {                                                              /* :509-515 */
            runningSum = 0.0f;
            for (int i = lowerBin; i <= upperBin; i += 2) runningSum += oldAmpl[i / 2];
        }
        float na = (float) (runningSum / binRangeLength);                     /* :519 */
        if (logAvg) na = expf(na);
        roundTo1TenQuadrillionth(&na);
        newAmpl[bin] = na;
        prevLowerBin = lowerBin;
        prevUpperBin = upperBin;
    }
    for (int bin = 0; bin <= N; bin += 2) {                                   /* :531-543 */
        float ampl = newAmpl[bin];
        roundToZero(buffer + bin, 1e-11f);
        roundToZero(buffer + bin + 1, 1e-11f);
        float phase = binPhase(buffer + bin);
        if (!includeAmplitude) ampl = 1.0f;
        if (!includePhase) phase = 0.0f;
        buffer[bin] = ampl * cosf(phase);
        buffer[bin + 1] = ampl * sinf(phase);
    }
    free(oldAmpl); free(newAmpl);
}

/* fp/ir.cpp:85-103 */
void fp_shifteroo(float *buf, int n)
{
    if (n < 2) return;
    int second = n / 2, first = second + (n % 2 != 0 ? 1 : 0);
    float *t = (float *) malloc(sizeof(float) * (size_t) n);
    memcpy(t, buf + first, sizeof(float) * (size_t) second);
    memcpy(t + second, buf, sizeof(float) * (size_t) first);
    memcpy(buf, t, sizeof(float) * (size_t) n);
    free(t);
}

/* fp/convolution.cpp:351-403.  setSize(.., keepExisting) = zero extension. */
int fp_deconvolve(const float *num, int len_num, const float *den, int len_den, double sampleRate,
                  int smoothing, int includePhase, int includeAmplitude, float *out)
{
    const int len = len_num > len_den ? len_num : len_den;                    /* :359-362 */
    const int N = fp_nextPowerOfTwo(len);
    float *a = (float *) calloc((size_t) len, sizeof(float)), *b = (float *) calloc((size_t) len, sizeof(float));
    memcpy(a, num, sizeof(float) * (size_t) len_num);
    memcpy(b, den, sizeof(float) * (size_t) len_den);
    float *A = (float *) malloc(sizeof(float) * 2 * (size_t) N), *Bf = (float *) malloc(sizeof(float) * 2 * (size_t) N);
    fp_fftTransform(a, len, A);                                               /* :365-366 */
    fp_fftTransform(b, len, Bf);
    for (int i = 0; i <= N; i += 2)                                           /* :373-378 */
        fp_complexDivCartesian(A + i, A + i + 1, Bf[i], Bf[i + 1]);
    if (smoothing) {                                                          /* :389-394 */
        float smoothPerAvg = 1.0 / 13.0;
        for (int i = 0; i < 3; i++)
            fp_averagingFilter(A, 2 * N, smoothPerAvg, sampleRate, 1, includePhase, includeAmplitude);
    }
    fp_fftInvTransform(A, 2 * N, out);                                        /* :397 */
    if (!includePhase) fp_shifteroo(out, N);                                  /* :400 */
    free(a); free(b); free(A); free(Bf);
    return N;
}

/* fp/ir.cpp:13-18 with tools::generatePulse(n) = delta at 0 (fp/tools.cpp:235-241, tools.hpp default) */
int fp_invertFilter(const float *ir, int len, int sampleRate, float *out)
{
    float *pulse = (float *) calloc((size_t) len, sizeof(float));
    pulse[0] = 1.0f;
    int N = fp_deconvolve(pulse, len, ir, len, (double) sampleRate, 1, 1, 1, out);
    free(pulse);
    return N;
}

/* ------------------------------------------------------------------------- */
/* fp::ExpSineSweep                                                            */
/* ------------------------------------------------------------------------- */
typedef struct { double SR, w1, w2, T, K, L; } sweep_par;

/* fp/ExpSineSweep.cpp:212-220 */
static sweep_par sweep_params(double durationSecs, double sampleRate, double lowFreq, double highFreq)
{
    sweep_par p;
    p.SR = sampleRate;
    p.T = p.SR * durationSecs;
    p.w1 = lowFreq / p.SR * 2 * M_PI;
    p.w2 = highFreq / p.SR * 2 * M_PI;
    p.K = p.T * p.w1 / log(p.w2 / p.w1);
    p.L = p.T / log(p.w2 / p.w1);
    return p;
}

/* fp/ExpSineSweep.cpp:26-41, gain via tools::dBToLin (fp/tools.cpp:94-96) */
int fp_sweep_generate(double durationSecs, double sampleRate, double lowFreq, double highFreq, double dBGain, double *out)
{
    const sweep_par p = sweep_params(durationSecs, sampleRate, lowFreq, highFreq);
    const int n = (int) p.T;
    const double gain = pow(10.0, dBGain / 20.0);
    for (int i = 0; i < n; i++) out[i] = gain * sin(p.K * (exp((double) i / p.L) - 1.0));
    return n;
}

/* fp/ExpSineSweep.cpp:59-79: reversed sweep with -6 dB/oct envelope */
int fp_sweep_generateInv(double durationSecs, double sampleRate, double lowFreq, double highFreq, double dBGain, double *out)
{
    const sweep_par p = sweep_params(durationSecs, sampleRate, lowFreq, highFreq);
    const int n = fp_sweep_generate(durationSecs, sampleRate, lowFreq, highFreq, dBGain, out);
    for (int i = 0; i < n / 2; i++) { double t = out[i]; out[i] = out[n - 1 - i]; out[n - 1 - i] = t; }
    const double k = pow(10.0, (-6.0 * log2(p.w2 / p.w1)) / 20.0 / p.T);
    double kIterator = k;
    for (int i = 0; i < n; i++) { out[i] *= kIterator; kIterator *= k; }
    return n;
}

/* fp/ExpSineSweep.cpp:243-258 */
int fp_sweep_getSampleIndexAtFreq(double freq, double durationSecs, double sampleRate, double lowFreq, double highFreq)
{
    const sweep_par p = sweep_params(durationSecs, sampleRate, lowFreq, highFreq);
    double lf = (p.w1 * p.SR / (2 * M_PI));
    double oct = log(freq / lf) / log(2);
    double totalOct = log(p.w2 / p.w1) / log(2);
    double t = oct / totalOct * p.T;
    int t_rounded = (int) round(t);
    if (t_rounded < 0 || t_rounded >= p.T) return -1;
    return t_rounded;
}

/* fp/ExpSineSweep.cpp:225-240 */
double fp_sweep_getFreqAtSampleIndex(int index, double durationSecs, double sampleRate, double lowFreq, double highFreq)
{
    const sweep_par p = sweep_params(durationSecs, sampleRate, lowFreq, highFreq);
    if (index < 0 || index >= (int) p.T) return -1;
    double lf = p.w1 * p.SR / (2 * M_PI);
    double totalOct = log(p.w2 / p.w1) / log(2);
    double oct = (double) index / p.T * totalOct;
    return lf * pow(2.0, oct);
}

/* fp/ExpSineSweep.cpp:147-164, index = getSampleIndexAtFreq(freq) */
void fp_sweep_linFadeout(double *sweep, int n, int index)
{
    if (n == 0 || index < 0) return;
    int fadeoutSampleLength = n - index;
    for (int i = index; i < n; i++)
        sweep[i] = sweep[i] * (fadeoutSampleLength - (i - index)) / fadeoutSampleLength;
}

/* ------------------------------------------------------------------------- */
/* fp::CircularBufferArray (fp/CircularBufferArray.cpp:94-128, 193-205)       */
/* ------------------------------------------------------------------------- */

typedef struct {
    int size, ch, n, rd, wr;
    float *data;                 /* [size][ch][n] */
} ring;

static void ring_init(ring *r, int size, int ch, int n)
{
    free(r->data);
    r->size = size; r->ch = ch; r->n = n; r->rd = 0; r->wr = 0;
    r->data = (float *) calloc((size_t) (size > 0 ? size : 1) * ch * n, sizeof(float));
}
static float *ring_at(ring *r, int idx, int c) { return r->data + ((size_t) idx * r->ch + c) * r->n; }
static void ring_incr_rd(ring *r) { if (++r->rd >= r->size) r->rd = 0; }          /* :110-114 */
static void ring_decr_rd(ring *r) { if (--r->rd < 0) r->rd = r->size - 1; }        /* :117-121 */
static void ring_incr_wr(ring *r) { if (++r->wr >= r->size) r->wr = 0; }          /* :124-128 */
static void ring_clear(ring *r, int idx) { memset(ring_at(r, idx, 0), 0, sizeof(float) * (size_t) r->ch * r->n); }

/* ------------------------------------------------------------------------- */
/* IRBaboonAudioProcessor, convolution part                                    */
/* ------------------------------------------------------------------------- */

struct fp_rt {
    int C, B, N, fftBlockSize, P, H;
    int irRingLen;
    ring input, audioFft, irFft, convResult, output, bypass;
    fp_fft *fft;
    float *inplace, *overlap;
    const float *ir; int irLen; float *irOwned;
    int inputBufferSampleIndex, outputBufferSampleIndex, outputSampleIndex;
    int blocksToProcess, blocksToOutputBuffer, savedIndex;
    int latency;
};

static int next_pow2(int x) { int p = 1; while (p < x) p *= 2; return p; }

/* ctor, Source/PluginProcessor.cpp:57-82 */
fp_rt *fp_rt_create(int channels, int block, int ir_ring_len)
{
    fp_rt *p = (fp_rt *) calloc(1, sizeof *p);
    p->C = channels; p->B = block; p->irRingLen = ir_ring_len;
    p->N = 1;
    while (p->N < block * 2 - 1) p->N *= 2;                               /* :61-64 */
    p->fftBlockSize = p->N * 2;                                           /* :65 */
    p->fft = fp_fft_create(highest_bit(p->N));                            /* :71-75 */
    p->inplace = (float *) calloc((size_t) channels * p->fftBlockSize, sizeof(float));   /* :78-79 */
    p->overlap = (float *) calloc((size_t) channels * block, sizeof(float));             /* :81-82 */
    p->latency = block;                                                   /* :59 */
    /* default IR = generatePulse(makeupIRLengthSamples, 100), :46 + fp/tools.cpp:235-241 */
    p->irOwned = (float *) calloc((size_t) ir_ring_len, sizeof(float));
    if (ir_ring_len > 100) p->irOwned[100] = 1.0f; else if (ir_ring_len > 0) p->irOwned[0] = 1.0f;
    p->ir = p->irOwned; p->irLen = ir_ring_len;
    return p;
}

void fp_rt_destroy(fp_rt *p)
{
    if (!p) return;
    free(p->input.data); free(p->audioFft.data); free(p->irFft.data);
    free(p->convResult.data); free(p->output.data); free(p->bypass.data);
    fp_fft_destroy(p->fft); free(p->inplace); free(p->overlap); free(p->irOwned); free(p);
}

/* prepareToPlay, Source/PluginProcessor.cpp:164-234 (convolution rings only).
 * The reference grows audioFftBufferArray with buffers it never clears
 * (fp/CircularBufferArray.cpp:48-53 clears the old ones); zeros are the intended
 * content and what this restatement uses. */
void fp_rt_prepareToPlay(fp_rt *p, int samplesPerBlock)
{
    p->H = samplesPerBlock;
    p->latency = (p->H > p->B) ? p->H : p->B;                             /* :169-174 */
    int inputArraySize = (int) ceilf((float) p->H / (float) p->B);        /* :203 */
    int outputArraySize = next_pow2((int) ceilf((float) p->B / (float) p->H) + 1);   /* :205 */
    ring_init(&p->input, inputArraySize, p->C, p->fftBlockSize);          /* :208 */
    ring_init(&p->convResult, inputArraySize, p->C, p->fftBlockSize);     /* :209 */
    ring_init(&p->output, outputArraySize, p->C, p->H);                   /* :210 */
    ring_init(&p->bypass, outputArraySize, p->C, p->H);                   /* :211 */
    p->output.rd = 1;                                                     /* :218 */
    p->inputBufferSampleIndex = 0;                                        /* :220 */
    p->outputBufferSampleIndex = 0;                                       /* :221 */
    p->P = (int) ceilf((float) p->irRingLen / (float) p->B);              /* :225 */
    ring_init(&p->irFft, p->P, 1, p->fftBlockSize);                       /* :226 */
    int audioSize = p->P > inputArraySize ? p->P : inputArraySize;        /* :229 */
    ring_init(&p->audioFft, audioSize, p->C, p->fftBlockSize);            /* :213-215,230 */
    p->audioFft.rd = p->audioFft.wr;                                      /* :231 */
    ring_decr_rd(&p->audioFft);                                           /* :232 */
    p->savedIndex = p->audioFft.rd;                                       /* :233 */
}

void fp_rt_set_ir(fp_rt *p, const float *ir, int len)
{
    free(p->irOwned);
    p->irOwned = (float *) malloc(sizeof(float) * (size_t) (len > 0 ? len : 1));
    memcpy(p->irOwned, ir, sizeof(float) * (size_t) len);
    p->ir = p->irOwned; p->irLen = len;
}

int fp_rt_latency(const fp_rt *p) { return p->latency; }
int fp_rt_partitions(const fp_rt *p) { return p->P; }

/* processBlock, IRCAP_IDLE branch, Source/PluginProcessor.cpp:408-562 */
void fp_rt_processBlock(fp_rt *p, float *const *buffer, int n)
{
    const int C = p->C, B = p->B, N = p->N, F = p->fftBlockSize;
    if (n == 0) return;                                                   /* :281-283 */

    for (int sample = 0; sample < n; sample++) {                          /* :421-445 */
        for (int c = 0; c < C; c++)
            ring_at(&p->input, p->input.wr, c)[p->inputBufferSampleIndex] = buffer[c][sample];
        p->inputBufferSampleIndex++;
        if (p->inputBufferSampleIndex >= B) {
            p->input.rd = p->input.wr;
            memcpy(ring_at(&p->audioFft, p->audioFft.wr, 0), ring_at(&p->input, p->input.rd, 0), sizeof(float) * (size_t) C * F);
            for (int c = 0; c < C; c++)
                fp_fft_forward_real(p->fft, ring_at(&p->audioFft, p->audioFft.wr, c));
            ring_incr_wr(&p->audioFft);
            ring_incr_wr(&p->input);
            ring_clear(&p->input, p->input.wr);
            p->inputBufferSampleIndex = 0;
            p->blocksToProcess++;
        }
    }

    while (p->blocksToProcess > 0) {                                      /* :452-518 */
        float *hw = ring_at(&p->irFft, p->irFft.wr, 0);
        memset(hw, 0, sizeof(float) * (size_t) F);                        /* :455 */
        memcpy(hw, p->ir + (size_t) p->irFft.wr * B, sizeof(float) * (size_t) B);   /* :457 */
        fp_fft_forward_real(p->fft, hw);                                  /* :459 */
        ring_incr_wr(&p->irFft);                                          /* :461 */

        ring_clear(&p->convResult, p->convResult.wr);                     /* :466 */
        p->audioFft.rd = p->savedIndex;                                   /* :468-470 */
        ring_incr_rd(&p->audioFft);
        p->savedIndex = p->audioFft.rd;

        for (int ch = 0; ch < C; ch++) {                                  /* :472-512 */
            float *ovl = p->overlap + (size_t) ch * B;
            float *conv = ring_at(&p->convResult, p->convResult.wr, ch);
            int k = 0;
            p->audioFft.rd = p->savedIndex;                               /* :478 */
            while (k < p->irFft.size) {                                   /* :480-501 */
                memcpy(p->inplace, ring_at(&p->audioFft, p->audioFft.rd, 0), sizeof(float) * (size_t) C * F);
                float *ip = p->inplace + (size_t) ch * F;
                const float *h = ring_at(&p->irFft, k, 0);                /* :485 IR channel 0 for every audio channel */
                for (int i = 0; i <= N; i += 2)
                    fp_complexMul(ip + i, ip + i + 1, h[i], h[i + 1]);
                for (int i = 0; i < F; i++)
                    conv[i] += ip[i];
                k++;
                ring_decr_rd(&p->audioFft);
            }
            fp_fft_inverse_real(p->fft, conv);                            /* :504 */
            for (int i = 0; i < B; i++) {                                 /* :507-510 */
                conv[i] += ovl[i];
                ovl[i] = conv[B + i];
            }
        }
        ring_incr_wr(&p->convResult);                                     /* :514-516 */
        p->blocksToOutputBuffer++;
        p->blocksToProcess--;
    }

    while (p->blocksToOutputBuffer > 0) {                                 /* :525-543 */
        for (int sample = 0; sample < B; sample++) {
            if (p->outputBufferSampleIndex == 0)
                ring_clear(&p->output, p->output.wr);
            for (int c = 0; c < C; c++)
                ring_at(&p->output, p->output.wr, c)[p->outputBufferSampleIndex] =
                    ring_at(&p->convResult, p->convResult.rd, c)[sample];
            p->outputBufferSampleIndex++;
            if (p->outputBufferSampleIndex >= p->H) {
                ring_incr_wr(&p->output);
                p->outputBufferSampleIndex = 0;
            }
        }
        ring_incr_rd(&p->convResult);
        p->blocksToOutputBuffer--;
    }

    for (int sample = 0; sample < n; sample++) {                          /* :550-560 */
        for (int c = 0; c < C; c++)
            buffer[c][sample] = ring_at(&p->output, p->output.rd, c)[p->outputSampleIndex];
        p->outputSampleIndex++;
        if (p->outputSampleIndex >= p->H) {
            ring_incr_rd(&p->output);
            p->outputSampleIndex = 0;
        }
    }
}

/* processBlockBypassed, Source/PluginProcessor.cpp:584-590 */
void fp_rt_processBlockBypassed(fp_rt *p, float *const *buffer, int n)
{
    if (n != p->H) return;                       /* makeCopyOf would resize the slot; only the prepared size is restated */
    for (int c = 0; c < p->C; c++)
        memcpy(ring_at(&p->bypass, p->bypass.wr, c), buffer[c], sizeof(float) * (size_t) n);      /* :586 */
    ring_incr_wr(&p->bypass);                                                                     /* :587 */
    ring_incr_rd(&p->bypass);                                                                     /* :588 */
    for (int c = 0; c < p->C; c++)
        memcpy(buffer[c], ring_at(&p->bypass, p->bypass.rd, c), sizeof(float) * (size_t) n);      /* :589 */
}
