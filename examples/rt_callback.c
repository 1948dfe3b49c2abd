/* rt_callback.c -- plain C11 host of the processBlock-shaped adapter (include/irb_fdl.h, irb_rt_*): what a plug-in host's
 * audio thread does, without JUCE.  Stereo, 256-sample processing block, 2048-sample IR ring (8 partitions); host blocks of
 * 256 then 512 samples; the default IR (unit delta at sample 100) must come back as a pure delay of latency + 100 samples.
 * Prints the callback times the library measured itself and what each callback asked of the HIP runtime.
 *   gcc -std=c11 -O2 -Iinclude examples/rt_callback.c -Lirbaboon_amd -l:libirbfdl.so -Wl,-rpath,$PWD/irbaboon_amd -lm */
#include "irb_fdl.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

static float lcg(unsigned *s) { *s = *s * 1664525u + 1013904223u; return (float) (*s >> 8) / 8388608.0f - 1.0f; }
static int cmp(const void *a, const void *b) { const float x = *(const float *) a, y = *(const float *) b; return (x > y) - (x < y); }

int main(void)
{
    enum { C = 2, B = 256, RING = 2048, NCB = 2000 };
    irb_rt *p = NULL;
    if (irb_rt_create(0, C, B, RING, &p) != IRB_OK) { fprintf(stderr, "create: %s\n", irb_last_error()); return 1; }
    int rc = 0;
    for (int hb = 256; hb <= 512 && rc == 0; hb *= 2) {
        if (irb_rt_prepare_to_play(p, hb) != IRB_OK) { fprintf(stderr, "prepare: %s\n", irb_last_error()); return 1; }
        const int lat = irb_rt_latency(p), n = hb * 40;
        float *x[C], *y[C], *buf[C];
        unsigned seed = 7;
        for (int c = 0; c < C; c++) {
            x[c] = malloc(sizeof(float) * n); y[c] = malloc(sizeof(float) * n);
            for (int i = 0; i < n; i++) x[c][i] = lcg(&seed);
        }
        for (int i = 0; i < n; i += hb) {                      /* the audio callback: in place, hb samples per channel */
            for (int c = 0; c < C; c++) { buf[c] = y[c] + i; for (int k = 0; k < hb; k++) buf[c][k] = x[c][i + k]; }
            if (irb_rt_process_block(p, buf, hb) != IRB_OK) { fprintf(stderr, "process_block: %s\n", irb_last_error()); return 1; }
        }
        double worst = 0;                                      /* default IR = delta at 100: output = input delayed by latency + 100 */
        const int d = lat + 100, from = d + 9 * B;             /* the rolling refresh has loaded every partition after 8 blocks */
        for (int c = 0; c < C; c++) for (int i = from; i < n; i++) worst = fmax(worst, fabs(y[c][i] - x[c][i - d]));
        float *us = malloc(sizeof(float) * NCB);
        unsigned long long cb0, l0, w0, k0, cb1, l1, w1, k1;
        irb_rt_time_callbacks(p, hb, 200, us);                 /* warm-up */
        irb_rt_counters(p, &cb0, &l0, &w0, &k0);
        if (irb_rt_time_callbacks(p, hb, NCB, us) != IRB_OK) { fprintf(stderr, "time_callbacks: %s\n", irb_last_error()); return 1; }
        irb_rt_counters(p, &cb1, &l1, &w1, &k1);
        qsort(us, NCB, sizeof(float), cmp);
        printf("host block %d: latency %d, delta-delay max err %.3g; callback median %.1f us, p99 %.1f us; per callback %.2f launches, %.2f waits, %.2f copies\n",
               hb, lat, worst, us[NCB / 2], us[NCB * 99 / 100], (double) (l1 - l0) / (cb1 - cb0), (double) (w1 - w0) / (cb1 - cb0),
               (double) (k1 - k0) / (cb1 - cb0));
        if (!(worst < 3e-6)) rc = 2;
        for (int c = 0; c < C; c++) { free(x[c]); free(y[c]); }
        free(us);
    }
    irb_rt_destroy(p);
    return rc;
}
