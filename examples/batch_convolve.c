/* batch_convolve.c -- plain C11 user of the C ABI (include/irb_fdl.h): N independent mono instances, block 256,
 * each with its own decaying-noise IR, streamed block by block; prints a checksum and the delta-IR sanity check.
 *   gcc -std=c11 -O2 -Iinclude examples/batch_convolve.c -Lirbaboon_amd -l:libirbfdl.so -Wl,-rpath,$PWD/irbaboon_amd -lm */
#include "irb_fdl.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

static float lcg(unsigned *s) { *s = *s * 1664525u + 1013904223u; return (float) (*s >> 8) / 8388608.0f - 1.0f; }

int main(void)
{
    enum { I = 8, B = 256, LH = 4096, T = 16 };
    irb_fdl_config cfg = {0};
    cfg.device = 0; cfg.n_instances = I; cfg.channels = 1; cfg.block = B; cfg.max_ir_len = LH; cfg.max_blocks_per_call = 4;
    irb_fdl *h = NULL;
    if (irb_fdl_create(&cfg, &h) != IRB_OK) { fprintf(stderr, "create: %s\n", irb_last_error()); return 1; }

    unsigned seed = 1;
    float *ir = malloc(sizeof(float) * LH), *x[I], *y[I];
    for (int i = 0; i < I; i++) {
        if (i == 0) { for (int n = 0; n < LH; n++) ir[n] = 0.0f; ir[100] = 1.0f; }        /* the plug-in's default IR */
        else for (int n = 0; n < LH; n++) ir[n] = lcg(&seed) * expf(-6.9f * n / LH) * 0.05f;
        const float *irp[1] = {ir};
        if (irb_fdl_set_ir(h, i, irp, 1, LH) != IRB_OK) { fprintf(stderr, "set_ir: %s\n", irb_last_error()); return 1; }
        x[i] = malloc(sizeof(float) * T * B);
        y[i] = malloc(sizeof(float) * T * B);
        for (int n = 0; n < T * B; n++) x[i][n] = lcg(&seed);
    }
    if (irb_fdl_process(h, (const float *const *) x, y, T) != IRB_OK) { fprintf(stderr, "process: %s\n", irb_last_error()); return 1; }

    double worst = 0, sum = 0;
    for (int n = 100; n < T * B; n++) worst = fmax(worst, fabs(y[0][n] - x[0][n - 100]));   /* delta at 100 = pure delay */
    for (int i = 0; i < I; i++) for (int n = 0; n < T * B; n++) sum += (double) y[i][n] * y[i][n];
    irb_fdl_info info;
    irb_fdl_get_info(h, &info);
    printf("N=%d P=%d streams=%d  delta-delay max err %.3g  energy %.6f\n", info.fft_len, info.partitions, info.streams, worst, sum);
    irb_fdl_destroy(h);
    for (int i = 0; i < I; i++) { free(x[i]); free(y[i]); }
    free(ir);
    return worst < 3e-6 ? 0 : 2;
}
