"""The oracle's C code under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only; GPU sanitizers are
not available on the pool).  Builds a small driver that runs the edge cases through every oracle entry."""
import os
import subprocess
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = textwrap.dedent(r"""
    #include "fp_oracle.h"
    #include <stdio.h>
    #include <stdlib.h>
    static float *noise(int n, unsigned s) { float *p = calloc(n > 0 ? n : 1, sizeof(float)); for (int i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; p[i] = (float) (s >> 8) / 8388608.0f - 1.0f; } return p; }
    int main(void) {
        int cases[][3] = {{1000, 300, 128}, {100, 300, 128}, {0, 10, 16}, {5000, 4096, 512}, {1000, 250, 100}, {257, 1, 256}, {1, 1, 1}, {64, 0, 16}};
        for (unsigned c = 0; c < sizeof cases / sizeof cases[0]; c++) {
            int La = cases[c][0], Lh = cases[c][1], B = cases[c][2];
            for (int ac = 1; ac <= 3; ac++) for (int ic = 1; ic <= 2; ic++) {
                float *a[3], *h[2], *o[3];
                for (int i = 0; i < ac; i++) { a[i] = noise(La, 7 + i); o[i] = calloc(La + Lh > 0 ? La + Lh : 1, sizeof(float)); }
                for (int i = 0; i < ic; i++) h[i] = noise(Lh, 70 + i);
                fp_convolvePeriodic((const float *const *) a, ac, La, (const float *const *) h, ic, Lh, B, o);
                if (La > 0 && Lh > 0) fp_convolveNonPeriodic((const float *const *) a, ac, La, (const float *const *) h, ic, Lh, o);
                for (int i = 0; i < ac; i++) { free(a[i]); free(o[i]); }
                for (int i = 0; i < ic; i++) free(h[i]);
            }
        }
        int hosts[] = {1, 64, 100, 256, 300, 512};
        for (unsigned k = 0; k < sizeof hosts / sizeof hosts[0]; k++) {
            fp_rt *p = fp_rt_create(2, 256, 2048);
            fp_rt_prepareToPlay(p, hosts[k]);
            float *ir = noise(2048, 3); fp_rt_set_ir(p, ir, 2048); free(ir);
            for (int it = 0; it < 40; it++) { float *b0 = noise(hosts[k], it), *b1 = noise(hosts[k], 99 + it); float *bb[2] = {b0, b1}; fp_rt_processBlock(p, bb, hosts[k]); free(b0); free(b1); }
            fp_rt_destroy(p);
        }
        int lens[] = {17, 100, 1024, 5000};
        for (unsigned k = 0; k < sizeof lens / sizeof lens[0]; k++) {
            int n = lens[k], N = fp_nextPowerOfTwo(n);
            float *x = noise(n, 5), *y = noise(n / 2 + 1, 6), *o = calloc(N, sizeof(float));
            for (int f = 0; f < 8; f++) fp_deconvolve(x, n, y, n / 2 + 1, 48000.0, f & 1, (f >> 1) & 1, (f >> 2) & 1, o);
            fp_invertFilter(x, n, 48000, o);
            fp_shifteroo(o, N); fp_shifteroo(o, N - 1);
            free(x); free(y); free(o);
        }
        puts("sanitizers ok");
        return 0;
    }
""")


def test_oracle_under_asan_ubsan(tmp_path):
    src = tmp_path / "driver.c"
    src.write_text(DRIVER)
    exe = tmp_path / "driver"
    subprocess.check_call(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-I", os.path.join(ROOT, "oracle"), "-o", str(exe), str(src),
                           os.path.join(ROOT, "oracle", "fp_oracle.c"), "-lm"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "sanitizers ok" in r.stdout
