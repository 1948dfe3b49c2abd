"""Regenerates tests/golden/convolve_cases.npz.

The reference ships no golden vectors and cannot be built in this image (JUCE/Boost absent),
so the expected outputs here are (1) float64 direct convolution of bit-portable seeded inputs
-- the mathematical definition of fp::convolution::convolvePeriodic's result -- and (2) the
number of trailing samples the reference leaves at zero (its dropped-tail quirk,
fp/convolution.cpp:233-238), which SURVEY.md 8a-1 records from a probe of the real reference:
La=10000,Lh=4096,B=256 -> 15 samples; La=5000,Lh=4096,B=512 -> 391 samples.
Inputs are stored too, so the fixture does not depend on regenerating them.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from irbaboon_amd.synth import decaying_ir, delta_ir, white_noise  # noqa: E402
from util import direct_f64  # noqa: E402

# name: (audio_ch, La, ir_ch, Lh, B, ir_kind)
CASES = {
    "mono_mono_b256": (1, 10000, 1, 4096, 256, "decay"),
    "mono_mono_b512_tail": (1, 5000, 1, 4096, 512, "decay"),
    "stereo_stereo_b128": (2, 3000, 2, 700, 128, "decay"),
    "irmono_astereo_b128": (2, 3000, 1, 700, 128, "decay"),
    "irstereo_amono_b128": (1, 3000, 2, 700, 128, "decay"),
    "short_ir_b256": (1, 2560, 1, 100, 256, "decay"),
    "audio_shorter_than_block": (1, 100, 1, 300, 128, "decay"),
    "nonpow2_block_100": (1, 1000, 1, 250, 100, "decay"),
    "delta100_b256": (2, 2048, 1, 2048, 256, "delta"),
    "exact_blocks": (1, 2560, 1, 1024, 256, "decay"),
}


def main():
    out = {}
    for i, (name, (ac, La, ic, Lh, B, kind)) in enumerate(sorted(CASES.items())):
        audio = np.stack([white_noise(1000 * (i + 1) + c, La) for c in range(ac)])
        if kind == "delta":
            ir = np.stack([delta_ir(Lh, 100) for _ in range(ic)])
        else:
            ir = np.stack([decaying_ir(9000 * (i + 1) + c, Lh) for c in range(ic)])
        out[name + "/audio"] = audio
        out[name + "/ir"] = ir
        out[name + "/block"] = np.int32(B)
        out[name + "/expected_f64"] = direct_f64(audio, ir)
    np.savez_compressed(os.path.join(HERE, "convolve_cases.npz"), **out)
    print("wrote", len(CASES), "cases")


if __name__ == "__main__":
    main()
