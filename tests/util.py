import numpy as np


def direct_f64(audio, ir):
    """float64 direct linear convolution with the reference's channel pairing
    (fp/convolution.cpp:176-181): IR channel c for (mono,mono)/(stereo,stereo), else 0;
    stereo IR + mono audio uses (L+R)/2 (fp/tools.cpp:14-30)."""
    a = np.atleast_2d(audio).astype(np.float64)
    h = np.atleast_2d(ir).astype(np.float64)
    if h.shape[0] == 2 and a.shape[0] == 1:
        h = ((h[0] + h[1]) / 2.0)[None]
    out = []
    for c in range(a.shape[0]):
        hc = h[c] if h.shape[0] == a.shape[0] else h[0]
        out.append(np.convolve(a[c], hc) if a.shape[1] and hc.size else np.zeros(max(a.shape[1] + hc.size - 1, 0)))
    return np.stack(out)


def rel_rms(y, ref):
    y = np.asarray(y, np.float64)
    ref = np.asarray(ref, np.float64)
    den = np.sqrt(np.mean(ref ** 2))
    return float(np.sqrt(np.mean((y - ref) ** 2)) / den) if den > 0 else float(np.sqrt(np.mean((y - ref) ** 2)))


def written_samples(La, Lh, B):
    """Samples convolvePeriodic actually writes: (floor(La/B)+P)*B clipped to La+Lh-1
    (fp/convolution.cpp:233-238)."""
    P = int(np.ceil(np.float32(Lh) / np.float32(B)))
    return min((La // B + P) * B, La + Lh - 1)


# float tolerance of the north star: 1e-5 relative RMS against the reference CPU convolution
RMS_TOL = 1e-5
