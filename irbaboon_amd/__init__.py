"""irbaboon_amd: MI355X-native uniformly partitioned convolution (the fp::convolution hot path
of Flixor/IRBaboon) behind the C ABI of include/irb_fdl.h.  Python here is only the host-side
mirror of the reference interface plus ctypes plumbing; the compute is HIP (csrc/)."""
from ._lib import IrbError, lib                     # noqa: F401
from .convolution import ChannelLayout, IRtoRealFFTRaw, convolveNonPeriodic, convolvePeriodic   # noqa: F401
from .engine import FdlEngine                       # noqa: F401
from .rt import RtConvolver                         # noqa: F401
from .deconvolution import (LargeFft, averagingFilter, deconvolve, fftInvTransform,   # noqa: F401,E402
                            fftTransform, invertFilter, nextPowerOfTwo, shifteroo)
from .sweep import ExpSineSweep   # noqa: F401,E402
