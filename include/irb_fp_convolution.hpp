// irb_fp_convolution.hpp -- C++ host-side mirror of the reference's fp::convolution::convolvePeriodic
// (fp/convolution.hpp:31, fp/convolution.cpp:14-242) on top of the C ABI (irb_fdl.h).
//
// Header only and templated on the buffer type, so it needs no JUCE here and compiles against
// juce::AudioBuffer<float> unchanged: the buffer type must offer the AudioBuffer members the
// reference function itself uses -- Buffer(int channels, int samples), clear(), getNumChannels(),
// getNumSamples(), getReadPointer(ch), getWritePointer(ch).  Same argument meaning (buffer1 = audio,
// buffer2 = IR), same result shape (channels of buffer1, La+Lh-1 samples) and same error behaviour
// (unknown channel layout: a debug line and a cleared buffer, fp/convolution.cpp:39-42).
#pragma once
#include "irb_fdl.h"

#include <cstdio>
#include <vector>

namespace irb {
namespace fp_convolution {

struct Options {
    int  device = 0;
    bool keep_reference_tail = true;   // fp/convolution.cpp:236-238: the last partial overlap is dropped
};

template <class Buffer>
Buffer convolvePeriodic(Buffer &buffer1, Buffer &buffer2, int processBlockSize = 256, Options opt = Options())
{
    const int ca = buffer1.getNumChannels(), ci = buffer2.getNumChannels();
    const int La = buffer1.getNumSamples(), Lh = buffer2.getNumSamples();
    Buffer outputBuffer(ca, La + Lh - 1 > 0 ? La + Lh - 1 : 0);
    outputBuffer.clear();
    std::vector<const float *> a(ca), h(ci);
    std::vector<float *> o(ca);
    for (int c = 0; c < ca; c++) { a[c] = buffer1.getReadPointer(c); o[c] = outputBuffer.getWritePointer(c); }
    for (int c = 0; c < ci; c++) h[c] = buffer2.getReadPointer(c);
    const int rc = irb_convolve_periodic(opt.device, a.data(), ca, La, h.data(), ci, Lh, processBlockSize,
                                         opt.keep_reference_tail ? 1 : 0, o.data());
    if (rc == IRB_ERR_LAYOUT)
        std::fprintf(stderr, "Either buffer1 or buffer2 is not mono nor stereo. Abort abort \n");
    else if (rc != IRB_OK)
        std::fprintf(stderr, "irb_convolve_periodic failed (%d): %s\n", rc, irb_last_error());
    return outputBuffer;   // cleared on any failure, like the reference's early return
}

}  // namespace fp_convolution
}  // namespace irb
