// C++ host-mirror test: irb::fp_convolution::convolvePeriodic with a minimal planar buffer type that
// has juce::AudioBuffer<float>'s member names (this repository's own test type, JUCE is not needed).
// usage: test_host_mirror layout   -> error convention only (no GPU needed)
//        test_host_mirror gpu      -> delta-at-100 IR must give a pure 100-sample delay
#include "../../include/irb_fp_convolution.hpp"
#include "../../include/irb_rt_adapter.hpp"

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

struct PlanarBuffer {
    PlanarBuffer(int ch, int n) : nch(ch), ns(n), d((size_t) ch * (size_t) (n > 0 ? n : 0)) {}
    void clear() { std::fill(d.begin(), d.end(), 0.0f); }
    int getNumChannels() const { return nch; }
    int getNumSamples() const { return ns; }
    const float *getReadPointer(int c) const { return d.data() + (size_t) c * ns; }
    float *getWritePointer(int c) { return d.data() + (size_t) c * ns; }
    int nch, ns;
    std::vector<float> d;
};

int main(int argc, char **argv)
{
    const bool gpu = argc > 1 && std::strcmp(argv[1], "gpu") == 0;
    {   // unknown layout: cleared buffer of the right shape (fp/convolution.cpp:20-21,39-42)
        PlanarBuffer a(3, 64), h(1, 16);
        for (auto &v : a.d) v = 1.0f;
        for (auto &v : h.d) v = 1.0f;
        PlanarBuffer y = irb::fp_convolution::convolvePeriodic(a, h, 16);
        if (y.getNumChannels() != 3 || y.getNumSamples() != 79) return 1;
        for (float v : y.d) if (v != 0.0f) return 2;
    }
    {   // an adapter whose create failed (0 channels: no GPU needed to fail): the failure is reported ONCE, by the constructor;
        // audio callbacks then clear their buffer and print nothing (this is the audio thread)
        irb::RtConvolverT<PlanarBuffer> bad(0, 0, 256, 2048);
        if (bad.ok()) return 7;
        bad.prepareToPlay(256);
        PlanarBuffer blk(2, 256);
        for (int k = 0; k < 5; k++) {
            for (auto &v : blk.d) v = 1.0f;
            bad.processBlock(blk);
            for (float v : blk.d) if (v != 0.0f) return 8;
        }
        bad.processBlockBypassed(blk);
        if (bad.failedCallbacks() != 6) return 9;
    }
    if (!gpu) { std::puts("layout ok"); return 0; }
    PlanarBuffer a(2, 4096), h(1, 2048);
    a.clear(); h.clear();
    uint64_t s = 12345;
    for (auto &v : a.d) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = (float) ((int64_t) (s >> 40) - (1 << 23)) / (float) (1 << 23); }
    h.getWritePointer(0)[100] = 1.0f;               // tools::generatePulse(2048, 100), Source/PluginProcessor.cpp:46
    PlanarBuffer y = irb::fp_convolution::convolvePeriodic(a, h);   // default processBlockSize = 256
    if (y.getNumSamples() != 4096 + 2048 - 1) return 3;
    double worst = 0;
    for (int c = 0; c < 2; c++)
        for (int i = 0; i < 4096; i++) worst = std::fmax(worst, std::fabs(y.getReadPointer(c)[i + 100] - a.getReadPointer(c)[i]));
    std::printf("delta delay max err %.3g\n", worst);
    if (!(worst < 3e-6)) return 4;

    // processBlock-shaped adapter with the plug-in's constants: stereo, 256-sample blocks, 2048-sample IR ring,
    // default IR = delta at 100 -> output = input delayed by 100 + one block (Source/PluginProcessor.cpp:46,169-174)
    irb::RtConvolverT<PlanarBuffer> rt(0, 2, 256, 2048);
    if (rt.prepareToPlay(256) != 256 || rt.partitions() != 8) return 5;
    const int nblk = 24, H = 256;
    PlanarBuffer in(2, nblk * H), outb(2, nblk * H);
    for (auto &v : in.d) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = (float) ((int64_t) (s >> 40) - (1 << 23)) / (float) (1 << 23); }
    for (int b = 0; b < nblk; b++) {
        PlanarBuffer blk(2, H);
        for (int c = 0; c < 2; c++) std::memcpy(blk.getWritePointer(c), in.getReadPointer(c) + b * H, sizeof(float) * H);
        rt.processBlock(blk);
        for (int c = 0; c < 2; c++) std::memcpy(outb.getWritePointer(c) + b * H, blk.getReadPointer(c), sizeof(float) * H);
    }
    worst = 0;
    const int d = 256 + 100;
    for (int c = 0; c < 2; c++)
        for (int i = d; i < nblk * H; i++) worst = std::fmax(worst, std::fabs(outb.getReadPointer(c)[i] - in.getReadPointer(c)[i - d]));
    std::printf("rt delta delay max err %.3g\n", worst);
    rt.releaseResources();
    return worst < 3e-6 ? 0 : 6;
}
