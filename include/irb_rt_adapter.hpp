// irb_rt_adapter.hpp -- C++ wrapper of irb_rt_* shaped like the convolution part of IRBaboonAudioProcessor
// (reference Source/PluginProcessor.{h,cpp}): prepareToPlay / processBlock / releaseResources on a planar
// buffer type.  Templated on the buffer type so that it compiles without JUCE (tests use a minimal planar
// buffer with juce::AudioBuffer<float>'s member names) and against juce::AudioBuffer<float> unchanged
// (adapters/juce/IRBaboonGpuConvolver.h).  Buffer members used: getNumSamples(), getWritePointer(ch),
// getReadPointer(ch), clear().
#pragma once
#include "irb_fdl.h"

#include <cstdio>
#include <vector>

namespace irb {

template <class Buffer>
class RtConvolverT
{
public:
    // ctor (Source/PluginProcessor.cpp:57-82): channels = generalInputAudioChannels, block = processBlockSize,
    // irRingLen = makeupIRLengthSamples (fixes irPartitions, :225); default IR = unit delta at sample 100 (:46)
    RtConvolverT(int device, int channels, int processBlockSize, int makeupIRLengthSamples)
        : nch(channels), ptrs((size_t) (channels > 0 ? channels : 0), nullptr)     // sized once: processBlock never allocates
    {
        if (irb_rt_create(device, channels, processBlockSize, makeupIRLengthSamples, &rt) != IRB_OK) report();
    }
    ~RtConvolverT() { irb_rt_destroy(rt); }
    RtConvolverT(const RtConvolverT &) = delete;
    RtConvolverT &operator=(const RtConvolverT &) = delete;

    // prepareToPlay (:164-234); returns the latency to pass to setLatencySamples (:169-174)
    int prepareToPlay(int samplesPerBlock)
    {
        if (irb_rt_prepare_to_play(rt, samplesPerBlock) != IRB_OK) report();
        return irb_rt_latency(rt);
    }
    void releaseResources() { irb_rt_release_resources(rt); }                 // :240-248

    // IRtoConvolve = playFiltered ? IRFilt : IRpulse (:411-414): call when the selection changes
    void setIR(const Buffer &ir)
    {
        if (irb_rt_set_ir(rt, ir.getReadPointer(0), ir.getNumSamples()) != IRB_OK) report();
    }

    // the IRCAP_IDLE branch of processBlock (:408-562), in place; on failure the buffer is cleared, which is the
    // reference's error convention.  Output gain and limiter (:567-574) stay with the caller.
    void processBlock(Buffer &buffer)
    {
        for (int c = 0; c < nch; ++c) ptrs[(size_t) c] = buffer.getWritePointer(c);
        if (irb_rt_process_block(rt, ptrs.data(), buffer.getNumSamples()) != IRB_OK) { report(); buffer.clear(); }
    }

    // processBlockBypassed (:584-590)
    void processBlockBypassed(Buffer &buffer)
    {
        for (int c = 0; c < nch; ++c) ptrs[(size_t) c] = buffer.getWritePointer(c);
        if (irb_rt_process_block_bypassed(rt, ptrs.data(), buffer.getNumSamples()) != IRB_OK) report();
    }

    int latency() const { return irb_rt_latency(rt); }
    int partitions() const { return irb_rt_partitions(rt); }

private:
    static void report() { std::fprintf(stderr, "irb_rt: %s\n", irb_last_error()); }
    irb_rt *rt = nullptr;
    int nch;
    std::vector<float *> ptrs;          // one pointer per channel of the processor (every channel, not the first 16)
};

}  // namespace irb
