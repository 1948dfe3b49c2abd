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
    // false after a failed create / prepareToPlay: every callback then clears its buffer (the reference's error convention)
    // WITHOUT printing -- the failure was reported once, where it happened, on the message thread
    bool ok() const { return rt != nullptr && prepared; }
    ~RtConvolverT() { irb_rt_destroy(rt); }
    RtConvolverT(const RtConvolverT &) = delete;
    RtConvolverT &operator=(const RtConvolverT &) = delete;

    // prepareToPlay (:164-234); returns the latency to pass to setLatencySamples (:169-174)
    int prepareToPlay(int samplesPerBlock)
    {
        prepared = rt != nullptr && irb_rt_prepare_to_play(rt, samplesPerBlock) == IRB_OK;
        if (rt != nullptr && !prepared) report();          // (a failed create has been reported by the constructor)
        failed_callbacks = 0;
        return irb_rt_latency(rt);
    }
    void releaseResources() { if (rt) irb_rt_release_resources(rt); prepared = false; }     // :240-248

    // IRtoConvolve = playFiltered ? IRFilt : IRpulse (:411-414): call when the selection changes
    void setIR(const Buffer &ir)
    {
        if (rt && irb_rt_set_ir(rt, ir.getReadPointer(0), ir.getNumSamples()) != IRB_OK) report();
    }

    // the IRCAP_IDLE branch of processBlock (:408-562), in place; on failure the buffer is cleared, which is the
    // reference's error convention.  Output gain and limiter (:567-574) stay with the caller.  This is the audio thread:
    // nothing is printed here after the first failure of a run of failures (failedCallbacks() counts them).
    void processBlock(Buffer &buffer)
    {
        if (!ok()) { buffer.clear(); ++failed_callbacks; return; }
        for (int c = 0; c < nch; ++c) ptrs[(size_t) c] = buffer.getWritePointer(c);
        if (irb_rt_process_block(rt, ptrs.data(), buffer.getNumSamples()) != IRB_OK) {
            if (failed_callbacks++ == 0) report();
            buffer.clear();
        }
    }

    // processBlockBypassed (:584-590)
    void processBlockBypassed(Buffer &buffer)
    {
        if (!ok()) { ++failed_callbacks; return; }                              // bypass = the dry signal, untouched
        for (int c = 0; c < nch; ++c) ptrs[(size_t) c] = buffer.getWritePointer(c);
        if (irb_rt_process_block_bypassed(rt, ptrs.data(), buffer.getNumSamples()) != IRB_OK && failed_callbacks++ == 0) report();
    }

    unsigned long long failedCallbacks() const { return failed_callbacks; }
    int latency() const { return irb_rt_latency(rt); }
    int partitions() const { return irb_rt_partitions(rt); }

private:
    static void report() { std::fprintf(stderr, "irb_rt: %s\n", irb_last_error()); }
    irb_rt *rt = nullptr;
    bool prepared = false;
    unsigned long long failed_callbacks = 0;
    int nch;
    std::vector<float *> ptrs;          // one pointer per channel of the processor (every channel, not the first 16)
};

}  // namespace irb
