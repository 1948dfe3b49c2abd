// IRBaboonGpuConvolver.h -- SOURCE-ONLY adapter for a JUCE build of IRBaboon (JUCE is not in this
// repository's image, so this file is not compiled by the tests; the same logic is exercised
// JUCE-free through irb_rt_* in tests/test_gpu_parity.py).
//
// Drop-in for the convolution part of IRBaboonAudioProcessor: keep the juce::AudioProcessor surface
// (prepareToPlay / processBlock / releaseResources, Source/PluginProcessor.h:73-81) and replace the
// IRCAP_IDLE branch (Source/PluginProcessor.cpp:408-562) by one call.  See INTEGRATION.md.
#pragma once
#include <JuceHeader.h>
#include "irb_fdl.h"

class IRBaboonGpuConvolver
{
public:
    IRBaboonGpuConvolver (int device, int channels, int processBlockSize, int makeupIRLengthSamples)
    {
        if (irb_rt_create (device, channels, processBlockSize, makeupIRLengthSamples, &rt) != IRB_OK)
            DBG (irb_last_error());
    }
    ~IRBaboonGpuConvolver()                       { irb_rt_destroy (rt); }

    // prepareToPlay (Source/PluginProcessor.cpp:164-234); returns the latency to report (:169-174)
    int prepareToPlay (int samplesPerBlock)
    {
        if (irb_rt_prepare_to_play (rt, samplesPerBlock) != IRB_OK) DBG (irb_last_error());
        return irb_rt_latency (rt);
    }
    void releaseResources()                       { irb_rt_release_resources (rt); }      // :240-248

    // IRtoConvolve = playFiltered ? IRFilt : IRpulse (:411-414); call when the selection changes
    void setIR (const juce::AudioBuffer<float>& ir) { irb_rt_set_ir (rt, ir.getReadPointer (0), ir.getNumSamples()); }

    // the IRCAP_IDLE branch of processBlock (:408-562); gain and limiter (:567-574) stay in the caller
    void processBlock (juce::AudioBuffer<float>& buffer, int channels)
    {
        float* ptrs[8];
        for (int c = 0; c < channels; ++c) ptrs[c] = buffer.getWritePointer (c);
        if (irb_rt_process_block (rt, ptrs, buffer.getNumSamples()) != IRB_OK)
        {
            DBG (irb_last_error());
            buffer.clear();                       // reference error convention: cleared buffer
        }
    }

private:
    irb_rt* rt = nullptr;
    JUCE_DECLARE_NON_COPYABLE_WITH_LEAK_DETECTOR (IRBaboonGpuConvolver)
};
