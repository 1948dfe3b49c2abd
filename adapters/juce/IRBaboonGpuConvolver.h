// IRBaboonGpuConvolver.h -- the JUCE-typed instantiation of irb::RtConvolverT for IRBaboon's
// IRBaboonAudioProcessor (Source/PluginProcessor.{h,cpp}).  JUCE is not part of this repository's image, so
// this file is source only; the template itself is compiled and run JUCE-free by tests/cpp/test_host_mirror.cpp.
// How it replaces the IRCAP_IDLE branch of processBlock: INTEGRATION.md section 2.
#pragma once
#include <JuceHeader.h>
#include "irb_rt_adapter.hpp"

using IRBaboonGpuConvolver = irb::RtConvolverT<juce::AudioBuffer<float>>;

// In IRBaboonAudioProcessor:
//   std::unique_ptr<IRBaboonGpuConvolver> gpuConvolver;                                              // PluginProcessor.h:194-217
//   gpuConvolver = std::make_unique<IRBaboonGpuConvolver> (0, generalInputAudioChannels,            // ctor :57-82
//                                                           processBlockSize, makeupIRLengthSamples);
//   setLatencySamples (gpuConvolver->prepareToPlay (samplesPerBlock));                                // prepareToPlay :164-234
//   gpuConvolver->releaseResources();                                                                 // releaseResources :240-248
//   if (IRCapture.state == IRCAP_IDLE) { gpuConvolver->setIR (*IRtoConvolve); gpuConvolver->processBlock (buffer); }   // :408-562
