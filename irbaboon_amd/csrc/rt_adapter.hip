// rt_adapter.hip -- processBlock-shaped host adapter (irb_rt_*) on top of the batched FDL engine.
//
// Mirrors the convolution part of IRBaboonAudioProcessor (reference Source/PluginProcessor.cpp):
// ctor :57-82, prepareToPlay :164-234, processBlock IRCAP_IDLE branch :408-562.  The host keeps
// exactly the reference's re-blocking state (input block buffer, output ring of host-block sized
// buffers with the read index started at 1, sample counters); the per-block convolution step
// (:452-518) runs on the GPU through irb_fdl_*.  Host-only C++; compiled into libirbfdl.so.
//
// Deliberate differences from the reference (SURVEY.md section 3.2 "quirks"):
//  * The FDL ring always holds P distinct past blocks.  The reference sizes it
//    max(P, ceil(hostBlock/256)) (:229-230) and writes all new spectra before processing any
//    (:421-445), so with hostBlock > 256 its last partitions see a FUTURE block.
//  * prepare_to_play always starts from a cleared FDL/overlap (the reference clears only when the
//    ring grows, :213-215) and resets outputSampleIndex when it would index past the new host block.
//  * irb_rt_set_ir checks the IR length instead of reading out of bounds (:457).
#include "../../include/irb_fdl.h"
#include "irb_env.hpp"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <new>
#include <vector>

struct irb_rt {
    int device = 0, C = 0, B = 0, irRingLen = 0, P = 0, H = 0, latency = 0;
    irb_fdl *eng = nullptr;
    std::vector<float> ir;                    // IRtoConvolve (:411-414)
    std::vector<float> inBlock;               // [C][B] block being filled (:421-426)
    std::vector<float> ready;                 // [C][maxBlocks*B] complete blocks of this call, stream-major
    std::vector<float> conv;                  // [C][maxBlocks*B] the processed blocks
    int maxBlocks = 1;                        // most complete blocks one callback can hold: (H + B - 1) / B + 1
    unsigned long long callbacks = 0;
    std::vector<float> outRing;               // [outSize][C][H] (:210)
    std::vector<const float *> inp;           // per-channel pointers handed to the engine (sized at create)
    std::vector<float *> outp;
    std::vector<float *> sub;                 // channel pointers of a piece of an oversized host block
    int outSize = 0, outRd = 0, outWr = 0;
    std::vector<float> bypassRing;            // [outSize][C][H] (:211)
    int bypRd = 0, bypWr = 0;
    int inputBufferSampleIndex = 0, outputBufferSampleIndex = 0, outputSampleIndex = 0;
    int irWriteIndex = 0;                     // irFftBufferArray write index (:457-461)
};

namespace {

int next_pow2(int x) { int p = 1; while (p < x) p *= 2; return p; }

}  // namespace

extern "C" int irb_set_error_(int code, const char *msg);   // fdl_engine.hip: sets irb_last_error(), returns code

extern "C" {

int irb_rt_create(int device, int channels, int block, int ir_ring_len, irb_rt **out)
{
    if (!out) return irb_set_error_(IRB_ERR_ARG, "null argument");
    *out = nullptr;
    if (channels < 1 || block < 1 || ir_ring_len < 1) return irb_set_error_(IRB_ERR_ARG, "channels, block, ir_ring_len must be >= 1");
    irb_rt *p = new (std::nothrow) irb_rt;
    if (!p) return irb_set_error_(IRB_ERR_ARG, "out of host memory");
    p->device = device; p->C = channels; p->B = block; p->irRingLen = ir_ring_len;
    p->latency = block;                                                         // :59
    p->P = (int) std::ceil((float) ir_ring_len / (float) block);                // :225
    // default IR = tools::generatePulse(makeupIRLengthSamples, 100) (:46, fp/tools.cpp:235-241)
    p->ir.assign((size_t) std::max(ir_ring_len, p->P * block), 0.0f);
    p->ir[ir_ring_len > 100 ? 100 : 0] = 1.0f;
    p->inBlock.assign((size_t) channels * block, 0.0f);
    p->inp.assign((size_t) channels, nullptr);
    p->outp.assign((size_t) channels, nullptr);
    p->sub.assign((size_t) channels, nullptr);
    *out = p;
    return IRB_OK;
}

void irb_rt_destroy(irb_rt *p)
{
    if (!p) return;
    irb_fdl_destroy(p->eng);
    delete p;
}

int irb_rt_release_resources(irb_rt *p)                                         // :240-248
{
    if (!p) return irb_set_error_(IRB_ERR_ARG, "null handle");
    irb_fdl_destroy(p->eng);
    p->eng = nullptr;
    p->outRing.clear(); p->outSize = 0;
    return IRB_OK;
}

int irb_rt_prepare_to_play(irb_rt *p, int samples_per_block)                    // :164-234
{
    if (!p) return irb_set_error_(IRB_ERR_ARG, "null handle");
    if (samples_per_block < 1) return irb_set_error_(IRB_ERR_ARG, "samples_per_block must be >= 1");
    p->H = samples_per_block;
    p->latency = std::max(p->H, p->B);                                          // :169-174
    p->outSize = next_pow2((int) std::ceil((float) p->B / (float) p->H) + 1);   // :205
    p->outRing.assign((size_t) p->outSize * p->C * p->H, 0.0f);                 // :210
    p->outRd = 1; p->outWr = 0;                                                 // :218
    p->bypassRing.assign((size_t) p->outSize * p->C * p->H, 0.0f);              // :211
    p->bypRd = 0; p->bypWr = 0;
    p->inputBufferSampleIndex = 0;                                              // :220
    p->outputBufferSampleIndex = 0;                                             // :221
    if (p->outputSampleIndex >= p->H) p->outputSampleIndex = 0;                 // reference: left as is (UB if too large)
    std::fill(p->inBlock.begin(), p->inBlock.end(), 0.0f);
    // every buffer the audio callback touches is sized here: process_block never allocates
    p->maxBlocks = (p->H + p->B - 1) / p->B + 1;
    p->ready.assign((size_t) p->C * p->maxBlocks * p->B, 0.0f);
    p->conv.assign((size_t) p->C * p->maxBlocks * p->B, 0.0f);
    p->irWriteIndex = 0;                                                        // :226 clearAndResize
    irb_fdl_destroy(p->eng);
    p->eng = nullptr;
    irb_fdl_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.device = p->device; cfg.n_instances = 1; cfg.channels = p->C; cfg.block = p->B;
    cfg.max_ir_len = p->P * p->B; cfg.max_blocks_per_call = p->maxBlocks;
    // the callback's blocks are stepped one by one (rolling IR refresh between them) but enqueued together; kernels read /
    // write the pinned staging directly (no copy round trips)
    // ... and, where the BAR lets the host store into device memory, by a workgroup that stays resident between callbacks (no launch
    // per callback; it leaves by itself 20 ms after the last callback); IRB_RT_RESIDENT=0: one launch per callback
    cfg.flags = IRB_FDL_HOST_IO | IRB_FDL_STEPWISE | IRB_FDL_RESIDENT;
    if (dbg_int("rt_host_io", 1) == 0) cfg.flags &= ~IRB_FDL_HOST_IO;          // A/B (IRB_DEBUG, irb_env.hpp)
    if (dbg_int("rt_one_launch", 1) == 0) cfg.flags |= IRB_FDL_NO_RT_KERNEL;   // A/B
    if (const char *v = std::getenv("IRB_RT_RESIDENT")) { if (std::atoi(v) == 0) cfg.flags &= ~IRB_FDL_RESIDENT; }        // A/B
    return irb_fdl_create(&cfg, &p->eng);
}

int irb_rt_set_ir(irb_rt *p, const float *ir, int len)
{
    if (!p || !ir) return irb_set_error_(IRB_ERR_ARG, "null argument");
    if (len < p->P * p->B) {
        char msg[160];
        std::snprintf(msg, sizeof msg, "IR has %d samples, the rolling refresh reads %d (partitions*block)", len, p->P * p->B);
        return irb_set_error_(IRB_ERR_ARG, msg);
    }
    p->ir.assign(ir, ir + len);
    return IRB_OK;
}

int irb_rt_process_block_bypassed(irb_rt *p, float *const *buffer, int n)       // :584-590
{
    if (!p || !buffer) return irb_set_error_(IRB_ERR_ARG, "null argument");
    if (p->outSize == 0) return irb_set_error_(IRB_ERR_STATE, "process_block_bypassed before prepare_to_play");
    if (n != p->H) return irb_set_error_(IRB_ERR_ARG, "bypassed blocks must have the prepared samples_per_block");
    const size_t slot = (size_t) p->C * p->H;
    for (int c = 0; c < p->C; c++) std::memcpy(p->bypassRing.data() + p->bypWr * slot + (size_t) c * p->H, buffer[c], sizeof(float) * n);   // :586
    if (++p->bypWr >= p->outSize) p->bypWr = 0;                                  // :587
    if (++p->bypRd >= p->outSize) p->bypRd = 0;                                  // :588
    for (int c = 0; c < p->C; c++) std::memcpy(buffer[c], p->bypassRing.data() + p->bypRd * slot + (size_t) c * p->H, sizeof(float) * n);   // :589
    return IRB_OK;
}

int irb_rt_counters(const irb_rt *p, unsigned long long *callbacks, unsigned long long *launches, unsigned long long *waits,
                    unsigned long long *copies)
{
    if (!p || !p->eng) return irb_set_error_(IRB_ERR_STATE, "counters before prepare_to_play");
    if (callbacks) *callbacks = p->callbacks;
    return irb_fdl_get_counters(p->eng, launches, waits, copies);
}

int irb_rt_latency(const irb_rt *p) { return p ? p->latency : 0; }
int irb_rt_partitions(const irb_rt *p) { return p ? p->P : 0; }

int irb_rt_process_block(irb_rt *p, float *const *buffer, int n)                // :408-562
{
    if (!p || !buffer) return irb_set_error_(IRB_ERR_ARG, "null argument");
    if (!p->eng) return irb_set_error_(IRB_ERR_STATE, "process_block before prepare_to_play");
    if (n == 0) return IRB_OK;                                                  // :281-283
    if (n < 0) return irb_set_error_(IRB_ERR_ARG, "negative block length");
    const int C = p->C, B = p->B, H = p->H;
    for (int c = 0; c < C; c++) if (!buffer[c]) return irb_set_error_(IRB_ERR_ARG, "null channel pointer");
    if (n > H) {
        // a host that hands over more than the prepared block size (JUCE promises it never does): in pieces of at most H, so
        // that the staging and the output ring sized by prepare_to_play always suffice (the reference would write its output
        // ring past the slots it owns)
        if (C > (int) p->sub.size()) return irb_set_error_(IRB_ERR_STATE, "adapter not prepared");
        for (int off = 0; off < n; off += H) {
            for (int c = 0; c < C; c++) p->sub[c] = buffer[c] + off;
            int rc = irb_rt_process_block(p, p->sub.data(), std::min(H, n - off));
            if (rc != IRB_OK) return rc;
        }
        return IRB_OK;
    }

    // :421-445 collect host samples into blocks of B (run-wise copies; the reference moves one sample at a time)
    const size_t chan = (size_t) p->maxBlocks * B;          // per-channel pitch of ready / conv
    int nb = 0;
    for (int pos = 0; pos < n;) {
        const int take = std::min(B - p->inputBufferSampleIndex, n - pos);
        for (int c = 0; c < C; c++)
            std::memcpy(p->inBlock.data() + (size_t) c * B + p->inputBufferSampleIndex, buffer[c] + pos, sizeof(float) * take);
        pos += take;
        p->inputBufferSampleIndex += take;
        if (p->inputBufferSampleIndex >= B) {
            for (int c = 0; c < C; c++)
                std::memcpy(p->ready.data() + c * chan + (size_t) nb * B, p->inBlock.data() + (size_t) c * B, sizeof(float) * B);
            std::fill(p->inBlock.begin(), p->inBlock.end(), 0.0f);
            p->inputBufferSampleIndex = 0;
            nb++;
        }
    }

    // :452-518 the rolling IR refresh + one FDL step per complete block -- ONE engine call (one wait) for all of them --
    // then :525-543 into the output ring
    if (nb > 0) {
        for (int c = 0; c < C; c++) { p->inp[c] = p->ready.data() + c * chan; p->outp[c] = p->conv.data() + c * chan; }
        int rc = irb_fdl_process_refresh(p->eng, 0, p->inp.data(), p->outp.data(), nb, p->ir.data(), p->irWriteIndex);   // :455-461
        if (rc != IRB_OK) return rc;
        p->irWriteIndex = (p->irWriteIndex + nb) % p->P;
        for (int pos = 0; pos < nb * B;) {
            float *w = p->outRing.data() + (size_t) p->outWr * C * H;
            if (p->outputBufferSampleIndex == 0) std::fill(w, w + (size_t) C * H, 0.0f);
            const int take = std::min(H - p->outputBufferSampleIndex, nb * B - pos);
            for (int c = 0; c < C; c++)
                std::memcpy(w + (size_t) c * H + p->outputBufferSampleIndex, p->conv.data() + c * chan + pos, sizeof(float) * take);
            pos += take;
            p->outputBufferSampleIndex += take;
            if (p->outputBufferSampleIndex >= H) {
                if (++p->outWr >= p->outSize) p->outWr = 0;
                p->outputBufferSampleIndex = 0;
            }
        }
    }
    p->callbacks++;

    // :550-560 output ring -> host buffer
    for (int pos = 0; pos < n;) {
        const float *r = p->outRing.data() + (size_t) p->outRd * C * H;
        const int take = std::min(H - p->outputSampleIndex, n - pos);
        for (int c = 0; c < C; c++) std::memcpy(buffer[c] + pos, r + (size_t) c * H + p->outputSampleIndex, sizeof(float) * take);
        pos += take;
        p->outputSampleIndex += take;
        if (p->outputSampleIndex >= H) {
            if (++p->outRd >= p->outSize) p->outRd = 0;
            p->outputSampleIndex = 0;
        }
    }
    return IRB_OK;
}

// Measurement support: n_callbacks host callbacks of n samples per channel (white noise, in place, as a plug-in host
// would call processBlock) timed one by one on the calling thread with the monotonic clock.
int irb_rt_time_callbacks(irb_rt *p, int n, int n_callbacks, float *us_out)
{
    if (!p || !us_out) return irb_set_error_(IRB_ERR_ARG, "null argument");
    if (!p->eng) return irb_set_error_(IRB_ERR_STATE, "time_callbacks before prepare_to_play");
    if (n < 1 || n_callbacks < 1) return irb_set_error_(IRB_ERR_ARG, "n and n_callbacks must be >= 1");
    std::vector<float> store((size_t) p->C * n);
    std::vector<float *> ch((size_t) p->C);
    for (int c = 0; c < p->C; c++) ch[c] = store.data() + (size_t) c * n;
    unsigned long long rng = 0x9E3779B97F4A7C15ull;
    for (int k = 0; k < n_callbacks; k++) {
        for (float &v : store) {                       // fresh noise every callback (xorshift; values in [-1, 1))
            rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
            v = (float) ((double) (rng >> 40) / 8388608.0 - 1.0);
        }
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        int rc = irb_rt_process_block(p, ch.data(), n);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (rc != IRB_OK) return rc;
        us_out[k] = (float) ((t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3);
    }
    return IRB_OK;
}

}  // extern "C"
