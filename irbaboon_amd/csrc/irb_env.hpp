// irb_env.hpp -- the environment the library reads (shared by fdl_engine.hip, rt_adapter.hip, deconv_engine.hip).
//
// For users:
//   IRB_OFFLINE_DIRECT      one-shot calls: unset / auto = zero-copy only for HIP host memory, 1 = also register pageable buffers, 0 = never
//   IRB_OFFLINE_THREADS     host threads of a staged one-shot call, the caller included (default 5, at most half of the host's cores)
//   IRB_RT_RESIDENT         0 = the plug-in adapter launches one kernel per callback instead of keeping a resident server
//   IRB_RT_SERVER_IDLE_US   how long a resident server waits for the next callback before it leaves (default 20 000)
//   IRB_RT_SERVER_LIFE_MS   the longest a resident server generation lives (default 250)
// For tests and tools only -- ONE variable, IRB_DEBUG="name=value,name=value" (read where a handle / plan is created or a call starts):
//   step kernels      fused_from fused_rows fused_u fused_nt colw psplits cols_handoff drop_granules
//   time-tiled calls  mfma_min_t mfma_wgs mfma_min_chunks tiled1 tiled_fused8
//   one-shot calls    offline_max_blocks offline_pieces offline_chunk offline_helpers_from offline_linger_us offline_in_one_launch input_checksums
//   real time         rt_bar rt_host_io rt_one_launch device_quiet
//   large FFT         lfft_log_m1 lfft_tile lfft_fuse_split
// Nothing else is read (rounds 1-4 had 41 IRB_* variables).
#pragma once
#include <cstdlib>
#include <cstring>

namespace {

inline int env_int(const char *name, int dflt)
{
    const char *v = std::getenv(name);
    return v ? std::atoi(v) : dflt;
}

// IRB_DEBUG="name=value,...": `present` tells a default of 0 from an explicit 0
inline int dbg_int(const char *name, int dflt, bool *present = nullptr)
{
    if (present) *present = false;
    const char *v = std::getenv("IRB_DEBUG");
    if (!v) return dflt;
    const size_t n = std::strlen(name);
    for (const char *p = v; *p;) {
        const char *e = std::strchr(p, ',');
        const size_t len = e ? (size_t) (e - p) : std::strlen(p);
        if (len > n && !std::strncmp(p, name, n) && p[n] == '=') { if (present) *present = true; return std::atoi(p + n + 1); }
        p += len + (e ? 1 : 0);
    }
    return dflt;
}

}  // namespace
