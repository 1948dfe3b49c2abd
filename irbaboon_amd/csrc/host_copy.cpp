// host_copy.cpp -- the host-side sample copies of the staged one-shot call (pinned staging -> the caller's result buffer).
// Plain C++ (no HIP): built by the host compiler so that the AVX2 path can carry its own target attribute.
//
// Why not memcpy: the result of a one-shot call is megabytes that nobody reads before the call returns; glibc's memcpy uses ordinary
// stores below ~3/4 of the last-level cache, so every destination line is first READ into the cache (read-for-ownership) and the
// copy moves three bytes per byte copied.  Streaming stores skip that read.
#include <cstddef>
#include <cstdint>
#include <cstring>

#if defined(__x86_64__)
#include <immintrin.h>

__attribute__((target("avx2"))) static void copy_stream_avx2(char *dst, const char *src, size_t n)
{
    // head: up to the destination's next 32-byte boundary
    const size_t head = (32 - (reinterpret_cast<uintptr_t>(dst) & 31)) & 31;
    if (head) { std::memcpy(dst, src, head); dst += head; src += head; n -= head; }
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 32));
        const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 64));
        const __m256i d = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 96));
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i), a);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 32), b);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 64), c);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 96), d);
    }
    for (; i + 32 <= n; i += 32)
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i), _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i)));
    if (i < n) std::memcpy(dst + i, src + i, n - i);
    _mm_sfence();                                      // the streamed lines are globally visible before the caller is told "done"
}
#endif

extern "C" void irb_copy_stream_(void *dst, const void *src, size_t bytes)
{
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2 && bytes >= 4096) { copy_stream_avx2(static_cast<char *>(dst), static_cast<const char *>(src), bytes); return; }
#endif
    std::memcpy(dst, src, bytes);
}
