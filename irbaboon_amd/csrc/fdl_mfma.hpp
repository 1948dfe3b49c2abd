// fdl_mfma.hpp -- the time-tiled multiply-accumulate of long calls on the matrix cores (round 3).  Included by fdl_kernels.hpp.
//
// Along the time axis a bin of the frequency-domain delay line is a 1-D convolution,
//     Y[t] = sum_{m < np} H[m] * X[t - m]                     (fp/convolution.cpp:185-205, one complex bin, all blocks of a call)
// and a 1-D convolution cut into segments of 16 is a sum of dense 16 x 16 matrix products: with O[n][i] = Y[16 n + i],
//     O[n, :] = sum_j  Xc[n - j, :] * M_j,     M_j[k][i] = H[16 j + i - k]   (Toeplitz; H = 0 outside [0, np)),
//                                              Xc[c][k]  = X[16 c + k]       (the FDL's time axis folded into rows of 16).
// Every M_j is dense (one triangle comes from chunk j of the IR spectra, the other from chunk j - 1), so the products carry no
// structural zeros apart from the padding of np to whole chunks.  This is genuinely GEMM-shaped fp32 work -- per bin a
// [T/16 x 16 J] x [16 J x 16] product -- and the exact-fp32 matrix instructions have the vector ALU's peak (157.3 TFLOP/s) but,
// measured (tools/mfma_peak.hip, profiles/r03_mfma_peak.txt), sustain 0.89-0.97 of it from ONE wave per SIMD at 2.1-2.4 GHz, where
// the packed-FMA kernel (fdl_tiled_pk.hpp) is issue-bound at 0.43 and pulls the clock to 1.7 GHz.
//
// A complex product is four real ones: Or = sum Xr Hr - sum Xi Hi (two accumulators, subtracted once: no sign flip in the loop),
// Oi += Xi Hr + Xr Hi (the packed bin 0 -- DC and Nyquist, two real products -- takes the first of each only).  Workgroup = 16
// neighbouring bins (one 128-byte line of every FDL / H / partial row) x one tile of 16 segments (256 output blocks) x one range
// of j; 16 waves, one bin each (12 accumulator registers), four per SIMD: ONE wave cannot keep the matrix pipe full (its operand
// reads and address updates sit between its own MFMAs: 39-49 cycles per 32-cycle MFMA measured, tools/mfma_stamps.hip), and two
// waves that meet at a barrier every stage still left it idle a quarter of the time.
// The range of j is walked in stages of kMfmaJC chunks: the H taps and X rows a stage needs go global -> registers -> LDS
// (rows of 128 B: every line is used whole; buffer loads with the row step in an SGPR: no vector instruction per load) into one
// half of the LDS while the other half is multiplied -- one barrier per stage of 9 chunks; 141 KiB of LDS for both --, X folded
// as [k][row]: the 16 lanes of an operand read that share a k read 16 consecutive rows = 128 contiguous bytes (hipcc pairs the
// reads into ds_read2_b64, which is banked per 16 lanes and 32 banks: with k % 2 innermost SQ_LDS_BANK_CONFLICT was 29 % of the
// LDS cycles, now 2 % -- the kernel's time did not move: LDS is not what it waits for).  Per chunk a wave reads 8 operands of 8 bytes from LDS -- half a chunk ahead, in registers: a wave that waits for a
// ds_read lets the matrix pipe run dry -- and issues 16 v_mfma_f32_16x16x4_f32 (512 cycles of matrix pipe).  The 256 x 16 results leave through LDS as whole 128-byte
// rows of part[s][t][split][M] (summed over splits by k_ifft_ola_run, like those of k_cmac_tiled_pk).
//
// Not bit-identical to the vector kernels (another summation order); parity with the oracle at the same tolerance (tests).
// The index arithmetic (tiles, splits, stage windows, LDS images, operand positions) is restated in tests/kernel_model.py
// (mfma_time_axis) and checked against the direct sum on a machine without a GPU.
// Non-finite input: as with the zero rows that pad H to whole ring periods in fdl_tiled_pk.hpp, 0 * X products exist for X blocks
// next to the ones an output really depends on (here: up to 255 blocks later / 31 blocks earlier), so an Inf or NaN sample
// can spread further along the time axis than in the reference; finite input is unaffected.
#pragma once

namespace irb {

typedef float v4f_t __attribute__((ext_vector_type(4)));

constexpr int kMfmaSeg = 16;                           // segment length = MFMA tile edge
constexpr int kMfmaTile = kMfmaSeg * kMfmaSeg;         // output blocks per workgroup tile
constexpr int kMfmaBins = 16;                          // bins per workgroup = waves per workgroup
constexpr int kMfmaThreads = kMfmaBins * kWave;        // 1024
constexpr int kMfmaSR = kMfmaThreads / kMfmaBins;      // FDL rows a staging pass covers: 64 = four X rows of 16 blocks
#ifndef IRB_MFMA_JC
#define IRB_MFMA_JC 9
#endif
constexpr int kMfmaJC = IRB_MFMA_JC;                   // Toeplitz chunks per LDS stage (tools/mfma_stamps.hip builds variants)
constexpr int kMfmaNC = kMfmaJC + kMfmaSeg - 1;        // X rows (of 16 blocks) a stage needs
constexpr int kMfmaNH = kMfmaJC + 2;                   // H chunks a stage needs (one before, up to one after)
constexpr int kMfmaXI = (kMfmaNC + 3) / 4, kMfmaHI = (kMfmaNH + 3) / 4;     // staging passes
// per-bin LDS strides in float2, 1 (mod 16): the 16 bins a 16-lane group of ds_write_b64 stages fall into the 32 banks once
constexpr int kMfmaXS = kMfmaSeg * kMfmaNC + 1;
constexpr int kMfmaHS = kMfmaSeg * kMfmaNH + 1;
constexpr int kMfmaStage = kMfmaBins * (kMfmaXS + kMfmaHS);    // float2 per stage image: X as [bin][k][X row], then H as [bin][m]
constexpr int kMfmaOS = kMfmaBins + 1;                 // row stride of the output tile in LDS
constexpr size_t kMfmaLds = 2 * kMfmaStage * sizeof(float2);   // dynamic LDS of a launch (two stage images)
static_assert(kMfmaSR == 4 * kMfmaSeg, "a staging pass is four X rows / four H chunks");
static_assert(kMfmaTile * kMfmaOS <= kMfmaStage, "the output tile reuses a stage image");

// chunks of the Toeplitz sum for np partitions: j = 0 .. J-1 covers every m = 16 j + i - k in [0, np)
__host__ __device__ constexpr int mfma_chunks(int np) { return np > 0 ? (np + kMfmaSeg - 2) / kMfmaSeg + 1 : 0; }

// One stage of one wave: jc chunks of its bin out of a stage image.  Unit = half a chunk: X operand lane (n, kq) <- Xc[s0 + n - j][4 g + kq]
// at xb[4 g NC - j], H operand lane (i, kq) <- H[16 j + i - 4 g - kq] at hb[16 j - 4 g], g = 0, 1 / 2, 3; the operands of the
// next unit are read while this one is multiplied.  PACKED (bin 0: DC and Nyquist, two real products) is a template parameter, the
// last chunk is peeled: no branch between the MFMAs of a stage but the loop's own (with a wave-uniform `if (packed)` around every
// unit the kernel executed 0.7 branches and 3.4 scalar instructions per MFMA, and one wave needed ~45 cycles per 32-cycle MFMA).
template <bool PACKED>
__device__ __forceinline__ void mfma_stage(const float2 *xb, const float2 *hb, int jc, v4f_t &acc_r, v4f_t &acc_n, v4f_t &acc_i)
{
    auto read_ops = [&](float2 (&x)[2], float2 (&h)[2], int j, int half) {
#pragma unroll
        for (int gg = 0; gg < 2; gg++) {
            const int g = 2 * half + gg;
            x[gg] = xb[4 * g * kMfmaNC - j];
            h[gg] = hb[kMfmaSeg * j - 4 * g];
        }
    };
    auto multiply = [&](const float2 (&x)[2], const float2 (&h)[2]) {
#pragma unroll
        for (int g = 0; g < 2; g++) {
            if constexpr (PACKED) {
                acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(x[g].x, h[g].x, acc_r, 0, 0, 0);
                acc_i = __builtin_amdgcn_mfma_f32_16x16x4f32(x[g].y, h[g].y, acc_i, 0, 0, 0);
            } else {
                acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(x[g].x, h[g].x, acc_r, 0, 0, 0);
                acc_i = __builtin_amdgcn_mfma_f32_16x16x4f32(x[g].y, h[g].x, acc_i, 0, 0, 0);
                acc_n = __builtin_amdgcn_mfma_f32_16x16x4f32(x[g].y, h[g].y, acc_n, 0, 0, 0);      // subtracted at the end: no sign flip in the loop
                acc_i = __builtin_amdgcn_mfma_f32_16x16x4f32(x[g].x, h[g].y, acc_i, 0, 0, 0);
            }
        }
    };
    float2 x0[2], h0[2], x1[2], h1[2];
    read_ops(x0, h0, 0, 0);
#pragma unroll 1
    for (int j = 0; j < jc - 1; j++) {
        read_ops(x1, h1, j, 1);
        __builtin_amdgcn_sched_barrier(0);                         // reads first: left alone, the compiler sinks each one to its use
        multiply(x0, h0);
        __builtin_amdgcn_sched_barrier(0);
        read_ops(x0, h0, j + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        multiply(x1, h1);
        __builtin_amdgcn_sched_barrier(0);
    }
    read_ops(x1, h1, jc - 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    multiply(x0, h0);
    multiply(x1, h1);
}

// grid = S * ceil(T / 256) * splits * (M / 16); block = 1024; dynamic LDS = kMfmaLds.  part[s][t][split][M] like k_cmac_tiled_pk
// (a.nq = a.splits).
template <int N>
__global__ __launch_bounds__(kMfmaThreads, 1) void k_cmac_mfma(StepArgs a)
{
    constexpr int M = N / 2;
    static_assert(M % kMfmaBins == 0, "bins per workgroup");
    constexpr int GROUPS = M / kMfmaBins;
    extern __shared__ float2 l_stage[];                            // [2][kMfmaStage]
    const int ntiles = (a.T + kMfmaTile - 1) / kMfmaTile;
    const int wg = xcd_contiguous_id(blockIdx.x, gridDim.x);       // the bin groups and splits of one tile share an XCD's L2 (the same FDL rows)
    const int bg = wg % GROUPS, split = (wg / GROUPS) % a.splits, tile = (wg / (GROUPS * a.splits)) % ntiles, s = wg / (GROUPS * a.splits * ntiles);
    const int2 si = a.stream_info[s];
    const int set = si.x, np = si.y;
    const int J = mfma_chunks(np);
    const int jb = (int) ((long) J * split / a.splits), je = (int) ((long) J * (split + 1) / a.splits);
    const int s0 = tile * kMfmaSeg;                                // first segment of the tile
    const int b0 = bg * kMfmaBins;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid / kWave), lane = tid % kWave;
    IRB_STAMP(a, 0);
#ifdef IRB_STAMPS
    if (threadIdx.x == 0) {                            // where this workgroup runs: HW_ID (wave / SIMD / CU / SE) and the XCD
        a.stamps[(size_t) blockIdx.x * 16 + 14] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
        a.stamps[(size_t) blockIdx.x * 16 + 15] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));
    }
    unsigned long long t_mult = 0, t_mark = 0;
#endif

    // ---- staging: thread (sbin, srow) brings FDL rows srow + 64 it of its bin = block k = sk of X rows 4 it + sc, and taps
    // srow + 64 it of H.  fetch(): buffer loads into registers (one VGPR offset per thread and stage, the row step in an SGPR: no
    // vector instruction per load on the common path -- every instruction here is issued between this SIMD's MFMAs); rows outside
    // (-np, T) x [0, np) read as zero: their offset is pushed out of the descriptor's range.  put(): registers -> image p.
    constexpr unsigned ROWB = M * sizeof(float2), STEP = kMfmaSR * ROWB, OOR = 0x80000000u;
    const int sbin = tid % kMfmaBins, srow = tid / kMfmaBins, sk = srow % kMfmaSeg, sc = srow / kMfmaSeg;
    const unsigned colb = (unsigned) (b0 + sbin) * (unsigned) sizeof(float2);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(a.X + (size_t) s * a.R * M), 0, (int) ((unsigned) a.R * ROWB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(a.H + (size_t) set * a.P * M), 0, (int) ((unsigned) a.P * ROWB), 0x00020000);
    typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
    auto as_f2 = [](v2u_t v) { return make_float2(__uint_as_float(v.x), __uint_as_float(v.y)); };
    float2 xr[kMfmaXI], hr[kMfmaHI];
    auto fetch = [&](int ja, int jc) {
        const int nx = jc + kMfmaSeg - 1, nh = jc + 2;             // X rows / H chunks this stage uses
        const int tau_lo = kMfmaSeg * (s0 - ja - jc + 1), m_lo = kMfmaSeg * ja - kMfmaSeg;      // uniform
        int first = (a.head + tau_lo) % a.R;                       // ring slot of the stage's first row (uniform)
        if (first < 0) first += a.R;
        if (tau_lo > -np && tau_lo + kMfmaSeg * nx <= a.T && nx == kMfmaNC && first + kMfmaSR * kMfmaXI <= a.R) {
            // every row exists and the ring does not wrap inside the passes (the last pass reads up to three rows the stage never uses)
            const unsigned vo = (unsigned) (first + srow) * ROWB + colb;
#pragma unroll
            for (int it = 0; it < kMfmaXI; it++) xr[it] = as_f2(__builtin_amdgcn_raw_buffer_load_b64(rx, vo, it * STEP, 0));
        } else {
#pragma unroll
            for (int it = 0; it < kMfmaXI; it++) {
                const int tau = tau_lo + srow + kMfmaSR * it;      // block time relative to the call's first block
                int sl = a.head + tau;                             // in (-R, 2 R) whenever the row exists
                if (sl < 0) sl += a.R;
                if (sl >= a.R) sl -= a.R;
                const bool ok = 4 * it + sc < nx && tau > -np && tau < a.T;
                xr[it] = as_f2(__builtin_amdgcn_raw_buffer_load_b64(rx, ok ? (unsigned) sl * ROWB + colb : OOR, 0, 0));
            }
        }
        if (m_lo >= 0 && m_lo + kMfmaSR * kMfmaHI <= np && nh == kMfmaNH) {
            const unsigned vo = (unsigned) (m_lo + srow) * ROWB + colb;
#pragma unroll
            for (int it = 0; it < kMfmaHI; it++) hr[it] = as_f2(__builtin_amdgcn_raw_buffer_load_b64(rh, vo, it * STEP, 0));
        } else {
#pragma unroll
            for (int it = 0; it < kMfmaHI; it++) {
                const int m = m_lo + srow + kMfmaSR * it;
                const bool ok = 4 * it + sc < nh && m >= 0 && m < np;
                hr[it] = as_f2(__builtin_amdgcn_raw_buffer_load_b64(rh, ok ? (unsigned) m * ROWB + colb : OOR, 0, 0));
            }
        }
    };
    auto put = [&](int p) {
        float2 *sx = l_stage + p * kMfmaStage + sbin * kMfmaXS + (sk * kMfmaNC + sc);
        float2 *sh = l_stage + p * kMfmaStage + kMfmaBins * kMfmaXS + sbin * kMfmaHS + srow;
#pragma unroll
        for (int it = 0; it < kMfmaXI; it++)
            if (4 * it + 3 < kMfmaNC || 4 * it + sc < kMfmaNC) sx[4 * it] = xr[it];
#pragma unroll
        for (int it = 0; it < kMfmaHI; it++)
            if (4 * it + 3 < kMfmaNH || 4 * it + sc < kMfmaNH) sh[kMfmaSR * it] = hr[it];
    };

    // ---- multiplying: wave w owns bin b0 + w
    const int n = lane % kMfmaSeg, kq = lane / kMfmaSeg;           // MFMA operand position: free index = lane % 16, k = 4 g + lane / 16
    v4f_t acc_r, acc_n, acc_i;                                     // O[n = 4 kq + v][i = lane % 16]: re = acc_r - acc_n (Xr Hr and Xi Hi apart), im
#pragma unroll
    for (int v = 0; v < 4; v++) { acc_r[v] = 0.f; acc_n[v] = 0.f; acc_i[v] = 0.f; }
    const bool packed = b0 + wave == 0;                            // wave-uniform: the bin that holds DC and Nyquist
    auto multiply_stage = [&](int jc, int p) {
        const float2 *xb = l_stage + p * kMfmaStage + wave * kMfmaXS + (kq * kMfmaNC + n + jc - 1);
        const float2 *hb = l_stage + p * kMfmaStage + kMfmaBins * kMfmaXS + wave * kMfmaHS + (kMfmaSeg + n - kq);
        if (packed) mfma_stage<true>(xb, hb, jc, acc_r, acc_n, acc_i);
        else        mfma_stage<false>(xb, hb, jc, acc_r, acc_n, acc_i);
    };

    // Stage st is multiplied out of image st % 2.  Behind the barrier that ends stage st - 1 every wave first stores the rows of stage
    // st + 1 (loaded a whole stage ago) into the image that barrier has just released and asks for the rows of stage st + 2, then
    // multiplies: a few dozen instructions per wave, and those of the younger waves are issued under the older waves' MFMAs -- the
    // matrix pipe stands still only from the last wave's last MFMA over the barrier to the first wave's first.  (With the store at the
    // END of a stage, after the wave's MFMAs, every stage cost ~2700 cycles of idle pipe on top of its 12 300 of MFMAs.)
    const int nst = (je - jb + kMfmaJC - 1) / kMfmaJC;
    auto stage_begin = [&](int st) { return min(jb + kMfmaJC * st, je); };
    if (nst > 0) { fetch(jb, stage_begin(1) - jb); put(0); }
    if (nst > 1) fetch(stage_begin(1), stage_begin(2) - stage_begin(1));
    __syncthreads();
    IRB_STAMP(a, 1);
#ifdef IRB_STAMPS
    t_mark = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll 1
    for (int st = 0; st < nst; st++) {
        // (at raised priority: at equal priority the arbiter serves the oldest wave's MFMA stream first, the youngest waves took 16 000 cycles
        // to get through these few dozen instructions and then multiplied alone at the end of the stage -- tools/mfma_stamps.hip ... waves)
        __builtin_amdgcn_s_setprio(3);
        if (st + 1 < nst) put((st & 1) ^ 1);
        if (st + 2 < nst) fetch(stage_begin(st + 2), stage_begin(st + 3) - stage_begin(st + 2));
        __builtin_amdgcn_s_setprio(0);
#ifdef IRB_STAMPS
        const unsigned long long t_in = __builtin_amdgcn_s_memtime();
#endif
        multiply_stage(stage_begin(st + 1) - stage_begin(st), st & 1);
#ifdef IRB_STAMPS
        const unsigned long long t_out = __builtin_amdgcn_s_memtime();
        t_mult += t_out - t_in;
        if (blockIdx.x == gridDim.x / 2 && lane == 0 && st < 4) {         // one workgroup's per-wave timeline, behind the per-workgroup records
            unsigned long long *w = a.stamps + (size_t) gridDim.x * 16 + wave * 16 + st * 4;
            w[0] = t_mark; w[1] = t_in; w[2] = t_out;
            w[3] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));      // HW_ID: SIMD in bits 5:4
        }
#endif
        __syncthreads();                                           // image st % 2 has been multiplied by everyone; image (st + 1) % 2 is complete
    }
    IRB_STAMP(a, 2);
#ifdef IRB_STAMPS
    if (threadIdx.x == 0) a.stamps[(size_t) blockIdx.x * 16 + 4] = t_mult;      // wave 0: cycles inside multiply_stage (the rest of the loop: staging stores, barriers)
#endif
    // ---- store: accumulator register v of lane (i, kq) is O[n = 4 kq + v][i], i.e. block 16 (s0 + n) + i of the call; through LDS,
    // so that each output row leaves as one whole 128-byte line
    float2 *l_o = l_stage;
#pragma unroll
    for (int v = 0; v < 4; v++) l_o[(kMfmaSeg * (4 * kq + v) + n) * kMfmaOS + wave] = make_float2(acc_r[v] - acc_n[v], acc_i[v]);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kMfmaTile / kMfmaSR; it++) {
        const int tl = srow + kMfmaSR * it, t = kMfmaTile * tile + tl;
        if (t < a.T) a.part[(((size_t) s * a.T + t) * a.nq + split) * M + b0 + sbin] = l_o[tl * kMfmaOS + sbin];
    }
    IRB_STAMP(a, 7);
}

}  // namespace irb
