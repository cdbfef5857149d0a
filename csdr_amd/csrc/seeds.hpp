// seeds.hpp -- chunk-seed tables for N streams with INDIVIDUAL shift rates (the per-stream variants of the fused WFM / receiver front-end chains).
//
// shift_addition_cc restarts its phasor at every 1024-sample chunk from (cos, sin) of a FLOAT phase that the CLI loop carries from chunk to chunk
// (libcsdr_gpl.c:33-35, 48-51; csdr.c:896-923):   ph <- ph + rate*PI*1024;  while (ph > PI) ph -= 2*PI;  while (ph < -PI) ph += 2*PI;   -- every step rounded
// to float.  The sequence depends on nothing but the rate (and the phase at the last retune), but it has to be replayed rounding for rounding: the float
// bookkeeping wanders by ~1e-5 rad * sqrt(chunks) against the exact phase.  With one rate for all streams the host does it (16 calls ahead, wfm.hip / ddc_mfma.hip);
// with a rate per stream it is one lane per stream on the device, on a side stream, a few calls ahead of the data.
#pragma once
#include "common.hpp"
#include <vector>

namespace csdr_amd {

// ---- the phase advance of one chunk without the wrap loop's iterations:   ph' = wrap(fl(ph + step)),   wrap(x):  while (x > PI_F) x -= 2 PI_F;  while (x < -PI_F) x += 2 PI_F;
// every subtraction rounded to float (libcsdr_gpl.c:48-51).  A rate of 0.4 means 410 iterations per chunk, and the chunks of a stream are a strictly sequential chain:
// what counts is the number of DEPENDENT operations per chunk.
//   * Why not fmod: each subtraction rounds on the grid of its RESULT's binade.  c = 2 PI_F = 0xC90FDB * 2^-21; for a result in [2^e, 2^(e+1)), e >= 4, the grid is
//     2^(e-23) >= 4 * 2^-21, c is never half-way between two grid points (0xC90FDB mod 2^k != 2^(k-1), k = 2 .. 10) and the minuend is on the grid already: the step
//     subtracts the CONSTANT c_e = c rounded to that grid, exactly.  Binades 9-11 share one c_e, 6-7 another: five groups (lo = 512, 256, 64, 32, 16).
//   * Steps stay inside a group while a - c_e >= lo (c_e > c) or > lo (c_e < c: the exact difference then lies below lo and rounds on the finer grid); their number
//     for a value a is floor((a - lo') / c_e) with lo' = lo or lo + grid.
//   * The values that ENTER a group span less than c + a rounding slack: from above they come out of [lo_above - c, lo_above), directly they are |ph + step| with
//     |ph| <= PI.  So the step count takes one of three consecutive values, told apart by two comparisons against thresholds that depend on nothing but the
//     stream's step (only its topmost group differs from the universal constants): a group costs two compares, two selects, two exact additions, the exact bulk
//     subtraction, the one rounded subtraction that leaves the group, and a select -- eight dependent operations instead of a loop.  Below 16 (ties occur there): at most three plain steps.
// tests/test_abi_cpu.py runs the host build of this against the loop.
struct WrapPlan {
    float t1[5], t2[5], k0[5];
};

__host__ __device__ inline void wrap_plan_init(WrapPlan &w, float step)
{
    const double PI = (double)(float)3.14159265358979323846, C = 2 * PI;
    const double lo[5] = {512, 256, 64, 32, 16};
    const double ce[5] = {0x1.922p+2, 0x1.921f8p+2, 0x1.921fcp+2, 0x1.921fbp+2, 0x1.921fb8p+2};      // c on the grids 2^-14 (binades 9-11), 2^-15, 2^-17 (6-7), 2^-18, 2^-19
    const double grid[5] = {0x1p-14, 0x1p-15, 0x1p-17, 0x1p-18, 0x1p-19};
    const double top = fabs((double)step) + PI + 0x1p-9;               // no |ph + step| exceeds this (the addition's rounding included)
#pragma unroll                                                        // (constant indices: the plan has to live in registers -- as a scratch array every group paid a memory round trip)
    for (int i = 0; i < 5; i++) {
        const double lop = ce[i] < C ? lo[i] + grid[i] : lo[i];       // steps stay in the group while the result is >= lop
        double hi = i ? lo[i - 1] : top; if (top < hi) hi = top;       // no value that enters the group reaches hi ...
        double amin = hi - C - 0x1p-8; if (amin < lo[i]) amin = lo[i]; // ... and none lies below amin
        double n0 = floor((amin - lop) / ce[i]); if (n0 < 0) n0 = 0;
        while (lop + (n0 + 1) * ce[i] <= amin) n0 += 1;
        while (n0 > 0 && lop + n0 * ce[i] > amin) n0 -= 1;
        w.t1[i] = (float)(lop + (n0 + 1) * ce[i]); w.t2[i] = (float)(lop + (n0 + 2) * ce[i]);      // exact: multiples of the grid below 2^(e+1)
        w.k0[i] = (float)(n0 * ce[i]);
        if (hi <= lo[i]) { w.t1[i] = w.t2[i] = 8192.f; w.k0[i] = 0.f; }                             // never entered
    }
}

// x = fl(ph + step) of the stream the plan was made for (|ph| <= PI)
__host__ __device__ inline float wrap_plan_apply(const WrapPlan &w, float x)
{
    const float PI = (float)3.14159265358979323846, C2 = 2 * PI;
    float a = fabsf(x);
    // (sums, not selects between table entries: the compiler turns a select of two array elements into a load through a selected ADDRESS, and the plan into scratch
    // memory -- one memory round trip per group.  Round 5, measured and dropped: the three possible results of a group side by side, then picked -- a dependent chain
    // of 29 instead of 46 operations per chunk step, bit-exact on 48 M values (tools/probes/plan_check.c), but 74 instead of ~62 instructions: the generator is ONE wave
    // per SIMD, every instruction of which occupies the SIMD for its four cycles whether it depends on the previous one or not -- k_seed_phases 1.59 -> 1.74 ms.)
#define CSDR_WRAP_GROUP(I, LO, CE)                                                                                    \
    {                                                                                                                 \
        const float k = (w.k0[I] + (a >= w.t1[I] ? CE : 0.f)) + (a >= w.t2[I] ? CE : 0.f);   /* exact: (n0 + 0..2) c_e */ \
        const float r = (a - k) - C2;                                 /* exact bulk, then the (rounded) step that leaves the group */ \
        a = a >= LO ? r : a;                                                                                          \
    }
    CSDR_WRAP_GROUP(0, 512.f, 0x1.922p+2f)
    CSDR_WRAP_GROUP(1, 256.f, 0x1.921f8p+2f)
    CSDR_WRAP_GROUP(2, 64.f, 0x1.921fcp+2f)
    CSDR_WRAP_GROUP(3, 32.f, 0x1.921fbp+2f)
    CSDR_WRAP_GROUP(4, 16.f, 0x1.921fb8p+2f)
#undef CSDR_WRAP_GROUP
#pragma unroll
    for (int i = 0; i < 3; i++) a = a > PI ? a - C2 : a;              // from below 16
    return x < 0 ? 0.0f - a : a;                                      // (-c + c is +0 in the loop as well)
}

struct SeedTables;

// rates: n_streams floats.  d_dtab / dtab_stride: per-stream tables D^(i - 2048), i in [0, 3072) (float2), for the drift corrections of the receiver front end
// (nullptr: no corrections -- the WFM chain, whose demodulator cancels them).  max_block_samples bounds the chunks one call may ask for.
SeedTables *seeds_create(csdr_amd_ctx *ctx, int n_streams, const float *rates, const float2 *d_dtab, size_t dtab_stride, size_t max_block_samples);
void seeds_destroy(SeedTables *t);
int seeds_set_drift(SeedTables *t, const std::vector<char> &drift);   // which streams get drift corrections (all at once: create time; synchronises)
int seeds_reset(SeedTables *t);                                        // stream start: phase 0 in front of chunk 0
int seeds_set_rate(SeedTables *t, int stream, float rate, bool drift); // from the next acquired call on; the phase carries over (csdr.c:881-923)

// device pointer to the 32 drift corrections of chunk `chunk` of `stream` in the table the data kernels use now, or nullptr (no corrections for the stream / chunk
// not in the table): what a retune has to keep for the samples in front of the next block
const float2 *seeds_corr_entry(SeedTables *t, int stream, long long chunk);

struct SeedView {
    const float2 *ctab;      // seed of chunk (first + k) of stream s: ctab[k * pitch + s]
    size_t pitch;
    int n_entries;           // entries from `first` on that the table holds (>= what was asked for)
    const float2 *corr;      // drift corrections or nullptr; row r = corr_row[s] >= 0: corr[((size_t)r * corr_chunks + k) * 32 + j]
    const int *corr_row;     // device, [n_streams]
    size_t corr_chunks;
};
// Before a call that covers chunks [first, first + n): the view is valid for kernels queued on the context's stream after this returns.
int seeds_acquire(SeedTables *t, long long first, size_t n, size_t n_next_hint, SeedView *v);

} // namespace csdr_amd
