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

// ---- the wrap loop without its iterations.  x: any float with |x| < 4096.  Returns exactly what
//          while (x > PI_F) x -= 2 * PI_F;   while (x < -PI_F) x += 2 * PI_F;
// leaves in float arithmetic.  Why this is not a plain fmod: every subtraction rounds to the grid of its RESULT's binade.  c = 2*PI_F = 0xC90FDB * 2^-21; for a result
// in [2^e, 2^(e+1)), e >= 4, the grid is 2^(e-23) >= 4 * 2^-21 and c is never half-way between two grid points (0xC90FDB mod 2^k != 2^(k-1) for k = 2 .. 10), while the
// minuend is on the grid already: the step subtracts the CONSTANT c_e = c rounded to that grid, exactly.  So all steps whose result stays in one binade collapse into
// one exact fused multiply-add; the step that leaves a binade, and everything below 16 (where the grid reaches c's own and ties occur), are plain float subtractions.
// A rate of 0.4 needs 410 iterations per chunk in the loop, this takes ~100 dependent operations for any rate (tests/test_abi_cpu.py checks it against the loop).
__host__ __device__ inline float wrap_phase_exact(float x)
{
    const float PI = (float)3.14159265358979323846, C2 = 2 * PI;
    float a = fabsf(x);
    if (!(a > PI)) return x;
    // c rounded to the grids of binades 11 .. 4 (2^-12 .. 2^-19) and the reciprocals used for the step counts
#define CSDR_WRAP_STAGE(E, CE)                                                                                       \
    {                                                                                                                \
        const float lo = (float)(1u << E), ce = CE;                                                                  \
        if (a >= lo) {                                                                                               \
            const float n = floorf((a - lo) * (1.0f / CE));           /* steps that stay in the binade (off by one at most) */ \
            float a1 = fmaf(-n, ce, a);                               /* exact: a multiple of the binade's grid, below 2^(E+1) */ \
            if (a1 < lo) a1 += ce;                                                                                   \
            if (a1 < a && ((a1 + ce) - C2) < lo) a1 += ce;            /* the last step counted left the binade: c_e != c */ \
            else if ((a1 - C2) >= lo) a1 -= ce;                       /* one more stays */                           \
            a = a1 - C2;                                              /* the step that leaves the binade: plain float subtraction */ \
        }                                                                                                            \
    }
    CSDR_WRAP_STAGE(11, 0x1.922p+2f)             // c on the 2^-12 grid
    CSDR_WRAP_STAGE(10, 0x1.922p+2f)             // 2^-13
    CSDR_WRAP_STAGE(9, 0x1.922p+2f)              // 2^-14
    CSDR_WRAP_STAGE(8, 0x1.921f8p+2f)            // 2^-15
    CSDR_WRAP_STAGE(7, 0x1.921fcp+2f)            // 2^-16
    CSDR_WRAP_STAGE(6, 0x1.921fcp+2f)            // 2^-17
    CSDR_WRAP_STAGE(5, 0x1.921fbp+2f)            // 2^-18
    CSDR_WRAP_STAGE(4, 0x1.921fb8p+2f)           // 2^-19
#undef CSDR_WRAP_STAGE
    while (a > PI) a -= C2;                                           // from below 16: at most three more
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
