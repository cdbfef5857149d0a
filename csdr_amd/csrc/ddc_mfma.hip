// ddc_mfma.hip -- fused receiver front end on the matrix cores:
//
//        convert_u8_f | shift_addition_cc <rate> | fir_decimate_cc <D> <tbw> <window>
//
// (libcsdr.c:2363-2368, libcsdr_gpl.c:27-52 with the CLI's 1024-sample chunks csdr.c:911-918, libcsdr.c:528-549 with the CLI's
// re-feed loop csdr.c:1160-1176) for N independent u8 IQ streams: the head of the reference's NFM, AM and SSB receive chains
// (README.md:87, 95, 110 -- D = 50, 801 taps).  Stream model:   y[k] = sum_t h[t] x'[D k + t],   x'[n] = u8_to_float(iq[n]) R[n].
//
// Like the WFM front end (wfm_mfma.hip) the whole thing is LINEAR in the input bytes, and u8 - 128 is exactly an int8:
//
//        y[k] = sum_n  a h[n - D k] R[n] (v_n - 128)  +  const ,          a = 2/255
//
// so conversion, rotation and FIR are one banded int8 product on v_mfma_i32_16x16x64_i8 with exact int32 accumulation and
// 23-bit weights (three base-256 digits).  What is new here:
//
//  * ONE weight set for every tile.  shift_addition_cc's phasor inside a 1024-chunk is R[n] = C_m D^(n - 1024 m) (C_m = float
//    (cos, sin) of the chunk's float starting phase, D = (cos d, sin d) rounded to float), so relative to a tile's first sample n0
//    R[n0 + t] = [C_m D^(n0 - 1024 m)] D^t: the weights a h[t - D o] D^t do not depend on the tile, the bracket is a complex
//    scalar per (tile, chunk) applied AFTER the product.  (The WFM kernel keeps 128 phase-specific weight sets and therefore has to
//    give every wave a fixed tile phase and jump through the input; here a workgroup can walk through time contiguously.)
//  * 16 rows = {Re, Im} x 8 consecutive outputs; the 8 D + L - 1 sample window (1151 samples = 36 K-steps for D = 50, L = 801) is
//    split over the 4 waves of a workgroup (9 K-steps = 27 weight fragments per wave, register resident for the whole launch),
//    partial sums are reduced through LDS.  The band is 70 % dense (WFM: 31 %).
//  * A 1024-chunk boundary inside a wave's K-range is handled without a second weight set: the accumulator chain is snapshotted
//    at the boundary (side 0 = snapshot, side 1 = total - snapshot); when the boundary falls in the middle of a K-step (it is
//    32-byte granular) that K-step is multiplied twice with the other half of the B operand zeroed.
//  * Input: a workgroup owns 16 streams x a contiguous run of tiles and slides over the input: every byte is fetched ONCE, as
//    whole 1-KiB runs of 128-byte lines, by LDS-DMA (global_load_lds_dwordx4, per-instruction M0) into a per-stream ring.
//  * Outputs that need the previous block's tail (history) or do not fill a tile run on k_ddc_direct, a plain one-thread-per-
//    output evaluation of the same model, which is also the fallback for shapes the matrix-core kernel does not cover.
#include "common.hpp"
#include "nfm_demod.hpp"
#include "seeds.hpp"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <complex>
#include <string>
#include <vector>
using namespace csdr_amd;

namespace {

constexpr int DDC_NKT = 36;                 // 64-byte K-steps per tile window (8 outputs): 2*(7 D + L) <= 2304 bytes
#ifndef DDC_WPT_N
#define DDC_WPT_N 4
#endif
constexpr int DDC_WPT = DDC_WPT_N;          // waves per team = K-ranges per tile.  -DDDC_WPT_N=6: 18 instead of 27 register-resident weight fragments per wave, 155 registers,
                                            // THREE waves per SIMD (two teams of six): measured 0.784 against 0.792 ms per NFM step with two fetchers per team, 0.797 with
                                            // four -- every K-range pays its own post-processing, the kernel is bound by instruction issue, not by latency (profiles/r3_notes.md)
constexpr int DDC_NKW = DDC_NKT / DDC_WPT;  // K-steps per wave
static_assert(DDC_NKT % DDC_WPT == 0 && (DDC_WPT == 4 || DDC_WPT == 6), "K-range split");
#ifndef DDC_FETCHERS
#define DDC_FETCHERS 2                      // fused epilogue: waves per team that fetch (of the DDC_WPT - 2 without an epilogue role)
#endif
constexpr int DDC_WIN = 64 * DDC_NKT;       // 2304 bytes
constexpr int DDC_HIST = 1024;              // complex samples of input history kept per stream between blocks (>= L - 1; one chunk, so that history has ONE phasor seed)
constexpr int DDC_DTAB = 3072;              // D^k for k in [-2048, 1024)
constexpr int DDC_NGRAN = 2 * DDC_NKT + 1;  // prefix sums of the weights at 32-byte granules

typedef int v4i __attribute__((ext_vector_type(4)));

struct DdcTable {
    int D, L; float scale;
    std::vector<int8_t> frags;              // [DDC_NKT][3 digits][64 lanes][16]  (lane l: row l%16, K bytes 16*(l/16) .. +15 of that K-step)
    std::vector<float> cum;                 // [DDC_NGRAN][16 rows]: 0.5 * sum of the (unquantised) weights of row r over bytes < 32 g
    std::vector<float2> dtab;               // D^(i - 2048)
};

bool ddc_mfma_supported(int D, int L)
{
    // D even: tile starts stay 32-byte granular (chunk boundaries fall on half K-steps); 16 D <= 3584: the 8 KiB ring holds the current window plus the next tile's bytes
    return D >= 2 && D <= 224 && (D % 2) == 0 && 2 * (7 * D + L) <= DDC_WIN && L - 1 <= DDC_HIST;
}

// shift_addition_init libcsdr_gpl.c:81-89: the per-sample phasor D as the reference's floats
std::complex<double> ddc_step_phasor(float shift_rate)
{
    const float rate2 = shift_rate * 2, inc = rate2 * PI_F;
    return std::complex<double>((double)(float)cos((double)inc), (double)(float)sin((double)inc));
}

void ddc_build_dtab(float shift_rate, std::vector<float2> &dtab)
{
    const std::complex<double> d = ddc_step_phasor(shift_rate);
    const double mag = std::abs(d), ang = std::arg(d);
    dtab.resize(DDC_DTAB);
    for (int i = 0; i < DDC_DTAB; i++) {
        const int k = i - 2048;
        const std::complex<double> v = std::polar(pow(mag, k), ang * k);
        dtab[i] = make_float2((float)v.real(), (float)v.imag());
    }
}

void ddc_build_table(int D, int L, float shift_rate, const float *taps, DdcTable &t)
{
    t.D = D; t.L = L;
    const std::complex<double> d = ddc_step_phasor(shift_rate);
    const double mag = std::abs(d), ang = std::arg(d);
    ddc_build_dtab(shift_rate, t.dtab);
    const double a = 2.0 / 255.0;
    const int span = 7 * D + L;                                       // samples of the tile window
    double gmax = 0, dmax = fmax(1.0, pow(mag, span));
    for (int k = 0; k < L; k++) gmax = fmax(gmax, fabs(a * (double)taps[k]));
    gmax *= dmax * 1.0001;
    if (gmax == 0) gmax = 1;
    const double qscale = 4194304.0 / gmax;                           // 2^22: three balanced base-256 digits stay inside int8
    t.scale = (float)(gmax / 4194304.0);
    t.frags.assign((size_t)DDC_NKT * 3 * 64 * 16, 0);
    std::vector<double> gsum((size_t)(DDC_NGRAN - 1) * 16, 0.0);
    std::vector<std::complex<double>> dpow((size_t)span);             // D^ts, once per sample of the window (a per-stream object builds one table per stream)
    for (int ts = 0; ts < span; ts++) dpow[ts] = std::polar(pow(mag, ts), ang * ts);
    for (int r = 0; r < 16; r++) {
        const int o = r / 2, comp = r % 2;                            // row = (output o of the tile, Re / Im)
        for (int tp = 0; tp < L; tp++) {
            const int ts = D * o + tp;                                // sample relative to the tile's first sample
            const std::complex<double> G = a * (double)taps[tp] * dpow[ts];
            for (int c = 0; c < 2; c++) {
                // real form of (Gr + j Gi)(I + j Q): Re row takes (Gr, -Gi) on (I, Q); Im row takes (Gi, Gr)
                const double val = comp == 0 ? (c == 0 ? G.real() : -G.imag()) : (c == 0 ? G.imag() : G.real());
                const int colb = 2 * ts + c, ks = colb / 64, b = colb % 64;
                long qv = lrint(val * qscale);
                const int w2 = (int)(((qv + 128) % 256 + 256) % 256) - 128; qv = (qv - w2) / 256;
                const int w1 = (int)(((qv + 128) % 256 + 256) % 256) - 128; qv = (qv - w1) / 256;
                const int w0 = (int)qv;
                const int lane = 16 * (b / 16) + r, byte = b % 16;
                const int dig[3] = {w0, w1, w2};
                for (int l = 0; l < 3; l++) t.frags[((size_t)(ks * 3 + l) * 64 + lane) * 16 + byte] = (int8_t)dig[l];
                gsum[(size_t)(colb / 32) * 16 + r] += val;
            }
        }
    }
    // u8 -> float is a (v - 128) + 1/255 = a (v - 128 + 0.5): the offset's way through the filter is 0.5 * (sum of the weights)
    t.cum.assign((size_t)DDC_NGRAN * 16, 0.f);
    for (int r = 0; r < 16; r++) {
        double run = 0;
        for (int g = 0; g < DDC_NGRAN; g++) { t.cum[(size_t)g * 16 + r] = (float)(0.5 * run); if (g + 1 < DDC_NGRAN) run += gsum[(size_t)g * 16 + r]; }
    }
}

__host__ __device__ __forceinline__ float2 cmulf(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// How well does R[n] = C_m D^k describe the reference's float32 phasor recurrence (libcsdr_gpl.c:44-45) for this rate?  For most rates the
// recurrence's rounding errors average out (relative RMS deviation ~8e-7); for rates whose orbit is short (0.05, 0.25, ...) they repeat and
// accumulate to a systematic drift of 1e-5 .. 4e-5 over a chunk.  Returns the RMS deviation over a chunk, averaged over a few seeds.
double ddc_rotator_model_rms(float shift_rate)
{
    const float rate2 = shift_rate * 2, inc = rate2 * PI_F;
    const float cd = (float)cos((double)inc), sd = (float)sin((double)inc);
    const std::complex<double> d((double)cd, (double)sd);
    const double mag = std::abs(d), ang = std::arg(d);
    double acc = 0; int cnt = 0;
    for (int seed = 0; seed < 12; seed++) {
        const float ph = -3.0f + 0.53f * seed;
        float c = (float)cos((double)ph), s = (float)sin((double)ph);
        const std::complex<double> C((double)c, (double)s);
        for (int k = 0; k < 1024; k++) {
            if (k >= 256 && (k & 63) == 0) {
                const std::complex<double> e = std::complex<double>(c, s) / (C * std::polar(pow(mag, k), ang * k)) - 1.0;
                acc += std::norm(e); cnt++;
            }
            const float c2 = c * cd - s * sd, s2 = s * cd + c * sd;   // -ffp-contract=off: separate products and sums, as the reference's SSE code
            c = c2; s = s2;
        }
    }
    return sqrt(acc / cnt);
}

// corr[m][j] = (float recurrence started at C_m, after k = 32 j + 16 steps) / (C_m D^k): the slowly varying factor by which the reference's
// phasor deviates from the model inside chunk m.  One lane per chunk replays the chunk (data independent, shared by all streams).
__global__ __launch_bounds__(64) void k_ddc_corr(const float2 *__restrict__ ctab, const float2 *__restrict__ dtab, float2 *__restrict__ corr, int n_rows, float cd, float sd)
{
    const int m = blockIdx.x * 64 + threadIdx.x;
    if (m >= n_rows) return;
    const float2 C = ctab[m];
    float c = C.x, s = C.y;
    // 1024 strictly sequential steps per lane: straight-line runs of 16 (the loop counter, the sample test and the branch were most of the 42 us this took)
    for (int j = 0; j < 32; j++) {
#pragma unroll
        for (int k = 0; k < 16; k++) { const float c2 = c * cd - s * sd, s2 = s * cd + c * sd; c = c2; s = s2; }      // libcsdr_gpl.c:44-45
        {
            const float2 ref = cmulf(C, dtab[32 * j + 16 + 2048]);
            const float inv = 1.0f / (ref.x * ref.x + ref.y * ref.y);
            corr[(size_t)m * 32 + j] = make_float2((c * ref.x + s * ref.y) * inv, (s * ref.x - c * ref.y) * inv);
        }
#pragma unroll
        for (int k = 0; k < 16; k++) { const float c2 = c * cd - s * sd, s2 = s * cd + c * sd; c = c2; s = s2; }
    }
}

// ---- geometry of one wave's K-range, shared by the kernel and the CPU evaluation (csdr_amd_debug_ddc_mfma_tile)
struct WaveGeom {
    int two;          // the K-range contains a 1024-chunk boundary
    int kb, half;     // boundary K-step (0..8) and whether it falls in the middle (32 bytes in) of that K-step
    int gb;           // boundary granule (32-byte units from the window start)
    int e0;           // exponent of the post factor of side 0: P0 = C_m D^e0 ; side 1: C_{m+1} D^(e0 - 1024)
    long long chunk;  // absolute chunk index of side 0
};
__host__ __device__ __forceinline__ WaveGeom ddc_wave_geom(long long n0, int w)
{
    WaveGeom g;
    const long long s = n0 + 32LL * DDC_NKW * w;                      // first sample of the wave's K-range
    const int off = (int)(s & 1023);
    g.chunk = s >> 10;
    g.e0 = (int)(n0 - (s & ~1023LL));
    g.two = off + 32 * DDC_NKW > 1024;
    const int bo = 2 * (1024 - off);                                  // byte offset of the next chunk's first sample inside the K-range
    g.kb = g.two ? (bo >> 6) : DDC_NKW;
    g.half = g.two ? ((bo >> 5) & 1) : 0;
    g.gb = 2 * DDC_NKW * w + (bo >> 5);
    return g;
}

__device__ __forceinline__ float combine_digits(int a0, int a1, int a2)
{   // exact integers (<= 23 bits each) recombined in float: value = a0*65536 + a1*256 + a2
    return fmaf((float)a0, 65536.0f, fmaf((float)a1, 256.0f, (float)a2));
}
__device__ __forceinline__ float u8_to_f(uint32_t v) { return fmaf((float)v, 0x1.010102p-7f, -1.0f); }   // v/127.5 - 1 (<= 1 ulp of the reference's double expression)

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// The accumulator chains of one wave's K-range with the chunk boundary at K-step KB (compile time; KB = DDC_NKW: no boundary), so that every
// variant is straight-line code (run-time `if (ks == kb)` branches inside the unrolled chain made the compiler shuffle the accumulators through
// AGPR copies at every merge point: 700 v_accvgpr moves per tile, 4x the kernel time).  half: the boundary lies 32 bytes into K-step KB: lanes
// q < 2 hold its chunk-m bytes.  Handled branch free: the K-step is multiplied twice with complementary halves of B zeroed (with half = 0 the
// first product is a product with zero).
template <int KB>
__device__ __forceinline__ void ddc_chain(const v4i (&A)[DDC_NKW * 3], const v4i (&Bf)[DDC_NKW], bool lo_lane, v4i (&acc)[3], v4i (&snap)[3])
{
    const v4i z = {0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < DDC_NKW; ks++) {
        if (ks == KB) {
            const v4i lo = lo_lane ? Bf[ks] : z, hi = lo_lane ? z : Bf[ks];
#pragma unroll
            for (int l = 0; l < 3; l++) acc[l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[ks * 3 + l], lo, acc[l], 0, 0, 0);
#pragma unroll
            for (int l = 0; l < 3; l++) snap[l] = acc[l];
#pragma unroll
            for (int l = 0; l < 3; l++) acc[l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[ks * 3 + l], hi, acc[l], 0, 0, 0);
        } else {
#pragma unroll
            for (int l = 0; l < 3; l++) acc[l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[ks * 3 + l], Bf[ks], acc[l], 0, 0, 0);
        }
    }
}

// the K-ranges' shares of a lane's two outputs, always summed in this order (the fused epilogue's roles and its predecessor logic must agree bit for bit)
template <int WPT>
__device__ __forceinline__ float4 ddc_reduce(const float4 *b)
{
    const float4 p0 = b[0], p1 = b[64], p2 = b[128], p3 = b[192];
    float4 r = make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z), (p0.w + p1.w) + (p2.w + p3.w));
    if constexpr (WPT == 6) {
        const float4 p4 = b[256], p5 = b[320];
        r.x += p4.x + p5.x; r.y += p4.y + p5.y; r.z += p4.z + p5.z; r.w += p4.w + p5.w;
    }
    return r;
}

#ifndef DDC_ASM_READS
#define DDC_ASM_READS 0     // 1 = the common tile as one hand-ordered line: nine ds_read_b128 back to back, then per K-step its own s_waitcnt, XOR and three products (so the first
                            // products run under the later reads' latency).  Parity green; measured round 5 against the plain loads (interleaved A/B, 2 x 200 steps each): per
                            // stream rates 0.719 / 0.727 ms vs 0.709 / 0.719, one rate 0.673 / 0.674 vs 0.668 / 0.665 -- no gain, a little worse (the SIMD's other wave already
                            // covers that latency).  Kept switchable, off.
#endif
#ifndef DDC_RING_PAD
#define DDC_RING_PAD 32       // bytes between the streams' rings beyond the ring itself (see k_ddc_mfma)
#endif
#ifndef DDC_DIAG
#define DDC_DIAG 0          // experiments (timing only): 1 = DMA ring + barriers only (no LDS reads / math), 2 = math only (ring filled once), 3 = the epilogue waves skip their
                            // K-range, 4 = the fetching waves skip theirs
#endif
struct DdcParams {
    int n_streams;
    long long B;                      // global sample index of the block start (multiple of 1024)
    long long tile_first; int n_tiles, tiles_per_seg;
    long long k_out0;                 // output index stored at out[stream][0]
    int D, nk_used;                   // decimation; K-steps of the window that hold weights at all
    float scale;
    int n_out;                        // outputs of this call: only k_out0 <= 8 tile + o < k_out0 + n_out are stored (the first / last tile of a call may be partial)
    const uint8_t *hist_in;           // != nullptr: the tiles start in the previous blocks' tail (2 KiB per stream in front of the block: ring positions < 0)
    long long two_T; uint8_t *hist_out;   // hist_out != nullptr: the last segment's workgroups keep the block's newest DDC_HIST samples for the next call (k_ddc_save_hist's work)
    int n_lead_store;                 // FUSE: the complex samples of outputs < n_lead_store are stored as well (k_nfm_demod_boundary redoes them after a retune's lead fix-up)
    // PS (a shift rate per stream: one workgroup = ONE stream x 16 time segments, see k_ddc_mfma)
    long long col_bytes; int col_chunks;      // input bytes / 1024-chunks between the starts of consecutive columns (tiles_per_seg tiles; a multiple of a whole chunk)
    size_t frag_stride, tab_pitch;            // v4i per stream in `frags`; streams per chunk row of the seed table
    int tab_len;                              // seed table rows from the call's first chunk on
    const float *scales; const int *corr_row; size_t corr_chunks;
};

// One workgroup = 16 streams x the tiles [t0, t1) of its segment, walked in time order, NT tiles at a time: the workgroup has NT teams of
// 4 waves; team j computes tile NT*g + j of group g (wave w of a team: K-range w), all teams read the same input ring.  NT = 2 puts two waves
// on every SIMD, so that one tile's LDS reads / epilogue overlap the other tile's matrix products (the per-tile chain is serial inside a wave).
//   LDS: 16 ring buffers of RB bytes (pitch RB + 32: a ds_read_b128 is served in four groups of 16 lanes -- {0-3, 12-15, 20-27}, ... -- and with the 16-byte slot of
//   lane (stream, q) = (2 stream + q) mod 16 every group touches 16 different slots; pitch RB + 16 left 4 two-way conflicts per read: 40 % of the kernel's LDS cycles
//   were conflict cycles, SQ_LDS_BANK_CONFLICT) + reduction buffer + prefix table.
//   DMA: a "row-step" fetches the next 1 KiB of all 16 streams (16 / (4 NT) instructions per wave).
// FUSE (the NFM chain, nfm.hip): the reducer does not store the decimated complex samples but demodulates them (fmdemod_quadri_cf | limit_ff) and stores the three
// digit planes the de-emphasis FIR reads.  y[k - 1] of a tile's first output comes from the previous tile of the same workgroup (re-reduced from the other team's
// partial sums, or handed over through `ylast` across groups); the FIRST output of a segment has its predecessor in another workgroup: it is left to
// k_nfm_demod_boundary, for which the segment's first and last complex samples are still stored.
#ifdef DDC_PROF
// diagnostic build (tools/diag_nfm.py): shader-clock cycles per wave summed over the launch: [wave][compute, wait vmcnt, barrier, DMA issue, reduce / demodulate / store, groups]
__device__ unsigned long long g_ddc_prof[16][8];
#define DPROF_T(k) { const long long t_now = __builtin_readcyclecounter(); prof[k] += t_now - t_prev; t_prev = t_now; }
#else
#define DPROF_T(k)
#endif

template <int RBL, int NT, bool FUSE, bool PS>
__global__ __launch_bounds__(64 * DDC_WPT * NT) void k_ddc_mfma(const uint8_t *__restrict__ in, size_t in_pitch, const v4i *__restrict__ frags,
                                                       const float *__restrict__ cum, const float2 *__restrict__ dtab, const float2 *__restrict__ ctab,
                                                       const float2 *__restrict__ corr, float2 *__restrict__ out, size_t out_pitch, DdcParams p, DdcFuse fz)
{
    // FUSE: two waves of every team own the epilogue (roles A and B below) and the others fetch: an LDS-DMA issue stalls its wave for as long as the memory
    // takes once the CU's 64 pieces are in flight (profiles/r3_notes.md), the epilogue is ~1200 cycles on the group's critical path -- a wave with both was the
    // last at every barrier.  Team 0: roles on K-ranges 0 / 2, team 1: on 1 / 3, so every SIMD hosts one role wave (wave id mod 4 = 0, 2 / 3, 1 with six waves per team).
    // Plain front end: the role rotates, the first four waves of a team fetch.
    // PS (csdr_amd_ddc_create_rates: a shift rate per stream): the weights a h D^t belong to ONE stream, so the 16 columns of the B operand are 16 TIME SEGMENTS
    // ("columns") of that stream instead of 16 streams.  Column starts are a whole number of tiles AND of 1024-chunks apart (lcm(8 D, 1024) samples = 64 tiles at D = 50),
    // so a tile's window sits at the same offset inside a chunk in every column: the chunk-boundary variant, the masks and D^e are the workgroup's, as before; only
    // the chunk SEEDS differ per column (and, for a stream whose rate drifts, the corrections).  The post factors of a group (seed x D^e x drift correction per
    // (team, K-range, side, column)) are put together by the role waves -- which never issue LDS-DMA, so their global loads do not disturb a ring -- into an LDS
    // table; every K-range wave picks its two entries up per tile (fewer vector instructions per K-range than the scalar loads and complex products of the
    // shared-rate kernel: the role waves are this kernel's critical path, the others wait for them).  Roles as in the fused kernel, also when the complex samples
    // are stored (!FUSE).  The first column of a call starts in the history buffer, every other one in the block itself; columns behind the block's
    // end re-read column 0 (their outputs are not stored).
    constexpr bool ROLES = FUSE || PS;
    constexpr int WPT = DDC_WPT, NTHR = 64 * WPT * NT;
    constexpr int RB = 1 << RBL, RP = RB + DDC_RING_PAD, NFW = (ROLES ? DDC_FETCHERS : 4) * NT, SPW = 16 / NFW;   // fetching waves; streams fetched per fetching wave in a row-step
    constexpr int PTN = NT * WPT * 2 * 16;                                             // PS: post factors per group
    static_assert(16 % NFW == 0, "rows per fetching wave");
    static_assert(!PS || (WPT == 4 && DDC_FETCHERS == 2), "the per-stream kernel's role waves fill 64 table entries each");
    extern __shared__ float4 lds_raw[];
    uint8_t *lds_in = reinterpret_cast<uint8_t *>(lds_raw);
    float4 *red = reinterpret_cast<float4 *>(lds_in + 16 * RP);                       // [2][NT teams][WPT waves][64 lanes]
    float *lcum = reinterpret_cast<float *>(red + 2 * NT * WPT * 64);                      // the prefix-sum table: a vector load from global memory inside the tile
                                                                                      // loop would need vmcnt(0), i.e. drain the whole DMA ring
    float2 *ylast = reinterpret_cast<float2 *>(lcum + DDC_NGRAN * 16);                // FUSE: [2][16] last sample of the previous group's last tile
    const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, col = lane & 15, q = lane >> 4;
    const int team = wv / WPT, w = wv % WPT;
    const int ra = team & 1, rb = ra + 2;                                              // FUSE: the K-ranges whose waves own the epilogue
    const bool has_role = ROLES && (w == ra || w == rb);                               // wave uniform
    const int rank = w - (w > ra) - (w > rb);                                          // FUSE: index among the team's waves without a role
    const bool fetches = ROLES ? (!has_role && rank < DDC_FETCHERS) : w < 4;
    const int fw = ROLES ? DDC_FETCHERS * team + rank : 4 * team + w;                  // index among the fetching waves
    const int sb = blockIdx.x;                                                         // block of 16 streams; PS: the stream
    float2 *ptab = ylast + 2 * 16;                                                     // PS: [2][PTN]
    const int col0 = PS ? 16 * (int)blockIdx.y : 0;                                    // PS: absolute index of the workgroup's first column
    float scale = p.scale;
    if constexpr (PS) {
        frags += (size_t)sb * p.frag_stride; cum += (size_t)sb * (DDC_NGRAN * 16); dtab += (size_t)sb * DDC_DTAB; ctab += sb;
        scale = p.scales[sb];
        const int crow = p.corr_row ? p.corr_row[sb] : -1;
        corr = (corr && crow >= 0) ? corr + (size_t)crow * p.corr_chunks * 32 : nullptr;
    }
    for (int i = tid; i < DDC_NGRAN * 16; i += NTHR) lcum[i] = cum[i];            // (visible after the barrier that ends the prologue)
    const long long t0 = p.tile_first + (long long)blockIdx.y * p.tiles_per_seg * (PS ? 16 : 1);      // PS: column 0's first tile
    long long t1 = t0 + p.tiles_per_seg; if (t1 > p.tile_first + p.n_tiles) t1 = p.tile_first + p.n_tiles;
    if (t0 >= t1) return;
    const int n_it = (int)(t1 - t0), n_grp = (n_it + NT - 1) / NT;
    const int last_stream = p.n_streams - 1;
    // ---- weights of this wave's K-range: once per workgroup
    v4i A[DDC_NKW * 3];
    {
        const v4i *fa = frags + (size_t)(DDC_NKW * w) * 3 * 64 + lane;
#pragma unroll
        for (int s = 0; s < DDC_NKW * 3; s++) A[s] = fa[s * 64];
    }
    int nact = p.nk_used - DDC_NKW * w; if (nact > DDC_NKW) nact = DDC_NKW; if (nact < 0) nact = 0;    // K-steps of this wave that hold weights
    nact = __builtin_amdgcn_readfirstlane(nact);
    const int g0 = 2 * DDC_NKW * w;
    float cfull[4], cg0[4], cg1[4];                                                   // offset constants of rows 4q .. 4q+3: whole K-range, and the prefix values at its ends
    {
        const float4 a = *reinterpret_cast<const float4 *>(cum + (size_t)g0 * 16 + 4 * q);
        const float4 b = *reinterpret_cast<const float4 *>(cum + (size_t)(g0 + 2 * DDC_NKW) * 16 + 4 * q);
        cg0[0] = a.x; cg0[1] = a.y; cg0[2] = a.z; cg0[3] = a.w; cg1[0] = b.x; cg1[1] = b.y; cg1[2] = b.z; cg1[3] = b.w;
#pragma unroll
        for (int r = 0; r < 4; r++) cfull[r] = cg1[r] - cg0[r];
    }
    // ---- DMA state
    const int tstride = 16 * p.D;                                                    // bytes of input per tile
    const long long org = PS ? (long long)col0 * p.col_bytes : 0LL;                  // PS: positions are counted from the start of the workgroup's column 0
    long long wg = t0 * tstride - 2 * p.B - org;                                     // window start of the current GROUP's first tile, bytes from the block start
    const long long F0 = wg & ~1023LL;                                               // (floor: a call's first tiles start in the history, at negative positions)
    long long F_end = ((t1 - 1) * tstride - 2 * p.B - org + DDC_WIN + 1023) & ~1023LL;
    if (p.hist_in && F_end > p.two_T - org) F_end = p.two_T - org;                   // a partial last tile's window reaches beyond the block: those bytes feed no stored output
    long long F = F0;                                                                // next row-step
    const uint32_t lds_in_addr = (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t *)lds_in;
    uint32_t voff[SPW], voff_h[SPW];
    long long row_lim[SPW];                                                          // PS: bytes of the block from the row's column start on (wave uniform)
#pragma unroll
    for (int r = 0; r < SPW; r++) {
        if constexpr (PS) {
            voff[r] = (uint32_t)((long long)(SPW * fw + r) * p.col_bytes) + 16u * lane;      // relative to column 0 of the workgroup
            voff_h[r] = 16u * lane;
            row_lim[r] = p.two_T - (long long)(col0 + SPW * fw + r) * p.col_bytes;
        } else {
            const int srow = max(min(sb * 16 + SPW * fw + r, last_stream) - sb * 16, 0); // rows past the last stream re-read it (results discarded)
            voff[r] = (uint32_t)srow * (uint32_t)in_pitch + 16u * lane;
            voff_h[r] = (uint32_t)srow * (uint32_t)(2 * DDC_HIST) + 16u * lane;
            row_lim[r] = 0;
        }
    }
    const uint8_t *sblock = PS ? in + (long long)sb * (long long)in_pitch + (long long)col0 * p.col_bytes : in + (long long)sb * 16 * (long long)in_pitch;
    const uint32_t voff_c0 = 16u * lane;
    auto row_step = [&]() {
        if constexpr (PS) {
            // positions F are column 0's, counted from ITS start (for the call's very first column that is the block start: F < 0 = the history buffer);
            // the other rows read the same position of their own column, which lies inside the block also for F < 0
            const uint32_t ldst = lds_in_addr + (SPW * fw) * RP + (uint32_t)(F & (RB - 1));
            if (F < 0 && col0 == 0) {                                                // (wave uniform; at most three row-steps of the call's first workgroups)
#pragma unroll
                for (int r = 0; r < SPW; r++) {
                    const bool hrow = SPW * fw + r == 0 || F + 1024 > row_lim[r];      // the call's first column; columns behind the block's end (any readable bytes)
                    const uint8_t *sbase = hrow ? p.hist_in + (size_t)sb * (2 * DDC_HIST) + (F < -2 * DDC_HIST ? 0 : F + 2 * DDC_HIST) : sblock + F;
                    const uint32_t vo = hrow ? voff_h[r] : voff[r];
                    const uint32_t la = __builtin_amdgcn_readfirstlane((int)(ldst + r * RP));
                    uint32_t keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(vo), "s"(sbase), "s"(la) : "memory");
                }
            } else {
                uint32_t vo[SPW];
#pragma unroll
                for (int r = 0; r < SPW; r++) vo[r] = F + 1024 <= row_lim[r] ? voff[r] : voff_c0;      // behind the block's end: column 0's bytes again
                dma_rows<SPW, RP>(vo, sblock + F, __builtin_amdgcn_readfirstlane((int)ldst));
            }
            F += 1024;
            return;
        }
        // F < 0: a run of the history (2 KiB per stream = positions [-2048, 0); a window may start up to 7 D samples in front of that: those bytes feed only
        // outputs the previous call has already delivered -- any readable address will do)
        const bool head = F < 0;                                                     // wave uniform; two straight-line copies (a select between the two offset arrays sent them to scratch)
        const uint32_t ldst = lds_in_addr + (SPW * fw) * RP + (uint32_t)(F & (RB - 1));
        if (head) {
            const uint8_t *sbase = p.hist_in + (size_t)sb * 16 * (2 * DDC_HIST) + (F < -2 * DDC_HIST ? 0 : F + 2 * DDC_HIST);
            dma_rows<SPW, RP>(voff_h, sbase, __builtin_amdgcn_readfirstlane((int)ldst));
        } else dma_rows<SPW, RP>(voff, sblock + F, __builtin_amdgcn_readfirstlane((int)ldst));
        F += 1024;
    };
    // every fetching wave issues exactly SPW VMEM loads per row-step and they return in order: vmcnt(SPW n) leaves at most the n newest row-steps in flight
    auto wait_newer = [&](long long newer) {
        switch ((int)newer) {
            case 0: wait_vmcnt<0>(); break;
            case 1: wait_vmcnt<SPW>(); break;
            case 2: wait_vmcnt<SPW * 2>(); break;
            case 3: wait_vmcnt<SPW * 3>(); break;
            case 4: wait_vmcnt<SPW * 4>(); break;
            case 5: wait_vmcnt<SPW * 5>(); break;
            case 6: wait_vmcnt<SPW * 6>(); break;
            default: wait_vmcnt<SPW * 7>(); break;
        }
    };
    auto wait_for = [&](long long last_window_start) {                               // everything below last_window_start + DDC_WIN has landed
        const long long need_end = (last_window_start + DDC_WIN + 1023) & ~1023LL;
        long long newer = (F - need_end) >> 10;                                      // row-steps issued beyond what the group needs
        if (newer < 0) newer = 0;
        if (newer > 7) newer = 7;
        wait_newer(newer);
    };
    // PS: the post factors of a group (window start of its first tile: wgn): this role wave's 64 of the PTN entries -- (K-range 2 role + lane / 32, side, column) of its
    // own team.  A vector load of this kernel queues behind the CU's LDS-DMA pieces: ~3000 cycles (per-wave cycle profile, -DDDC_PROF: issued at the top of a group
    // and used at its end the wait cost the role waves ~900 cycles on their way to the barrier).  So the three table loads of group g + 3 are issued right behind
    // barrier g and stay in flight for a whole group; behind barrier g + 1 they are multiplied (ps_mul -> psP) and the registers reloaded; psP is written in front of
    // barrier g + 2.  The loads are inline asm with the wait counted by hand: written as plain loads the compiler copied the destination registers at the first merge
    // of control flow -- a wait for the full latency right behind the issue (the same finding as in fastddc_mfma.hip's fold).  A role wave's only other vector memory
    // operations are its epilogue's stores, a group old at the wait.  (Measured alternatives, profiles/r4_notes.md: the seeds alone through an LDS table and the
    // products in the K-range waves -- slower, the role waves' wait is what counts; two register sets two groups ahead -- the wait then needs the number of stores
    // the epilogue in between has issued.)
    typedef float ps_v2f __attribute__((ext_vector_type(2)));
    ps_v2f psC = {1.f, 0.f}, psD = {1.f, 0.f}, psK = {1.f, 0.f};
    float2 psP = make_float2(1.f, 0.f);
    const int ps_e = (2 * team + (w == rb ? 1 : 0)) * 64 + lane;
    int ps_tab_pitch = PS ? (int)p.tab_pitch : 0, ps_tab_len = p.tab_len;            // (seed tables hold < 2^31 entries; opaque copies as below: no s_load inside the loop)
    if constexpr (PS) asm volatile("" : "+v"(ps_tab_pitch), "+v"(ps_tab_len));
    auto ps_load = [&](long long wgn) {
        const int tcol = ps_e & 15, side = (ps_e >> 4) & 1, w2 = (ps_e >> 5) & 3;
        const long long n02 = p.B + ((wgn + (long long)team * tstride) >> 1);       // (wave uniform; ddc_wave_geom in 32-bit lane arithmetic)
        const int o0 = (int)(n02 & 1023), chunk0 = (int)((n02 >> 10) - (p.B >> 10));
        const int so = o0 + 32 * DDC_NKW * w2, off2 = so & 1023;
        const int chunk_rel2 = chunk0 + (so >> 10) + (col0 + tcol) * p.col_chunks;
        const bool two2 = off2 + 32 * DDC_NKW > 1024;
        const int e02 = off2 - 32 * DDC_NKW * w2;
        int ci = side ? chunk_rel2 + 2 : max(chunk_rel2 + 1, 0);
        ci = min(ci, ps_tab_len - 1);                                                // (columns behind the block's end; groups behind the segment's last)
        const int ex = side ? e02 - 1024 : e02;
        const int cj = side ? max(off2 + 32 * DDC_NKW - 1024, 0) / 2 : off2 + (two2 ? (1024 - off2) / 2 : 16 * DDC_NKW);
        const float2 *pc = ctab + (size_t)((unsigned)ci * (unsigned)ps_tab_pitch), *pd = dtab + (ex + 2048);      // (ci >= 0; the table has < 2^32 entries)
        asm volatile("global_load_dwordx2 %0, %1, off" : "+v"(psC) : "v"(pc) : "memory");
        asm volatile("global_load_dwordx2 %0, %1, off" : "+v"(psD) : "v"(pd) : "memory");
        if (corr) { const float2 *pk = corr + ((size_t)ci * 32 + (cj >> 5)); asm volatile("global_load_dwordx2 %0, %1, off" : "+v"(psK) : "v"(pk) : "memory"); }
    };
    auto ps_mul = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(psC), "+v"(psD), "+v"(psK) :: "memory");
        float2 P = cmulf(make_float2(psC.x, psC.y), make_float2(psD.x, psD.y));
        if (corr) P = cmulf(P, make_float2(psK.x, psK.y));
        psP = P;
    };
    auto ps_store = [&](int g) { ptab[(g & 1) * PTN + ps_e] = psP; };
    if (PS && has_role) {
        ps_load(wg); ps_mul(); ps_store(0);
        ps_load(wg + (long long)NT * tstride); ps_mul();                             // psP = group 1
        ps_load(wg + 2LL * NT * tstride);                                            // in flight: group 2
    }
    if (fetches) {
        while (F < F_end && F + 1024 <= wg + RB) row_step();
        wait_for(wg + (long long)(NT - 1) * tstride);
    }
    __syncthreads();
    const uint8_t *lrow = lds_in + col * RP;
    const int kk_seg = (int)(8 * t0 - p.k_out0);                                     // output index of the segment's first tile's first output (< 0: a partial first tile)
    // ---- round 5: what the epilogue needs of the kernel's arguments, in registers of the lane that uses it.  The kernel holds 106 scalar registers and spilled 26; the
    // compiler re-read these arguments from memory INSIDE the tile loop (six s_load + s_waitcnt lgkmcnt(0) per tile in the role waves, which are the last at every
    // barrier) and rebuilt three 64-bit plane addresses per store.  Opaque copies ("+v") cannot be rematerialised from the argument block.
    typedef float e_v2f __attribute__((ext_vector_type(2))); typedef __attribute__((address_space(1))) e_v2f *gp_f2; typedef __attribute__((address_space(1))) int8_t *gp_i8;      // (global pointers: behind an opaque asm a generic pointer means flat_store)
    const int e_stream = PS ? sb : sb * 16 + col;
    gp_f2 e_ybase = (gp_f2)(out + (size_t)e_stream * out_pitch);                     // y of output kk at e_ybase[kk]
    gp_i8 e_pl0 = nullptr, e_pl1 = nullptr, e_pl2 = nullptr;
    float e_max_amp = 0.f, e_q_per_amp = 0.f;
    int e_n_out = p.n_out, e_n_lead = p.n_lead_store;
    if constexpr (FUSE) {
        e_pl0 = (gp_i8)(fz.planes + (size_t)e_stream * fz.dl_pitch + fz.dl_fill); e_pl1 = e_pl0 + fz.plane_bytes; e_pl2 = e_pl1 + fz.plane_bytes;
        e_max_amp = fz.max_amp; e_q_per_amp = fz.q_per_amp;
        asm volatile("" : "+v"(e_pl0), "+v"(e_pl1), "+v"(e_pl2), "+v"(e_max_amp), "+v"(e_q_per_amp));
    }
    asm volatile("" : "+v"(e_ybase), "+v"(e_n_out), "+v"(e_n_lead));
#ifdef DDC_PROF
    long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = __builtin_readcyclecounter();
#endif
    // Round 5, measured and NOT kept (profiles/r5_notes.md; the code of the experiments is in git history, commit "ddc: staggered teams ..."): the two teams in different
    // phases (team 0 runs a group's epilogue / ring refill only after its next tile's products: shared-rate kernel 0.673 -> 0.670 ms, per-stream 0.725 -> 0.868), a
    // tile's LDS reads issued in front of the previous group's epilogue (+3 %), s_setprio for the role waves (+-0).
    for (int gi = 0; gi < n_grp; gi++, wg += (long long)NT * tstride) {
        const int it = gi * NT + team;                                               // this team's tile of the group
        const bool active = it < n_it;
        const long long ws = wg + (long long)team * tstride;                         // window start of this team's tile
#if DDC_DIAG == 1
        float4 part = make_float4((float)it, 0.f, 0.f, 0.f);
#else
        float4 part = make_float4(0.f, 0.f, 0.f, 0.f);
        if (active && nact > 0 && !(DDC_DIAG == 3 && has_role) && !(DDC_DIAG == 4 && fetches)) {   // (nact is loop invariant: waves beyond a short window have nothing to add)
            const long long n0 = p.B + (ws >> 1);                                    // global index of the tile's first sample
            const WaveGeom g = ddc_wave_geom(n0, w);
            const int kb = __builtin_amdgcn_readfirstlane(g.kb), half = __builtin_amdgcn_readfirstlane(g.half);
            const int chunk_rel = (int)(g.chunk - (p.B >> 10));                      // -1: the previous block's last chunk (history); -2: in front of it (feeds only outputs that are not stored)
            const int ci0 = max(chunk_rel + 1, 0);                                   // table row of side 0
            // ---- B fragments of this wave's K-range from the ring
            // (round 5, SQ counters in profiles/r5_nfm_pmc_issue.json: ~200 vector instructions per 27 matrix products and wave.  The ring position of a K-step was three
            // vector instructions -- add, mask, add -- in front of every read: 27 per tile; the wave's 9 reads only wrap around the ring's end in 1 tile of 14, so the
            // common case is ONE address and nine immediate offsets.)
            const int b0 = __builtin_amdgcn_readfirstlane((int)(ws & (RB - 1)) + 64 * DDC_NKW * w);      // wave uniform
            v4i Bf[DDC_NKW];
            v4i acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}, snap[3];
            const bool lo_lane = half && q < 2;
            const bool two = kb < DDC_NKW;                                            // == g.two (ddc_wave_geom), as the branch the chains were picked by
            const bool fast = b0 + 64 * DDC_NKW <= RB;                                // (16 q <= 48 stays inside the last K-step's 64 bytes)
#if DDC_ASM_READS
            if (fast && !two) {
                // The common tile (no ring wrap inside the wave's nine reads, no chunk boundary in its K-range: two tiles in three) as ONE straight line: nine reads
                // issued back to back, then per K-step its own wait, its XOR and its three products -- the first products run under the later reads' latency.
                // (Reads as plain loads were followed by one s_waitcnt lgkmcnt(0) and all 36 XORs in front of the first product.)
                const uint32_t pa = (uint32_t)(size_t)(__attribute__((address_space(3))) const uint8_t *)(lrow + b0 + 16 * q);
#pragma unroll
                for (int ks = 0; ks < DDC_NKW; ks++) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(Bf[ks]) : "v"(pa), "n"(64 * ks) : "memory");
#pragma unroll
                for (int ks = 0; ks < DDC_NKW; ks++) {
                    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(Bf[ks]) : "n"(DDC_NKW - 1 - ks));
                    Bf[ks] ^= (int)0x80808080;
#pragma unroll
                    for (int l = 0; l < 3; l++) acc[l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[ks * 3 + l], Bf[ks], acc[l], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);                               // (without it the scheduler pulls six of the nine waits in front of the first product)
                }
            } else
#endif
            {
                if (fast) {
                    const uint8_t *pb = lrow + b0 + 16 * q;
#pragma unroll
                    for (int ks = 0; ks < DDC_NKW; ks++) Bf[ks] = *reinterpret_cast<const v4i *>(pb + 64 * ks);
                    asm volatile("" ::: "memory");                                   // (keeps these reads in this branch: merged with the other branch's they get nine computed addresses again)
#pragma unroll
                    for (int ks = 0; ks < DDC_NKW; ks++) Bf[ks] ^= (int)0x80808080;
                } else {
                    const int base = b0 + 16 * q;
#pragma unroll
                    for (int ks = 0; ks < DDC_NKW; ks++) Bf[ks] = *reinterpret_cast<const v4i *>(lrow + ((base + 64 * ks) & (RB - 1))) ^ (int)0x80808080;
                }
                // ---- one accumulator chain per digit; snapshot at the chunk boundary.  (round 5: the tile without a boundary in this K-range -- 72 % of them -- is its own
                // straight-line path: merged with the ten boundary variants it paid 12 moves for a snapshot it does not have.)
                if (!two) ddc_chain<DDC_NKW>(A, Bf, lo_lane, acc, snap);               // (snap untouched and unused)
                else {
#pragma unroll
                    for (int l = 0; l < 3; l++) snap[l] = v4i{0, 0, 0, 0};
                    switch (kb) {
                        case 0: ddc_chain<0>(A, Bf, lo_lane, acc, snap); break;
                        case 1: ddc_chain<1>(A, Bf, lo_lane, acc, snap); break;
                        case 2: ddc_chain<2>(A, Bf, lo_lane, acc, snap); break;
                        case 3: ddc_chain<3>(A, Bf, lo_lane, acc, snap); break;
                        case 4: ddc_chain<(4 < DDC_NKW ? 4 : DDC_NKW)>(A, Bf, lo_lane, acc, snap); break;
                        case 5: ddc_chain<(5 < DDC_NKW ? 5 : DDC_NKW)>(A, Bf, lo_lane, acc, snap); break;
                        case 6: ddc_chain<(6 < DDC_NKW ? 6 : DDC_NKW)>(A, Bf, lo_lane, acc, snap); break;
                        case 7: ddc_chain<(7 < DDC_NKW ? 7 : DDC_NKW)>(A, Bf, lo_lane, acc, snap); break;
                        default: ddc_chain<(8 < DDC_NKW ? 8 : DDC_NKW)>(A, Bf, lo_lane, acc, snap); break;
                    }
                }
            }
            // ---- this wave's share of rows 4q .. 4q+3 = (Re, Im) of outputs 2q and 2q+1, after the post factors
            const float2 *pt = ptab + (gi & 1) * PTN + ((team * WPT + w) * 2) * 16 + col;      // PS: prepared by the role waves
            float2 P0;
            const int off = (int)((n0 + 32LL * DDC_NKW * w) & 1023);
            if constexpr (PS) P0 = pt[0];
            else {
                P0 = cmulf(ctab[ci0], dtab[g.e0 + 2048]);
                // optional per-chunk correction (rates for which the reference's float recurrence drifts away from C_m D^k): sampled at the
                // centre of the K-range's part in each chunk
                if (corr) P0 = cmulf(P0, corr[ci0 * 32 + ((off + (two ? (1024 - off) / 2 : 16 * DDC_NKW)) >> 5)]);
            }
            float u[4];
            if (two) {
                const float4 cb = *reinterpret_cast<const float4 *>(lcum + g.gb * 16 + 4 * q);
                const float cbv[4] = {cb.x, cb.y, cb.z, cb.w};
                float2 P1;
                if constexpr (PS) P1 = pt[16];
                else {
                    P1 = cmulf(ctab[chunk_rel + 2], dtab[g.e0 - 1024 + 2048]);
                    if (corr) P1 = cmulf(P1, corr[(chunk_rel + 2) * 32 + (((off + 32 * DDC_NKW - 1024) / 2) >> 5)]);
                }
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    u[r] = fmaf(combine_digits(snap[0][r], snap[1][r], snap[2][r]), scale, cbv[r] - cg0[r]);
                    v[r] = fmaf(combine_digits(acc[0][r] - snap[0][r], acc[1][r] - snap[1][r], acc[2][r] - snap[2][r]), scale, cg1[r] - cbv[r]);
                }
                part.x = P0.x * u[0] - P0.y * u[1] + (P1.x * v[0] - P1.y * v[1]); part.y = P0.x * u[1] + P0.y * u[0] + (P1.x * v[1] + P1.y * v[0]);
                part.z = P0.x * u[2] - P0.y * u[3] + (P1.x * v[2] - P1.y * v[3]); part.w = P0.x * u[3] + P0.y * u[2] + (P1.x * v[3] + P1.y * v[2]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) u[r] = fmaf(combine_digits(acc[0][r], acc[1][r], acc[2][r]), scale, cfull[r]);
                part.x = P0.x * u[0] - P0.y * u[1]; part.y = P0.x * u[1] + P0.y * u[0];
                part.z = P0.x * u[2] - P0.y * u[3]; part.w = P0.x * u[3] + P0.y * u[2];
            }
        }
#endif
        float4 *rbuf = red + ((gi & 1) * NT + team) * (WPT * 64);
        rbuf[w * 64 + lane] = part;
        if (PS && has_role && gi + 1 < n_grp) ps_store(gi + 1);
        // ---- the next group's windows must have landed before anyone passes the barrier; the ring space behind them is refilled right after
        const long long wg_n = wg + (long long)NT * tstride;
        DPROF_T(0)
#if DDC_DIAG != 2
        if (fetches && gi + 1 < n_grp) wait_for(wg_n + (long long)(NT - 1) * tstride);
#endif
        DPROF_T(1)
        __syncthreads();
        DPROF_T(2)
#if DDC_DIAG != 2
        if (fetches && gi + 1 < n_grp) { while (F < F_end && F + 1024 <= wg_n + RB) row_step(); }
#endif
        DPROF_T(3)
        // ---- reduction of the four K-range shares and store.  Plain front end: the waves of a team take turns.  FUSE: TWO fixed waves of the team share the epilogue
        // -- role A demodulates the tile's even outputs (it needs the predecessor logic), role B the odd ones; fmdemod_quadri_cf | limit_ff + the digit split is two
        // thirds of the work --, the other two waves fetch.  Both roles sum the same partials in the same order: the values are those of a one-wave epilogue, bit for bit.
        const bool role_a = ROLES ? w == ra : w == gi % WPT, role_b = ROLES && w == rb;
        if (PS && has_role) { ps_mul(); ps_load(wg + 3LL * NT * tstride); }          // psP = group gi + 2; in flight: group gi + 3
#ifndef DDC_NOSTORE
#define DDC_NOSTORE 0       // experiment (timing only): 1 = no epilogue, nothing stored (what the stores cost the input stream)
#endif
        if (active && (role_a || role_b) && !DDC_NOSTORE) {
            const float4 yy = ddc_reduce<WPT>(rbuf + lane);                           // (y0.re, y0.im, y1.re, y1.im): outputs 2q, 2q + 1 of stream col
            const int stream = e_stream;
            const float2 y0 = make_float2(yy.x, yy.y), y1 = make_float2(yy.z, yy.w);
            const int kk = kk_seg + 8 * it + 2 * q + (PS ? col * 8 * p.tiles_per_seg : 0);  // index of y0 in the call's outputs; the first / last tile of a call may be partial
            const bool ok0 = (unsigned)kk < (unsigned)e_n_out, ok1 = (unsigned)(kk + 1) < (unsigned)e_n_out;
            if (!FUSE) {
                if (stream < p.n_streams) {
                    gp_f2 dst = e_ybase + kk;
                    if (ok0 && (!PS || role_a)) dst[0] = e_v2f{y0.x, y0.y};           // (PS: the two role waves share the stores)
                    if (ok1 && (!PS || role_b)) dst[1] = e_v2f{y1.x, y1.y};
                }
            } else if (role_a) {
                // predecessor of output 2q: lane (col, q - 1)'s second output; for q = 0 the previous tile's last output
                float2 prev = make_float2(__shfl_up(y1.x, 16), __shfl_up(y1.y, 16));
                if (q == 0 && it > 0) {
                    if (team > 0) {                                                  // same group, previous team: its partial sums are complete (same barrier)
                        const float4 pv = ddc_reduce<WPT>(red + ((gi & 1) * NT + team - 1) * (WPT * 64) + 48 + col);
                        prev = make_float2(pv.z, pv.w);
                    } else prev = ylast[((gi - 1) & 1) * 16 + col];                  // previous group's last team: handed over (double buffered: the writer of this group is on its way)
                }
                if (team == NT - 1 && q == 3) ylast[(gi & 1) * 16 + col] = y1;
                if (stream < p.n_streams) {
                    // complex samples k_nfm_demod_boundary needs: the call's first output and every segment's first one (their predecessors live elsewhere), the segments'
                    // last ones (the next segment's predecessor) and the call's last one (the next call's)
                    gp_f2 ydst = e_ybase + kk;
                    const bool seg_first = it == 0 && q == 0;
                    if (ok0 && (seg_first || kk == 0 || kk == e_n_out - 1 || kk < e_n_lead)) ydst[0] = e_v2f{y0.x, y0.y};
                    if (ok0 && !seg_first && kk != 0) {
                        int dg[3];
                        nfm_demod_digits(y0, prev, e_max_amp, e_q_per_amp, dg);
                        e_pl0[kk] = (int8_t)dg[0]; e_pl1[kk] = (int8_t)dg[1]; e_pl2[kk] = (int8_t)dg[2];
                    }
                }
            } else {                                                                 // role B: the odd outputs (their predecessor is this lane's own even one)
                if (stream < p.n_streams) {
                    gp_f2 ydst = e_ybase + kk;
                    const bool seg_last = it == n_it - 1 && q == 3;
                    if (ok1 && (seg_last || kk + 1 == 0 || kk + 1 == e_n_out - 1 || kk + 1 < e_n_lead)) ydst[1] = e_v2f{y1.x, y1.y};
                    if (ok1 && kk + 1 != 0) {
                        int dg[3];
                        nfm_demod_digits(y1, y0, e_max_amp, e_q_per_amp, dg);
                        e_pl0[kk + 1] = (int8_t)dg[0]; e_pl1[kk + 1] = (int8_t)dg[1]; e_pl2[kk + 1] = (int8_t)dg[2];
                    }
                }
            }
        }
        DPROF_T(4)
    }
#ifdef DDC_PROF
    if (lane == 0 && wv < 16) { for (int k = 0; k < 5; k++) atomicAdd(&g_ddc_prof[wv][k], (unsigned long long)prof[k]); atomicAdd(&g_ddc_prof[wv][5], (unsigned long long)n_grp); }
#endif
    if (PS && p.hist_out && blockIdx.y + 1 == gridDim.y) {                           // this stream's 2 KiB: the next call's history
        const uint8_t *srow = in + (size_t)sb * in_pitch;
        for (int i = tid; i < 2 * DDC_HIST / 16; i += NTHR)
            *reinterpret_cast<uint4 *>(p.hist_out + (size_t)sb * (2 * DDC_HIST) + 16 * i) = *reinterpret_cast<const uint4 *>(srow + (p.two_T - 2 * DDC_HIST) + 16 * i);
    } else
    if (p.hist_out && blockIdx.y + 1 == gridDim.y) {                                 // 16 streams x 2 KiB: the next call's history
        for (int i = tid; i < 16 * (2 * DDC_HIST / 16); i += NTHR) {
            const int srow = i / (2 * DDC_HIST / 16), piece = i % (2 * DDC_HIST / 16);
            if (sb * 16 + srow < p.n_streams)
                *reinterpret_cast<uint4 *>(p.hist_out + (size_t)(sb * 16 + srow) * (2 * DDC_HIST) + 16 * piece) =
                    *reinterpret_cast<const uint4 *>(sblock + (size_t)srow * in_pitch + (p.two_T - 2 * DDC_HIST) + 16 * piece);
        }
    }
}

#ifdef DDC_PROF
extern "C" int csdr_amd_debug_ddc_prof(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ddc_prof), sizeof(unsigned long long) * 128) != hipSuccess) return -1;
    if (reset) { static unsigned long long z[128]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_ddc_prof), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif

// Plain evaluation of the same model, one WAVE per output (lanes split the taps): outputs [ka0, ka0 + na) and [kb0, kb0 + nb) of every stream.
// Samples before the block come from the history buffer (the previous blocks' last DDC_HIST samples).
struct DirectParams {
    int n_streams; long long B; int T, D, L; long long k_out0, ka0, kb0; int na, nb;
    // a shift rate per stream (0 / nullptr otherwise): seed of chunk row c of stream s at ctab[c * tab_pitch + s], D^k table of stream s at dtab + s * dtab_stride
    size_t tab_pitch, dtab_stride; const int *corr_row; size_t corr_chunks;
    const int *list;                  // != nullptr: blockIdx.y indexes this list of streams (the lead outputs of retuned streams)
    const float2 *dtab_old;           // != nullptr: samples in front of the block were rotated at the rate of [stream]'s table here (a retune at the block boundary)
    const float2 *corr_old;           // with dtab_old: [stream][32] drift corrections of that rate for the chunk in front of the block ((1, 0) where it had none)
};

__global__ __launch_bounds__(256) void k_ddc_direct(const uint8_t *__restrict__ in, size_t in_pitch, const uint8_t *__restrict__ hist,
                                                    const float *__restrict__ taps, const float2 *__restrict__ dtab, const float2 *__restrict__ ctab,
                                                    const float2 *__restrict__ corr, float2 *__restrict__ out, size_t out_pitch, DirectParams p)
{
    const int s = p.list ? p.list[blockIdx.y] : blockIdx.y, lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= p.na + p.nb) return;
    const long long k = idx < p.na ? p.ka0 + idx : p.kb0 + (idx - p.na);
    const uint8_t *row = in + (size_t)s * in_pitch;
    const uint8_t *hrow = hist + (size_t)s * (2 * DDC_HIST);
    const long long c0 = p.B >> 10;
    const size_t cp = p.tab_pitch ? p.tab_pitch : 1;
    const float2 *dtab_o = dtab;
    if (p.tab_pitch) {
        ctab += s; dtab += (size_t)s * p.dtab_stride;
        const int crow = p.corr_row ? p.corr_row[s] : -1;
        corr = (corr && crow >= 0) ? corr + (size_t)crow * p.corr_chunks * 32 : nullptr;
        dtab_o = p.dtab_old ? p.dtab_old + (size_t)s * p.dtab_stride : dtab;
    }
    float ai = 0.f, aq = 0.f;
    // (four taps per lane and round: the loop is a chain of dependent gathers -- sample bytes, chunk phasor, model phasor, correction -- and one round trip per
    // tap made this edge kernel 32 us of the NFM step)
#pragma unroll 4
    for (int t = lane; t < p.L; t += 64) {
        const long long n = (long long)p.D * k + t, rel = n - p.B;
        uint32_t vi, vq;
        if (rel < 0) { vi = hrow[2 * (rel + DDC_HIST)]; vq = hrow[2 * (rel + DDC_HIST) + 1]; }
        else { vi = row[2 * rel]; vq = row[2 * rel + 1]; }
        float2 R = cmulf(ctab[((n >> 10) - c0 + 1) * cp], (rel < 0 ? dtab_o : dtab)[(int)(n & 1023) + 2048]);
        if (rel < 0 && p.dtab_old) R = cmulf(R, p.corr_old[(size_t)s * 32 + ((int)(n & 1023) >> 5)]);
        else if (corr) R = cmulf(R, corr[((n >> 10) - c0 + 1) * 32 + ((int)(n & 1023) >> 5)]);
        const float xi = u8_to_f(vi), xq = u8_to_f(vq);
        const float h = taps[t];
        ai = fmaf(h, xi * R.x - xq * R.y, ai);
        aq = fmaf(h, xq * R.x + xi * R.y, aq);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { ai += __shfl_xor(ai, o, 64); aq += __shfl_xor(aq, o, 64); }
    if (lane == 0) out[(size_t)s * out_pitch + (k - p.k_out0)] = make_float2(ai, aq);
}

// new_hist = last DDC_HIST samples of (old_hist ++ block)
__global__ __launch_bounds__(256) void k_ddc_save_hist(const uint8_t *__restrict__ in, size_t in_pitch, int T, const uint8_t *__restrict__ old_hist,
                                                       uint8_t *__restrict__ new_hist)
{
    const int s = blockIdx.x;
    const uint16_t *src = reinterpret_cast<const uint16_t *>(in + (size_t)s * in_pitch);
    const uint16_t *oh = reinterpret_cast<const uint16_t *>(old_hist + (size_t)s * (2 * DDC_HIST));
    uint16_t *nh = reinterpret_cast<uint16_t *>(new_hist + (size_t)s * (2 * DDC_HIST));
    for (int i = threadIdx.x; i < DDC_HIST; i += 256) {
        const long long rel = (long long)T - DDC_HIST + i;
        nh[i] = rel >= 0 ? src[rel] : oh[rel + DDC_HIST];
    }
}

} // namespace

namespace csdr_amd { long ddc_process_fused(csdr_amd_ddc *d, const uint8_t *in, size_t in_pitch, size_t block_samples, csdr_complexf *out, size_t out_pitch, const DdcFuse *fuse, DdcFuseInfo *info); }

struct csdr_amd_ddc {
    csdr_amd_ctx *ctx;
    int n_streams, D, L;
    float shift_rate;
    size_t max_block;
    float *d_taps; uint8_t *d_hist[2]; int hflip;
    void *d_frags; float *d_cum; float2 *d_dtab, *d_ctab, *d_corr; size_t ctab_cap; bool need_corr;
    float scale; int nk_used;
    bool use_mfma, ended, whole_off;
    float phase; float2 c_prev;
    long long tab_first; bool tab_valid;      // the device tables of chunk seeds (and drift corrections) cover chunks [tab_first, tab_first + ctab_cap)
    long long B, next_k;
    std::string kernel_name;
    bool profiling; std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool; size_t ev_used; double prof_ms; long prof_launches;
    // a shift rate per stream (csdr_amd_ddc_create_rates)
    bool ps; std::vector<float> rates, h_taps; csdr_amd::SeedTables *seeds; float *d_scales;
    float2 *d_dtab_old, *d_corr_old; int *d_list; std::vector<int> retuned;      // streams retuned since the last call: their first outputs straddle two rates (lead fix-up)
    int fallback;                                                     // 1: the last call ran outside the matrix-core kernel (csdr_amd_ddc_fallback)
};

// tables of ONE stream of a per-stream object: weights, prefix sums, scale, D^k
static int ddc_upload_stream_tables(csdr_amd_ddc *d, int s, float rate)
{
    if (d->use_mfma) {
        DdcTable t;
        ddc_build_table(d->D, d->L, rate, d->h_taps.data(), t);
        CSDR_HIP(hipMemcpy((uint8_t *)d->d_frags + (size_t)s * t.frags.size(), t.frags.data(), t.frags.size(), hipMemcpyHostToDevice));
        CSDR_HIP(hipMemcpy(d->d_cum + (size_t)s * t.cum.size(), t.cum.data(), t.cum.size() * sizeof(float), hipMemcpyHostToDevice));
        CSDR_HIP(hipMemcpy(d->d_dtab + (size_t)s * DDC_DTAB, t.dtab.data(), sizeof(float2) * DDC_DTAB, hipMemcpyHostToDevice));
        CSDR_HIP(hipMemcpy(d->d_scales + s, &t.scale, sizeof(float), hipMemcpyHostToDevice));
    } else {
        std::vector<float2> dt; ddc_build_dtab(rate, dt);
        CSDR_HIP(hipMemcpy(d->d_dtab + (size_t)s * DDC_DTAB, dt.data(), sizeof(float2) * DDC_DTAB, hipMemcpyHostToDevice));
    }
    return 0;
}

extern "C" {

static csdr_amd_ddc *ddc_create_impl(csdr_amd_ctx *ctx, int n_streams, const float *rates, bool ps, int decimation, const float *host_taps, int taps_length,
                                     size_t max_block_samples)
{
    if (!ctx || n_streams < 1 || decimation < 1 || taps_length < 1 || !host_taps || !rates) { fail_msg(-3, "ddc_create: bad arguments"); return nullptr; }
    if (taps_length - 1 > DDC_HIST) { fail_msg(-3, "ddc_create: %d taps exceed the %d-sample history", taps_length, DDC_HIST + 1); return nullptr; }
    if (max_block_samples < 1024) max_block_samples = 1024;
    const float shift_rate = rates[0];
    csdr_amd_ddc *d = new csdr_amd_ddc();
    d->ctx = ctx; d->n_streams = n_streams; d->D = decimation; d->L = taps_length; d->shift_rate = shift_rate; d->max_block = max_block_samples;
    d->d_taps = nullptr; d->d_hist[0] = d->d_hist[1] = nullptr; d->d_frags = nullptr; d->d_cum = nullptr; d->d_dtab = nullptr; d->d_ctab = nullptr; d->d_corr = nullptr;
    d->profiling = false; d->ev_used = 0; d->prof_ms = 0; d->prof_launches = 0;
    d->ps = ps; d->seeds = nullptr; d->d_scales = nullptr; d->d_dtab_old = nullptr; d->d_corr_old = nullptr; d->d_list = nullptr; d->fallback = 0;
    d->ctab_cap = 16 * (max_block_samples / 1024 + 8);                 // seeds for 16 calls of the largest block ahead (see ddc_process_fused)
    hipError_t e = hipSuccess;
    auto alloc = [&](void **p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
    const size_t nt = ps ? (size_t)n_streams : 1;                      // table sets
    alloc((void **)&d->d_taps, sizeof(float) * taps_length);
    alloc((void **)&d->d_hist[0], (size_t)2 * DDC_HIST * n_streams);
    alloc((void **)&d->d_hist[1], (size_t)2 * DDC_HIST * n_streams);
    alloc((void **)&d->d_dtab, sizeof(float2) * DDC_DTAB * nt);
    if (!ps) alloc((void **)&d->d_ctab, sizeof(float2) * d->ctab_cap);
    const char *ce = getenv("CSDR_AMD_DDC_CORR");                     // 0 = never, 1 = always, default = only for rates whose float recurrence drifts (model deviation > 2e-6 RMS)
    auto drifts = [&](float r) { return ce ? atoi(ce) != 0 : ddc_rotator_model_rms(r) > 2e-6; };
    d->need_corr = false;
    if (!ps) {
        d->need_corr = drifts(shift_rate);
        if (d->need_corr) alloc((void **)&d->d_corr, sizeof(float2) * d->ctab_cap * 32);
    }
    if (e == hipSuccess) e = hipMemcpy(d->d_taps, host_taps, sizeof(float) * taps_length, hipMemcpyHostToDevice);
    const char *force = getenv("CSDR_AMD_DDC_PATH");                  // "direct" forces the plain kernel (A/B comparisons)
    d->use_mfma = ddc_mfma_supported(decimation, taps_length) && !(force && !strcmp(force, "direct"));
    d->whole_off = getenv("CSDR_AMD_DDC_WHOLE") && atoi(getenv("CSDR_AMD_DDC_WHOLE")) == 0;      // A/B: interior tiles only, the edges on k_ddc_direct (round 2's split)
    d->scale = 0; d->nk_used = 0;
    if (ps) {
        d->rates.assign(rates, rates + n_streams); d->h_taps.assign(host_taps, host_taps + taps_length);
        alloc((void **)&d->d_scales, sizeof(float) * n_streams);
        if (d->use_mfma) {
            d->nk_used = (2 * (7 * decimation + taps_length) + 63) / 64;
            alloc(&d->d_frags, (size_t)DDC_NKT * 3 * 64 * 16 * nt);
            alloc((void **)&d->d_cum, (size_t)DDC_NGRAN * 16 * sizeof(float) * nt);
        }
        for (int s = 0; s < n_streams && e == hipSuccess; s++) {
            const int rc = ddc_upload_stream_tables(d, s, rates[s]);
            if (rc) { csdr_amd_ddc_destroy(d); return nullptr; }
        }
        if (e == hipSuccess) {
            d->seeds = seeds_create(ctx, n_streams, rates, d->d_dtab, DDC_DTAB, max_block_samples);
            if (!d->seeds) { csdr_amd_ddc_destroy(d); return nullptr; }
            std::vector<char> dr(n_streams);
            for (int s = 0; s < n_streams; s++) dr[s] = drifts(rates[s]) ? 1 : 0;
            if (seeds_set_drift(d->seeds, dr)) { csdr_amd_ddc_destroy(d); return nullptr; }
        }
    } else if (d->use_mfma) {
        DdcTable t;
        ddc_build_table(decimation, taps_length, shift_rate, host_taps, t);
        d->scale = t.scale; d->nk_used = (2 * (7 * decimation + taps_length) + 63) / 64;
        alloc(&d->d_frags, t.frags.size());
        alloc((void **)&d->d_cum, t.cum.size() * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(d->d_frags, t.frags.data(), t.frags.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d->d_cum, t.cum.data(), t.cum.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d->d_dtab, t.dtab.data(), sizeof(float2) * DDC_DTAB, hipMemcpyHostToDevice);
    } else {
        std::vector<float2> dt; ddc_build_dtab(shift_rate, dt);
        if (e == hipSuccess) e = hipMemcpy(d->d_dtab, dt.data(), sizeof(float2) * DDC_DTAB, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) { fail(e, "hipMalloc/hipMemcpy(ddc state)", __FILE__, __LINE__); csdr_amd_ddc_destroy(d); return nullptr; }
    d->kernel_name = d->use_mfma ? "k_ddc_mfma" : "k_ddc_direct";
    if (csdr_amd_ddc_reset(d)) { csdr_amd_ddc_destroy(d); return nullptr; }
    return d;
}

csdr_amd_ddc *csdr_amd_ddc_create(csdr_amd_ctx *ctx, int n_streams, float shift_rate, int decimation, const float *host_taps, int taps_length,
                                  size_t max_block_samples)
{
    return ddc_create_impl(ctx, n_streams, &shift_rate, false, decimation, host_taps, taps_length, max_block_samples);
}

csdr_amd_ddc *csdr_amd_ddc_create_rates(csdr_amd_ctx *ctx, int n_streams, const float *shift_rates, int decimation, const float *host_taps, int taps_length,
                                        size_t max_block_samples)
{
    return ddc_create_impl(ctx, n_streams, shift_rates, true, decimation, host_taps, taps_length, max_block_samples);
}

// Retune of one stream: effective from the next call's first sample (a chunk boundary), the phase carries over -- `csdr shift_addition_cc --fifo` (csdr.c:881-923:
// a new rate is picked up between two reads, starting_phase survives).  The first outputs of the next call still reach into samples rotated at the old rate:
// they are recomputed by the plain kernel with both tables (ddc_process_fused).
int csdr_amd_ddc_set_rate(csdr_amd_ddc *d, int stream, float shift_rate)
{
    if (!d->ps) return fail_msg(-3, "ddc_set_rate: the object shares one rate (create it with csdr_amd_ddc_create_rates)");
    if (stream < 0 || stream >= d->n_streams) return fail_msg(-3, "ddc_set_rate: stream %d out of range", stream);
    if (d->rates[stream] == shift_rate) return 0;
    CSDR_HIP(hipStreamSynchronize(d->ctx->stream));                   // calls in flight read this stream's tables
    if (!d->d_dtab_old) CSDR_HIP(hipMalloc((void **)&d->d_dtab_old, sizeof(float2) * DDC_DTAB * (size_t)d->n_streams));
    if (!d->d_corr_old) CSDR_HIP(hipMalloc((void **)&d->d_corr_old, sizeof(float2) * 32 * (size_t)d->n_streams));
    bool listed = false;
    for (int v : d->retuned) listed |= v == stream;
    if (!listed) {                                                    // (two retunes between calls: the history was rotated at the first old rate)
        CSDR_HIP(hipMemcpy(d->d_dtab_old + (size_t)stream * DDC_DTAB, d->d_dtab + (size_t)stream * DDC_DTAB, sizeof(float2) * DDC_DTAB, hipMemcpyDeviceToDevice));
        // ... and the old rate's drift corrections for the chunk in front of the next block
        const float2 *co = seeds_corr_entry(d->seeds, stream, d->B / 1024 - 1);
        if (co) CSDR_HIP(hipMemcpy(d->d_corr_old + (size_t)stream * 32, co, sizeof(float2) * 32, hipMemcpyDeviceToDevice));
        else {
            float2 one[32]; for (int i = 0; i < 32; i++) one[i] = make_float2(1.f, 0.f);
            CSDR_HIP(hipMemcpy(d->d_corr_old + (size_t)stream * 32, one, sizeof one, hipMemcpyHostToDevice));
        }
        d->retuned.push_back(stream);
    }
    const int rc = ddc_upload_stream_tables(d, stream, shift_rate); if (rc) return rc;
    d->rates[stream] = shift_rate;
    const char *ce = getenv("CSDR_AMD_DDC_CORR");
    return seeds_set_rate(d->seeds, stream, shift_rate, ce ? atoi(ce) != 0 : ddc_rotator_model_rms(shift_rate) > 2e-6);
}

float csdr_amd_ddc_get_rate(const csdr_amd_ddc *d, int stream)
{
    if (d->ps) return (stream >= 0 && stream < d->n_streams) ? d->rates[stream] : 0.f;
    return d->shift_rate;
}

int csdr_amd_ddc_fallback(const csdr_amd_ddc *d) { return d->fallback; }

void csdr_amd_ddc_destroy(csdr_amd_ddc *d)
{
    if (!d) return;
    (void)hipStreamSynchronize(d->ctx->stream);
    if (d->seeds) seeds_destroy(d->seeds);
    (void)hipFree(d->d_taps); (void)hipFree(d->d_hist[0]); (void)hipFree(d->d_hist[1]); (void)hipFree(d->d_frags); (void)hipFree(d->d_cum);
    (void)hipFree(d->d_dtab); (void)hipFree(d->d_ctab); (void)hipFree(d->d_corr); (void)hipFree(d->d_scales); (void)hipFree(d->d_dtab_old); (void)hipFree(d->d_corr_old); (void)hipFree(d->d_list);
    for (auto &pr : d->ev_pool) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    delete d;
}

int csdr_amd_ddc_reset(csdr_amd_ddc *d)
{
    d->phase = 0.f; d->c_prev = make_float2(1.f, 0.f); d->B = 0; d->next_k = 0; d->ended = false; d->hflip = 0; d->tab_valid = false; d->tab_first = 0;
    d->retuned.clear();
    if (d->seeds) { const int rc = seeds_reset(d->seeds); if (rc) return rc; }
    CSDR_HIP(hipMemsetAsync(d->d_hist[0], 0x80, (size_t)2 * DDC_HIST * d->n_streams, d->ctx->stream));
    CSDR_HIP(hipMemsetAsync(d->d_hist[1], 0x80, (size_t)2 * DDC_HIST * d->n_streams, d->ctx->stream));
    return 0;
}

const char *csdr_amd_ddc_kernel_name(const csdr_amd_ddc *d) { return d->kernel_name.c_str(); }

int csdr_amd_ddc_set_profiling(csdr_amd_ddc *d, int on)
{
    d->profiling = on != 0; d->ev_used = 0; d->prof_ms = 0; d->prof_launches = 0;
    return 0;
}

int csdr_amd_ddc_kernel_time(csdr_amd_ddc *d, double *total_ms, long *launches)
{
    CSDR_HIP(hipStreamSynchronize(d->ctx->stream));
    for (size_t i = 0; i < d->ev_used; i++) {
        float ms = 0; CSDR_HIP(hipEventElapsedTime(&ms, d->ev_pool[i].first, d->ev_pool[i].second));
        d->prof_ms += ms; d->prof_launches++;
    }
    d->ev_used = 0;
    if (total_ms) *total_ms = d->prof_ms;
    if (launches) *launches = d->prof_launches;
    return 0;
}

long csdr_amd_ddc_process(csdr_amd_ddc *d, const uint8_t *in, size_t in_pitch, size_t block_samples, csdr_complexf *out, size_t out_pitch)
{
    return csdr_amd::ddc_process_fused(d, in, in_pitch, block_samples, out, out_pitch, nullptr, nullptr);
}

} // extern "C"

// The front end's call; fuse != nullptr (the NFM chain): the matrix-core kernel writes the limited demodulator output as digit planes instead of y, *info says
// which outputs are left for k_nfm_demod_boundary (their complex samples are in `out`).
long csdr_amd::ddc_process_fused(csdr_amd_ddc *d, const uint8_t *in, size_t in_pitch, size_t block_samples, csdr_complexf *out, size_t out_pitch, const DdcFuse *fuse, DdcFuseInfo *info)
{
    if (info) { memset(info, 0, sizeof *info); }
    csdr_amd_ctx *c = d->ctx; hipStream_t st = c->stream;
    if (d->ended) return fail_msg(-3, "ddc: stream already ended by a block that was not a multiple of 1024 samples; reset first");
    if (block_samples == 0) return 0;
    if (block_samples > d->max_block) return fail_msg(-3, "ddc: block of %zu samples exceeds max_block_samples %zu", block_samples, d->max_block);
    if (((uintptr_t)in & 1) || (in_pitch & 1)) return fail_msg(-3, "ddc: input pointer and pitch must be 2-byte aligned");
    if (in_pitch < 2 * block_samples && d->n_streams > 1) return fail_msg(-3, "ddc: in_pitch smaller than the block");
    const int T = (int)block_samples;
    // 1. per-chunk phasor seeds C_m = (cos, sin)(starting_phase_m) with the reference's float phase bookkeeping (libcsdr_gpl.c:33-34, 48-51;
    //    chunks of 1024 per csdr.c:911-918), and -- for rates whose recurrence drifts -- the per-chunk corrections.  Both depend on nothing but the shift
    //    rate: the device holds TABLES that run 16 calls ahead of the stream, a call passes an offset (round 2: an upload and a 21-us k_ddc_corr in front of
    //    every call's kernel).  Entry k = chunk tab_first + k; a call needs the chunk in front of its block (history) up to two behind it.
    const size_t nch = ((size_t)T + 1023) / 1024;
    if (nch + 3 > d->ctab_cap) return fail_msg(-3, "ddc: chunk table too small");
    const long long first = d->B / 1024 - 1;
    const float inc = (d->shift_rate * 2) * PI_F;
    int rc = 0;
    SeedView sv; memset(&sv, 0, sizeof sv);
    if (d->ps) {
        // a rate per stream: the same bookkeeping, one lane per stream on a side stream, a few calls ahead (seeds.hip)
        rc = seeds_acquire(d->seeds, first, nch + 3, (T % 1024) ? 0 : nch, &sv); if (rc) return rc;
    } else
    if (!d->tab_valid || first < d->tab_first || first + (long long)nch + 3 > d->tab_first + (long long)d->ctab_cap) {
        float2 *hc = (float2 *)c->pinned_acquire(sizeof(float2) * d->ctab_cap);
        if (!hc) return -2;
        float ph = d->phase;
        hc[0] = d->c_prev;
        for (size_t k = 1; k < d->ctab_cap; k++) {
            hc[k] = make_float2((float)cos((double)ph), (float)sin((double)ph));
            float nx = ph + inc * (float)1024;
            while (nx > PI_F) nx -= 2 * PI_F;
            while (nx < -PI_F) nx += 2 * PI_F;
            ph = nx;
        }
        rc = c->pinned_upload(d->d_ctab, sizeof(float2) * d->ctab_cap); if (rc) return rc;
        if (d->need_corr) {
            hipLaunchKernelGGL(k_ddc_corr, dim3(cdiv(d->ctab_cap, 64)), dim3(64), 0, st, d->d_ctab, d->d_dtab, d->d_corr, (int)d->ctab_cap, (float)cos((double)inc), (float)sin((double)inc));
            CSDR_LAUNCH_CHECK();
        }
        d->tab_first = first; d->tab_valid = true;
    }
    const float2 *ctab = d->ps ? sv.ctab : d->d_ctab + (first - d->tab_first);
    const float2 *corr = d->ps ? sv.corr : d->need_corr ? d->d_corr + (first - d->tab_first) * 32 : nullptr;
    if (!d->ps) {   // the stream's phase behind this block (and the seed of its last chunk, for a table rebuilt at the next call)
        float ph = d->phase;
        for (size_t m = 0; m < nch; m++) {
            if (m + 1 == nch) d->c_prev = make_float2((float)cos((double)ph), (float)sin((double)ph));
            const int len = ((size_t)T - m * 1024 < 1024) ? (int)((size_t)T - m * 1024) : 1024;
            float nx = ph + inc * (float)len;
            while (nx > PI_F) nx -= 2 * PI_F;
            while (nx < -PI_F) nx += 2 * PI_F;
            ph = nx;
        }
        d->phase = ph;
    }
    // 2. outputs that become computable with this block: y[k] needs input up to D k + L - 1
    const long long avail_last = d->B + T - 1;
    long long k_hi = -1;
    if (avail_last - (d->L - 1) >= 0) k_hi = (avail_last - (d->L - 1)) / d->D;
    const long long n_out_ll = k_hi - d->next_k + 1;
    const long n_out = n_out_ll > 0 ? (long)n_out_ll : 0;
    bool hist_saved = false;
    if (n_out > 0) {
        if ((size_t)n_out > out_pitch && d->n_streams > 1) return fail_msg(-3, "ddc: out_pitch %zu smaller than the %ld outputs of this block", out_pitch, n_out);
        const long long k_first = d->next_k;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (d->profiling) {
            if (d->ev_used == d->ev_pool.size()) { hipEvent_t a, b; CSDR_HIP(hipEventCreate(&a)); CSDR_HIP(hipEventCreate(&b)); d->ev_pool.emplace_back(a, b); }
            e0 = d->ev_pool[d->ev_used].first; e1 = d->ev_pool[d->ev_used].second; d->ev_used++;
        }
        // whole tiles (8 outputs) whose window lies inside this block, fetched as 1-KiB runs of whole lines
        long long ta = 0, tb = -1;
        const long long tstride = 16LL * d->D;
        bool whole = false;                                           // the matrix-core kernel takes the whole call (history head, partial tiles): no k_ddc_direct
        if (d->use_mfma && (((uintptr_t)in | in_pitch) & 127) == 0 && in_pitch * 16 + 4096 < ((size_t)1 << 32)) {
            if ((T & 511) == 0 && T >= 2 * DDC_HIST && !d->whole_off) {
                ta = k_first / 8; tb = k_hi / 8; whole = true;           // (a window starts at most 7 D + L - 1 samples in front of the block: inside the 3 head runs)
                if (tb - ta + 1 < 4 || ta * tstride - 2 * d->B < -3072) { ta = 0; tb = -1; whole = false; }
            } else {
                // a ragged / very short block (it ends the stream): whole tiles whose window lies inside this block, fetched as 1-KiB runs of whole lines; the rest below
                ta = (k_first + 7) / 8; tb = (k_hi + 1) / 8 - 1;
                while (ta <= tb && ta * tstride - 2 * d->B < 0) ta++;
                while (tb >= ta && ((tb * tstride - 2 * d->B + DDC_WIN + 1023) & ~1023LL) > 2LL * T) tb--;
                if (tb - ta + 1 < 4) { ta = 0; tb = -1; }
            }
        }
        const int n_cu = current_device_cu_count();
        if (d->ps && !whole) { ta = 0; tb = -1; }                     // a rate per stream: the matrix-core kernel takes whole calls only, everything else is k_ddc_direct's
        // retuned streams: their outputs with D k < B reach into samples that were rotated at the old rate
        long n_fix = 0;
        if (d->ps && !d->retuned.empty()) { n_fix = (long)((d->B + d->D - 1) / d->D - k_first); if (n_fix < 0) n_fix = 0; if (n_fix > n_out) n_fix = n_out; }
        if (tb >= ta) {
            DdcParams p;
            memset(&p, 0, sizeof p);
            p.n_streams = d->n_streams; p.B = d->B; p.tile_first = ta; p.n_tiles = (int)(tb - ta + 1); p.k_out0 = k_first; p.D = d->D; p.nk_used = d->nk_used; p.scale = d->scale;
            p.two_T = 2LL * T; p.hist_out = (T >= DDC_HIST && (T & 7) == 0) ? d->d_hist[d->hflip ^ 1] : nullptr; hist_saved = p.hist_out != nullptr;
            p.n_out = (int)n_out; p.hist_in = whole ? d->d_hist[d->hflip] : nullptr;
            const int n_wsb = (d->n_streams + 15) / 16;
            // 8 KiB ring per stream (window 2304 B + the next group's bytes + 1-KiB fetch granularity on both sides need > 4 KiB), one workgroup per CU.
            // Two teams (512 threads, two waves per SIMD) when the ring also holds a second tile per group: 3 * 16 D <= 3842.
            constexpr int rbl = 13;
            static int teams_env = -1;
            if (teams_env < 0) { const char *e = getenv("CSDR_AMD_DDC_TEAMS"); teams_env = e ? atoi(e) : 2; if (teams_env != 1) teams_env = 2; }
            const int nt = (teams_env == 2 && 3 * 16 * d->D <= 3842) ? 2 : 1;
            int n_seg = (n_cu + n_wsb - 1) / n_wsb; if (n_seg < 1) n_seg = 1;
            if (n_seg > p.n_tiles / 16) n_seg = p.n_tiles / 16;
            if (n_seg < 1) n_seg = 1;
            p.tiles_per_seg = (p.n_tiles + n_seg - 1) / n_seg;
            n_seg = (p.n_tiles + p.tiles_per_seg - 1) / p.tiles_per_seg;
            int n_cols = 0, gx = n_wsb, gy = n_seg;
            if (d->ps) {
                // columns: a whole number of periods lcm(8 D, 1024) samples each, 16 per workgroup; about two workgroups per CU at most
                long long per = 8LL * d->D; { long long a = per, b = 1024; while (b) { const long long t = a % b; a = b; b = t; } per = per / a * 1024; }
                const int tpp = (int)(per / (8 * d->D));               // tiles per period
                const int n_per = (p.n_tiles + tpp - 1) / tpp;
                int n_ss = (2 * n_cu + d->n_streams - 1) / d->n_streams; if (n_ss < 1) n_ss = 1;
                int np = (n_per + 16 * n_ss - 1) / (16 * n_ss); if (np < 1) np = 1;
                p.tiles_per_seg = np * tpp;
                n_cols = (p.n_tiles + p.tiles_per_seg - 1) / p.tiles_per_seg;
                gx = d->n_streams; gy = (n_cols + 15) / 16;
                p.col_bytes = (long long)p.tiles_per_seg * tstride; p.col_chunks = (int)(p.col_bytes / 2048);
                p.frag_stride = (size_t)DDC_NKT * 3 * 64; p.tab_pitch = sv.pitch; p.tab_len = sv.n_entries;
                p.scales = d->d_scales; p.corr_row = sv.corr_row; p.corr_chunks = sv.corr_chunks;
            }
            p.n_lead_store = (fuse && n_fix > 0) ? (int)n_fix + 1 : 0;
            const size_t lds = (size_t)16 * ((1u << rbl) + DDC_RING_PAD) + (size_t)2 * nt * DDC_WPT * 64 * sizeof(float4) + DDC_NGRAN * 16 * sizeof(float) + 2 * 16 * sizeof(float2)
                               + (d->ps ? (size_t)2 * nt * DDC_WPT * 2 * 16 * sizeof(float2) : 0);      // the per-stream kernel's post factors
            DdcFuse fz; memset(&fz, 0, sizeof fz); if (fuse) fz = *fuse;
#define DDC_LAUNCH(NTV, FV, PSV, THREADS) do {                                                                                                             \
                const int arc = lds_attr_once((const void *)k_ddc_mfma<rbl, NTV, FV, PSV>, lds); if (arc) return arc;                                         \
                if (e0) CSDR_HIP(hipEventRecord(e0, st));                                                                                                      \
                hipLaunchKernelGGL((k_ddc_mfma<rbl, NTV, FV, PSV>), dim3(gx, gy), dim3(THREADS), lds, st, in, in_pitch, (const v4i *)d->d_frags, d->d_cum, d->d_dtab, ctab, \
                                   corr, reinterpret_cast<float2 *>(out), out_pitch, p, fz); } while (0)
            if (d->ps) {
                if (fuse) { if (nt == 2) DDC_LAUNCH(2, true, true, 128 * DDC_WPT); else DDC_LAUNCH(1, true, true, 64 * DDC_WPT); }
                else      { if (nt == 2) DDC_LAUNCH(2, false, true, 128 * DDC_WPT); else DDC_LAUNCH(1, false, true, 64 * DDC_WPT); }
            } else {
                if (fuse) { if (nt == 2) DDC_LAUNCH(2, true, false, 128 * DDC_WPT); else DDC_LAUNCH(1, true, false, 64 * DDC_WPT); }
                else      { if (nt == 2) DDC_LAUNCH(2, false, false, 128 * DDC_WPT); else DDC_LAUNCH(1, false, false, 64 * DDC_WPT); }
            }
#undef DDC_LAUNCH
            CSDR_LAUNCH_CHECK();
            if (fuse && info) {
                info->fused = true; info->seg_outputs = 8L * p.tiles_per_seg; info->n_seg = d->ps ? n_cols : n_seg; info->seg_first = (long)(8 * ta - k_first);
                if (whole) { info->n_lead = 8 * ta < k_first ? 1 : 0; info->trail_first = n_out; info->n_trail = 0; }      // lead = the call's first output when its tile is partial
                else { info->n_lead = (long)(8 * ta - k_first); info->trail_first = (long)(8 * (tb + 1) - k_first); info->n_trail = (long)(k_hi - 8 * (tb + 1) + 1); }
                if (info->n_lead < p.n_lead_store) info->n_lead = p.n_lead_store;                                            // (retuned streams: the plain kernel below rewrites their first samples)
            }
            if (e1) CSDR_HIP(hipEventRecord(e1, st));
            d->kernel_name = "k_ddc_mfma";
        } else d->kernel_name = "k_ddc_direct";
        // everything else: leading outputs (history), trailing outputs of the last partial tile, or the whole block
        DirectParams q;
        memset(&q, 0, sizeof q);
        q.n_streams = d->n_streams; q.B = d->B; q.T = T; q.D = d->D; q.L = d->L; q.k_out0 = k_first;
        if (d->ps) { q.tab_pitch = sv.pitch; q.dtab_stride = DDC_DTAB; q.corr_row = sv.corr_row; q.corr_chunks = sv.corr_chunks; }
        if (whole) { q.ka0 = k_first; q.na = 0; q.kb0 = 0; q.nb = 0; }
        else if (tb >= ta) { q.ka0 = k_first; q.na = (int)(8 * ta - k_first); q.kb0 = 8 * (tb + 1); q.nb = (int)(k_hi - q.kb0 + 1); }
        else { q.ka0 = k_first; q.na = (int)n_out; q.kb0 = 0; q.nb = 0; }
        if (q.na + q.nb > 0) {
            if (tb < ta && e0) CSDR_HIP(hipEventRecord(e0, st));
            hipLaunchKernelGGL(k_ddc_direct, dim3(cdiv(q.na + q.nb, 4), d->n_streams), dim3(256), 0, st, in, in_pitch, d->d_hist[d->hflip], d->d_taps, d->d_dtab, ctab, corr,
                               reinterpret_cast<float2 *>(out), out_pitch, q);
            CSDR_LAUNCH_CHECK();
            if (tb < ta && e1) CSDR_HIP(hipEventRecord(e1, st));
        }
        d->fallback = tb >= ta ? 0 : 1;
        if (n_fix > 0) {
            // the first outputs of the streams retuned since the last call, with the old rate's table for the samples in front of the block
            const int nr = (int)d->retuned.size();
            if (!d->d_list) CSDR_HIP(hipMalloc((void **)&d->d_list, sizeof(int) * d->n_streams));
            int *hl = (int *)c->pinned_acquire(sizeof(int) * nr); if (!hl) return -2;
            memcpy(hl, d->retuned.data(), sizeof(int) * nr);
            rc = c->pinned_upload(d->d_list, sizeof(int) * nr); if (rc) return rc;
            DirectParams f = q;
            f.ka0 = k_first; f.na = (int)n_fix; f.kb0 = 0; f.nb = 0; f.list = d->d_list; f.dtab_old = d->d_dtab_old; f.corr_old = d->d_corr_old;
            hipLaunchKernelGGL(k_ddc_direct, dim3(cdiv(f.na, 4), nr), dim3(256), 0, st, in, in_pitch, d->d_hist[d->hflip], d->d_taps, d->d_dtab, ctab, corr,
                               reinterpret_cast<float2 *>(out), out_pitch, f);
            CSDR_LAUNCH_CHECK();
        }
    }
    if (n_out > 0) d->retuned.clear();
    // 3. history for the next block (the matrix-core kernel's last segment has done it when it ran on a whole block)
    if (!hist_saved) hipLaunchKernelGGL(k_ddc_save_hist, dim3(d->n_streams), dim3(256), 0, st, in, in_pitch, T, d->d_hist[d->hflip], d->d_hist[d->hflip ^ 1]);
    CSDR_LAUNCH_CHECK();
    d->hflip ^= 1;
    if (T % 1024) d->ended = true;
    d->B += T; d->next_k += n_out;
    return n_out;
}

extern "C" {

// Test hook (tests/test_ddc_table_cpu.py): evaluates ONE tile on the CPU exactly the way k_ddc_mfma does -- same table, same lane/byte
// layout, same K-range split over four waves, same snapshot / half-K-step handling of chunk boundaries, same post factors -- so the
// table builder and the index arithmetic are validated without a GPU.
//   n0: global index of the tile's first sample (multiple of 16); window: DDC_WIN raw u8 bytes starting at sample n0;
//   ctab: (cos, sin) pairs of the chunks n0 >> 10, +1, +2;  out16: (Re, Im) of the tile's 8 outputs.
int csdr_amd_debug_ddc_mfma_tile(int D, int L, float shift_rate, const float *taps, long long n0, const uint8_t *window, const float *ctab3, float *out16)
{
    if (!ddc_mfma_supported(D, L) || (n0 & 15)) return -1;
    static DdcTable t; static int cD = 0, cL = 0; static float crate = 0; static std::vector<float> ctaps;
    if (cD != D || cL != L || crate != shift_rate || ctaps != std::vector<float>(taps, taps + L)) {
        ddc_build_table(D, L, shift_rate, taps, t); cD = D; cL = L; crate = shift_rate; ctaps.assign(taps, taps + L);
    }
    const int nk_used = (2 * (7 * D + L) + 63) / 64;
    for (int r = 0; r < 16; r++) out16[r] = 0.f;
    for (int w = 0; w < DDC_WPT; w++) {
        const WaveGeom g = ddc_wave_geom(n0, w);
        int nact = nk_used - DDC_NKW * w; if (nact > DDC_NKW) nact = DDC_NKW; if (nact < 0) nact = 0;
        const long long chunk_rel = g.chunk - (n0 >> 10);
        const float2 C0 = make_float2(ctab3[2 * chunk_rel], ctab3[2 * chunk_rel + 1]), C1 = make_float2(ctab3[2 * chunk_rel + 2], ctab3[2 * chunk_rel + 3]);
        const float2 P0 = cmulf(C0, t.dtab[g.e0 + 2048]);
        const float2 P1 = g.two ? cmulf(C1, t.dtab[g.e0 - 1024 + 2048]) : make_float2(0.f, 0.f);
        float u[16], v[16];
        for (int r = 0; r < 16; r++) {
            long acc[3] = {0, 0, 0}, snap[3] = {0, 0, 0};
            auto step = [&](int ks, int kg_lo, int kg_hi) {           // K-step ks of this wave, k-groups (16-byte lane groups) [kg_lo, kg_hi)
                const int K = DDC_NKW * w + ks;
                for (int kg = kg_lo; kg < kg_hi; kg++) for (int b = 0; b < 16; b++) {
                    const int x = (int)(int8_t)(window[64 * K + 16 * kg + b] ^ 0x80);
                    for (int l = 0; l < 3; l++) acc[l] += (long)t.frags[((size_t)(K * 3 + l) * 64 + (16 * kg + r)) * 16 + b] * x;
                }
            };
            for (int ks = 0; ks < nact; ks++) {
                if (ks == g.kb) {
                    if (g.half) step(ks, 0, 2);
                    for (int l = 0; l < 3; l++) snap[l] = acc[l];
                    if (g.half) step(ks, 2, 4); else step(ks, 0, 4);
                } else step(ks, 0, 4);
            }
            const float cg0 = t.cum[(size_t)(2 * DDC_NKW * w) * 16 + r], cg1 = t.cum[(size_t)(2 * DDC_NKW * (w + 1)) * 16 + r];
            if (g.two && g.kb < nact) {
                const float cb = t.cum[(size_t)g.gb * 16 + r];
                u[r] = fmaf(fmaf((float)snap[0], 65536.0f, fmaf((float)snap[1], 256.0f, (float)snap[2])), t.scale, cb - cg0);
                v[r] = fmaf(fmaf((float)(acc[0] - snap[0]), 65536.0f, fmaf((float)(acc[1] - snap[1]), 256.0f, (float)(acc[2] - snap[2]))), t.scale, cg1 - cb);
            } else {
                u[r] = fmaf(fmaf((float)acc[0], 65536.0f, fmaf((float)acc[1], 256.0f, (float)acc[2])), t.scale, cg1 - cg0);
                v[r] = 0.f;
            }
        }
        for (int o = 0; o < 8; o++) {
            out16[2 * o] += P0.x * u[2 * o] - P0.y * u[2 * o + 1] + (P1.x * v[2 * o] - P1.y * v[2 * o + 1]);
            out16[2 * o + 1] += P0.x * u[2 * o + 1] + P0.y * u[2 * o] + (P1.x * v[2 * o + 1] + P1.y * v[2 * o]);
        }
    }
    return 0;
}

} // extern "C"
