// wfm_mfma.hip -- matrix-core front end of the fused WFM chain (u8 IQ -> rotate -> decimating FIR -> FM demod -> de-emphasis -> s16): ONE kernel per call.
//
// Why matrix cores on a "memory-bound DSP" path: per complex input sample the VALU formulation needs
// 2 cvt + 4 fma (u8->float, rotate) + 6.3 fma (the 2-of-5 FIR outputs the audio needs) + LDS traffic of 8 B write /
// 19 B read, i.e. it is VALU/LDS bound at ~11 % of the HBM roofline (profiles/r1_*).  But the whole front end is
// LINEAR in the input bytes with weights that are shared by every stream:
//
//     y[k] = sum_n  a h[n-Dk] R[n] (v_n - 128)  +  const_k ,      v_n = raw u8 I/Q bytes,  a = 2/255
//
// and u8-128 is exactly an int8.  So the rotation, the conversion and the FIR collapse into one banded
// [outputs x bytes] x [bytes x streams] product on v_mfma_i32_16x16x64_i8 with EXACT int32 accumulation:
//   * B operand  = the raw input: lane l holds 16 consecutive bytes of stream (l%16), XOR 0x80 to recentre; no conversion, no rotation
//     instructions at all;
//   * A operand  = the weights a*h*D^t split into three signed base-256 digits (23-bit fixed point; 1.3e-7 end-to-end
//     error measured against the oracle), held in registers and reused for every tile of every stream;
//   * 16 rows    = {Re,Im} x {y[Fj+9], y[Fj+10]} x 4 audio samples -> the quadrature demodulator is lane local.
// The band wastes ~5x MACs, but i8 MFMA has ~50x the rate of the f32 VALU path; the kernel becomes bound by the input stream.
//
// Rotator model: shift_addition_cc restarts its float32 phasor at every 1024-sample chunk from cos/sin of a float
// phase (libcsdr_gpl.c:33-35) and advances it by multiplying with the ROUNDED (cos d, sin d) (:44-45), so inside a
// chunk R[n] = C_m * D[n mod 1024], D[k] = (cosdelta_f32 + j sindelta_f32)^k (evaluated in double; the float
// recurrence's own rounding noise is 8.7e-7 RMS).  Relative to a tile's window base n0, R[n0 + t] = [C_m D^(n0 - 1024 m)] D^t: the weights
// a h D^t are the same for every tile, the bracket is one complex scalar per (tile, chunk), applied AFTER the matrix product (C_m exact per
// chunk from the host's float phase bookkeeping).  A window that contains a chunk boundary accumulates the two sides in ONE accumulator
// chain that is snapshotted at the boundary.
//
// Round 3: the kernel families this file grew through (per-wave kernel with 128 phase-specific weight sets, quad and octet workgroup kernels:
// profiles/r1_notes.md) are gone, and so are the bounds-checked edge launch, k_wfm_tail and k_wfm_save_hist: the sequential kernel takes any
// 16-byte-aligned pitch, starts in the history of the previous block (a 1-KiB head per stream in front of the ring), masks its fetches at a
// ragged end, handles partial tiles at both ends of a call, carries the de-emphasis state in and out and keeps the next call's history itself.
#include "common.hpp"
#include "wfm_mfma.hpp"
#include <hip/hip_ext.h>
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <complex>
#include <vector>
using namespace csdr_amd;

namespace csdr_amd {

bool wfm_mfma_supported(int D, int L, int F)
{
    if ((D * F) & 1) return false;                                  // tile stride 8*D*F bytes must keep 16-byte alignment
    const int win_off = (18 * D) & ~15;
    const int span_bytes = 2 * (D * (3 * F + 10) + L) - win_off;     // last row ends at sample D*(3F+10)+L-1 (tile-relative)
    if (span_bytes > 64 * WFM_NK) return false;                     // window must fit the 8 K-steps (<= 256 samples: at most one chunk boundary)
    if (D + L + 3 * D * F + 8 > WFM_HIST) return false;              // a block's first tile may reach this far into the history
    return true;
}

// Builds the phase-independent weight set of k_wfm_mfma_seq: weights a h D^t relative to the window base, post factors D^e.
//   seq_frags : [WFM_NK][3 digits][64 lanes] int8x16     (lane l: row l%16, K bytes 16*(l/16) .. +15 of that K-step)
//   seq_cum   : [4 WFM_NK + 1][16 rows] float            (prefix sums of the weights at 16-byte granules: the +1/255 offset of u8->float through the filter)
//   dtab      : D^(i - 2048), i in [0, 3072)
void wfm_mfma_build_table(int D, int L, int F, float shift_rate, const float *taps, WfmMfmaTable &t)
{
    t.D = D; t.L = L; t.F = F;
    t.tile_stride_bytes = 8 * D * F;
    t.win_off_bytes = (18 * D) & ~15;
    const float rate2 = shift_rate * 2, inc = rate2 * PI_F;         // libcsdr_gpl.c:83-86
    const float sd = (float)sin((double)inc), cd = (float)cos((double)inc);
    const std::complex<double> d((double)cd, (double)sd);
    const double mag = std::abs(d), ang = std::arg(d);
    const double a = 2.0 / 255.0;
    const int base_off_samples = t.win_off_bytes / 2;
    {
        t.dtab.resize(3072);
        for (int i = 0; i < 3072; i++) { const std::complex<double> v = std::polar(pow(mag, i - 2048), ang * (i - 2048)); t.dtab[i] = make_float2((float)v.real(), (float)v.imag()); }
        const double dmax = fmax(1.0, pow(mag, 32 * WFM_NK));
        double g2 = 0;
        for (int k = 0; k < L; k++) g2 = fmax(g2, fabs(a * (double)taps[k]));
        g2 *= dmax * 1.0001; if (g2 == 0) g2 = 1;
        const double qs = 4194304.0 / g2;
        t.seq_scale = (float)(g2 / 4194304.0);
        t.seq_frags.assign((size_t)WFM_NK * 3 * 64 * 16, 0);
        const int ngr = 4 * WFM_NK;                                   // 16-byte granules of the window
        std::vector<double> gsum((size_t)ngr * 16, 0.0);
        for (int r = 0; r < 16; r++) {
            const int q = r / 4, which = (r % 4) / 2, comp = r % 2;
            const long off = (long)D * (F * q + 9 + which) - base_off_samples;
            for (int tp = 0; tp < L; tp++) {
                const long ts = off + tp;
                const std::complex<double> G = a * (double)taps[tp] * std::polar(pow(mag, (double)ts), ang * (double)ts);
                for (int c = 0; c < 2; c++) {
                    const double val = comp == 0 ? (c == 0 ? G.real() : -G.imag()) : (c == 0 ? G.imag() : G.real());
                    const long colb = 2 * ts + c;
                    const int ks = (int)(colb / 64), b = (int)(colb % 64);
                    long qv = lrint(val * qs);
                    const int w2 = (int)(((qv + 128) % 256 + 256) % 256) - 128; qv = (qv - w2) / 256;
                    const int w1 = (int)(((qv + 128) % 256 + 256) % 256) - 128; qv = (qv - w1) / 256;
                    const int dig[3] = {(int)qv, w1, w2};
                    for (int l = 0; l < 3; l++) t.seq_frags[((size_t)(ks * 3 + l) * 64 + (16 * (b / 16) + r)) * 16 + b % 16] = (int8_t)dig[l];
                    gsum[(size_t)(colb / 16) * 16 + r] += val;
                }
            }
        }
        t.seq_cum.assign((size_t)(ngr + 1) * 16, 0.f);
        for (int r = 0; r < 16; r++) {
            double run = 0;
            for (int g = 0; g <= ngr; g++) { t.seq_cum[(size_t)g * 16 + r] = (float)(0.5 * run); if (g < ngr) run += gsum[(size_t)g * 16 + r]; }
        }
    }
}

} // namespace csdr_amd

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float combine_digits(int a0, int a1, int a2)
{   // exact integers (<= 22 bits each) recombined in float: value = a0*65536 + a1*256 + a2
    return fmaf((float)a0, 65536.0f, fmaf((float)a1, 256.0f, (float)a2));
}

// rows 4q..4q+3 of the tile after the chunk phasors: (Re, Im) of y[Fj+9] and of y[Fj+10];  y = C_m u0 + C_{m+1} u1
__device__ __forceinline__ void tile_rows(const v4i (&acc)[3], const v4i (&snap)[3], bool two, float scale, const float (&k0)[4], const float (&k1)[4],
                                          float2 C0, float2 C1, float &pI, float &pQ, float &cI, float &cQ)
{
    float u0[4];
    if (two) {
        float u1[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            u0[r] = fmaf(combine_digits(snap[0][r], snap[1][r], snap[2][r]), scale, k0[r]);
            u1[r] = fmaf(combine_digits(acc[0][r] - snap[0][r], acc[1][r] - snap[1][r], acc[2][r] - snap[2][r]), scale, k1[r]);
        }
        pI = C0.x * u0[0] - C0.y * u0[1] + (C1.x * u1[0] - C1.y * u1[1]); pQ = C0.x * u0[1] + C0.y * u0[0] + (C1.x * u1[1] + C1.y * u1[0]);
        cI = C0.x * u0[2] - C0.y * u0[3] + (C1.x * u1[2] - C1.y * u1[3]); cQ = C0.x * u0[3] + C0.y * u0[2] + (C1.x * u1[3] + C1.y * u1[2]);
    } else {
#pragma unroll
        for (int r = 0; r < 4; r++) u0[r] = fmaf(combine_digits(acc[0][r], acc[1][r], acc[2][r]), scale, k0[r]);
        pI = C0.x * u0[0] - C0.y * u0[1]; pQ = C0.x * u0[1] + C0.y * u0[0];
        cI = C0.x * u0[2] - C0.y * u0[3]; cQ = C0.x * u0[3] + C0.y * u0[2];
    }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int KB>
__device__ __forceinline__ void wfm_chain(const v4i (&A)[WFM_NK * 3], const v4i (&Bf)[WFM_NK], bool lo_lane, v4i (&acc)[3], v4i (&snap)[3])
{   // straight-line code per boundary position (see ddc_mfma.hip: run-time branches inside the unrolled chain cost hundreds of AGPR moves)
    const v4i z = {0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < WFM_NK; ks++) {
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

// ---------------------------------------------------------------------------------------------------------------------
// The sequential kernel.  One 24-fragment weight set, no wave tied to a tile phase: a workgroup owns 16 streams x a contiguous run of
// tiles (a "segment"), its SEQ_TPG = 6 compute waves take 6 consecutive tiles per step, and the input slides through a per-stream LDS ring
// filled by LDS-DMA in whole 1-KiB runs -- every input byte is fetched exactly once, in long sequential runs per stream (PMC: 1.005 x the
// algorithmic bytes).  A chunk boundary inside the 256-sample window is 16-byte granular (window bases are multiples of 8 samples): the
// boundary K-step is multiplied twice with complementary lane groups of the B operand zeroed, the accumulator chain is snapshotted in
// between; each boundary position is its own straight-line code variant (wfm_chain<KB>).
//
// Who fetches (round 3, profiles/r3_notes.md): SEQ_NLD = 2 LOADER waves, 8 streams each.  The CU accepts 64 LDS-DMA pieces in flight; beyond
// that an issuing wave stalls for as long as the memory takes (160-330 cycles per piece at the kernel's rate).  While the eight waves of the
// earlier layout each fetched two streams, every wave sat in that stall for a quarter to a third of each step (per-wave cycle counts,
// -DWFM_PROF: tools/diag_wfm.py) and did its tile afterwards.  Now the stall belongs to two waves that do nothing else, a row-step is one
// straight piece of code (M0 saved once, stepped by s_add: dma_rows), and the six compute waves run tile -> barrier -> tile.
//
// The back end of the chain runs inside: the workgroup produces its streams' audio in time order, so the one-pole de-emphasis
// (libcsdr.c:1081-1097) is a state carried in wave 0 from step to step, and convert_f_s16 follows one step later out of a ring of SEQ_OUTS
// steps' samples per stream, done by the loader waves, which hold the s16 lines in registers until 5.5 KiB per stream go out together (see "Stores"
// in the kernel): the demodulated audio never goes to HBM as floats.  A segment
// that starts in the middle of a call demodulates two steps (48 audio samples) ahead of its range from zero state without storing them --
// the filter forgets as 0.706^k --; the first segment starts from the exact state the previous call left, the last one leaves its own.
//
// A call's edges, all inside this launch:
//   * history: the first tiles' windows start up to 512 bytes in front of the block.  Ring coordinates run from -1024: the run [-1024, 0) is
//     fetched from a 1-KiB head per stream (bytes 512.. = the previous block's newest 512 bytes; 0x80 = zero samples before the stream),
//     every later run from the block itself;
//   * the last segment's workgroups copy the block's newest 512 bytes into the OTHER head buffer (the next call's history; two buffers,
//     because the first segment of this launch is still reading the current one);
//   * partial tiles: a call's first / last audio sample need not be a multiple of 4: such a step's de-emphasis runs sample by sample over
//     the valid range only (one lane per stream; at most two steps per call), and nothing outside [j_first, j_first + n_audio) is stored;
//   * ragged end (a last block that is not a multiple of 512 samples): the fetch of a run is masked at the row's last 16-byte piece.
struct SeqParams {
    int n_streams; long long B2; long long tile_first; int n_tiles, tiles_per_seg;
    int stride, win_off; float scale;
    long long two_T;                                                                  // bytes per stream of this block
    float alpha; const float *last_in; float *last_out; int16_t *s16; float *af; size_t out_pitch; long long j_first; int n_audio;
    const uint8_t *head_in; uint8_t *head_out;
    // PS (a shift rate per stream, csdr_amd_wfm_create_rates): one workgroup = ONE stream x 16 time segments ("columns", tiles_per_seg tiles each, a whole number
    // of 1024-chunks and of output lines apart), see k_wfm_mfma_seq
    long long col_bytes; int col_chunks, col_audio, n_cols;      // input bytes / chunks / audio samples between column starts; columns of the call
    size_t frag_stride, tab_pitch; int tab_len;                   // v4i per stream in `frags`; streams per chunk row of the seed table; its rows from the call's first chunk on
    const float *scales;
    const float *lead_d; const int *lead_n; int lead_stride;      // retuned streams: lead_n[stream] audio samples at the call's start come from lead_d[stream * lead_stride + i]
};

#ifndef SEQ_RB_KIB
#define SEQ_RB_KIB 9
#endif
constexpr int SEQ_RB = 1024 * SEQ_RB_KIB;  // ring bytes per stream: 9 x 1 KiB (a step's window (SEQ_TPG - 1) x stride + 512 B, the next step's, what is in flight beyond, fetch granularity)
#ifndef SEQ_RING_PAD
#define SEQ_RING_PAD 32
#endif
constexpr int SEQ_RP = SEQ_RB + SEQ_RING_PAD;   // LDS pitch.  A ds_read_b128 is served in four groups of 16 lanes ({0-3, 12-15, 20-27}, ...): with the 16-byte slot of lane
                                           // (stream, q) = (2 stream + q) mod 16 every group touches 16 different slots; + 16 (slot = stream + q) left 4 two-way conflicts per read
constexpr int SEQ_NGR = 4 * WFM_NK;        // 16-byte granules per window
#ifndef SEQ_TPG_N
#define SEQ_TPG_N 6
#endif
#ifndef SEQ_NLD_N
#define SEQ_NLD_N 2
#endif
constexpr int SEQ_TPG = SEQ_TPG_N;         // tiles per step = compute waves per workgroup
constexpr int SEQ_NLD = SEQ_NLD_N;         // loader waves: they issue the LDS-DMA (a wave stalls 160-330 cycles per 1-KiB piece once the CU's 64 pieces are in flight), so that
                                           // the compute waves never wait at an issue, and they collect the finished audio in their otherwise empty registers
constexpr int SEQ_NW = SEQ_TPG + SEQ_NLD;  // waves per workgroup
constexpr int SEQ_NLW = SEQ_NLD;           // waves that fetch
constexpr int SEQ_LINE = 64;               // audio samples stored at a time per stream: one 128-byte line of the s16 output row
constexpr int SEQ_OUTS = (SEQ_LINE - 1 + 3 * 4 * SEQ_TPG + 4 * SEQ_TPG - 1) / (4 * SEQ_TPG);   // audio staging: a ring of this many steps' samples per stream (a step = 4 SEQ_TPG
constexpr int SEQ_OUTR = SEQ_OUTS * 4 * SEQ_TPG;   // samples): written by the tile waves, filtered in place by the de-emphasis one step later, stored one step after that in whole
constexpr int SEQ_OUTP = SEQ_OUTR + 4;     // lines, which may reach SEQ_LINE - 1 samples back; row pitch in floats
static_assert(16 % SEQ_NLW == 0 && (16 / SEQ_NLW) * 7 <= 63 && SEQ_TPG % 2 == 0, "rows per fetching wave (vmcnt counts to 63); the de-emphasis scan works on pairs");
constexpr int SEQ_HEAD = 1024;             // bytes of the per-stream head (one fetch run)

#ifndef WFM_DIAG
#define WFM_DIAG 0      // diagnostic builds (tools/diag_wfm.py; the compute waves still take part in every barrier): 1 = no ring reads (B operand from registers),
#endif                  // 2 = ring reads but no matrix products, 3 = compute waves idle, 4 = 3 + no audio stores, 5 = 4 + no de-emphasis
#ifdef WFM_PROF
// diagnostic build (tools/diag_wfm.sh): shader-clock cycles per wave summed over the launch: [wave][compute, wait vmcnt, barrier, DMA issue, emit, de-emphasis, steps]
__device__ unsigned long long g_wfm_prof[SEQ_NW][8];
__device__ unsigned long long g_wfm_life[SEQ_NW][4];      // per wave, summed over workgroups: cycles from kernel entry to the first step, inside the step loop, behind it; workgroups
#define PROF_T(k) { const long long t_now = __builtin_readcyclecounter(); prof[k] += t_now - t_prev; t_prev = t_now; }
#else
#define PROF_T(k)
#endif

// PS: the weights a h D^t belong to one stream, so the 16 columns of the B operand are 16 TIME SEGMENTS of that stream (ddc_mfma.hip has the same arrangement and the
// reasoning): column starts a whole number of tiles, chunks and 64-sample output lines apart, so that the chunk-boundary variant, D^e and the line bookkeeping stay
// the workgroup's; per column only the chunk seeds differ.  A step's six windows touch three consecutive chunks: compute wave 1 -- no wave of this kernel but
// the loaders issues other vector memory operations -- brings the 3 x 16 seeds of a step into an LDS table two steps ahead (inline-asm loads, one step in flight:
// a vector load queues behind the CU's LDS-DMA pieces for ~3000 cycles).  Every column but the call's first warms its de-emphasis up over two steps like a
// segment of the shared-rate kernel; columns behind the block's end re-read column 0 and store nothing.
// RES (the resident form, csdr_amd_wfm_ring_*: north_star's "persistent-kernel ring buffer"; the reference's unit of work is one the_bufsize block per loop iteration,
// csdr.c:189-193, 232-247, 330-392): ONE grid stays on the GPU and walks a ring of blocks.  The host posts a block by writing its descriptor -- tagged 64-byte lines in
// host-coherent memory: the block's chunk seeds -- ; a workgroup takes the work items (block k, 16-stream group sb) with item = k n_wsb + sb = its id (mod the grid), polls
// the descriptor of its next block (s_sleep between polls), runs this kernel's body on it and counts the item in; the workgroup that completes a block writes the block's
// done line (host memory).  Weights, prefix table and the LDS ring stay with the workgroup.  State between blocks: NONE -- every block warms its de-emphasis up over the
// two steps in front of it, read from the previous block's slot of the input ring (what every segment but the first of a long call does anyway), and its first partial
// tile is recomputed from there; so a workgroup that leaves can be replaced by a new launch at any block boundary: it leaves when the host says stop, when no block
// arrived for idle_ticks, or when the launch is older than life_ticks (a stalled host or a bug cannot hold the GPU), by raising `exiting` for all others and storing its
// next item.  The host relaunches on demand (wfm_ring.hip).
constexpr int RES_WARM = 48;              // audio samples in front of a block over which a retune's de-emphasis state is rebuilt (0.706^48 = 5e-8)
struct ResCtl {
    const uint32_t *desc;                  // host: [n_slots][desc_lines][16]: dword 0 = tag of the block, 1 = lead samples (a retune), 2.. = 7 x (cos, sin) of chunks first_chunk - 4 ...
    const uint32_t *ctrl;                  // host: [0] = stop
    uint32_t *done;                        // host: [n_slots][16]: tag, n_audio, t_first (2), t_done (2)
    unsigned *cnt;                         // device [n_slots]: items of the block in that slot that are finished
    unsigned long long *t_first;           // device [n_slots]: earliest start of an item of the block (wall clock ticks)
    unsigned long long *next_item;         // device [grid]: the item a workgroup takes next
    unsigned *exiting;                     // device: a workgroup has left
    const uint8_t *in_ring; int16_t *out_ring; size_t in_slot_bytes, out_slot_elems;
    int n_slots, desc_lines, n_wsb, T, D, L, F;
    long long idle_ticks, life_ticks;
    const float *lead_d; int lead_stride;  // [n_streams][lead_stride]: a retuned stream's first audio samples of the block (k_wfm_lead)
    const float *lead_state;               // [n_streams]: the de-emphasis state in front of the first block behind a retune
    unsigned long long *stats;             // device [grid][4]: ticks spent waiting for a block, in the body, in the completion; items (accumulated over launches)
    int fence_mode;                        // experiments: 0 = write-through stores, no fence (default), 3 = plus one system fence per item, 2 = plus one by every thread
};
__device__ __forceinline__ uint32_t sys_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ uint32_t res_tag(long long k) { return (uint32_t)((unsigned long long)k % 0xfffffffeull) + 1u; }

template <bool PS, bool RES = false>
__global__ __launch_bounds__(64 * SEQ_NW) void k_wfm_mfma_seq(const uint8_t *__restrict__ in, size_t in_pitch, const v4i *__restrict__ frags, const float *__restrict__ cum,
                                                               const float2 *__restrict__ dtab, const float2 *__restrict__ ctab, SeqParams p_in, ResCtl rc)
{
    static_assert(!(PS && RES), "the resident form exists for the shared-rate kernel");
    constexpr int TPG = SEQ_TPG, SPW = 16 / SEQ_NLW, NTHR = 64 * SEQ_NW;              // tiles per step; streams fetched per fetching wave in a row-step
    extern __shared__ float4 lds_raw[];
    uint8_t *lds_in = reinterpret_cast<uint8_t *>(lds_raw);
    float *lds_out = reinterpret_cast<float *>(lds_in + 16 * SEQ_RP);                // 16 x SEQ_OUTP floats
    float *lcum = lds_out + 16 * SEQ_OUTP;                                       // prefix table (a global vector load inside the loop would drain the DMA ring)
    float2 *ctl = reinterpret_cast<float2 *>(lcum + (SEQ_NGR + 1) * 16);          // PS: [2][3][16] chunk seeds of a step (class, column);  RES: [2 words of control][chunk seeds of the block]
    const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, col = lane & 15, q = lane >> 4;
    const bool fetches = wv >= TPG;
#ifdef WFM_PROF
    const long long t_entry = __builtin_readcyclecounter();
    long long t_first = t_entry, t_last = t_entry;
#endif
    const int fw = wv - TPG;                                                         // index among the fetching waves
    float scale = p_in.scale;
    if constexpr (PS) { const int sbp = blockIdx.x; frags += (size_t)sbp * p_in.frag_stride; cum += (size_t)sbp * ((SEQ_NGR + 1) * 16); dtab += (size_t)sbp * 3072; ctab += sbp; scale = p_in.scales[sbp]; }
    for (int i = tid; i < (SEQ_NGR + 1) * 16; i += NTHR) lcum[i] = cum[i];
    v4i A[WFM_NK * 3];
#pragma unroll
    for (int s = 0; s < WFM_NK * 3; s++) A[s] = frags[s * 64 + lane];

    float c_lo[4], c_hi[4];                                                          // prefix values at the window's ends, rows 4q .. 4q+3
    {
        const float4 a = *reinterpret_cast<const float4 *>(cum + 4 * q), b = *reinterpret_cast<const float4 *>(cum + (size_t)SEQ_NGR * 16 + 4 * q);
        c_lo[0] = a.x; c_lo[1] = a.y; c_lo[2] = a.z; c_lo[3] = a.w; c_hi[0] = b.x; c_hi[1] = b.y; c_hi[2] = b.z; c_hi[3] = b.w;
    }
    const float K = 0.340447550238101026565118445432744920253753662109375f;
    // ---- RES: the walk over the work items
    SeqParams p_res;
    if constexpr (RES) p_res = p_in;
    const SeqParams &p = RES ? p_res : p_in;
    unsigned long long item = 0;
    long long res_t_launch = 0;
    uint32_t *const rctl = reinterpret_cast<uint32_t *>(ctl);                        // RES: [0] = go / leave, [1] = lead samples
    const float2 *const lseed = ctl + 1;                                             // RES: seeds of chunks first_chunk - 4 ...
    long long st_wait = 0, st_body = 0, st_done = 0, st_items = 0, st_t = 0;        // RES: thread 0's clock readings
    if constexpr (RES) { item = rc.next_item[blockIdx.x]; res_t_launch = wall_clock64(); st_t = res_t_launch; }
    // RES, thread 0: the previous item's completion in flight (its count returns while the next block's descriptor is polled); wave 0: the next descriptor's lines
    bool pend = false; unsigned pend_c = 0; int pend_slot = 0, pend_n = 0; long long pend_k = 0;
    uint32_t pre_v[4] = {0u, 0u, 0u, 0u}; bool pre_valid = false;
    auto res_complete = [&]() {
        if constexpr (RES) {
            if (tid == 0 && pend) {
                pend = false;
                if (pend_c + 1 == (unsigned)rc.n_wsb) {                              // this workgroup completed the block: every item's audio is in memory (write-through stores, waited for)
                    const unsigned long long tf = __hip_atomic_exchange(rc.t_first + pend_slot, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), td = (unsigned long long)wall_clock64();
                    __hip_atomic_store(rc.cnt + pend_slot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    uint32_t *dn = rc.done + (size_t)pend_slot * 16;
                    __hip_atomic_store(dn + 1, (uint32_t)pend_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(dn + 2, (uint32_t)tf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __hip_atomic_store(dn + 3, (uint32_t)(tf >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(dn + 4, (uint32_t)td, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __hip_atomic_store(dn + 5, (uint32_t)(td >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the line's fields have left before its tag does
                    __hip_atomic_store(dn, res_tag(pend_k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
    };
    for (;;) {                                                                       // (not RES: one pass)
    int res_slot = 0, res_lead = 0; long long res_k = 0; bool res_retuned = false;
    const uint8_t *pblock = nullptr;                                                 // RES: the previous block's rows of this stream group (what lies in front of the block)
    if constexpr (RES) {
        res_k = (long long)(item / (unsigned)rc.n_wsb);
        res_slot = (int)(res_k % rc.n_slots);
        if (wv == 0) {
            const uint32_t *dsc = rc.desc + (size_t)res_slot * rc.desc_lines * 16;
            const uint32_t tag = res_tag(res_k);
            const int nd = rc.desc_lines * 16;
            const long long t_wait = wall_clock64();
            uint32_t state = 0;
            // Polling costs PCIe reads: 256 idle workgroups that each re-read a whole descriptor (256 B) and the stop word every microsecond are ~40 GB/s -- the link's own
            // rate; on a box whose host memory sits far from the GPU that traffic tripled every workgroup's waiting time and slowed the busy ones' stores and prefetches
            // (round 6: 0.52 of the roofline on one box, 0.22 on another).  So: the lines asked for at the end of the previous item are looked at first (the usual case: the
            // block is already posted); after that a poll reads ONE word -- line 0's tag --, the whole descriptor only once that word matches, the stop word every eighth
            // poll, and the pause between polls grows from ~0.6 to ~5 us.
            for (int iter = 0;; iter++) {
                uint32_t v[4] = {0u, 0u, 0u, 0u};
                bool have = false;
                if (iter == 0 && pre_valid) {                                        // the lines this wave asked for while the previous item's audio was on its way out
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = pre_v[j];
                    have = true;
                } else {
                    const uint32_t t0 = sys_load(dsc);                               // (uniform address: one request)
                    if (t0 == tag) {
#pragma unroll
                        for (int j = 0; j < 4; j++) { const int d = lane + 64 * j; v[j] = d < nd ? sys_load(dsc + d) : 0u; }
                        have = true;
                    }
                }
                const uint32_t ex = __hip_atomic_load(rc.exiting, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t stop = (iter & 7) == 7 ? sys_load(rc.ctrl) : 0u;
                res_complete();                                                      // the previous item's count has come back by now: the block's done line, if it was the last
                bool ok = have;
#pragma unroll
                for (int j = 0; j < 4; j++) { const int d = lane + 64 * j; if (d < nd && (d & 15) == 0 && v[j] != tag) ok = false; }
                const bool ready = __all(ok);
                const long long now = wall_clock64();
                const bool old = now - res_t_launch > rc.life_ticks;
                if (ready && !ex && !old) {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int d = lane + 64 * j, w = d & 15;
                        if (d < nd && w >= 2) rctl[2 + (d >> 4) * 14 + (w - 2)] = v[j];                 // seed floats (cos, sin) x 7 per line -> lseed[]
                        if (d == 1) rctl[1] = v[j];
                    }
                    state = 1; break;
                }
                if (ex || old || stop || now - t_wait > rc.idle_ticks) { state = 2; break; }
                if (iter < 2) __builtin_amdgcn_s_sleep(24); else if (iter < 6) __builtin_amdgcn_s_sleep(64); else { __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(100); }
            }
            pre_valid = false;
            if (lane == 0) rctl[0] = state;
        }
        __syncthreads();
        if (rctl[0] != 1u) {
            if (tid == 0) {
                __hip_atomic_store(rc.exiting, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); rc.next_item[blockIdx.x] = item;
                unsigned long long *sp = rc.stats + (size_t)blockIdx.x * 4;
                sp[0] += (unsigned long long)st_wait; sp[1] += (unsigned long long)st_body; sp[2] += (unsigned long long)st_done; sp[3] += (unsigned long long)st_items;
            }
            return;
        }
        if (tid == 0) { const long long t = wall_clock64(); st_wait += t - st_t; st_t = t; }
        res_lead = (int)(rctl[1] & 0xffu); res_retuned = (rctl[1] >> 8) & 1u;
        // the block's geometry (what csdr_amd_wfm_process derives on the host, wfm.hip): audio samples that become computable with block k
        auto j_hi = [&](long long kk) -> long long {
            if (kk < 0) return -1;
            const long long avail_last = (kk + 1) * rc.T - 1;
            if (avail_last - (rc.L - 1) < 0) return -1;
            const long long k_max = (avail_last - (rc.L - 1)) / rc.D;
            return k_max >= 10 ? (k_max - 10) / rc.F : -1;
        };
        const long long jp = j_hi(res_k - 1), jn = j_hi(res_k);
        p_res.B2 = 2 * res_k * rc.T;
        p_res.j_first = jp + 1; p_res.n_audio = (int)(jn - jp);
        p_res.tile_first = p_res.j_first / 4; p_res.n_tiles = (int)((p_res.j_first + p_res.n_audio - 1) / 4 - p_res.tile_first + 1);
        p_res.tiles_per_seg = p_res.n_tiles;
        p_res.s16 = rc.out_ring + (size_t)res_slot * rc.out_slot_elems;
        if (tid == 0) (void)__hip_atomic_fetch_min(rc.t_first + res_slot, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int sb = RES ? (int)(item % (unsigned)rc.n_wsb) : (int)blockIdx.x;         // block of 16 streams; PS: the stream
    const int seg_y = RES ? 0 : (int)blockIdx.y;                                      // time segment of the call (RES: a block is one segment)
    const int col0 = PS ? 16 * seg_y : 0;                                            // PS: absolute index of the workgroup's first column
    if constexpr (RES) {
        in = rc.in_ring + (size_t)res_slot * rc.in_slot_bytes;
        pblock = rc.in_ring + (size_t)((res_slot + rc.n_slots - 1) % rc.n_slots) * rc.in_slot_bytes + (long long)sb * 16 * (long long)in_pitch;
    }
    const long long t0 = p.tile_first + (long long)seg_y * p.tiles_per_seg * (PS ? 16 : 1);      // PS: column 0's first tile
    long long t1 = t0 + p.tiles_per_seg; if (t1 > p.tile_first + p.n_tiles) t1 = p.tile_first + p.n_tiles;
    if (!RES && t0 >= t1) return;
    const int n_it = (int)(t1 - t0), n_grp = (n_it + TPG - 1) / TPG;
    const int n_lead = RES ? res_lead : ((PS && p.lead_n && seg_y == 0) ? p.lead_n[sb] : 0);      // PS / RES: audio samples at the call's start that k_wfm_lead has evaluated (a retuned stream)
    const int n_warm = RES ? ((res_k > 0 && !res_retuned) ? 2 : 0) : ((PS || seg_y > 0) ? 2 : 0);       // steps demodulated ahead of the segment to warm the de-emphasis up (RES: the first
                                                                                     // block behind a retune gets its state from k_wfm_lead_state: the samples in front of it want the old weights)
    const int last_stream = p.n_streams - 1;
    // ---- DMA ring.  Positions are bytes from the block start; they start at -1024 (the head), hence the + 2 SEQ_RB in the slot arithmetic
    const int tstride = p.stride;
    const long long org = PS ? (long long)col0 * p.col_bytes : 0LL;                  // PS: positions are counted from the start of the workgroup's column 0
    long long wg = (t0 - (long long)n_warm * TPG) * tstride + p.win_off - p.B2 - org; // window start of the step's first tile
    const long long F0 = wg & ~1023LL;                                               // (floor, also for negative positions)
    long long F_end = ((t1 - 1) * tstride + p.win_off - p.B2 - org + 64 * WFM_NK + 1023) & ~1023LL;
    { const long long row_end = ((p.two_T + 1023) & ~1023LL) - org; if (F_end > row_end) F_end = row_end; }      // a partial last tile's window reaches beyond the block: those bytes feed no stored sample
    long long F = F0;
    int fslot = (int)((F0 + 2 * SEQ_RB) % SEQ_RB), wslot = (int)((wg + 2 * SEQ_RB) % SEQ_RB);      // ring positions of F and of wg (wg already includes the warm-up steps)
    const uint32_t lds_in_addr = (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t *)lds_in;
    uint32_t voff[SPW], voff_h[SPW];
    long long row_lim[SPW];                                                          // PS: bytes of the block from the row's column start on (wave uniform)
#pragma unroll
    for (int r = 0; r < SPW; r++) {
        if constexpr (PS) {
            voff[r] = (uint32_t)((long long)(SPW * fw + r) * p.col_bytes) + 16u * lane;      // relative to the workgroup's column 0
            voff_h[r] = 16u * lane;
            row_lim[r] = p.two_T - (long long)(col0 + SPW * fw + r) * p.col_bytes;
        } else {
            const int srow = max(min(sb * 16 + SPW * fw + r, last_stream) - sb * 16, 0);
            voff[r] = (uint32_t)srow * (uint32_t)in_pitch + 16u * lane;
            voff_h[r] = (uint32_t)srow * (uint32_t)SEQ_HEAD + 16u * lane;
            row_lim[r] = 0;
        }
    }
    const uint8_t *sblock = PS ? in + (long long)sb * (long long)in_pitch + org : in + (long long)sb * 16 * (long long)in_pitch;
    const uint8_t *hblock = PS ? p.head_in + (size_t)sb * SEQ_HEAD : p.head_in + (size_t)sb * 16 * SEQ_HEAD;
    const bool ragged = (p.two_T & 1023) != 0;                                       // only a stream's last block
    auto row_step = [&]() {
        if constexpr (PS) {
            // positions are column 0's, counted from its start; the other rows read the same position of their own columns (inside the block also for F < 0).  In
            // front of the call's very first column lies the head (1 KiB; further back -- the warm-up steps of that column, whose audio is replaced by the carried
            // state -- any readable bytes do); rows behind the block's end re-read column 0; a ragged end is masked at the row's last 16-byte piece.
            const uint32_t ldst = lds_in_addr + (SPW * fw) * SEQ_RP + (uint32_t)fslot;
            const bool head = F < 0 && col0 == 0;                                    // wave uniform
            if (!head && !ragged) {                                                  // the common case: whole runs (row_lim is a multiple of 1024 like F), inside the block or behind it
                uint32_t vo[SPW];
#pragma unroll
                for (int r = 0; r < SPW; r++) vo[r] = F < row_lim[r] ? voff[r] : 16u * lane;      // behind the block's end: column 0's bytes again (the partial last column of
                dma_rows<SPW, SEQ_RP>(vo, sblock + F, __builtin_amdgcn_readfirstlane((int)ldst)); // a call spends a good part of its steps there: no row by row path for it)
            } else
#pragma unroll
            for (int r = 0; r < SPW; r++) {
                const bool hrow = head && (SPW * fw + r == 0 || F + 1024 > row_lim[r]);
                const bool beyond = !hrow && F >= row_lim[r];                        // nothing of this run lies inside the block: column 0's bytes again
                const uint8_t *sbase = hrow ? hblock : sblock + F;
                const uint32_t vo = hrow ? voff_h[r] : (beyond ? 16u * lane : voff[r]);
                const bool live = hrow || F + 16 * lane < (beyond ? p.two_T - org : row_lim[r]);      // (a run cut by the block's end: whole 16-byte pieces inside only; F < F_end: lane 0 is always live)
                const uint32_t la = __builtin_amdgcn_readfirstlane((int)(ldst + r * SEQ_RP));
                uint32_t keep;
                if (live)
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(vo), "s"(sbase), "s"(la) : "memory");
            }
            F += 1024; fslot += 1024; if (fslot >= SEQ_RB) fslot -= SEQ_RB;
            return;
        }
        if constexpr (RES) {                                                         // in front of the block lies the previous block's slot of the input ring; blocks are whole chunks
            const uint8_t *sb_res = F < 0 ? pblock + (p.two_T + F) : sblock + F;
            dma_rows<SPW, SEQ_RP>(voff, sb_res, __builtin_amdgcn_readfirstlane((int)(lds_in_addr + (SPW * fw) * SEQ_RP + (uint32_t)fslot)));
            F += 1024; fslot += 1024; if (fslot >= SEQ_RB) fslot -= SEQ_RB;
            return;
        }
        const bool head = F < 0;                                                     // wave uniform: the run [-1024, 0)
        const uint8_t *sbase = head ? hblock : sblock + F;
        const uint32_t ldst = lds_in_addr + (SPW * fw) * SEQ_RP + (uint32_t)fslot;
        const bool mask = ragged && !head;                                           // wave uniform; ragged end: the row's last 16-byte piece is the last one fetched
        const bool live = !mask || F + 16 * lane < p.two_T;
        if (!head && !ragged) dma_rows<SPW, SEQ_RP>(voff, sbase, __builtin_amdgcn_readfirstlane((int)ldst));      // the common case
        else
#pragma unroll
        for (int r = 0; r < SPW; r++) {
            const uint32_t la = __builtin_amdgcn_readfirstlane((int)(ldst + r * SEQ_RP));
            const uint32_t vo = head ? voff_h[r] : voff[r];
            uint32_t keep;
            if (!mask)
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(vo), "s"(sbase), "s"(la) : "memory");
            else if (live)
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(vo), "s"(sbase), "s"(la) : "memory");
        }
        F += 1024; fslot += 1024; if (fslot >= SEQ_RB) fslot -= SEQ_RB;
    };
    auto wait_newer = [&](long long newer) {                                         // SPW loads per wave per row-step, returned in order
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
    auto wait_for = [&](long long last_window_start) {
        const long long need_end = (last_window_start + 64 * WFM_NK + 1023) & ~1023LL;
        long long newer = (F - need_end) >> 10;
        if (newer < 0) newer = 0;
        if (newer > 7) newer = 7;
        wait_newer(newer);
    };
    // PS: chunk seeds of a step (window start of its first tile: wgs): class k = 0 .. 2 <-> chunk (that of the step's first window) + k, of every column; lanes 0 .. 47
    // of compute wave 1.  Loaded one step ahead of their write into the table half the step two ahead will read (see the kernel's comment; ddc_mfma.hip's ps_load).
    typedef float ps_v2f __attribute__((ext_vector_type(2)));
    ps_v2f psC = {1.f, 0.f};
    const bool ps_wave = PS && wv == 1, ps_lane = ps_wave && lane < 48;
    auto ps_base = [&](long long wgs) { return (int)((((wgs + p.B2) >> 1) >> 10) - (p.B2 >> 11)); };      // chunk of a step's first window, relative to the block's first chunk + column 0's
    auto ps_load = [&](long long wgs) {
        if (ps_lane) {
            int ci = ps_base(wgs) + 1 + (lane >> 4) + (col0 + (lane & 15)) * p.col_chunks;
            ci = min(max(ci, 0), p.tab_len - 1);                                     // (in front of the history chunk; columns behind the block's end, steps behind the last)
            const float2 *pc = ctab + (size_t)ci * p.tab_pitch;
            asm volatile("global_load_dwordx2 %0, %1, off" : "+v"(psC) : "v"(pc) : "memory");
        }
    };
    auto ps_store = [&](int g) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(psC) :: "memory");
        if (ps_lane) ctl[((g + 2) & 1) * 48 + lane] = make_float2(psC.x, psC.y);     // (g >= -2: the parity of g + 2)
    };
    if (ps_wave) {
        const long long step_b = (long long)TPG * tstride;
        // the first two steps' seeds in flight together (one after the other they cost the workgroup 3.7 us in front of its first step: tools/diag_wfm_life.py)
        ps_v2f c0 = {1.f, 0.f}, c1 = {1.f, 0.f};
        if (ps_lane) {
            auto entry = [&](long long wgs) {
                int ci = ps_base(wgs) + 1 + (lane >> 4) + (col0 + (lane & 15)) * p.col_chunks;
                ci = min(max(ci, 0), p.tab_len - 1);
                return ctab + (size_t)ci * p.tab_pitch;
            };
            const float2 *pa = entry(wg), *pb = entry(wg + step_b);
            asm volatile("global_load_dwordx2 %0, %2, off\n\tglobal_load_dwordx2 %1, %3, off" : "+v"(c0), "+v"(c1) : "v"(pa), "v"(pb) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(c0), "+v"(c1) :: "memory");
        if (ps_lane) {                                                               // (n_warm = 2: steps -2 and -1 -> halves 0 and 1)
            ctl[((-n_warm + 2) & 1) * 48 + lane] = make_float2(c0.x, c0.y);
            ctl[((-n_warm + 1 + 2) & 1) * 48 + lane] = make_float2(c1.x, c1.y);
        }
        ps_load(wg + 2 * step_b);                                                    // in flight: the third step
    }
    if (fetches) {
        while (F < F_end && F + 1024 <= wg + SEQ_RB) row_step();
        wait_for(wg + (long long)(TPG - 1) * tstride);
    }
    __syncthreads();
    const uint8_t *lrow = lds_in + col * SEQ_RP;
    const int s0 = sb * 16;
    // ---- de-emphasis state of stream s0 + col (wave 0).  The recurrence y_k = alpha x_k + b y_{k-1} (b = 1 - alpha) over a step's 32 samples of a
    // stream is split over 4 lanes (lane = stream + 16 * quarter): each lane runs its 8 samples from zero state, the quarters' end values are combined
    // with b^8, b^16, b^24 (three shuffles), and every output gets its share b^(j+1) of its quarter's start state -- a dependent chain of ~12 instead of
    // 32 steps on the one wave that all others wait for at the next barrier.  (Rounding differs from the sequential recurrence by a few 1e-8.)
    float yst = 0.f;                                                                 // state after the last sample processed so far (same value in the stream's 4 lanes)
    const float one_minus = 1 - p.alpha;
    const bool iir_wave = wv == 0, iir_lane = iir_wave && lane < 16;
    constexpr int QL = TPG, SPS = 4 * TPG;                                           // samples per lane of the scan (a quarter of a step), samples per step and stream
    float bp[QL + 1];                                                                // b^0 .. b^QL
    bp[0] = 1.f;
#pragma unroll
    for (int j = 1; j <= QL; j++) bp[j] = bp[j - 1] * one_minus;
    const float b2q = bp[QL] * bp[QL], b3q = b2q * bp[QL];
    float yst_first = 0.f;                                                           // PS: the call's first column starts from the carried state at step 0 (its warm-up steps ran on whatever lay in front)
    if (!RES && iir_wave && seg_y == 0) {                                            // first segment: the exact carried state (NaN reset as libcsdr.c:1092)
        yst = p.last_in[PS ? sb : min(s0 + col, last_stream)]; if (yst != yst) yst = 0.f;
        yst_first = yst;
    }
    if (RES && iir_wave && res_retuned) yst = rc.lead_state[min(s0 + col, last_stream)];
    // the segment's audio samples, counted from its first tile's first sample (ints: this bookkeeping runs every step, on wave 0 between two barriers)
    const long long j_end = p.j_first + p.n_audio;
    const int seg_lo = (int)max(0LL, p.j_first - 4 * t0);                           // > 0 only in the call's first segment, when j_first is not a multiple of 4
    const int seg_hi = (int)min(4LL * n_it, j_end - 4 * t0);                        // samples of the segment that exist in this call
    const long long idx0 = 4 * t0 - p.j_first;                                      // output index of the segment's sample 0
    // PS: the same per column (segment of column c = tiles t0 + c tiles_per_seg ...): only the call's first column can start late, only its last one ends early
    auto col_lo = [&](int c) { return PS ? (col0 + c == 0 ? seg_lo : 0) : seg_lo; };
    auto col_hi = [&](int c) { return PS ? (int)max(0LL, min(4LL * n_it, j_end - 4 * (t0 + (long long)c * p.tiles_per_seg))) : seg_hi; };
    // ---- Stores (convert_f_s16, libcsdr.c:2397, x86 truncation semantics).  Usual case (16-byte aligned rows; emit_vec): the LOADER waves take the audio out of the
    // staging ring, one step after the de-emphasis, in whole 128-byte LINES of the s16 output row (64 samples, 8 lanes x 16 bytes; lines of the OUTPUT row, whatever the
    // call's first sample is: after step g the line whose last sample lies in step g is complete -- the same step for all 16 streams; it may start up to 63 samples
    // back, hence the ring of SEQ_OUTS steps) and keep them in registers -- a loader uses none otherwise -- until 44 lines = 5.5 KiB per stream are together.  Why: beside a
    // saturated read stream the memory charges 2 % of the bytes written as stores with 17-20 % of the read rate, whatever the cache policy or the instruction count,
    // nothing if the lines stay in L2, half of it for 4 KiB per row at a time, a quarter for 8 KiB (tools/microbench/dma_write_mix.hip, profiles/r3_notes.md).
    // Lines cut by the call's first / last sample or the segment's ends, and every line when the float audio is wanted too, are stored at once.
    const bool emit_vec = ((((size_t)p.s16 | (size_t)p.af) & 15) == 0) && (p.out_pitch & 7) == 0;
    const int emit_a = (int)((SEQ_LINE - (idx0 & (SEQ_LINE - 1))) & (SEQ_LINE - 1));  // segment samples n = emit_a (mod 64) start a line
    // RES: the audio is stored WRITE-THROUGH (sc0 sc1: system scope), so that "every store acknowledged" means "in memory" and an item can be counted in without writing
    // the XCD's L2 back (one such write-back per item cost 4-8 us of every workgroup's 50: profiles/r6_notes.md)
    typedef unsigned wfm_st4 __attribute__((ext_vector_type(4)));
    auto put16 = [&](int16_t *dst, unsigned a, unsigned b, unsigned c, unsigned d) __attribute__((always_inline)) {
        if constexpr (RES) { const wfm_st4 v = {a, b, c, d}; asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(dst), "v"(v) : "memory"); }
        else *reinterpret_cast<uint4 *>(dst) = make_uint4(a, b, c, d);
    };
    auto put2 = [&](int16_t *dst, int v) __attribute__((always_inline)) {
        if constexpr (RES) asm volatile("global_store_short %0, %1, off sc0 sc1" :: "v"(dst), "v"(v) : "memory");
        else *dst = (int16_t)v;
    };
    auto s16_of = [](float e) { const float scaled = e * 32767.0f; return (scaled >= -2147483648.0f && scaled < 2147483648.0f) ? (int)scaled : (int)0x80000000; };
    // rows or pitches that are not 16-byte aligned: sample by sample, by the compute waves except wave 0 (which runs the de-emphasis)
    constexpr int EMIT_T0 = 64, EMIT_N = 64 * (TPG - 1);
    auto emit_scalar = [&](int g) {
        if (g < 0 || g >= n_grp || tid < EMIT_T0) return;
#pragma unroll
        for (int e0 = 0; e0 < 16 * SPS; e0 += EMIT_N) {
            const int ei = e0 + tid - EMIT_T0;
            const int srow = ei / SPS, k = ei % SPS, kr = g * SPS + k;
            if (ei >= 16 * SPS) continue;
            if (PS ? (kr < col_lo(srow) || kr >= col_hi(srow)) : (kr < seg_lo || kr >= seg_hi || s0 + srow >= p.n_streams)) continue;
            const float e = lds_out[srow * SEQ_OUTP + (g % SEQ_OUTS) * SPS + k];
            const long long idx = idx0 + kr;
            const size_t at = PS ? (size_t)sb * p.out_pitch + (size_t)srow * p.col_audio + idx : (size_t)(s0 + srow) * p.out_pitch + idx;      // (idx0 is column 0's)
            p.s16[at] = (int16_t)s16_of(e);
            if (p.af) p.af[at] = e;
        }
    };
#ifdef WFM_PROF
    long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = __builtin_readcyclecounter();
#endif
    if (fetches) {
        // ======================================================================== the loader waves: fetch, and take the finished audio out
        typedef unsigned wfm_u4 __attribute__((ext_vector_type(4)));
        constexpr int NST = RES ? 8 : 44;                                            // lines held per stream: 5.5 KiB, 176 registers (32 / 40 / 44 lines: 0.859 / 0.848 / 0.849 ms; 48: 256 registers and spills)
                                                                                     // (RES: short blocks -- 16384 samples are five lines --, and the registers are wanted for the walk)
        // (named registers, written through selects: any array or switch form went to scratch memory)
#define WFM_ST_ALL(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) \
                      X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32) X(33) X(34) X(35) X(36) X(37) X(38) X(39) X(40) X(41) X(42) X(43)
#define X(k) wfm_u4 st##k = {0u, 0u, 0u, 0u};
        WFM_ST_ALL(X)
#undef X
        int n_st = 0, st_n0 = 0;                                                     // lines held, first sample of the first one (wave uniform)
        const int srow = 8 * fw + (lane >> 3), pc = lane & 7;                        // this lane: stream row of the workgroup, 16-byte piece of a line
        // PS: this lane's row is a COLUMN of the stream: its own valid range [lo_r, hi_r) (the uniform decisions below use column 0's, the widest) and output offset
        const int lo_r = col_lo(srow), hi_r = col_hi(srow);
        const bool row_ok = PS ? hi_r > lo_r : s0 + srow < p.n_streams;
        const size_t row_at = PS ? (size_t)sb * p.out_pitch + (size_t)srow * p.col_audio : (size_t)(s0 + srow) * p.out_pitch;      // (idx0 is column 0's)
        int16_t *const orow = p.s16 + row_at;
        auto flush = [&]() __attribute__((always_inline)) {
            if constexpr (PS) {
                // a held line is whole for column 0; for the call's last column it may be cut, for columns behind the block's end it does not exist
#define X(j) if (j < NST && j < n_st && row_ok) { const int a = st_n0 + SEQ_LINE * j + 8 * pc;                                                             \
                 if (a + 8 <= hi_r) *reinterpret_cast<uint4 *>(orow + (idx0 + a)) = make_uint4(st##j[0], st##j[1], st##j[2], st##j[3]);        \
                 else for (int i = 0; i < 8; i++) if (a + i < hi_r) orow[idx0 + a + i] = (int16_t)(st##j[i >> 1] >> (16 * (i & 1))); }
                WFM_ST_ALL(X)
#undef X
            } else {
#define X(j) if (j < NST && j < n_st && row_ok) put16(orow + (idx0 + st_n0 + SEQ_LINE * j + 8 * pc), st##j[0], st##j[1], st##j[2], st##j[3]);
                WFM_ST_ALL(X)
#undef X
            }
            n_st = 0;
        };
        auto take_line = [&](int g) __attribute__((always_inline)) {                                              // g = n_grp: the segment's last, incomplete line
            if (g < 0 || !emit_vec) return;
            // lines k = 0, 1, ...: samples [emit_a + 64 (k - 1), emit_a + 64 k); complete once sample emit_a + 64 k - 1 exists (floor divisions: arithmetic shifts)
            const int k_before = (g * SPS - emit_a) >> 6, k_now = g < n_grp ? ((g + 1) * SPS - emit_a) >> 6 : k_before + 1;
            if (k_now <= k_before) return;                                           // (at most one line per step: SPS < 64)
            static_assert(4 * SEQ_TPG < SEQ_LINE && SEQ_LINE == 64 && SEQ_NLD == 2, "one line per step at most; 8 streams x 8 pieces per loader");
            const int nl = emit_a + SEQ_LINE * (k_now - 1), n0 = nl + 8 * pc;        // the line's first sample, this lane's
            if (nl + SEQ_LINE <= (PS ? 0 : seg_lo) || nl >= seg_hi) return;          // (PS: seg_lo is the call's first column's; every other column starts at its sample 0 -- a line in front of
                                                                                     // column 0's first sample still holds the first samples of the columns behind it, cut per lane below)
            const float *src = lds_out + srow * SEQ_OUTP;
            float e[8];
#pragma unroll
            for (int i = 0; i < 8; i++) e[i] = src[(n0 + i + SEQ_OUTR) % SEQ_OUTR];  // (n0 >= -64)
            int v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = s16_of(e[i]);
            const wfm_u4 w = {(unsigned)(v[0] & 0xffff) | ((unsigned)v[1] << 16), (unsigned)(v[2] & 0xffff) | ((unsigned)v[3] << 16),
                              (unsigned)(v[4] & 0xffff) | ((unsigned)v[5] << 16), (unsigned)(v[6] & 0xffff) | ((unsigned)v[7] << 16)};
            if (nl >= seg_lo && nl + SEQ_LINE <= seg_hi && !p.af) {                  // a whole line (wave uniform): held
                if (n_st == 0) st_n0 = nl;
#define X(k) if (k < NST) st##k = n_st == k ? w : st##k;           // (selects: a switch became a store through a pointer and sent the registers to scratch memory)
                WFM_ST_ALL(X)
#undef X
                if (++n_st == NST) flush();
            } else if (row_ok) {
                const size_t o = (size_t)(idx0 + n0);                                // (may wrap below zero in front of a cut line's first valid sample: never dereferenced there)
                float *const arow = p.af ? p.af + row_at : nullptr;
                if (n0 >= lo_r && n0 + 8 <= hi_r) {
                    put16(orow + o, w[0], w[1], w[2], w[3]);
                    if (arow) { *reinterpret_cast<float4 *>(arow + o) = make_float4(e[0], e[1], e[2], e[3]); *reinterpret_cast<float4 *>(arow + o + 4) = make_float4(e[4], e[5], e[6], e[7]); }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; i++)
                        if (n0 + i >= lo_r && n0 + i < hi_r) { put2(orow + (o + i), v[i]); if (arow) arow[o + i] = e[i]; }
                }
            }
        };
        PROF_T(7)
#ifdef WFM_PROF
        t_first = __builtin_readcyclecounter();
#endif
        // (Round 5, measured and dropped: a dozen steps before its end a workgroup touched the head and first 2 KiB per stream of workgroup id + 256 -- the one that runs
        // next on this XCD -- so that its first windows would wait in L2: 6.8 us pass between kernel entry and the first step (tools/diag_wfm_life.py), a sixth of a
        // workgroup's life at 65536 x 24576.  65536 x 24576: 0.674 -> 0.697 ms (0.685 with the touches switched off: the code alone cost 1.6 %).)
        for (int gi = -n_warm; gi < n_grp; gi++) {
            PROF_T(0)
            const long long wg_n = wg + (long long)TPG * tstride;
            if (gi + 1 < n_grp) wait_for(wg_n + (long long)(TPG - 1) * tstride);
            PROF_T(1)
            __syncthreads();
            PROF_T(2)
            if (gi + 1 < n_grp) { while (F < F_end && F + 1024 <= wg_n + SEQ_RB) row_step(); }
            PROF_T(3)
            if (WFM_DIAG < 4) take_line(gi - 1);                                    // the previous step's audio: filtered by wave 0 before it came to this barrier
            PROF_T(4)
#ifdef WFM_PROF
            prof[6] += 1;
#endif
            wg = wg_n;
        }
#ifdef WFM_PROF
        if (lane == 0) for (int k = 0; k < 8; k++) atomicAdd(&g_wfm_prof[wv][k], (unsigned long long)prof[k]);
        t_last = __builtin_readcyclecounter();
#endif
        __syncthreads();
        if (WFM_DIAG < 4) { take_line(n_grp - 1); take_line(n_grp); flush(); }
#undef WFM_ST_ALL
    } else {
    // ============================================================================ the compute waves
    PROF_T(7)
#ifdef WFM_PROF
    t_first = __builtin_readcyclecounter();
#endif
    for (int gi = -n_warm; gi < n_grp; gi++) {
        const int it = gi * TPG + wv;
        float *lout = lds_out + ((gi + SEQ_OUTS) % SEQ_OUTS) * SPS;                   // this step's samples in every stream's ring
        if (it < n_it && (WFM_DIAG < 3)) {
            const long long ws = wg + (long long)wv * tstride;
            const long long n0 = (ws + p.B2) >> 1;                                   // global index of the window's first sample
            const int off = (int)(n0 & 1023);
            const bool two = off + 32 * WFM_NK > 1024;
            const int bo = 2 * (1024 - off);                                         // byte offset of the next chunk's first sample inside the window
            const int kb = __builtin_amdgcn_readfirstlane(two ? (bo >> 6) : WFM_NK), hq = __builtin_amdgcn_readfirstlane((bo >> 4) & 3);
            const long long chunk_rel = (n0 >> 10) - (p.B2 >> 11);                   // -1: the window starts in the previous block's last chunk
            // ---- B fragments from the ring
            unsigned a0 = (unsigned)(wslot + wv * tstride + 16 * q);
            a0 = min(a0, a0 - (unsigned)SEQ_RB);                                     // wrap (at most once: wslot < RB, the rest < RB)
            v4i Bf[WFM_NK];
#pragma unroll
            for (int ks = 0; ks < WFM_NK; ks++) {
                unsigned a = a0 + 64u * ks; a = min(a, a - (unsigned)SEQ_RB);
                if (WFM_DIAG == 1) Bf[ks] = v4i{(int)a, lane, ks, gi};
                else Bf[ks] = *reinterpret_cast<const v4i *>(lrow + a) ^ (int)0x80808080;
            }
            v4i acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}, snap[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
            const bool lo_lane = q < hq;
            if (WFM_DIAG == 2) {
#pragma unroll
                for (int ks = 0; ks < WFM_NK; ks++) { acc[ks % 3] += Bf[ks]; snap[ks % 3] ^= Bf[ks]; }
            } else
            switch (kb) {
                case 0: wfm_chain<0>(A, Bf, lo_lane, acc, snap); break;
                case 1: wfm_chain<1>(A, Bf, lo_lane, acc, snap); break;
                case 2: wfm_chain<2>(A, Bf, lo_lane, acc, snap); break;
                case 3: wfm_chain<3>(A, Bf, lo_lane, acc, snap); break;
                case 4: wfm_chain<4>(A, Bf, lo_lane, acc, snap); break;
                case 5: wfm_chain<5>(A, Bf, lo_lane, acc, snap); break;
                case 6: wfm_chain<6>(A, Bf, lo_lane, acc, snap); break;
                case 7: wfm_chain<7>(A, Bf, lo_lane, acc, snap); break;
                default: wfm_chain<WFM_NK>(A, Bf, lo_lane, acc, snap); break;
            }
            // ---- post factors and offset constants of the window's part(s)
            const int kcls = PS ? (int)chunk_rel - ps_base(wg) : 0;                   // PS: the window's chunk class in the step's seed table (0 .. 2; its second chunk: the next)
            const float2 *ct = ctl + ((gi + 2) & 1) * 48 + kcls * 16 + col;
            const float2 C0 = PS ? ct[0] : (RES ? lseed[chunk_rel + 4] : ctab[chunk_rel + 1]), D0 = dtab[off + 2048];
            const float2 P0 = make_float2(C0.x * D0.x - C0.y * D0.y, C0.x * D0.y + C0.y * D0.x);
            float2 P1 = make_float2(0.f, 0.f);
            float k0[4], k1[4];
            if (two) {
                const float2 C1 = PS ? ct[16] : (RES ? lseed[chunk_rel + 5] : ctab[chunk_rel + 2]), D1 = dtab[off - 1024 + 2048];
                P1 = make_float2(C1.x * D1.x - C1.y * D1.y, C1.x * D1.y + C1.y * D1.x);
                const float4 cb = *reinterpret_cast<const float4 *>(lcum + (bo >> 4) * 16 + 4 * q);
                k0[0] = cb.x - c_lo[0]; k0[1] = cb.y - c_lo[1]; k0[2] = cb.z - c_lo[2]; k0[3] = cb.w - c_lo[3];
                k1[0] = c_hi[0] - cb.x; k1[1] = c_hi[1] - cb.y; k1[2] = c_hi[2] - cb.z; k1[3] = c_hi[3] - cb.w;
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) { k0[r] = c_hi[r] - c_lo[r]; k1[r] = 0.f; }
            }
            float pI, pQ, cI, cQ;
            tile_rows(acc, snap, two, scale, k0, k1, P0, P1, pI, pQ, cI, cQ);
            const float dq = cQ - pQ, di = cI - pI;                                  // fmdemod_quadri_cf (libcsdr.c:1040-1071) on (previous = y[Fj+9], current = y[Fj+10])
            const float num = cI * dq - cQ * di, den = cI * cI + cQ * cQ;
            float rd = __builtin_amdgcn_rcpf(den);
            rd = fmaf(fmaf(-den, rd, 1.0f), rd, rd);
            float dval = (den != 0.f) ? (K * num) * rd : 0.f;
            if constexpr (PS) {
                // a retuned stream's first samples of the call: their windows reach into bytes that were rotated at the OLD rate -- evaluated with both tables
                // by k_wfm_lead in front of this kernel
                const long long ja = 4 * (t0 + it) + q - p.j_first;                  // this audio sample's index in the call (column 0)
                if (n_lead > 0 && col0 + col == 0 && ja >= 0 && ja < n_lead) dval = p.lead_d[(size_t)sb * p.lead_stride + ja];      // (n_lead in a register: a vector load here waits behind the DMA ring)
            }
            if constexpr (RES) {                                                     // the first block after a retune: the samples whose windows straddle the two rates
                if (n_lead > 0) {
                    const long long ja = 4 * (t0 + it) + q - p.j_first;
                    if (ja >= 0 && ja < n_lead) dval = rc.lead_d[(size_t)min(sb * 16 + col, last_stream) * rc.lead_stride + ja];
                }
            }
            lout[col * SEQ_OUTP + 4 * wv + q] = dval;                                 // audio 4 * tile + q of stream col
        }
        PROF_T(0)
        const long long wg_n = wg + (long long)TPG * tstride;
        __syncthreads();
        PROF_T(2)
        if (!emit_vec) emit_scalar(gi - 1);                                                                // the previous step's audio: filtered by wave 0 before it came to this barrier
        if (ps_wave) { ps_store(gi + 2); ps_load(wg + 3LL * TPG * tstride); }        // step gi + 2's seeds into the half step gi has finished with; in flight: step gi + 3
        PROF_T(4)
        if (iir_wave && WFM_DIAG < 5) {                                              // this step's 32 samples of stream s0 + col through the de-emphasis, in place
            // (PS: per column -- lane col -- ; the scan runs when every column's range allows it)
            if (PS && gi == 0 && col0 + col == 0) yst = yst_first;                   // the call's first column: the carried state instead of its warm-up's
            const int lo = (gi == 0 && !(RES && n_warm)) ? col_lo(col) : 0;          // samples [lo, hi) of the step exist in this call (warm-up steps: all; RES: a block's first, partial tile
                                                                                     // continues the recurrence of the warm-up over the samples in front of the block -- they are not stored)
            const int hi = gi < 0 ? SPS : max(0, min(SPS, col_hi(col) - gi * SPS));
            if (PS ? __all(lo == 0 && (hi & 3) == 0) : (lo == 0 && (hi & 3) == 0)) {
                float *row = lout + col * SEQ_OUTP + QL * q;                         // this lane's quarter: samples QL q .. QL q + QL - 1
                const int nv = hi;                                                   // valid samples of the step (a multiple of 4 here; all except in a segment's last step)
                float x[QL], z[QL], y[QL];
#pragma unroll
                for (int j = 0; j < QL; j += 2) {                                   // (pairs: QL q + j is even and nv a multiple of 4, a pair is valid or not as a whole)
                    const float2 v = *reinterpret_cast<const float2 *>(row + j);
                    const bool ok = QL * q + j < nv;                                 // samples behind the valid ones do not exist: feed zeros (their outputs are not stored)
                    x[j] = ok ? v.x : 0.f; x[j + 1] = ok ? v.y : 0.f;
                }
                z[0] = p.alpha * x[0];
#pragma unroll
                for (int j = 1; j < QL; j++) z[j] = p.alpha * x[j] + one_minus * z[j - 1];
                // start state of this quarter: S_q = b^(QL q) y_prev + sum_{i<q} b^(QL (q-1-i)) z_end(i)
                const float ze_0 = __shfl(z[QL - 1], col, 64), ze_1 = __shfl(z[QL - 1], col + 16, 64), ze_2 = __shfl(z[QL - 1], col + 32, 64);
                float S = yst;
                if (q == 1) S = bp[QL] * yst + ze_0;
                else if (q == 2) S = b2q * yst + (bp[QL] * ze_0 + ze_1);
                else if (q == 3) S = b3q * yst + (b2q * ze_0 + (bp[QL] * ze_1 + ze_2));
#pragma unroll
                for (int j = 0; j < QL; j++) y[j] = z[j] + bp[j + 1] * S;
#pragma unroll
                for (int j = 0; j < QL; j += 2) *reinterpret_cast<float2 *>(row + j) = make_float2(y[j], y[j + 1]);
                // new carried state = the value after the last VALID sample (nv - 1): it lives in quarter (nv - 1) / QL at position (nv - 1) % QL (odd)
                if (nv > 0) {
                    const int lq = (nv - 1) / QL, lj = (nv - 1) % QL;
                    float ylast = y[QL - 1];
#pragma unroll
                    for (int j = 1; j < QL - 1; j += 2) if (lj == j) ylast = y[j];
                    yst = __shfl(ylast, col + 16 * lq, 64);
                }
            } else {                                                                 // a call's first / last step with a partial tile: sample by sample over the valid range
                float yv = yst;
                if (lane < 16) {
                    float *row = lout + col * SEQ_OUTP;
                    for (int i = lo; i < hi; i++) { yv = p.alpha * row[i] + one_minus * yv; row[i] = yv; }
                }
                yst = __shfl(yv, col, 64);
            }
        }
        PROF_T(5)
#ifdef WFM_PROF
        prof[6] += 1;
#endif
        wg = wg_n; wslot += TPG * tstride; if (wslot >= SEQ_RB) wslot -= SEQ_RB;
    }
#ifdef WFM_PROF
    if (lane == 0) for (int k = 0; k < 8; k++) atomicAdd(&g_wfm_prof[wv][k], (unsigned long long)prof[k]);
    t_last = __builtin_readcyclecounter();
#endif
    __syncthreads();
    if (!emit_vec) emit_scalar(n_grp - 1);
    }
    if (PS && seg_y + 1 == (int)gridDim.y) {                                         // the workgroup that holds the call's last column
        if (iir_lane && col0 + lane == p.n_cols - 1) p.last_out[sb] = yst;
        if (tid < 32 && p.two_T >= 512 && (p.two_T & 15) == 0) {                     // the block's newest 512 bytes -> bytes 512.. of the other head buffer
            const uint4 v = *reinterpret_cast<const uint4 *>(in + (size_t)sb * in_pitch + (p.two_T - 512) + 16 * tid);
            *reinterpret_cast<uint4 *>(p.head_out + (size_t)sb * SEQ_HEAD + 512 + 16 * tid) = v;
        }
    } else
    if (!PS && !RES && seg_y + 1 == (int)gridDim.y) {                                // the call's last segment: what the next call starts from
        if (iir_lane && s0 + lane < p.n_streams) p.last_out[s0 + lane] = yst;
        // the block's newest 512 bytes -> bytes 512.. of the other head buffer (16 streams x 32 pieces of 16 bytes)
        if (tid < 16 * 32 && p.two_T >= 512 && (p.two_T & 15) == 0 && s0 + tid / 32 < p.n_streams) {
            const int srow = tid / 32, piece = tid % 32;
            const uint4 v = *reinterpret_cast<const uint4 *>(sblock + (size_t)srow * in_pitch + (p.two_T - 512) + 16 * piece);
            *reinterpret_cast<uint4 *>(p.head_out + (size_t)(s0 + srow) * SEQ_HEAD + 512 + 16 * piece) = v;
        }
    }
#ifdef WFM_PROF
    if (lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                             // (the wave's stores have left)
        const long long t_exit = __builtin_readcyclecounter();
        atomicAdd(&g_wfm_life[wv][0], (unsigned long long)(t_first - t_entry)); atomicAdd(&g_wfm_life[wv][1], (unsigned long long)(t_last - t_first));
        atomicAdd(&g_wfm_life[wv][2], (unsigned long long)(t_exit - t_last)); atomicAdd(&g_wfm_life[wv][3], 1ULL);
    }
#endif
    if constexpr (!RES) break;
    else {
        // The item's audio is on its way (write-through stores: put16 / put2): once every wave's stores are acknowledged the item is counted in; the count comes
        // back while wave 0 polls the next block's descriptor (res_complete), whose lines it asks for NOW -- the round trip over PCIe runs beside the loaders' last stores.
        if (tid == 0) { const long long t = wall_clock64(); st_body += t - st_t; st_t = t; }
        if (wv == 0) {
            const unsigned long long nx = item + gridDim.x;
            const long long nk = (long long)(nx / (unsigned)rc.n_wsb);
            const uint32_t *dsc = rc.desc + (size_t)(nk % rc.n_slots) * rc.desc_lines * 16;
            const int nd = rc.desc_lines * 16;
#pragma unroll
            for (int j = 0; j < 4; j++) { const int d = lane + 64 * j; pre_v[j] = d < nd ? sys_load(dsc + d) : 0u; }
            pre_valid = true;
        }
        if (rc.fence_mode == 2) __threadfence_system();
        if (wv != 0 || !emit_vec) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (wave 0 stores nothing on the vector path: it need not wait for its descriptor lines here)
        __syncthreads();
        if (tid == 0) {
            if (rc.fence_mode == 3) __threadfence_system();                          // (experiment switch: the L2 write-back that plain stores would need)
            pend_c = __hip_atomic_fetch_add(rc.cnt + res_slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pend = true; pend_slot = res_slot; pend_n = p.n_audio; pend_k = res_k;
        }
        item += gridDim.x;
        if (tid == 0) { const long long t = wall_clock64(); st_done += t - st_t; st_t = t; st_items++; }
    }
    }
}

// the history of a call that produced no audio (a block shorter than one tile's reach): the head buffers still have to roll
__global__ __launch_bounds__(256) void k_wfm_roll_head(const uint8_t *__restrict__ in, size_t in_pitch, long long two_T, const uint8_t *__restrict__ head_in, uint8_t *__restrict__ head_out)
{
    const int s = blockIdx.x, t = threadIdx.x;                                       // 512 bytes per stream: 256 threads x 2 bytes
    for (int k = 2 * t; k < 2 * t + 2; k++) {
        const long long pos = two_T - 512 + k;                                       // byte position in the block; < 0: still history
        head_out[(size_t)s * SEQ_HEAD + 512 + k] = pos >= 0 ? in[(size_t)s * in_pitch + pos] : head_in[(size_t)s * SEQ_HEAD + 512 + (pos + 512)];
    }
}

// Retuned streams (csdr_amd_wfm_set_rate): the call's first audio samples have windows that reach into bytes in front of the block, rotated at the OLD rate
// (shift_addition_cc --fifo picks a new rate up between two reads, csdr.c:881-923).  Those samples -- at most four -- are evaluated here with both tables, one
// wave per (listed stream, sample, which of y[Fj+9] / y[Fj+10]): the chain kernel takes the demodulated value from lead_d.  Seeds: row 0 of the stream's table =
// the chunk in front of the block.
struct LeadParams { int D, L, F; long long B, j_first, c_first; int n_lead, ja0, lead_stride; float *warm; size_t tab_pitch, dtab_stride, ct_stride, head_pitch, head_off; };      // (a ring's tables are shared: strides 0; its history
                                                                                                                                               //  is the previous slot's tail)
__global__ __launch_bounds__(128) void k_wfm_lead(const uint8_t *__restrict__ in, size_t in_pitch, const uint8_t *__restrict__ head, const float *__restrict__ taps,
                                                 const float2 *__restrict__ ctab, const float2 *__restrict__ dtab, const float2 *__restrict__ dtab_old,
                                                 const int *__restrict__ list, float *__restrict__ lead_d, LeadParams p)
{
    __shared__ float2 yv[2];
    const int s = list[blockIdx.y], ja = (int)blockIdx.x + p.ja0, which = threadIdx.x >> 6, lane = threadIdx.x & 63;      // (ja < 0: a ring's warm-up samples in front of the block)
    const long long j = p.j_first + ja, k = (long long)p.F * j + 9 + which;
    const uint8_t *row = in + (size_t)s * in_pitch, *hrow = head + (size_t)s * p.head_pitch + p.head_off;      // (the head's second half:) the 256 samples in front of the block
    const float2 *ct = ctab + (size_t)s * p.ct_stride, *dn = dtab + (size_t)s * p.dtab_stride, *dold = dtab_old + (size_t)s * p.dtab_stride;
    float ai = 0.f, aq = 0.f;
    for (int t = lane; t < p.L; t += 64) {
        const long long n = (long long)p.D * k + t, rel = n - p.B;
        uint32_t vi, vq;
        if (rel < 0) { vi = hrow[2 * (rel + 256)]; vq = hrow[2 * (rel + 256) + 1]; } else { vi = row[2 * rel]; vq = row[2 * rel + 1]; }
        const float2 C = ct[(size_t)((n >> 10) - p.c_first) * p.tab_pitch], Dv = (rel < 0 ? dold : dn)[(int)(n & 1023) + 2048];
        const float2 R = make_float2(C.x * Dv.x - C.y * Dv.y, C.x * Dv.y + C.y * Dv.x);
        const float xi = fmaf((float)vi, 0x1.010102p-7f, -1.0f), xq = fmaf((float)vq, 0x1.010102p-7f, -1.0f);      // v / 127.5 - 1
        const float h = taps[t];
        ai = fmaf(h, xi * R.x - xq * R.y, ai); aq = fmaf(h, xq * R.x + xi * R.y, aq);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { ai += __shfl_xor(ai, o, 64); aq += __shfl_xor(aq, o, 64); }
    if (lane == 0) yv[which] = make_float2(ai, aq);
    __syncthreads();
    if (threadIdx.x == 0) {
        const float K = 0.340447550238101026565118445432744920253753662109375f;
        const float pI = yv[0].x, pQ = yv[0].y, cI = yv[1].x, cQ = yv[1].y;
        const float dq = cQ - pQ, di = cI - pI, num = cI * dq - cQ * di, den = cI * cI + cQ * cQ;
        float rd = __builtin_amdgcn_rcpf(den); rd = fmaf(fmaf(-den, rd, 1.0f), rd, rd);
        const float dv = (den != 0.f) ? (K * num) * rd : 0.f;
        if (ja >= 0) lead_d[(size_t)s * p.lead_stride + ja] = dv; else p.warm[(size_t)s * RES_WARM + (ja + RES_WARM)] = dv;
    }
}

// a ring's retune: the de-emphasis state in front of the block (libcsdr.c:1081-1097) from the RES_WARM samples k_wfm_lead has demodulated there with the old tables
__global__ __launch_bounds__(64) void k_wfm_lead_state(const float *__restrict__ warm, float *__restrict__ state, int n_streams, float alpha)
{
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n_streams) return;
    float y = 0.f;
    for (int i = 0; i < RES_WARM; i++) y = alpha * warm[(size_t)s * RES_WARM + i] + (1 - alpha) * y;
    state[s] = y;
}

} // namespace

namespace csdr_amd {

size_t wfm_mfma_head_bytes(int n_streams) { return (size_t)((n_streams + 15) / 16 * 16) * SEQ_HEAD; }

int wfm_mfma_lead(hipStream_t st, const uint8_t *in, size_t in_pitch, const uint8_t *head, const float *d_taps, const float2 *ctab, size_t tab_pitch, const float2 *d_dtab,
                  const float2 *d_dtab_old, const int *d_list, int n_list, float *d_lead_d, int lead_stride, int D, int L, int F, long long B, long long j_first, int n_lead)
{
    if (n_list <= 0 || n_lead <= 0) return 0;
    LeadParams lp; lp.D = D; lp.L = L; lp.F = F; lp.B = B; lp.j_first = j_first; lp.n_lead = n_lead; lp.tab_pitch = tab_pitch; lp.dtab_stride = 3072;
    lp.ct_stride = 1; lp.head_pitch = SEQ_HEAD; lp.head_off = 512; lp.c_first = (B >> 10) - 1; lp.ja0 = 0; lp.warm = nullptr; lp.lead_stride = lead_stride;
    hipLaunchKernelGGL(k_wfm_lead, dim3(n_lead, n_list), dim3(128), 0, st, in, in_pitch, head, d_taps, ctab, d_dtab, d_dtab_old, d_list, d_lead_d, lp);
    CSDR_LAUNCH_CHECK();
    return 0;
}

// the same for a ring (csdr_amd_wfm_ring_set_rate): one table set and one seed sequence for all streams (ctab[0] = the chunk in front of the block), the samples in front of
// the block = the previous slot's last 512 bytes per row
// ctab[0] = chunk (B >> 10) - 4.  Also the RES_WARM audio samples in front of the block (old tables only) and from them the de-emphasis state there (d_state[stream]): a
// ring's blocks carry no state, and the warm-up the grid would run over those samples uses the NEW weights.
int wfm_mfma_lead_shared(hipStream_t st, const uint8_t *in, size_t in_pitch, const uint8_t *prev, size_t two_T, const float *d_taps, const float2 *ctab, const float2 *d_dtab,
                         const float2 *d_dtab_old, const int *d_list, int n_list, float *d_lead_d, int lead_stride, float *d_warm, float *d_state, float alpha,
                         int D, int L, int F, long long B, long long j_first, int n_lead)
{
    if (n_list <= 0) return 0;
    if (j_first < RES_WARM) return fail_msg(-3, "wfm ring: a retune needs %d audio samples in front of the block", RES_WARM);
    LeadParams lp; lp.D = D; lp.L = L; lp.F = F; lp.B = B; lp.j_first = j_first; lp.n_lead = n_lead; lp.tab_pitch = 1; lp.dtab_stride = 0;
    lp.ct_stride = 0; lp.head_pitch = in_pitch; lp.head_off = two_T - 512; lp.c_first = (B >> 10) - 4; lp.ja0 = -RES_WARM; lp.warm = d_warm; lp.lead_stride = lead_stride;
    hipLaunchKernelGGL(k_wfm_lead, dim3(n_lead + RES_WARM, n_list), dim3(128), 0, st, in, in_pitch, prev, d_taps, ctab, d_dtab, d_dtab_old, d_list, d_lead_d, lp);
    CSDR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_wfm_lead_state, dim3((n_list + 63) / 64), dim3(64), 0, st, d_warm, d_state, n_list, alpha);
    CSDR_LAUNCH_CHECK();
    return 0;
}

int wfm_mfma_launch(hipStream_t st, hipEvent_t ev_begin, hipEvent_t ev_end, const uint8_t *in, size_t in_pitch, const WfmMfmaDevice &dev, const float2 *ctab,
                    int n_streams, int T, long long B, long long j_first, int n_audio, const WfmBackArgs &back, const WfmPerStream *ps)
{
    if (in_pitch * 16 + 4096 >= ((size_t)1 << 32)) return fail_msg(-3, "wfm: in_pitch %zu too large for the matrix-core front end (32-bit row offsets)", in_pitch);
    if (n_audio <= 0) {
        hipLaunchKernelGGL(k_wfm_roll_head, dim3(n_streams), dim3(256), 0, st, in, in_pitch, 2LL * T, back.head_in, back.head_out);
        CSDR_LAUNCH_CHECK();
        return 0;
    }
    SeqParams sp;
    memset((void *)&sp, 0, sizeof sp);
    ResCtl no_res; memset((void *)&no_res, 0, sizeof no_res);
    sp.n_streams = n_streams; sp.B2 = 2 * B; sp.two_T = 2LL * T;
    sp.tile_first = j_first / 4; sp.n_tiles = (int)((j_first + n_audio - 1) / 4 - sp.tile_first + 1);
    sp.stride = dev.tile_stride_bytes; sp.win_off = dev.win_off_bytes; sp.scale = dev.seq_scale;
    // grid: 16-stream blocks x time segments, about one workgroup per CU; a segment is at least 24 tiles (the two warm-up steps = 12 tiles in front of it
    // must stay inside the call; short blocks -- the reference's 16384 samples are 82 tiles -- would otherwise leave three quarters of the CUs idle) and a
    // multiple of SEQ_TPG
    const int n_cu = current_device_cu_count();
    const int n_wsb = (n_streams + 15) / 16;
    int n_seg = (n_cu + n_wsb - 1) / n_wsb; if (n_seg < 1) n_seg = 1;
    if (n_seg > sp.n_tiles / 24) n_seg = sp.n_tiles / 24;
    if (n_seg < 1) n_seg = 1;
    sp.tiles_per_seg = ((sp.n_tiles + n_seg - 1) / n_seg + SEQ_TPG - 1) / SEQ_TPG * SEQ_TPG;
    n_seg = (sp.n_tiles + sp.tiles_per_seg - 1) / sp.tiles_per_seg;
    sp.alpha = back.alpha; sp.last_in = back.last_in; sp.last_out = back.last_out; sp.s16 = back.s16; sp.af = back.af; sp.out_pitch = back.out_pitch;
    sp.j_first = j_first; sp.n_audio = n_audio; sp.head_in = back.head_in; sp.head_out = back.head_out;
    if (ps) {
        // a rate per stream: grid = streams x groups of 16 columns; a column = a whole number of periods (lcm of the tile stride, a 1024-chunk and a 64-sample
        // output line), about four workgroups per CU at most
        const long long ts = sp.stride / 2;                                         // samples per tile (4 audio samples)
        long long per = ts; { long long a = per, b = 1024; while (b) { const long long t = a % b; a = b; b = t; } per = per / a * 1024; }
        int tpp = (int)(per / ts);                                                  // tiles per period
        while ((4 * tpp) % SEQ_LINE) tpp *= 2;
        const int n_per = (sp.n_tiles + tpp - 1) / tpp;
        int n_ss = (4 * n_cu + n_streams - 1) / n_streams; if (n_ss < 1) n_ss = 1;
        int np = (n_per + 16 * n_ss - 1) / (16 * n_ss); if (np < 1) np = 1;
        sp.tiles_per_seg = np * tpp;
        sp.n_cols = (sp.n_tiles + sp.tiles_per_seg - 1) / sp.tiles_per_seg;
        n_seg = (sp.n_cols + 15) / 16;
        sp.col_bytes = (long long)sp.tiles_per_seg * sp.stride; sp.col_chunks = (int)(sp.col_bytes / 2048); sp.col_audio = 4 * sp.tiles_per_seg;
        sp.frag_stride = (size_t)WFM_NK * 3 * 64; sp.tab_pitch = ps->tab_pitch; sp.tab_len = ps->tab_len; sp.scales = ps->d_scales;
        sp.lead_d = ps->d_lead_d; sp.lead_n = ps->d_lead_n; sp.lead_stride = ps->lead_stride;
        if (sp.col_bytes * 16 + 4096 >= (1LL << 32)) return fail_msg(-3, "wfm: block too large for the per-stream kernel's 32-bit row offsets");
        const size_t ldsp = (size_t)16 * SEQ_RP + 16 * SEQ_OUTP * sizeof(float) + (SEQ_NGR + 1) * 16 * sizeof(float) + 2 * 48 * sizeof(float2);
        { const int arc = lds_attr_once((const void *)k_wfm_mfma_seq<true>, ldsp); if (arc) return arc; }
        if (ev_begin && ev_end)
            hipExtLaunchKernelGGL(k_wfm_mfma_seq<true>, dim3(n_streams, n_seg), dim3(64 * SEQ_NW), ldsp, st, ev_begin, ev_end, 0, in, in_pitch, (const v4i *)dev.d_seq_frags, dev.d_seq_cum, dev.d_dtab, ctab, sp, no_res);
        else
            hipLaunchKernelGGL(k_wfm_mfma_seq<true>, dim3(n_streams, n_seg), dim3(64 * SEQ_NW), ldsp, st, in, in_pitch, (const v4i *)dev.d_seq_frags, dev.d_seq_cum, dev.d_dtab, ctab, sp, no_res);
        CSDR_LAUNCH_CHECK();
        if (T < 256 || (T & 7)) {
            hipLaunchKernelGGL(k_wfm_roll_head, dim3(n_streams), dim3(256), 0, st, in, in_pitch, 2LL * T, back.head_in, back.head_out);
            CSDR_LAUNCH_CHECK();
        }
        return 0;
    }
    const size_t lds = (size_t)16 * SEQ_RP + 16 * SEQ_OUTP * sizeof(float) + (SEQ_NGR + 1) * 16 * sizeof(float);
    { const int arc = lds_attr_once((const void *)k_wfm_mfma_seq<false>, lds); if (arc) return arc; }
    // timing events ride on the kernel's own dispatch (start / completion signal of its packet): hipEventRecord in front of and behind it would put two
    // marker packets into the stream, ~10 us of bubbles that the un-profiled path does not have
    if (ev_begin && ev_end)
        hipExtLaunchKernelGGL(k_wfm_mfma_seq<false>, dim3(n_wsb, n_seg), dim3(64 * SEQ_NW), lds, st, ev_begin, ev_end, 0, in, in_pitch, (const v4i *)dev.d_seq_frags, dev.d_seq_cum, dev.d_dtab, ctab, sp, no_res);
    else
        hipLaunchKernelGGL(k_wfm_mfma_seq<false>, dim3(n_wsb, n_seg), dim3(64 * SEQ_NW), lds, st, in, in_pitch, (const v4i *)dev.d_seq_frags, dev.d_seq_cum, dev.d_dtab, ctab, sp, no_res);
    CSDR_LAUNCH_CHECK();
    if (T < 256 || (T & 7)) {      // a block shorter than the history, or a ragged last block: the kernel's epilogue skipped the copy
        hipLaunchKernelGGL(k_wfm_roll_head, dim3(n_streams), dim3(256), 0, st, in, in_pitch, 2LL * T, back.head_in, back.head_out);
        CSDR_LAUNCH_CHECK();
    }
    return 0;
}

// The resident form: `grid` workgroups that walk the ring described by rv until they are told to stop, see nothing for rv.idle_ticks, or are older than rv.life_ticks
// (k_wfm_mfma_seq<false, true>).  ev_end (may be null) is recorded behind the launch: it fires when the last workgroup has left.
int wfm_mfma_launch_resident(hipStream_t st, hipEvent_t ev_end, const WfmMfmaDevice &dev, int n_streams, size_t in_pitch, float alpha, size_t out_pitch, const WfmResident &rv, int grid)
{
    if (in_pitch * 16 + 4096 >= ((size_t)1 << 32)) return fail_msg(-3, "wfm ring: in_pitch %zu too large (32-bit row offsets)", in_pitch);
    SeqParams sp; memset((void *)&sp, 0, sizeof sp);
    sp.n_streams = n_streams; sp.two_T = 2LL * rv.T; sp.stride = dev.tile_stride_bytes; sp.win_off = dev.win_off_bytes; sp.scale = dev.seq_scale;
    sp.alpha = alpha; sp.out_pitch = out_pitch;
    ResCtl rc; memset((void *)&rc, 0, sizeof rc);
    rc.desc = rv.desc; rc.ctrl = rv.ctrl; rc.done = rv.done; rc.cnt = rv.cnt; rc.t_first = rv.t_first; rc.next_item = rv.next_item; rc.exiting = rv.exiting;
    rc.in_ring = rv.in_ring; rc.out_ring = rv.out_ring; rc.in_slot_bytes = rv.in_slot_bytes; rc.out_slot_elems = rv.out_slot_elems;
    rc.n_slots = rv.n_slots; rc.desc_lines = rv.desc_lines; rc.n_wsb = (n_streams + 15) / 16; rc.T = rv.T; rc.D = rv.D; rc.L = rv.L; rc.F = rv.F;
    rc.idle_ticks = rv.idle_ticks; rc.life_ticks = rv.life_ticks; rc.lead_d = rv.lead_d; rc.lead_stride = rv.lead_stride; rc.lead_state = rv.lead_state; rc.stats = rv.stats; rc.fence_mode = rv.fence_mode;
    if (rv.desc_lines < 1 || rv.desc_lines > 16) return fail_msg(-3, "wfm ring: %d descriptor lines", rv.desc_lines);
    const size_t lds = (size_t)16 * SEQ_RP + 16 * SEQ_OUTP * sizeof(float) + (SEQ_NGR + 1) * 16 * sizeof(float) + 8 + (size_t)rv.desc_lines * 14 * sizeof(float);
    { const int arc = lds_attr_once((const void *)k_wfm_mfma_seq<false, true>, lds); if (arc) return arc; }
    hipLaunchKernelGGL((k_wfm_mfma_seq<false, true>), dim3(grid), dim3(64 * SEQ_NW), lds, st, (const uint8_t *)nullptr, in_pitch, (const v4i *)dev.d_seq_frags, dev.d_seq_cum, dev.d_dtab,
                       (const float2 *)nullptr, sp, rc);
    CSDR_LAUNCH_CHECK();
    if (ev_end) CSDR_HIP(hipEventRecord(ev_end, st));
    return 0;
}
int wfm_resident_max_grid() { return current_device_cu_count(); }      // one workgroup per CU: eight waves of up to 256 registers fill a CU's register files

} // namespace csdr_amd

#ifdef WFM_PROF
extern "C" int csdr_amd_debug_wfm_prof(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wfm_prof), sizeof(unsigned long long) * SEQ_NW * 8) != hipSuccess) return -1;
    if (reset) { static unsigned long long z[SEQ_NW * 8]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_wfm_prof), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
extern "C" int csdr_amd_debug_wfm_life(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wfm_life), sizeof(unsigned long long) * SEQ_NW * 4) != hipSuccess) return -1;
    if (reset) { static unsigned long long z[SEQ_NW * 4]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_wfm_life), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif

// Test hook (tests/test_mfma_table_cpu.py): ONE tile of the sequential kernel on the CPU -- the phase-independent weight set, the
// snapshot / lane-group masking at a chunk boundary, the post factors C_m D^e and the prefix-sum offset constants, exactly as
// k_wfm_mfma_seq evaluates them.  n0: global index of the window's first sample (multiple of 8); window: 64*WFM_NK raw u8 bytes from n0;
// ctab2: (cos, sin) of the chunks n0>>10 and +1;  out16: the 16 rows (Re, Im of y[Fj+9], y[Fj+10] for the tile's 4 audio samples).
extern "C" int csdr_amd_debug_wfm_seq_tile(int D, int L, int F, float shift_rate, const float *taps, long long n0, const uint8_t *window,
                                           const float *ctab2, float *out16)
{
    if (!wfm_mfma_supported(D, L, F) || (n0 & 7)) return -1;
    static WfmMfmaTable t; static int cD = 0, cL = 0, cF = 0; static float crate = 0; static std::vector<float> ctaps;
    if (cD != D || cL != L || cF != F || crate != shift_rate || ctaps != std::vector<float>(taps, taps + L)) {
        wfm_mfma_build_table(D, L, F, shift_rate, taps, t); cD = D; cL = L; cF = F; crate = shift_rate; ctaps.assign(taps, taps + L);
    }
    const int off = (int)(n0 & 1023);
    const bool two = off + 32 * WFM_NK > 1024;
    const int bo = 2 * (1024 - off), kb = two ? (bo >> 6) : WFM_NK, hq = (bo >> 4) & 3, gb = bo >> 4;
    const float2 D0 = t.dtab[off + 2048], C0 = make_float2(ctab2[0], ctab2[1]);
    const float2 P0 = make_float2(C0.x * D0.x - C0.y * D0.y, C0.x * D0.y + C0.y * D0.x);
    float2 P1 = make_float2(0.f, 0.f);
    if (two) { const float2 D1 = t.dtab[off - 1024 + 2048], C1 = make_float2(ctab2[2], ctab2[3]); P1 = make_float2(C1.x * D1.x - C1.y * D1.y, C1.x * D1.y + C1.y * D1.x); }
    const int ngr = 4 * WFM_NK;
    float u[16], v[16];
    for (int r = 0; r < 16; r++) {
        long acc[3] = {0, 0, 0}, snap[3] = {0, 0, 0};
        auto step = [&](int ks, int kg_lo, int kg_hi) {
            for (int kg = kg_lo; kg < kg_hi; kg++) for (int b = 0; b < 16; b++) {
                const int x = (int)(int8_t)(window[64 * ks + 16 * kg + b] ^ 0x80);
                for (int l = 0; l < 3; l++) acc[l] += (long)t.seq_frags[((size_t)(ks * 3 + l) * 64 + (16 * kg + r)) * 16 + b] * x;
            }
        };
        for (int ks = 0; ks < WFM_NK; ks++) {
            if (ks == kb) { step(ks, 0, hq); for (int l = 0; l < 3; l++) snap[l] = acc[l]; step(ks, hq, 4); }
            else step(ks, 0, 4);
        }
        const float clo = t.seq_cum[r], chi = t.seq_cum[(size_t)ngr * 16 + r];
        if (two) {
            const float cb = t.seq_cum[(size_t)gb * 16 + r];
            u[r] = fmaf(fmaf((float)snap[0], 65536.0f, fmaf((float)snap[1], 256.0f, (float)snap[2])), t.seq_scale, cb - clo);
            v[r] = fmaf(fmaf((float)(acc[0] - snap[0]), 65536.0f, fmaf((float)(acc[1] - snap[1]), 256.0f, (float)(acc[2] - snap[2]))), t.seq_scale, chi - cb);
        } else {
            u[r] = fmaf(fmaf((float)acc[0], 65536.0f, fmaf((float)acc[1], 256.0f, (float)acc[2])), t.seq_scale, chi - clo);
            v[r] = 0.f;
        }
    }
    for (int k = 0; k < 8; k++) {
        out16[2 * k] = P0.x * u[2 * k] - P0.y * u[2 * k + 1] + (P1.x * v[2 * k] - P1.y * v[2 * k + 1]);
        out16[2 * k + 1] = P0.x * u[2 * k + 1] + P0.y * u[2 * k] + (P1.x * v[2 * k + 1] + P1.y * v[2 * k]);
    }
    return 0;
}
