// wfm_mfma.hip -- matrix-core front end of the fused WFM chain (u8 IQ -> rotate -> decimating FIR -> FM demod).
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
//   * B operand  = the raw input: lane l holds 16 consecutive bytes of stream (l%16) -> one global_load_dwordx4 per
//     64-byte K-step, XOR 0x80 to recentre; no LDS, no conversion, no rotation instructions at all;
//   * A operand  = the weights a*h*R split into three signed base-256 digits (23-bit fixed point; 2.8e-7 end-to-end
//     error measured against the oracle), held in registers and reused for 64 streams;
//   * 16 rows    = {Re,Im} x {y[Fj+9], y[Fj+10]} x 4 audio samples -> the quadrature demodulator is lane local.
// The band wastes ~5x MACs, but i8 MFMA has ~50x the rate of the f32 VALU path; the kernel becomes bound by the
// input stream (measured access pattern: 5.2-6.1 TB/s, tools/probes/probe_mfma_i8.hip).
//
// Rotator model: shift_addition_cc restarts its float32 phasor at every 1024-sample chunk from cos/sin of a float
// phase (libcsdr_gpl.c:33-35) and advances it by multiplying with the ROUNDED (cos d, sin d) (:44-45), so inside a
// chunk R[n] = C_m * D[n mod 1024], D[k] = (cosdelta_f32 + j sindelta_f32)^k (evaluated in double; the float
// recurrence's own rounding noise is 8.7e-7 RMS).  C_m is exact per chunk (host float phase bookkeeping) and is
// applied AFTER the matrix product, so the weight table depends only on the window's offset inside a chunk: it is
// periodic (128 tile phases for D*F = 50) and built once per filter, not per block.  A window that straddles a chunk
// boundary accumulates the two sides separately (acc0 / acc1) and combines them with C_m and C_{m+1}.
#include "common.hpp"
#include "wfm_mfma.hpp"
#include <math.h>
#include <stdlib.h>
#include <complex>
#include <vector>
using namespace csdr_amd;

namespace csdr_amd {

static int gcd_i(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

bool wfm_mfma_supported(int D, int L, int F)
{
    if ((D * F) & 1) return false;                                  // tile stride 8*D*F bytes must keep 16-byte alignment
    const int win_off = (18 * D) & ~15;
    const int span_bytes = 2 * (D * (3 * F + 10) + L) - win_off;     // last row ends at sample D*(3F+10)+L-1 (tile-relative)
    if (span_bytes > 64 * WFM_NK) return false;                     // window must fit the 8 K-steps (<= 256 samples: at most one chunk boundary)
    if (D + L + 3 * D * F + 8 > WFM_HIST) return false;              // a block's first tile may reach this far into the history
    return true;
}

// Builds the periodic weight table: ONE weight set per tile phase.  A window that contains a shift_addition_cc chunk boundary
// (25 % of the phases) is split at K-step granularity: K-steps before the boundary belong to chunk m, K-steps after it to chunk
// m+1, and the one K-step `kb` that contains the boundary appears twice -- its chunk-m samples in the extra fragment (index
// WFM_NK*3 + digit), its chunk-(m+1) samples in its regular slot.  The kernels run one accumulator chain, snapshot it after the
// extra fragment and get the two sides as (snapshot, total - snapshot): exact in int32, 27 MFMAs instead of 2 x 24.
//   frags : [n_phases][WFM_NFRAG][64 lanes] int8x16        (lane l: row l%16, K bytes 16*(l/16) .. +15 of that K-step)
//   kb_of : [n_phases]                                     (WFM_NK when the window lies inside one chunk)
//   consts: [n_phases][2 sides][16 rows] float             (the +1/255 offset of u8->float through the filter)
void wfm_mfma_build_table(int D, int L, int F, float shift_rate, const float *taps, WfmMfmaTable &t)
{
    t.D = D; t.L = L; t.F = F;
    t.tile_stride_bytes = 8 * D * F;
    t.win_off_bytes = (18 * D) & ~15;
    t.n_phases = 1024 / gcd_i((4 * D * F) % 1024 ? (4 * D * F) % 1024 : 1024, 1024);
    const float rate2 = shift_rate * 2, inc = rate2 * PI_F;         // libcsdr_gpl.c:83-86
    const float sd = (float)sin((double)inc), cd = (float)cos((double)inc);
    const std::complex<double> d((double)cd, (double)sd);
    const double mag = std::abs(d), ang = std::arg(d);
    std::vector<std::complex<double>> Dk(1024);
    for (int k = 0; k < 1024; k++) Dk[k] = std::polar(pow(mag, k), ang * k);
    const double a = 2.0 / 255.0, c0 = 1.0 / 255.0;
    double gmax = 0;
    for (int k = 0; k < L; k++) gmax = fmax(gmax, fabs(a * (double)taps[k]) * 1.0000005);
    if (gmax == 0) gmax = 1;
    const double qscale = 4194304.0 / gmax;                          // 2^22: three balanced base-256 digits stay inside int8
    t.scale = (float)(gmax / 4194304.0);
    const size_t ph_bytes = (size_t)WFM_FRAG_V4 * 16;
    t.frags.assign((size_t)t.n_phases * ph_bytes, 0);
    t.consts.assign((size_t)t.n_phases * 32, 0.f);
    t.kb_of.assign((size_t)t.n_phases, WFM_NK);
    const int base_off_samples = t.win_off_bytes / 2;
    for (int ph = 0; ph < t.n_phases; ph++) {
        const long s0 = (long)4 * D * F * ph + base_off_samples;     // window base sample in the periodic frame
        const long chunk0 = s0 / 1024;
        const long bb = 2 * ((chunk0 + 1) * 1024 - s0);              // byte offset of the next chunk's first sample inside the window
        const int kb = bb < 64 * WFM_NK ? (int)(bb / 64) : WFM_NK;
        t.kb_of[ph] = kb;
        int8_t *fr = t.frags.data() + (size_t)ph * ph_bytes;
        float *cst = t.consts.data() + (size_t)ph * 32;
        for (int r = 0; r < 16; r++) {
            const int q = r / 4, which = (r % 4) / 2, comp = r % 2;
            const long off = (long)D * (F * q + 9 + which) - base_off_samples;        // row's first sample relative to the window base
            std::complex<double> csum[2] = {0, 0};
            for (int tp = 0; tp < L; tp++) {
                const long rel = off + tp, g = s0 + rel;
                const int side = (int)(g / 1024 - chunk0);
                const std::complex<double> G = a * (double)taps[tp] * Dk[g % 1024];
                csum[side] += (double)taps[tp] * Dk[g % 1024];
                for (int c = 0; c < 2; c++) {
                    // real form of (Gr + j Gi)(I + j Q): Re row takes (Gr, -Gi) on (I, Q); Im row takes (Gi, Gr)
                    const double val = comp == 0 ? (c == 0 ? G.real() : -G.imag()) : (c == 0 ? G.imag() : G.real());
                    const long colb = 2 * rel + c;
                    const int ks = (int)(colb / 64), b = (int)(colb % 64);
                    long qv = lrint(val * qscale);
                    const int w2 = (int)(((qv + 128) % 256 + 256) % 256) - 128; qv = (qv - w2) / 256;
                    const int w1 = (int)(((qv + 128) % 256 + 256) % 256) - 128; qv = (qv - w1) / 256;
                    const int w0 = (int)qv;
                    const int lane = 16 * (b / 16) + r, byte = b % 16;
                    const int dig[3] = {w0, w1, w2};
                    // side 0 inside the boundary K-step goes to the extra fragment; everything else to its regular slot
                    // (ks < kb is side 0 and ks > kb is side 1 by construction)
                    const bool extra = (ks == kb && side == 0);
                    for (int l = 0; l < 3; l++) fr[((size_t)(extra ? WFM_NK * 3 + l : ks * 3 + l) * 64 + lane) * 16 + byte] = (int8_t)dig[l];
                }
            }
            for (int p = 0; p < 2; p++) {
                const std::complex<double> k = std::complex<double>(1, 1) * csum[p] * c0;
                cst[p * 16 + r] = (float)(comp == 0 ? k.real() : k.imag());
            }
        }
    }
    // ---- phase-independent form (k_wfm_mfma_seq): R[n0 + t] = [C_m D^(n0 - 1024 m)] D^t, so the weights a h D^t are the same for every tile
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

struct MfmaParams {
    int n_streams, T;                 // complex samples in this block
    long long B;                      // global sample index of the block start (multiple of 1024)
    long long j_first; int n_audio;   // audio samples produced by this call: j_first .. j_first+n_audio-1
    long long tile_first; int n_tiles, tiles_per_wave;
    long long tile_first_b; int n_tiles_b; int two_ranges;   // edge launch: blockIdx.z picks [tile_first, +n_tiles) or [tile_first_b, +n_tiles_b), no z segmentation
    long long tile_out0;              // tile whose 4 audio samples land at demod[stream][0..3]
    int tile_stride_bytes, win_off_bytes, n_phases;
    float scale;
};

__device__ __forceinline__ float combine_digits(int a0, int a1, int a2)
{   // exact integers (<= 22 bits each) recombined in float: value = a0*65536 + a1*256 + a2
    return fmaf((float)a0, 65536.0f, fmaf((float)a1, 256.0f, (float)a2));
}

// The banded product of one (tile, 16 streams): ONE accumulator chain per digit over the window's K-steps.  When the window contains
// a 1024-chunk boundary (kb < WFM_NK, wave uniform) the boundary K-step contributes twice -- first its chunk-m samples (extra
// fragment), then, after the chain has been snapshotted, its chunk-(m+1) samples -- so that  side 0 = snap,  side 1 = acc - snap.
__device__ __forceinline__ void tile_product(const v4i (&A)[WFM_NFRAG], const v4i (&Bf)[WFM_NK], int kb, v4i (&acc)[3], v4i (&snap)[3])
{
#pragma unroll
    for (int ks = 0; ks < WFM_NK; ks++) {
        if (ks == kb) {
#pragma unroll
            for (int l = 0; l < 3; l++) acc[l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[WFM_NK * 3 + l], Bf[ks], acc[l], 0, 0, 0);
#pragma unroll
            for (int l = 0; l < 3; l++) snap[l] = acc[l];
        }
#pragma unroll
        for (int l = 0; l < 3; l++) acc[l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[ks * 3 + l], Bf[ks], acc[l], 0, 0, 0);
    }
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

// B operand of one (tile, stream group): 8 x 16 raw bytes per lane straight from the input rows.
// EDGE = false: the window lies inside this block (no checks).  EDGE = true: it reaches into the history kept from
// the previous block and/or beyond the ragged end of the last block (only the first/last few tiles of a call).
template <bool EDGE>
__device__ __forceinline__ void load_B(v4i (&Bf)[WFM_NK], const uint8_t *__restrict__ in, size_t in_pitch, const uint8_t *__restrict__ hist,
                                       int stream, long long wbr, long long two_T, int q)
{
    const uint8_t *row = in + (size_t)stream * in_pitch;
    if (!EDGE) {
        const uint8_t *src = row + wbr + 16 * q;
#pragma unroll
        for (int ks = 0; ks < WFM_NK; ks++) Bf[ks] = *reinterpret_cast<const v4i *>(src + 64 * ks);
    } else {
        const uint8_t *hrow = hist + (size_t)stream * (2 * WFM_HIST);
        // All fetches of the window in flight at once, from an always-valid address chosen per lane (history / block / a dummy), then a select: with one
        // conditional fetch per K-step in a rolled loop the edge launch was 32 serial memory round trips per wave (22 us for a handful of tiles).
        bool ragged = false;
#pragma unroll
        for (int ks = 0; ks < WFM_NK; ks++) {
            const long long off = wbr + 64 * ks + 16 * q;
            const bool in_hist = off < 0 && off >= -2 * WFM_HIST, in_blk = off >= 0 && off + 16 <= two_T;
            const uint8_t *src = in_hist ? hrow + (off + 2 * WFM_HIST) : row + (in_blk ? off : 0);
            const v4i v = *reinterpret_cast<const v4i *>(src);
            const v4i z = {(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};
            Bf[ks] = (in_hist || in_blk) ? v : z;
            ragged = ragged || (off >= 0 && off < two_T && off + 16 > two_T);
        }
        if (ragged) {                                                   // ragged end of the last block: byte-wise (a block that is not a multiple of 1024 samples ends the stream)
#pragma unroll
            for (int ks = 0; ks < WFM_NK; ks++) {
                const long long off = wbr + 64 * ks + 16 * q;
                if (off >= 0 && off < two_T && off + 16 > two_T) {
                    uint32_t w[4] = {0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};
                    for (int k = 0; k < 16; k++) if (off + k < two_T) { const uint32_t by = row[off + k]; w[k / 4] = (w[k / 4] & ~(0xffu << (8 * (k & 3)))) | (by << (8 * (k & 3))); }
                    Bf[ks] = v4i{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
                }
            }
        }
    }
}

// One wave = 64 streams (4 groups of 16) x ONE TILE PHASE: it owns the tiles ti = ph, ph + n_phases, ph + 2 n_phases, ...
// of its launch range.  All those tiles use the same weights, so the 24 (48 when the phase's window contains a chunk
// boundary) int8x16 weight fragments are loaded ONCE per wave and stay in registers for the whole launch.  (With
// consecutive tiles per wave the weights had to be re-read for every tile: 5.9 GB per step against 4.9 GB of input, and the
// kernel sat at 1.65 ms = the sum of both streams at the ~7 TB/s the memory side delivers; see profiles/r1_notes.md.)
//   * input  : ring of 4 B-operand buffers (one per stream group); a group's buffer is refilled with the window of the
//              wave's NEXT tile as soon as its MFMAs have been issued (4 x 8 KiB per wave in flight, one wave per SIMD);
//   * grid   : x = stream block, y = tile phase, z = segment of the tile range.  Blocks are placed on XCD (linear id % 8)
//              and gridDim.x is a multiple of 8 for >= 512 streams, so every phase of one stream block runs on the same XCD:
//              the 28 % window overlap between neighbouring tiles (different waves) is served by that XCD's L2;
//   * output : the quadrature demodulator is lane local; 4 audio samples per stream leave as one aligned 16-byte store.
template <bool EDGE>
__global__ __launch_bounds__(64) void k_wfm_mfma(const uint8_t *__restrict__ in, size_t in_pitch, const uint8_t *__restrict__ hist,
                                                 const v4i *__restrict__ frags, const float *__restrict__ consts, const int *__restrict__ kb_of,
                                                 const float2 *__restrict__ ctab, float *__restrict__ demod, size_t demod_pitch, MfmaParams p)
{
    const int lane = threadIdx.x, col = lane & 15, q = lane >> 4;
    const int ph = blockIdx.y;
    // tiles of this phase inside [tile_first, tile_first + n_tiles): ti = t0 + m * n_phases, m in this segment
    const bool rb = p.two_ranges && blockIdx.z == 1;
    const long long r_first = rb ? p.tile_first_b : p.tile_first;
    const long long t_lim = r_first + (rb ? p.n_tiles_b : p.n_tiles);
    long long t0 = r_first + (((long long)ph - r_first) % p.n_phases + p.n_phases) % p.n_phases;
    if (t0 >= t_lim) return;
    const long long m_total = (t_lim - 1 - t0) / p.n_phases + 1;
    const int zseg = p.two_ranges ? 1 : (int)gridDim.z, zidx = p.two_ranges ? 0 : (int)blockIdx.z;
    const long long m_per = (m_total + zseg - 1) / zseg;
    const long long m_begin = (long long)zidx * m_per;
    long long m_end = m_begin + m_per; if (m_end > m_total) m_end = m_total;
    if (m_begin >= m_end) return;
    const long long step = (long long)p.n_phases * p.tile_stride_bytes;               // bytes between this wave's consecutive windows
    const long long two_T = 2LL * p.T, B2 = 2 * p.B;
    const int stream_base = blockIdx.x * 64 + col, last_stream = p.n_streams - 1;
    // ---- weights: once per wave
    const int kb = __builtin_amdgcn_readfirstlane(kb_of[ph]);
    const bool two = kb < WFM_NK;                                                     // the window contains a 1024-chunk boundary
    v4i A0[WFM_NFRAG];
    {
        const v4i *fa = frags + (size_t)ph * WFM_FRAG_V4 + lane;
#pragma unroll
        for (int s = 0; s < WFM_NFRAG; s++) A0[s] = fa[s * 64];
    }
    const float4 k0v = *reinterpret_cast<const float4 *>(consts + (size_t)ph * 32 + 4 * q);
    const float4 k1v = *reinterpret_cast<const float4 *>(consts + (size_t)ph * 32 + 16 + 4 * q);
    const float k0[4] = {k0v.x, k0v.y, k0v.z, k0v.w}, k1[4] = {k1v.x, k1v.y, k1v.z, k1v.w};
    // ---- per-group row pointers, computed once (the hot loop only adds the running window offset)
    const uint8_t *rowp[4]; float *dstp[4]; int sidx[4];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        sidx[g] = stream_base + 16 * g;
        const int sc = min(sidx[g], last_stream);
        rowp[g] = in + (size_t)sc * in_pitch + 16 * q;
        dstp[g] = demod + (size_t)sc * demod_pitch;
    }
    // ---- input ring
    long long ti = t0 + m_begin * p.n_phases;
    long long wbr = ti * p.tile_stride_bytes + p.win_off_bytes - B2;                   // window base relative to this block's first byte
    v4i Bq[4][WFM_NK];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        if (!EDGE) {
#pragma unroll
            for (int ks = 0; ks < WFM_NK; ks++) Bq[g][ks] = *reinterpret_cast<const v4i *>(rowp[g] + wbr + 64 * ks);
        } else load_B<true>(Bq[g], in, in_pitch, hist, min(sidx[g], last_stream), wbr, two_T, q);
    }
    const float K = 0.340447550238101026565118445432744920253753662109375f;
    for (long long m = m_begin; m < m_end; m++, ti += p.n_phases, wbr += step) {
        const bool refill = m + 1 < m_end;
        const long long chunk_rel = ((wbr + B2) >> 11) - (B2 >> 11);                  // chunk of the window base relative to the block; -1 = history
        const float2 C0 = ctab[chunk_rel + 1], C1 = ctab[chunk_rel + 2];
        const long long out_off = 4 * (ti - p.tile_out0);
#pragma unroll
        for (int g = 0; g < 4; g++) {
#pragma unroll
            for (int ks = 0; ks < WFM_NK; ks++) Bq[g][ks] ^= (int)0x80808080;          // u8 - 128 as int8, in place
            v4i acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}, snap[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
            tile_product(A0, Bq[g], kb, acc, snap);
            if (refill) {
                if (!EDGE) {
#pragma unroll
                    for (int ks = 0; ks < WFM_NK; ks++) Bq[g][ks] = *reinterpret_cast<const v4i *>(rowp[g] + (wbr + step) + 64 * ks);
                } else load_B<true>(Bq[g], in, in_pitch, hist, min(sidx[g], last_stream), wbr + step, two_T, q);
            }
            // lane (col, q): rows 4q..4q+3 = Re/Im of y[Fj+9], Re/Im of y[Fj+10] for audio j = 4*ti+q of stream col;  y = C_m u0 + C_{m+1} u1
            float pI, pQ, cI, cQ;
            tile_rows(acc, snap, two, p.scale, k0, k1, C0, C1, pI, pQ, cI, cQ);
            // fmdemod_quadri_cf (libcsdr.c:1040-1071) on (previous = y[Fj+9], current = y[Fj+10])
            const float dq = cQ - pQ, di = cI - pI;
            const float num = cI * dq - cQ * di, den = cI * cI + cQ * cQ;
            const float a = (den != 0.f) ? (K * num) / den : 0.f;
            const float a1 = __shfl(a, col + 16, 64), a2 = __shfl(a, col + 32, 64), a3 = __shfl(a, col + 48, 64);
            if (q == 0 && sidx[g] < p.n_streams)
                *reinterpret_cast<float4 *>(dstp[g] + out_off) = make_float4(a, a1, a2, a3);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Workgroup variant: 4 waves = 4 CONSECUTIVE tile phases (a "quad": 16 audio samples) x 32 streams.
// PMC on the per-wave kernel above (profiles/r1_pmc_traffic.json): 1.49 x the algorithmic bytes cross the fabric -- the 512-byte
// windows of neighbouring tiles overlap by 112 bytes (28 %) and live in different waves, and the 16-byte demod stores are
// written back as partial lines (2.2 x write amplification).  Here the quad's input (3*stride + 512 = 1712 bytes per stream)
// is fetched ONCE per workgroup with row-contiguous 16-byte loads into LDS (double buffered: the next quad is in flight while
// the current one is multiplied), each wave takes its window from LDS with ds_read_b128 (row pitch 107 x 16 B: odd, so the 16
// streams of a group fall into different bank slots), and the 16 audio samples of a stream leave as one contiguous 64-byte row.
// Every wave still owns one tile phase, so the weights stay register resident.
struct WgParams {
    int n_streams; long long B2; long long quad_first; int n_quads; long long tile_out0;
    int stride, win_off, n_phases, row_bytes; float scale;
};

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// SB = streams per workgroup (SB/16 groups per wave), NB = input ring depth in quads (NB-1 quads in flight).
// Measured (profiles/r1_notes.md): SB=32/NB=2 streams 4.4 TB/s with one quad (54.8 KB per CU) in flight -- latency bound;
// SB=16/NB=5 keeps 4 quads (110 KB per CU) in flight.
template <int SB, int NB>
__global__ __launch_bounds__(256) void k_wfm_mfma_wg(const uint8_t *__restrict__ in, size_t in_pitch,
                                                     const v4i *__restrict__ frags, const float *__restrict__ consts, const int *__restrict__ kb_of,
                                                     const float2 *__restrict__ ctab, float *__restrict__ demod, size_t demod_pitch, WgParams p)
{
    extern __shared__ float4 lds_raw[];
    uint8_t *lds_in = reinterpret_cast<uint8_t *>(lds_raw);                         // NB quad buffers, rows back to back inside each
    constexpr int quad_bytes = 4 * ((SB * 107 + 255) / 256) * 1024;                 // buffer stride = what 4 waves x DMA_PW x 1 KiB cover (>= SB*1712)
    float *lds_out = reinterpret_cast<float *>(lds_in + NB * quad_bytes);           // 2 x SB x 16 floats
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, col = lane & 15, q = lane >> 4;
    const int n_qph = p.n_phases / 4;                                               // quad phases
    const int qph = blockIdx.x;                                                     // x = quad phase: the 32 phases of a stream block are co-resident and
                                                                                    // together sweep each input row contiguously (DRAM page locality)
    // quads of this quad-phase inside [quad_first, quad_first + n_quads): Qg = q0 + m * n_qph, m in this segment
    const long long q_lim = p.quad_first + p.n_quads;
    const long long q0 = p.quad_first + (((long long)qph - p.quad_first) % n_qph + n_qph) % n_qph;
    if (q0 >= q_lim) return;
    const long long m_total = (q_lim - 1 - q0) / n_qph + 1;
    const long long m_per = (m_total + gridDim.z - 1) / gridDim.z;
    const long long m_begin = (long long)blockIdx.z * m_per;
    long long m_end = m_begin + m_per; if (m_end > m_total) m_end = m_total;
    if (m_begin >= m_end) return;
    const int ph = 4 * qph + w;                                                     // this wave's tile phase
    const int last_stream = p.n_streams - 1;
    // persistent over stream blocks: this workgroup owns stream blocks blockIdx.y, blockIdx.y + gridDim.y, ... and walks the item
    // sequence (stream block, quad m) without draining the input ring in between; the weights are loaded once per WORKGROUP LIFETIME
    const int n_wsb = (p.n_streams + SB - 1) / SB;
    const int my_sb = (n_wsb - (int)blockIdx.y + (int)gridDim.y - 1) / (int)gridDim.y;      // stream blocks of this workgroup
    if (my_sb <= 0) return;
    const long long M = m_end - m_begin, n_items = (long long)my_sb * M;
    // ---- weights: once per wave
    const int kb = __builtin_amdgcn_readfirstlane(kb_of[ph]);
    const bool two = kb < WFM_NK;                                                     // the window contains a 1024-chunk boundary
    v4i A0[WFM_NFRAG];
    {
        const v4i *fa = frags + (size_t)ph * WFM_FRAG_V4 + lane;
#pragma unroll
        for (int s = 0; s < WFM_NFRAG; s++) A0[s] = fa[s * 64];
    }
    const float4 k0v = *reinterpret_cast<const float4 *>(consts + (size_t)ph * 32 + 4 * q);
    const float4 k1v = *reinterpret_cast<const float4 *>(consts + (size_t)ph * 32 + 16 + 4 * q);
    const float k0[4] = {k0v.x, k0v.y, k0v.z, k0v.w}, k1[4] = {k1v.x, k1v.y, k1v.z, k1v.w};
    const float K = 0.340447550238101026565118445432744920253753662109375f;
    // ---- input staging by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B land at M0 + 16*lane, no VGPR round trip).
    // A quad buffer is one contiguous run of SB x n_cols sixteen-byte pieces; wave w issues pieces (DMA_PW*w + k)*64 + lane.
    const int n_cols = p.row_bytes / 16;                                            // 107
    const int n_pieces = SB * n_cols;
    constexpr int DMA_PW = (SB * 107 + 255) / 256;                                  // DMA instructions per wave per quad (row_bytes = 1712)
    int prow[DMA_PW], pcol[DMA_PW];                                                 // per-lane (row, 16-byte column) of each piece (iteration invariant)
#pragma unroll
    for (int k = 0; k < DMA_PW; k++) {
        int P = (w * DMA_PW + k) * 64 + lane;
        if (P >= n_pieces) P = 0;                                                   // lanes past the end re-read piece 0 into the pad area
        prow[k] = P / n_cols; pcol[k] = 16 * (P - prow[k] * n_cols);
    }
    const long long quad_step = (long long)n_qph * 4 * p.stride;                    // bytes between this workgroup's consecutive quads
    const long long Qg0 = q0 + m_begin * n_qph;
    const long long wq0 = Qg0 * 4 * p.stride + p.win_off - p.B2;                    // first quad's window base relative to the block start
    // Inline asm on purpose: hipcc models __builtin_amdgcn_global_load_lds as a store to LDS and puts `s_waitcnt vmcnt(0)` in
    // front of the ds_reads of the OTHER buffers (it cannot prove they do not alias), serialising the DMA with the math.
    // Completion is counted by hand: every wave issues exactly DMA_PW instructions per quad, VMEM returns in order, so
    // `vmcnt(DMA_PW * n)` leaves at most the n newest quads in flight (wave 0's demod store sits in the same queue, which only
    // makes its wait slightly conservative).
    const uint32_t lds_in_addr = (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t *)lds_in;
    // DMA cursor: the next item to fetch (stream block cs, quad cm); advances independently of the compute position
    int cs = blockIdx.y; long long cm = 0, c_issued = 0;
    // per-lane source row of each DMA instruction for the cursor's stream block: recomputed once per stream block (M items), so the
    // per-item address is one 64-bit add instead of a 64-bit multiply-add (VALU instructions are on every wave's serial chain)
    const uint8_t *rowp[DMA_PW];
    auto set_rows = [&](int sb) {
#pragma unroll
        for (int k = 0; k < DMA_PW; k++) rowp[k] = in + (long long)min(sb * SB + prow[k], last_stream) * (long long)in_pitch + pcol[k];
    };
    set_rows(cs);
    auto dma_next = [&](int buf) {
        const uint32_t ldst = __builtin_amdgcn_readfirstlane((int)(lds_in_addr + buf * quad_bytes + (w * DMA_PW) * 1024));
        const long long base = wq0 + cm * quad_step;
#pragma unroll
        for (int k = 0; k < DMA_PW; k++) {                                          // every lane executes every instruction (fixed count per wave)
            const uint8_t *gp = rowp[k] + base;
            const uint32_t la = __builtin_amdgcn_readfirstlane((int)(ldst + k * 1024));
            uint32_t keep;
            // nt: the input is read exactly once, by one CU (MI355X_MICROARCH.md row nt-weights; measured here 1.135 -> 1.110 ms)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gp), "s"(la) : "memory");
        }
        c_issued++;
        if (++cm == M) { cm = 0; cs += gridDim.y; set_rows(cs); }
    };
    auto wait_newer = [&](long long newer) {                                        // all but the `newer` most recent quads have landed
        switch ((int)newer) {
            case 0: wait_vmcnt<0>(); break;
            case 1: wait_vmcnt<DMA_PW>(); break;
            case 2: wait_vmcnt<DMA_PW * 2>(); break;
            case 3: wait_vmcnt<DMA_PW * 3>(); break;
            default: wait_vmcnt<(DMA_PW * 4 < 63 ? DMA_PW * 4 : DMA_PW * 3)>(); break;
        }
    };
    // Ring discipline: quad n lives in buffer n % NB.  Prologue: quads 0 .. NB-1 in flight; at the end of iteration m (after the
    // barrier) buffer m % NB is free and takes quad m + NB.  At the wait point of iteration m the quads issued beyond m+1 are
    // m+2 .. min(m+NB-1, last): that many may stay in flight.
    for (int k = 0; k < NB; k++) if (c_issued < n_items) dma_next(k);
    {
        long long newer = c_issued - 1; if (newer > 4) newer = 4; if (DMA_PW * 4 >= 63 && newer > 3) newer = 3;
        wait_newer(newer);
    }
    __syncthreads();
    int buf = 0;
    int s0 = blockIdx.y * SB; long long m = 0, Qg = Qg0, wq = wq0;                  // compute position
    for (long long it = 0; it < n_items; it++) {
        const long long wb2 = wq + p.B2 + (long long)w * p.stride;                  // global byte index of this wave's window base
        const long long chunk_rel = (wb2 >> 11) - (p.B2 >> 11);
        const float2 C0 = ctab[chunk_rel + 1], C1 = ctab[chunk_rel + 2];
        const uint8_t *lrow = lds_in + buf * quad_bytes + w * p.stride + 16 * q;
        float *lout = lds_out + (int)(it & 1) * (SB * 16);
#pragma unroll
        for (int g = 0; g < SB / 16; g++) {
            v4i Bf[WFM_NK];
            const uint8_t *src = lrow + (16 * g + col) * p.row_bytes;
#pragma unroll
            for (int ks = 0; ks < WFM_NK; ks++) Bf[ks] = *reinterpret_cast<const v4i *>(src + 64 * ks) ^ (int)0x80808080;
            v4i acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}, snap[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
            tile_product(A0, Bf, kb, acc, snap);
            float pI, pQ, cI, cQ;
            tile_rows(acc, snap, two, p.scale, k0, k1, C0, C1, pI, pQ, cI, cQ);
            const float dq = cQ - pQ, di = cI - pI;
            const float num = cI * dq - cQ * di, den = cI * cI + cQ * cQ;
            // K*num/den with a Newton-refined reciprocal (v_rcp_f32 + one step: ~1 ulp) instead of the 10-instruction IEEE division sequence
            float rd = __builtin_amdgcn_rcpf(den);
            rd = fmaf(fmaf(-den, rd, 1.0f), rd, rd);
            lout[(16 * g + col) * 16 + 4 * w + q] = (den != 0.f) ? (K * num) * rd : 0.f;      // audio 4*ti+q of stream 16g+col
        }
        // quad m+1 must have landed before anyone passes the barrier; quads m+2 .. m+NB-1 stay in flight.  After the barrier this
        // quad's buffer is free and takes quad m+NB: the input stream never pauses.
        {
            long long left = c_issued - 1 - (it + 1);                               // items issued beyond it+1
            if (left < 0) left = 0;
            if (left > 4) left = 4;
            if (DMA_PW * 4 >= 63 && left > 3) left = 3;
            wait_newer(left);
        }
        __syncthreads();
        if (c_issued < n_items) dma_next(buf);
        if (tid < SB * 4) {                                                         // 4 threads x 16 B = one contiguous 64-byte row per stream
            const int srow = tid >> 2, part = tid & 3;
            if (s0 + srow < p.n_streams)
                *reinterpret_cast<float4 *>(demod + (size_t)(s0 + srow) * demod_pitch + 4 * (4 * Qg + part - p.tile_out0)) =
                    *reinterpret_cast<const float4 *>(lout + srow * 16 + 4 * part);
        }
        buf = (buf + 1 == NB) ? 0 : buf + 1;
        if (++m == M) { m = 0; Qg = Qg0; wq = wq0; s0 += gridDim.y * SB; } else { Qg += n_qph; wq += quad_step; }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Octet variant: one workgroup item = 8 CONSECUTIVE tiles (32 audio samples) x 16 streams, fetched as WHOLE 128-byte lines.
// PMC on the quad kernel above (profiles/r1_notes.md): the vector-memory request stream of each CU is the limiter and it asks
// for 1.29 x the useful bytes -- 16.1 line requests per 1600-byte row: the 112-byte overlap between consecutive quads (7 %),
// 16-byte-granular row starts and 1-KiB DMA instructions that straddle a line each.  Here
//   * the item is twice as long in time (3312-byte rows, overlap 3.4 %), every wave owns TWO tile phases (w and w+4; 2 x 27
//     weight fragments stay in registers, which the single-chain scheme of tile_product() made affordable);
//   * every row is fetched from its line-aligned base (window base & ~127): 27 lines = 3 DMA instructions of 8 lines + one of
//     3 lines (exec-masked to 24 lanes), 27 requests per 3200 useful bytes = 1.08 x.  Each instruction has its own M0, so rows
//     sit in LDS at a pitch of 217 x 16 B (odd: the 16 streams of a B-fragment read hit different banks) although the fetches
//     are line granular; the window's offset inside its first line is one scalar add on the LDS read address;
//   * global addresses are saddr + 32-bit lane offsets: the per-item address work is scalar only;
//   * 32 audio samples per stream leave as one whole 128-byte line.
// Requires the input base and pitch to be multiples of 128 bytes (else the quad kernel runs).
#ifndef WFM_OCT_DIAG
#define WFM_OCT_DIAG 0
#endif
struct OctParams {
    int n_streams; long long B2; long long oct_first; int n_octs; long long tile_out0;
    int stride, win_off, n_phases; float scale;
    int swap_xy;                      // grid roles: 0 = x octet phase / y stream-block slot, 1 = the reverse
};
constexpr int OCT_ROW_BYTES = 7 * 400 + 64 * WFM_NK;      // 3312: the kernel is specialised for D*F = 50 (tile stride 400 bytes)
constexpr int OCT_LINES = 27;                             // (112 + 3312 + 127) / 128
constexpr int OCT_PITCH = 217 * 16;                       // LDS row pitch: 27 lines + one 16-byte pad slot
constexpr int OCT_UNIT = 8 * OCT_PITCH;                   // DMA ring unit: 8 streams (half an item), so that 5 units = 2.5 items fit the 160 KB of LDS
constexpr int OCT_OUTP = 36;                              // floats per stream row of the output staging (32 + pad, 16-byte multiple)

template <int NU>
__global__ __launch_bounds__(256) void k_wfm_mfma_oct(const uint8_t *__restrict__ in, size_t in_pitch,
                                                      const v4i *__restrict__ frags, const float *__restrict__ consts, const int *__restrict__ kb_of,
                                                      const float2 *__restrict__ ctab, float *__restrict__ demod, size_t demod_pitch, OctParams p)
{
    extern __shared__ float4 lds_raw[];
    uint8_t *lds_in = reinterpret_cast<uint8_t *>(lds_raw);
    float *lds_out = reinterpret_cast<float *>(lds_in + NU * OCT_UNIT);               // 2 x 16 x OCT_OUTP floats
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, col = lane & 15, q = lane >> 4;
    const int n_oph = p.n_phases / 8;                                                // octet phases
    const int oph = p.swap_xy ? blockIdx.y : blockIdx.x;
    const int slot_y = p.swap_xy ? blockIdx.x : blockIdx.y, n_slots = p.swap_xy ? gridDim.x : gridDim.y;   // stream-block slot of this workgroup
    const long long o_lim = p.oct_first + p.n_octs;
    const long long o0 = p.oct_first + (((long long)oph - p.oct_first) % n_oph + n_oph) % n_oph;
    if (o0 >= o_lim) return;
    const long long m_total = (o_lim - 1 - o0) / n_oph + 1;
    const long long m_per = (m_total + gridDim.z - 1) / gridDim.z;
    const long long m_begin = (long long)blockIdx.z * m_per;
    long long m_end = m_begin + m_per; if (m_end > m_total) m_end = m_total;
    if (m_begin >= m_end) return;
    const int last_stream = p.n_streams - 1;
    const int n_wsb = (p.n_streams + 15) / 16;
    const int my_sb = (n_wsb - slot_y + n_slots - 1) / n_slots;      // stream blocks of this (persistent) workgroup
    if (my_sb <= 0) return;
    const long long M = m_end - m_begin, n_items = (long long)my_sb * M;
    // ---- weights of this wave's two tile phases: once per workgroup lifetime
    v4i A[2][WFM_NFRAG];
    int kb[2]; float k0[2][4], k1[2][4];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int ph = 8 * oph + w + 4 * t;
        kb[t] = __builtin_amdgcn_readfirstlane(kb_of[ph]);
        const v4i *fa = frags + (size_t)ph * WFM_FRAG_V4 + lane;
#pragma unroll
        for (int s = 0; s < WFM_NFRAG; s++) A[t][s] = fa[s * 64];
        const float4 a = *reinterpret_cast<const float4 *>(consts + (size_t)ph * 32 + 4 * q);
        const float4 b = *reinterpret_cast<const float4 *>(consts + (size_t)ph * 32 + 16 + 4 * q);
        k0[t][0] = a.x; k0[t][1] = a.y; k0[t][2] = a.z; k0[t][3] = a.w;
        k1[t][0] = b.x; k1[t][1] = b.y; k1[t][2] = b.z; k1[t][3] = b.w;
    }
    const float K = 0.340447550238101026565118445432744920253753662109375f;
    // ---- input staging by LDS-DMA.  Ring of NU units of 8 streams: unit u = (item u/2, streams 8(u%2) .. +7) lives in slot u % NU;
    //      wave w fetches rows 2w, 2w+1 of every unit, 4 instructions per row (8 VMEM instructions per wave per unit).
    const long long oct_step = (long long)n_oph * 8 * p.stride;                     // bytes between this workgroup's consecutive octets
    const long long Og0 = o0 + m_begin * n_oph;
    const long long wo0 = Og0 * 8 * p.stride + p.win_off - p.B2;                    // first octet's window base relative to the block start
    const uint32_t lds_in_addr = (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t *)lds_in;
    int cs = slot_y, ch = 0, cslot = 0; long long cm = 0, u_issued = 0;         // DMA cursor (stream block, octet, half), its ring slot
    const long long n_units = 2 * n_items;
    uint32_t voff[2][2];                                                            // per-lane byte offset of (half h, row 2w+r, lane's 16-byte piece) from the stream block's base
    auto set_rows = [&](int sb) {
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int srow = min(sb * 16 + 8 * h + 2 * w + r, last_stream) - sb * 16;   // rows past the last stream re-read it (results discarded)
                voff[h][r] = (uint32_t)max(srow, 0) * (uint32_t)in_pitch + 16u * lane;
            }
    };
    set_rows(cs);
    auto dma_unit = [&]() {
        const long long wa = (wo0 + cm * oct_step) & ~127LL;                        // line-aligned fetch base (same for all rows: base and pitch are line multiples)
        const uint8_t *sbase = in + (long long)cs * 16 * (long long)in_pitch + wa;
        const uint32_t ldst = lds_in_addr + cslot * OCT_UNIT + (2 * w) * OCT_PITCH;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const uint32_t vo = ch ? voff[1][r] : voff[0][r];
#pragma unroll
            for (int part = 0; part < 4; part++) {
                const uint8_t *sb = sbase + 1024 * part;
                const uint32_t la = __builtin_amdgcn_readfirstlane((int)(ldst + r * OCT_PITCH + 1024 * part));
                uint32_t keep;
                // nt: the input is read exactly once, by one CU.  Inline asm on purpose (see the quad kernel): hand-counted vmcnt.
                if (part < 3)
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(vo), "s"(sb), "s"(la) : "memory");
                else                                                                // lines 24..26 only: lanes 0..23
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_mov_b64 exec, 0xffffff\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b64 exec, -1\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(vo), "s"(sb), "s"(la) : "memory");
            }
        }
        u_issued++;
        cslot = (cslot + 1 == NU) ? 0 : cslot + 1;
        ch ^= 1;
        if (!ch && ++cm == M) { cm = 0; cs += n_slots; set_rows(cs); }
    };
    // every wave issues exactly 8 VMEM loads per unit and they return in order, so vmcnt(8 n) leaves at most the n newest units in
    // flight (the demod store of waves 0/1 sits in the same queue: it can only make the wait stricter by one instruction)
    auto wait_newer = [&](long long newer) {
        switch ((int)newer) {
            case 0: wait_vmcnt<0>(); break;
            case 1: wait_vmcnt<8>(); break;
            case 2: wait_vmcnt<16>(); break;
            default: wait_vmcnt<24>(); break;
        }
    };
    for (int k = 0; k < NU; k++) if (u_issued < n_units) dma_unit();
    { long long newer = u_issued - 2; if (newer < 0) newer = 0; if (newer > 3) newer = 3; wait_newer(newer); }     // item 0 = units 0, 1
    __syncthreads();
    int ua = 0;                                                                     // slot of the current item's first unit
    int s0 = slot_y * 16; long long m = 0, Og = Og0, wo = wo0;                  // compute position
    for (long long it = 0; it < n_items; it++) {
        const int wbm = (int)(wo & 127);                                            // the window's offset inside its first fetched line
        float *lout = lds_out + (int)(it & 1) * (16 * OCT_OUTP);
        const int ub = (ua + 1 == NU) ? 0 : ua + 1;
        const uint8_t *lrow = lds_in + (col < 8 ? ua : ub) * OCT_UNIT + (col & 7) * OCT_PITCH + 16 * q;     // this lane's stream row in the ring
#if WFM_OCT_DIAG == 1                                                                 // experiment: DMA stream only (no LDS reads, no math)
        if (lane == 0) { lout[w] = (float)it; }
#else
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int jt = w + 4 * t;                                               // tile inside the octet
            const long long wb2 = wo + p.B2 + (long long)jt * p.stride;             // global byte index of the tile's window base
            const long long chunk_rel = (wb2 >> 11) - (p.B2 >> 11);
            const float2 C0 = ctab[chunk_rel + 1], C1 = ctab[chunk_rel + 2];
            const uint8_t *src = lrow + wbm + jt * p.stride;
            v4i Bf[WFM_NK];
#pragma unroll
            for (int ks = 0; ks < WFM_NK; ks++) Bf[ks] = *reinterpret_cast<const v4i *>(src + 64 * ks) ^ (int)0x80808080;
            v4i acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}, snap[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
            tile_product(A[t], Bf, kb[t], acc, snap);
            float pI, pQ, cI, cQ;
            tile_rows(acc, snap, kb[t] < WFM_NK, p.scale, k0[t], k1[t], C0, C1, pI, pQ, cI, cQ);
            const float dq = cQ - pQ, di = cI - pI;
            const float num = cI * dq - cQ * di, den = cI * cI + cQ * cQ;
            float rd = __builtin_amdgcn_rcpf(den);                                  // Newton-refined reciprocal (~1 ulp), as in the quad kernel
            rd = fmaf(fmaf(-den, rd, 1.0f), rd, rd);
            lout[col * OCT_OUTP + 4 * jt + q] = (den != 0.f) ? (K * num) * rd : 0.f;  // audio 4*(8 Og + jt) + q of stream col
        }
#endif
        {   // item it+1 = units 2it+2, 2it+3 must have landed before anyone passes the barrier; units issued beyond them stay in flight
            long long left = u_issued - (2 * it + 4);
            if (left < 0) left = 0;
            if (left > 3) left = 3;
            wait_newer(left);
        }
        __syncthreads();
#if WFM_OCT_DIAG != 2                                                                 // experiment 2: math only (the ring is filled once)
        if (u_issued < n_units) dma_unit();                                         // this item's two slots are free again
        if (u_issued < n_units) dma_unit();
#endif
        if (tid < 128) {                                                            // 8 threads x 16 B = one whole 128-byte line per stream
            const int srow = tid >> 3, part = tid & 7;
            if (s0 + srow < p.n_streams)
                *reinterpret_cast<float4 *>(demod + (size_t)(s0 + srow) * demod_pitch + 4 * (8 * Og - p.tile_out0) + 4 * part) =
                    *reinterpret_cast<const float4 *>(lout + srow * OCT_OUTP + 4 * part);
        }
        ua += 2; if (ua >= NU) ua -= NU;
        if (++m == M) { m = 0; Og = Og0; wo = wo0; s0 += n_slots * 16; } else { Og += n_oph; wo += oct_step; }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Sequential variant (default): ONE phase-independent weight set, a workgroup walks through time.
// shift_addition_cc's phasor inside a 1024-chunk is R[n] = C_m D^(n - 1024 m), so relative to a tile's window base n0
// R[n0 + t] = [C_m D^(n0 - 1024 m)] D^t: the weights a h D^t are the same for every tile and the bracket is a complex scalar per (tile, chunk)
// applied after the product (the same idea as ddc_mfma.hip).  Without phase-specific weights a wave no longer has to own a tile phase and
// jump through the input in steps of 128 tiles: a workgroup owns 16 streams x a contiguous run of tiles, its 8 waves take 8 consecutive tiles
// per step, and the input slides through a per-stream LDS ring filled by LDS-DMA in whole 1-KiB runs of lines -- every input byte is fetched
// exactly once (quad kernel: 1.29 x, octet kernel: 1.08 x requested / useful bytes), in long sequential runs per stream.
// A chunk boundary inside the 256-sample window is 16-byte granular here (window bases are multiples of 8 samples): the boundary K-step is
// multiplied twice with complementary lane groups of the B operand zeroed, the accumulator chain is snapshotted in between.
struct SeqParams {
    int n_streams; long long B2; long long tile_first; int n_tiles, tiles_per_seg; long long tile_out0;
    int stride, win_off; float scale;
    // fused back end (FUSE): de-emphasis + convert_f_s16 inside the kernel
    float alpha; const float *last_in; float *seg_state; int16_t *s16; float *af; size_t out_pitch; long long j_first; int skip;
};

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

constexpr int SEQ_RB = 9216;               // ring bytes per stream: 9 x 1 KiB (the 8-tile step's window 3312 B + the next step's 3200 B + fetch granularity)
#ifndef SEQ_RING_PAD
#define SEQ_RING_PAD 32
#endif
constexpr int SEQ_RP = SEQ_RB + SEQ_RING_PAD;   // LDS pitch.  A ds_read_b128 is served in four groups of 16 lanes ({0-3, 12-15, 20-27}, ...): with the 16-byte slot of lane
                                           // (stream, q) = (2 stream + q) mod 16 every group touches 16 different slots; + 16 (slot = stream + q) left 4 two-way conflicts per read
constexpr int SEQ_NGR = 4 * WFM_NK;        // 16-byte granules per window

// FUSE: the back end of the chain inside this kernel.  The workgroup produces its streams' audio in time order, so the one-pole de-emphasis
// (libcsdr.c:1081-1097) is a state carried in 16 lanes of wave 0 from step to step, and convert_f_s16 + the stores are done by all threads one
// step later (one sample per thread, three staging buffers so that nobody waits): the demodulated audio never goes to HBM as floats
// (0.2 GB written + read per step), and k_wfm_back (0.077 ms of a 1.08 ms step) disappears.  A segment that starts in the middle of a stream
// demodulates two steps (64 audio samples) ahead of its range from zero state without storing them -- the filter forgets as 0.706^k --,
// the first segment starts from the exact carried state and first walks through the leading edge tiles' samples (computed by the per-wave
// edge kernel, launched before); the last segment exports its state for the trailing edge tiles (k_wfm_tail).
template <int NT, bool FUSE>
__global__ __launch_bounds__(256 * NT) void k_wfm_mfma_seq(const uint8_t *__restrict__ in, size_t in_pitch, const v4i *__restrict__ frags, const float *__restrict__ cum,
                                                           const float2 *__restrict__ dtab, const float2 *__restrict__ ctab, float *__restrict__ demod, size_t demod_pitch, SeqParams p)
{
    constexpr int TPG = 4 * NT, SPW = 16 / (4 * NT);                                 // tiles per step; streams fetched per wave in a row-step
    extern __shared__ float4 lds_raw[];
    uint8_t *lds_in = reinterpret_cast<uint8_t *>(lds_raw);
    constexpr int NOUT = FUSE ? 3 : 2;
    float *lds_out = reinterpret_cast<float *>(lds_in + 16 * SEQ_RP);                // NOUT x 16 x OCT_OUTP floats
    float *lcum = lds_out + NOUT * 16 * OCT_OUTP;                                    // prefix table (a global vector load inside the loop would drain the DMA ring)
    const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, col = lane & 15, q = lane >> 4;
    for (int i = tid; i < (SEQ_NGR + 1) * 16; i += 256 * NT) lcum[i] = cum[i];
    const int sb = blockIdx.x;
    const long long t0 = p.tile_first + (long long)blockIdx.y * p.tiles_per_seg;
    long long t1 = t0 + p.tiles_per_seg; if (t1 > p.tile_first + p.n_tiles) t1 = p.tile_first + p.n_tiles;
    if (t0 >= t1) return;
    const int n_it = (int)(t1 - t0), n_grp = (n_it + TPG - 1) / TPG;
    const int n_warm = (FUSE && blockIdx.y > 0) ? 2 : 0;                             // steps demodulated ahead of the segment to warm the de-emphasis up
    const int last_stream = p.n_streams - 1;
    v4i A[WFM_NK * 3];
#pragma unroll
    for (int s = 0; s < WFM_NK * 3; s++) A[s] = frags[s * 64 + lane];
    float c_lo[4], c_hi[4];                                                          // prefix values at the window's ends, rows 4q .. 4q+3
    {
        const float4 a = *reinterpret_cast<const float4 *>(cum + 4 * q), b = *reinterpret_cast<const float4 *>(cum + (size_t)SEQ_NGR * 16 + 4 * q);
        c_lo[0] = a.x; c_lo[1] = a.y; c_lo[2] = a.z; c_lo[3] = a.w; c_hi[0] = b.x; c_hi[1] = b.y; c_hi[2] = b.z; c_hi[3] = b.w;
    }
    const float K = 0.340447550238101026565118445432744920253753662109375f;
    // ---- DMA ring
    const int tstride = p.stride;
    long long wg = (t0 - (long long)n_warm * TPG) * tstride + p.win_off - p.B2;      // window start of the step's first tile, bytes from the block start
    const long long F0 = wg & ~1023LL;
    const long long F_end = ((t1 - 1) * tstride + p.win_off - p.B2 + 64 * WFM_NK + 1023) & ~1023LL;
    long long F = F0;
    int fslot = (int)(F0 % SEQ_RB), wslot = (int)(wg % SEQ_RB);                      // ring positions of F and of wg (wg already includes the warm-up steps)
    const uint32_t lds_in_addr = (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t *)lds_in;
    uint32_t voff[SPW];
#pragma unroll
    for (int r = 0; r < SPW; r++) {
        const int srow = min(sb * 16 + SPW * wv + r, last_stream) - sb * 16;
        voff[r] = (uint32_t)max(srow, 0) * (uint32_t)in_pitch + 16u * lane;
    }
    const uint8_t *sblock = in + (long long)sb * 16 * (long long)in_pitch;
    auto row_step = [&]() {
        const uint8_t *sbase = sblock + F;
        const uint32_t ldst = lds_in_addr + (SPW * wv) * SEQ_RP + (uint32_t)fslot;
#pragma unroll
        for (int r = 0; r < SPW; r++) {
            const uint32_t la = __builtin_amdgcn_readfirstlane((int)(ldst + r * SEQ_RP));
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff[r]), "s"(sbase), "s"(la) : "memory");
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
    while (F < F_end && F + 1024 <= wg + SEQ_RB) row_step();
    wait_for(wg + (long long)(TPG - 1) * tstride);
    __syncthreads();
    const uint8_t *lrow = lds_in + col * SEQ_RP;
    int s0 = sb * 16;
    // ---- fused back end: de-emphasis state of stream s0 + lane (wave 0, lanes 0..15)
    // The recurrence y_k = alpha x_k + b y_{k-1} (b = 1 - alpha) over a step's 32 samples of a stream is split over 4 lanes (wave 0: lane =
    // stream + 16 * quarter): each lane runs its 8 samples from zero state, the quarters' end values are combined with b^8, b^16, b^24 (three
    // shuffles), and every output gets its share b^(j+1) of its quarter's start state -- a dependent chain of ~12 instead of 32 steps on the
    // one wave that all others wait for at the next barrier.  (Rounding differs from the sequential recurrence by a few 1e-8.)
    float yst = 0.f;                                                                 // state after the last sample processed so far (same value in the stream's 4 lanes)
    const float one_minus = 1 - p.alpha;
    const bool iir_lane = FUSE && wv == 0 && lane < 16;
    const bool iir_wave = FUSE && wv == 0;
    constexpr int QL = TPG, SPS = 4 * TPG;                                           // samples per lane of the scan (a quarter of a step), samples per step and stream
    float bp[QL + 1];                                                                // b^0 .. b^QL
    bp[0] = 1.f;
#pragma unroll
    for (int j = 1; j <= QL; j++) bp[j] = bp[j - 1] * one_minus;
    const float b2q = bp[QL] * bp[QL], b3q = b2q * bp[QL];
    if (FUSE && iir_lane && blockIdx.y == 0 && s0 + lane < p.n_streams) {
        // first segment: exact carried state (NaN reset as libcsdr.c:1092), then the leading edge tiles' samples in order
        yst = p.last_in[s0 + lane]; if (yst != yst) yst = 0.f;
        const int n_lead = (int)(4 * p.tile_first - p.j_first);
        const float *row = demod + (size_t)(s0 + lane) * demod_pitch + p.skip;
        for (int i = 0; i < n_lead; i++) {
            yst = p.alpha * row[i] + one_minus * yst;
            const float scaled = yst * 32767.0f;
            const int iv = (scaled >= -2147483648.0f && scaled < 2147483648.0f) ? (int)scaled : (int)0x80000000;
            p.s16[(size_t)(s0 + lane) * p.out_pitch + i] = (int16_t)iv;
            if (p.af) p.af[(size_t)(s0 + lane) * p.out_pitch + i] = yst;
        }
    }
    if (iir_wave) yst = __shfl(yst, col, 64);                                        // the stream's state into all four of its lanes
    auto emit = [&](int g) {                                                         // convert_f_s16 + store of step g's 16 x 32 samples, one per thread
        if (g < 0) return;
        const int vt = min(TPG, n_it - g * TPG);
        const int srow = tid / SPS, k = tid % SPS;
        if (tid < 16 * SPS && (k >> 2) < vt && s0 + srow < p.n_streams) {
            const float e = lds_out[(g % 3) * (16 * OCT_OUTP) + srow * OCT_OUTP + k];
            const long long idx = 4 * (t0 + (long long)g * TPG) + k - p.j_first;
            const float scaled = e * 32767.0f;                                       // convert_f_s16 libcsdr.c:2397 (x86 truncation semantics)
            const int iv = (scaled >= -2147483648.0f && scaled < 2147483648.0f) ? (int)scaled : (int)0x80000000;
            p.s16[(size_t)(s0 + srow) * p.out_pitch + idx] = (int16_t)iv;
            if (p.af) p.af[(size_t)(s0 + srow) * p.out_pitch + idx] = e;
        }
    };
    for (int gi = -n_warm; gi < n_grp; gi++) {
        const int it = gi * TPG + wv;
        float *lout = lds_out + (FUSE ? ((gi + 3) % 3) : (gi & 1)) * (16 * OCT_OUTP);
        if (it < n_it) {
            const long long ws = wg + (long long)wv * tstride;
            const long long n0 = (ws + p.B2) >> 1;                                   // global index of the window's first sample
            const int off = (int)(n0 & 1023);
            const bool two = off + 32 * WFM_NK > 1024;
            const int bo = 2 * (1024 - off);                                         // byte offset of the next chunk's first sample inside the window
            const int kb = __builtin_amdgcn_readfirstlane(two ? (bo >> 6) : WFM_NK), hq = __builtin_amdgcn_readfirstlane((bo >> 4) & 3);
            const long long chunk_rel = (n0 >> 10) - (p.B2 >> 11);
            // ---- B fragments from the ring
            unsigned a0 = (unsigned)(wslot + wv * tstride + 16 * q);
            a0 = min(a0, a0 - (unsigned)SEQ_RB);                                     // wrap (at most once: wslot < RB, the rest < RB)
            v4i Bf[WFM_NK];
#pragma unroll
            for (int ks = 0; ks < WFM_NK; ks++) {
                unsigned a = a0 + 64u * ks; a = min(a, a - (unsigned)SEQ_RB);
                Bf[ks] = *reinterpret_cast<const v4i *>(lrow + a) ^ (int)0x80808080;
            }
            v4i acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}, snap[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
            const bool lo_lane = q < hq;
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
            const float2 C0 = ctab[chunk_rel + 1], D0 = dtab[off + 2048];
            const float2 P0 = make_float2(C0.x * D0.x - C0.y * D0.y, C0.x * D0.y + C0.y * D0.x);
            float2 P1 = make_float2(0.f, 0.f);
            float k0[4], k1[4];
            if (two) {
                const float2 C1 = ctab[chunk_rel + 2], D1 = dtab[off - 1024 + 2048];
                P1 = make_float2(C1.x * D1.x - C1.y * D1.y, C1.x * D1.y + C1.y * D1.x);
                const float4 cb = *reinterpret_cast<const float4 *>(lcum + (bo >> 4) * 16 + 4 * q);
                k0[0] = cb.x - c_lo[0]; k0[1] = cb.y - c_lo[1]; k0[2] = cb.z - c_lo[2]; k0[3] = cb.w - c_lo[3];
                k1[0] = c_hi[0] - cb.x; k1[1] = c_hi[1] - cb.y; k1[2] = c_hi[2] - cb.z; k1[3] = c_hi[3] - cb.w;
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) { k0[r] = c_hi[r] - c_lo[r]; k1[r] = 0.f; }
            }
            float pI, pQ, cI, cQ;
            tile_rows(acc, snap, two, p.scale, k0, k1, P0, P1, pI, pQ, cI, cQ);
            const float dq = cQ - pQ, di = cI - pI;
            const float num = cI * dq - cQ * di, den = cI * cI + cQ * cQ;
            float rd = __builtin_amdgcn_rcpf(den);
            rd = fmaf(fmaf(-den, rd, 1.0f), rd, rd);
            lout[col * OCT_OUTP + 4 * wv + q] = (den != 0.f) ? (K * num) * rd : 0.f;   // audio 4 * tile + q of stream col
        }
        const long long wg_n = wg + (long long)TPG * tstride;
        if (gi + 1 < n_grp) wait_for(wg_n + (long long)(TPG - 1) * tstride);
        __syncthreads();
        if (gi + 1 < n_grp) { while (F < F_end && F + 1024 <= wg_n + SEQ_RB) row_step(); }
        if (FUSE) {
            emit(gi - 1);                                                            // the previous step's audio: filtered by wave 0 before it came to this barrier
            if (iir_wave) {                                                          // this step's 32 samples of stream s0 + col through the de-emphasis, in place
                float *row = lout + col * OCT_OUTP + QL * q;                         // this lane's quarter: samples QL q .. QL q + QL - 1
                const int nv = 4 * min(TPG, n_it - gi * TPG);                        // valid samples of the step (a multiple of 4; all except in a segment's last step)
                float x[QL], z[QL], y[QL];
#pragma unroll
                for (int j = 0; j < QL; j += 4) {
                    const float4 v = *reinterpret_cast<const float4 *>(row + j);
                    const bool ok = QL * q + j < nv;                                 // samples behind the valid ones do not exist: feed zeros (their outputs are not stored)
                    x[j] = ok ? v.x : 0.f; x[j + 1] = ok ? v.y : 0.f; x[j + 2] = ok ? v.z : 0.f; x[j + 3] = ok ? v.w : 0.f;
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
                for (int j = 0; j < QL; j += 4) *reinterpret_cast<float4 *>(row + j) = make_float4(y[j], y[j + 1], y[j + 2], y[j + 3]);
                // new carried state = the value after the last VALID sample (nv - 1): it lives in quarter (nv - 1) / QL at position (nv - 1) % QL
                const int lq = (nv - 1) / QL, lj = (nv - 1) % QL;
                float ylast = y[QL - 1];
                if (lj == 3) ylast = y[3];
                yst = __shfl(ylast, col + 16 * lq, 64);
            }
        } else {   // TPG tiles x 4 audio samples per stream leave as one run of 16-byte pieces (a whole 128-byte line for 8 tiles)
            const int vt = min(TPG, n_it - gi * TPG);
            if (tid < 16 * TPG) {
                const int srow = tid / TPG, part = tid % TPG;
                if (part < vt && s0 + srow < p.n_streams)
                    *reinterpret_cast<float4 *>(demod + (size_t)(s0 + srow) * demod_pitch + 4 * (t0 + (long long)gi * TPG - p.tile_out0) + 4 * part) =
                        *reinterpret_cast<const float4 *>(lout + srow * OCT_OUTP + 4 * part);
            }
        }
        wg = wg_n; wslot += TPG * tstride; if (wslot >= SEQ_RB) wslot -= SEQ_RB;
    }
    if (FUSE) {
        __syncthreads();
        emit(n_grp - 1);
        if (iir_lane && blockIdx.y + 1 == gridDim.y && s0 + lane < p.n_streams) p.seg_state[s0 + lane] = yst;   // for the trailing edge tiles (lanes 0..15: quarter 0 holds the state too)
    }
}

// trailing samples of a call (the edge tiles behind the sequential kernel's range): one lane per stream continues the de-emphasis
__global__ __launch_bounds__(64) void k_wfm_tail(const float *__restrict__ demod, size_t demod_pitch, int skip, int first, int n_audio, float alpha,
                                                 const float *__restrict__ state_in, float *__restrict__ last_out, int16_t *__restrict__ s16, float *__restrict__ af,
                                                 size_t out_pitch, int n_streams)
{
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n_streams) return;
    float y = state_in[s];
    const float one_minus = 1 - alpha;
    const float *row = demod + (size_t)s * demod_pitch + skip;
    for (int i = first; i < n_audio; i++) {
        y = alpha * row[i] + one_minus * y;
        const float scaled = y * 32767.0f;
        const int iv = (scaled >= -2147483648.0f && scaled < 2147483648.0f) ? (int)scaled : (int)0x80000000;
        s16[(size_t)s * out_pitch + i] = (int16_t)iv;
        if (af) af[(size_t)s * out_pitch + i] = y;
    }
    last_out[s] = y;
}

} // namespace

namespace csdr_amd {

static const char *g_last_kernel = "k_wfm_mfma";
static int g_wfm_select = -1;      // test hook csdr_amd_debug_wfm_select: -1 = environment / default, 0 = sequential, 1 = octet, 2 = quad, 3 = per-wave kernel
const char *wfm_mfma_last_kernel() { return g_last_kernel; }

int wfm_mfma_launch(hipStream_t st, hipStream_t st_edge, hipEvent_t ev_begin, hipEvent_t ev_end, const uint8_t *in, size_t in_pitch, const uint8_t *hist, const WfmMfmaDevice &dev, const float2 *ctab,
                    float *demod, size_t demod_pitch, int n_streams, int T, long long B, long long j_first, int n_audio, WfmBackArgs *back)
{
    MfmaParams p;
    if (back) back->done = false;
    p.n_streams = n_streams; p.T = T; p.B = B; p.j_first = j_first; p.n_audio = n_audio;
    p.tile_stride_bytes = dev.tile_stride_bytes; p.win_off_bytes = dev.win_off_bytes; p.n_phases = dev.n_phases; p.scale = dev.scale;
    const long long tile_first = j_first / 4, tile_last = (j_first + n_audio - 1) / 4;
    p.tile_out0 = 8 * (tile_first / 8);                              // the scratch rows hold whole octets (128-byte lines); k_wfm_back skips j_first % 32 samples
    // interior tiles: window entirely inside [0, 2T) of this block (no history, no ragged end)
    long long t_a = tile_first, t_b = tile_last;
    while (t_a <= tile_last && t_a * p.tile_stride_bytes + p.win_off_bytes - 2 * B < 0) t_a++;
    while (t_b >= t_a && t_b * p.tile_stride_bytes + p.win_off_bytes - 2 * B + 64 * WFM_NK > 2LL * T) t_b--;
    const int n_sb = (n_streams + 63) / 64;
    static int target = 0;
    if (!target) { const char *e = getenv("CSDR_AMD_WFM_WAVES"); target = e ? atoi(e) : 2048; if (target < 1) target = 2048; }
    auto launch = [&](hipStream_t ls, long long first, long long last, bool edge) -> int {
        if (last < first) return 0;
        p.tile_first = first; p.n_tiles = (int)(last - first + 1); p.tiles_per_wave = 0; p.two_ranges = 0; p.tile_first_b = 0; p.n_tiles_b = 0;
        // waves = stream blocks x phases x segments: aim at `target` waves, keep >= 4 tiles per wave
        const long long per_phase = (p.n_tiles + p.n_phases - 1) / p.n_phases;
        int z = (int)((target + (long long)n_sb * p.n_phases - 1) / ((long long)n_sb * p.n_phases));
        if (z > per_phase / 4) z = (int)(per_phase / 4);
        if (z < 1) z = 1;
        dim3 grid(n_sb, p.n_phases, z);
        if (edge) hipLaunchKernelGGL((k_wfm_mfma<true>), grid, dim3(64), 0, ls, in, in_pitch, hist, (const v4i *)dev.d_frags, dev.d_consts, dev.d_kb_of, ctab, demod, demod_pitch, p);
        else      hipLaunchKernelGGL((k_wfm_mfma<false>), grid, dim3(64), 0, ls, in, in_pitch, hist, (const v4i *)dev.d_frags, dev.d_consts, dev.d_kb_of, ctab, demod, demod_pitch, p);
        CSDR_LAUNCH_CHECK();
        return 0;
    };
    static int env_wg = -1, env_seq = -1, env_oct = -1;
    if (env_wg < 0) {
        const char *e = getenv("CSDR_AMD_WFM_WG"); env_wg = e ? atoi(e) : 1;
        e = getenv("CSDR_AMD_WFM_SEQ"); env_seq = e ? atoi(e) : 1;
        e = getenv("CSDR_AMD_WFM_OCT"); env_oct = e ? atoi(e) : 1;
    }
    const int use_wg = g_wfm_select >= 0 ? g_wfm_select <= 2 : env_wg;
    const int use_seq = g_wfm_select >= 0 ? g_wfm_select == 0 : env_seq;
    const int use_oct = g_wfm_select >= 0 ? g_wfm_select <= 1 : env_oct;
    int rc = 0;
    const int n_cu = current_device_cu_count();
    // leftovers around the workgroup kernels' range run on the per-wave kernel (bounds-checked variant: a handful of tiles), both ends in ONE launch
    auto launch_edges = [&](long long a0, long long a1, long long b0, long long b1) -> int {
        if (a1 >= a0 && b1 >= b0) {                                          // (each is latency bound: ~22 us)
            p.tile_first = a0; p.n_tiles = (int)(a1 - a0 + 1); p.tile_first_b = b0; p.n_tiles_b = (int)(b1 - b0 + 1); p.two_ranges = 1; p.tiles_per_wave = 0;
            hipLaunchKernelGGL((k_wfm_mfma<true>), dim3(n_sb, p.n_phases, 2), dim3(64), 0, st_edge, in, in_pitch, hist, (const v4i *)dev.d_frags, dev.d_consts, dev.d_kb_of, ctab, demod, demod_pitch, p);
            CSDR_LAUNCH_CHECK();
            return 0;
        }
        int r2 = launch(st_edge, a0, a1, true); if (r2) return r2;
        return launch(st_edge, b0, b1, true);
    };
    // ---- sequential kernel (default): one weight set, a workgroup walks its 16 streams through time, every byte fetched once
    if (use_seq && use_wg && dev.d_seq_frags && p.tile_stride_bytes == 400 && (((uintptr_t)in | in_pitch) & 127) == 0 && in_pitch * 16 + 4096 < ((size_t)1 << 32)) {
        long long sa = (t_a + 7) / 8 * 8, sb_ = t_b;                                     // first tile: multiple of 8, so that every step writes one whole output line
        while (sb_ >= sa && ((sb_ * p.tile_stride_bytes + p.win_off_bytes - 2 * B + 64 * WFM_NK + 1023) & ~1023LL) > 2LL * T) sb_--;   // 1-KiB fetch granularity
        if (sb_ - sa + 1 >= 64) {
            SeqParams sp;
            sp.n_streams = n_streams; sp.B2 = 2 * B; sp.tile_first = sa; sp.n_tiles = (int)(sb_ - sa + 1); sp.tile_out0 = p.tile_out0;
            sp.stride = p.tile_stride_bytes; sp.win_off = p.win_off_bytes; sp.scale = dev.seq_scale;
            const int n_wsb = (n_streams + 15) / 16;
            int n_seg = (n_cu + n_wsb - 1) / n_wsb; if (n_seg < 1) n_seg = 1;
            if (n_seg > sp.n_tiles / 64) n_seg = sp.n_tiles / 64;
            if (n_seg < 1) n_seg = 1;
            sp.tiles_per_seg = ((sp.n_tiles + n_seg - 1) / n_seg + 7) / 8 * 8;
            n_seg = (sp.n_tiles + sp.tiles_per_seg - 1) / sp.tiles_per_seg;
            static int env_fuse = -1;
            if (env_fuse < 0) { const char *e = getenv("CSDR_AMD_WFM_FUSE_BACK"); env_fuse = e ? atoi(e) : 1; }
            const bool fuse = back && env_fuse && sp.tiles_per_seg >= 64 && st_edge == st;       // (edge tiles on a side stream would not be ordered before the first segment)
            sp.alpha = 0; sp.last_in = nullptr; sp.seg_state = nullptr; sp.s16 = nullptr; sp.af = nullptr; sp.out_pitch = 0; sp.j_first = j_first; sp.skip = 0;
            if (fuse) { sp.alpha = back->alpha; sp.last_in = back->last_in; sp.seg_state = back->seg_state; sp.s16 = back->s16; sp.af = back->af; sp.out_pitch = back->out_pitch; sp.skip = back->skip; }
            const size_t lds = (size_t)16 * SEQ_RP + (fuse ? 3 : 2) * 16 * OCT_OUTP * sizeof(float) + (SEQ_NGR + 1) * 16 * sizeof(float);
            static int env_nt = 0;                                                       // CSDR_AMD_WFM_SEQ_NT: 2 = 8 waves take 8 tiles per step (default), 1 = 4 waves x 4 tiles (deeper look-ahead in the same ring)
            if (!env_nt) { const char *e = getenv("CSDR_AMD_WFM_SEQ_NT"); env_nt = (e && atoi(e) == 1) ? 1 : 2; }
            const int nt = env_nt;
            {
                const void *fn = nt == 2 ? (fuse ? (const void *)k_wfm_mfma_seq<2, true> : (const void *)k_wfm_mfma_seq<2, false>)
                                         : (fuse ? (const void *)k_wfm_mfma_seq<1, true> : (const void *)k_wfm_mfma_seq<1, false>);
                const int arc = lds_attr_once(fn, lds); if (arc) return arc;
            }
            // the edge tiles first: the fused back end of the first segment walks through the leading ones' samples
            rc = launch_edges(tile_first, sa - 1, sb_ + 1, tile_last); if (rc) return rc;
            if (ev_begin) CSDR_HIP(hipEventRecord(ev_begin, st));
#define SEQ_LAUNCH(NTV, FV) hipLaunchKernelGGL((k_wfm_mfma_seq<NTV, FV>), dim3(n_wsb, n_seg), dim3(256 * NTV), lds, st, in, in_pitch, (const v4i *)dev.d_seq_frags, dev.d_seq_cum, dev.d_dtab, ctab, demod, demod_pitch, sp)
            if (nt == 2) { if (fuse) SEQ_LAUNCH(2, true); else SEQ_LAUNCH(2, false); }
            else         { if (fuse) SEQ_LAUNCH(1, true); else SEQ_LAUNCH(1, false); }
#undef SEQ_LAUNCH
            CSDR_LAUNCH_CHECK();
            if (ev_end) CSDR_HIP(hipEventRecord(ev_end, st));
            if (fuse) {
                hipLaunchKernelGGL(k_wfm_tail, dim3((n_streams + 63) / 64), dim3(64), 0, st, demod, demod_pitch, back->skip, (int)(4 * (sb_ + 1) - j_first), n_audio, back->alpha,
                                   back->seg_state, back->last_out, back->s16, back->af, back->out_pitch, n_streams);
                CSDR_LAUNCH_CHECK();
                back->done = true;
            }
            g_last_kernel = "k_wfm_mfma_seq";
            return 0;
        }
    }
    // ---- octet kernel: 8 consecutive tiles x 16 streams per item, line-aligned fetch
    if (use_oct && use_wg && (p.n_phases % 8) == 0 && 7 * p.tile_stride_bytes + 64 * WFM_NK == OCT_ROW_BYTES &&
        (((uintptr_t)in | in_pitch) & 127) == 0 && in_pitch * 16 + 4096 < ((size_t)1 << 32)) {
        const long long oa = (t_a + 7) / 8; long long ob = (t_b + 1) / 8 - 1;            // whole octets inside the interior tile range
        auto fetch_end = [&](long long o) { return ((o * 8 * p.tile_stride_bytes + p.win_off_bytes - 2 * B) & ~127LL) + 128LL * OCT_LINES; };
        while (ob >= oa && fetch_end(ob) > 2LL * T) ob--;                                // the line-aligned fetch must stay inside the row
        const int n_oph = p.n_phases / 8;
        if (ob - oa + 1 >= 2 * n_oph) {
            OctParams op;
            op.n_streams = n_streams; op.B2 = 2 * B; op.oct_first = oa; op.n_octs = (int)(ob - oa + 1); op.tile_out0 = p.tile_out0;
            op.stride = p.tile_stride_bytes; op.win_off = p.win_off_bytes; op.n_phases = p.n_phases; op.scale = p.scale;
            static int swap = -1;
            if (swap < 0) { const char *e = getenv("CSDR_AMD_WFM_OCT_SWAP"); swap = e ? atoi(e) != 0 : 1; }
            op.swap_xy = swap;
            const int n_wsb = (n_streams + 15) / 16;
            const long long per_oph = (op.n_octs + n_oph - 1) / n_oph;
            int gy = n_cu / n_oph; if (gy < 1) gy = 1; if (gy > n_wsb) gy = n_wsb;      // persistent: one workgroup per CU
            int z = n_cu / (n_oph * gy); if (z < 1) z = 1;
            if (z > per_oph / 8) z = (int)(per_oph / 8);
            if (z < 1) z = 1;
            static int nu = 0;                                                           // ring depth in half-item units (experiment switch; 5 = 2.5 items)
            if (!nu) { const char *e = getenv("CSDR_AMD_WFM_OCT_UNITS"); nu = e ? atoi(e) : 5; if (nu != 4) nu = 5; }
            const size_t lds = (size_t)nu * OCT_UNIT + 2 * 16 * OCT_OUTP * sizeof(float);
            static bool done[2] = {false, false};
            auto go = [&](auto kern, bool &dn) -> int {
                (void)dn; { const int arc = lds_attr_once((const void *)kern, lds); if (arc) return arc; }
                if (ev_begin) CSDR_HIP(hipEventRecord(ev_begin, st));
                hipLaunchKernelGGL(kern, swap ? dim3(gy, n_oph, z) : dim3(n_oph, gy, z), dim3(256), lds, st, in, in_pitch, (const v4i *)dev.d_frags, dev.d_consts, dev.d_kb_of, ctab, demod, demod_pitch, op);
                return 0;
            };
            const int grc = nu == 4 ? go(k_wfm_mfma_oct<4>, done[0]) : go(k_wfm_mfma_oct<5>, done[1]);
            if (grc) return grc;
            CSDR_LAUNCH_CHECK();
            if (ev_end) CSDR_HIP(hipEventRecord(ev_end, st));
            g_last_kernel = "k_wfm_mfma_oct";
            return launch_edges(tile_first, 8 * oa - 1, 8 * (ob + 1), tile_last);
        }
    }
    const long long qa = (t_a + 3) / 4, qb = (t_b + 1) / 4 - 1;             // whole quads inside the interior tile range
    if (use_wg && (p.n_phases % 4) == 0 && 3 * p.tile_stride_bytes + 64 * WFM_NK == 1712 && qb - qa + 1 >= 2 * (p.n_phases / 4)) {
        WgParams wp;
        wp.n_streams = n_streams; wp.B2 = 2 * B; wp.quad_first = qa; wp.n_quads = (int)(qb - qa + 1); wp.tile_out0 = p.tile_out0;
        wp.stride = p.tile_stride_bytes; wp.win_off = p.win_off_bytes; wp.n_phases = p.n_phases; wp.scale = p.scale;
        wp.row_bytes = 3 * wp.stride + 64 * WFM_NK;                       // rows back to back (the DMA fills one contiguous run); 107 slots: odd
        // CSDR_AMD_WFM_WGCFG: 0 = 16 streams x ring of 5 quads, one workgroup per CU (default); 1 = 32 streams x ring of 2;
        // 2 = 16 streams x ring of 2, TWO workgroups per CU (two waves per SIMD); 3 = 16 streams x ring of 3, one per CU
        static int cfg = -1;
        if (cfg < 0) { const char *e = getenv("CSDR_AMD_WFM_WGCFG"); cfg = e ? atoi(e) : 0; if (cfg < 0 || cfg > 3) cfg = 0; }
        const int SBv = cfg == 1 ? 32 : 16, NBv = cfg == 0 ? 5 : (cfg == 3 ? 3 : 2), per_cu = cfg == 2 ? 2 : 1;
        const int n_qph = p.n_phases / 4, n_wsb = (n_streams + SBv - 1) / SBv;
        const long long per_qph = (wp.n_quads + n_qph - 1) / n_qph;
        // persistent grid: `per_cu` workgroups per CU; a workgroup owns a quad phase and walks its share of the stream blocks
        const int slots = n_cu * per_cu;
        int gy = slots / n_qph; if (gy < 1) gy = 1; if (gy > n_wsb) gy = n_wsb;
        int z = slots / (n_qph * gy); if (z < 1) z = 1;
        if (z > per_qph / 8) z = (int)(per_qph / 8);
        if (z < 1) z = 1;
        const size_t lds = (size_t)NBv * (4 * ((SBv * 107 + 255) / 256) * 1024) + 2 * SBv * 16 * sizeof(float);
        if (wp.row_bytes != 1712) return fail_msg(-3, "wfm: workgroup kernel is specialised for 1712-byte quad rows");
        if (ev_begin) CSDR_HIP(hipEventRecord(ev_begin, st));
        auto go = [&](auto kern, bool &done) -> int {
            (void)done; { const int arc = lds_attr_once((const void *)kern, lds); if (arc) return arc; }
            hipLaunchKernelGGL(kern, dim3(n_qph, gy, z), dim3(256), lds, st, in, in_pitch, (const v4i *)dev.d_frags, dev.d_consts, dev.d_kb_of, ctab, demod, demod_pitch, wp);
            return 0;
        };
        static bool done[4] = {false, false, false, false};
        int grc = 0;
        switch (cfg) {
            case 1: grc = go(k_wfm_mfma_wg<32, 2>, done[1]); break;
            case 2: grc = go(k_wfm_mfma_wg<16, 2>, done[2]); break;
            case 3: grc = go(k_wfm_mfma_wg<16, 3>, done[3]); break;
            default: grc = go(k_wfm_mfma_wg<16, 5>, done[0]); break;
        }
        if (grc) return grc;
        CSDR_LAUNCH_CHECK();
        if (ev_end) CSDR_HIP(hipEventRecord(ev_end, st));
        g_last_kernel = "k_wfm_mfma_wg";
        return launch_edges(tile_first, 4 * qa - 1, 4 * (qb + 1), tile_last);
    }
    g_last_kernel = "k_wfm_mfma";
    if (ev_begin) CSDR_HIP(hipEventRecord(ev_begin, st));
    rc = launch(st, t_a, t_b, false); if (rc) return rc;                    // the bulk: no bounds logic at all
    if (ev_end) CSDR_HIP(hipEventRecord(ev_end, st));
    rc = launch(st_edge, tile_first, t_a - 1 < tile_last ? t_a - 1 : tile_last, true); if (rc) return rc;      // leading tiles (history)
    if (t_b + 1 > t_a - 1) rc = launch(st_edge, t_b + 1 > t_a ? t_b + 1 : t_a, tile_last, true);               // trailing tiles (ragged end / partial tile)
    return rc;
}

} // namespace csdr_amd

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

// Test hook: which front-end kernel the following csdr_amd_wfm_process calls use for the bulk of a block (-1 = default order: sequential,
// octet, quad, per-wave -- the first one whose preconditions hold).
extern "C" void csdr_amd_debug_wfm_select(int kernel) { csdr_amd::g_wfm_select = kernel; }

// Test hook (tests/test_mfma_table_cpu.py): evaluates ONE tile on the CPU exactly the way the kernel does -- same table,
// same lane/byte layout, same digit recombination -- so the table builder and the indexing are validated without a GPU.
// window: 64*WFM_NK raw u8 bytes of one stream starting at the tile's window base; out: the 16 rows after C_m/C_{m+1}.
extern "C" int csdr_amd_debug_wfm_mfma_tile(int D, int L, int F, float shift_rate, const float *taps, int phase, const uint8_t *window,
                                            const float *C0, const float *C1, float *out16, int *n_phases, int *straddle, int *tile_stride_bytes, int *win_off_bytes)
{
    if (!wfm_mfma_supported(D, L, F)) return -1;
    static WfmMfmaTable t; static int cD = 0, cL = 0, cF = 0; static float crate = 0; static std::vector<float> ctaps;
    if (cD != D || cL != L || cF != F || crate != shift_rate || ctaps != std::vector<float>(taps, taps + L)) {
        wfm_mfma_build_table(D, L, F, shift_rate, taps, t); cD = D; cL = L; cF = F; crate = shift_rate; ctaps.assign(taps, taps + L);
    }
    if (n_phases) *n_phases = t.n_phases;
    if (tile_stride_bytes) *tile_stride_bytes = t.tile_stride_bytes;
    if (win_off_bytes) *win_off_bytes = t.win_off_bytes;
    if (phase < 0 || phase >= t.n_phases) return -2;
    const int kb = t.kb_of[phase];
    if (straddle) *straddle = kb < WFM_NK;
    const float *cst = t.consts.data() + (size_t)phase * 32;
    const int8_t *fr = t.frags.data() + (size_t)phase * WFM_FRAG_V4 * 16;
    for (int r = 0; r < 16; r++) {
        long acc[3] = {0, 0, 0}, snap[3] = {0, 0, 0};
        auto step = [&](int frag0, int ks) {                          // one K-step of v_mfma_i32_16x16x64_i8, row r, three digit fragments
            for (int kg = 0; kg < 4; kg++) for (int b = 0; b < 16; b++) {
                const int v = (int)(int8_t)(window[64 * ks + 16 * kg + b] ^ 0x80);
                for (int l = 0; l < 3; l++) acc[l] += (long)fr[((size_t)(frag0 + l) * 64 + (16 * kg + r)) * 16 + b] * v;
            }
        };
        for (int ks = 0; ks < WFM_NK; ks++) {                         // same order as tile_product()
            if (ks == kb) { step(WFM_NK * 3, ks); for (int l = 0; l < 3; l++) snap[l] = acc[l]; }
            step(ks * 3, ks);
        }
        if (kb < WFM_NK) {
            out16[r] = fmaf(fmaf((float)snap[0], 65536.0f, fmaf((float)snap[1], 256.0f, (float)snap[2])), t.scale, cst[r]);
            out16[16 + r] = fmaf(fmaf((float)(acc[0] - snap[0]), 65536.0f, fmaf((float)(acc[1] - snap[1]), 256.0f, (float)(acc[2] - snap[2]))), t.scale, cst[16 + r]);
        } else {
            out16[r] = fmaf(fmaf((float)acc[0], 65536.0f, fmaf((float)acc[1], 256.0f, (float)acc[2])), t.scale, cst[r]);
            out16[16 + r] = 0.f;
        }
    }
    float res[16];
    for (int q = 0; q < 4; q++) for (int w = 0; w < 2; w++) {
        const float *u0 = out16 + 4 * q + 2 * w, *u1 = out16 + 16 + 4 * q + 2 * w;
        res[4 * q + 2 * w] = C0[0] * u0[0] - C0[1] * u0[1] + (C1[0] * u1[0] - C1[1] * u1[1]);
        res[4 * q + 2 * w + 1] = C0[0] * u0[1] + C0[1] * u0[0] + (C1[0] * u1[1] + C1[1] * u1[0]);
    }
    for (int r = 0; r < 16; r++) out16[r] = res[r];
    return 0;
}
