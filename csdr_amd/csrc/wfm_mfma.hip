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

// Builds the periodic weight table.  A window that contains a shift_addition_cc chunk boundary gets TWO weight sets
// (samples before / after the boundary, the other side zero) so that each side can be scaled by its own chunk phasor.
//   frags : [n_sets][WFM_NK][3 digits][64 lanes] int8x16   (lane l: row l%16, K bytes 16*(l/16) .. +15 of that K-step)
//   set_of: [n_phases][2]                                  (weight set of each side; -1 when the window has one side only)
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
    const size_t set_bytes = (size_t)WFM_FRAG_V4 * 16;
    t.frags.clear();
    t.consts.assign((size_t)t.n_phases * 32, 0.f);
    t.set_of.assign((size_t)t.n_phases * 2, -1);
    const int base_off_samples = t.win_off_bytes / 2;
    int n_sets = 0;
    for (int ph = 0; ph < t.n_phases; ph++) {
        const long s0 = (long)4 * D * F * ph + base_off_samples;     // window base sample in the periodic frame
        const long chunk0 = s0 / 1024;
        const bool two_sides = (chunk0 + 1) * 1024 - s0 < 32 * WFM_NK;   // a chunk boundary inside the 256-sample window
        t.set_of[2 * ph] = n_sets++;
        if (two_sides) t.set_of[2 * ph + 1] = n_sets++;
        t.frags.resize((size_t)n_sets * set_bytes, 0);
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
                int8_t *fr = t.frags.data() + (size_t)t.set_of[2 * ph + side] * set_bytes;
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
                    for (int l = 0; l < 3; l++) fr[((size_t)(ks * 3 + l) * 64 + lane) * 16 + byte] = (int8_t)dig[l];
                }
            }
            for (int p = 0; p < 2; p++) {
                const std::complex<double> k = std::complex<double>(1, 1) * csum[p] * c0;
                cst[p * 16 + r] = (float)(comp == 0 ? k.real() : k.imag());
            }
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
#pragma unroll 1
        for (int ks = 0; ks < WFM_NK; ks++) {
            const long long off = wbr + 64 * ks + 16 * q;
            v4i v = {(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};
            if (off < 0) { if (off >= -2 * WFM_HIST) v = *reinterpret_cast<const v4i *>(hrow + off + 2 * WFM_HIST); }
            else if (off + 16 <= two_T) v = *reinterpret_cast<const v4i *>(row + off);
            else { // ragged end of the last block: byte-wise
                uint32_t w[4] = {0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};
                for (int k = 0; k < 16; k++) if (off + k < two_T) { const uint32_t by = row[off + k]; w[k / 4] = (w[k / 4] & ~(0xffu << (8 * (k & 3)))) | (by << (8 * (k & 3))); }
                v = v4i{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
            }
            Bf[ks] = v;
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
                                                 const v4i *__restrict__ frags, const float *__restrict__ consts, const int *__restrict__ set_of,
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
    const int set0 = set_of[2 * ph], set1 = set_of[2 * ph + 1];
    const bool two = set1 >= 0;
    v4i A0[WFM_NK * 3], A1[WFM_NK * 3];
    {
        const v4i *fa = frags + (size_t)set0 * WFM_FRAG_V4 + lane;
#pragma unroll
        for (int s = 0; s < WFM_NK * 3; s++) A0[s] = fa[s * 64];
        const v4i *fb = frags + (size_t)(two ? set1 : set0) * WFM_FRAG_V4 + lane;
#pragma unroll
        for (int s = 0; s < WFM_NK * 3; s++) A1[s] = fb[s * 64];
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
            v4i acc0[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}, acc1[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
            for (int ks = 0; ks < WFM_NK; ks++)
#pragma unroll
                for (int l = 0; l < 3; l++) acc0[l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A0[ks * 3 + l], Bq[g][ks], acc0[l], 0, 0, 0);
            if (two) {
#pragma unroll
                for (int ks = 0; ks < WFM_NK; ks++)
#pragma unroll
                    for (int l = 0; l < 3; l++) acc1[l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A1[ks * 3 + l], Bq[g][ks], acc1[l], 0, 0, 0);
            }
            if (refill) {
                if (!EDGE) {
#pragma unroll
                    for (int ks = 0; ks < WFM_NK; ks++) Bq[g][ks] = *reinterpret_cast<const v4i *>(rowp[g] + (wbr + step) + 64 * ks);
                } else load_B<true>(Bq[g], in, in_pitch, hist, min(sidx[g], last_stream), wbr + step, two_T, q);
            }
            // lane (col, q): rows 4q..4q+3 = Re/Im of y[Fj+9], Re/Im of y[Fj+10] for audio j = 4*ti+q of stream col;  y = C_m u0 + C_{m+1} u1
            float pI, pQ, cI, cQ;
            {
                float u0[4];
#pragma unroll
                for (int r = 0; r < 4; r++) u0[r] = fmaf(combine_digits(acc0[0][r], acc0[1][r], acc0[2][r]), p.scale, k0[r]);
                pI = C0.x * u0[0] - C0.y * u0[1]; pQ = C0.x * u0[1] + C0.y * u0[0];
                cI = C0.x * u0[2] - C0.y * u0[3]; cQ = C0.x * u0[3] + C0.y * u0[2];
            }
            if (two) {
                float u1[4];
#pragma unroll
                for (int r = 0; r < 4; r++) u1[r] = fmaf(combine_digits(acc1[0][r], acc1[1][r], acc1[2][r]), p.scale, k1[r]);
                pI += C1.x * u1[0] - C1.y * u1[1]; pQ += C1.x * u1[1] + C1.y * u1[0];
                cI += C1.x * u1[2] - C1.y * u1[3]; cQ += C1.x * u1[3] + C1.y * u1[2];
            }
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
                                                     const v4i *__restrict__ frags, const float *__restrict__ consts, const int *__restrict__ set_of,
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
    const int set0 = set_of[2 * ph], set1 = set_of[2 * ph + 1];
    const bool two = set1 >= 0;
    v4i A0[WFM_NK * 3], A1[WFM_NK * 3];
    {
        const v4i *fa = frags + (size_t)set0 * WFM_FRAG_V4 + lane;
#pragma unroll
        for (int s = 0; s < WFM_NK * 3; s++) A0[s] = fa[s * 64];
        const v4i *fb = frags + (size_t)(two ? set1 : set0) * WFM_FRAG_V4 + lane;
#pragma unroll
        for (int s = 0; s < WFM_NK * 3; s++) A1[s] = fb[s * 64];
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
            v4i acc0[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
            for (int ks = 0; ks < WFM_NK; ks++)
#pragma unroll
                for (int l = 0; l < 3; l++) acc0[l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A0[ks * 3 + l], Bf[ks], acc0[l], 0, 0, 0);
            // the second weight set (window straddles a 1024-chunk) lives entirely inside one branch: its accumulators need no zero
            // initialisation on the common path (they sit in AGPRs: every touch is a VALU instruction of the wave's serial chain)
            float sI = 0.f, sQ = 0.f, tI = 0.f, tQ = 0.f;
            if (two) {
                v4i acc1[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
                for (int ks = 0; ks < WFM_NK; ks++)
#pragma unroll
                    for (int l = 0; l < 3; l++) acc1[l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A1[ks * 3 + l], Bf[ks], acc1[l], 0, 0, 0);
                float u1[4];
#pragma unroll
                for (int r = 0; r < 4; r++) u1[r] = fmaf(combine_digits(acc1[0][r], acc1[1][r], acc1[2][r]), p.scale, k1[r]);
                sI = C1.x * u1[0] - C1.y * u1[1]; sQ = C1.x * u1[1] + C1.y * u1[0];
                tI = C1.x * u1[2] - C1.y * u1[3]; tQ = C1.x * u1[3] + C1.y * u1[2];
            }
            float pI, pQ, cI, cQ;
            {
                float u0[4];
#pragma unroll
                for (int r = 0; r < 4; r++) u0[r] = fmaf(combine_digits(acc0[0][r], acc0[1][r], acc0[2][r]), p.scale, k0[r]);
                pI = C0.x * u0[0] - C0.y * u0[1]; pQ = C0.x * u0[1] + C0.y * u0[0];
                cI = C0.x * u0[2] - C0.y * u0[3]; cQ = C0.x * u0[3] + C0.y * u0[2];
            }
            if (two) { pI += sI; pQ += sQ; cI += tI; cQ += tQ; }
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

} // namespace

namespace csdr_amd {

static const char *g_last_kernel = "k_wfm_mfma";
const char *wfm_mfma_last_kernel() { return g_last_kernel; }

int wfm_mfma_launch(hipStream_t st, hipStream_t st_edge, hipEvent_t ev_begin, hipEvent_t ev_end, const uint8_t *in, size_t in_pitch, const uint8_t *hist, const WfmMfmaDevice &dev, const float2 *ctab,
                    float *demod, size_t demod_pitch, int n_streams, int T, long long B, long long j_first, int n_audio)
{
    MfmaParams p;
    p.n_streams = n_streams; p.T = T; p.B = B; p.j_first = j_first; p.n_audio = n_audio;
    p.tile_stride_bytes = dev.tile_stride_bytes; p.win_off_bytes = dev.win_off_bytes; p.n_phases = dev.n_phases; p.scale = dev.scale;
    const long long tile_first = j_first / 4, tile_last = (j_first + n_audio - 1) / 4;
    p.tile_out0 = tile_first;                                        // the scratch rows hold whole tiles; k_wfm_back skips j_first - 4*tile_first samples
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
        if (edge) hipLaunchKernelGGL((k_wfm_mfma<true>), grid, dim3(64), 0, ls, in, in_pitch, hist, (const v4i *)dev.d_frags, dev.d_consts, dev.d_set_of, ctab, demod, demod_pitch, p);
        else      hipLaunchKernelGGL((k_wfm_mfma<false>), grid, dim3(64), 0, ls, in, in_pitch, hist, (const v4i *)dev.d_frags, dev.d_consts, dev.d_set_of, ctab, demod, demod_pitch, p);
        CSDR_LAUNCH_CHECK();
        return 0;
    };
    static int use_wg = -1;
    if (use_wg < 0) { const char *e = getenv("CSDR_AMD_WFM_WG"); use_wg = e ? atoi(e) : 1; }
    int rc = 0;
    const long long qa = (t_a + 3) / 4, qb = (t_b + 1) / 4 - 1;             // whole quads inside the interior tile range
    if (use_wg && (p.n_phases % 4) == 0 && 3 * p.tile_stride_bytes + 64 * WFM_NK == 1712 && qb - qa + 1 >= 2 * (p.n_phases / 4)) {
        WgParams wp;
        wp.n_streams = n_streams; wp.B2 = 2 * B; wp.quad_first = qa; wp.n_quads = (int)(qb - qa + 1); wp.tile_out0 = tile_first;
        wp.stride = p.tile_stride_bytes; wp.win_off = p.win_off_bytes; wp.n_phases = p.n_phases; wp.scale = p.scale;
        wp.row_bytes = 3 * wp.stride + 64 * WFM_NK;                       // rows back to back (the DMA fills one contiguous run); 107 slots: odd
        static int cfg = -1;                                             // 0: 16 streams x ring of 5 quads (default), 1: 32 streams x ring of 2
        if (cfg < 0) { const char *e = getenv("CSDR_AMD_WFM_WGCFG"); cfg = e ? atoi(e) : 0; }
        const int SBv = cfg == 1 ? 32 : 16, NBv = cfg == 1 ? 2 : 5;
        const int n_qph = p.n_phases / 4, n_wsb = (n_streams + SBv - 1) / SBv;
        const long long per_qph = (wp.n_quads + n_qph - 1) / n_qph;
        // persistent grid: one workgroup per CU (256); a workgroup owns a quad phase and walks its share of the stream blocks
        static int n_cu = 0;
        if (!n_cu) { hipDeviceProp_t pr; int d = 0; (void)hipGetDevice(&d); n_cu = (hipGetDeviceProperties(&pr, d) == hipSuccess) ? pr.multiProcessorCount : 256; if (n_cu < 1) n_cu = 256; }
        int gy = n_cu / n_qph; if (gy < 1) gy = 1; if (gy > n_wsb) gy = n_wsb;
        int z = n_cu / (n_qph * gy); if (z < 1) z = 1;
        if (z > per_qph / 8) z = (int)(per_qph / 8);
        if (z < 1) z = 1;
        const size_t lds = (size_t)NBv * (4 * ((SBv * 107 + 255) / 256) * 1024) + 2 * SBv * 16 * sizeof(float);
        if (wp.row_bytes != 1712) return fail_msg(-3, "wfm: workgroup kernel is specialised for 1712-byte quad rows");
        if (ev_begin) CSDR_HIP(hipEventRecord(ev_begin, st));
        if (cfg == 1) {
            static bool done1 = false;
            if (!done1) { CSDR_HIP(hipFuncSetAttribute((const void *)k_wfm_mfma_wg<32, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); done1 = true; }
            hipLaunchKernelGGL((k_wfm_mfma_wg<32, 2>), dim3(n_qph, gy, z), dim3(256), lds, st, in, in_pitch, (const v4i *)dev.d_frags, dev.d_consts, dev.d_set_of, ctab, demod, demod_pitch, wp);
        } else {
            static bool done0 = false;
            if (!done0) { CSDR_HIP(hipFuncSetAttribute((const void *)k_wfm_mfma_wg<16, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); done0 = true; }
            hipLaunchKernelGGL((k_wfm_mfma_wg<16, 5>), dim3(n_qph, gy, z), dim3(256), lds, st, in, in_pitch, (const v4i *)dev.d_frags, dev.d_consts, dev.d_set_of, ctab, demod, demod_pitch, wp);
        }
        CSDR_LAUNCH_CHECK();
        if (ev_end) CSDR_HIP(hipEventRecord(ev_end, st));
        g_last_kernel = "k_wfm_mfma_wg";
        // leftovers around the quad range run on the per-wave kernel (bounds-checked variant: a handful of tiles)
        const long long a0 = tile_first, a1 = 4 * qa - 1, b0 = 4 * (qb + 1), b1 = tile_last;
        if (a1 >= a0 && b1 >= b0) {                                          // both ends in ONE launch (each is latency bound: ~22 us)
            p.tile_first = a0; p.n_tiles = (int)(a1 - a0 + 1); p.tile_first_b = b0; p.n_tiles_b = (int)(b1 - b0 + 1); p.two_ranges = 1; p.tiles_per_wave = 0;
            hipLaunchKernelGGL((k_wfm_mfma<true>), dim3(n_sb, p.n_phases, 2), dim3(64), 0, st_edge, in, in_pitch, hist, (const v4i *)dev.d_frags, dev.d_consts, dev.d_set_of, ctab, demod, demod_pitch, p);
            CSDR_LAUNCH_CHECK();
            return 0;
        }
        rc = launch(st_edge, a0, a1, true); if (rc) return rc;
        rc = launch(st_edge, b0, b1, true);
        return rc;
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
    if (straddle) *straddle = t.set_of[2 * phase + 1] >= 0;
    const float *cst = t.consts.data() + (size_t)phase * 32;
    for (int side = 0; side < 2; side++) {
        const int set = t.set_of[2 * phase + side];
        for (int r = 0; r < 16; r++) {
            long acc[3] = {0, 0, 0};
            if (set >= 0) {
                const int8_t *fr = t.frags.data() + (size_t)set * WFM_FRAG_V4 * 16;
                for (int ks = 0; ks < WFM_NK; ks++) for (int kg = 0; kg < 4; kg++) for (int b = 0; b < 16; b++) {
                    const int v = (int)(int8_t)(window[64 * ks + 16 * kg + b] ^ 0x80);
                    for (int l = 0; l < 3; l++) acc[l] += (long)fr[((size_t)(ks * 3 + l) * 64 + (16 * kg + r)) * 16 + b] * v;
                }
                out16[16 * side + r] = fmaf(fmaf((float)acc[0], 65536.0f, fmaf((float)acc[1], 256.0f, (float)acc[2])), t.scale, cst[side * 16 + r]);
            } else out16[16 * side + r] = 0.f;
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
