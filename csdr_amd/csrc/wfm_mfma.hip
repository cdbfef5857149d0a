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

// Builds the periodic weight table.  Layout:
//   frags : [n_phases][WFM_SLOTS][3 digits][64 lanes] int8x16   (lane l: row l%16, K bytes 16*(l/16) .. +15 of the slot's K-step)
//   consts: [n_phases][2 parts][16 rows] float                  (the +1/255 offset of u8->float through the filter)
//   straddle: [n_phases] int                                    (-1, or the K-step that contains the chunk boundary)
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
    const size_t frag_bytes = (size_t)WFM_SLOTS * 3 * 64 * 16;
    t.frags.assign((size_t)t.n_phases * frag_bytes, 0);
    t.consts.assign((size_t)t.n_phases * 32, 0.f);
    t.straddle.assign(t.n_phases, -1);
    const int base_off_samples = t.win_off_bytes / 2;
    for (int ph = 0; ph < t.n_phases; ph++) {
        const long s0 = (long)4 * D * F * ph + base_off_samples;     // window base sample in the periodic frame
        const long chunk0 = s0 / 1024;
        const long bb = (chunk0 + 1) * 1024 - s0;                    // samples from the window base to the next chunk boundary
        const int S = (2 * bb < 64 * WFM_NK) ? (int)(2 * bb / 64) : -1;
        t.straddle[ph] = S;
        int8_t *fr = t.frags.data() + (size_t)ph * frag_bytes;
        float *cst = t.consts.data() + (size_t)ph * 32;
        for (int r = 0; r < 16; r++) {
            const int q = r / 4, which = (r % 4) / 2, comp = r % 2;
            const long off = (long)D * (F * q + 9 + which) - base_off_samples;        // row's first sample relative to the window base
            std::complex<double> csum[2] = {0, 0};
            for (int tp = 0; tp < L; tp++) {
                const long rel = off + tp, g = s0 + rel;
                const int part = (int)(g / 1024 - chunk0);
                const std::complex<double> G = a * (double)taps[tp] * Dk[g % 1024];
                csum[part] += (double)taps[tp] * Dk[g % 1024];
                for (int c = 0; c < 2; c++) {
                    // real form of (Gr + j Gi)(I + j Q): Re row takes (Gr, -Gi) on (I, Q); Im row takes (Gi, Gr)
                    const double val = comp == 0 ? (c == 0 ? G.real() : -G.imag()) : (c == 0 ? G.imag() : G.real());
                    const long col = 2 * rel + c;
                    const int ks = (int)(col / 64), b = (int)(col % 64);
                    int slot;
                    if (S < 0) slot = ks;
                    else if (ks < S) slot = ks;
                    else if (ks == S) slot = S + part;
                    else slot = ks + 1;
                    long qv = lrint(val * qscale);
                    const int w2 = (int)(((qv + 128) % 256 + 256) % 256) - 128; qv = (qv - w2) / 256;
                    const int w1 = (int)(((qv + 128) % 256 + 256) % 256) - 128; qv = (qv - w1) / 256;
                    const int w0 = (int)qv;
                    const int lane = 16 * (b / 16) + r, byte = b % 16;
                    const int dig[3] = {w0, w1, w2};
                    for (int l = 0; l < 3; l++) fr[((size_t)(slot * 3 + l) * 64 + lane) * 16 + byte] = (int8_t)dig[l];
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
    int tile_stride_bytes, win_off_bytes, n_phases;
    float scale;
};

template <int S>
__device__ __forceinline__ void tile_mfma(const v4i (&A)[WFM_SLOTS * 3], const v4i (&Bf)[WFM_NK], v4i (&acc0)[3], v4i (&acc1)[3])
{
#pragma unroll
    for (int ks = 0; ks < WFM_NK; ks++) {
        if (S < 0 || ks < S) {
#pragma unroll
            for (int l = 0; l < 3; l++) acc0[l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[ks * 3 + l], Bf[ks], acc0[l], 0, 0, 0);
        } else if (ks == S) {
#pragma unroll
            for (int l = 0; l < 3; l++) acc0[l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[ks * 3 + l], Bf[ks], acc0[l], 0, 0, 0);
#pragma unroll
            for (int l = 0; l < 3; l++) acc1[l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[(ks + 1) * 3 + l], Bf[ks], acc1[l], 0, 0, 0);
        } else {
#pragma unroll
            for (int l = 0; l < 3; l++) acc1[l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[(ks + 1) * 3 + l], Bf[ks], acc1[l], 0, 0, 0);
        }
    }
}

__device__ __forceinline__ float combine_digits(int a0, int a1, int a2)
{   // exact integers (<= 22 bits each) recombined in float: value = a0*65536 + a1*256 + a2
    return fmaf((float)a0, 65536.0f, fmaf((float)a1, 256.0f, (float)a2));
}

// One wave = one block of 64 streams (4 groups of 16) x tiles_per_wave consecutive tiles of 4 audio samples.
__global__ __launch_bounds__(64) void k_wfm_mfma(const uint8_t *__restrict__ in, size_t in_pitch, const uint8_t *__restrict__ hist,
                                                 const v4i *__restrict__ frags, const float *__restrict__ consts, const int *__restrict__ straddle,
                                                 const float2 *__restrict__ ctab, float *__restrict__ demod, size_t demod_pitch, MfmaParams p)
{
    const int lane = threadIdx.x;
    const int col = lane & 15, q = lane >> 4;
    const int sb = blockIdx.x;
    const long long t_begin = p.tile_first + (long long)blockIdx.y * p.tiles_per_wave;
    long long t_end = t_begin + p.tiles_per_wave;
    const long long t_last = p.tile_first + p.n_tiles;
    if (t_end > t_last) t_end = t_last;
    const long long two_T = 2LL * p.T;
    for (long long ti = t_begin; ti < t_end; ti++) {
        const int ph = (int)(ti % p.n_phases);
        const int S = straddle[ph];
        // weights of this tile phase -> registers (reused by the 4 stream groups)
        v4i A[WFM_SLOTS * 3];
        const v4i *fa = frags + (size_t)ph * (WFM_SLOTS * 3 * 64) + lane;
#pragma unroll
        for (int s = 0; s < WFM_SLOTS * 3; s++) A[s] = fa[s * 64];
        const float4 k0 = *reinterpret_cast<const float4 *>(consts + (size_t)ph * 32 + 4 * q);
        const float4 k1 = *reinterpret_cast<const float4 *>(consts + (size_t)ph * 32 + 16 + 4 * q);
        const long long wb = ti * p.tile_stride_bytes + p.win_off_bytes;          // global byte index of the window base
        const long long wbr = wb - 2 * p.B;                                        // relative to this block's first byte
        const long long chunk_rel = (wb / 2) / 1024 - p.B / 1024;                  // -1 for windows that start in the history
        const float2 C0 = ctab[chunk_rel + 1], C1 = ctab[chunk_rel + 2];
        const bool interior = (wbr >= 0) && (wbr + 64 * WFM_NK <= two_T);
        const long long j0 = 4 * ti;
#pragma unroll 1
        for (int g = 0; g < 4; g++) {
            int stream = sb * 64 + g * 16 + col;
            const bool stream_ok = stream < p.n_streams;
            if (stream >= p.n_streams) stream = p.n_streams - 1;
            const uint8_t *row = in + (size_t)stream * in_pitch;
            v4i Bf[WFM_NK];
            if (interior) {
                const uint8_t *src = row + wbr + 16 * q;
#pragma unroll
                for (int ks = 0; ks < WFM_NK; ks++) Bf[ks] = *reinterpret_cast<const v4i *>(src + 64 * ks);
            } else {
                const uint8_t *hrow = hist + (size_t)stream * (2 * WFM_HIST);
#pragma unroll
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
#pragma unroll
            for (int ks = 0; ks < WFM_NK; ks++) Bf[ks] ^= (int)0x80808080;              // u8 - 128 as int8
            v4i acc0[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}, acc1[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
            switch (S) {
                case 0: tile_mfma<0>(A, Bf, acc0, acc1); break;
                case 1: tile_mfma<1>(A, Bf, acc0, acc1); break;
                case 2: tile_mfma<2>(A, Bf, acc0, acc1); break;
                case 3: tile_mfma<3>(A, Bf, acc0, acc1); break;
                case 4: tile_mfma<4>(A, Bf, acc0, acc1); break;
                case 5: tile_mfma<5>(A, Bf, acc0, acc1); break;
                case 6: tile_mfma<6>(A, Bf, acc0, acc1); break;
                case 7: tile_mfma<7>(A, Bf, acc0, acc1); break;
                default: tile_mfma<-1>(A, Bf, acc0, acc1); break;
            }
            // lane (col, q): rows 4q..4q+3 = Re/Im of y[Fj+9], Re/Im of y[Fj+10] for audio j = j0+q of stream `col`
            float u0[4], u1[4];
            const float kk0[4] = {k0.x, k0.y, k0.z, k0.w}, kk1[4] = {k1.x, k1.y, k1.z, k1.w};
#pragma unroll
            for (int r = 0; r < 4; r++) {
                u0[r] = fmaf(combine_digits(acc0[0][r], acc0[1][r], acc0[2][r]), p.scale, kk0[r]);
                u1[r] = fmaf(combine_digits(acc1[0][r], acc1[1][r], acc1[2][r]), p.scale, kk1[r]);
            }
            // y = C_m * u0 + C_{m+1} * u1   (complex)
            const float pI = C0.x * u0[0] - C0.y * u0[1] + (C1.x * u1[0] - C1.y * u1[1]);
            const float pQ = C0.x * u0[1] + C0.y * u0[0] + (C1.x * u1[1] + C1.y * u1[0]);
            const float cI = C0.x * u0[2] - C0.y * u0[3] + (C1.x * u1[2] - C1.y * u1[3]);
            const float cQ = C0.x * u0[3] + C0.y * u0[2] + (C1.x * u1[3] + C1.y * u1[2]);
            // fmdemod_quadri_cf (libcsdr.c:1040-1071) on (previous = y[Fj+9], current = y[Fj+10])
            const float dq = cQ - pQ, di = cI - pI;
            const float num = cI * dq - cQ * di, den = cI * cI + cQ * cQ;
            const float K = 0.340447550238101026565118445432744920253753662109375f;
            const float a = (den != 0.f) ? (K * num) / den : 0.f;
            // gather the 4 audio samples of a stream into lane `col` and store them
            const float a1 = __shfl(a, col + 16, 64), a2 = __shfl(a, col + 32, 64), a3 = __shfl(a, col + 48, 64);
            if (q == 0 && stream_ok) {
                const long long jr = j0 - p.j_first;                       // position of the tile's first audio sample in this call's output
                float *dst = demod + (size_t)stream * demod_pitch;
                const float vals[4] = {a, a1, a2, a3};
                if (jr >= 0 && jr + 4 <= p.n_audio) {
#pragma unroll
                    for (int k = 0; k < 4; k++) dst[jr + k] = vals[k];
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++) if (jr + k >= 0 && jr + k < p.n_audio) dst[jr + k] = vals[k];
                }
            }
        }
    }
}

} // namespace

namespace csdr_amd {

int wfm_mfma_launch(hipStream_t st, const uint8_t *in, size_t in_pitch, const uint8_t *hist, const WfmMfmaDevice &dev, const float2 *ctab,
                    float *demod, size_t demod_pitch, int n_streams, int T, long long B, long long j_first, int n_audio)
{
    MfmaParams p;
    p.n_streams = n_streams; p.T = T; p.B = B; p.j_first = j_first; p.n_audio = n_audio;
    p.tile_first = j_first / 4;
    const long long tile_last = (j_first + n_audio - 1) / 4;
    p.n_tiles = (int)(tile_last - p.tile_first + 1);
    p.tile_stride_bytes = dev.tile_stride_bytes; p.win_off_bytes = dev.win_off_bytes; p.n_phases = dev.n_phases; p.scale = dev.scale;
    const int n_sb = (n_streams + 63) / 64;
    // aim at >= 2048 waves (2 per SIMD) while keeping >= 8 tiles per wave
    int chunks = (2048 + n_sb - 1) / n_sb;
    if (chunks > p.n_tiles / 8) chunks = p.n_tiles / 8;
    if (chunks < 1) chunks = 1;
    if (chunks > 65535) chunks = 65535;
    p.tiles_per_wave = (p.n_tiles + chunks - 1) / chunks;
    chunks = (p.n_tiles + p.tiles_per_wave - 1) / p.tiles_per_wave;
    hipLaunchKernelGGL(k_wfm_mfma, dim3(n_sb, chunks), dim3(64), 0, st, in, in_pitch, hist, (const v4i *)dev.d_frags, dev.d_consts, dev.d_straddle,
                       ctab, demod, demod_pitch, p);
    CSDR_LAUNCH_CHECK();
    return 0;
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
    const int S = t.straddle[phase];
    if (straddle) *straddle = S;
    const int8_t *fr = t.frags.data() + (size_t)phase * WFM_SLOTS * 3 * 64 * 16;
    const float *cst = t.consts.data() + (size_t)phase * 32;
    for (int r = 0; r < 16; r++) {
        long acc[2][3] = {{0, 0, 0}, {0, 0, 0}};
        for (int ks = 0; ks < WFM_NK; ks++) {
            for (int pass = 0; pass < 2; pass++) {
                int slot, part;
                if (S < 0 || ks < S) { if (pass) continue; slot = ks; part = 0; }
                else if (ks == S) { slot = ks + pass; part = pass; }
                else { if (pass) continue; slot = ks + 1; part = 1; }
                for (int kg = 0; kg < 4; kg++) for (int b = 0; b < 16; b++) {
                    const int v = (int)(int8_t)(window[64 * ks + 16 * kg + b] ^ 0x80);
                    for (int l = 0; l < 3; l++) acc[part][l] += (long)fr[((size_t)(slot * 3 + l) * 64 + (16 * kg + r)) * 16 + b] * v;
                }
            }
        }
        float u[2];
        for (int p = 0; p < 2; p++) u[p] = fmaf(fmaf((float)acc[p][0], 65536.0f, fmaf((float)acc[p][1], 256.0f, (float)acc[p][2])), t.scale, cst[p * 16 + r]);
        out16[r] = u[0]; out16[16 + r] = u[1];
    }
    // complex recombination exactly as the kernel's epilogue
    float res[16];
    for (int q = 0; q < 4; q++) for (int w = 0; w < 2; w++) {
        const float *u0 = out16 + 4 * q + 2 * w, *u1 = out16 + 16 + 4 * q + 2 * w;
        res[4 * q + 2 * w] = C0[0] * u0[0] - C0[1] * u0[1] + (C1[0] * u1[0] - C1[1] * u1[1]);
        res[4 * q + 2 * w + 1] = C0[0] * u0[1] + C0[1] * u0[0] + (C1[0] * u1[1] + C1[1] * u1[0]);
    }
    for (int r = 0; r < 16; r++) out16[r] = res[r];
    return 0;
}
