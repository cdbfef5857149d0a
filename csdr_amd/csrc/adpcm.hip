// adpcm.hip -- IMA ADPCM codec: encode_ima_adpcm_i16_u8 / decode_ima_adpcm_u8_i16 (ima_adpcm.c:110-174) and the
// waterfall compressor `csdr compress_fft_adpcm_f_u8` (csdr.c:1745-1768).  Integer, bit exact.
// ENCODING is a strictly serial state machine (the code of a sample depends on the predictor the previous codes left): the parallel axis is the STREAM
// (one lane per stream), and the BLOCK for the waterfall compressor, whose encoder restarts from the zero state for every FFT row (one lane per row).
// DECODING is not serial: the step index is a chain of clamped adds of table values of the nibbles, the predictor a chain of clamped adds of
// diff(step[index], nibble) (ima_adpcm.c:109-131).  Maps x -> clamp(x + a, lo, hi) compose into maps of the same form (exactly, in integers), so both chains
// are parallel scans over time: k_adpcm_decode_scan, one workgroup per stream walking chunks of 8192 samples, 32 samples per lane.
#include "common.hpp"
#include <stdlib.h>
using namespace csdr_amd;

namespace {

__constant__ int c_step[89] = { 7, 8, 9, 10, 11, 12, 13, 14, 16, 17, 19, 21, 23, 25, 28, 31, 34, 37, 41, 45, 50, 55, 60, 66, 73, 80, 88, 97, 107, 118, 130, 143,
    157, 173, 190, 209, 230, 253, 279, 307, 337, 371, 408, 449, 494, 544, 598, 658, 724, 796, 876, 963, 1060, 1166, 1282, 1411, 1552, 1707, 1878, 2066,
    2272, 2499, 2749, 3024, 3327, 3660, 4026, 4428, 4871, 5358, 5894, 6484, 7132, 7845, 8630, 9493, 10442, 11487, 12635, 13899, 15289, 16818, 18500,
    20350, 22385, 24623, 27086, 29794, 32767 };                     // the standard IMA step table (ima_adpcm.c:98-108)

struct St { int index, prev; };

__device__ __forceinline__ int dec_one(unsigned code, St &s)
{   // ima_adpcm.c:110-134
    const int step = c_step[s.index];
    int diff = step >> 3;
    if (code & 1) diff += step >> 2;
    if (code & 2) diff += step >> 1;
    if (code & 4) diff += step;
    if (code & 8) diff = -diff;
    s.prev += diff;
    s.prev = s.prev > 32767 ? 32767 : (s.prev < -32768 ? -32768 : s.prev);
    s.index += (code & 4) ? 2 * (int)(code & 3) + 2 : -1;           // indexAdjustTable {-1,-1,-1,-1,2,4,6,8} twice (ima_adpcm.c:90-95)
    s.index = s.index < 0 ? 0 : (s.index > 88 ? 88 : s.index);
    return s.prev;
}
__device__ __forceinline__ unsigned enc_one(int sample, St &s)
{   // ima_adpcm.c:136-152
    int diff = sample - s.prev, step = c_step[s.index];
    unsigned code = 0;
    if (diff < 0) { code = 8; diff = -diff; }
    if (diff >= step) { code |= 4; diff -= step; }
    step >>= 1;
    if (diff >= step) { code |= 2; diff -= step; }
    step >>= 1;
    if (diff >= step) code |= 1;
    dec_one(code, s);
    return code;
}

__global__ void k_adpcm_encode(const int16_t *__restrict__ in, uint8_t *__restrict__ out, int n_streams, size_t n, size_t in_pitch, size_t out_pitch, int *__restrict__ state_io)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    St st{min(max(state_io[2 * s], 0), 88), state_io[2 * s + 1]};               // (a table index: clamped, see k_adpcm_decode_scan)
    const int16_t *x = in + (size_t)s * in_pitch; uint8_t *y = out + (size_t)s * out_pitch;
    for (size_t k = 0; k < n / 2; k++) { const unsigned lo = enc_one(x[2 * k], st), hi = enc_one(x[2 * k + 1], st); y[k] = (uint8_t)(lo | (hi << 4)); }
    state_io[2 * s] = st.index; state_io[2 * s + 1] = st.prev;
}
// The encoder for many streams: one wave = 64 streams, one lane each (the state machine is serial in time), but the samples reach the lanes through LDS: a chunk
// of 128 samples of 64 streams is fetched as 16-byte pieces (a lane that walks its own row in global memory touches a different line with every lane of every
// load), the codes leave the same way, and the step table sits in LDS.  What remains is the chain itself: ~35 dependent instructions and one LDS read per sample,
// ~385 cycles with one wave per SIMD (4096 x 48000: 7.7 ms against 9.1; 65536 x 12000: 2.2 against 5.4).  Requesting the five steps the next sample can see
// (index - 1, + 2, + 4, + 6, + 8) ahead of the compare chain was tried: the select among them costs more than the read it hides (11.9 / 3.3 ms).
// Bit exact (the same integer operations as enc_one / dec_one).  16-byte aligned rows only; anything else takes k_adpcm_encode.
constexpr int ENC_CH = 128;                                         // samples per stream and chunk
constexpr int ENC_IP = 2 * ENC_CH + 16, ENC_OP = ENC_CH / 2 + 16;   // LDS row pitches in bytes (16-byte multiples, bank spreading)
__global__ __launch_bounds__(64) void k_adpcm_encode_lds(const int16_t *__restrict__ in, uint8_t *__restrict__ out, int n_streams, size_t n, size_t in_pitch, size_t out_pitch,
                                                          int *__restrict__ state_io)
{
    __shared__ __attribute__((aligned(16))) uint8_t l_in[64 * ENC_IP];
    __shared__ __attribute__((aligned(16))) uint8_t l_out[64 * ENC_OP];
    __shared__ int l_step[89 + 8];
    const int lane = threadIdx.x, s0 = blockIdx.x * 64, s = s0 + lane;
    for (int k = lane; k < 89 + 8; k += 64) l_step[k] = c_step[k < 89 ? k : 88];      // (entries behind 88 repeat it: index + 8 needs no clamp of its own)
    const bool mine = s < n_streams;
    int index = mine ? min(max(state_io[2 * s], 0), 88) : 0, prev = mine ? state_io[2 * s + 1] : 0;
    const size_t n_codes = n & ~(size_t)1;                                              // samples that yield a code (pairs: ima_adpcm.c:154-163)
    __syncthreads();
    int step = l_step[index];
    for (size_t c0 = 0; c0 < n_codes; c0 += ENC_CH) {
        const int len = (int)min((size_t)ENC_CH, n_codes - c0);                         // even
        // ---- fetch: 64 rows x 256 bytes = 16 pieces per row; a wave instruction takes 4 rows
#pragma unroll 4
        for (int r4 = 0; r4 < 16; r4++) {
            const int row = 4 * r4 + (lane >> 4), pc = lane & 15;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (s0 + row < n_streams && 8 * pc < len) {
                const int16_t *src = in + (size_t)(s0 + row) * in_pitch + c0 + 8 * pc;
                if (8 * pc + 8 <= len) v = *reinterpret_cast<const uint4 *>(src);
                else { int16_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0}; for (int i = 0; 8 * pc + i < len; i++) t[i] = src[i]; v = *reinterpret_cast<const uint4 *>(t); }
            }
            *reinterpret_cast<uint4 *>(l_in + row * ENC_IP + 16 * pc) = v;
        }
        __syncthreads();
        // ---- this lane's stream, 8 samples (one 16-byte read) -> 4 code bytes at a time
        const uint8_t *my = l_in + lane * ENC_IP;
        uint32_t *mo = reinterpret_cast<uint32_t *>(l_out + lane * ENC_OP);
        for (int g = 0; 8 * g < len; g++) {
            const uint4 xv = *reinterpret_cast<const uint4 *>(my + 16 * g);
            const uint32_t xw[4] = {xv.x, xv.y, xv.z, xv.w};
            uint32_t packed = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (8 * g + i >= len) break;                                            // (wave uniform: only in a stream's last chunk)
                const int sample = (int)(int16_t)(xw[i >> 1] >> (16 * (i & 1)));
                int diff = sample - prev;                                               // ima_adpcm.c:136-152
                unsigned code = 0;
                if (diff < 0) { code = 8; diff = -diff; }
                int dq = step >> 3, st = step;
                if (diff >= st) { code |= 4; diff -= st; dq += st; }
                st >>= 1;
                if (diff >= st) { code |= 2; diff -= st; dq += st; }
                st >>= 1;
                if (diff >= st) { code |= 1; dq += st; }
                prev += (code & 8) ? -dq : dq;                                          // the decoder's update (ima_adpcm.c:110-134)
                prev = prev > 32767 ? 32767 : (prev < -32768 ? -32768 : prev);
                const unsigned m = code & 7;
                index = min(max(index + (m < 4 ? -1 : 2 * (int)(m & 3) + 2), 0), 88);
                step = l_step[index];
                packed |= code << (4 * i);
            }
            mo[g] = packed;
        }
        __syncthreads();
        // ---- store: 64 rows x 64 bytes = 4 pieces per row; a wave instruction takes 16 rows
#pragma unroll
        for (int r16 = 0; r16 < 4; r16++) {
            const int row = 16 * r16 + (lane >> 2), pc = lane & 3;
            if (s0 + row < n_streams && 32 * pc < len) {
                uint8_t *dst = out + (size_t)(s0 + row) * out_pitch + c0 / 2 + 16 * pc;
                const uint4 v = *reinterpret_cast<const uint4 *>(l_out + row * ENC_OP + 16 * pc);
                if (32 * pc + 32 <= len) *reinterpret_cast<uint4 *>(dst) = v;
                else { const uint8_t *b = reinterpret_cast<const uint8_t *>(&v); for (int i = 0; 32 * pc + 2 * i < len; i++) dst[i] = b[i]; }
            }
        }
        __syncthreads();
    }
    if (mine) { state_io[2 * s] = index; state_io[2 * s + 1] = prev; }
}
__global__ void k_adpcm_decode(const uint8_t *__restrict__ in, int16_t *__restrict__ out, int n_streams, size_t n, size_t in_pitch, size_t out_pitch, int *__restrict__ state_io)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    St st{min(max(state_io[2 * s], 0), 88), state_io[2 * s + 1]};               // (a table index: clamped, see k_adpcm_decode_scan)
    const uint8_t *x = in + (size_t)s * in_pitch; int16_t *y = out + (size_t)s * out_pitch;
    for (size_t k = 0; k < n; k++) { const unsigned b = x[k]; y[2 * k] = (int16_t)dec_one(b & 0xf, st); y[2 * k + 1] = (int16_t)dec_one((b >> 4) & 0xf, st); }
    state_io[2 * s] = st.index; state_io[2 * s + 1] = st.prev;
}
// ---- decode as two scans.  A clamped add f(x) = min(max(x + a, lo), hi); (f2 o f1)(x) = clamp(x + a1 + a2, clamp(lo1 + a2, lo2, hi2), clamp(hi1 + a2, lo2, hi2)).
struct CMap { int a, lo, hi; };
__device__ __forceinline__ int cclamp(int x, int lo, int hi) { return min(max(x, lo), hi); }
__device__ __forceinline__ CMap cmap_then(const CMap &f1, const CMap &f2) { return CMap{f1.a + f2.a, cclamp(f1.lo + f2.a, f2.lo, f2.hi), cclamp(f1.hi + f2.a, f2.lo, f2.hi)}; }
__device__ __forceinline__ int cmap_apply(const CMap &f, int x) { return cclamp(x + f.a, f.lo, f.hi); }
constexpr int CM_INF = 1 << 29;
__device__ __forceinline__ CMap cmap_id() { return CMap{0, -CM_INF, CM_INF}; }
__device__ __forceinline__ CMap cmap_shfl_up(const CMap &f, int d) { return CMap{__shfl_up(f.a, d, 64), __shfl_up(f.lo, d, 64), __shfl_up(f.hi, d, 64)}; }
// exclusive prefix of the lanes' maps over the 256 threads of a workgroup (time order = thread order), and the workgroup's total
__device__ __forceinline__ void cmap_scan256(CMap mine, CMap *wave_tot /* LDS [4] */, int tid, CMap &excl, CMap &total)
{
    const int lane = tid & 63, wv = tid >> 6;
    CMap inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const CMap up = cmap_shfl_up(inc, d); if (lane >= d) inc = cmap_then(up, inc); }
    if (lane == 63) wave_tot[wv] = inc;
    __syncthreads();
    CMap before = cmap_id(); total = cmap_id();
#pragma unroll
    for (int w = 0; w < 4; w++) { const CMap t = wave_tot[w]; if (w < wv) before = cmap_then(before, t); total = cmap_then(total, t); }
    CMap ex = cmap_shfl_up(inc, 1); if (lane == 0) ex = cmap_id();
    excl = cmap_then(before, ex);
    __syncthreads();                                                // wave_tot is reused by the next scan
}

constexpr int DEC_S = 32;                                           // samples (16 input bytes, 64 output bytes) per lane and chunk
__global__ __launch_bounds__(256) void k_adpcm_decode_scan(const uint8_t *__restrict__ in, int16_t *__restrict__ out, size_t n, size_t in_pitch, size_t out_pitch, int *__restrict__ state_io)
{
    // two look-up tables in LDS (divergent indices: not the scalar constant cache):
    //   l_dn[index][code] = (signed difference << 8) | next index   -- one read per sample gives both (|difference| < 2^17)
    //   l_pair[byte]      = the index map of the byte's TWO nibbles (low first), packed (a + 2) | lo << 8 | hi << 16: scan 1 composes per byte, not per sample
    __shared__ int l_dn[89 * 16];
    __shared__ int l_pair[256];
    __shared__ CMap wave_tot[4];
    const int s = blockIdx.x, tid = threadIdx.x;
    for (int k = tid; k < 89 * 16; k += 256) {
        const int idx = k >> 4, step = c_step[idx], code = k & 15;
        int d = step >> 3;
        if (code & 1) d += step >> 2;
        if (code & 2) d += step >> 1;
        if (code & 4) d += step;
        if (code & 8) d = -d;
        const int nx = cclamp(idx + ((code & 4) ? 2 * (code & 3) + 2 : -1), 0, 88);
        l_dn[k] = d * 256 + nx;
    }
    {
        const int c0 = tid & 15, c1 = tid >> 4;
        const CMap m = cmap_then(CMap{(c0 & 4) ? 2 * (c0 & 3) + 2 : -1, 0, 88}, CMap{(c1 & 4) ? 2 * (c1 & 3) + 2 : -1, 0, 88});
        l_pair[tid] = (m.a + 2) | (m.lo << 8) | (m.hi << 16);
    }
    const uint8_t *x = in + (size_t)s * in_pitch; int16_t *y = out + (size_t)s * out_pitch;
    int index0 = cclamp(state_io[2 * s], 0, 88), prev0 = state_io[2 * s + 1];      // state at the chunk's first sample (same value in every thread); the index is a table
                                                                                      // index from here on: a caller's value outside the table (the reference reads past its table, ima_adpcm.c:110) is clamped
    const bool in_al = (((uintptr_t)x) & 7) == 0, out_al = (((uintptr_t)y) & 15) == 0;
    __syncthreads();
    for (size_t c0 = 0; c0 < n; c0 += 256 * (DEC_S / 2)) {          // c0: first input byte of the chunk
        const size_t b0 = c0 + (size_t)tid * (DEC_S / 2);           // this lane's first byte
        const int nb = b0 >= n ? 0 : (int)min((size_t)(DEC_S / 2), n - b0);      // valid bytes of this lane (0 .. 8)
        unsigned long long bits[DEC_S / 16];
#pragma unroll
        for (int k = 0; k < DEC_S / 16; k++) bits[k] = 0;
        if (nb == DEC_S / 2 && in_al) {
#pragma unroll
            for (int k = 0; k < DEC_S / 16; k++) bits[k] = reinterpret_cast<const unsigned long long *>(x + b0)[k];
        } else {
#pragma unroll
            for (int k = 0; k < DEC_S / 2; k++) if (k < nb) bits[k / 8] |= (unsigned long long)x[b0 + k] << (8 * (k % 8));
        }
        const int ns = 2 * nb;                                       // valid samples: nibble j of `bits` = sample j (low nibble first, ima_adpcm.c:171-172)
        // ---- scan 1: the step index.  index += {-1,-1,-1,-1,2,4,6,8}[code & 7], clamped to [0, 88]
        CMap mi = cmap_id();
#pragma unroll
        for (int k = 0; k < DEC_S / 2; k++) {
            const int e = l_pair[(unsigned)(bits[k / 8] >> (8 * (k % 8))) & 0xff];
            if (k < nb) mi = cmap_then(mi, CMap{(e & 0xff) - 2, (e >> 8) & 0xff, e >> 16});
        }
        CMap ex, tot;
        cmap_scan256(mi, wave_tot, tid, ex, tot);
        int idx = cmap_apply(ex, index0);
        index0 = cmap_apply(tot, index0);
        // ---- the differences (they need the index in front of every sample), and scan 2: the predictor, clamped to int16
        int diff[DEC_S];
        CMap mp = cmap_id();
#pragma unroll
        for (int j = 0; j < DEC_S; j++) {
            const unsigned code = (unsigned)(bits[j / 16] >> (4 * (j % 16))) & 0xf;
            const int e = l_dn[idx * 16 + (int)code];
            const int d = e >> 8;
            diff[j] = d;
            if (j < ns) { mp = cmap_then(mp, CMap{d, -32768, 32767}); idx = e & 0xff; }
        }
        cmap_scan256(mp, wave_tot, tid, ex, tot);
        int pv = cmap_apply(ex, prev0);
        prev0 = cmap_apply(tot, prev0);
        // ---- outputs
        int16_t o[DEC_S];
#pragma unroll
        for (int j = 0; j < DEC_S; j++) { pv = cclamp(pv + diff[j], -32768, 32767); o[j] = (int16_t)pv; }
        int16_t *dst = y + 2 * b0;
        if (ns == DEC_S && out_al) {
#pragma unroll
            for (int k = 0; k < DEC_S / 8; k++) {
                uint4 v;
                v.x = (uint16_t)o[8 * k] | ((unsigned)(uint16_t)o[8 * k + 1] << 16); v.y = (uint16_t)o[8 * k + 2] | ((unsigned)(uint16_t)o[8 * k + 3] << 16);
                v.z = (uint16_t)o[8 * k + 4] | ((unsigned)(uint16_t)o[8 * k + 5] << 16); v.w = (uint16_t)o[8 * k + 6] | ((unsigned)(uint16_t)o[8 * k + 7] << 16);
                reinterpret_cast<uint4 *>(dst)[k] = v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < DEC_S; j++) if (j < ns) dst[j] = o[j];
        }
    }
    if (tid == 0) { state_io[2 * s] = index0; state_io[2 * s + 1] = prev0; }
}

__device__ __forceinline__ int db_to_short(float v)
{   // temp = input*100 stored to a short (csdr.c:1763): float product, truncation towards zero (x86 cvttss2si: 0x80000000 when out of range), low 16 bits
    const float p = v * 100;
    const int i = (p >= -2147483648.0f && p < 2147483648.0f) ? (int)p : (int)0x80000000;
    return (int)(int16_t)(i & 0xffff);
}
__global__ void k_compress_fft(const float *__restrict__ in, uint8_t *__restrict__ out, int n_blocks, int fft_size)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    const float *x = in + (size_t)b * fft_size;
    uint8_t *y = out + (size_t)b * ((fft_size + 10) / 2);
    St st{0, 0};                                                    // "we always return to original d at any new buffer" (csdr.c:1764)
    const int pad = db_to_short(x[0]);
    for (int k = 0; k < 5; k++) { const unsigned lo = enc_one(pad, st), hi = enc_one(pad, st); y[k] = (uint8_t)(lo | (hi << 4)); }
    for (int k = 0; k < fft_size / 2; k++) { const unsigned lo = enc_one(db_to_short(x[2 * k]), st), hi = enc_one(db_to_short(x[2 * k + 1]), st); y[5 + k] = (uint8_t)(lo | (hi << 4)); }
}

} // namespace

extern "C" {

int csdr_amd_encode_ima_adpcm_i16_u8(csdr_amd_ctx *c, const int16_t *in, uint8_t *out, int n_streams, size_t n, size_t in_pitch, size_t out_pitch, int *state_io)
{
    if (n < 2 || n_streams <= 0) return 0;
    // rows that start on 16-byte boundaries go through LDS (k_adpcm_encode_lds); CSDR_AMD_ADPCM_SERIAL (A/B, read once per process): the plain one-lane-per-stream walk
    static const bool serial = getenv("CSDR_AMD_ADPCM_SERIAL") != nullptr;
    const bool aligned = ((((size_t)in) | (in_pitch * 2) | ((size_t)out) | out_pitch) & 15) == 0;
    if (aligned && !serial) hipLaunchKernelGGL(k_adpcm_encode_lds, dim3(cdiv(n_streams, 64)), dim3(64), 0, c->stream, in, out, n_streams, n, in_pitch, out_pitch, state_io);
    else hipLaunchKernelGGL(k_adpcm_encode, dim3(cdiv(n_streams, 64)), dim3(64), 0, c->stream, in, out, n_streams, n, in_pitch, out_pitch, state_io);
    CSDR_LAUNCH_CHECK();
    return 0;
}
int csdr_amd_decode_ima_adpcm_u8_i16(csdr_amd_ctx *c, const uint8_t *in, int16_t *out, int n_streams, size_t n, size_t in_pitch, size_t out_pitch, int *state_io)
{
    if (!n || n_streams <= 0) return 0;
    // Two parallel scans per stream (k_adpcm_decode_scan: one 256-thread workgroup per stream, ~6 KiB of LDS tables built per workgroup, two 256-wide scans per 4096
    // input bytes) -- or one lane per stream (k_adpcm_decode), which wins only when the streams are many AND short.  Measured (tools/bench_adpcm_decode.py, ms per
    // call, lane per stream / scans): 65536 streams x 256 B 0.157 / 0.427; 65536 x 1024 B 0.955 / 0.433; 8192 x 1024 B 0.301 / 0.060; 4096 x 100 B 0.029 / 0.034;
    // 2048 x 4096 B 1.21 / 0.025; 1024 x 65536 B 19.7 / 0.155.  The walk costs ~0.3 us per input byte (twice that once more than 32768 lanes share the memory
    // system), whatever the stream count; the scans ~4 us per workgroup + ~9 us per chunk, 2048 streams in flight at a time.  The choice below is that model;
    // CSDR_AMD_ADPCM_SERIAL / CSDR_AMD_ADPCM_SCAN force one (read once per process).
    static const bool force_serial = getenv("CSDR_AMD_ADPCM_SERIAL") != nullptr, force_scan = getenv("CSDR_AMD_ADPCM_SCAN") != nullptr;
    const double t_serial = 0.3 * (double)n * (n_streams > 32768 ? (double)n_streams / 32768.0 : 1.0);
    const double t_scan = (double)(((size_t)n_streams + 2047) / 2048) * (4.0 + 9.0 * (double)((n + 4095) / 4096));
    const bool serial = force_serial || (!force_scan && t_serial < t_scan);
    if (serial) hipLaunchKernelGGL(k_adpcm_decode, dim3(cdiv(n_streams, 64)), dim3(64), 0, c->stream, in, out, n_streams, n, in_pitch, out_pitch, state_io);
    else hipLaunchKernelGGL(k_adpcm_decode_scan, dim3(n_streams), dim3(256), 0, c->stream, in, out, n, in_pitch, out_pitch, state_io);
    CSDR_LAUNCH_CHECK();
    return 0;
}
int csdr_amd_compress_fft_adpcm_f_u8(csdr_amd_ctx *c, const float *in, uint8_t *out, int n_blocks, int fft_size)
{
    if (n_blocks <= 0) return 0;
    if (fft_size <= 0 || (fft_size & 1)) return fail_msg(-3, "compress_fft_adpcm_f_u8: fft_size must be positive and even");
    hipLaunchKernelGGL(k_compress_fft, dim3(cdiv(n_blocks, 64)), dim3(64), 0, c->stream, in, out, n_blocks, fft_size);
    CSDR_LAUNCH_CHECK();
    return 0;
}

} // extern "C"
