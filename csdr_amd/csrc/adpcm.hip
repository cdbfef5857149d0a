// adpcm.hip -- IMA ADPCM codec: encode_ima_adpcm_i16_u8 / decode_ima_adpcm_u8_i16 (ima_adpcm.c:110-174) and the
// waterfall compressor `csdr compress_fft_adpcm_f_u8` (csdr.c:1745-1768).  Integer, bit exact.
// The codec is a strictly serial state machine (predictor + step index) per stream, so the parallel axis is the STREAM for the two codec calls
// (one lane per stream) and the BLOCK for the waterfall compressor, whose encoder restarts from the zero state for every FFT row (one lane per row).
#include "common.hpp"
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
    St st{state_io[2 * s], state_io[2 * s + 1]};
    const int16_t *x = in + (size_t)s * in_pitch; uint8_t *y = out + (size_t)s * out_pitch;
    for (size_t k = 0; k < n / 2; k++) { const unsigned lo = enc_one(x[2 * k], st), hi = enc_one(x[2 * k + 1], st); y[k] = (uint8_t)(lo | (hi << 4)); }
    state_io[2 * s] = st.index; state_io[2 * s + 1] = st.prev;
}
__global__ void k_adpcm_decode(const uint8_t *__restrict__ in, int16_t *__restrict__ out, int n_streams, size_t n, size_t in_pitch, size_t out_pitch, int *__restrict__ state_io)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    St st{state_io[2 * s], state_io[2 * s + 1]};
    const uint8_t *x = in + (size_t)s * in_pitch; int16_t *y = out + (size_t)s * out_pitch;
    for (size_t k = 0; k < n; k++) { const unsigned b = x[k]; y[2 * k] = (int16_t)dec_one(b & 0xf, st); y[2 * k + 1] = (int16_t)dec_one((b >> 4) & 0xf, st); }
    state_io[2 * s] = st.index; state_io[2 * s + 1] = st.prev;
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
    hipLaunchKernelGGL(k_adpcm_encode, dim3(cdiv(n_streams, 64)), dim3(64), 0, c->stream, in, out, n_streams, n, in_pitch, out_pitch, state_io);
    CSDR_LAUNCH_CHECK();
    return 0;
}
int csdr_amd_decode_ima_adpcm_u8_i16(csdr_amd_ctx *c, const uint8_t *in, int16_t *out, int n_streams, size_t n, size_t in_pitch, size_t out_pitch, int *state_io)
{
    if (!n || n_streams <= 0) return 0;
    hipLaunchKernelGGL(k_adpcm_decode, dim3(cdiv(n_streams, 64)), dim3(64), 0, c->stream, in, out, n_streams, n, in_pitch, out_pitch, state_io);
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
