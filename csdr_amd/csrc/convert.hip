// convert.hip -- sample-format converters, bit exact against the reference binary (libcsdr.c:2363-2437).
// Pure streaming, HBM bound: 16-byte vector accesses on the float side, grid-stride over 2048 blocks.
#include "common.hpp"
#include "convert_dev.hpp"
using namespace csdr_amd;

namespace {

// x86 cvttss2si semantics (the reference's float->int): NaN / out of int32 range -> 0x80000000.
// gfx950 v_cvt_i32_f32 saturates instead, so the indefinite value is selected explicitly.
__device__ __forceinline__ int trunc_i32(float x)
{
    return (x >= -2147483648.0f && x < 2147483648.0f) ? (int)x : (int)0x80000000;
}
__device__ __forceinline__ int trunc_i32(double x)
{
    return (x > -2147483649.0 && x < 2147483648.0) ? (int)x : (int)0x80000000;
}

// ---- X -> float: one thread produces 4 floats (one 16-byte store) per step; to_float<KIND> lives in convert_dev.hpp (the channelizer's integer ingest shares it)
template <int KIND, typename IN_T>
__global__ __launch_bounds__(256) void k_to_float(const IN_T *__restrict__ in, float *__restrict__ out, size_t n)
{
    const size_t nvec = n / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        float4 o;
        if (sizeof(IN_T) == 1) {
            const uint32_t w = reinterpret_cast<const uint32_t *>(in)[v];
            int a = w & 0xff, b = (w >> 8) & 0xff, c = (w >> 16) & 0xff, d = w >> 24;
            if (KIND == 1) { a = (int8_t)a; b = (int8_t)b; c = (int8_t)c; d = (int8_t)d; }
            o = make_float4(to_float<KIND>(a), to_float<KIND>(b), to_float<KIND>(c), to_float<KIND>(d));
        } else {
            const uint2 w = reinterpret_cast<const uint2 *>(in)[v];
            o = make_float4(to_float<KIND>((int16_t)(w.x & 0xffff)), to_float<KIND>((int16_t)(w.x >> 16)),
                            to_float<KIND>((int16_t)(w.y & 0xffff)), to_float<KIND>((int16_t)(w.y >> 16)));
        }
        reinterpret_cast<float4 *>(out)[v] = o;
    }
    // tail (n not a multiple of 4)
    const size_t t = nvec * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = to_float<KIND>((int)in[t]);
}

// ---- float -> X
template <int KIND>   // 0: u8, 1: s8, 2: s16
__device__ __forceinline__ int from_float(float x)
{
    if (KIND == 0) return trunc_i32((double)__fmul_rn(x, 255.0f) * 0.5 + 128.0) & 0xff;   // libcsdr.c:2380
    if (KIND == 1) return trunc_i32(__fmul_rn(x, 127.0f)) & 0xff;                          // :2387
    return trunc_i32(__fmul_rn(x, 32767.0f)) & 0xffff;                                     // :2397
}

template <int KIND, typename OUT_T>
__global__ __launch_bounds__(256) void k_from_float(const float *__restrict__ in, OUT_T *__restrict__ out, size_t n)
{
    const size_t nvec = n / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        const float4 x = reinterpret_cast<const float4 *>(in)[v];
        const int a = from_float<KIND>(x.x), b = from_float<KIND>(x.y), c = from_float<KIND>(x.z), d = from_float<KIND>(x.w);
        if (sizeof(OUT_T) == 1) reinterpret_cast<uint32_t *>(out)[v] = (uint32_t)a | ((uint32_t)b << 8) | ((uint32_t)c << 16) | ((uint32_t)d << 24);
        else reinterpret_cast<uint2 *>(out)[v] = make_uint2((uint32_t)a | ((uint32_t)b << 16), (uint32_t)c | ((uint32_t)d << 16));
    }
    const size_t t = nvec * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = (OUT_T)from_float<KIND>(in[t]);
}

// ---- 24-bit packed (libcsdr.c:2403-2437).  4 samples = 12 bytes = three dwords per thread.
__global__ __launch_bounds__(256) void k_f_s24(const float *__restrict__ in, uint8_t *__restrict__ out, size_t n, int lsb_first)
{
    const size_t nvec = n / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        const float4 x = reinterpret_cast<const float4 *>(in)[v];
        uint32_t s[4] = { (uint32_t)trunc_i32(__fmul_rn(x.x, 8388607.0f)), (uint32_t)trunc_i32(__fmul_rn(x.y, 8388607.0f)),
                          (uint32_t)trunc_i32(__fmul_rn(x.z, 8388607.0f)), (uint32_t)trunc_i32(__fmul_rn(x.w, 8388607.0f)) };
        uint32_t t[4];
        for (int k = 0; k < 4; k++) {
            const uint32_t b0 = s[k] & 0xff, b1 = (s[k] >> 8) & 0xff, b2 = (s[k] >> 16) & 0xff;
            t[k] = lsb_first ? (b0 | (b1 << 8) | (b2 << 16)) : (b2 | (b1 << 8) | (b0 << 16));   // 3 bytes in stream order
        }
        uint32_t *o = reinterpret_cast<uint32_t *>(out) + 3 * v;
        o[0] = t[0] | (t[1] << 24);
        o[1] = (t[1] >> 8) | (t[2] << 16);
        o[2] = (t[2] >> 16) | (t[3] << 8);
    }
    const size_t k = nvec * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) {
        const uint32_t s = (uint32_t)trunc_i32(__fmul_rn(in[k], 8388607.0f));
        const uint8_t b0 = s & 0xff, b1 = (s >> 8) & 0xff, b2 = (s >> 16) & 0xff;
        out[3 * k] = lsb_first ? b0 : b2; out[3 * k + 1] = b1; out[3 * k + 2] = lsb_first ? b2 : b0;
    }
}

__device__ __forceinline__ float s24_value(uint32_t three, int lsb_first)
{   // `three` holds the 3 stream bytes in its low 24 bits, first byte lowest
    const uint32_t p0 = three & 0xff, p1 = (three >> 8) & 0xff, p2 = (three >> 16) & 0xff;
    const uint32_t u = lsb_first ? ((p2 << 24) | (p1 << 16) | (p0 << 8)) : ((p2 << 8) | (p1 << 16) | (p0 << 24));
    return __fmul_rn((float)(int)u, 1.0f / 2147483392.0f);   // "/(float)(INT_MAX-256)" as reciprocal (fast-math build)
}

__global__ __launch_bounds__(256) void k_s24_f(const uint8_t *__restrict__ in, float *__restrict__ out, size_t n, int lsb_first)
{
    const size_t nvec = n / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        const uint32_t *w = reinterpret_cast<const uint32_t *>(in) + 3 * v;
        const uint32_t a = w[0], b = w[1], c = w[2];
        float4 o;
        o.x = s24_value(a & 0xffffff, lsb_first);
        o.y = s24_value((a >> 24) | ((b & 0xffff) << 8), lsb_first);
        o.z = s24_value((b >> 16) | ((c & 0xff) << 16), lsb_first);
        o.w = s24_value(c >> 8, lsb_first);
        reinterpret_cast<float4 *>(out)[v] = o;
    }
    const size_t k = nvec * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = s24_value((uint32_t)in[3 * k] | ((uint32_t)in[3 * k + 1] << 8) | ((uint32_t)in[3 * k + 2] << 16), lsb_first);
}

inline unsigned grid_for(size_t n_vec) { size_t g = (n_vec + 255) / 256; if (g < 1) g = 1; if (g > 2048) g = 2048; return (unsigned)g; }

} // namespace

#define ALIGN_CHECK(p, a) if (((uintptr_t)(p)) % (a)) return fail_msg(-3, "%s: pointer %s must be %d-byte aligned", __func__, #p, (int)(a))

extern "C" {

int csdr_amd_convert_u8_f(csdr_amd_ctx *c, const uint8_t *in, float *out, size_t n)
{ ALIGN_CHECK(in, 4); ALIGN_CHECK(out, 16); if (!n) return 0;
  hipLaunchKernelGGL((k_to_float<0, uint8_t>), dim3(grid_for(n / 4)), dim3(256), 0, c->stream, in, out, n); CSDR_LAUNCH_CHECK(); return 0; }
int csdr_amd_convert_s8_f(csdr_amd_ctx *c, const int8_t *in, float *out, size_t n)
{ ALIGN_CHECK(in, 4); ALIGN_CHECK(out, 16); if (!n) return 0;
  hipLaunchKernelGGL((k_to_float<1, int8_t>), dim3(grid_for(n / 4)), dim3(256), 0, c->stream, in, out, n); CSDR_LAUNCH_CHECK(); return 0; }
int csdr_amd_convert_s16_f(csdr_amd_ctx *c, const int16_t *in, float *out, size_t n)
{ ALIGN_CHECK(in, 8); ALIGN_CHECK(out, 16); if (!n) return 0;
  hipLaunchKernelGGL((k_to_float<2, int16_t>), dim3(grid_for(n / 4)), dim3(256), 0, c->stream, in, out, n); CSDR_LAUNCH_CHECK(); return 0; }
int csdr_amd_convert_f_u8(csdr_amd_ctx *c, const float *in, uint8_t *out, size_t n)
{ ALIGN_CHECK(in, 16); ALIGN_CHECK(out, 4); if (!n) return 0;
  hipLaunchKernelGGL((k_from_float<0, uint8_t>), dim3(grid_for(n / 4)), dim3(256), 0, c->stream, in, out, n); CSDR_LAUNCH_CHECK(); return 0; }
int csdr_amd_convert_f_s8(csdr_amd_ctx *c, const float *in, int8_t *out, size_t n)
{ ALIGN_CHECK(in, 16); ALIGN_CHECK(out, 4); if (!n) return 0;
  hipLaunchKernelGGL((k_from_float<1, int8_t>), dim3(grid_for(n / 4)), dim3(256), 0, c->stream, in, out, n); CSDR_LAUNCH_CHECK(); return 0; }
int csdr_amd_convert_f_s16(csdr_amd_ctx *c, const float *in, int16_t *out, size_t n)
{ ALIGN_CHECK(in, 16); ALIGN_CHECK(out, 8); if (!n) return 0;
  hipLaunchKernelGGL((k_from_float<2, int16_t>), dim3(grid_for(n / 4)), dim3(256), 0, c->stream, in, out, n); CSDR_LAUNCH_CHECK(); return 0; }
int csdr_amd_convert_f_s24(csdr_amd_ctx *c, const float *in, uint8_t *out, size_t n, int bigendian)
{ ALIGN_CHECK(in, 16); ALIGN_CHECK(out, 4); if (!n) return 0;
  hipLaunchKernelGGL(k_f_s24, dim3(grid_for(n / 4)), dim3(256), 0, c->stream, in, out, n, bigendian ? 1 : 0); CSDR_LAUNCH_CHECK(); return 0; }
int csdr_amd_convert_s24_f(csdr_amd_ctx *c, const uint8_t *in, float *out, size_t n, int bigendian)
{ ALIGN_CHECK(in, 4); ALIGN_CHECK(out, 16); if (!n) return 0;
  hipLaunchKernelGGL(k_s24_f, dim3(grid_for(n / 4)), dim3(256), 0, c->stream, in, out, n, bigendian ? 1 : 0); CSDR_LAUNCH_CHECK(); return 0; }

} // extern "C"
