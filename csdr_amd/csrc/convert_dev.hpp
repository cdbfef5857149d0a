// convert_dev.hpp -- the integer -> float sample conversions of convert.hip (libcsdr.c:2363-2437) as device functions: one definition for the stand-alone
// converters and for kernels that take integer samples directly (the channelizer's forward transform, fastddc_mfma.hip), so that "convert_s16_f | fastddc_fwd_cc"
// fused is bit-equal to the two stages.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace csdr_amd {

template <int KIND>   // 0: u8, 1: s8, 2: s16
__device__ __forceinline__ float to_float(int raw)
{
    if (KIND == 0) {
        // (float)v/(255/2.0)-1.0 in double, rounded once (libcsdr.c:2365).  For all 256 codes this equals the
        // correctly rounded float quotient (2v-255)/255, which one Newton step on the reciprocal product
        // reproduces exactly (checked exhaustively against the oracle in tests/): 4 VALU ops, no fp64 divide.
        const float num = fmaf((float)raw, 2.0f, -255.0f);          // exact integer in [-255, 255]
        const float rcp = 0x1.010102p-8f;                            // RN(1/255) = 0x3b808081
        const float q = __fmul_rn(num, rcp);
        const float err = fmaf(-q, 255.0f, num);                     // exact residual
        return fmaf(err, rcp, q);
    } else if (KIND == 1) {
        // "/SCHAR_MAX" is a multiplication by the rounded reciprocal in the reference's -ffast-math build
        return __fmul_rn((float)raw, 1.0f / 127.0f);
    } else {
        return __fmul_rn((float)raw, 1.0f / 32767.0f);
    }
}


} // namespace csdr_amd
