// nfm_demod.hpp -- fmdemod_quadri_cf | limit_ff of ONE sample and its three balanced base-256 digits (the input format of the de-emphasis FIR on the
// matrix cores, nfm.hip), shared by the stand-alone pass k_nfm_demod_limit, the boundary pass k_nfm_demod_boundary and the fused epilogue of the front
// end's matrix-core kernel (ddc_mfma.hip): one definition, so every path rounds alike.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace csdr_amd {

// where the fused front end puts the limited demodulator output: planes [3][n_streams][dl_pitch] int8, sample k of the call at offset dl_fill + k
struct DdcFuse { int8_t *planes; size_t plane_bytes, dl_pitch; int dl_fill; float max_amp, q_per_amp; };
// which outputs of a call the fused kernel could NOT demodulate itself (their predecessor was computed by another workgroup or kernel); y holds them
struct DdcFuseInfo { bool fused; long n_lead, seg_outputs, n_seg, trail_first, n_trail; };

// x = y[k], p = y[k - 1] (libcsdr.c:1040-1071, :1130-1137); digit j of round(v / max_amp * NFM_XQ) = d[0] * 65536 + d[1] * 256 + d[2]
__device__ __forceinline__ void nfm_demod_digits(float2 x, float2 p, float max_amp, float q_per_amp, int (&d)[3])
{
    const float Kf = 0.340447550238101026565118445432744920253753662109375f;   // libcsdr.c:1021
    const float dq = x.y - p.y, di = x.x - p.x;
    const float num = x.x * dq - x.y * di, den = x.x * x.x + x.y * x.y;
    float rd = __builtin_amdgcn_rcpf(den);                                     // same evaluation as k_fmdemod (audio.hip)
    rd = fmaf(fmaf(-den, rd, 1.0f), rd, rd);
    float v = (den != 0.f) ? (Kf * num) * rd : 0.f;
    v = (max_amp < v) ? max_amp : v; v = (-max_amp > v) ? -max_amp : v;        // limit_ff libcsdr.c:1133-1136
    int qv = __float2int_rn(v * q_per_amp);                                    // |qv| <= NFM_XQ
    d[2] = ((qv + 128) & 255) - 128; qv = (qv - d[2]) >> 8;
    d[1] = ((qv + 128) & 255) - 128; qv = (qv - d[1]) >> 8;
    d[0] = qv;
}

} // namespace csdr_amd
