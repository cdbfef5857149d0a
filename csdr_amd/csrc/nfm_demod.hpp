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
struct DdcFuseInfo { bool fused; long n_lead, seg_first, seg_outputs, n_seg, trail_first, n_trail; };      // seg_first: output index of segment 0's first sample (may be < 0: a partial first tile)

// x = y[k], p = y[k - 1] (libcsdr.c:1040-1071, :1130-1137); round(v / max_amp * NFM_XQ) = D0 * 65536 + D1 * 256 + D2 with Dj = (int8_t)d[j]
__device__ __forceinline__ void nfm_demod_digits(float2 x, float2 p, float max_amp, float q_per_amp, int (&d)[3])
{
    const float Kf = 0.340447550238101026565118445432744920253753662109375f;   // libcsdr.c:1021
    const float dq = x.y - p.y, di = x.x - p.x;
    const float num = x.x * dq - x.y * di, den = x.x * x.x + x.y * x.y;
    float rd = __builtin_amdgcn_rcpf(den);                                     // same evaluation as k_fmdemod (audio.hip)
    rd = fmaf(fmaf(-den, rd, 1.0f), rd, rd);
    float v = (den != 0.f) ? (Kf * num) * rd : 0.f;
    v = (max_amp < v) ? max_amp : v; v = (-max_amp > v) ? -max_amp : v;        // limit_ff libcsdr.c:1133-1136
    const int qv = __float2int_rn(v * q_per_amp);                              // |qv| <= NFM_XQ
    // balanced base-256 digits: d2 = sext8(qv), then (qv - d2) >> 8 = (qv + 128) >> 8, and so on.  The callers store the LOW BYTE of each entry, so the
    // sign extensions are never computed: two add / shift pairs instead of ten integer operations per sample
    d[2] = qv; d[1] = (qv + 128) >> 8; d[0] = (d[1] + 128) >> 8;
}

} // namespace csdr_amd
