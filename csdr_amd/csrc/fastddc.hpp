// fastddc.hpp -- pieces shared by the fastddc inverse of fftpath.hip (general geometry, hipFFT) and its matrix-core path fastddc_mfma.hip
// (config 4's geometry: fft_inv_size 512).
#pragma once
#include "common.hpp"

namespace csdr_amd {

struct ChanGeom { int offsetbin; float sindelta, cosdelta, rate2; };   // per channel: fastddc_t.offsetbin, dsadata (fastddc.h:5-24)
struct DdcChanState { int remain; float phase; };                     // decimating_shift_addition_status_t carried between blocks (libcsdr_gpl.h:39-44)

struct DdcMfma;   // device-side plan of the matrix-core path

// nullptr when the geometry is not the one this path implements (the caller keeps the general kernels)
DdcMfma *ddc_mfma_create(csdr_amd_ctx *ctx, int fft, int inv, int pre, int n_channels, int max_blocks, int scrap, int post_in, int post_dec, int input_size, int overlap);
void ddc_mfma_destroy(DdcMfma *m);
// (re)build the kernel-side layout of the taps spectra of channels [c_first, c_first + c_count) from the natural [channel][fft] array
int ddc_mfma_set_taps(DdcMfma *m, hipStream_t st, const cf32 *d_H, int c_first, int c_count);
// One call = ddc_mfma_begin_chains, then EITHER ddc_mfma_load_spectra (natural [n_blocks][fft] spectra, the wire format between fastddc_fwd_cc and
// fastddc_inv_cc) OR ddc_mfma_forward (new input samples; overlap-save framing + own 65536-point transform straight into the fold's layout), then
// ddc_mfma_process (fold + inverse transforms + scrap + residual shift).
int ddc_mfma_begin_chains(DdcMfma *m, hipStream_t st, int n_blocks, DdcChanState *d_state, const ChanGeom *d_geom, int *d_blk_remain, float *d_blk_phase, int *d_blk_off, int *d_counts);
int ddc_mfma_load_spectra(DdcMfma *m, hipStream_t st, const cf32 *spectra, int n_blocks);
bool ddc_mfma_can_forward(const DdcMfma *m);
int ddc_mfma_forward(DdcMfma *m, hipStream_t st, const cf32 *in, int n_blocks);
int ddc_mfma_process(DdcMfma *m, hipStream_t st, int n_blocks, const ChanGeom *d_geom, const int *d_blk_remain, const int *d_blk_off, cf32 *out, size_t out_pitch);
int ddc_mfma_set_profiling(DdcMfma *m, int on);
int ddc_mfma_kernel_time(DdcMfma *m, double *total_ms, long *launches);
// device pointer / pitches of the transposed spectrum buffer Xt[residue][block][q] (for the multi-GPU exchange)
cf32 *ddc_mfma_xt(DdcMfma *m, size_t *bytes, int *block_pitch);

} // namespace csdr_amd
