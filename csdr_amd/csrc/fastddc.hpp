// fastddc.hpp -- pieces shared by the fastddc inverse of fftpath.hip (general geometry, hipFFT) and its matrix-core path fastddc_mfma.hip
// (config 4's geometry: fft_inv_size 512).
#pragma once
#include "common.hpp"

namespace csdr_amd {

struct ChanGeom { int offsetbin; float sindelta, cosdelta, rate2; };   // per channel: fastddc_t.offsetbin, dsadata (fastddc.h:5-24)
struct DdcChanState { int remain; float phase; };                     // decimating_shift_addition_status_t carried between blocks (libcsdr_gpl.h:39-44)

struct DdcMfma;   // device-side plan of the matrix-core path

// the communicator of a sharded bank as the channelizer sees it (comm.cpp: RCCL over xGMI, one process per GPU); counts in floats
struct DdcComm {
    int rank, world; void *impl;
    int (*group_start)(const DdcComm *c);
    int (*group_end)(const DdcComm *c);
    int (*send)(const DdcComm *c, const void *dev_buf, size_t n_floats, int peer, hipStream_t st);
    int (*recv)(const DdcComm *c, void *dev_buf, size_t n_floats, int peer, hipStream_t st);
    int (*all_gather)(const DdcComm *c, void *dev_buf_all, size_t n_floats_per_rank, hipStream_t st);      // in place: rank g's piece at offset g * n
};

// nullptr when the geometry is not the one this path implements (the caller keeps the general kernels).  comm: nullptr = one GPU.
DdcMfma *ddc_mfma_create(csdr_amd_ctx *ctx, int fft, int inv, int pre, int n_channels, int max_blocks, int scrap, int post_in, int post_dec, int input_size, int overlap,
                         const DdcComm *comm);
void ddc_mfma_destroy(DdcMfma *m);
int ddc_mfma_quiesce(DdcMfma *m);
// (re)build the kernel-side layout of the taps spectra of channels [c_first, c_first + c_count) from the natural [channel][fft] array
int ddc_mfma_set_taps(DdcMfma *m, hipStream_t st, const cf32 *d_H, int c_first, int c_count);
// One call = ddc_mfma_submit (chains + EITHER the natural [n_blocks][fft] spectra, the wire format between fastddc_fwd_cc and fastddc_inv_cc, OR new
// input samples: overlap-save framing + own 65536-point transform straight into the fold's layout, sharded by blocks and all-gathered when the bank
// spans several GPUs; all on a side stream) followed by ddc_mfma_collect (fold + inverse transforms + scrap + residual shift on the context's stream).
// Two calls may be staged: submit(N + 1) overlaps collect(N).
bool ddc_mfma_can_forward(const DdcMfma *m);
// in: complexf, or -- fmt -- s16 / u8 IQ pairs converted inside the forward transform (bit-equal to convert_s16_f / convert_u8_f in front of it)
enum { DDC_IN_CF32 = 0, DDC_IN_S16 = 1, DDC_IN_U8 = 2 };
static inline size_t ddc_in_bytes(int fmt) { return fmt == DDC_IN_S16 ? 4 : fmt == DDC_IN_U8 ? 2 : 8; }      // per complex sample
int ddc_mfma_submit(DdcMfma *m, const void *in, const cf32 *spectra, int n_blocks, DdcChanState *d_state, const ChanGeom *d_geom, bool inline_call, const cf32 *ext_tail = nullptr,
                    int fmt = DDC_IN_CF32, bool tail_in_front = false);
int ddc_mfma_convert_samples(hipStream_t st, const void *in, int fmt, long long first, int n, cf32 *out);      // samples [first, first + n) of `in` as complexf
// time-sliced bank (fftpath.hip): position of the next call's blocks inside a batch dealt to `world` ranks in runs of nbl blocks; the per-rank sample counts
// offsets [world + 1][n_channels] (device, written into the caller's buffers); a batch in which this rank has no blocks
int ddc_mfma_set_segment(DdcMfma *m, int nbl, int first, int total, int world, int *pref_cur, int *pref_next);
int ddc_mfma_pending_blocks(const DdcMfma *m);      // block count of the call collect() would fold next (0: nothing staged)
int ddc_mfma_skip_batch(DdcMfma *m, DdcChanState *d_state, const ChanGeom *d_geom);
int ddc_mfma_collect(DdcMfma *m, const ChanGeom *d_geom, cf32 *out, size_t out_pitch, const int **d_counts, hipEvent_t after_inverse = nullptr);      // after_inverse: recorded when the call's last kernel completes
int ddc_mfma_set_profiling(DdcMfma *m, int on);
int ddc_mfma_kernel_time(DdcMfma *m, double *total_ms, long *launches);
int ddc_mfma_stage_time(DdcMfma *m, int stage, double *total_ms, long *launches);
const char *ddc_mfma_kernel_name(const DdcMfma *m);       // the fold kernel the last collect() launched

} // namespace csdr_amd
