// wfm_mfma.hpp -- interface between wfm.hip (chain object, bookkeeping) and wfm_mfma.hip (matrix-core chain kernel)
#pragma once
#include "common.hpp"
#include <vector>

namespace csdr_amd {

constexpr int WFM_HIST = 256;      // complex samples of input history kept per stream between blocks
constexpr int WFM_NK = 8;          // 64-byte K-steps per tile window (4 audio samples)

struct WfmMfmaTable {              // host side: the phase-independent weight set (weights a h D^t relative to a tile's window base, post factors C_m D^e)
    int D, L, F, tile_stride_bytes, win_off_bytes;
    std::vector<int8_t> seq_frags; // [WFM_NK][3][64][16]
    std::vector<float> seq_cum;    // [4 * WFM_NK + 1][16]: 0.5 * sum of the weights of row r over bytes < 16 g
    std::vector<float2> dtab;      // D^(i - 2048), i in [0, 3072)
    float seq_scale;
};

struct WfmMfmaDevice {             // device copies
    int tile_stride_bytes, win_off_bytes;
    void *d_seq_frags; float *d_seq_cum; float2 *d_dtab; float seq_scale;
};

// what the chain kernel carries between calls (de-emphasis state per stream; the 1-KiB head per stream whose second half is the previous block's newest
// 512 bytes) and where its audio goes
struct WfmBackArgs {
    float alpha; const float *last_in; float *last_out; int16_t *s16; float *af; size_t out_pitch; const uint8_t *head_in; uint8_t *head_out;
};

// a shift rate per stream (csdr_amd_wfm_create_rates): dev.d_seq_frags / d_seq_cum / d_dtab hold one table set per stream, the chunk seeds come from a seed table
// (seeds.hpp: ctab[k * tab_pitch + stream]); lead: the first audio samples of retuned streams, evaluated by wfm_mfma_lead
struct WfmPerStream { size_t tab_pitch; int tab_len; const float *d_scales; const float *d_lead_d; const int *d_lead_n; int lead_stride; };
// audio samples whose windows can straddle a retune: a window spans D + L - 1 samples, audio samples lie D F apart
inline int wfm_lead_max(int D, int L, int F) { return (D + L - 1) / (D * F) + 2; }

// the resident form (csdr_amd_wfm_ring_*, wfm_ring.hip): what the persistent grid of k_wfm_mfma_seq<false, true> walks
struct WfmResident {
    const uint32_t *desc, *ctrl; uint32_t *done;                      // host-coherent: descriptor lines, stop word, done lines
    unsigned *cnt; unsigned long long *t_first, *next_item; unsigned *exiting;      // device
    const uint8_t *in_ring; int16_t *out_ring; size_t in_slot_bytes, out_slot_elems;
    int n_slots, desc_lines, T, D, L, F;
    long long idle_ticks, life_ticks;
    const float *lead_d, *lead_state; int lead_stride;
    unsigned long long *stats; int fence_mode;
};
constexpr int WFM_RES_WARM = 48;   // = RES_WARM (wfm_mfma.hip)
int wfm_mfma_launch_resident(hipStream_t st, hipEvent_t ev_end, const WfmMfmaDevice &dev, int n_streams, size_t in_pitch, float alpha, size_t out_pitch, const WfmResident &rv, int grid);
int wfm_resident_max_grid();
int wfm_mfma_lead_shared(hipStream_t st, const uint8_t *in, size_t in_pitch, const uint8_t *prev, size_t two_T, const float *d_taps, const float2 *ctab, const float2 *d_dtab,
                         const float2 *d_dtab_old, const int *d_list, int n_list, float *d_lead_d, int lead_stride, float *d_warm, float *d_state, float alpha,
                         int D, int L, int F, long long B, long long j_first, int n_lead);

bool wfm_mfma_supported(int D, int L, int F);
void wfm_mfma_build_table(int D, int L, int F, float shift_rate, const float *taps, WfmMfmaTable &t);
size_t wfm_mfma_head_bytes(int n_streams);
// One launch per call: audio j_first .. j_first + n_audio - 1 of every stream from the block in[stream * in_pitch + 2 * (0 .. T)) (global sample index B at its
// start) and the head; ev_begin / ev_end (may be null) are recorded around the kernel.  n_audio = 0: only the head rolls.
int wfm_mfma_launch(hipStream_t st, hipEvent_t ev_begin, hipEvent_t ev_end, const uint8_t *in, size_t in_pitch, const WfmMfmaDevice &dev, const float2 *ctab,
                    int n_streams, int T, long long B, long long j_first, int n_audio, const WfmBackArgs &back, const WfmPerStream *ps = nullptr);
int wfm_mfma_lead(hipStream_t st, const uint8_t *in, size_t in_pitch, const uint8_t *head, const float *d_taps, const float2 *ctab, size_t tab_pitch, const float2 *d_dtab,
                  const float2 *d_dtab_old, const int *d_list, int n_list, float *d_lead_d, int lead_stride, int D, int L, int F, long long B, long long j_first, int n_lead);

} // namespace csdr_amd
