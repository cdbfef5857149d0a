// wfm_mfma.hpp -- interface between wfm.hip (chain object, bookkeeping) and wfm_mfma.hip (matrix-core front end)
#pragma once
#include "common.hpp"
#include <vector>

namespace csdr_amd {

constexpr int WFM_HIST = 256;      // complex samples of input history kept per stream between blocks
constexpr int WFM_NK = 8;          // 64-byte K-steps per tile window (4 audio samples)
constexpr int WFM_NFRAG = WFM_NK * 3 + 3;      // weight fragments per tile phase: [K-step][digit] + the boundary K-step's side-0 part [digit]
constexpr int WFM_FRAG_V4 = WFM_NFRAG * 64;    // int8x16 vectors per tile phase: [fragment][lane]

struct WfmMfmaTable {              // host side
    int D, L, F, tile_stride_bytes, win_off_bytes, n_phases;
    float scale;
    std::vector<int8_t> frags;     // [n_phases][WFM_NFRAG][64][16]
    std::vector<float> consts;     // [n_phases][2][16]
    std::vector<int> kb_of;        // [n_phases]: K-step that contains the first sample of the NEXT 1024-chunk; WFM_NK = the window has one side only
    // phase-independent form for the sequential kernel (k_wfm_mfma_seq): weights a h D^t relative to the window base, post factors C_m D^e
    std::vector<int8_t> seq_frags; // [WFM_NK][3][64][16]
    std::vector<float> seq_cum;    // [2 * 4 * WFM_NK + 1][16]: 0.5 * sum of the weights of row r over bytes < 16 g
    std::vector<float2> dtab;      // D^(i - 2048), i in [0, 3072)
    float seq_scale;
};

struct WfmMfmaDevice {             // device copies
    int tile_stride_bytes, win_off_bytes, n_phases;
    float scale;
    void *d_frags; float *d_consts; int *d_kb_of;
    void *d_seq_frags; float *d_seq_cum; float2 *d_dtab; float seq_scale;
};

// back end (de-emphasis + convert_f_s16) the sequential kernel can take over; `done` is set when it did (the caller then skips k_wfm_back)
struct WfmBackArgs {
    float alpha; const float *last_in; float *last_out; float *seg_state; int16_t *s16; float *af; size_t out_pitch; int skip; bool done;
};

bool wfm_mfma_supported(int D, int L, int F);
void wfm_mfma_build_table(int D, int L, int F, float shift_rate, const float *taps, WfmMfmaTable &t);
const char *wfm_mfma_last_kernel();
// st_edge: stream for the few bounds-checked tiles around the whole-quad range (may equal st); ev_begin/ev_end (may be null) are recorded on st
// around the dominant kernel only.
int wfm_mfma_launch(hipStream_t st, hipStream_t st_edge, hipEvent_t ev_begin, hipEvent_t ev_end, const uint8_t *in, size_t in_pitch, const uint8_t *hist, const WfmMfmaDevice &dev, const float2 *ctab,
                    float *demod, size_t demod_pitch, int n_streams, int T, long long B, long long j_first, int n_audio, WfmBackArgs *back);

} // namespace csdr_amd
