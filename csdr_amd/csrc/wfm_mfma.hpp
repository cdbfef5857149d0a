// wfm_mfma.hpp -- interface between wfm.hip (chain object, bookkeeping) and wfm_mfma.hip (matrix-core front end)
#pragma once
#include "common.hpp"
#include <vector>

namespace csdr_amd {

constexpr int WFM_HIST = 256;      // complex samples of input history kept per stream between blocks
constexpr int WFM_NK = 8;          // 64-byte K-steps per tile window (4 audio samples)
constexpr int WFM_FRAG_V4 = WFM_NK * 3 * 64;   // int8x16 vectors per weight set: [K-step][digit][lane]

struct WfmMfmaTable {              // host side
    int D, L, F, tile_stride_bytes, win_off_bytes, n_phases;
    float scale;
    std::vector<int8_t> frags;     // [n_sets][WFM_NK][3][64][16]; set index from set_of
    std::vector<float> consts;     // [n_phases][2][16]
    std::vector<int> set_of;       // [n_phases][2]: weight set of (phase, side of the chunk boundary); -1 = no second side
};

struct WfmMfmaDevice {             // device copies
    int tile_stride_bytes, win_off_bytes, n_phases;
    float scale;
    void *d_frags; float *d_consts; int *d_set_of;
};

bool wfm_mfma_supported(int D, int L, int F);
void wfm_mfma_build_table(int D, int L, int F, float shift_rate, const float *taps, WfmMfmaTable &t);
const char *wfm_mfma_last_kernel();
// st_edge: stream for the few bounds-checked tiles around the whole-quad range (may equal st); ev_begin/ev_end (may be null) are recorded on st
// around the dominant kernel only.
int wfm_mfma_launch(hipStream_t st, hipStream_t st_edge, hipEvent_t ev_begin, hipEvent_t ev_end, const uint8_t *in, size_t in_pitch, const uint8_t *hist, const WfmMfmaDevice &dev, const float2 *ctab,
                    float *demod, size_t demod_pitch, int n_streams, int T, long long B, long long j_first, int n_audio);

} // namespace csdr_amd
