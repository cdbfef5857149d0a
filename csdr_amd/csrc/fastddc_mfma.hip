// fastddc_mfma.hip -- the fastddc inverse (fastddc.c:106-166 x N channels, csdr.c:2302-2378) at BASELINE config 4's geometry
// (fft_size 65536, fft_inv_size 512, pre_decimation 128, 256 channels, tens of blocks per call) on the fp32 matrix cores.
//
// Per bin residue r = i mod fft_inv_size the alias fold of all channels and blocks is ONE complex matrix product
//     C_r[channel][block] = sum_q  H[channel][r + q inv] * Xs[block][r + q inv]          q = 0 .. pre-1      (fastddc.c:126-141, Xs = swapped spectrum)
// i.e. per residue a (channels x pre) . (pre x blocks) contraction: 8 flop per (bin, channel, block) = 8.6 GFLOP per 64-block call against
// 128 MiB of taps spectra: 51 flop per byte, above the fp32 ridge (157 TF / 8 TB/s = 20): COMPUTE bound on v_mfma_f32_32x32x2_f32 (exact fp32
// fma chains, MI355X_MICROARCH.md: 64 cycles per instruction per SIMD = the fp32 vector rate, but without the VALU's operand traffic).
// The general kernel k_ddc_fold_ct (fftpath.hip) re-read every taps value 16 times per call from L2 / Infinity Cache and ran at 13 % of that peak.
//
// Layouts (all private to this path; built once per retune / once per call):
//   Ht[r][g][channel][8]   the taps spectra, 8 floats = (re, im) of q = 4g .. 4g+3: a wave's A operand for one k-group of all its 32 channels
//                          is ONE contiguous KiB, fetched exactly once per call straight into registers (no LDS), prefetched 4 groups ahead;
//   Xt[r][block][q]        the spectra with the first fft_swap_sides folded into q (q' = (q + pre/2) mod pre): 64 KiB contiguous per residue,
//                          staged once per workgroup in LDS (row pitch + 16 B: the 16 lanes of a ds_read_b128 group hit 16 different bank quads);
//   Ct[m][channel][block]  the folded bins, m = (r - offsetbin_channel) mod inv = the IFFT input bin after the second fft_swap_sides
//                          (fastddc.c:129, 150): 32 blocks x 8 B = 256-byte runs per store instruction.
// Complex product on real MFMAs: rows = channels, k' = (q, re/im of H) = the natural interleaved order of H, columns = (block, re/im of C):
//   C_re = [H_re | H_im] . [X_re ; -X_im],  C_im = [H_re | H_im] . [X_im ; X_re]: the SAME A registers feed both, B is the loaded complex value with
//   a swap / sign flip.  A lane's k pair inside a group is (float j of its half, j = 0..3): lanes 0-31 hold q = 4g, 4g+1, lanes 32-63 q = 4g+2, 4g+3,
//   so one 16-byte A load and one 16-byte B read feed 4 MFMA steps x 2 output tiles (the summation order over q differs from the reference's
//   sequential one: fp32 rounding noise, well inside the 1e-5 gate).
// Then, per (channel, 16 blocks): 512-point inverse transforms in LDS (radix 8 x 8 x 8, one wave per transform, in place), scrap,
// decimating_shift_addition_cc with the reference's float32 phasor recurrence REPLAYED (k_ddc_rot: one lane per (channel, block) chain, data
// independent) -- the folded bins are read once, the output written once; no hipFFT plan, no [channel][block][inv] round trips.
#include "fastddc.hpp"
#include "convert_dev.hpp"
#include <hip/hip_ext.h>
#include "fft_butterflies.hpp"
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace csdr_amd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct DdcMfma {
    csdr_amd_ctx *ctx;
    int fft, inv, pre, G, C, Cpad, nbp, scrap, post_in, post_dec, kmax, rpitch, max_blocks;
    float *d_Ht; cf32 *d_Ct; float2 *d_tw;
    // sharding (a bank over several GPUs): this rank's channels [C] of all, the forward transform split by blocks (nbl per rank), Xt = one chunk per rank
    int rank, world, nbl; const DdcComm *comm; cf32 *d_in_local;
    // time-sliced bank (fftpath.hip: the bank deals the BLOCKS of a batch to the ranks, every rank runs this object unsharded on its run): where the next call's
    // blocks sit in the batch (ddc_mfma_set_segment), and per set the samples every rank's run produces [seg_world][C]
    int seg_nbl = 0, seg_first = 0, seg_total = 0, seg_world = 0, spec_seg_first = 0, spec_seg_total = 0; int *seg_cur = nullptr, *seg_next = nullptr;
    // Two sets of everything a call produces before the fold (transposed spectra, chain tables, phasor checkpoints): submit() fills one set on the side
    // stream -- exchange + forward transform + chains -- while collect() folds the other on the context's stream.
    cf32 *d_Xt[2]; float2 *d_R[2]; int *d_blk_remain[2], *d_blk_off[2], *d_counts[2]; float *d_blk_phase[2];
    int pending_blocks[2]; int fill, drain;                            // set being filled next / folded next
    bool inline_set[2], chains_on_side[2];
    bool gemm_three = false, gemm_narrow = false;
    bool y_holds[2] = {false, false};                                  // set k's spectra are still pass-1 output in d_Y (the fold runs pass 2 itself)
    // the NEXT call's chain tables, computed one call ahead by riders of the inverse-transform kernel (data independent; valid for process() calls of equal size
    // with no retune in between).  On the side stream beside the fold they cost more than they hid: 0.182 vs 0.167 ms per step.
    DdcChanState *d_state_spec = nullptr, *last_state = nullptr; bool spec_valid = false, ahead_ok = false; int spec_set = 0, spec_blocks = 0;                                           // the fold kernel of the last collect(): k_ddc_gemm3 (three real products) or k_ddc_gemm
    hipStream_t side; hipEvent_t ev_ready[2], ev_free[2], ev_fork; bool free_recorded[2];
    // fused forward transform (65536 = 512 x 128): intermediate Y[block][k1][n2], the kept overlap tail of the input stream, W_65536^lo table
    cf32 *d_Y, *d_tail[2]; float2 *d_twb; int flip, input_size, overlap;
    // A/B switches, read ONCE when the object is created: a call never looks at the environment
    struct Opt { bool fwd_off, riders_off, spec_off, pass2_own; } opt;      // test hooks (tests/test_configs_gpu.py::test_c4_bank_alternative_paths): each turns one default choice off, so that the
                                                                            // path other conditions select (a sharded bank, a size change, > 4096 channels) is crossed on one GPU
    // HIP-event timing of the fold kernel on the context's stream (bench_fastddc.py's roofline leg)
    bool profiling = false; size_t ev_used = 0; double prof_ms = 0; long prof_launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    // the same for the forward transform's first pass (stage 1: k_ddc_fwd512) and the inverse transforms (stage 2: k_ddc_ifft256d_post / k_ddc_ifft512_post)
    bool profile_stages = false;
    struct StageProf { size_t used = 0; double ms = 0; long launches = 0; std::vector<std::pair<hipEvent_t, hipEvent_t>> pool; } stage[2];
};

static int ddc_stage_begin(DdcMfma *m, int s, hipStream_t st, hipEvent_t *e1)
{
    *e1 = nullptr;
    if (!m->profiling || !m->profile_stages) return 0;
    DdcMfma::StageProf &p = m->stage[s];
    if (p.used == p.pool.size()) { hipEvent_t a, b; CSDR_HIP(hipEventCreate(&a)); CSDR_HIP(hipEventCreate(&b)); p.pool.emplace_back(a, b); }
    CSDR_HIP(hipEventRecord(p.pool[p.used].first, st)); *e1 = p.pool[p.used].second; p.used++;
    return 0;
}

namespace {

// ------------------------------------------------------------------ layout builders
// Ht[((r G + q/4) Cpad + c) 8 + 2 (q%4)] = H[c][r + q inv]
__global__ __launch_bounds__(256) void k_ddc_ht(const float2 *__restrict__ H, float *__restrict__ Ht, int fft, int inv, int G, int Cpad, int c_first)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= inv) return;
    const int q = blockIdx.y, c = c_first + blockIdx.z;
    const float2 v = H[(size_t)c * fft + r + (size_t)q * inv];
    *reinterpret_cast<float2 *>(Ht + (((size_t)r * G + (q >> 2)) * Cpad + c) * 8 + 2 * (q & 3)) = v;
}

// Xt[(r nbp + b) pre + q] = X[b][r + q' inv],  q' = (q + pre/2) mod pre     (32 x 32 tiles through LDS: both sides move 256-byte runs)
__global__ __launch_bounds__(256) void k_ddc_xt(const float2 *__restrict__ X, float2 *__restrict__ Xt, int fft, int inv, int pre, int nbp)
{
    __shared__ float2 tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.x * 32, qp0 = blockIdx.y * 32; const size_t b = blockIdx.z;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int qp = qp0 + ty + 8 * j, r = r0 + tx;
        if (qp < pre && r < inv) tile[ty + 8 * j][tx] = X[b * fft + (size_t)qp * inv + r];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int r = r0 + ty + 8 * j, qp = qp0 + tx;
        if (qp < pre && r < inv) Xt[((size_t)r * nbp + b) * pre + ((qp - pre / 2) & (pre - 1))] = tile[tx][ty + 8 * j];
    }
}


// ------------------------------------------------------------------ the fold as a matrix product
// grid (residue slots, ceil(Cpad / 256) channel groups, ceil(n_blocks / (32 NBT)) block groups); 512 threads: wave w owns channels 32 w .. 32 w + 31 of
// the group, all 32 NBT blocks of the group and both output parts: 2 NBT accumulator tiles of 32 x 32.
// A workgroup walks the residues r = blockIdx.x, blockIdx.x + gridDim.x, ...  One residue per workgroup (gridDim.x = inv) is the simple form: but then
// all workgroups of the launch load their spectra, multiply, and store their bins in lockstep, and the matrix cores idle during the first and the last phase
// (64 KiB in and 128 KiB out per workgroup).  PERSIST (gridDim.x = one workgroup per CU, two LDS buffers): the next residue's spectra are fetched into
// registers while the current one is multiplied and go to the other buffer afterwards; the bins' stores drain under the next residue's product.
template <int NBT, bool PERSIST>
__global__ __launch_bounds__(512, PERSIST ? 2 : 4) void k_ddc_gemm(const float *__restrict__ Ht, const float2 *__restrict__ Xt, float2 *__restrict__ Ct,
                                                  const ChanGeom *__restrict__ geom, int inv, int pre, int Cpad, int n_channels, int nbp, int nbl, int n_blocks, float scale)
{
    extern __shared__ float4 xs_all[];                              // (PERSIST ? 2 : 1) x [32 NBT rows][pre / 2 + 1] float4
    constexpr int NX = 4 * NBT;                                     // float4 per thread of one residue's spectra (PERSIST: pre <= 128)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c_base = blockIdx.y * 256 + wave * 32, b_base = blockIdx.z * 32 * NBT;
    const int G = pre >> 2, P4 = (pre >> 1) + 1, rows = 32 * NBT, row4 = pre >> 1, BUF = rows * P4;
    const int i = lane & 31, hi = lane >> 5;
    const size_t gstride = (size_t)Cpad * 2;                          // float4 per k-group
    const bool active = c_base < Cpad;
    // Xt is kept in chunks of nbl blocks (one chunk per rank of a sharded bank: the all-gather's receive layout; a single chunk otherwise):
    // row (r, b) = Xt[((b / nbl) inv + r) nbl + b % nbl][0 .. pre)
    auto xt_row = [&](int r, int b) { return reinterpret_cast<const float4 *>(Xt + (((size_t)(b / nbl) * inv + r) * nbl + (b % nbl)) * pre); };
    {   // stage the first residue's spectra (pre complex = pre / 2 float4 per block row)
        for (int idx = threadIdx.x; idx < rows * row4; idx += 512) {
            const int row = idx / row4, col = idx - row * row4, b = b_base + row;
            xs_all[row * P4 + col] = (b < n_blocks) ? xt_row(blockIdx.x, b)[col] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();
    int cur = 0;
    for (int r = blockIdx.x; r < inv; r += gridDim.x) {
        const int rn = r + gridDim.x;
        float4 nx[NX];
        if (PERSIST && rn < inv) {
#pragma unroll
            for (int k = 0; k < NX; k++) {
                const int idx = threadIdx.x + 512 * k, row = idx / row4, col = idx - row * row4, b = b_base + row;
                nx[k] = (idx < rows * row4 && b < n_blocks) ? xt_row(rn, b)[col] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if (active) {
            const float4 *xs = xs_all + cur * BUF;
            const float4 *ap = reinterpret_cast<const float4 *>(Ht) + ((size_t)r * G * Cpad + c_base + i) * 2 + hi;
            f32x16 acc[NBT][2];
#pragma unroll
            for (int bt = 0; bt < NBT; bt++)
#pragma unroll
                for (int p = 0; p < 2; p++)
#pragma unroll
                    for (int e = 0; e < 16; e++) acc[bt][p][e] = 0.f;
            // four k-groups in flight (indices clamped: a short K loop re-loads its last group instead of branching)
            float4 a0 = ap[0], a1 = ap[(size_t)min(1, G - 1) * gstride], a2 = ap[(size_t)min(2, G - 1) * gstride], a3 = ap[(size_t)min(3, G - 1) * gstride];
            const float4 *xrow = xs + i * P4 + hi;
#define DDC_STEP(AV, GG)                                                                                                   \
            {                                                                                                              \
                _Pragma("unroll") for (int bt = 0; bt < NBT; bt++) {                                                       \
                    const float4 xv = xrow[bt * 32 * P4 + 2 * (GG)];                                                       \
                    acc[bt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).x, xv.x, acc[bt][0], 0, 0, 0);                  \
                    acc[bt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).x, xv.y, acc[bt][1], 0, 0, 0);                  \
                    acc[bt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).y, -xv.y, acc[bt][0], 0, 0, 0);                 \
                    acc[bt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).y, xv.x, acc[bt][1], 0, 0, 0);                  \
                    acc[bt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).z, xv.z, acc[bt][0], 0, 0, 0);                  \
                    acc[bt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).z, xv.w, acc[bt][1], 0, 0, 0);                  \
                    acc[bt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).w, -xv.w, acc[bt][0], 0, 0, 0);                 \
                    acc[bt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).w, xv.z, acc[bt][1], 0, 0, 0);                  \
                }                                                                                                          \
            }
            int g = 0;
            for (; g + 4 <= G; g += 4) {                              // straight-line body: the loads stay four groups ahead of their use
                { const float4 av = a0; a0 = ap[(size_t)min(g + 4, G - 1) * gstride]; DDC_STEP(av, g); }
                { const float4 av = a1; a1 = ap[(size_t)min(g + 5, G - 1) * gstride]; DDC_STEP(av, g + 1); }
                { const float4 av = a2; a2 = ap[(size_t)min(g + 6, G - 1) * gstride]; DDC_STEP(av, g + 2); }
                { const float4 av = a3; a3 = ap[(size_t)min(g + 7, G - 1) * gstride]; DDC_STEP(av, g + 3); }
            }
            if (g < G) { DDC_STEP(a0, g); }                           // pre_decimation = 8 ... (G not a multiple of 4)
            if (g + 1 < G) { DDC_STEP(a1, g + 1); }
            if (g + 2 < G) { DDC_STEP(a2, g + 2); }
#undef DDC_STEP
            // C / D layout of the 32 x 32 tile: column = lane & 31 (block), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (channel)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int c = c_base + (e & 3) + 8 * (e >> 2) + 4 * hi;
                if (c >= n_channels) continue;
                int m = (r - geom[c].offsetbin) % inv; if (m < 0) m += inv;
                float2 *dst = Ct + ((size_t)m * Cpad + c) * nbp + b_base + i;
#pragma unroll
                for (int bt = 0; bt < NBT; bt++)
                    if (b_base + bt * 32 + i < n_blocks) dst[bt * 32] = make_float2(acc[bt][0][e] * scale, acc[bt][1][e] * scale);
            }
        }
        if (PERSIST && rn < inv) {
            float4 *nxt = xs_all + (cur ^ 1) * BUF;
#pragma unroll
            for (int k = 0; k < NX; k++) {
                const int idx = threadIdx.x + 512 * k, row = idx / row4, col = idx - row * row4;
                if (idx < rows * row4) nxt[row * P4 + col] = nx[k];
            }
            // the other buffer is complete; nobody reads this one any more.  An LDS-only barrier: __syncthreads() also waits for vmcnt(0), i.e. for this
            // residue's bin stores to land, which are meant to drain under the next residue's product
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            cur ^= 1;
        }
    }
}

// ------------------------------------------------------------------ the fold with three real products per complex one (pre_decimation 128)
// k_ddc_gemm3: the persistent fold above with two changes.
//  (1) Gauss's three-multiplication form:  P1 = sum Hre Xre,  P2 = sum Him Xim,  P3 = sum (Hre + Him)(Xre + Xim);  C = (P1 - P2) + j (P3 - P1 - P2).
//      Six matrix instructions per k-group and block tile instead of eight (the sums cost one VALU add each and replace the sign flips); three accumulator
//      tiles instead of two.  The kernel is bound by the fp32 matrix pipe (SQ_VALU_MFMA_BUSY_CYCLES = 60 % of the kernel with the four-product form), so
//      the product shrinks by a quarter.  Rounding: C_im carries the rounding of three sums of comparable size instead of one -- a few 1e-7, gate 1e-5.
//  (2) The next residue's spectra go from HBM straight into the other LDS buffer (global_load_lds_dwordx4: a row = 128 complex = 1 KiB = one wave instruction),
//      which frees the 32 staging registers the third accumulator tile needs.  Issued after the first k-groups so that the taps fetches in front of it are
//      not queued behind it (vector memory returns in order); the barrier at the end of a residue waits for this wave's own pieces and sits BEFORE the bin
//      stores, which therefore drain under the next residue's product together with its first taps fetches.
//  FWD (one GPU, process()): the second pass of the forward transform happens HERE -- the workgroup reads the residue's 64 rows of pass-1 output Y[block][r][0..127]
//      (1 KiB each), runs the 128-point transforms (16 x 8, eight lanes per row, exchange inside the row's own LDS space) and leaves the result where the staging
//      would have put it: k_ddc_fwd128, its 34 MB round trip and one kernel boundary are gone; the butterflies run on the vector ALUs beside the matrix pipe.
template <int NBT, bool FWD>
__global__ __launch_bounds__(512, 2) void k_ddc_gemm3(const float *__restrict__ Ht, const float2 *__restrict__ Xt, float2 *__restrict__ Ct,
                                                       const ChanGeom *__restrict__ geom, int inv, int Cpad, int n_channels, int nbp, int nbl, int n_blocks, float scale,
                                                       const float2 *__restrict__ g_tw)
{
    extern __shared__ float4 xs_all[];                              // 2 x [32 NBT rows][65] float4
    constexpr int PRE = 128, G = PRE / 4, P4 = PRE / 2 + 1, ROWS = 32 * NBT, BUF = ROWS * P4, RPW = ROWS / 8;      // rows per wave
    constexpr int AD = FWD ? 4 : 8;                                     // taps fetches in flight per wave, in k-groups (FWD: 8 spills)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int c_base = blockIdx.y * 256 + wave * 32, b_base = blockIdx.z * ROWS;
    const int i = lane & 31, hi = lane >> 5;
    const size_t gstride = (size_t)Cpad * 2;                          // float4 per k-group
    const bool active = c_base < Cpad;
    const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t *)xs_all;
    // rows wave * RPW .. + RPW - 1 of residue r into buffer `buf`: one 1-KiB piece per row (blocks past the end re-read the last one: their columns are never stored)
    auto dma_rows = [&](int r, int buf) {
#pragma unroll
        for (int k = 0; k < RPW; k++) {
            const int row = wave * RPW + k, b = min(b_base + row, n_blocks - 1);
            const float2 *src = (FWD ? Xt + ((size_t)b * 512 + r) * PRE                                            // pass-1 output Y[block][r][0..127]
                                     : Xt + (((size_t)(b / nbl) * inv + r) * nbl + (b % nbl)) * PRE) + 2 * lane;      // 16 bytes per lane
            const uint32_t la = __builtin_amdgcn_readfirstlane((int)(lds_base + ((uint32_t)buf * BUF + (uint32_t)row * P4) * 16u));
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(la) : "memory");
        }
    };
    // Taps fetches by hand: as plain loads the compiler sinks each one to just in front of its use (to shorten live ranges) and waits vmcnt(0) right behind it --
    // the prefetch distance is gone and every k-group pays a memory round trip.  taps_fetch issues, taps_ready waits until at most AD - 1 younger operations are out
    // (the fetches of the following groups, issued in order; anything else that is younger only makes the wait conservative) and ties the value to the wait.
    typedef float g3_v4f __attribute__((ext_vector_type(4)));
    auto taps_fetch = [&](g3_v4f &dst, const float4 *p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory"); };
#define G3_TAPS_READY(V, K) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(V) : "n"(K))
    // A value in flight must never meet a merge of definitions (a branch around a fetch, the residue loop's back edge): the compiler, which takes the asm's output for
    // ready, may copy the register there.  Hence: every wave fetches (inactive ones from a clamped channel base), and a residue's fetches start at the top of its
    // own iteration -- their latency is covered by the PREVIOUS residue's bin stores, which are issued right behind them.
    // FWD: thread (tr, c) = lane c of the 128-point transform of row tr (rows past the tile's NBT * 32: idle)
    const int tr = wave * RPW + (lane >> 3), fc = lane & 7;           // a wave transforms exactly the rows it staged (RPW = 8: all lanes; 4: the lower half)
    const bool frow = (lane >> 3) < RPW;
    auto fft_rows = [&](int buf) {                                    // 128 = 16 (a) x 8 (c), as k_ddc_fwd128, IN PLACE in the row's 1040 bytes of LDS.  Row tr was staged by
        if (!frow) return;                                            // this very wave (rows 8 w .. 8 w + 7), so no workgroup barrier is needed in front -- only its own pieces
        float2 *rowp = reinterpret_cast<float2 *>(xs_all + (size_t)buf * BUF + (size_t)tr * P4);
        float2 v[16];
#pragma unroll
        for (int a16 = 0; a16 < 16; a16++) v[a16] = rowp[8 * a16 + fc];
        dft16<false>(v);
        __builtin_amdgcn_wave_barrier();                              // (the eight lanes of a row sit in one wave: every lane has read before anyone writes; LDS operations of a wave complete in order)
#pragma unroll
        for (int ka = 0; ka < 16; ka++) rowp[fc * 16 + ka] = cmul(v[ka], g_tw[(4 * fc * ka) & 511]);
        __builtin_amdgcn_wave_barrier();
        float2 e[8], o[8];
#pragma unroll
        for (int cc = 0; cc < 8; cc++) {
            const float4 two = *reinterpret_cast<const float4 *>(&rowp[cc * 16 + 2 * fc]);
            e[cc] = make_float2(two.x, two.y); o[cc] = make_float2(two.z, two.w);
        }
        dft8<false>(e); dft8<false>(o);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int kc = 0; kc < 8; kc++) {
            const int q = (2 * fc + 16 * kc + 64) & 127;               // q' = ka + 16 kc with ka = 2 c (and 2 c + 1); q = (q' - pre/2) mod pre
            *reinterpret_cast<float4 *>(&rowp[q]) = make_float4(e[kc].x, e[kc].y, o[kc].x, o[kc].y);
        }
    };
    int r = blockIdx.x;
    if (r >= inv) return;
    dma_rows(r, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // this wave's pieces of the first buffer
    const int c_eff = min(c_base, Cpad - 32);                         // channel base the taps are fetched from (waves past the last channel tile compute a copy nobody stores)
    f32x16 acc[NBT][3];
    // C / D layout of the 32 x 32 tile: column = lane & 31 (block), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (channel)
    auto store_bins = [&](int rr) {
        if (!active) return;
        int mm[16];                                                   // all sixteen offsetbin fetches in flight at once (inside the store loop each one was waited for on its own)
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int c = min(c_base + (e & 3) + 8 * (e >> 2) + 4 * hi, n_channels - 1);
            mm[e] = (rr - geom[c].offsetbin) & (inv - 1);             // inv is 512 here (a power of two): no division on the store path
        }
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int c = c_base + (e & 3) + 8 * (e >> 2) + 4 * hi;
            if (c >= n_channels) continue;
            float2 *dst = Ct + ((size_t)mm[e] * Cpad + c) * nbp + b_base + i;
#pragma unroll
            for (int bt = 0; bt < NBT; bt++)
                if (b_base + bt * 32 + i < n_blocks) {
                    const float p1 = acc[bt][0][e], p2 = acc[bt][1][e], p3 = acc[bt][2][e];
                    dst[bt * 32] = make_float2((p1 - p2) * scale, (p3 - p1 - p2) * scale);      // (streaming `nt` stores here and in the forward passes: the consumers got 3-5 us slower each)
                }
        }
    };
    int cur = 0, r_prev = -1;
    for (;;) {
        const int rn = r + gridDim.x;
        const float4 *ap = reinterpret_cast<const float4 *>(Ht) + ((size_t)r * G * Cpad + c_eff + i) * 2 + hi;
        g3_v4f a[AD];
#pragma unroll
        for (int d = 0; d < AD; d++) taps_fetch(a[d], ap + (size_t)d * gstride);
        if (r_prev >= 0) store_bins(r_prev);                          // the previous residue's bins: their stores cover the first fetches' latency and drain under this product
        // FWD: each wave transforms the rows it staged (landed: they are older than the previous product's last taps fetches, all consumed), with the accumulators
        // dead -- inside the product the transform's registers pushed the kernel into spills.  ONE barrier per residue: every wave has left the previous product
        // (the other buffer may be overwritten) and this buffer is complete.
        if (FWD) fft_rows(cur);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#pragma unroll
        for (int bt = 0; bt < NBT; bt++)
#pragma unroll
            for (int p = 0; p < 3; p++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[bt][p][e] = 0.f;
        const float4 *xrow = xs_all + cur * BUF + i * P4 + hi;
        // the spectra operand of k-group GG + 1 is read from LDS while group GG multiplies
        float4 xn[NBT];
#pragma unroll
        for (int bt = 0; bt < NBT; bt++) xn[bt] = xrow[bt * 32 * P4];
#define DDC_STEP3(AV, GG)                                                                                                  \
        {                                                                                                                  \
            const float hs0 = (AV).x + (AV).y, hs1 = (AV).z + (AV).w;                                                      \
            float4 xc[NBT];                                                                                                \
            _Pragma("unroll") for (int bt = 0; bt < NBT; bt++) { xc[bt] = xn[bt]; xn[bt] = xrow[bt * 32 * P4 + 2 * min((GG) + 1, G - 1)]; }  \
            _Pragma("unroll") for (int bt = 0; bt < NBT; bt++) {                                                           \
                const float4 xv = xc[bt];                                                                                  \
                const float xs0 = xv.x + xv.y, xs1 = xv.z + xv.w;                                                          \
                acc[bt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).x, xv.x, acc[bt][0], 0, 0, 0);                      \
                acc[bt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).y, xv.y, acc[bt][1], 0, 0, 0);                      \
                acc[bt][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(hs0, xs0, acc[bt][2], 0, 0, 0);                          \
                acc[bt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).z, xv.z, acc[bt][0], 0, 0, 0);                      \
                acc[bt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).w, xv.w, acc[bt][1], 0, 0, 0);                      \
                acc[bt][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(hs1, xs1, acc[bt][2], 0, 0, 0);                          \
            }                                                                                                              \
        }
        static_assert(G % AD == 0 && G >= 2 * AD, "k loop");
        // straight-line (fully unrolled): across a loop back edge the compiler's wait-count bookkeeping gives up.  A fetch is ready when at most AD - 1 younger
        // operations are out (the fetches of the following groups, issued in order; anything else that is younger only makes the wait conservative)
#pragma unroll
        for (int d = 0; d < AD; d++) { G3_TAPS_READY(a[d], AD - 1); const g3_v4f av = a[d]; taps_fetch(a[d], ap + (size_t)(AD + d) * gstride); DDC_STEP3(av, d); }
        if (rn < inv) dma_rows(rn, cur ^ 1);                          // behind the first 2 AD taps fetches
#pragma unroll
        for (int g = AD; g < G - AD; g += AD) {
#pragma unroll
            for (int d = 0; d < AD; d++) { G3_TAPS_READY(a[d], AD - 1); const g3_v4f av = a[d]; taps_fetch(a[d], ap + (size_t)(g + AD + d) * gstride); DDC_STEP3(av, g + d); }
        }
#define G3_LAST(D) { G3_TAPS_READY(a[D], AD - 1 - (D)); const g3_v4f av = a[D]; DDC_STEP3(av, G - AD + (D)); }      /* the last AD groups: nothing new is fetched */
        G3_LAST(0) G3_LAST(1) G3_LAST(2) G3_LAST(3)
        if constexpr (AD == 8) { G3_LAST(4) G3_LAST(5) G3_LAST(6) G3_LAST(7) }
#undef G3_LAST
#undef DDC_STEP3
        r_prev = r;
        if (rn >= inv) break;
        r = rn; cur ^= 1;
    }
    store_bins(r_prev);
}

// k_ddc_gemm3n: the same product for FEW channel rows -- a bank of up to 128 channels, or one rank's slice of a channel-sharded bank (32 of 256 at eight ranks).
// k_ddc_gemm3 gives each of its eight waves 32 channels and lets all of them share the staged spectra: with 32 channels seven waves multiply a copy nobody stores, and
// the rank's fold costs what the whole bank's does (75 us; the emulated channel-shard scaling of 1.3 x at eight ranks).  Here a workgroup is NW = channels / 32 waves
// (1, 2 or 4) x ONE residue x 32 blocks: 33 KiB of LDS, no second buffer, nothing persistent -- four workgroups per CU overlap one another's staging, products and
// bin stores, and 512 residues x (blocks / 32) workgroups fill the chip by themselves.  Operand layouts, the three-product form, the taps fetches by hand and the bin
// stores are k_ddc_gemm3's (non-FWD: the spectra are complete, the forward transform's second pass runs as its own kernel).
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_ddc_gemm3n(const float *__restrict__ Ht, const float2 *__restrict__ Xt, float2 *__restrict__ Ct,
                                                        const ChanGeom *__restrict__ geom, int inv, int Cpad, int n_channels, int nbp, int nbl, int n_blocks, float scale)
{
    extern __shared__ float4 xs_all[];                              // [32 rows][65] float4
    constexpr int PRE = 128, G = PRE / 4, P4 = PRE / 2 + 1, ROWS = 32, RPW = ROWS / NW, AD = 8;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int r = blockIdx.x;
    const int c_base = ((int)blockIdx.y * NW + wave) * 32, b_base = blockIdx.z * ROWS;
    const int i = lane & 31, hi = lane >> 5;
    const size_t gstride = (size_t)Cpad * 2;                          // float4 per k-group
    const bool active = c_base < Cpad;
    const int c_eff = min(c_base, Cpad - 32);                         // (a wave past the last channel tile multiplies a copy nobody stores)
    const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t *)xs_all;
    typedef float g3_v4f __attribute__((ext_vector_type(4)));
    auto taps_fetch = [&](g3_v4f &dst, const float4 *p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory"); };
    const float4 *ap = reinterpret_cast<const float4 *>(Ht) + ((size_t)r * G * Cpad + c_eff + i) * 2 + hi;
    g3_v4f a[AD];
#pragma unroll
    for (int d = 0; d < AD; d++) taps_fetch(a[d], ap + (size_t)d * gstride);
    // rows wave * RPW .. + RPW - 1 of the residue: one 1-KiB piece per row (blocks past the end re-read the last one: their columns are never stored)
#pragma unroll
    for (int k = 0; k < RPW; k++) {
        const int row = wave * RPW + k, b = min(b_base + row, n_blocks - 1);
        const float2 *src = Xt + (((size_t)(b / nbl) * inv + r) * nbl + (b % nbl)) * PRE + 2 * lane;      // 16 bytes per lane
        const uint32_t la = __builtin_amdgcn_readfirstlane((int)(lds_base + (uint32_t)row * P4 * 16u));
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(la) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // this wave's rows (and its first taps) have landed
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    if (NW > 1) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    f32x16 acc[3];
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[p][e] = 0.f;
    const float4 *xrow = xs_all + i * P4 + hi;
    float4 xn = xrow[0];                                              // the spectra operand of k-group GG + 1 is read from LDS while group GG multiplies
#define DDC_STEP3N(AV, GG)                                                                                                 \
    {                                                                                                                      \
        const float hs0 = (AV).x + (AV).y, hs1 = (AV).z + (AV).w;                                                          \
        const float4 xv = xn; xn = xrow[2 * min((GG) + 1, G - 1)];                                                         \
        const float xs0 = xv.x + xv.y, xs1 = xv.z + xv.w;                                                                  \
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).x, xv.x, acc[0], 0, 0, 0);                                      \
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).y, xv.y, acc[1], 0, 0, 0);                                      \
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(hs0, xs0, acc[2], 0, 0, 0);                                          \
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).z, xv.z, acc[0], 0, 0, 0);                                      \
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).w, xv.w, acc[1], 0, 0, 0);                                      \
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(hs1, xs1, acc[2], 0, 0, 0);                                          \
    }
    static_assert(G % AD == 0 && G >= 2 * AD, "k loop");
#pragma unroll
    for (int g = 0; g < G - AD; g += AD) {
#pragma unroll
        for (int d = 0; d < AD; d++) { G3_TAPS_READY(a[d], AD - 1); const g3_v4f av = a[d]; taps_fetch(a[d], ap + (size_t)(g + AD + d) * gstride); DDC_STEP3N(av, g + d); }
    }
#define G3N_LAST(D) { G3_TAPS_READY(a[D], AD - 1 - (D)); const g3_v4f av = a[D]; DDC_STEP3N(av, G - AD + (D)); }      /* the last AD groups: nothing new is fetched */
    G3N_LAST(0) G3N_LAST(1) G3N_LAST(2) G3N_LAST(3) G3N_LAST(4) G3N_LAST(5) G3N_LAST(6) G3N_LAST(7)
#undef G3N_LAST
#undef DDC_STEP3N
    if (!active) return;
    // C / D layout of the 32 x 32 tile: column = lane & 31 (block), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (channel)
    int mm[16];
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const int c = min(c_base + (e & 3) + 8 * (e >> 2) + 4 * hi, n_channels - 1);
        mm[e] = (r - geom[c].offsetbin) & (inv - 1);                  // inv is 512 here (a power of two)
    }
    if (b_base + i >= n_blocks) return;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const int c = c_base + (e & 3) + 8 * (e >> 2) + 4 * hi;
        if (c >= n_channels) continue;
        const float p1 = acc[0][e], p2 = acc[1][e], p3 = acc[2][e];
        Ct[((size_t)mm[e] * Cpad + c) * nbp + b_base + i] = make_float2((p1 - p2) * scale, (p3 - p1 - p2) * scale);
    }
}
#undef G3_TAPS_READY

// ------------------------------------------------------------------ the residual shift: data-independent bookkeeping
// Both tables are BLOCK-MAJOR (index b * n_channels + c): a wave's lanes are consecutive channels, so every access below is coalesced (the general
// path's [channel][block] arrays cost one scattered 4-byte store per lane and step: 20 us for 256 x 64 entries).
// k_ddc_chain_t: per channel, the (decimation_remain, starting_phase, output offset) of every block of the call (libcsdr_gpl.c:153-158, float32 phase
// bookkeeping exactly as decimating_shift_addition_cc returns it) and the samples produced.
struct DdcChainJob {                                                 // one call's data-independent bookkeeping (set k of the plan)
    DdcChanState *state, *state_out; const ChanGeom *geom; int n_channels, n_blocks, post_in, post_dec, kmax;      // state_out: where the advanced state goes (= state, or the shadow of tables computed ahead)
    int mode;                                                       // riders of forward pass 1: 1 = the chain tables; 2 = the tables exist already (computed one call ahead): commit their state, phasor checkpoints
    int *blk_remain; float *blk_phase; int *blk_off; int *counts; float2 *R;
    // time-sliced bank (the blocks of a batch dealt to the ranks in runs of seg_nbl): this rank's n_blocks blocks are global blocks [seg_first, seg_first + n_blocks) of
    // seg_total; `state` is the state at global block 0, state_out the one behind block seg_total - 1; seg_counts[rank][channel] = samples every rank's run produces
    int seg_nbl, seg_first, seg_total, seg_world; int *seg_pref;      // seg_pref[g][channel] = samples of the runs of ranks < g; row seg_world = the batch's total
};
// one block of decimating_shift_addition_cc's bookkeeping (libcsdr_gpl.c:153-158): samples produced, state advanced
__host__ __device__ __forceinline__ int ddc_chain_step(DdcChanState &s, float r, int post_in, int post_dec, int sh)
{
    int k = 0, pos = s.remain;
    if (pos < post_in) { k = (sh >= 0 ? (post_in - 1 - pos) >> sh : (post_in - 1 - pos) / post_dec) + 1; pos += k * post_dec; }
    s.remain = pos - post_in;
    float p = s.phase + r * PI_F * (float)k;                           // libcsdr_gpl.c:155
    while (p > PI_F) p -= 2 * PI_F;
    while (p < -PI_F) p += 2 * PI_F;
    s.phase = p;
    return k;
}
// When post_input_size is a multiple of post_decimation (every power-of-two decimation: 448 / 2) and 0 <= remain < post_decimation, a block neither changes
// `remain` nor the sample count k = post_in / post_dec, and the phase advances by the same float d = r PI k every block; for |d| < 6 (channels near the bin
// grid: config 4's have d = 0) the reference's two while loops (libcsdr_gpl.c:156-157) run at most once, so a step is five dependent float operations --
// the same values, bit for bit, as ddc_chain_step (larger |d|: the loops themselves).
struct DdcChainFast { bool ok, big; int k; float d; };
__host__ __device__ __forceinline__ DdcChainFast ddc_chain_fast(const DdcChanState &s, float r, int post_in, int post_dec)
{
    DdcChainFast f;
    f.k = post_in / post_dec; f.d = r * PI_F * (float)f.k;
    f.ok = post_in % post_dec == 0 && s.remain >= 0 && s.remain < post_dec;
    f.big = !(fabsf(f.d) < 6.0f);                                    // a residual shift of a turn or more per block: the wraps may run more than once
    return f;
}
__host__ __device__ __forceinline__ float ddc_phase_step(float p, float d, bool big = false)
{
    p = p + d;
    if (big) { while (p > PI_F) p -= 2 * PI_F; while (p < -PI_F) p += 2 * PI_F; return p; }
    const float lo = p - 2 * PI_F; p = p > PI_F ? lo : p;
    const float hi = p + 2 * PI_F; p = p < -PI_F ? hi : p;
    return p;
}
// The chain of a time-sliced bank: every rank walks ALL blocks of the batch (the state is a strictly sequential float recurrence, but data independent, so
// nothing has to be exchanged), keeps the tables of its own run only -- offsets local to the run -- and notes how many samples every run produces (what the
// output exchange needs to stitch a channel's stream together).
__device__ __forceinline__ void ddc_chain_body_seg(const DdcChainJob &j, int c)
{
    if (c >= j.n_channels) return;
    DdcChanState s = j.state[c];
    const float r = j.geom[c].rate2;
    const int post_in = j.post_in, post_dec = j.post_dec, n_channels = j.n_channels;
    const int sh = (post_dec & (post_dec - 1)) == 0 ? __ffs(post_dec) - 1 : -1;
    const DdcChainFast f = ddc_chain_fast(s, r, post_in, post_dec);
    int total = 0;
    for (int g = 0; g < j.seg_world; g++) {
        const int b0 = min(g * j.seg_nbl, j.seg_total), b1 = min(b0 + j.seg_nbl, j.seg_total);
        const bool mine = b0 == j.seg_first && j.n_blocks > 0;
        int cnt = 0;
        if (f.ok) {
            float p = s.phase;
            if (mine) for (int b = b0; b < b1; b++) {
                const size_t id = (size_t)(b - b0) * n_channels + c;
                j.blk_remain[id] = s.remain; j.blk_phase[id] = p; j.blk_off[id] = (b - b0) * f.k;
                p = ddc_phase_step(p, f.d, f.big);
            } else for (int b = b0; b < b1; b++) p = ddc_phase_step(p, f.d, f.big);
            s.phase = p; cnt = (b1 - b0) * f.k;
        } else if (mine) {
            for (int b = b0; b < b1; b++) {
                const size_t id = (size_t)(b - b0) * n_channels + c;
                j.blk_remain[id] = s.remain; j.blk_phase[id] = s.phase; j.blk_off[id] = cnt;
                cnt += ddc_chain_step(s, r, post_in, post_dec, sh);
            }
        } else for (int b = b0; b < b1; b++) cnt += ddc_chain_step(s, r, post_in, post_dec, sh);
        if (mine) j.counts[c] = cnt;
        j.seg_pref[(size_t)g * n_channels + c] = total; total += cnt;
    }
    j.seg_pref[(size_t)j.seg_world * n_channels + c] = total;
    if (j.n_blocks <= 0) j.counts[c] = 0;
    j.state_out[c] = s;
}
__device__ __forceinline__ void ddc_chain_body(const DdcChainJob &j, int c)
{
    if (j.seg_nbl) { ddc_chain_body_seg(j, c); return; }
    if (c >= j.n_channels) return;
    DdcChanState s = j.state[c];
    const float r = j.geom[c].rate2;
    const int post_in = j.post_in, post_dec = j.post_dec, n_channels = j.n_channels;
    const int sh = (post_dec & (post_dec - 1)) == 0 ? __ffs(post_dec) - 1 : -1;    // post_decimation is 2 for every power-of-two decimation: a shift, not a division
    const DdcChainFast f = ddc_chain_fast(s, r, post_in, post_dec);
    if (f.ok) {                                                       // (lanes of a wave may differ: both forms give the same values)
        float p = s.phase;
        for (int b = 0; b < j.n_blocks; b++) {
            const size_t id = (size_t)b * n_channels + c;
            j.blk_remain[id] = s.remain; j.blk_phase[id] = p; j.blk_off[id] = b * f.k;
            p = ddc_phase_step(p, f.d, f.big);
        }
        s.phase = p;
        j.state_out[c] = s; j.counts[c] = j.n_blocks * f.k;
        return;
    }
    int off = 0;
    for (int b = 0; b < j.n_blocks; b++) {                            // strictly sequential steps: every dependent instruction counts
        const size_t id = (size_t)b * n_channels + c;
        j.blk_remain[id] = s.remain; j.blk_phase[id] = s.phase; j.blk_off[id] = off;
        off += ddc_chain_step(s, r, post_in, post_dec, sh);
    }
    j.state_out[c] = s; j.counts[c] = off;
}
__global__ __launch_bounds__(64) void k_ddc_chain_t(DdcChainJob j) { ddc_chain_body(j, blockIdx.x * 64 + threadIdx.x); }
// k_ddc_rot: the phasor recurrence (c, s) <- (c cd - s sd, s cd + c sd) of every (block, channel) chain from (cos, sin)(its starting phase), replayed in
// float32 like libcsdr_gpl.c:141-152; every ROT_CK-th state is kept (R[kc * n_chains + id]), the consumer replays the < ROT_CK steps in between itself.
constexpr int ROT_CK = 16;
__device__ __forceinline__ void ddc_rot_body(const DdcChainJob &j, int id)
{
    const int n_chains = j.n_channels * j.n_blocks;
    if (id >= n_chains) return;
    const ChanGeom g = j.geom[id % j.n_channels];
    const float cd = g.cosdelta, sd = g.sindelta, ph = j.blk_phase[id];
    float co = (float)cos((double)ph), sn = (float)sin((double)ph);
    for (int k0 = 0; k0 < j.kmax; k0 += ROT_CK) {
        j.R[(size_t)(k0 / ROT_CK) * n_chains + id] = make_float2(co, sn);
#pragma unroll
        for (int kk = 0; kk < ROT_CK; kk++) { const float c1 = co * cd - sn * sd, s1 = sn * cd + co * sd; co = c1; sn = s1; }
    }
}
__global__ __launch_bounds__(64) void k_ddc_rot(DdcChainJob j) { ddc_rot_body(j, blockIdx.x * 64 + threadIdx.x); }

// ------------------------------------------------------------------ 512-point transforms in LDS: N = 512 = 8 x 8 x 8
// n = 64 n1 + 8 n2 + n3, k = k1 + 8 k2 + 64 k3;  W^(nk) = W8^(n1 k1) W512^((8 n2 + n3) k1) W8^(n2 k2) W64^(n3 k2) W8^(n3 k3).
// A workgroup of 256 threads holds 16 transforms (rows of I512_PITCH cells, one pad cell per 8: every stage's accesses are conflict free or 2-way).
// Stage 1 works on values straight from global memory: thread (j = t & 15: transform, i = t >> 4) loads x_j[64 a + tp], a = 0..7, for its four
// tp = i + 16 s, so the 16 lanes j of one load instruction read one 128-byte run; stages 2 and 3 are done by one wave per transform, in place.
__device__ __forceinline__ int pad8(int idx) { return idx + (idx >> 3); }
// row pitch for NT transforms per workgroup: the NT lanes of one global-load instruction (one per transform) then write NT rows: 580 (16 rows) and
// 578 (8 rows) spread them over the banks (2-way at worst)
template <int NT> struct I512 { static constexpr int pitch = NT == 16 ? 580 : 578; static constexpr int tw_n = NT == 16 ? 512 : 448; };   // twiddle indices stay below 7 * 63 + 1

template <bool INV>
__device__ __forceinline__ void fft512_stage1_store(float2 (&v)[8], float2 *row, int tp, const float2 *tw)
{
    dft8<INV>(v);
#pragma unroll
    for (int k1 = 0; k1 < 8; k1++) { float2 w = tw[(k1 * tp) & 511]; if (INV) w.y = -w.y; row[pad8(64 * k1 + tp)] = cmul(v[k1], w); }
}
// stages 2 and 3 of the NT rows: wave w takes rows w, w + 4, ...; all 256 threads must call it (barriers inside)
template <bool INV, int NT>
__device__ __forceinline__ void fft512_stages23(float2 *data, int t, const float2 *tw)
{
    const int wave = t >> 6, lane = t & 63, hi3 = lane >> 3, lo3 = lane & 7;
    for (int round = 0; round < NT / 4; round++) {
        float2 *row = data + (4 * round + wave) * I512<NT>::pitch;
        float2 v[8];
#pragma unroll
        for (int n2 = 0; n2 < 8; n2++) v[n2] = row[pad8(64 * hi3 + 8 * n2 + lo3)];            // lane = (k1, n3), over n2; in place
        dft8<INV>(v);
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++) { float2 w = tw[(8 * k2 * lo3) & 511]; if (INV) w.y = -w.y; row[pad8(64 * hi3 + 8 * k2 + lo3)] = cmul(v[k2], w); }
        __syncthreads();
#pragma unroll
        for (int n3 = 0; n3 < 8; n3++) v[n3] = row[pad8(64 * hi3 + 8 * lo3 + n3)];            // lane = (k1, k2), over n3
        dft8<INV>(v);
        __syncthreads();                                                                      // results land on other lanes' inputs
#pragma unroll
        for (int k3 = 0; k3 < 8; k3++) row[pad8(hi3 + 8 * lo3 + 64 * k3)] = v[k3];            // natural position k1 + 8 k2 + 64 k3
    }
    __syncthreads();
}

// ------------------------------------------------------------------ fused forward transform, 65536 = 512 (k1 = bin residue) x 128 (k2 = q')
//   X[k1 + 512 k2] = sum_n2 W_N^(n2 k1) W_128^(n2 k2) [ sum_n1 x[128 n1 + n2] W_512^(n1 k1) ]          (csdr.c:2289-2299: window b = stream[b inp - ovl ..))
// pass 1: 512-point transforms over n1 for 16 consecutive n2 per workgroup (128-byte runs on both sides), times W_N^(n2 k1), to Y[block][k1][n2];
// pass 2: 128-point transforms over n2 -- one per (residue, block), input and output 1 KiB contiguous -- written straight in the fold's layout
// Xt[residue][block][q] (q = q' with the first fft_swap_sides folded in).  The natural-order spectrum never exists; no framing copy.
// FMT: what `in` holds -- 0 complexf, 1 s16 IQ pairs, 2 u8 IQ pairs: the reference feeds fastddc_fwd_cc from convert_s16_f / convert_u8_f (README.md:66-87, csdr.c:2255-2300);
// converted here with the converters' own arithmetic (convert_dev.hpp: bit-equal to the two stages), so the stream crosses PCIe / xGMI at 4 or 2 bytes per sample
// instead of 8.  The overlap in front of the first window is either `tail` (complexf: the previous call's newest samples, zeros at the stream's start, csdr.c:2279)
// or -- tail == nullptr -- lies in front of `in` itself (a rank's run of a sharded bank: the bytes arrive with their overlap).  tail_out: complexf.
template <int FMT> __device__ __forceinline__ float2 ddc_in_sample(const void *in, long long pos)
{
    if (FMT == 0) return reinterpret_cast<const float2 *>(in)[pos];
    if (FMT == 1) { const uint32_t w = reinterpret_cast<const uint32_t *>(in)[pos]; return make_float2(to_float<2>((int16_t)(w & 0xffff)), to_float<2>((int16_t)(w >> 16))); }
    const uint32_t w = reinterpret_cast<const uint16_t *>(in)[pos]; return make_float2(to_float<0>((int)(w & 0xff)), to_float<0>((int)(w >> 8)));
}

template <int NT, int FMT>
__global__ __launch_bounds__(256) void k_ddc_fwd512(const void *__restrict__ in, const float2 *__restrict__ tail, float2 *__restrict__ tail_out, float2 *__restrict__ Y,
                                                    const float2 *__restrict__ g_tw, const float2 *__restrict__ g_twb, int inp, int ovl, int n_blocks, DdcChainJob cj)
{
    // rows blockIdx.y >= n_blocks are riders -- the call's data-independent chain tables (k_ddc_chain_t's work), done beside the transforms instead of in front
    // of them (12 us of strictly sequential float bookkeeping on a handful of waves).  Measured: kernels of their own 0.175 ms per step, riders at the end of
    // the grid 0.168, riders at the front (dispatched first, s_setprio 3) 0.172 -- beside a CU full of transform waves the chain runs 3 x slower than alone.
    if ((int)blockIdx.y >= n_blocks) {
        const int id = (((int)blockIdx.y - n_blocks) * (int)gridDim.x + (int)blockIdx.x) * 256 + (int)threadIdx.x;
        if (cj.mode == 2) {                                             // tables computed one call ahead: make their end state the current one, then the checkpoints
            if (id < cj.n_channels) cj.state[id] = cj.state_out[id];
            ddc_rot_body(cj, id);
        } else ddc_chain_body(cj, id);
        return;
    }
    extern __shared__ float4 lds_raw[];
    constexpr int PITCH = I512<NT>::pitch, NS = NT / 4, IW = 256 / NT;
    float2 *data = reinterpret_cast<float2 *>(lds_raw), *tw = data + NT * PITCH, *twb = tw + 512;
    const int t = threadIdx.x, j = t & (NT - 1), i = t / NT;
    const int n2 = NT * blockIdx.x + j; const long long b = blockIdx.y;
    tw[t] = g_tw[t]; tw[t + 256] = g_tw[t + 256];
    if (t < 128) twb[t] = g_twb[t];
    const long long base = b * inp - ovl + n2, tail_first = (long long)n_blocks * inp - ovl;
    float2 v[NS][8];
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int a = 0; a < 8; a++) {
            const long long pos = base + 128LL * (64 * a + i + IW * s);
            v[s][a] = (pos < 0 && tail) ? tail[ovl + pos] : ddc_in_sample<FMT>(in, pos);
            // the last window ends with the stream's newest `ovl` samples = the next call's overlap (csdr.c:2292)
            if (tail_out && b == n_blocks - 1 && pos >= tail_first) tail_out[pos - tail_first] = v[s][a];
        }
    __syncthreads();                                                  // twiddle tables
#pragma unroll
    for (int s = 0; s < NS; s++) fft512_stage1_store<false>(v[s], data + j * PITCH, i + IW * s, tw);
    __syncthreads();
    fft512_stages23<false, NT>(data, t, tw);
    float2 *dst = Y + (size_t)b * 65536 + n2;
#pragma unroll 8
    for (int p = 0; p < 512 / IW; p++) {
        const int k1 = i + IW * p, m = n2 * k1;                        // W_65536^m = W_512^(m >> 7) W_65536^(m & 127)
        dst[(size_t)k1 * 128] = cmul(data[j * PITCH + pad8(k1)], cmul(tw[m >> 7], twb[m & 127]));
    }
}

// 128 = 16 (a) x 8 (c): n2 = 8 a + c, k2 = ka + 16 kc; W128^(n2 k2) = W16^(a ka) W128^(c ka) W8^(c kc).  8 lanes per transform, 32 transforms per workgroup.
__global__ __launch_bounds__(256) void k_ddc_fwd128(const float2 *__restrict__ Y, float2 *__restrict__ Xt, const float2 *__restrict__ g_tw, int nbp, int n_blocks, DdcChainJob cj)
{
    // the last row of workgroups (when the call carries riders): the phasor checkpoints of every (block, channel) chain (k_ddc_rot's work)
    if (cj.R && blockIdx.y == gridDim.y - 1) { ddc_rot_body(cj, (int)blockIdx.x * 256 + (int)threadIdx.x); return; }      // (host: only behind a mode-1 pass 1)
    __shared__ __attribute__((aligned(16))) float2 ex[32 * 144];       // [transform][c][18]: ka fastest, pitch 18
    const int t = threadIdx.x, tr = t >> 3, c = t & 7, r = blockIdx.x, b = blockIdx.y * 32 + tr;
    const bool ok = b < n_blocks;
    const float2 *src = Y + ((size_t)(ok ? b : 0) * 512 + r) * 128;
    float2 v[16];
#pragma unroll
    for (int a = 0; a < 16; a++) v[a] = src[8 * a + c];
    dft16<false>(v);
#pragma unroll
    for (int ka = 0; ka < 16; ka++) ex[tr * 144 + c * 18 + ka] = cmul(v[ka], g_tw[(4 * c * ka) & 511]);
    __syncthreads();
    float2 e[8], o[8];
#pragma unroll
    for (int cc = 0; cc < 8; cc++) {
        const float4 two = *reinterpret_cast<const float4 *>(&ex[tr * 144 + cc * 18 + 2 * c]);
        e[cc] = make_float2(two.x, two.y); o[cc] = make_float2(two.z, two.w);
    }
    dft8<false>(e); dft8<false>(o);
    if (!ok) return;
    float2 *dst = Xt + ((size_t)r * nbp + b) * 128;
#pragma unroll
    for (int kc = 0; kc < 8; kc++) {
        const int q = (2 * c + 16 * kc + 64) & 127;                   // q' = ka + 16 kc with ka = 2 c (and 2 c + 1); q = (q' - pre/2) mod pre
        *reinterpret_cast<float4 *>(dst + q) = make_float4(e[kc].x, e[kc].y, o[kc].x, o[kc].y);
    }
}

// ------------------------------------------------------------------ 512-point inverse transforms + scrap + residual shift
// One workgroup (256 threads) = one channel x NT consecutive blocks: the bins of those blocks (runs of NT x 8 bytes per bin) are transformed, the first
// `scrap` samples dropped (overlap & scrap, fastddc.c:153), every post_dec-th sample from decimation_remain on rotated by the replayed phasor and written.
// NT = 8: 40 KiB of LDS, four workgroups per CU (the load, transform and store phases of different workgroups overlap); its two halves of a 128-byte
// bin line are workgroup ids 8 apart = the same XCD (same L2), dispatched together.  NT = 16: whole lines per workgroup, two workgroups per CU.
template <int NT>
__global__ __launch_bounds__(256) void k_ddc_ifft512_post(const float2 *__restrict__ Ct, float2 *__restrict__ out, size_t out_pitch, const float2 *__restrict__ R,
                                                          const float2 *__restrict__ g_tw, const int *__restrict__ blk_remain, const int *__restrict__ blk_off,
                                                          const ChanGeom *__restrict__ geom, int Cpad, int nbp, int n_blocks, int n_channels, int scrap, int post_in, int post_dec)
{
    extern __shared__ float4 lds_raw[];
    constexpr int PITCH = I512<NT>::pitch, TWN = I512<NT>::tw_n, NS = NT / 4, IW = 256 / NT;       // NS sets of 8 loads per thread; IW threads per transform
    float2 *data = reinterpret_cast<float2 *>(lds_raw), *tw = data + NT * PITCH;
    const int t = threadIdx.x, j = t & (NT - 1), i = t / NT;
    int c, b0;
    if (NT == 16) { c = blockIdx.y; b0 = blockIdx.x * 16; }
    else {   // linear id L = 16 g + 8 h + u: (channel, line) pair P = 8 g + u, half h
        const int L = blockIdx.x, h = (L >> 3) & 1, P = (L >> 4) * 8 + (L & 7), nl = (n_blocks + 15) / 16;
        c = P / nl; b0 = (P - c * nl) * 16 + 8 * h;
        if (c >= n_channels) return;
    }
    for (int k = t; k < TWN; k += 256) tw[k] = g_tw[k];
    const bool ok = b0 + j < n_blocks;
    const float2 *src = Ct + (size_t)c * nbp + b0 + (ok ? j : 0);
    const size_t mstride = (size_t)Cpad * nbp;
    float2 v[NS][8];
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int a = 0; a < 8; a++) v[s][a] = src[(size_t)(64 * a + i + IW * s) * mstride];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NS; s++) {
        if (!ok) {
#pragma unroll
            for (int a = 0; a < 8; a++) v[s][a] = make_float2(0.f, 0.f);
        }
        fft512_stage1_store<true>(v[s], data + j * PITCH, i + IW * s, tw);
    }
    __syncthreads();
    fft512_stages23<true, NT>(data, t, tw);
    // fastddc.c:153-162: /size, drop `scrap` samples, rotate every post_dec-th sample from decimation_remain on
    const float inv_n = 1.0f / 512.0f;
    const float cd = geom[c].cosdelta, sd = geom[c].sindelta;
    const size_t n_chains = (size_t)n_blocks * n_channels;
    for (int bl = 0; bl < NT; bl++) {
        const int b = b0 + bl;
        if (b >= n_blocks) break;
        const size_t id = (size_t)b * n_channels + c;
        const int rem = blk_remain[id];
        const int cnt = rem < post_in ? (post_in - 1 - rem) / post_dec + 1 : 0;
        float2 *dst = out + (size_t)c * out_pitch + blk_off[id];
        for (int k = t; k < cnt; k += 256) {
            const float2 x = data[bl * PITCH + pad8(scrap + rem + post_dec * k)];
            const float vi = x.x * inv_n, vq = x.y * inv_n;
            float2 w = R[(size_t)(k / ROT_CK) * n_chains + id];            // the chain's state at the checkpoint below k, then the steps in between
            const int steps = k % ROT_CK;
#pragma unroll
            for (int s2 = 0; s2 < ROT_CK - 1; s2++)
                if (s2 < steps) { const float c1 = w.x * cd - w.y * sd, s1 = w.y * cd + w.x * sd; w.x = c1; w.y = s1; }
            dst[k] = make_float2(w.x * vi - w.y * vq, w.y * vi + w.x * vq);
        }
    }
}

// ------------------------------------------------------------------ the same for post_decimation = 2: HALF-size inverse transforms
// Only every second output sample survives decimating_shift_addition_cc (libcsdr_gpl.c:141: i += decimation), and decimating the OUTPUT of an inverse
// transform by two is aliasing its INPUT: with n = 2k + rho,
//     x[2k + rho] = sum_{m < 256} (X[m] + (-1)^rho X[m + 256]) e^(2 pi i m rho / 512) . e^(2 pi i m k / 256)
// i.e. ONE 256-point inverse transform of the folded, phase-ramped bins Z[m] -- half the butterflies, half the LDS traffic, and the parity rho of the
// first kept sample (scrap + decimation_remain) is known per (block, channel) before the data exists.  Every power-of-two decimation has post_decimation 2
// (fastddc.c:44-48).  256 = 4 x 8 x 8: m = 64 a + 8 n2 + n3, k = k1 + 4 k2 + 32 k3;  W^(mk) = W4^(a k1) W256^((8 n2 + n3) k1) W8^(n2 k2) W64^(n3 k2) W8^(n3 k3).
template <int NT> struct I256 { static constexpr int pitch = NT == 16 ? 292 : 290; };     // 256 cells + one pad per 8 (+ row stagger)

template <int NT>
__global__ __launch_bounds__(256) void k_ddc_ifft256d_post(const float2 *__restrict__ Ct, float2 *__restrict__ out, size_t out_pitch, const float2 *__restrict__ R,
                                                           const float2 *__restrict__ g_tw, const int *__restrict__ blk_remain, const int *__restrict__ blk_off,
                                                           const ChanGeom *__restrict__ geom, int Cpad, int nbp, int n_blocks, int n_channels, int scrap, int post_in,
                                                           DdcChainJob ahead, int n_riders)
{
    // riders (NT = 8 launch only; a multiple of 16 workgroups so that the main ones keep their XCD pairing): the NEXT call's chain tables, data independent, from the
    // state this call's chain ended with (into the other table set and a shadow state: the next call commits them if it has the size assumed and nothing retuned)
    if ((int)blockIdx.x < n_riders) { ddc_chain_body(ahead, (int)blockIdx.x * 256 + (int)threadIdx.x); return; }
    const int bx = (int)blockIdx.x - n_riders;
    extern __shared__ float4 lds_raw[];
    constexpr int PITCH = I256<NT>::pitch, NS = NT / 4, IW = 256 / NT;
    float2 *data = reinterpret_cast<float2 *>(lds_raw), *tw = data + NT * PITCH;
    const int t = threadIdx.x, j = t & (NT - 1), i = t / NT;
    int c, b0;
    if (NT == 16) { c = blockIdx.y; b0 = blockIdx.x * 16; }
    else {   // linear id L = 16 g + 8 h + u: (channel, line) pair P = 8 g + u, half h: the two halves of a 128-byte bin line share an XCD
        const int L = bx, h = (L >> 3) & 1, P = (L >> 4) * 8 + (L & 7), nl = (n_blocks + 15) / 16;
        c = P / nl; b0 = (P - c * nl) * 16 + 8 * h;
        if (c >= n_channels) return;
    }
    tw[t] = g_tw[t]; tw[t + 256] = g_tw[t + 256];
    const bool ok = b0 + j < n_blocks;
    const size_t idj = (size_t)(ok ? b0 + j : 0) * n_channels + c;
    const int remj = blk_remain[idj], rho = (scrap + remj) & 1;
    const float2 *src = Ct + (size_t)c * nbp + b0 + (ok ? j : 0);
    const size_t mstride = (size_t)Cpad * nbp;
    float2 v[NS][8];
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int a = 0; a < 8; a++) v[s][a] = src[(size_t)(64 * a + i + IW * s) * mstride];
    __syncthreads();
    float2 *row = data + j * PITCH;
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int tp = i + IW * s;
        float2 z[4];
#pragma unroll
        for (int a = 0; a < 4; a++) {
            float2 lo = v[s][a], hi = v[s][a + 4];
            if (!ok) { lo = make_float2(0.f, 0.f); hi = lo; }
            float2 sum = rho ? csub(lo, hi) : cadd(lo, hi);
            if (rho) { float2 w = tw[64 * a + tp]; w.y = -w.y; sum = cmul(sum, w); }
            z[a] = sum;
        }
        dft4<true>(z[0], z[1], z[2], z[3]);
#pragma unroll
        for (int k1 = 0; k1 < 4; k1++) { float2 w = tw[(2 * k1 * tp) & 511]; w.y = -w.y; row[pad8(64 * k1 + tp)] = cmul(z[k1], w); }
    }
    __syncthreads();
    {   // stages 2 and 3: 32 lanes per transform, a wave takes two rows per round
        const int wave = t >> 6, lane = t & 63, half = lane >> 5, l5 = lane & 31, k1 = l5 >> 3, lo3 = l5 & 7;
        for (int round = 0; round < NT / 8; round++) {
            float2 *rw = data + (8 * round + 2 * wave + half) * PITCH;
            float2 u[8];
#pragma unroll
            for (int n2 = 0; n2 < 8; n2++) u[n2] = rw[pad8(64 * k1 + 8 * n2 + lo3)];            // lane = (k1, n3), over n2; in place
            dft8<true>(u);
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) { float2 w = tw[(8 * k2 * lo3) & 511]; w.y = -w.y; rw[pad8(64 * k1 + 8 * k2 + lo3)] = cmul(u[k2], w); }
            __syncthreads();
#pragma unroll
            for (int n3 = 0; n3 < 8; n3++) u[n3] = rw[pad8(64 * k1 + 8 * lo3 + n3)];            // lane = (k1, k2), over n3
            dft8<true>(u);
            __syncthreads();
#pragma unroll
            for (int k3 = 0; k3 < 8; k3++) rw[pad8(k1 + 4 * lo3 + 32 * k3)] = u[k3];            // natural position k1 + 4 k2 + 32 k3
        }
        __syncthreads();
    }
    const float inv_n = 1.0f / 512.0f;
    const float cd = geom[c].cosdelta, sd = geom[c].sindelta;
    const size_t n_chains = (size_t)n_blocks * n_channels;
    for (int bl = 0; bl < NT; bl++) {
        const int b = b0 + bl;
        if (b >= n_blocks) break;
        const size_t id = (size_t)b * n_channels + c;
        const int rem = blk_remain[id];
        const int cnt = rem < post_in ? (post_in - 1 - rem) / 2 + 1 : 0;
        const int kbase = (scrap + rem) >> 1;                            // sample scrap + rem + 2 k of the full transform = sample kbase + k of the half-size one
        float2 *dst = out + (size_t)c * out_pitch + blk_off[id];
        for (int k = t; k < cnt; k += 256) {
            const float2 x = data[bl * PITCH + pad8(kbase + k)];
            const float vi = x.x * inv_n, vq = x.y * inv_n;
            float2 w = R[(size_t)(k / ROT_CK) * n_chains + id];
            const int steps = k % ROT_CK;
#pragma unroll
            for (int s2 = 0; s2 < ROT_CK - 1; s2++)
                if (s2 < steps) { const float c1 = w.x * cd - w.y * sd, s1 = w.y * cd + w.x * sd; w.x = c1; w.y = s1; }
            dst[k] = make_float2(w.x * vi - w.y * vq, w.y * vi + w.x * vq);
        }
    }
}

} // namespace

// ====================================================================================== host side
DdcMfma *ddc_mfma_create(csdr_amd_ctx *ctx, int fft, int inv, int pre, int n_channels, int max_blocks, int scrap, int post_in, int post_dec, int input_size, int overlap,
                         const DdcComm *comm)
{
    if (getenv("CSDR_AMD_DDC_MFMA_OFF")) return nullptr;
    if (inv != 512 || pre < 8 || (pre & (pre - 1)) || fft != inv * pre || post_dec < 1 || scrap + post_in > inv) return nullptr;
    DdcMfma *m = new DdcMfma();
    m->ctx = ctx; m->fft = fft; m->inv = inv; m->pre = pre; m->G = pre / 4; m->C = n_channels; m->Cpad = (n_channels + 31) / 32 * 32;
    m->max_blocks = max_blocks; m->nbp = (max_blocks + 31) / 32 * 32; m->scrap = scrap; m->post_in = post_in; m->post_dec = post_dec;
    m->kmax = (post_in - 1) / post_dec + 1; m->rpitch = (m->kmax + ROT_CK - 1) / ROT_CK;      // checkpoints per chain
    m->input_size = input_size; m->overlap = overlap;
    m->comm = comm; m->rank = comm ? comm->rank : 0; m->world = comm ? comm->world : 1;
    {
        DdcMfma::Opt &o = m->opt;
        o.fwd_off = getenv("CSDR_AMD_DDC_FWD_OFF") != nullptr;
        o.riders_off = getenv("CSDR_AMD_DDC_RIDERS_OFF") != nullptr; o.spec_off = getenv("CSDR_AMD_DDC_SPEC_OFF") != nullptr;
        o.pass2_own = getenv("CSDR_AMD_DDC_PASS2") != nullptr;
    }
    m->nbl = m->world > 1 ? (max_blocks + m->world - 1) / m->world : m->nbp;
    const size_t xt_elems = (size_t)m->world * inv * m->nbl * pre;
    hipError_t e = hipMalloc((void **)&m->d_Ht, sizeof(float) * 2 * (size_t)m->Cpad * fft);
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_Ct, sizeof(cf32) * (size_t)inv * m->Cpad * m->nbp);
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_tw, sizeof(float2) * 512);
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_twb, sizeof(float2) * 128);
    for (int k = 0; k < 2 && e == hipSuccess; k++) {
        const size_t n_tab = (size_t)n_channels * max_blocks;
        e = hipMalloc((void **)&m->d_Xt[k], sizeof(cf32) * xt_elems);
        if (e == hipSuccess) e = hipMalloc((void **)&m->d_R[k], sizeof(float2) * n_tab * m->rpitch);
        if (e == hipSuccess) e = hipMalloc((void **)&m->d_blk_remain[k], sizeof(int) * n_tab);
        if (e == hipSuccess) e = hipMalloc((void **)&m->d_blk_off[k], sizeof(int) * n_tab);
        if (e == hipSuccess) e = hipMalloc((void **)&m->d_blk_phase[k], sizeof(float) * n_tab);
        if (e == hipSuccess) e = hipMalloc((void **)&m->d_counts[k], sizeof(int) * n_channels);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&m->ev_ready[k], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&m->ev_free[k], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_state_spec, sizeof(DdcChanState) * (size_t)n_channels);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMemsetAsync(m->d_Ht, 0, sizeof(float) * 2 * (size_t)m->Cpad * fft, ctx->stream);      // padded channel rows stay zero
    // the fused forward transform's buffers (allocated here, not in the first call: a call allocates nothing)
    m->input_size = input_size; m->overlap = overlap;
    if (e == hipSuccess && fft == 65536 && pre == 128) {
        const int y_blocks = m->world > 1 ? m->nbl : m->max_blocks;
        e = hipMalloc((void **)&m->d_Y, sizeof(cf32) * (size_t)y_blocks * fft);
        for (int t = 0; t < 2 && e == hipSuccess; t++) {
            e = hipMalloc((void **)&m->d_tail[t], sizeof(cf32) * (size_t)(overlap + 1));
            if (e == hipSuccess) e = hipMemsetAsync(m->d_tail[t], 0, sizeof(cf32) * (size_t)(overlap + 1), ctx->stream);        // csdr.c:2279: the first window starts with zeros
        }
        if (e == hipSuccess && m->world > 1) e = hipMalloc((void **)&m->d_in_local, sizeof(cf32) * ((size_t)m->nbl * input_size + overlap));
    }
    if (e != hipSuccess) { fail(e, "hipMalloc(fastddc matrix-core path)", __FILE__, __LINE__); ddc_mfma_destroy(m); return nullptr; }
    std::vector<float2> tw(512), twb(128);
    for (int k = 0; k < 512; k++) { const double a = -2.0 * M_PI * k / 512.0; tw[k] = make_float2((float)cos(a), (float)sin(a)); }
    for (int k = 0; k < 128; k++) { const double a = -2.0 * M_PI * k / 65536.0; twb[k] = make_float2((float)cos(a), (float)sin(a)); }
    if (hipMemcpy(m->d_tw, tw.data(), sizeof(float2) * 512, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(m->d_twb, twb.data(), sizeof(float2) * 128, hipMemcpyHostToDevice) != hipSuccess) { ddc_mfma_destroy(m); return nullptr; }
    return m;
}

void ddc_mfma_destroy(DdcMfma *m)
{
    if (!m) return;
    if (m->side) { (void)hipStreamSynchronize(m->side); (void)hipStreamDestroy(m->side); }
    (void)hipFree(m->d_Ht); (void)hipFree(m->d_Ct); (void)hipFree(m->d_tw); (void)hipFree(m->d_twb);
    (void)hipFree(m->d_Y); (void)hipFree(m->d_tail[0]); (void)hipFree(m->d_tail[1]); (void)hipFree(m->d_in_local);
    for (int k = 0; k < 2; k++) {
        (void)hipFree(m->d_Xt[k]); (void)hipFree(m->d_R[k]); (void)hipFree(m->d_blk_remain[k]); (void)hipFree(m->d_blk_off[k]); (void)hipFree(m->d_blk_phase[k]); (void)hipFree(m->d_counts[k]);
        if (m->ev_ready[k]) (void)hipEventDestroy(m->ev_ready[k]);
        if (m->ev_free[k]) (void)hipEventDestroy(m->ev_free[k]);
    }
    if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
    (void)hipFree(m->d_state_spec);
    for (auto &pr : m->ev_pool) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    delete m;
}

// everything queued by submit / collect has finished (retunes and destruction)
int ddc_mfma_quiesce(DdcMfma *m)
{   // (before a retune: the tables computed ahead belong to the old rate / state)
    CSDR_HIP(hipStreamSynchronize(m->side)); CSDR_HIP(hipStreamSynchronize(m->ctx->stream));
    m->spec_valid = false;
    return 0;
}

int ddc_mfma_set_taps(DdcMfma *m, hipStream_t st, const cf32 *d_H, int c_first, int c_count)
{
    if (c_count <= 0) return 0;
    hipLaunchKernelGGL(k_ddc_ht, dim3(cdiv(m->inv, 256), m->pre, c_count), dim3(256), 0, st, reinterpret_cast<const float2 *>(d_H), m->d_Ht, m->fft, m->inv, m->G, m->Cpad, c_first);
    CSDR_LAUNCH_CHECK();
    return 0;
}

bool ddc_mfma_can_forward(const DdcMfma *m) { return m && m->fft == 65536 && m->pre == 128 && !m->opt.fwd_off; }

// chain tables + phasor checkpoints of one call into set k (data independent)
static DdcChainJob mfma_chain_job(DdcMfma *m, int k, int n_blocks, DdcChanState *d_state, const ChanGeom *d_geom)
{
    DdcChainJob j;
    j.state = d_state; j.state_out = d_state; j.mode = 1; j.geom = d_geom; j.n_channels = m->C; j.n_blocks = n_blocks; j.post_in = m->post_in; j.post_dec = m->post_dec; j.kmax = m->kmax;
    j.blk_remain = m->d_blk_remain[k]; j.blk_phase = m->d_blk_phase[k]; j.blk_off = m->d_blk_off[k]; j.counts = m->d_counts[k]; j.R = m->d_R[k];
    j.seg_nbl = m->seg_nbl; j.seg_first = m->seg_first; j.seg_total = m->seg_total; j.seg_world = m->seg_world; j.seg_pref = m->seg_cur;
    return j;
}

// Time-sliced bank: the next submit()'s n_blocks blocks are global blocks [first, first + n_blocks) of a batch of `total`, dealt to `world` ranks in runs of nbl.
// The state handed to submit() is the one at the batch's block 0; the call leaves it behind the batch's last block.  pref_cur receives this call's run
// offsets ([world + 1][n_channels] ints, device), pref_next those of the NEXT call when its chain is computed one call ahead (the caller alternates two
// buffers: this call's pref_next is the next call's pref_cur).  nbl = 0: back to a plain call.
int ddc_mfma_set_segment(DdcMfma *m, int nbl, int first, int total, int world, int *pref_cur, int *pref_next)
{
    m->seg_nbl = nbl; m->seg_first = first; m->seg_total = total; m->seg_world = world; m->seg_cur = pref_cur; m->seg_next = pref_next;
    return 0;
}
int ddc_mfma_pending_blocks(const DdcMfma *m) { return m->pending_blocks[m->drain]; }
// a rank whose run of the batch is empty still has to carry its channels' states over the batch (and to know every run's sample counts)
int ddc_mfma_skip_batch(DdcMfma *m, DdcChanState *d_state, const ChanGeom *d_geom)
{
    if (!m->seg_nbl) return fail_msg(-3, "fastddc: skip_batch outside a time-sliced bank");
    DdcChainJob j = mfma_chain_job(m, m->fill, 0, d_state, d_geom);
    hipLaunchKernelGGL(k_ddc_chain_t, dim3(cdiv(m->C, 64)), dim3(64), 0, m->ctx->stream, j);
    CSDR_LAUNCH_CHECK();
    m->spec_valid = false;
    return 0;
}
static int mfma_chains(DdcMfma *m, hipStream_t st, int k, int n_blocks, DdcChanState *d_state, const ChanGeom *d_geom)
{
    const DdcChainJob j = mfma_chain_job(m, k, n_blocks, d_state, d_geom);
    hipLaunchKernelGGL(k_ddc_chain_t, dim3(cdiv(m->C, 64)), dim3(64), 0, st, j);
    CSDR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_ddc_rot, dim3(cdiv((size_t)m->C * n_blocks, 64)), dim3(64), 0, st, j);
    CSDR_LAUNCH_CHECK();
    return 0;
}

// forward transform of n_loc windows: window b starts at in[b inp - ovl] (positions < 0 come from `tail`); the result goes to Xt chunk `xt` with block pitch nbl.
// riders != nullptr: the call's chain tables and phasor checkpoints are computed by extra workgroups of the two passes (pass 1 carries the chains, pass 2 the
// checkpoints, which need the chains' phases: stream order) instead of by kernels of their own.
// the newest n samples of an integer stream as complexf (the next call's overlap, where the forward pass itself does not write it)
template <int FMT> __global__ __launch_bounds__(256) void k_ddc_cvt_tail(const void *__restrict__ in, long long first, int n, float2 *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = ddc_in_sample<FMT>(in, first + i);
}
int ddc_mfma_convert_samples(hipStream_t st, const void *in, int fmt, long long first, int n, cf32 *out)
{
    if (fmt == 1) hipLaunchKernelGGL(k_ddc_cvt_tail<1>, dim3(cdiv(n, 256)), dim3(256), 0, st, in, first, n, reinterpret_cast<float2 *>(out));
    else if (fmt == 2) hipLaunchKernelGGL(k_ddc_cvt_tail<2>, dim3(cdiv(n, 256)), dim3(256), 0, st, in, first, n, reinterpret_cast<float2 *>(out));
    else CSDR_HIP(hipMemcpyAsync(out, reinterpret_cast<const cf32 *>(in) + first, sizeof(cf32) * (size_t)n, hipMemcpyDeviceToDevice, st));
    CSDR_LAUNCH_CHECK();
    return 0;
}

static int mfma_forward(DdcMfma *m, hipStream_t st, const void *in, int fmt, const cf32 *tail, cf32 *tail_out, int n_loc, cf32 *xt, const DdcChainJob *riders, bool skip_pass2 = false)
{
    if (n_loc <= 0) return 0;
    DdcChainJob cj; memset(&cj, 0, sizeof cj);
    if (riders) cj = *riders;
    const size_t rider_lanes = riders ? (cj.mode == 2 ? (size_t)cj.n_channels * cj.n_blocks : (size_t)cj.n_channels) : 0;      // mode 2: one lane per (block, channel) chain
    // pass 1: 16 columns n2 per workgroup (128-byte runs)
#define DDC_FWD_ARGS in, reinterpret_cast<const float2 *>(tail), reinterpret_cast<float2 *>(tail_out), \
                     reinterpret_cast<float2 *>(m->d_Y), m->d_tw, m->d_twb, m->input_size, m->overlap, n_loc, cj
    hipEvent_t pe1 = nullptr;
    { const int rc = ddc_stage_begin(m, 0, st, &pe1); if (rc) return rc; }
    {
        const size_t lds = (size_t)(16 * I512<16>::pitch + 512 + 128) * sizeof(float2);
        const dim3 grid(8, n_loc + (riders ? cdiv(rider_lanes, 8 * 256) : 0));
#define DDC_FWD_LAUNCH(F) do { const int rc = lds_attr_once((const void *)k_ddc_fwd512<16, F>, lds); if (rc) return rc; \
                               hipLaunchKernelGGL((k_ddc_fwd512<16, F>), grid, dim3(256), lds, st, DDC_FWD_ARGS); } while (0)
        if (fmt == 1) DDC_FWD_LAUNCH(1); else if (fmt == 2) DDC_FWD_LAUNCH(2); else DDC_FWD_LAUNCH(0);
#undef DDC_FWD_LAUNCH
    }
#undef DDC_FWD_ARGS
    CSDR_LAUNCH_CHECK();
    if (pe1) CSDR_HIP(hipEventRecord(pe1, st));
    const bool rot_rides = !skip_pass2 && riders && cj.mode == 1 && (size_t)cj.n_channels * cj.n_blocks <= 512u * 256u;      // mode 2: pass 1 did the checkpoints
    if (!rot_rides) cj.R = nullptr;
    if (!skip_pass2)
    hipLaunchKernelGGL(k_ddc_fwd128, dim3(512, cdiv(n_loc, 32) + (rot_rides ? 1 : 0)), dim3(256), 0, st, reinterpret_cast<const float2 *>(m->d_Y), reinterpret_cast<float2 *>(xt),
                       m->d_tw, m->nbl, n_loc, cj);
    CSDR_LAUNCH_CHECK();
    if (riders && cj.mode == 1 && !rot_rides) {
        hipLaunchKernelGGL(k_ddc_rot, dim3(cdiv((size_t)m->C * cj.n_blocks, 64)), dim3(64), 0, st, *riders);
        CSDR_LAUNCH_CHECK();
    }
    return 0;
}

// will collect() fold n_blocks blocks with k_ddc_gemm3 (persistent shape, pre_decimation 128)?  submit() needs to know: only that kernel can run pass 2 itself
static bool ddc_folds_with_gemm3(const DdcMfma *m, int n_blocks)
{
    const int nbt = n_blocks > 32 ? 2 : 1;
    const size_t lds = (size_t)32 * nbt * (m->pre / 2 + 1) * sizeof(float4);
    const int per_res = (int)(cdiv(m->Cpad, 256) * cdiv(n_blocks, 32 * nbt));
    int slots = current_device_cu_count() / per_res; if (slots < 1) slots = 1; if (slots > m->inv) slots = m->inv;
    bool persist = m->pre <= 128 && 2 * lds <= 160 * 1024 - 512 && slots * 2 <= m->inv;
    return persist && m->pre == 128;
}

// few channel rows (a small bank, a rank's slice of a channel-sharded one): k_ddc_gemm3n; CSDR_AMD_DDC_NARROW=0 keeps the eight-wave kernel
static bool ddc_fold_is_narrow(const DdcMfma *m, int n_blocks)
{
    static const bool off = getenv("CSDR_AMD_DDC_NARROW") && atoi(getenv("CSDR_AMD_DDC_NARROW")) == 0;
    return !off && m->Cpad <= 128 && (m->inv & (m->inv - 1)) == 0 && ddc_folds_with_gemm3(m, n_blocks);
}

// Stage one call: `in` = n_blocks x input_size NEW wideband samples (on rank 0 of a sharded bank; ignored elsewhere), or `spectra` = the natural
// [n_blocks][fft] spectra of csdr fastddc_fwd_cc (single GPU only).  Runs on the side stream: chains, [scatter of the input windows by blocks -> local
// forward transforms -> all-gather of the transposed spectra], into the set that collect() folds next.  At most two calls may be staged.
// inline = true (process(): submit immediately followed by collect, nothing else staged): the transforms go on the context's stream itself and only the
// chains use the side stream, beside them -- on one GPU the forward transforms cannot overlap the previous batch's fold anyway (the fold's workgroups
// hold the whole LDS of every CU), and a stream hand-off per kernel group costs more than it hides.
int ddc_mfma_submit(DdcMfma *m, const void *in_v, const cf32 *spectra, int n_blocks, DdcChanState *d_state, const ChanGeom *d_geom, bool inline_call, const cf32 *ext_tail,
                    int fmt, bool tail_in_front)
{
    // fmt: DDC_IN_CF32 / _S16 / _U8 samples in `in`.  tail_in_front: the overlap of the first window lies in front of `in` in the same format (a run of a sharded
    // bank) -- otherwise ext_tail (complexf) or, when that is null too, the object's own carried tail.
    if (fmt < 0 || fmt > 2) return fail_msg(-3, "fastddc: unknown input format %d", fmt);
    const size_t es = ddc_in_bytes(fmt);
    const uint8_t *in = reinterpret_cast<const uint8_t *>(in_v);
    if (n_blocks <= 0) return fail_msg(-3, "fastddc: nothing to submit");
    if (n_blocks > m->max_blocks) return fail_msg(-3, "fastddc: %d blocks exceed max_blocks %d", n_blocks, m->max_blocks);
    const int k = m->fill;
    if (m->pending_blocks[k]) return fail_msg(-3, "fastddc: two calls are already staged; collect one first");
    hipStream_t mainst = m->ctx->stream;
    const bool inl = inline_call && m->world == 1 && !m->pending_blocks[k ^ 1];
    const int chains_side = 0;                        // (1 = the chains on the side stream beside the transforms: measured 0.190 vs 0.186 ms per step, round 3; never selected)
    hipStream_t st = inl ? mainst : m->side;
    int rc = 0;
    const bool riders_off = m->opt.riders_off, spec_off = m->opt.spec_off;
    const bool fused_fwd = inl && !chains_side && !spectra && ddc_mfma_can_forward(m);
    // Chain tables one call ahead: the previous process() call's inverse-transform kernel carried riders that computed the tables of THIS call (set k) from the
    // state it ended with -- valid when this call has the size that was assumed and nothing retuned in between (ddc_mfma_quiesce).
    const bool spec_hit = fused_fwd && m->spec_valid && m->spec_set == k && m->spec_blocks == n_blocks && m->spec_seg_first == m->seg_first && m->spec_seg_total == m->seg_total;
    m->spec_valid = false;
    const bool ride = fused_fwd && !riders_off;                        // chain work done by extra workgroups of the forward passes (hit: only the commit + the checkpoints)
    m->last_state = d_state; m->ahead_ok = fused_fwd && !spec_off && !riders_off;
    m->y_holds[k] = false;
    if (ride) {
    } else if (inl && !chains_side) {
        rc = mfma_chains(m, mainst, k, n_blocks, d_state, d_geom); if (rc) return rc;
    } else {
        CSDR_HIP(hipEventRecord(m->ev_fork, mainst));                   // the producers of `in` queued so far; the readers of this set's tables (inline: same stream order)
        CSDR_HIP(hipStreamWaitEvent(m->side, m->ev_fork, 0));
        // (the fold that read this set two calls ago was queued on the context's stream before this call: ev_fork orders the side stream behind it)
        rc = mfma_chains(m, m->side, k, n_blocks, d_state, d_geom); if (rc) return rc;
    }
    if (spectra) {
        if (m->world > 1) return fail_msg(-3, "fastddc: natural-order spectra cannot feed a sharded bank");
        hipLaunchKernelGGL(k_ddc_xt, dim3(cdiv(m->inv, 32), cdiv(m->pre, 32), n_blocks), dim3(256), 0, st, reinterpret_cast<const float2 *>(spectra),
                           reinterpret_cast<float2 *>(m->d_Xt[k]), m->fft, m->inv, m->pre, m->nbl);
        CSDR_LAUNCH_CHECK();
    } else {
        if (!ddc_mfma_can_forward(m)) return fail_msg(-3, "fastddc: the fused forward transform covers fft_size 65536 / pre_decimation 128 only");
        const int inp = m->input_size, ovl = m->overlap;
        if (!m->d_Y) return fail_msg(-3, "fastddc: forward buffers missing");
        if (m->world == 1) {
            DdcChainJob job = mfma_chain_job(m, k, n_blocks, d_state, d_geom);
            if (spec_hit) { job.mode = 2; job.state_out = m->d_state_spec; }
            const bool fuse2_off = m->opt.pass2_own;                              // CSDR_AMD_DDC_PASS2: k_ddc_fwd128 stays a kernel of its own
            const bool skip2 = inl && !fuse2_off && ddc_folds_with_gemm3(m, n_blocks) && !ddc_fold_is_narrow(m, n_blocks);      // the fold runs pass 2 itself (d_Y is this call's until its collect())
            // ext_tail: the overlap in front of the first window comes from the caller (a time-sliced bank: the stream before this rank's run is another rank's)
            if (tail_in_front) rc = mfma_forward(m, st, in, fmt, nullptr, nullptr, n_blocks, m->d_Xt[k], ride ? &job : nullptr, skip2);
            else if (ext_tail) rc = mfma_forward(m, st, in, fmt, ext_tail, nullptr, n_blocks, m->d_Xt[k], ride ? &job : nullptr, skip2);
            else { rc = mfma_forward(m, st, in, fmt, m->d_tail[m->flip], m->d_tail[m->flip ^ 1], n_blocks, m->d_Xt[k], ride ? &job : nullptr, skip2); m->flip ^= 1; }
            if (rc) return rc;
            m->y_holds[k] = skip2;
        } else {
            // rank g transforms blocks [g nbl, (g + 1) nbl): the root sends it the samples of its windows, stream[g nbl inp - ovl, min((g + 1) nbl, n) inp),
            // over its own link (seven transfers in flight from the root), then the chunks are all-gathered over the full mesh
            const DdcComm *cm = m->comm;
            auto first_of = [&](int g) { return g * m->nbl < n_blocks ? g * m->nbl : n_blocks; };
            const int b0 = first_of(m->rank), b1 = first_of(m->rank + 1), n_loc = b1 - b0;
            rc = cm->group_start(cm); if (rc) return rc;
            if (m->rank == 0) {
                for (int g = 1; g < m->world; g++) {
                    const int g0 = first_of(g), g1 = first_of(g + 1);
                    // (the raw samples: 8, 4 or 2 bytes each -- the root's egress is what bounds the scaling of a single-ingest bank; counts in 4-byte words)
                    if (g1 > g0) { rc = cm->send(cm, in + ((size_t)g0 * inp - ovl) * es, ((size_t)(g1 - g0) * inp + ovl) * es / 4, g, st); if (rc) return rc; }
                }
            } else if (n_loc > 0) { rc = cm->recv(cm, m->d_in_local, ((size_t)n_loc * inp + ovl) * es / 4, 0, st); if (rc) return rc; }
            rc = cm->group_end(cm); if (rc) return rc;
            cf32 *chunk = m->d_Xt[k] + (size_t)m->rank * m->inv * m->nbl * m->pre;
            if (m->rank == 0) {
                rc = mfma_forward(m, st, in, fmt, m->d_tail[m->flip], nullptr, n_loc, chunk, nullptr); if (rc) return rc;
                // the next call's overlap = the newest ovl samples of the stream (input_size >= overlap_length at this geometry), as complexf
                rc = ddc_mfma_convert_samples(st, in, fmt, (long long)n_blocks * inp - ovl, ovl, m->d_tail[m->flip ^ 1]); if (rc) return rc;
                m->flip ^= 1;
            } else { rc = mfma_forward(m, st, reinterpret_cast<const uint8_t *>(m->d_in_local) + (size_t)ovl * es, fmt, nullptr, nullptr, n_loc, chunk, nullptr); if (rc) return rc; }
            rc = cm->all_gather(cm, m->d_Xt[k], 2 * (size_t)m->inv * m->nbl * m->pre, st); if (rc) return rc;      // in place: every rank's chunk sits at its offset
        }
    }
    if (!(inl && !chains_side)) CSDR_HIP(hipEventRecord(m->ev_ready[k], m->side));      // inline: the side stream carries only the chains
    m->inline_set[k] = inl; m->chains_on_side[k] = !(inl && !chains_side);
    m->pending_blocks[k] = n_blocks; m->fill ^= 1;
    return 0;
}

const char *ddc_mfma_kernel_name(const DdcMfma *m) { return m->gemm_three ? (m->gemm_narrow ? "k_ddc_gemm3n" : "k_ddc_gemm3") : "k_ddc_gemm"; }
int ddc_mfma_set_profiling(DdcMfma *m, int on)
{
    m->profiling = on != 0; m->ev_used = 0; m->prof_ms = 0; m->prof_launches = 0;
    m->profile_stages = on == 2;                                         // (2: the kernels around the fold too -- two more event pairs per call, not for a timed region)
    for (auto &p : m->stage) { p.used = 0; p.ms = 0; p.launches = 0; }
    return 0;
}
// stage 1: the forward transform's first pass (k_ddc_fwd512), stage 2: the inverse transforms + scrap + residual shift (k_ddc_ifft256d_post / k_ddc_ifft512_post)
int ddc_mfma_stage_time(DdcMfma *m, int stage, double *total_ms, long *launches)
{
    if (stage < 1 || stage > 2) return -3;
    DdcMfma::StageProf &p = m->stage[stage - 1];
    for (size_t k = 0; k < p.used; k++) {
        CSDR_HIP(hipEventSynchronize(p.pool[k].second));
        float ms = 0; CSDR_HIP(hipEventElapsedTime(&ms, p.pool[k].first, p.pool[k].second));
        p.ms += ms; p.launches++;
    }
    p.used = 0;
    *total_ms = p.ms; *launches = p.launches;
    return 0;
}
int ddc_mfma_kernel_time(DdcMfma *m, double *total_ms, long *launches)
{
    for (size_t k = 0; k < m->ev_used; k++) {
        CSDR_HIP(hipEventSynchronize(m->ev_pool[k].second));
        float ms = 0; CSDR_HIP(hipEventElapsedTime(&ms, m->ev_pool[k].first, m->ev_pool[k].second));
        m->prof_ms += ms; m->prof_launches++;
    }
    m->ev_used = 0;
    *total_ms = m->prof_ms; *launches = m->prof_launches;
    return 0;
}

// Fold + inverse transforms + scrap + residual shift of the oldest staged call, on the context's stream.  Returns its block count (or < 0);
// *d_counts receives the device array of samples written per channel.
int ddc_mfma_collect(DdcMfma *m, const ChanGeom *d_geom, cf32 *out, size_t out_pitch, const int **d_counts, hipEvent_t after_inverse)
{
    const int k = m->drain, n_blocks = m->pending_blocks[k];
    if (!n_blocks) return fail_msg(-3, "fastddc: nothing staged to collect");
    hipStream_t st = m->ctx->stream;
    if (m->chains_on_side[k]) CSDR_HIP(hipStreamWaitEvent(st, m->ev_ready[k], 0));
    const float scale = 1.0f / (float)m->pre;                              // fastddc.c:144-148 (a power of two: exact)
    const int nbt = n_blocks > 32 ? 2 : 1;
    const size_t lds = (size_t)32 * nbt * (m->pre / 2 + 1) * sizeof(float4);
    const dim3 grid(m->inv, cdiv(m->Cpad, 256), cdiv(n_blocks, 32 * nbt));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (m->profiling) {
        if (m->ev_used == m->ev_pool.size()) {
            hipEvent_t a, b; CSDR_HIP(hipEventCreate(&a)); CSDR_HIP(hipEventCreate(&b));
            m->ev_pool.emplace_back(a, b);
        }
        e0 = m->ev_pool[m->ev_used].first; e1 = m->ev_pool[m->ev_used].second; m->ev_used++;
        CSDR_HIP(hipEventRecord(e0, st));
    }
    // persistent form (one workgroup per CU walks several residues, double-buffered spectra) when a residue's spectra fit the register staging
    // and there are at least two residues per workgroup
    const int n_cu = current_device_cu_count();
    const int per_res = (int)(grid.y * grid.z);
    int slots = n_cu / per_res; if (slots < 1) slots = 1; if (slots > m->inv) slots = m->inv;
    bool persist = m->pre <= 128 && 2 * lds <= 160 * 1024 - 512 && slots * 2 <= m->inv;
    const size_t lds_use = persist ? 2 * lds : lds;
    const dim3 grid_use(persist ? (unsigned)slots : grid.x, grid.y, grid.z);
#define DDC_GEMM_LAUNCH(NBTV, PV) do {                                                                                                               \
        if (lds_use > 64 * 1024) { const int rc = lds_attr_once((const void *)k_ddc_gemm<NBTV, PV>, lds_use); if (rc) return rc; }                    \
        hipLaunchKernelGGL((k_ddc_gemm<NBTV, PV>), grid_use, dim3(512), lds_use, st, m->d_Ht, reinterpret_cast<const float2 *>(m->d_Xt[k]),           \
                           reinterpret_cast<float2 *>(m->d_Ct), d_geom, m->inv, m->pre, m->Cpad, m->C, m->nbp, m->nbl, n_blocks, scale); } while (0)
    // three-product form with LDS-DMA staging: pre_decimation 128 (a spectra row = one 1-KiB piece), persistent shape; other geometries keep the four-product kernel k_ddc_gemm
    const bool three = ddc_folds_with_gemm3(m, n_blocks);
    m->gemm_three = three; m->gemm_narrow = false;
    if (three) {
        // the second pass of the forward transform inside the fold: one GPU, process() (pass 1's output Y belongs to this call), submit() skipped k_ddc_fwd128
        const bool fwd = m->y_holds[k];
        const float2 *src = fwd ? reinterpret_cast<const float2 *>(m->d_Y) : reinterpret_cast<const float2 *>(m->d_Xt[k]);
#define DDC_GEMM3_LAUNCH(NBTV, FV) do { const int rc = lds_attr_once((const void *)k_ddc_gemm3<NBTV, FV>, lds_use); if (rc) return rc;                     \
        hipLaunchKernelGGL((k_ddc_gemm3<NBTV, FV>), grid_use, dim3(512), lds_use, st, m->d_Ht, src, reinterpret_cast<float2 *>(m->d_Ct), d_geom, m->inv, m->Cpad, m->C,  \
                           m->nbp, m->nbl, n_blocks, scale, m->d_tw); } while (0)
        m->gemm_narrow = !fwd && ddc_fold_is_narrow(m, n_blocks);
        if (m->gemm_narrow) {
            const dim3 gn(m->inv, 1, cdiv(n_blocks, 32));
            const size_t ldsn = (size_t)32 * (m->pre / 2 + 1) * sizeof(float4);
#define DDC_GEMM3N_LAUNCH(NWV) hipLaunchKernelGGL((k_ddc_gemm3n<NWV>), gn, dim3(64 * NWV), ldsn, st, m->d_Ht, src, reinterpret_cast<float2 *>(m->d_Ct), d_geom, m->inv, m->Cpad, m->C, \
                                                  m->nbp, m->nbl, n_blocks, scale)
            if (m->Cpad <= 32) DDC_GEMM3N_LAUNCH(1); else if (m->Cpad <= 64) DDC_GEMM3N_LAUNCH(2); else DDC_GEMM3N_LAUNCH(4);
#undef DDC_GEMM3N_LAUNCH
        }
        else if (nbt == 2) { if (fwd) DDC_GEMM3_LAUNCH(2, true); else DDC_GEMM3_LAUNCH(2, false); }
        else          { if (fwd) DDC_GEMM3_LAUNCH(1, true); else DDC_GEMM3_LAUNCH(1, false); }
#undef DDC_GEMM3_LAUNCH
    }
    else if (nbt == 2) { if (persist) DDC_GEMM_LAUNCH(2, true); else DDC_GEMM_LAUNCH(2, false); }
    else               { if (persist) DDC_GEMM_LAUNCH(1, true); else DDC_GEMM_LAUNCH(1, false); }
#undef DDC_GEMM_LAUNCH
    CSDR_LAUNCH_CHECK();
    if (e1) CSDR_HIP(hipEventRecord(e1, st));
    // inverse transforms.  post_decimation 2 (every power-of-two decimation): half-size transforms of the aliased bins; otherwise the full-size form.  Half 128-byte
    // bin lines per workgroup (8 blocks: more workgroups per CU than whole lines -- the <16> instantiations measured slower in round 2 and are gone)
    const bool full = m->post_dec != 2;
    const int pairs = m->C * cdiv(n_blocks, 16);
    const dim3 g8(cdiv(pairs, 8) * 16);
    hipEvent_t pe2 = nullptr;
    { const int rc = ddc_stage_begin(m, 1, st, &pe2); if (rc) return rc; }
#define DDC_IFFT_ARGS reinterpret_cast<const float2 *>(m->d_Ct), reinterpret_cast<float2 *>(out), out_pitch, m->d_R[k], m->d_tw, m->d_blk_remain[k], m->d_blk_off[k], d_geom, m->Cpad, m->nbp, n_blocks, m->C, m->scrap, m->post_in
    DdcChainJob ahead; memset(&ahead, 0, sizeof ahead);
    if (full) {
        hipLaunchKernelGGL(k_ddc_ifft512_post<8>, g8, dim3(256), (size_t)(8 * I512<8>::pitch + I512<8>::tw_n) * sizeof(float2), st, DDC_IFFT_ARGS, m->post_dec);
    } else {
        int n_riders = 0;
        if (m->ahead_ok && m->inline_set[k] && m->last_state && m->C <= 16 * 256) {      // the next call's chain tables into the other set, state into the shadow
            ahead = mfma_chain_job(m, k ^ 1, n_blocks, m->last_state, d_geom);
            ahead.state_out = m->d_state_spec; ahead.seg_pref = m->seg_next;
            n_riders = 16;
            m->spec_valid = true; m->spec_set = k ^ 1; m->spec_blocks = n_blocks; m->spec_seg_first = m->seg_first; m->spec_seg_total = m->seg_total;
        }
        // after_inverse: an event that completes with THIS kernel (its own completion signal: a hipEventRecord behind it would put a marker packet between this
        // call's last kernel and the next call's first one)
        if (after_inverse) hipExtLaunchKernelGGL(k_ddc_ifft256d_post<8>, dim3(g8.x + n_riders), dim3(256), (size_t)(8 * I256<8>::pitch + 512) * sizeof(float2), st, nullptr, after_inverse, 0, DDC_IFFT_ARGS, ahead, n_riders);
        else
        hipLaunchKernelGGL(k_ddc_ifft256d_post<8>, dim3(g8.x + n_riders), dim3(256), (size_t)(8 * I256<8>::pitch + 512) * sizeof(float2), st, DDC_IFFT_ARGS, ahead, n_riders);
        after_inverse = nullptr;
    }
#undef DDC_IFFT_ARGS
    CSDR_LAUNCH_CHECK();
    if (pe2) CSDR_HIP(hipEventRecord(pe2, st));
    if (after_inverse) CSDR_HIP(hipEventRecord(after_inverse, st));     // (the other inverse-transform variants)
    if (d_counts) *d_counts = m->d_counts[k];
    m->pending_blocks[k] = 0; m->drain ^= 1;
    return n_blocks;
}

} // namespace csdr_amd

// Test hook (tests/test_abi_cpu.py): the residual-shift bookkeeping of `n_blocks` consecutive blocks of one channel on the CPU, through the GENERAL step (mode 0: the
// reference's arithmetic, libcsdr_gpl.c:153-158) or the FAST path the kernels take when a block changes neither `remain` nor the sample count (mode 1; returns -1 when its
// precondition does not hold): phases_out[b] = the phase in front of block b, *remain_io / *phase_io = the state in front of block 0, behind the last block on return.
extern "C" int csdr_amd_debug_ddc_chain(int mode, float rate2, int post_in, int post_dec, int n_blocks, int *remain_io, float *phase_io, float *phases_out, int *count_out)
{
    using namespace csdr_amd;
    DdcChanState s{*remain_io, *phase_io};
    const int sh = (post_dec & (post_dec - 1)) == 0 ? __builtin_ctz(post_dec) : -1;
    int total = 0;
    if (mode == 1) {
        const DdcChainFast f = ddc_chain_fast(s, rate2, post_in, post_dec);
        if (!f.ok) return -1;
        float p = s.phase;
        for (int b = 0; b < n_blocks; b++) { phases_out[b] = p; p = ddc_phase_step(p, f.d, f.big); total += f.k; }
        s.phase = p;
    } else {
        for (int b = 0; b < n_blocks; b++) { phases_out[b] = s.phase; total += ddc_chain_step(s, rate2, post_in, post_dec, sh); }
    }
    *remain_io = s.remain; *phase_io = s.phase; *count_out = total;
    return 0;
}

// Test hook: the 8-point butterfly of the 512-point transform on the CPU; 8 interleaved complex floats
extern "C" void csdr_amd_debug_dft8(const float *in16, float *out16, int inverse)
{
    float2 v[8];
    for (int k = 0; k < 8; k++) v[k] = make_float2(in16[2 * k], in16[2 * k + 1]);
    if (inverse) dft8<true>(v); else dft8<false>(v);
    for (int k = 0; k < 8; k++) { out16[2 * k] = v[k].x; out16[2 * k + 1] = v[k].y; }
}
