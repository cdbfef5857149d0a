// f2blocks.hip -- the remaining simple blocks of the AM/SSB receive chains and the waterfall path.
//   amdemod_cf / amdemod_estimator_cf   libcsdr.c:861-901
//   fmdemod_atan_cf                     libcsdr.c:1004-1019
//   dcblock_ff / fastdcblock_ff         libcsdr.c:903-941
//   agc_ff                              libcsdr_gpl.c:163-260
//   realpart_cf                         csdr.c:634-645
//   logpower_cf                         libcsdr.c:1296-1303
//   precalculate_window + apply_precalculated_window_c, the `fft_cc` framing   libcsdr.c:1256-1276, csdr.c:1569-1641
// All HBM-bound element-wise work except the two recursive ones:
//   dcblock_ff  y[i] = x[i] - x[i-1] + a*y[i-1] is a linear recurrence: every lane owns a 32-sample segment, computes its zero-state
//               end value, one lane per stream chains the segment carries (c' = a^32*c + e), the segments are then replayed from their
//               exact carries -- the reference's own arithmetic inside a segment, carries within float rounding of the sequential ones.
//   agc_ff      a data-dependent state machine (hang / attack-wait counters): strictly sequential per stream, so one lane per stream;
//               it is an audio-rate block and parallel over streams only.
// Built with -ffp-contract=off like the rest of the library.
#include "common.hpp"
#include <stdlib.h>
#include <hipfft/hipfft.h>
#include <math.h>
#include <map>
#include <vector>
using namespace csdr_amd;

namespace {

constexpr int SEG = 32;                                             // dcblock segment length (one lane, eight 16-byte accesses)

template <int OP> __device__ __forceinline__ float cf_to_f(cf32 x, float p0, float p1)
{
    if (OP == 0) { const float s = x.i * x.i + x.q * x.q; return (float)sqrt((double)s); }                 // amdemod_cf: double sqrt of the float sum
    if (OP == 1) {                                                                                          // amdemod_estimator_cf
        const float ai = x.i < 0 ? -x.i : x.i, aq = x.q < 0 ? -x.q : x.q;
        const float mx = aq > ai ? aq : ai, mn = aq < ai ? aq : ai;
        return p0 * mx + p1 * mn;
    }
    if (OP == 2) return x.i;                                                                                // realpart_cf
    const float s = x.i * x.i + x.q * x.q;                                                                  // logpower_cf
    return 10 * (float)log10((double)s) + p0;
}
template <int OP>
__global__ __launch_bounds__(256) void k_cf_to_f(const cf32 *__restrict__ in, float *__restrict__ out, size_t n, float p0, float p1)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) out[k] = cf_to_f<OP>(in[k], p0, p1);
}

// fmdemod_atan_cf: the phase of every sample is needed twice (as "now" and as "last"); a block computes its 256 phases once, shares
// them through LDS, lane 0 also evaluates the sample in front of the block (or takes the carried phase).
__global__ __launch_bounds__(256) void k_fmdemod_atan(const cf32 *__restrict__ in, float *__restrict__ out, size_t n, size_t in_pitch, size_t out_pitch,
                                                      const float *__restrict__ last_phase)
{
    __shared__ float ph[257];
    const float PIf = (float)3.14159265358979323846;                // libcsdr.h:65: PI is a float constant
    const cf32 *src = in + (size_t)blockIdx.y * in_pitch;
    float *dst = out + (size_t)blockIdx.y * out_pitch;
    const size_t k0 = (size_t)blockIdx.x * 256, k = k0 + threadIdx.x;
    if (k < n) { const cf32 x = src[k]; ph[threadIdx.x + 1] = (float)atan2((double)x.q, (double)x.i); }
    if (threadIdx.x == 0) {
        if (k0 == 0) ph[0] = last_phase[blockIdx.y];
        else { const cf32 x = src[k0 - 1]; ph[0] = (float)atan2((double)x.q, (double)x.i); }
    }
    __syncthreads();
    if (k < n) {
        float d = ph[threadIdx.x + 1] - ph[threadIdx.x];
        if (d < -PIf) d += 2 * PIf;
        if (d > PIf) d -= 2 * PIf;
        dst[k] = d / PIf;
    }
}
__global__ void k_store_last_phase(const cf32 *__restrict__ in, size_t n, size_t in_pitch, float *__restrict__ last_phase, int n_streams)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_streams) { const cf32 x = in[(size_t)s * in_pitch + n - 1]; last_phase[s] = (float)atan2((double)x.q, (double)x.i); }
}

// ------------------------------------------------------------------ dcblock_ff
// pass 1: zero-state end value of every segment.  pass 2: carries.  pass 3: replay from the carries.
__global__ __launch_bounds__(256) void k_dc_local(const float *__restrict__ in, size_t n, size_t in_pitch, float a, const float *__restrict__ state,
                                                  float *__restrict__ seg_end, int n_seg)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_seg) return;
    const float *x = in + (size_t)blockIdx.y * in_pitch;
    const size_t b = (size_t)j * SEG;
    const int len = (int)((n - b < (size_t)SEG) ? n - b : SEG);
    float prev = b ? x[b - 1] : state[2 * blockIdx.y], y = 0.f;
    for (int i = 0; i < len; i++) { const float v = x[b + i]; y = v - prev + a * y; prev = v; }
    seg_end[(size_t)blockIdx.y * n_seg + j] = y;
}
__global__ void k_dc_carry(float *__restrict__ seg_end, int n_seg, size_t n, float a, const float *__restrict__ state, int n_streams)
{   // seg_end[j] becomes the y value in FRONT of segment j
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    float *e = seg_end + (size_t)s * n_seg;
    float aL = 1.f;
    for (int i = 0; i < SEG; i++) aL *= a;
    float c = state[2 * s + 1];
    for (int j = 0; j < n_seg; j++) {
        const float end_local = e[j];
        e[j] = c;
        const size_t b = (size_t)j * SEG;
        float ap = aL;
        if (n - b < (size_t)SEG) { ap = 1.f; for (size_t i = 0; i < n - b; i++) ap *= a; }
        c = ap * c + end_local;
    }
}
__global__ __launch_bounds__(256) void k_dc_apply(const float *__restrict__ in, float *__restrict__ out, size_t n, size_t in_pitch, size_t out_pitch, float a,
                                                  const float *__restrict__ state, const float *__restrict__ seg_carry, int n_seg, float *__restrict__ state_out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_seg) return;
    const float *x = in + (size_t)blockIdx.y * in_pitch;
    float *yo = out + (size_t)blockIdx.y * out_pitch;
    const size_t b = (size_t)j * SEG;
    const int len = (int)((n - b < (size_t)SEG) ? n - b : SEG);
    float prev = b ? x[b - 1] : state[2 * blockIdx.y], y = seg_carry[(size_t)blockIdx.y * n_seg + j];
    for (int i = 0; i < len; i++) { const float v = x[b + i]; y = v - prev + a * y; yo[b + i] = y; prev = v; }
    if (j == n_seg - 1) { state_out[2 * blockIdx.y] = prev; state_out[2 * blockIdx.y + 1] = y; }     // state_out is a scratch tail, not state[]
}
__global__ void k_copy_f(const float *__restrict__ src, float *__restrict__ dst, int n)
{ const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) dst[i] = src[i]; }

// ------------------------------------------------------------------ fastdcblock_ff
__global__ __launch_bounds__(256) void k_block_mean(const float *__restrict__ in, size_t in_pitch, int block, int n_blocks, float *__restrict__ avg)
{
    __shared__ float part[256];
    const float *x = in + (size_t)blockIdx.y * in_pitch + (size_t)blockIdx.x * block;
    float s = 0.f;
    for (int i = threadIdx.x; i < block; i += 256) s += x[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w]; __syncthreads(); }
    if (threadIdx.x == 0) avg[(size_t)blockIdx.y * (n_blocks + 1) + blockIdx.x + 1] = part[0] / block;
}
__global__ __launch_bounds__(256) void k_fastdc_apply(const float *__restrict__ in, float *__restrict__ out, size_t in_pitch, size_t out_pitch, int block, int n_blocks,
                                                      const float *__restrict__ avg)
{   // avg[s][0] = the level carried in, avg[s][b+1] = mean of block b
    const float last = avg[(size_t)blockIdx.y * (n_blocks + 1) + blockIdx.x], cur = avg[(size_t)blockIdx.y * (n_blocks + 1) + blockIdx.x + 1];
    const float diff = cur - last;
    const float *x = in + (size_t)blockIdx.y * in_pitch + (size_t)blockIdx.x * block;
    float *y = out + (size_t)blockIdx.y * out_pitch + (size_t)blockIdx.x * block;
    for (int i = threadIdx.x; i < block; i += 256) y[i] = x[i] - (last + diff * ((float)i / block));
}
__global__ void k_fastdc_seed(float *__restrict__ avg, const float *__restrict__ last_dc, int n_blocks, int n_streams, int store)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    if (store) const_cast<float *>(last_dc)[s] = avg[(size_t)s * (n_blocks + 1) + n_blocks];
    else avg[(size_t)s * (n_blocks + 1)] = last_dc[s];
}

// ------------------------------------------------------------------ agc_ff: one lane per stream
__global__ void k_agc(const float *__restrict__ in, float *__restrict__ out, int n_streams, size_t n, int block, size_t in_pitch, size_t out_pitch,
                      float reference, float attack_rate, float decay_rate, float max_gain, short hang_time, short attack_wait_time, float alpha,
                      float *__restrict__ last_gain_io)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    const float *x = in + (size_t)s * in_pitch;
    float *y = out + (size_t)s * out_pitch;
    float last_gain = last_gain_io[s];
    for (size_t at = 0; at < n; at += block) {                      // one libcsdr call per block: counters and the peak estimate restart
        const size_t len = (n - at < (size_t)block) ? n - at : block;
        short hang_counter = 0, attack_wait_counter = 0;
        float gain = last_gain, last_peak = reference / last_gain, dgain;
        y[at] = last_gain * x[at];
        for (size_t k = 1; k < len; k++) {
            const float v = x[at + k];
            const float av = fabsf(v);
            const float error = reference / av - gain;
            if (v != 0) {
                if (error < 0) {
                    if (last_peak < av) { attack_wait_counter = attack_wait_time; last_peak = av; }
                    if (attack_wait_counter > 0) { attack_wait_counter--; dgain = 0; }
                    else { dgain = error * attack_rate; hang_counter = hang_time; }
                } else {
                    if (hang_counter > 0) { hang_counter--; dgain = 0; }
                    else dgain = error * decay_rate;
                }
                gain = gain + dgain;
            }
            if (gain > max_gain) gain = max_gain;
            if (gain < 0) gain = 0;
            gain = gain + last_gain - alpha * last_gain;
            y[at + k] = gain * v;
            last_gain = gain;
        }
        last_gain = gain;
    }
    last_gain_io[s] = last_gain;
}

// The same for FEW streams (a CLI process has one; the lane-per-stream walk above then waits for a global load and a division per sample: 3.8 M samples/s, the
// slowest stage of the README's AM and SSB pipelines by two orders of magnitude).  One wave per stream: all lanes bring 1024 samples into LDS and form
// reference / |x| (the state-independent division), lane 0 runs the state machine over them from LDS -- the reference's operations in the reference's order, so
// bit exact like k_agc --, all lanes apply the gains and store.  The chain itself stays serial: it is the algorithm (libcsdr_gpl.c:163-260).
constexpr int AGC_C = 1024;
__global__ __launch_bounds__(64) void k_agc_coop(const float *__restrict__ in, float *__restrict__ out, size_t n, int block, size_t in_pitch, size_t out_pitch,
                                                 float reference, float attack_rate, float decay_rate, float max_gain, short hang_time, short attack_wait_time, float alpha,
                                                 float *__restrict__ last_gain_io)
{
    __shared__ float l_x[AGC_C], l_r[AGC_C];                         // samples; reference / |x|, then the gains
    const int lane = threadIdx.x;
    const size_t s = blockIdx.x;
    const float *x = in + s * in_pitch;
    float *y = out + s * out_pitch;
    float last_gain = last_gain_io[s], gain = last_gain, last_peak = 0.f;
    short hang_counter = 0, attack_wait_counter = 0;
    size_t call_pos = 0;                                             // position inside the current libcsdr call (block samples each: counters and the peak estimate restart)
    for (size_t at = 0; at < n; at += AGC_C) {
        const int len = (int)((n - at < (size_t)AGC_C) ? n - at : AGC_C);
        for (int k = lane; k < len; k += 64) { const float v = x[at + k]; l_x[k] = v; l_r[k] = reference / fabsf(v); }
        __syncthreads();
        if (lane == 0) {
            // Branch free: every path of the reference's nested ifs is evaluated and the taken one selected -- the operations on the taken path are the
            // reference's, so the values are; the dependent chain per sample is error -> compare -> select -> add -> two clamps -> filter (~10 operations) instead
            // of a walk through exec-mask branches.  Eight samples' operands are read from LDS ahead of their chain.
            const int aw_time = attack_wait_time, h_time = hang_time;
            int hang = hang_counter, aw = attack_wait_counter;
            for (int k0 = 0; k0 < len; k0 += 8) {
                float xv[8], rv[8];
#pragma unroll
                for (int j = 0; j < 8; j++) { const int k = k0 + j < len ? k0 + j : len - 1; xv[j] = l_x[k]; rv[j] = l_r[k]; }
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    if (k0 + j >= len) break;
                    float g_out;
                    if (call_pos == 0) {                             // (uniform, once per `block` samples) a call's first sample: out[0] = last_gain * in[0]; the state
                        hang = 0; aw = 0;                            // is set up from the carried gain
                        gain = last_gain; last_peak = reference / last_gain;
                        g_out = last_gain;
                    } else {
                        const float v = xv[j], av = fabsf(v);
                        const float error = rv[j] - gain;
                        const bool nz = v != 0, neg = error < 0;
                        // error < 0: attack (with its wait counter and the peak estimate)
                        const bool newpeak = last_peak < av;
                        const int aw_a = newpeak ? aw_time : aw;
                        const float lp_a = newpeak ? av : last_peak;
                        const bool waiting = aw_a > 0;
                        const float dg_a = waiting ? 0.f : error * attack_rate;
                        const int aw_a2 = waiting ? aw_a - 1 : aw_a;
                        const int hang_a = waiting ? hang : h_time;
                        // error >= 0: decay (behind the hang counter)
                        const bool hanging = hang > 0;
                        const float dg_d = hanging ? 0.f : error * decay_rate;
                        const int hang_d = hanging ? hang - 1 : hang;
                        const float dgain = neg ? dg_a : dg_d;
                        const float g1 = nz ? gain + dgain : gain;
                        if (nz) { hang = neg ? hang_a : hang_d; aw = neg ? aw_a2 : aw; last_peak = neg ? lp_a : last_peak; }
                        float g2 = g1 > max_gain ? max_gain : g1;
                        g2 = g2 < 0 ? 0.f : g2;
                        gain = g2 + last_gain - alpha * last_gain;
                        g_out = gain;
                        last_gain = gain;
                    }
                    l_r[k0 + j] = g_out;
                    if (++call_pos == (size_t)block) call_pos = 0;
                }
            }
            hang_counter = (short)hang; attack_wait_counter = (short)aw;
        }
        __syncthreads();
        for (int k = lane; k < len; k += 64) y[at + k] = l_r[k] * l_x[k];
        __syncthreads();
    }
    if (lane == 0) last_gain_io[s] = last_gain;
}

// ------------------------------------------------------------------ fft_cc framing + window
// frame f, element i <- sample (f+1)*every - fft + i of [history | in] (overlapped mode) or f*every + i (every > fft)
__global__ __launch_bounds__(256) void k_fft_frame(const cf32 *__restrict__ in, const cf32 *__restrict__ hist, const float *__restrict__ w, cf32 *__restrict__ frames,
                                                   int fft, int every, int n_frames)
{
    const int f = blockIdx.y;
    const int H = fft > every ? fft - every : 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < fft; i += gridDim.x * blockDim.x) {
        cf32 v;
        if (every > fft) v = in[(size_t)f * every + i];
        else { const long g = (long)(f + 1) * every - fft + i; v = g >= 0 ? in[g] : hist[H + g]; }
        const float ww = w[i];
        frames[(size_t)f * fft + i] = cf32{v.i * ww, v.q * ww};
    }
}
__global__ __launch_bounds__(256) void k_fft_hist(const cf32 *__restrict__ in, const cf32 *__restrict__ hist_old, cf32 *__restrict__ hist_new, int H, long consumed)
{   // new history = last H samples of [hist_old | in[0..consumed)]
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H; i += gridDim.x * blockDim.x) {
        const long g = consumed - H + i;
        hist_new[i] = g >= 0 ? in[g] : hist_old[H + g];
    }
}

inline unsigned grid1(size_t n) { size_t g = (n + 255) / 256; if (g < 1) g = 1; if (g > 4096) g = 4096; return (unsigned)g; }

} // namespace

struct csdr_amd_fftcc {
    csdr_amd_ctx *c; int fft, every, max_frames; float *d_w; cf32 *d_hist[2]; int cur; cf32 *d_frames; std::map<int, hipfftHandle> plans;
};

extern "C" {

int csdr_amd_amdemod_cf(csdr_amd_ctx *c, const csdr_complexf *in, float *out, size_t n)
{ if (!n) return 0; hipLaunchKernelGGL((k_cf_to_f<0>), dim3(grid1(n)), dim3(256), 0, c->stream, (const cf32 *)in, out, n, 0.f, 0.f); CSDR_LAUNCH_CHECK(); return 0; }
int csdr_amd_amdemod_estimator_cf(csdr_amd_ctx *c, const csdr_complexf *in, float *out, size_t n, float alpha, float beta)
{
    if (!n) return 0;
    if (alpha == 0) { alpha = 0.947543636291; beta = 0.392485425092; }   // libcsdr.c:881-885
    hipLaunchKernelGGL((k_cf_to_f<1>), dim3(grid1(n)), dim3(256), 0, c->stream, (const cf32 *)in, out, n, alpha, beta); CSDR_LAUNCH_CHECK(); return 0;
}
int csdr_amd_realpart_cf(csdr_amd_ctx *c, const csdr_complexf *in, float *out, size_t n)
{ if (!n) return 0; hipLaunchKernelGGL((k_cf_to_f<2>), dim3(grid1(n)), dim3(256), 0, c->stream, (const cf32 *)in, out, n, 0.f, 0.f); CSDR_LAUNCH_CHECK(); return 0; }
int csdr_amd_logpower_cf(csdr_amd_ctx *c, const csdr_complexf *in, float *out, size_t n, float add_db)
{ if (!n) return 0; hipLaunchKernelGGL((k_cf_to_f<3>), dim3(grid1(n)), dim3(256), 0, c->stream, (const cf32 *)in, out, n, add_db, 0.f); CSDR_LAUNCH_CHECK(); return 0; }

int csdr_amd_fmdemod_atan_cf(csdr_amd_ctx *c, const csdr_complexf *in, float *out, int n_streams, size_t n, size_t in_pitch, size_t out_pitch, float *last_phase_io)
{
    if (!n || n_streams <= 0) return 0;
    hipLaunchKernelGGL(k_fmdemod_atan, dim3(cdiv(n, 256), (unsigned)n_streams), dim3(256), 0, c->stream, (const cf32 *)in, out, n, in_pitch, out_pitch, last_phase_io);
    CSDR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_store_last_phase, dim3(cdiv(n_streams, 64)), dim3(64), 0, c->stream, (const cf32 *)in, n, in_pitch, last_phase_io, n_streams);
    CSDR_LAUNCH_CHECK();
    return 0;
}

int csdr_amd_dcblock_ff(csdr_amd_ctx *c, const float *in, float *out, int n_streams, size_t n, size_t in_pitch, size_t out_pitch, float a, float *state_io)
{
    if (!n || n_streams <= 0) return 0;
    if (a == 0) a = 0.999;                                           // libcsdr.c:909
    const int n_seg = (int)((n + SEG - 1) / SEG);
    float *seg = (float *)c->get_scratch(0, sizeof(float) * ((size_t)n_seg + 2) * n_streams + 64);
    if (!seg) return fail_msg(-2, "dcblock_ff: scratch allocation failed");
    float *new_state = seg + (size_t)n_seg * n_streams;             // written by the last segment of every stream, copied back once all reads of state_io are done
    dim3 grid(cdiv(n_seg, 256), (unsigned)n_streams);
    hipLaunchKernelGGL(k_dc_local, grid, dim3(256), 0, c->stream, in, n, in_pitch, a, state_io, seg, n_seg); CSDR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_dc_carry, dim3(cdiv(n_streams, 64)), dim3(64), 0, c->stream, seg, n_seg, n, a, state_io, n_streams); CSDR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_dc_apply, grid, dim3(256), 0, c->stream, in, out, n, in_pitch, out_pitch, a, state_io, seg, n_seg, new_state); CSDR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_copy_f, dim3(cdiv(2 * n_streams, 64)), dim3(64), 0, c->stream, new_state, state_io, 2 * n_streams); CSDR_LAUNCH_CHECK();
    return 0;
}

int csdr_amd_fastdcblock_ff(csdr_amd_ctx *c, const float *in, float *out, int n_streams, int n_blocks, int block, size_t in_pitch, size_t out_pitch, float *last_dc_io)
{
    if (n_blocks <= 0 || n_streams <= 0 || block <= 0) return 0;
    float *avg = (float *)c->get_scratch(0, sizeof(float) * (size_t)(n_blocks + 1) * n_streams + 64);
    if (!avg) return fail_msg(-2, "fastdcblock_ff: scratch allocation failed");
    hipLaunchKernelGGL(k_fastdc_seed, dim3(cdiv(n_streams, 64)), dim3(64), 0, c->stream, avg, last_dc_io, n_blocks, n_streams, 0); CSDR_LAUNCH_CHECK();
    dim3 grid((unsigned)n_blocks, (unsigned)n_streams);
    hipLaunchKernelGGL(k_block_mean, grid, dim3(256), 0, c->stream, in, in_pitch, block, n_blocks, avg); CSDR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_fastdc_apply, grid, dim3(256), 0, c->stream, in, out, in_pitch, out_pitch, block, n_blocks, avg); CSDR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_fastdc_seed, dim3(cdiv(n_streams, 64)), dim3(64), 0, c->stream, avg, last_dc_io, n_blocks, n_streams, 1); CSDR_LAUNCH_CHECK();
    return 0;
}

int csdr_amd_agc_ff(csdr_amd_ctx *c, const float *in, float *out, int n_streams, size_t n, int block, size_t in_pitch, size_t out_pitch,
                    float reference, float attack_rate, float decay_rate, float max_gain, short hang_time, short attack_wait_time, float gain_filter_alpha,
                    float *last_gain_io)
{
    if (!n || n_streams <= 0) return 0;
    if (block <= 0) return fail_msg(-3, "agc_ff: block must be positive");
    static const bool lane_env = getenv("CSDR_AMD_AGC_LANE") != nullptr;                // (A/B, read once per process)
    if (n_streams < 64 && n >= 256 && !lane_env)
        hipLaunchKernelGGL(k_agc_coop, dim3(n_streams), dim3(64), 0, c->stream, in, out, n, block, in_pitch, out_pitch, reference, attack_rate, decay_rate,
                           max_gain, hang_time, attack_wait_time, gain_filter_alpha, last_gain_io);
    else
    hipLaunchKernelGGL(k_agc, dim3(cdiv(n_streams, 64)), dim3(64), 0, c->stream, in, out, n_streams, n, block, in_pitch, out_pitch, reference, attack_rate, decay_rate,
                       max_gain, hang_time, attack_wait_time, gain_filter_alpha, last_gain_io);
    CSDR_LAUNCH_CHECK();
    return 0;
}

void csdr_amd_precalculate_window(float *windowt, int size, int window)
{   // libcsdr.c:1256-1267 on the host: kernel(2*rate + 1), rate = (float)i/(size-1); same window kernels as the filter design (libcsdr.c:76-97)
    for (int i = 0; i < size; i++) {
        const float rate = (float)i / (size - 1);
        const float r = (float)(2.0 * rate + 1.0);
        if (window == CSDR_WINDOW_BOXCAR) { windowt[i] = 1.0f; continue; }
        const float x = (float)(0.5 + r / 2);
        const float PI_F = (float)3.14159265358979323846;
        windowt[i] = window == CSDR_WINDOW_BLACKMAN ? (float)(0.42 - 0.5 * cos((double)(2 * PI_F * x)) + 0.08 * cos((double)(4 * PI_F * x)))
                                                    : (float)(0.54 - 0.46 * cos((double)(2 * PI_F * x)));
    }
}

csdr_amd_fftcc *csdr_amd_fftcc_create(csdr_amd_ctx *c, int fft_size, int every_n_samples, int window, int max_frames)
{
    if (fft_size <= 0 || (fft_size & (fft_size - 1)) || every_n_samples <= 0 || max_frames <= 0) { fail_msg(-3, "fft_cc: fft_size must be a power of two, every_n and max_frames positive"); return nullptr; }
    csdr_amd_fftcc *f = new csdr_amd_fftcc();
    f->c = c; f->fft = fft_size; f->every = every_n_samples; f->max_frames = max_frames; f->cur = 0;
    const int H = fft_size > every_n_samples ? fft_size - every_n_samples : 0;
    f->d_w = (float *)csdr_amd_malloc(c, 4 * (size_t)fft_size);
    f->d_hist[0] = (cf32 *)csdr_amd_malloc(c, 8 * (size_t)(H + 1)); f->d_hist[1] = (cf32 *)csdr_amd_malloc(c, 8 * (size_t)(H + 1));
    f->d_frames = (cf32 *)csdr_amd_malloc(c, 8 * (size_t)fft_size * max_frames);
    if (!f->d_w || !f->d_hist[0] || !f->d_hist[1] || !f->d_frames) { delete f; return nullptr; }
    std::vector<float> w(fft_size); csdr_amd_precalculate_window(w.data(), fft_size, window);
    csdr_amd_h2d(c, f->d_w, w.data(), 4 * (size_t)fft_size);
    csdr_amd_memset(c, f->d_hist[0], 0, 8 * (size_t)(H + 1));     // the sliding buffer starts empty (the reference's fresh allocation)
    return f;
}
void csdr_amd_fftcc_destroy(csdr_amd_fftcc *f)
{
    if (!f) return;
    for (auto &kv : f->plans) hipfftDestroy(kv.second);
    csdr_amd_free(f->c, f->d_w); csdr_amd_free(f->c, f->d_hist[0]); csdr_amd_free(f->c, f->d_hist[1]); csdr_amd_free(f->c, f->d_frames);
    delete f;
}
// in: n_in new samples (device).  Emits floor(n_in / every_n) spectra of fft_size bins into out; *consumed = frames * every_n.
int csdr_amd_fftcc_process(csdr_amd_fftcc *f, const csdr_complexf *in, size_t n_in, csdr_complexf *out, size_t *consumed)
{
    csdr_amd_ctx *c = f->c;
    int n_frames = (int)(n_in / f->every);
    if (n_frames > f->max_frames) n_frames = f->max_frames;
    if (consumed) *consumed = (size_t)n_frames * f->every;
    if (n_frames <= 0) return 0;
    const int H = f->fft > f->every ? f->fft - f->every : 0;
    hipLaunchKernelGGL(k_fft_frame, dim3(cdiv(f->fft, 256) > 64 ? 64 : cdiv(f->fft, 256), (unsigned)n_frames), dim3(256), 0, c->stream, (const cf32 *)in, f->d_hist[f->cur],
                       f->d_w, f->d_frames, f->fft, f->every, n_frames);
    CSDR_LAUNCH_CHECK();
    if (H) {
        hipLaunchKernelGGL(k_fft_hist, dim3(cdiv(H, 256) > 64 ? 64 : cdiv(H, 256)), dim3(256), 0, c->stream, (const cf32 *)in, f->d_hist[f->cur], f->d_hist[f->cur ^ 1], H,
                           (long)n_frames * f->every);
        CSDR_LAUNCH_CHECK();
        f->cur ^= 1;
    }
    if (!f->plans.count(n_frames)) {
        hipfftHandle h; int n[1] = {f->fft};
        if (hipfftPlanMany(&h, 1, n, nullptr, 1, f->fft, nullptr, 1, f->fft, HIPFFT_C2C, n_frames) != HIPFFT_SUCCESS) return fail_msg(-5, "fft_cc: hipfftPlanMany(%d x %d) failed", f->fft, n_frames);
        hipfftSetStream(h, c->stream);
        f->plans[n_frames] = h;
    }
    if (hipfftExecC2C(f->plans[n_frames], (hipfftComplex *)f->d_frames, (hipfftComplex *)out, HIPFFT_FORWARD) != HIPFFT_SUCCESS) return fail_msg(-5, "fft_cc: hipfftExecC2C failed");
    return n_frames;
}

} // extern "C"
