// fftfilt_lds.hip -- bandpass_fir_fft_cc (csdr.c:1810-1886 = apply_fir_fft_cc, libcsdr.c:814-849, per block) as ONE pass over HBM.
//
// What the reference computes per stream is the linear convolution of the input with the taps: every block of input_size = fft_size - taps + 1 samples is
// zero padded, transformed, multiplied by the taps' spectrum, transformed back, and the last taps - 1 results are added onto the start of the next
// block (overlap ADD).  The block size is bookkeeping: any partition of the stream yields the same samples up to float rounding.  A 65536-point transform
// (BASELINE config 3) does not fit a CU (512 KiB), so fft64k.hip needs three passes over HBM (3 x 16 B per sample).  The taps, however, are short
// (63 ... 4095): here the stream is cut into windows of N = 4096 / 8192 / 16384 samples that DO fit the 160 KiB of LDS, each window overlapping its
// predecessor by taps - 1 samples (overlap SAVE: no dependency between windows, hence between workgroups), and one workgroup does
//     load window -> N-point transform -> x taps spectrum (N-point, 1/N folded in) -> inverse transform -> store the N - (taps - 1) valid samples
// with every intermediate in registers / LDS: 8 B read + 8 B written per sample, the algorithmic minimum.  The caller still sees the reference's framing
// (n_blocks x input_size samples in, the same count out, state carried between calls): the state is the last taps - 1 INPUT samples per stream instead of the
// reference's last taps - 1 partial OUTPUT sums -- the same information.
//
// Transform: in-place decimation-in-frequency stages (natural order in, digit-reversed out), the taps spectrum stored in that digit-reversed order, then the
// mirrored decimation-in-time stages (digit-reversed in, natural out): no reordering pass.  N / 16 threads, 16 points each; first and last stage (radix 16,
// stride N / 16) work on the registers the global loads / stores use; the last forward stage, the bin product and the first inverse stage share registers.
// Plans: 4096 = 16.16.16, 8192 = 16.8.8.8, 16384 = 16.16.8.8.  Every stage function is __host__ __device__: tests/test_abi_cpu.py runs the very same index algebra
// on the CPU (csdr_amd_debug_fftfilt_lds), thread by thread, phase by phase.
#include "common.hpp"
// no bit-exact contract on this path (float FFT filtering, 1e-5 relative RMS): let the butterflies and twiddle products contract into FMAs
#pragma clang fp contract(fast)
#include "fft_butterflies.hpp"
#include <math.h>
#include <stdlib.h>
#include <vector>

using namespace csdr_amd;

typedef int ffl_i32x4 __attribute__((ext_vector_type(4)));
typedef float ffl_f32x2 __attribute__((ext_vector_type(2)));
__device__ ffl_f32x2 ffl_buf_load(ffl_i32x4 rsrc, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");
__device__ void ffl_buf_store(ffl_f32x2 v, ffl_i32x4 rsrc, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.store.v2f32");

#define FFL_HD __host__ __device__ __forceinline__

namespace {

template <int N> struct FflPlan;
template <> struct FflPlan<4096>  { static constexpr int NS = 3, R1 = 16, R2 = 16, R3 = 16, R4 = 1, PADSH = 4; };
template <> struct FflPlan<8192>  { static constexpr int NS = 4, R1 = 16, R2 = 8,  R3 = 8,  R4 = 8, PADSH = 3; };
template <> struct FflPlan<16384> { static constexpr int NS = 4, R1 = 16, R2 = 8,  R3 = 8,  R4 = 16, PADSH = 3; };

template <int N> struct FflGeom {
    using P = FflPlan<N>;
    static constexpr int T = N / 16;                       // threads
    static constexpr int TWN = N / 16;                     // sub-stage twiddle table: exp(-2 pi i e / TWN)
    static constexpr int DATA = N + (N >> P::PADSH);       // padded points
    static constexpr size_t LDS_BYTES = (size_t)(DATA + TWN) * sizeof(float2);
};

template <int PADSH> FFL_HD int ffl_pad(int p) { return p + (p >> PADSH); }
FFL_HD float2 cconj(float2 a) { return make_float2(a.x, -a.y); }

template <int R, bool INV> struct FflDft;
template <bool INV> struct FflDft<16, INV> { static FFL_HD void run(float2 (&a)[16]) { dft16<INV>(a); } };
template <bool INV> struct FflDft<8, INV>  { static FFL_HD void run(float2 (&a)[8])  { dft8<INV>(a); } };

// w^k, k = 0..15, by a multiplication tree of depth <= 4 (keeps the rounding of the high powers at a few ulp)
FFL_HD void ffl_powers16(float2 w1, float2 (&w)[16])
{
    w[0] = make_float2(1.f, 0.f); w[1] = w1; w[2] = cmul(w1, w1); w[3] = cmul(w[2], w1); w[4] = cmul(w[2], w[2]);
    w[5] = cmul(w[4], w1); w[6] = cmul(w[3], w[3]); w[7] = cmul(w[4], w[3]); w[8] = cmul(w[4], w[4]);
    w[9] = cmul(w[8], w1); w[10] = cmul(w[5], w[5]); w[11] = cmul(w[8], w[3]); w[12] = cmul(w[6], w[6]);
    w[13] = cmul(w[8], w[5]); w[14] = cmul(w[7], w[7]); w[15] = cmul(w[8], w[7]);
}

// ---- phase 1: v[j] = x[t + T j] -> radix 16 over j -> x W_N^(t k) -> lds[t + T k]
template <int N>
FFL_HD void ffl_first(float2 (&v)[16], float2 *lds, float2 w1, int t)
{
    using G = FflGeom<N>; constexpr int PS = G::P::PADSH;
    dft16<false>(v);
    float2 w[16]; ffl_powers16(w1, w);
#pragma unroll
    for (int k = 0; k < 16; k++) lds[ffl_pad<PS>(t + G::T * k)] = k ? cmul(v[k], w[k]) : v[k];
}

// ---- last phase: lds[t + T k] x conj(W_N^(t k)) -> inverse radix 16 -> v[j] = y[t + T j]
template <int N>
FFL_HD void ffl_last(float2 (&v)[16], const float2 *lds, float2 w1, int t)
{
    using G = FflGeom<N>; constexpr int PS = G::P::PADSH;
    float2 w[16]; ffl_powers16(cconj(w1), w);
#pragma unroll
    for (int k = 0; k < 16; k++) { const float2 a = lds[ffl_pad<PS>(t + G::T * k)]; v[k] = k ? cmul(a, w[k]) : a; }
    dft16<true>(v);
}

// ---- a middle stage on sub-transforms of length L, radix R, in place.  Forward: butterfly then twiddle W_L^(o k); inverse: conj twiddle then butterfly.
template <int N, int L, int R, bool INV>
FFL_HD void ffl_mid(float2 *lds, const float2 *tws, int t)
{
    using G = FflGeom<N>; constexpr int PS = G::P::PADSH, U = 16 / R, S = L / R, TS = G::TWN / L;
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int q = t + G::T * u, blk = q / S, o = q % S, base = blk * L + o;
        float2 a[R];
#pragma unroll
        for (int j = 0; j < R; j++) {
            a[j] = lds[ffl_pad<PS>(base + S * j)];
            if (INV && j) a[j] = cmul(a[j], cconj(tws[TS * o * j]));
        }
        FflDft<R, INV>::run(a);
#pragma unroll
        for (int k = 0; k < R; k++) lds[ffl_pad<PS>(base + S * k)] = (!INV && k) ? cmul(a[k], tws[TS * o * k]) : a[k];
    }
}

// ---- centre: last forward stage (L = R, stride 1) from LDS, bin product with the taps spectrum (slot order: Hperm[slot * T + t]), first inverse stage back to LDS
// A thread's R points are consecutive.  R = 16 under pad-one-per-eight (the 16384-point plan): lane stride 18 points = 36 dwords, as 8-byte accesses only 16 different
// bank pairs per 32 lanes -- every access a two-way conflict (SQ_LDS_BANK_CONFLICT = 52 % of the LDS's active cycles at 4095 taps, profiles/r6_fftfilt_pmc_issue.json).
// As 16-byte accesses of the aligned pairs the same stride is conflict free (16 lanes x 4 banks, starting banks = the 16 multiples of 4): points 0-7 are four aligned
// pairs, 8 and 15 stay single, 9-14 are three pairs.
template <int PS> FFL_HD void ffl_pair_load(float2 &a, float2 &b, const float2 *p)
{
#ifdef __HIP_DEVICE_COMPILE__
    const float4 v = *reinterpret_cast<const float4 *>(p); a = make_float2(v.x, v.y); b = make_float2(v.z, v.w);
#else
    a = p[0]; b = p[1];
#endif
}
template <int PS> FFL_HD void ffl_pair_store(float2 *p, float2 a, float2 b)
{
#ifdef __HIP_DEVICE_COMPILE__
    *reinterpret_cast<float4 *>(p) = make_float4(a.x, a.y, b.x, b.y);
#else
    p[0] = a; p[1] = b;
#endif
}
template <int N, int R>
FFL_HD void ffl_centre(float2 *lds, const float2 *hperm, int t)
{
    using G = FflGeom<N>; constexpr int PS = G::P::PADSH, U = 16 / R;
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int q = t + G::T * u, base = q * R;
        float2 a[R];
        if constexpr (R == 16 && PS == 3) {
            float2 *p0 = lds + ffl_pad<PS>(base);                       // 18 q: even; points j < 8 at p0 + j, points j >= 8 at p0 + j + 1
#pragma unroll
            for (int j = 0; j < 8; j += 2) ffl_pair_load<PS>(a[j], a[j + 1], p0 + j);
            a[8] = p0[9];
#pragma unroll
            for (int j = 9; j < 15; j += 2) ffl_pair_load<PS>(a[j], a[j + 1], p0 + j + 1);
            a[15] = p0[16];
        } else {
#pragma unroll
            for (int j = 0; j < R; j++) a[j] = lds[ffl_pad<PS>(base + j)];
        }
        FflDft<R, false>::run(a);
#pragma unroll
        for (int k = 0; k < R; k++) a[k] = cmul(a[k], hperm[(size_t)(u * R + k) * G::T + t]);      // libcsdr.c:826-830 (and 836-839: the 1/N is in the table)
        FflDft<R, true>::run(a);
        if constexpr (R == 16 && PS == 3) {
            float2 *p0 = lds + ffl_pad<PS>(base);
#pragma unroll
            for (int j = 0; j < 8; j += 2) ffl_pair_store<PS>(p0 + j, a[j], a[j + 1]);
            p0[9] = a[8];
#pragma unroll
            for (int j = 9; j < 15; j += 2) ffl_pair_store<PS>(p0 + j + 1, a[j], a[j + 1]);
            p0[16] = a[15];
        } else {
#pragma unroll
            for (int j = 0; j < R; j++) lds[ffl_pad<PS>(base + j)] = a[j];
        }
    }
}

// The phases between the first and the last one, in order; `sync` separates them (a barrier on the device, "all threads done" in the CPU harness).
// PHASE counts from 0; returns false when there is no such phase.
template <int N> struct FflMidPhases {
    using P = FflPlan<N>;
    static constexpr int COUNT = 2 * (P::NS - 2) + 1;
    template <int PHASE> static FFL_HD void run(float2 *lds, const float2 *tws, const float2 *hperm, int t)
    {
        constexpr int L2 = N / P::R1, L3 = L2 / P::R2, L4 = L3 / P::R3;
        if constexpr (P::NS == 3) {
            if constexpr (PHASE == 0) ffl_mid<N, L2, P::R2, false>(lds, tws, t);
            else if constexpr (PHASE == 1) ffl_centre<N, P::R3>(lds, hperm, t);
            else ffl_mid<N, L2, P::R2, true>(lds, tws, t);
        } else {
            if constexpr (PHASE == 0) ffl_mid<N, L2, P::R2, false>(lds, tws, t);
            else if constexpr (PHASE == 1) ffl_mid<N, L3, P::R3, false>(lds, tws, t);
            else if constexpr (PHASE == 2) ffl_centre<N, P::R4>(lds, hperm, t);
            else if constexpr (PHASE == 3) ffl_mid<N, L3, P::R3, true>(lds, tws, t);
            else ffl_mid<N, L2, P::R2, true>(lds, tws, t);
            (void)L4;
        }
    }
};

// frequency index held at in-place position p after the forward stages: p = k1 N/R1 + k2 N/(R1 R2) + ... holds X[k1 + R1 k2 + R1 R2 k3 + ...]
template <int N> int ffl_freq_of_position(int p)
{
    using P = FflPlan<N>;
    const int r[4] = {P::R1, P::R2, P::R3, P::R4};
    int len = N, f = 0, mul = 1;
    for (int s = 0; s < P::NS; s++) { len /= r[s]; const int k = p / len; p -= k * len; f += k * mul; mul *= r[s]; }
    return f;
}
// slot (u, k) of thread t in the centre phase <-> position (t + T u) R + k
template <int N> int ffl_slot_position(int slot, int t)
{
    using P = FflPlan<N>;
    const int R = P::NS == 3 ? P::R3 : P::R4;
    return (t + FflGeom<N>::T * (slot / R)) * R + slot % R;
}

// ------------------------------------------------------------------------------------------------ device kernel
// Persistent workgroups (a few per CU) walk the windows of the call; the NEXT window's samples are fetched into registers while the current one is
// transformed, so the HBM pipe never waits for the butterflies.  Window w = (stream, chunk); workgroup ids are dealt round robin over the 8 XCDs, so each
// XCD gets a contiguous range of windows and the workgroups of one XCD take consecutive windows at the same time: the taps - 1 samples neighbouring
// windows share are served by that XCD's L2.
// Buffer loads / stores with the hardware range check do the edges: a window's samples in front of the call's input come from the history (a second
// descriptor), samples behind its end read as zero, results outside [0, m_new) are dropped -- no divergent branch anywhere.  (The LLVM intrinsics are declared
// directly, see fir.hip.)
// The phases exchange data through LDS only.  __syncthreads() is a workgroup-scope fence over EVERY address space: it puts `s_waitcnt vmcnt(0)` in front of the
// barrier, i.e. every barrier drains the global loads and stores in flight -- the previous window's 16 output stores at this window's first barrier, and a prefetch of the
// next window could never fly under the butterflies at all (round 6: found when the 512-thread form of the 16384-point window ran SLOWER with the prefetch than without).
// Here: LDS operations complete (lgkmcnt), then the barrier; global memory operations stay in flight across it.
__device__ __forceinline__ void ffl_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int N>
__device__ __forceinline__ void ffl_load_window(float2 (&v)[16], const float2 *x, const float2 *h, int w0, int k1p, int m_new, int t)
{
    constexpr int T = FflGeom<N>::T;
    const unsigned long long bx = (unsigned long long)x, bh = (unsigned long long)h;
    const ffl_i32x4 rx = {(int)(unsigned)bx, (int)((bx >> 32) & 0xffffu), m_new * 8, 0x00020000};
    const int v0 = (w0 + t) * 8;                                        // negative (history) -> out of range as unsigned -> 0
#pragma unroll
    for (int j = 0; j < 16; j++) { const ffl_f32x2 r = ffl_buf_load(rx, v0 + T * 8 * j, 0, 0); v[j] = make_float2(r.x, r.y); }
    if (w0 < 0) {                                                       // uniform: the stream's first window
        const ffl_i32x4 rh = {(int)(unsigned)bh, (int)((bh >> 32) & 0xffffu), k1p * 8, 0x00020000};
        const int vh = (k1p + w0 + t) * 8;
#pragma unroll
        for (int j = 0; j < 16; j++) { const ffl_f32x2 r = ffl_buf_load(rh, vh + T * 8 * j, 0, 0); v[j].x += r.x; v[j].y += r.y; }
    }
}

// Prefetch of the NEXT window into registers, issued by hand: written as plain loads the compiler sinks them to their use at the top of the next iteration (and the
// prefetching variants measured no better than the plain ones).  The values are made "real" for the compiler at the END of the iteration (ffl_prefetch_ready:
// by then they have landed; vmcnt(16) lets this window's 16 output stores stay in flight), so nothing in flight crosses the loop's back edge.
template <int N>
__device__ __forceinline__ void ffl_prefetch_issue(ffl_f32x2 (&nx)[16], const float2 *x, int w0, int m_new, int t)
{
    constexpr int T = FflGeom<N>::T;
    const unsigned long long bx = (unsigned long long)x;
    const ffl_i32x4 rx = {(int)(unsigned)bx, (int)((bx >> 32) & 0xffffu), m_new * 8, 0x00020000};
    const int v0 = (w0 + t) * 8;
#pragma unroll
    for (int j = 0; j < 16; j++) { const int vo = v0 + T * 8 * j; asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(nx[j]) : "v"(vo), "s"(rx) : "memory"); }
}
template <int INFLIGHT>
__device__ __forceinline__ void ffl_prefetch_ready(ffl_f32x2 (&nx)[16])
{
    asm volatile("s_waitcnt vmcnt(%16)" : "+v"(nx[0]), "+v"(nx[1]), "+v"(nx[2]), "+v"(nx[3]), "+v"(nx[4]), "+v"(nx[5]), "+v"(nx[6]), "+v"(nx[7]),
                                         "+v"(nx[8]), "+v"(nx[9]), "+v"(nx[10]), "+v"(nx[11]), "+v"(nx[12]), "+v"(nx[13]), "+v"(nx[14]), "+v"(nx[15]) : "n"(INFLIGHT));
}

// LPT (round 6): logical threads per physical thread.  The stage functions are written for N / 16 logical threads of 16 points each; with LPT = 2 a physical thread runs
// two of them one after the other (t and t + N / 32), so a 16384-point window is ONE workgroup of 512 threads = two waves per SIMD with 256 registers each instead of four
// with 128: room for the next window's 32 input samples per thread in flight under this window's butterflies (PF), half as many waves at every barrier.  Counters of the
// 1024-thread form (profiles/r6_fftfilt_pmc_issue.json): the vector ALU busy 46 % of the kernel's time, every wave parked (barrier / waitcnt) 52 % of its life -- one
// workgroup per CU whose memory phase and butterflies do not overlap at all.
template <int N, bool PF, int MINWG, bool HOIST, int LPT = 1>
__global__ __launch_bounds__(N / 16 / LPT, MINWG) void k_fftfilt_lds(const float2 *__restrict__ in, size_t in_pitch, const float2 *__restrict__ hist, int k1p, int m_new,
                                                        int n_chunks, int n_windows, float2 *__restrict__ out, size_t out_pitch, const float2 *hperm,
                                                        const float2 *__restrict__ g_tw1, const float2 *__restrict__ g_tws)
{
    using G = FflGeom<N>;
    constexpr int TP = G::T / LPT;                                      // physical threads
    extern __shared__ float4 ffl_raw[];
    float2 *lds = reinterpret_cast<float2 *>(ffl_raw), *tws = lds + G::DATA;
    const int t = threadIdx.x;
    for (int i = t; i < G::TWN; i += TP) tws[i] = g_tws[i];
    float2 w1[LPT];
#pragma unroll
    for (int h = 0; h < LPT; h++) w1[h] = g_tw1[t + h * TP];
    const int V = N - k1p;
    const int per_xcd = (n_windows + 7) >> 3, xcd = blockIdx.x & 7, stride = gridDim.x >> 3;      // gridDim.x is a multiple of 8
    const int w_end = min(n_windows, (xcd + 1) * per_xcd);
    int w = xcd * per_xcd + (blockIdx.x >> 3);
    if (w >= w_end) return;
    float2 v[LPT][16]; ffl_f32x2 nx[LPT][16];
    {
        const int s = w / n_chunks, c = w - s * n_chunks;
#pragma unroll
        for (int h = 0; h < LPT; h++) ffl_load_window<N>(v[h], in + (size_t)s * in_pitch, hist + (size_t)s * k1p, c * V - k1p, k1p, m_new, t + h * TP);
    }
    for (; w < w_end; w += stride) {
        const int s = w / n_chunks, c = w - s * n_chunks;
        if (!HOIST) {                                                   // keep the loop-invariant twiddle powers and taps spectrum OUT of registers (residency over reuse)
#pragma unroll
            for (int h = 0; h < LPT; h++) asm volatile("" : "+v"(w1[h].x), "+v"(w1[h].y));
            asm volatile("" : "+s"(hperm));
        }
#pragma unroll
        for (int h = 0; h < LPT; h++) ffl_first<N>(v[h], lds, w1[h], t + h * TP);
        const int wn = w + stride;
        // the window prefetched: the next one if there is one and it lies inside the call's input (a stream's first window, which starts in the history, is loaded the
        // ordinary way below); otherwise this one again, ignored -- the fetches are unconditional so that no branch surrounds a value in flight
        int sn = wn / n_chunks, cn = wn - sn * n_chunks;
        const bool pre = PF && wn < w_end && cn > 0;
        if (!pre) { sn = s; cn = c; }
        if (PF) {
#pragma unroll
            for (int h = 0; h < LPT; h++) ffl_prefetch_issue<N>(nx[h], in + (size_t)sn * in_pitch, cn * V - k1p, m_new, t + h * TP);
        }
        ffl_barrier();
#define FFL_PHASE(K) { _Pragma("unroll") for (int h = 0; h < LPT; h++) FflMidPhases<N>::template run<K>(lds, tws, hperm, t + h * TP); ffl_barrier(); }
        FFL_PHASE(0) FFL_PHASE(1) FFL_PHASE(2)
        if constexpr (FflMidPhases<N>::COUNT == 5) { FFL_PHASE(3) FFL_PHASE(4) }
#undef FFL_PHASE
        // results n = k1p .. N-1 of the window are outputs c V + (n - k1p): descriptor based at output c V, range = what is left of the call
        const unsigned long long by = (unsigned long long)(out + (size_t)s * out_pitch + (size_t)c * V);
        const ffl_i32x4 ry = {(int)(unsigned)by, (int)((by >> 32) & 0xffffu), (m_new - c * V) * 8, 0x00020000};
#pragma unroll
        for (int h = 0; h < LPT; h++) {
            ffl_last<N>(v[h], lds, w1[h], t + h * TP);
            if (h + 1 == LPT) ffl_barrier();                          // the next window's first stage overwrites the exchange buffer
            const int vy = (t + h * TP - k1p) * 8;                      // negative (the window's overlap part) -> dropped
#pragma unroll
            for (int j = 0; j < 16; j++) { const ffl_f32x2 r = {v[h][j].x, v[h][j].y}; ffl_buf_store(r, ry, vy + G::T * 8 * j, 0, 0); }
        }
        if (PF) {                                                       // the loads were issued before every store of this window: all 16 LPT stores may stay in flight
            if constexpr (LPT == 1) ffl_prefetch_ready<16>(nx[0]);
            else { ffl_prefetch_ready<32>(nx[0]); ffl_prefetch_ready<32>(nx[LPT - 1]); }
        }
        if (pre) {
#pragma unroll
            for (int h = 0; h < LPT; h++)
#pragma unroll
                for (int j = 0; j < 16; j++) v[h][j] = make_float2(nx[h][j].x, nx[h][j].y);
        } else if (wn < w_end) {
            const int s2 = wn / n_chunks, c2 = wn - s2 * n_chunks;
#pragma unroll
            for (int h = 0; h < LPT; h++) ffl_load_window<N>(v[h], in + (size_t)s2 * in_pitch, hist + (size_t)s2 * k1p, c2 * V - k1p, k1p, m_new, t + h * TP);
        }
    }
}

// the last k1p input samples of (history ++ this call's input) become the next call's history
__global__ __launch_bounds__(256) void k_fftfilt_hist(const float2 *__restrict__ in, size_t in_pitch, const float2 *__restrict__ hist_old, float2 *__restrict__ hist_new,
                                                      int k1p, long m_new)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= k1p) return;
    const size_t s = blockIdx.y;
    const long p = m_new - k1p + i;
    hist_new[s * k1p + i] = p >= 0 ? in[s * in_pitch + p] : hist_old[s * k1p + (k1p + p)];
}

// double-precision transform of the zero-padded taps on the host (once per set_taps; csdr.c:1869-1871 does it in float with the FFT library)
void host_dft_pow2(std::vector<double> &re, std::vector<double> &im)
{
    const size_t n = re.size();
    for (size_t i = 1, j = 0; i < n; i++) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        for (size_t k = 0; k < len / 2; k++) {
            const double ang = -2.0 * M_PI * (double)k / (double)len, wr = cos(ang), wi = sin(ang);
            for (size_t i = k; i < n; i += len) {
                const size_t j = i + len / 2;
                const double xr = re[j] * wr - im[j] * wi, xi = re[j] * wi + im[j] * wr;
                re[j] = re[i] - xr; im[j] = im[i] - xi; re[i] += xr; im[i] += xi;
            }
        }
    }
}

template <int N>
void ffl_host_tables(const cf32 *taps, int taps_len, std::vector<float2> &hperm, std::vector<float2> &tw1, std::vector<float2> &tws)
{
    using G = FflGeom<N>;
    std::vector<double> re(N, 0.0), im(N, 0.0);
    for (int k = 0; k < taps_len; k++) { re[k] = taps[k].i; im[k] = taps[k].q; }
    host_dft_pow2(re, im);
    hperm.resize(N);
    for (int slot = 0; slot < 16; slot++)
        for (int t = 0; t < G::T; t++) {
            const int f = ffl_freq_of_position<N>(ffl_slot_position<N>(slot, t));
            hperm[(size_t)slot * G::T + t] = make_float2((float)(re[f] / N), (float)(im[f] / N));
        }
    tw1.resize(G::T); tws.resize(G::TWN);
    for (int t = 0; t < G::T; t++) { const double a = -2.0 * M_PI * t / N; tw1[t] = make_float2((float)cos(a), (float)sin(a)); }
    for (int e = 0; e < G::TWN; e++) { const double a = -2.0 * M_PI * e / G::TWN; tws[e] = make_float2((float)cos(a), (float)sin(a)); }
}

// the kernel's algorithm on the CPU: same stage functions, one "thread" after the other, phases in order
template <int N>
void ffl_host_run(const cf32 *taps, int taps_len, const cf32 *x, long m_new, cf32 *y)
{
    using G = FflGeom<N>;
    std::vector<float2> hperm, tw1, tws; ffl_host_tables<N>(taps, taps_len, hperm, tw1, tws);
    const int k1p = (taps_len - 1 + 15) & ~15, V = N - k1p;
    const long n_chunks = (m_new + V - 1) / V;
    std::vector<float2> lds(G::DATA);
    for (long c = 0; c < n_chunks; c++) {
        const long w0 = c * V - k1p;
        for (int t = 0; t < G::T; t++) {
            float2 v[16];
            for (int j = 0; j < 16; j++) { const long p = w0 + t + G::T * j; v[j] = (p < 0 || p >= m_new) ? make_float2(0.f, 0.f) : make_float2(x[p].i, x[p].q); }
            ffl_first<N>(v, lds.data(), tw1[t], t);
        }
        for (int t = 0; t < G::T; t++) FflMidPhases<N>::template run<0>(lds.data(), tws.data(), hperm.data(), t);
        for (int t = 0; t < G::T; t++) FflMidPhases<N>::template run<1>(lds.data(), tws.data(), hperm.data(), t);
        for (int t = 0; t < G::T; t++) FflMidPhases<N>::template run<2>(lds.data(), tws.data(), hperm.data(), t);
        if constexpr (FflMidPhases<N>::COUNT == 5) {
            for (int t = 0; t < G::T; t++) FflMidPhases<N>::template run<3>(lds.data(), tws.data(), hperm.data(), t);
            for (int t = 0; t < G::T; t++) FflMidPhases<N>::template run<4>(lds.data(), tws.data(), hperm.data(), t);
        }
        for (int t = 0; t < G::T; t++) {
            float2 v[16];
            ffl_last<N>(v, lds.data(), tw1[t], t);
            for (int j = 0; j < 16; j++) {
                const int n = t + G::T * j; const long o = c * V + n - k1p;
                if (n >= k1p && o < m_new) y[o] = cf32{v[j].x, v[j].y};
            }
        }
    }
}

} // namespace

// ------------------------------------------------------------------------------------------------ one wave per window (4096 points)
#include "fftfilt_wave.hpp"

namespace {

// Lane l of a wave holds the rows n1 = fw_pi(l) of the 64 x 64 window (both in the time half, n = n1 + 64 n2, and in the frequency half, k = 64 k1 + k2 with
// k2 = fw_pi(l)): the order in which 16-byte loads of two neighbouring samples, halves swapped between lanes l and l + 32, leave them (see fw_rows_to_regs).
FFL_HD constexpr int fw_pi(int l) { return 2 * (l & 31) + (l >> 5); }
FFL_HD constexpr int fw_pi_inv(int r) { return (r >> 1) + 32 * (r & 1); }
// taps spectrum in the order pass 1 asks for it: radix-16 group g (g = k1 & 3), pair m of its outputs k1 = g + 4 (2 m), g + 4 (2 m + 1), lane l: 16 bytes
FFL_HD constexpr int fw_h_index(int k1, int l) { return (((k1 & 3) * 8 + (k1 >> 3)) * 64 + l) * 2 + ((k1 >> 2) & 1); }

// hw[fw_h_index(k1, lane)] = H[64 k1 + fw_pi(lane)] / N;  tw[e * 64 + lane]: e < 3: W^(n (e + 1)), e >= 3: W^(4 n (e - 2)), n = fw_pi(lane), W = exp(-2 pi i / 4096)
void fw_host_tables(const cf32 *taps, int taps_len, std::vector<float2> &hw, std::vector<float2> &tw)
{
    std::vector<double> re(FW_N, 0.0), im(FW_N, 0.0);
    for (int k = 0; k < taps_len; k++) { re[k] = taps[k].i; im[k] = taps[k].q; }
    host_dft_pow2(re, im);
    hw.resize(FW_N); tw.resize((size_t)FW_TWE * 64);
    for (int k1 = 0; k1 < 64; k1++)
        for (int l = 0; l < 64; l++) { const int f = 64 * k1 + fw_pi(l); hw[fw_h_index(k1, l)] = make_float2((float)(re[f] / FW_N), (float)(im[f] / FW_N)); }
    for (int e = 0; e < FW_TWE; e++)
        for (int l = 0; l < 64; l++) {
            const int p = e < 3 ? e + 1 : 4 * (e - 2);
            const double a = -2.0 * M_PI * (double)(fw_pi(l) * p) / FW_N;
            tw[(size_t)e * 64 + l] = make_float2((float)cos(a), (float)sin(a));
        }
}

// wave-private 64 x 64 transpose of one float per (lane, register): with rows and columns numbered by fw_pi, register r of the lane of row a -> register a of the lane
// of row r.  Lane l writes register r to LDS row l, column fw_pi_inv(r); lane l reads column l of every row p into register fw_pi(p): both sides conflict free
// (row pitch 65), the permutation costs nothing (register names).  LDS operations of one wave execute in order, so the reads see the writes without a barrier; the
// wavefront-scope fences only keep the COMPILER from moving a read above a write it cannot see the other lanes make.
template <bool IM> __device__ __forceinline__ void fw_transpose_half(fw_pk2 (&v)[64], float *L, int lane)
{
    float *wr = L + lane * FW_LP;
#pragma unroll
    for (int r = 0; r < 64; r++) wr[fw_pi_inv(r)] = IM ? v[r].y : v[r].x;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const float *rd = L + lane;
#pragma unroll
    for (int p = 0; p < 64; p++) { const float x = rd[p * FW_LP]; if (IM) v[fw_pi(p)].y = x; else v[fw_pi(p)].x = x; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void fw_transpose(fw_pk2 (&v)[64], float *L, int lane) { fw_transpose_half<false>(v, L, lane); fw_transpose_half<true>(v, L, lane); }

// The device forms of the passes.  A wave's 256 registers hold one window (128) and little else, and a wave cannot prefetch what it has no room for -- so what the
// first version waited for, it waited for in full: the taps spectrum as 64 load-wait-multiply round trips to the L2 (25 of its 37 us per window), then, batched, still
// 3 us; the next window's samples 8 us.  And a wave has 63 memory operations in flight at most: 64 stores + 64 loads of 8 bytes per lane made the loads wait for the
// stores to retire (two memory latencies, ~4 us each, per window, wherever the schedule put them).  Now:
//  * 16 bytes per lane and operation -- two neighbouring samples -- for window, results and spectrum: 32 + 32 + 32 operations per window.  A 16-byte load gives lane l
//    the samples 2 l, 2 l + 1 (+ 128 j); one v_permlane32_swap per register pair (lanes >= 32 hand their first sample to lanes < 32 for the second one of those) turns
//    that into rows n1 = fw_pi(l) with two consecutive j per operation, and the same swap undoes it in front of the stores;
//  * pass 1: the spectrum values of a radix-16 group are requested one group ahead, by hand (two sets of 16 registers), and waited for behind the group's butterflies;
//  * the twiddle bases live in LDS (one table per workgroup), not in 36 registers: that is what pays for the second set;
//  * pass 3: the moment two radix-16 groups' 32 result rows are stored, the same 32 rows of the NEXT window are requested into the registers just freed.  Rows of
//    groups g, g + 1 are j = g, g + 1 (mod 4), which is what the next pass 0's first step (radix 4 over rows n2, n2 + 16, n2 + 32, n2 + 48) consumes together.
typedef float ffl_f32x4 __attribute__((ext_vector_type(4)));
__device__ ffl_f32x4 ffl_buf_load4(ffl_i32x4 rsrc, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ void ffl_buf_store4(ffl_f32x4 v, ffl_i32x4 rsrc, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4f32");

// P = (sample 2 l, rows j), Q = (sample 2 l + 1, rows j) in lanes l < 32 / (2 l - 64, rows j + 1), (2 l - 63, rows j + 1) in lanes l >= 32  <->
// P = (n1, row j), Q = (n1, row j + 1) with n1 = fw_pi(l): lanes [32, 64) of P change places with lanes [0, 32) of Q.  Its own inverse.
__device__ __forceinline__ void fw_swap_halves(fw_pk2 &P, fw_pk2 &Q)
{
    const auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(P.x), __float_as_uint(Q.x), false, false);
    const auto ry = __builtin_amdgcn_permlane32_swap(__float_as_uint(P.y), __float_as_uint(Q.y), false, false);
    P = fw_pk2{__uint_as_float(rx[0]), __uint_as_float(ry[0])}; Q = fw_pk2{__uint_as_float(rx[1]), __uint_as_float(ry[1])};
}

__device__ __forceinline__ void fw_h_issue(ffl_f32x4 (&h)[8], ffl_i32x4 rh, int voff, int g)
{
#pragma unroll
    for (int m = 0; m < 8; m++) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(h[m]) : "v"(voff), "s"(rh), "s"((g * 8 + m) * 1024) : "memory");
}
template <int INFLIGHT> __device__ __forceinline__ void fw_h_ready(ffl_f32x4 (&h)[8])
{
    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(h[4]), "+v"(h[5]), "+v"(h[6]), "+v"(h[7]) : "n"(INFLIGHT));
}
__device__ __forceinline__ void fw_h_mul(fw_pk2 (&v)[64], const ffl_f32x4 (&h)[8], int g)                          // group g: v[16 g + k2] = X[k1 = g + 4 k2]
{
#pragma unroll
    for (int m = 0; m < 8; m++) {                                       // libcsdr.c:826-830 (and 836-839: the 1/N is in the table)
        v[16 * g + 2 * m] = fw_pk_cmul<false>(v[16 * g + 2 * m], fw_pk2{h[m].x, h[m].y});
        v[16 * g + 2 * m + 1] = fw_pk_cmul<false>(v[16 * g + 2 * m + 1], fw_pk2{h[m].z, h[m].w});
    }
}
__device__ __forceinline__ void fw_pass1_dev(fw_pk2 (&v)[64], const float2 *hw, int lane)
{
    const unsigned long long bh = (unsigned long long)hw;
    const ffl_i32x4 rh = {(int)(unsigned)bh, (int)((bh >> 32) & 0xffffu), FW_N * 8, 0x00020000};
    const int voff = lane * 16;
    ffl_f32x4 ha[8], hb[8];
    fw_h_issue(ha, rh, voff, 0);
    fw_pk_dft64_head<false>(v);
    fw_h_issue(hb, rh, voff, 1);
    fw_pk_dft16<0, false>(v); fw_h_ready<8>(ha); fw_h_mul(v, ha, 0); fw_h_issue(ha, rh, voff, 2);
    fw_pk_dft16<16, false>(v); fw_h_ready<8>(hb); fw_h_mul(v, hb, 1); fw_h_issue(hb, rh, voff, 3);
    fw_pk_dft16<32, false>(v); fw_h_ready<8>(ha); fw_h_mul(v, ha, 2);
    fw_pk_dft16<48, false>(v); fw_h_ready<0>(hb); fw_h_mul(v, hb, 3);
    fw_pk_dft64_tail(v);
}

// the lane's twiddle bases from the workgroup's table in LDS (tl = table + lane): the 18 values are live inside this function only
template <bool CONJ> __device__ __forceinline__ void fw_twiddle_lds(fw_pk2 (&v)[64], const fw_pk2 *tl)
{
    fw_pk2 tw[FW_TWE];
#pragma unroll
    for (int e = 0; e < FW_TWE; e++) tw[e] = tl[e * 64];
    fw_pk_twiddle<CONJ>(v, tw);
}

struct FwRows {                                                          // where a window's samples are: buffer descriptor + the lane's byte offset of sample 2 lane
    ffl_i32x4 r; int v0;
};
// half h of a window's rows: the row pairs (4 m + 2 h, 4 m + 2 h + 1), m = 0..15 -- nx[2 m + h] = samples 2 lane, 2 lane + 1 (+ 128 (2 m + h))
__device__ __forceinline__ void fw_load_half(ffl_f32x4 (&nx)[32], const FwRows &x, int h)
{
#pragma unroll
    for (int m = 0; m < 16; m++) nx[2 * m + h] = ffl_buf_load4(x.r, x.v0 + 1024 * (2 * m + h), 0, 0);
}
__device__ __forceinline__ void fw_rows_to_regs(fw_pk2 (&v)[64], const ffl_f32x4 (&nx)[32])
{
#pragma unroll
    for (int jp = 0; jp < 32; jp++) {
        fw_pk2 P = {nx[jp].x, nx[jp].y}, Q = {nx[jp].z, nx[jp].w};
        fw_swap_halves(P, Q);
        v[2 * jp] = P; v[2 * jp + 1] = Q;
    }
}
// pass 3, the stores of this window and the loads of the next one, half by half (group k1's outputs are the rows k1 + 4 k2)
template <int H> __device__ __forceinline__ void fw_pass3_half(fw_pk2 (&v)[64], const FwRows &y, ffl_f32x4 (&nx)[32], const FwRows &xn)
{
    constexpr int h = H;
    {
        fw_pk_dft16<32 * H, true>(v); fw_pk_dft16<32 * H + 16, true>(v);
#pragma unroll
        for (int m = 0; m < 16; m++) {                                  // rows 4 m + 2 h (group 2 h, k2 = m) and 4 m + 2 h + 1 (group 2 h + 1, k2 = m)
            fw_pk2 P = v[16 * (2 * h) + m], Q = v[16 * (2 * h + 1) + m];
            fw_swap_halves(P, Q);
            const ffl_f32x4 r = {P.x, P.y, Q.x, Q.y};
            ffl_buf_store4(r, y.r, y.v0 + 1024 * (2 * m + h), 0, 0);
        }
        fw_load_half(nx, xn, h);
    }
}
__device__ __forceinline__ void fw_pass3_dev(fw_pk2 (&v)[64], const FwRows &y, ffl_f32x4 (&nx)[32], const FwRows &xn)
{
    fw_pk_dft64_head<true>(v);
    fw_pass3_half<0>(v, y, nx, xn); fw_pass3_half<1>(v, y, nx, xn);
}

#ifdef FW_PROF
// -DFW_PROF (tools/probes/fw_prof.py): 100-MHz clock stamps between the phases of wave 0 of every workgroup, summed; every stamp pins the window's registers so that
// no arithmetic drifts across it.
__device__ unsigned long long g_fw_prof[8];
__device__ __forceinline__ void fw_pin(fw_pk2 (&v)[64])
{
#pragma unroll
    for (int j = 0; j < 64; j += 8)
        asm volatile("" : "+v"(v[j].x), "+v"(v[j].y), "+v"(v[j + 1].x), "+v"(v[j + 1].y), "+v"(v[j + 2].x), "+v"(v[j + 2].y), "+v"(v[j + 3].x), "+v"(v[j + 3].y),
                          "+v"(v[j + 4].x), "+v"(v[j + 4].y), "+v"(v[j + 5].x), "+v"(v[j + 5].y), "+v"(v[j + 6].x), "+v"(v[j + 6].y), "+v"(v[j + 7].x), "+v"(v[j + 7].y) :: "memory");
}
#define FW_T(k) { fw_pin(v); long long t_now; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_now) :: "memory"); fw_pin(v); prof[k] += t_now - t_prev; t_prev = t_now; }
#else
#define FW_T(k)
#endif

// Windows are dealt like the 256-thread kernel's: every XCD a contiguous range, consecutive waves consecutive windows (the taps - 1 samples two neighbours share come
// from that XCD's L2).  Edges by the buffer range check: samples in front of the call's input come from the history (second descriptor, the stream's first window only),
// samples behind its end read as zero, results outside [0, m_new) are dropped.  m_new is even (fw_launch), so a 16-byte access never straddles an edge.
__global__ __launch_bounds__(64 * FW_WAVES, 2) void k_fftfilt_wave(const float2 *__restrict__ in, size_t in_pitch, const float2 *__restrict__ hist, int k1p, int m_new,
                                                                    int n_chunks, int n_windows, float2 *__restrict__ out, size_t out_pitch, const float2 *__restrict__ hw,
                                                                    const float2 *__restrict__ g_tw)
{
    extern __shared__ float4 ffl_raw[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // (wave-uniform: window, stream and descriptors stay in scalar registers)
    float *L = reinterpret_cast<float *>(ffl_raw) + wv * 64 * FW_LP;
    float2 *twl = reinterpret_cast<float2 *>(reinterpret_cast<float *>(ffl_raw) + FW_WAVES * 64 * FW_LP);
    for (int i = threadIdx.x; i < FW_TWE * 64; i += 64 * FW_WAVES) twl[i] = g_tw[i];
    __syncthreads();                                                    // the kernel's only barrier
    const fw_pk2 *tl = reinterpret_cast<const fw_pk2 *>(twl) + lane;
    const int V = FW_N - k1p;
    const int per_xcd = (n_windows + 7) >> 3, xcd = blockIdx.x & 7, stride = (gridDim.x >> 3) * FW_WAVES;      // gridDim.x is a multiple of 8
    const int w_end = min(n_windows, (xcd + 1) * per_xcd);
    int w = xcd * per_xcd + (blockIdx.x >> 3) * FW_WAVES + wv;
    if (w >= w_end) return;
    auto rows_in = [&](int win) {                                       // window `win` of the input (negative offset -> out of range as unsigned -> 0)
        const int s = win / n_chunks, c = win - s * n_chunks;
        const unsigned long long bx = (unsigned long long)(in + (size_t)s * in_pitch);
        return FwRows{ffl_i32x4{(int)(unsigned)bx, (int)((bx >> 32) & 0xffffu), m_new * 8, 0x00020000}, (c * V - k1p + 2 * lane) * 8};
    };
    ffl_f32x4 nx[32];
    {
        const FwRows x0 = rows_in(w);
        fw_load_half(nx, x0, 0); fw_load_half(nx, x0, 1);
    }
#ifdef FW_PROF
    long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = wall_clock64();
#endif
    for (;;) {
        const int s = w / n_chunks, c = w - s * n_chunks, w0 = c * V - k1p;
        fw_pk2 v[64];
        fw_rows_to_regs(v, nx);
        if (w0 < 0) {                                                   // uniform: the stream's first window
            const unsigned long long bh = (unsigned long long)(hist + (size_t)s * k1p);
            const ffl_i32x4 rh = {(int)(unsigned)bh, (int)((bh >> 32) & 0xffffu), k1p * 8, 0x00020000};
            const int vh = (k1p + w0 + fw_pi(lane)) * 8;
#pragma unroll
            for (int j = 0; j < 64; j++) { const ffl_f32x2 r = ffl_buf_load(rh, vh + 512 * j, 0, 0); v[j].x += r.x; v[j].y += r.y; }
        }
#ifdef FW_PROF
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        FW_T(0)
        fw_pk_dft64<false>(v); fw_twiddle_lds<false>(v, tl);
        FW_T(1)
        fw_transpose(v, L, lane);
        FW_T(2)
        fw_pass1_dev(v, hw, lane);
        FW_T(3)
        fw_pk_dft64<true>(v); fw_twiddle_lds<true>(v, tl);
        FW_T(4)
        fw_transpose(v, L, lane);
        FW_T(5)
        // results n = k1p .. N-1 of the window are outputs c V + (n - k1p): descriptor based at output c V, range = what is left of the call; samples in front of
        // k1p (the window's overlap part) have negative offsets and are dropped
        const unsigned long long by = (unsigned long long)(out + (size_t)s * out_pitch + (size_t)c * V);
        const FwRows y = {ffl_i32x4{(int)(unsigned)by, (int)((by >> 32) & 0xffffu), (m_new - c * V) * 8, 0x00020000}, (2 * lane - k1p) * 8};
        const int wn = w + stride;
        const bool more = wn < w_end;
        const FwRows xn = rows_in(more ? wn : w);                       // (the last window: this one again, ignored -- no branch around values in flight)
        fw_pass3_dev(v, y, nx, xn);
        FW_T(6)
#ifdef FW_PROF
        prof[7] += 1;
#endif
        if (!more) break;
        w = wn;
    }
#ifdef FW_PROF
    if (threadIdx.x == 0) for (int k = 0; k < 8; k++) atomicAdd(&g_fw_prof[k], (unsigned long long)prof[k]);
#endif
}

// the wave kernel's algorithm on the CPU: same pass functions and tables, lane after lane, the transposes as the index maps of fw_transpose_half
void fw_host_run(const cf32 *taps, int taps_len, const cf32 *x, long m_new, cf32 *y)
{
    std::vector<float2> hw, twt; fw_host_tables(taps, taps_len, hw, twt);
    const int k1p = (taps_len - 1 + 15) & ~15, V = FW_N - k1p;
    const long n_chunks = (m_new + V - 1) / V;
    std::vector<float2> a((size_t)64 * 64), b((size_t)64 * 64);
    auto lane_tw = [&](int l, float2 (&tw)[FW_TWE]) { for (int e = 0; e < FW_TWE; e++) tw[e] = twt[(size_t)e * 64 + l]; };
    for (long c = 0; c < n_chunks; c++) {
        const long w0 = c * V - k1p;
        for (int l = 0; l < 64; l++) {
            float2 v[64], tw[FW_TWE]; lane_tw(l, tw);
            for (int j = 0; j < 64; j++) { const long p = w0 + fw_pi(l) + 64 * j; v[j] = (p < 0 || p >= m_new) ? make_float2(0.f, 0.f) : make_float2(x[p].i, x[p].q); }
            fw_pass0(v, tw);
            for (int r = 0; r < 64; r++) a[(size_t)l * 64 + fw_pi_inv(r)] = v[r];
        }
        for (int l = 0; l < 64; l++) {
            float2 v[64], h[64], tw[FW_TWE]; lane_tw(l, tw);
            for (int p = 0; p < 64; p++) v[fw_pi(p)] = a[(size_t)p * 64 + l];
            for (int k1 = 0; k1 < 64; k1++) h[k1] = hw[fw_h_index(k1, l)];
            fw_pass1(v, h, 1);
            fw_pass2(v, tw);
            for (int r = 0; r < 64; r++) b[(size_t)l * 64 + fw_pi_inv(r)] = v[r];
        }
        for (int l = 0; l < 64; l++) {
            float2 v[64];
            for (int p = 0; p < 64; p++) v[fw_pi(p)] = b[(size_t)p * 64 + l];
            fw_pass3(v);
            for (int j = 0; j < 64; j++) {
                const int n = fw_pi(l) + 64 * j; const long o = c * V + n - k1p;
                if (n >= k1p && o < m_new) y[o] = cf32{v[j].x, v[j].y};
            }
        }
    }
}

} // namespace

// ------------------------------------------------------------------------------------------------ a team of waves per window (8192 / 16384 points)
#include "fftfilt_team.hpp"

namespace {

template <int M>
void ft_host_tables(const cf32 *taps, int taps_len, std::vector<float2> &hw, std::vector<float2> &tw1, std::vector<float2> &tw2)
{
    using G = FtGeom<M>;
    std::vector<double> re(G::N, 0.0), im(G::N, 0.0);
    for (int k = 0; k < taps_len; k++) { re[k] = taps[k].i; im[k] = taps[k].q; }
    host_dft_pow2(re, im);
    hw.resize(G::N); tw1.resize((size_t)FW_TWE * G::T); tw2.resize((size_t)FW_TWE * M);
    for (int p = 0; p < G::T; p++)
        for (int kc = 0; kc < 64; kc++) {
            const int ka = p >> G::LOGM, kd = ft_slot_kd(p & (M - 1), M), f = ka + 64 * (kc + 64 * kd);
            hw[ft_h_index<M>(kc, p)] = make_float2((float)(re[f] / G::N), (float)(im[f] / G::N));
        }
    for (int e = 0; e < FW_TWE; e++) {
        const int pw = e < 3 ? e + 1 : 4 * (e - 2);
        for (int p = 0; p < G::T; p++) { const double a = -2.0 * M_PI * (double)((long)ft_logical(p) * pw) / G::N; tw1[(size_t)e * G::T + p] = make_float2((float)cos(a), (float)sin(a)); }
        for (int m = 0; m < M; m++) { const double a = -2.0 * M_PI * (double)(m * pw) / G::T; tw2[(size_t)e * M + m] = make_float2((float)cos(a), (float)sin(a)); }
    }
}

// the team kernel's algorithm on the CPU: its tables (ft_host_tables), its index maps (ft_logical, the exchanges' addresses t <-> (k_a, m), ft_slot_kd, ft_h_index) and the
// scalar forms of its passes, thread after thread; the radix-M step over the M lanes of a group written out as the kernel's two butterfly stages
template <int M>
void ft_host_run(const cf32 *taps, int taps_len, const cf32 *x, long m_new, cf32 *y)
{
    using G = FtGeom<M>;
    constexpr int T = G::T, N = G::N;
    std::vector<float2> hw, t1, t2; ft_host_tables<M>(taps, taps_len, hw, t1, t2);
    const int k1p = (taps_len - 1 + 15) & ~15, V = N - k1p;
    const long n_chunks = (m_new + V - 1) / V;
    std::vector<float2> lds((size_t)64 * G::P), regs((size_t)T * 64);
    auto tw_of = [&](const std::vector<float2> &tab, int stride, int idx, float2 (&tw)[FW_TWE]) { for (int e = 0; e < FW_TWE; e++) tw[e] = tab[(size_t)e * stride + idx]; };
    auto bfly = [](float2 x, float2 p, float sig) { return make_float2(p.x + sig * x.x, p.y + sig * x.y); };
    auto radix_lanes = [&](float2 (&q)[M], bool inv) {                   // q[m]: the M lanes of a group, in place (ft_radix_lanes)
        float2 a[M];
        if (M == 2) { a[0] = bfly(q[0], q[1], 1.f); a[1] = bfly(q[1], q[0], -1.f); }
        else if (!inv) {
            float2 s[4]; for (int m = 0; m < 4; m++) s[m] = bfly(q[m], q[m ^ 2], (m & 2) ? -1.f : 1.f);
            s[3] = make_float2(s[3].y, -s[3].x);
            for (int m = 0; m < 4; m++) a[m] = bfly(s[m], s[m ^ 1], (m & 1) ? -1.f : 1.f);
        } else {
            float2 s[4]; for (int m = 0; m < 4; m++) s[m] = bfly(q[m], q[m ^ 1], (m & 1) ? -1.f : 1.f);
            s[3] = make_float2(-s[3].y, s[3].x);
            for (int m = 0; m < 4; m++) a[m] = bfly(s[m], s[m ^ 2], (m & 2) ? -1.f : 1.f);
        }
        for (int m = 0; m < M; m++) q[m] = a[m];
    };
    for (long c = 0; c < n_chunks; c++) {
        const long w0 = c * V - k1p;
        for (int p = 0; p < T; p++) {                                    // pass 0 and the exchange's writes: a[k_a P + t]
            const int t = ft_logical(p);
            float2 v[64], tw[FW_TWE]; tw_of(t1, T, p, tw);
            for (int j = 0; j < 64; j++) { const long n = w0 + t + (long)T * j; v[j] = (n < 0 || n >= m_new) ? make_float2(0.f, 0.f) : make_float2(x[n].i, x[n].q); }
            dft64<false>(v); fw_twiddle<false>(v, tw);
            for (int r = 0; r < 64; r++) lds[(size_t)r * G::P + t] = v[r];
        }
        for (int p = 0; p < T; p++) {                                    // reads b[M i], pass 1, x W_T^(m k_c)
            float2 v[64], tw[FW_TWE]; tw_of(t2, M, p & (M - 1), tw);
            for (int i = 0; i < 64; i++) v[i] = lds[(size_t)(p >> G::LOGM) * G::P + (p & (M - 1)) + M * i];
            dft64<false>(v); fw_twiddle<false>(v, tw);
            for (int r = 0; r < 64; r++) regs[(size_t)p * 64 + r] = v[r];
        }
        for (int g = 0; g < T / M; g++)                                  // radix M over the group's lanes, x spectrum, inverse
            for (int r = 0; r < 64; r++) {
                float2 q[M];
                for (int m = 0; m < M; m++) q[m] = regs[(size_t)(M * g + m) * 64 + r];
                radix_lanes(q, false);
                for (int m = 0; m < M; m++) q[m] = cmul(q[m], hw[ft_h_index<M>(r, M * g + m)]);
                radix_lanes(q, true);
                for (int m = 0; m < M; m++) regs[(size_t)(M * g + m) * 64 + r] = q[m];
            }
        for (int p = 0; p < T; p++) {                                    // x conj W_T^(m k_c), pass 2, the exchange back: writes b[M i]
            float2 v[64], tw[FW_TWE]; tw_of(t2, M, p & (M - 1), tw);
            for (int r = 0; r < 64; r++) v[r] = regs[(size_t)p * 64 + r];
            fw_twiddle<true>(v, tw); dft64<true>(v);
            for (int i = 0; i < 64; i++) lds[(size_t)(p >> G::LOGM) * G::P + (p & (M - 1)) + M * i] = v[i];
        }
        for (int p = 0; p < T; p++) {                                    // reads a[k_a P + t], x conj W_N^(t k_a), pass 3
            const int t = ft_logical(p);
            float2 v[64], tw[FW_TWE]; tw_of(t1, T, p, tw);
            for (int r = 0; r < 64; r++) v[r] = lds[(size_t)r * G::P + t];
            fw_twiddle<true>(v, tw); dft64<true>(v);
            for (int j = 0; j < 64; j++) {
                const int n = t + T * j; const long o = c * V + n - k1p;
                if (n >= k1p && o < m_new) y[o] = cf32{v[j].x, v[j].y};
            }
        }
    }
}

} // namespace

namespace csdr_amd {

struct FftfiltLds {
    int n, taps_len, k1p, n_streams;
    float2 *d_hperm, *d_tw1, *d_tws, *d_hist[2]; int flip;
    const char *last;                               // the window kernel the last call ran (a call's size and parity pick it)
    float2 *d_hw, *d_twl, *d_tw2; bool wave, team;  // the tables of the wave-per-window kernel (4096-point windows) / of the team kernel (8192, 16384)
    int mode;                                       // CSDR_AMD_FFTFILT_LDS_MODE (A/B: 5 = the kernels of rounds 2-5, 6 = the wave kernel at every call size), read at create
};

// window size for a filter of taps_len taps: the smallest plan that keeps >= 3/4 of every window as output; 0 = none fits (the caller keeps its other paths)
int fftfilt_lds_pick(int taps_len)
{
    if (getenv("CSDR_AMD_FFTFILT_LDS_OFF")) return 0;
    if (const char *e = getenv("CSDR_AMD_FFTFILT_LDS_N")) { const int n = atoi(e); if ((n == 4096 || n == 8192 || n == 16384) && taps_len - 1 + 15 < n / 2) return n; }
    const int k1p = (taps_len - 1 + 15) & ~15;
    for (int n : {4096, 8192, 16384}) if (4 * k1p <= n) return n;
    return 0;
}

void fftfilt_lds_destroy(FftfiltLds *p)
{
    if (!p) return;
    (void)hipFree(p->d_hperm); (void)hipFree(p->d_tw1); (void)hipFree(p->d_tws); (void)hipFree(p->d_hist[0]); (void)hipFree(p->d_hist[1]);
    (void)hipFree(p->d_hw); (void)hipFree(p->d_twl); (void)hipFree(p->d_tw2);
    delete p;
}

int fftfilt_lds_set_taps(FftfiltLds *p, hipStream_t st, const cf32 *taps, int taps_len)
{
    std::vector<float2> hperm, tw1, tws;
    if (p->wave) {
        std::vector<float2> hw, twl; fw_host_tables(taps, taps_len, hw, twl);
        CSDR_HIP(hipStreamSynchronize(st));
        CSDR_HIP(hipMemcpy(p->d_hw, hw.data(), sizeof(float2) * hw.size(), hipMemcpyHostToDevice));
        CSDR_HIP(hipMemcpy(p->d_twl, twl.data(), sizeof(float2) * twl.size(), hipMemcpyHostToDevice));
    }                                                                   // (and the 256-thread kernel's tables: it takes the calls with an odd sample count)
    if (p->team) {
        std::vector<float2> hw, t1, t2;
        if (p->n == 8192) ft_host_tables<2>(taps, taps_len, hw, t1, t2); else ft_host_tables<4>(taps, taps_len, hw, t1, t2);
        CSDR_HIP(hipStreamSynchronize(st));
        CSDR_HIP(hipMemcpy(p->d_hw, hw.data(), sizeof(float2) * hw.size(), hipMemcpyHostToDevice));
        CSDR_HIP(hipMemcpy(p->d_twl, t1.data(), sizeof(float2) * t1.size(), hipMemcpyHostToDevice));
        CSDR_HIP(hipMemcpy(p->d_tw2, t2.data(), sizeof(float2) * t2.size(), hipMemcpyHostToDevice));
    }
    if (p->n == 4096) ffl_host_tables<4096>(taps, taps_len, hperm, tw1, tws);
    else if (p->n == 8192) ffl_host_tables<8192>(taps, taps_len, hperm, tw1, tws);
    else ffl_host_tables<16384>(taps, taps_len, hperm, tw1, tws);
    CSDR_HIP(hipStreamSynchronize(st));
    CSDR_HIP(hipMemcpy(p->d_hperm, hperm.data(), sizeof(float2) * hperm.size(), hipMemcpyHostToDevice));
    CSDR_HIP(hipMemcpy(p->d_tw1, tw1.data(), sizeof(float2) * tw1.size(), hipMemcpyHostToDevice));
    CSDR_HIP(hipMemcpy(p->d_tws, tws.data(), sizeof(float2) * tws.size(), hipMemcpyHostToDevice));
    return 0;
}

int fftfilt_lds_reset(FftfiltLds *p, hipStream_t st)
{
    CSDR_HIP(hipMemsetAsync(p->d_hist[0], 0, sizeof(float2) * (size_t)p->n_streams * (p->k1p + 16), st));
    CSDR_HIP(hipMemsetAsync(p->d_hist[1], 0, sizeof(float2) * (size_t)p->n_streams * (p->k1p + 16), st));
    p->flip = 0;
    return 0;
}

FftfiltLds *fftfilt_lds_create(hipStream_t st, int n, const cf32 *taps, int taps_len, int n_streams)
{
    FftfiltLds *p = new FftfiltLds();
    p->n = n; p->taps_len = taps_len; p->k1p = (taps_len - 1 + 15) & ~15; p->n_streams = n_streams; p->flip = 0;
    p->mode = getenv("CSDR_AMD_FFTFILT_LDS_MODE") ? atoi(getenv("CSDR_AMD_FFTFILT_LDS_MODE")) : 0;
    p->d_hperm = p->d_tw1 = p->d_tws = p->d_hist[0] = p->d_hist[1] = p->d_hw = p->d_twl = p->d_tw2 = nullptr; p->last = nullptr;
    p->wave = n == 4096 && (p->mode == 0 || p->mode == 6);        // 4096-point windows: one wave per window; CSDR_AMD_FFTFILT_LDS_MODE=5 (A/B): the 256-thread kernel of rounds 2-5
    p->team = (n == 8192 || n == 16384) && (p->mode == 0 || p->mode == 6);      // a team of 2 / 4 waves per window; CSDR_AMD_FFTFILT_LDS_MODE=5: the 512-thread kernels of rounds 2-6
    hipError_t e = hipMalloc((void **)&p->d_hperm, sizeof(float2) * n);
    if ((p->wave || p->team) && e == hipSuccess) e = hipMalloc((void **)&p->d_hw, sizeof(float2) * n);
    if ((p->wave || p->team) && e == hipSuccess) e = hipMalloc((void **)&p->d_twl, sizeof(float2) * FW_TWE * (n / 64));
    if (p->team && e == hipSuccess) e = hipMalloc((void **)&p->d_tw2, sizeof(float2) * FW_TWE * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_tw1, sizeof(float2) * (n / 16));
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_tws, sizeof(float2) * (n / 16));
    for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipMalloc((void **)&p->d_hist[i], sizeof(float2) * (size_t)n_streams * (p->k1p + 16));
    if (e != hipSuccess) { fail(e, "hipMalloc(fftfilt_lds)", __FILE__, __LINE__); fftfilt_lds_destroy(p); return nullptr; }
    if (fftfilt_lds_set_taps(p, st, taps, taps_len) || fftfilt_lds_reset(p, st)) { fftfilt_lds_destroy(p); return nullptr; }
    return p;
}

const char *fftfilt_lds_kernel_name(const FftfiltLds *p) { return p->last ? p->last : p->wave ? "k_fftfilt_wave" : p->team ? (p->n == 8192 ? "k_fftfilt_team<2>" : "k_fftfilt_team<4>") : p->n == 4096 ? "k_fftfilt_lds<4096>" : p->n == 8192 ? "k_fftfilt_lds<8192>" : "k_fftfilt_lds<16384>"; }
int fftfilt_lds_window(const FftfiltLds *p) { return p->n; }

template <int N, bool PF, int MINWG, bool HOIST, int LPT = 1>
static int ffl_launch(FftfiltLds *p, hipStream_t st, const cf32 *in, size_t in_pitch, long m_new, cf32 *out, size_t out_pitch)
{
    using G = FflGeom<N>;
    int rc = lds_attr_once((const void *)k_fftfilt_lds<N, PF, MINWG, HOIST, LPT>, G::LDS_BYTES); if (rc) return rc;
    const int V = N - p->k1p;
    const int n_chunks = (int)((m_new + V - 1) / V);
    const long n_windows = (long)n_chunks * p->n_streams;
    if (n_windows > 0x7fffffffL || m_new > (1L << 27)) return fail_msg(-3, "fftfilt: call too large (2^27 samples per stream at most)");
    constexpr int wpe = N / 16 / LPT / 64 / 4;                         // waves per SIMD of one workgroup (MINWG counts waves per SIMD)
    constexpr int by_regs = MINWG / wpe > 0 ? MINWG / wpe : 1, by_lds = (int)(160 * 1024 / G::LDS_BYTES);
    long grid = (long)current_device_cu_count() * (by_regs < by_lds ? by_regs : by_lds);
    if (grid > n_windows) grid = n_windows;
    grid = (grid + 7) & ~7L;
    hipLaunchKernelGGL((k_fftfilt_lds<N, PF, MINWG, HOIST, LPT>), dim3((unsigned)grid), dim3(G::T / LPT), G::LDS_BYTES, st, (const float2 *)in, in_pitch,
                       (const float2 *)p->d_hist[p->flip], p->k1p, (int)m_new, n_chunks, (int)n_windows, (float2 *)out, out_pitch, (const float2 *)p->d_hperm,
                       (const float2 *)p->d_tw1, (const float2 *)p->d_tws);
    CSDR_LAUNCH_CHECK();
    return 0;
}

static int fw_launch(FftfiltLds *p, hipStream_t st, const cf32 *in, size_t in_pitch, long m_new, cf32 *out, size_t out_pitch)
{
    constexpr size_t lds_bytes = (size_t)FW_WAVES * 64 * FW_LP * sizeof(float) + (size_t)FW_TWE * 64 * sizeof(float2);      // transposes + the workgroup's twiddle table
    int rc = lds_attr_once((const void *)k_fftfilt_wave, lds_bytes); if (rc) return rc;
    const int V = FW_N - p->k1p;
    const int n_chunks = (int)((m_new + V - 1) / V);
    const long n_windows = (long)n_chunks * p->n_streams;
    if (n_windows > 0x7fffffffL || m_new > (1L << 27)) return fail_msg(-3, "fftfilt: call too large (2^27 samples per stream at most)");
    long grid = (long)current_device_cu_count() * (8 / FW_WAVES);      // eight waves per CU: two per SIMD at 256 registers
    const long need = (n_windows + FW_WAVES - 1) / FW_WAVES;
    if (grid > need) grid = need;
    grid = (grid + 7) & ~7L;
    hipLaunchKernelGGL(k_fftfilt_wave, dim3((unsigned)grid), dim3(64 * FW_WAVES), lds_bytes, st, (const float2 *)in, in_pitch, (const float2 *)p->d_hist[p->flip], p->k1p,
                       (int)m_new, n_chunks, (int)n_windows, (float2 *)out, out_pitch, (const float2 *)p->d_hw, (const float2 *)p->d_twl);
    CSDR_LAUNCH_CHECK();
    return 0;
}

template <int M>
static int ft_launch(FftfiltLds *p, hipStream_t st, const cf32 *in, size_t in_pitch, long m_new, cf32 *out, size_t out_pitch)
{
    using G = FtGeom<M>;
    int rc = lds_attr_once((const void *)k_fftfilt_team<M>, G::LDS_BYTES); if (rc) return rc;
    const int V = G::N - p->k1p;
    const int n_chunks = (int)((m_new + V - 1) / V);
    const long n_windows = (long)n_chunks * p->n_streams;
    if (n_windows > 0x7fffffffL || m_new > (1L << 27)) return fail_msg(-3, "fftfilt: call too large (2^27 samples per stream at most)");
    long grid = (long)current_device_cu_count() * (8 / M);             // eight waves per CU: two per SIMD at 256 registers
    if (grid > n_windows) grid = n_windows;
    grid = (grid + 7) & ~7L;
    hipLaunchKernelGGL(k_fftfilt_team<M>, dim3((unsigned)grid), dim3(64 * M), G::LDS_BYTES, st, (const float2 *)in, in_pitch, (const float2 *)p->d_hist[p->flip], p->k1p,
                       (int)m_new, n_chunks, (int)n_windows, (float2 *)out, out_pitch, (const float2 *)p->d_hw, (const float2 *)p->d_twl, (const float2 *)p->d_tw2);
    CSDR_LAUNCH_CHECK();
    return 0;
}

// m_new new samples per stream in, m_new filtered samples out
int fftfilt_lds_process(FftfiltLds *p, hipStream_t st, const cf32 *in, size_t in_pitch, long m_new, cf32 *out, size_t out_pitch)
{
    if (m_new <= 0) return 0;
    int rc;
    // the wave kernel from four windows per wave on: below that its one or two rounds take a window's full latency with a quarter of the 256-thread kernel's waves on it
    // (1 stream x 64 blocks = 1345 windows: 0.028 against 0.024 ms; 5380 windows 0.075 / 0.067; 10760 equal; 21520 0.266 / 0.285 -- tools/probes/fftfilt_sizes.py)
    const bool wave_pays = p->wave && (p->mode == 6 || (m_new + (FW_N - p->k1p) - 1) / (FW_N - p->k1p) * p->n_streams >= 8192);
    const char *old_name = p->n == 4096 ? "k_fftfilt_lds<4096>" : p->n == 8192 ? "k_fftfilt_lds<8192>" : "k_fftfilt_lds<16384>";
    p->last = p->wave && !(m_new & 1) && wave_pays ? "k_fftfilt_wave" : p->team && !(m_new & 1) ? (p->n == 8192 ? "k_fftfilt_team<2>" : "k_fftfilt_team<4>") : old_name;
    if (p->wave && !(m_new & 1) && wave_pays) rc = fw_launch(p, st, in, in_pitch, m_new, out, out_pitch);      // (16-byte accesses: an even sample count; odd ones take the 256-thread kernel)
    else if (p->team && !(m_new & 1)) rc = p->n == 8192 ? ft_launch<2>(p, st, in, in_pitch, m_new, out, out_pitch) : ft_launch<4>(p, st, in, in_pitch, m_new, out, out_pitch);
    // The kernels of rounds 2-5 (odd sample counts, small calls at 4096 points, CSDR_AMD_FFTFILT_LDS_MODE=5), each in the one form that measured best (profiles/r2_notes.md,
    // r6_notes.md; the prefetch / residency variants that lost -- modes 1-4 of earlier rounds -- are gone): 4096 points: four resident workgroups per CU, no register
    // prefetch; 8192: one 512-thread workgroup that prefetches the next window and keeps twiddle powers and spectrum in registers; 16384: 512 threads x two logical threads.
    else if (p->n == 4096) rc = ffl_launch<4096, false, 4, false>(p, st, in, in_pitch, m_new, out, out_pitch);
    else if (p->n == 8192) rc = ffl_launch<8192, true, 1, true>(p, st, in, in_pitch, m_new, out, out_pitch);
    else rc = ffl_launch<16384, false, 1, false, 2>(p, st, in, in_pitch, m_new, out, out_pitch);
    if (rc) return rc;
    if (p->k1p > 0) {
        hipLaunchKernelGGL(k_fftfilt_hist, dim3(cdiv(p->k1p, 256), p->n_streams), dim3(256), 0, st, (const float2 *)in, in_pitch, (const float2 *)p->d_hist[p->flip],
                           p->d_hist[p->flip ^ 1], p->k1p, m_new);
        CSDR_LAUNCH_CHECK();
        p->flip ^= 1;
    }
    return 0;
}


} // namespace csdr_amd

#ifdef FW_PROF
extern "C" int csdr_amd_debug_fw_prof(unsigned long long *out8, int reset)
{
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_fw_prof), 64) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_fw_prof), z, 64) != hipSuccess) return -1; }
    return 0;
}
#endif

// Test hook (CPU, no device): the LDS kernel's algorithm -- same stage functions, same tables -- on m_new samples of one stream from the zero state.
extern "C" int csdr_amd_debug_fftfilt_lds(int n, const float *taps_iq, int taps_len, const float *x_iq, long m_new, float *y_iq)
{
    const cf32 *taps = reinterpret_cast<const cf32 *>(taps_iq), *x = reinterpret_cast<const cf32 *>(x_iq); cf32 *y = reinterpret_cast<cf32 *>(y_iq);
    if (n == -8192 || n == -16384) {                                    // the team form of the 8192- / 16384-point windows
        if (taps_len < 1 || ((taps_len - 1 + 15) & ~15) >= -n) return -3;
        if (n == -8192) ft_host_run<2>(taps, taps_len, x, m_new, y); else ft_host_run<4>(taps, taps_len, x, m_new, y);
        return 0;
    }
    if (n == -4096) {                                                   // the wave-per-window form of the 4096-point window
        if (taps_len < 1 || ((taps_len - 1 + 15) & ~15) >= FW_N) return -3;
        fw_host_run(taps, taps_len, x, m_new, y); return 0;
    }
    if (taps_len < 1 || ((taps_len - 1 + 15) & ~15) >= n) return -3;
    if (n == 4096) ffl_host_run<4096>(taps, taps_len, x, m_new, y);
    else if (n == 8192) ffl_host_run<8192>(taps, taps_len, x, m_new, y);
    else if (n == 16384) ffl_host_run<16384>(taps, taps_len, x, m_new, y);
    else return -3;
    return 0;
}
