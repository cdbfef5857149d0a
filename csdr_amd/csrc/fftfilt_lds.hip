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

namespace csdr_amd {

struct FftfiltLds {
    int n, taps_len, k1p, n_streams;
    float2 *d_hperm, *d_tw1, *d_tws, *d_hist[2]; int flip;
    int mode;                                       // CSDR_AMD_FFTFILT_LDS_MODE (A/B: prefetch / residency variant of the 4096-point kernel), read at create
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
    delete p;
}

int fftfilt_lds_set_taps(FftfiltLds *p, hipStream_t st, const cf32 *taps, int taps_len)
{
    std::vector<float2> hperm, tw1, tws;
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
    p->d_hperm = p->d_tw1 = p->d_tws = p->d_hist[0] = p->d_hist[1] = nullptr;
    hipError_t e = hipMalloc((void **)&p->d_hperm, sizeof(float2) * n);
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_tw1, sizeof(float2) * (n / 16));
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_tws, sizeof(float2) * (n / 16));
    for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipMalloc((void **)&p->d_hist[i], sizeof(float2) * (size_t)n_streams * (p->k1p + 16));
    if (e != hipSuccess) { fail(e, "hipMalloc(fftfilt_lds)", __FILE__, __LINE__); fftfilt_lds_destroy(p); return nullptr; }
    if (fftfilt_lds_set_taps(p, st, taps, taps_len) || fftfilt_lds_reset(p, st)) { fftfilt_lds_destroy(p); return nullptr; }
    return p;
}

const char *fftfilt_lds_kernel_name(const FftfiltLds *p) { return p->n == 4096 ? "k_fftfilt_lds<4096>" : p->n == 8192 ? "k_fftfilt_lds<8192>" : "k_fftfilt_lds<16384>"; }
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

// m_new new samples per stream in, m_new filtered samples out
int fftfilt_lds_process(FftfiltLds *p, hipStream_t st, const cf32 *in, size_t in_pitch, long m_new, cf32 *out, size_t out_pitch)
{
    if (m_new <= 0) return 0;
    int rc;
    // Measured on one box (profiles/r2_notes.md): 4096-point windows run best with four resident workgroups per CU and no register prefetch (0.318 ms per
    // 64 x 16 blocks; three workgroups 0.346, prefetching variants 0.33-0.36); 8192-point windows with one 512-thread workgroup that prefetches the next window
    // and keeps the twiddle powers and the taps spectrum in registers.  CSDR_AMD_FFTFILT_LDS_MODE=1 selects the prefetching variant for 4096 too.
    const int mode = p->mode;
    if (p->n == 4096) {
        if (mode == 1) rc = ffl_launch<4096, true, 2, true>(p, st, in, in_pitch, m_new, out, out_pitch);
        else if (mode == 3) rc = ffl_launch<4096, true, 3, false>(p, st, in, in_pitch, m_new, out, out_pitch);
        else if (mode == 4) rc = ffl_launch<4096, true, 4, false>(p, st, in, in_pitch, m_new, out, out_pitch);
        else rc = ffl_launch<4096, false, 4, false>(p, st, in, in_pitch, m_new, out, out_pitch);
    } else if (p->n == 8192) {
        if (mode == 2) rc = ffl_launch<8192, false, 2, false>(p, st, in, in_pitch, m_new, out, out_pitch);
        else if (mode == 4) rc = ffl_launch<8192, false, 4, false>(p, st, in, in_pitch, m_new, out, out_pitch);      // (the second __launch_bounds__ argument is waves per SIMD: 4 = 128 registers = TWO workgroups per CU)
        else rc = ffl_launch<8192, true, 1, true>(p, st, in, in_pitch, m_new, out, out_pitch);
    } else if (mode == 1) rc = ffl_launch<16384, false, 1, false>(p, st, in, in_pitch, m_new, out, out_pitch);      // (A/B: the 1024-thread form of rounds 2-5)
    // 512 threads, two logical threads each, no prefetch: 0.490 against 0.513 ms (1024 threads) per 64 x 16 blocks at 4095 taps on one box.  WITH the next window's 32
    // samples per thread in flight (PF) the compiler, at 256 registers, serialises the centre phase's 32 loads of the taps spectrum: 0.601 ms -- not instantiated.
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

// Test hook (CPU, no device): the LDS kernel's algorithm -- same stage functions, same tables -- on m_new samples of one stream from the zero state.
extern "C" int csdr_amd_debug_fftfilt_lds(int n, const float *taps_iq, int taps_len, const float *x_iq, long m_new, float *y_iq)
{
    const cf32 *taps = reinterpret_cast<const cf32 *>(taps_iq), *x = reinterpret_cast<const cf32 *>(x_iq); cf32 *y = reinterpret_cast<cf32 *>(y_iq);
    if (taps_len < 1 || ((taps_len - 1 + 15) & ~15) >= n) return -3;
    if (n == 4096) ffl_host_run<4096>(taps, taps_len, x, m_new, y);
    else if (n == 8192) ffl_host_run<8192>(taps, taps_len, x, m_new, y);
    else if (n == 16384) ffl_host_run<16384>(taps, taps_len, x, m_new, y);
    else return -3;
    return 0;
}
