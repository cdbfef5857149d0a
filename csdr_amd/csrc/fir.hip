// fir.hip -- time-domain FIR kernels.
//   fir_decimate_cc   libcsdr.c:528-549  (real taps on interleaved complexf, decimation D)
//   deemphasis_nfm_ff libcsdr.c:1101-1128 (fixed real FIR on floats, no decimation)
//
// k_fir_poly (the default): one wave = one (stream, tile of 64*R consecutive outputs).  The input window is staged ONCE through LDS,
// but stored polyphase-decomposed: sample n goes to row n % D, column n / D.  Output o needs x[D*o + a*D + p] = row p, column o + a, so
// for a fixed tap phase p the lanes of a wave read CONSECUTIVE columns (conflict-free 8/16-byte ds_reads) instead of addresses D*8 bytes
// apart (which for D = 10 hit the same LDS banks 4 ways -- the round-1 generic kernel ran at 31-37 % of the HBM roofline because of that).
// Each lane produces R consecutive outputs from a sliding register window, so an LDS value is used R times; the taps sit in LDS in the same
// phase-major order, zero padded to a multiple of R (uniform address = broadcast read).  One wave per workgroup and ~21 KiB of LDS at
// D = 10 / 79 taps -> 7 independent waves per CU overlap their staging and their arithmetic.
// Summation order differs from the reference's t = 0..taps-1 (phase-major here); both are float32 sums, parity gate 1e-5 relative RMS.
// Algorithmic traffic: 8 B in + 8/D B out per input sample => HBM bound on paper (3.6 flop/B).
// k_fir_generic: natural-layout fallback for shapes whose polyphase tile does not fit in LDS.
#include "common.hpp"
#include <stdlib.h>
using namespace csdr_amd;

// Buffer loads with the hardware range check (out-of-window lanes read 0).  Declared as the LLVM intrinsics directly: hipcc 7.2's
// __builtin_amdgcn_raw_buffer_load_b64 emits a single-dword load and duplicates it (checked in the ISA), so the builtins are avoided.
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ f32x2_t buf_load_f32x2(i32x4_t rsrc, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");
__device__ float buf_load_f32(i32x4_t rsrc, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");

namespace {

// Generic kernel.  blockDim = 256; each thread handles outputs o = tid, tid+256, ... < tile_outputs.
// LDS holds (tile_outputs-1)*D + taps samples.
template <bool COMPLEX>
__global__ __launch_bounds__(256) void k_fir_generic(const float *__restrict__ in, float *__restrict__ out, int n_out, int tile_outputs,
                                                     size_t in_pitch, size_t out_pitch, int D, const float *__restrict__ taps, int ntaps)
{
    extern __shared__ float4 lds_raw[];
    constexpr int W = COMPLEX ? 2 : 1;                       // floats per sample
    float *win = reinterpret_cast<float *>(lds_raw);
    const int o0 = blockIdx.x * tile_outputs;
    const int outs = min(tile_outputs, n_out - o0);
    if (outs <= 0) return;
    const size_t s = blockIdx.y;
    const float *src = in + (s * in_pitch + (size_t)o0 * D) * W;
    const int win_samples = (outs - 1) * D + ntaps;
    const int win_floats = win_samples * W;
    // coalesced staging; 16-byte path when the window start is 16-byte aligned
    if ((((uintptr_t)src) & 15) == 0) {
        const int nv = win_floats / 4;
        for (int v = threadIdx.x; v < nv; v += blockDim.x) reinterpret_cast<float4 *>(win)[v] = reinterpret_cast<const float4 *>(src)[v];
        for (int k = nv * 4 + threadIdx.x; k < win_floats; k += blockDim.x) win[k] = src[k];
    } else {
        for (int k = threadIdx.x; k < win_floats; k += blockDim.x) win[k] = src[k];
    }
    __syncthreads();
    float *dst = out + (s * out_pitch + (size_t)o0) * W;
    for (int o = threadIdx.x; o < outs; o += blockDim.x) {
        const float *x = win + (size_t)o * D * W;
        if (COMPLEX) {
            float ai = 0.f, aq = 0.f;
            for (int t = 0; t < ntaps; t++) {
                const float h = taps[t];                      // uniform address -> scalar load
                const float2 v = reinterpret_cast<const float2 *>(x)[t];
                ai = fmaf(v.x, h, ai); aq = fmaf(v.y, h, aq);
            }
            reinterpret_cast<float2 *>(dst)[o] = make_float2(ai, aq);
        } else {
            float a = 0.f;
            for (int t = 0; t < ntaps; t++) a = fmaf(x[t], taps[t], a);
            dst[o] = a;
        }
    }
}

// Polyphase-layout kernel, see the file header.  blockDim = 64.  LDS: x[D][Q] of T, then ht[D][NA] floats.
// NA = ceil(ntaps / D) rounded up to a multiple of 2R;  Q = 64*R + NA + R (+2 so that Q = 2 mod 4).
template <typename T> __device__ __forceinline__ T zero_of();
template <> __device__ __forceinline__ float zero_of<float>() { return 0.f; }
template <> __device__ __forceinline__ float2 zero_of<float2>() { return make_float2(0.f, 0.f); }
__device__ __forceinline__ void mac(float &acc, float v, float h) { acc = fmaf(v, h, acc); }
__device__ __forceinline__ void mac(float2 &acc, float2 v, float h) { acc.x = fmaf(v.x, h, acc.x); acc.y = fmaf(v.y, h, acc.y); }

#ifndef FIR_STORE_TILES
#define FIR_STORE_TILES 8
#endif
template <typename T, int R, int U>
__global__ __launch_bounds__(64) void k_fir_poly(const T *__restrict__ in, T *__restrict__ out, int n_out, size_t in_pitch, size_t out_pitch,
                                                 int D, const float *__restrict__ taps, int ntaps, int NA, int Q, int n_tiles, int tiles_per_wg)
{
    extern __shared__ float4 lds_raw[];
    T *x = reinterpret_cast<T *>(lds_raw);
    float *ht = reinterpret_cast<float *>(x + (size_t)D * Q);
    constexpr int TO = 64 * R;
    constexpr bool SWZ = sizeof(T) * R == 32;
    const int lane = threadIdx.x;
    int tile = blockIdx.x * tiles_per_wg;
    const int tile_end = min(n_tiles, tile + tiles_per_wg);
    if (tile >= tile_end) return;
    const T *in_s = in + (size_t)blockIdx.y * in_pitch;
    T *out_s = out + (size_t)blockIdx.y * out_pitch;
    for (int k = lane; k < D * NA; k += 64) {                      // taps, phase major, zero padded: once per workgroup
        const int p = k / NA, a = k - p * NA, t = a * D + p;
        ht[k] = t < ntaps ? taps[t] : 0.f;
    }
    const int total = D * Q;                                       // total <= 64*U (host guarantees)
    // LDS byte address of sample n = lane + 64u of a window: fixed per lane, computed once (row n % D, column n / D, swizzled), two 16-bit
    // values per register.  Samples past the tile (n >= total) go to one spare cell behind the taps.
    unsigned idx2[(U + 1) / 2];
    {
        const int dp = 64 % D, dq = 64 / D;
        const unsigned spare = (unsigned)((size_t)D * Q * sizeof(T) + (size_t)D * NA * 4);
        int p = lane % D, q = lane / D;
#pragma unroll
        for (int u = 0; u < U; u++) {
            unsigned byte = (unsigned)(p * Q + ((SWZ && q < (Q & ~3)) ? (q ^ ((q >> 3) & 2)) : q)) * (unsigned)sizeof(T);
            if (lane + 64 * u >= total) byte = spare;
            if (u & 1) idx2[u / 2] |= byte << 16; else idx2[u / 2] = byte;
            p += dp; q += dq;
            if (p >= D) { p -= D; q++; }
        }
    }
    // The NEXT tile's window is register resident and in flight while the current tile is computed (U x 512 B per wave for complexf).
    // A second window in flight (tiles t+1 and t+2) was tried: 303 VGPRs -> one wave per SIMD -> slower (3.2 vs 4.1 TB/s).
    // Loads are buffer loads whose descriptor covers exactly the samples the tile reads: lanes past the window get 0 from the hardware
    // range check -- no per-load compares, no branches, and every LDS cell of the tile is (re)written each time with data or zero.
    T va[U];
    auto fetch = [&](T (&v)[U], int t) {
        const int outs = min(TO, n_out - t * TO);
        const unsigned win_bytes = (unsigned)((outs - 1) * D + ntaps) * (unsigned)sizeof(T);
        const unsigned long long base = (unsigned long long)(in_s + (size_t)t * TO * D);       // wave uniform
        const i32x4_t rsrc = {(int)(unsigned)base, (int)((base >> 32) & 0xffffu), (int)win_bytes, 0x00020000};
        const int voff = lane * (int)sizeof(T);
#pragma unroll
        for (int u = 0; u < U; u++) {
            if constexpr (sizeof(T) == 8) {
                const f32x2_t r = buf_load_f32x2(rsrc, voff, 64 * u * 8, 0);
                v[u] = make_float2(r.x, r.y);
            } else {
                v[u] = buf_load_f32(rsrc, voff, 64 * u * 4, 0);
            }
        }
    };
    auto stage = [&](T (&v)[U]) {
        char *xb = reinterpret_cast<char *>(x);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const unsigned byte = (u & 1) ? (idx2[u / 2] >> 16) : (idx2[u / 2] & 0xffffu);
            *reinterpret_cast<T *>(xb + byte) = v[u];
        }
    };
    // R consecutive samples m*R .. m*R+R-1 of a row.  For complexf with R = 4 a lane's chunk is 32 bytes, so the two 16-byte halves of
    // consecutive lanes would only ever touch half of the LDS banks per instruction; every other 128-byte group is stored with its
    // halves exchanged (column index bit 1 ^= bit 4), which makes both ds_read_b128 of a chunk conflict free.
    auto chunk = [&](const char *rowb, int m, T *dstw) {
        if constexpr (SWZ) {
            const int lo_off = 32 * m + ((m & 4) << 2);             // bit 4 of the column index 4m -> exchange the 16-byte halves
            const float4 lo = *reinterpret_cast<const float4 *>(rowb + lo_off);
            const float4 hi = *reinterpret_cast<const float4 *>(rowb + (lo_off ^ 16));
            dstw[0] = make_float2(lo.x, lo.y); dstw[1] = make_float2(lo.z, lo.w);
            dstw[2] = make_float2(hi.x, hi.y); dstw[3] = make_float2(hi.z, hi.w);
        } else {
            const T *row = reinterpret_cast<const T *>(rowb);
#pragma unroll
            for (int r = 0; r < R; r++) dstw[r] = row[R * m + r];
        }
    };
    auto block = [&](T (&acc)[R], const T *w0, const T *w1, const float *h) {   // R taps x R outputs on the window (w0, w1)
#pragma unroll
        for (int j = 0; j < R; j++)
#pragma unroll
            for (int r = 0; r < R; r++) mac(acc[r], (r + j < R) ? w0[r + j] : w1[r + j - R], h[j]);
    };
    auto compute = [&](int t, T (&acc)[R]) {
#pragma unroll
        for (int r = 0; r < R; r++) acc[r] = zero_of<T>();
        const char *rowb = reinterpret_cast<const char *>(x);
        const float *hp = ht;
        for (int p = 0; p < D; p++, rowb += (size_t)Q * sizeof(T), hp += NA) {
            T wa[R], wb[R];
            chunk(rowb, lane, wa);
            for (int a = 0, m = lane + 1; a < NA; a += 2 * R, m += 2) {   // NA is a multiple of 2R: two blocks per trip, no window moves
                float h[2 * R];
                chunk(rowb, m, wb);
#pragma unroll
                for (int r = 0; r < 2 * R; r++) h[r] = hp[a + r];
                block(acc, wa, wb, h);
                chunk(rowb, m + 1, wa);
                block(acc, wb, wa, h + R);
            }
        }
    };
#ifndef FIR_DIAG
#define FIR_DIAG 0      // experiments: 1 = every tile's results go to the stream's first tile (stores that never leave L2): what the output stream costs the input stream
#endif
    auto store = [&](int t, const T (&acc)[R]) {
        const int outs = min(TO, n_out - t * TO);
        T *dst = out_s + (size_t)(FIR_DIAG == 1 ? 0 : t) * TO + R * lane;
        if (R * lane + R <= outs && sizeof(T) * R >= 16 && (((uintptr_t)dst) & 15) == 0) {          // whole lane: 16-byte stores
            float4 *d4 = reinterpret_cast<float4 *>(dst);
            const float *af = reinterpret_cast<const float *>(acc);
#pragma unroll
            for (int k = 0; k < (int)(sizeof(T) * R / 16); k++) d4[k] = make_float4(af[4 * k], af[4 * k + 1], af[4 * k + 2], af[4 * k + 3]);
        } else {
#pragma unroll
            for (int r = 0; r < R; r++) if (R * lane + r < outs) dst[r] = acc[r];
        }
    };
    // The results of KD consecutive tiles are stored together (KD x 64 R outputs, contiguous): beside a saturated read stream the memory charges a store event of 2 KiB
    // almost what it charges one of 128 bytes per byte, one of 4 - 8 KiB half of that (tools/microbench/dma_write_mix.hip, profiles/r3_notes.md).
    constexpr int KD = FIR_STORE_TILES;
    __syncthreads();
    fetch(va, tile);
    for (; tile < tile_end; tile += KD) {
        T hold[KD][R];
#pragma unroll
        for (int j = 0; j < KD; j++) {
            if (tile + j < tile_end) {                             // (wave uniform)
                stage(va);
                __syncthreads();
                if (tile + j + 1 < tile_end) fetch(va, tile + j + 1);
                compute(tile + j, hold[j]);
                __syncthreads();                                   // everyone is done reading x[][] before the next tile overwrites it
            }
        }
#pragma unroll
        for (int j = 0; j < KD; j++) if (tile + j < tile_end) store(tile + j, hold[j]);
    }
}

struct PolyCfg { int R, U, NA, Q; size_t lds; };
static bool poly_cfg(int D, int ntaps, size_t elem, PolyCfg &c)
{
    const int rs[2] = {4, 2}, us[2] = {24, 44};
    for (int i = 0; i < 2; i++) {
        const int R = rs[i];
        const int NA = ((ntaps + D - 1) / D + 2 * R - 1) / (2 * R) * (2 * R);
        int Q = (64 * R + NA + R + 1) & ~1;
        if (Q % 4 == 0) Q += 2;                                      // row pitch = 2 (mod 4) samples: the D rows of the staging scatter start in different banks
        const size_t lds = (size_t)D * Q * elem + (size_t)D * NA * 4 + 32;   // + the spare cell
        if (lds > 48u * 1024) continue;
        for (int k = 0; k < 2; k++)
            if ((size_t)D * Q <= 64u * us[k]) { c = {R, us[k], NA, Q, lds}; return true; }
    }
    return false;
}
template <typename T>
static void launch_poly(csdr_amd_ctx *c, const PolyCfg &g, const T *in, T *out, int n_out, int n_streams, size_t in_pitch, size_t out_pitch,
                        int D, const float *taps, int ntaps)
{
    // few, fat workgroups: ~16 resident-wave generations per CU at most; each walks a contiguous range of tiles with the next tile's loads in flight
    const int n_tiles = cdiv(n_out, 64 * g.R);
    const long want = 256L * 7 * 8;
    const int per = (int)(((long)n_tiles * n_streams + want - 1) / want);
    const int tiles_per_wg = per < 1 ? 1 : per;
    dim3 grid(cdiv(n_tiles, tiles_per_wg), (unsigned)n_streams);
#define POLY(RR, UU) hipLaunchKernelGGL((k_fir_poly<T, RR, UU>), grid, dim3(64), g.lds, c->stream, in, out, n_out, in_pitch, out_pitch, D, taps, ntaps, g.NA, g.Q, n_tiles, tiles_per_wg)
    if (g.R == 4) { if (g.U == 24) POLY(4, 24); else POLY(4, 44); }
    else          { if (g.U == 24) POLY(2, 24); else POLY(2, 44); }
#undef POLY
}

// ------------------------------------------------------------------ long filters on complexf: the FIR as a banded product on the fp32 matrix cores
// fir_decimate_cc 50 / 801 taps (the head of the NFM / AM / SSB chains when they start from complexf, README.md:87, 95, 110) reads 16 inputs per
// output tap: k_fir_generic is LDS-read bound at 18 % of the HBM roofline (one 8-byte ds_read per two FMAs).  On v_mfma_f32_16x16x4_f32
// (exact fp32 FMA chains, MI355X_MICROARCH.md) the same window costs two 4-byte ds_reads per 2048 flops:
//   C[i][n] = sum_k A[i][k] B[k][n],   i = output within a group of 16,   n = (group g of NT consecutive groups, re / im),
//   A[i][k] = h[k - D i]  (Toeplitz band of the taps, zero outside [0, L): read from a zero-padded copy in LDS; the band is 52 % dense at D = 50 / L = 801),
//   B[k][n] = x[16 D g + k].part   (the staged input window, natural interleaved order).
// One workgroup = (stream, 16 NT consecutive outputs): its (16 NT - 1) D + L input samples are staged ONCE (8 in + 8 / D out bytes per sample of HBM
// traffic: every input is fetched by exactly one tile plus the L - 1 overlap); the K range (15 D + L) is split over the four waves, the partial tiles are
// reduced through LDS.  Summation order differs from the reference's t = 0 .. L-1: fp32 rounding noise, parity gate 1e-5.
typedef float f32x4_mfma __attribute__((ext_vector_type(4)));
// A workgroup walks `tiles_per_wg` consecutive tiles of its stream: the NEXT tile's window is fetched into registers (NS x 256 float2 per workgroup) while
// the matrix cores work on the current one, so the global-memory latency of a tile is hidden behind the previous tile's arithmetic.
template <int NT, int NS>
__global__ __launch_bounds__(256) void k_fir_mfma(const float2 *__restrict__ in, float2 *__restrict__ out, int n_out, int input_size, size_t in_pitch, size_t out_pitch,
                                                  int D, const float *__restrict__ taps, int L, int tiles_per_wg)
{
    extern __shared__ float4 lds_raw[];
    const int TO = 16 * NT, W = (TO - 1) * D + L, PAD = 15 * D, KT = 15 * D + L, steps = (KT + 3) / 4;
    float *xw = reinterpret_cast<float *>(lds_raw);                 // 2 (W + 8) floats: the window, interleaved, zero tail, XOR-swizzled (below)
    float *hz = xw + ((2 * (W + 8) + 31) & ~31);                      // PAD + 4 steps + 4 floats: the taps with PAD zeros in front
    float *red = hz + PAD + 4 * steps + 4;                            // 4 x 256 partial results
    const size_t s = blockIdx.y;
    const int t = threadIdx.x;
    const int n_tiles = (n_out + TO - 1) / TO, tile0 = blockIdx.x * tiles_per_wg, tile1 = min(tile0 + tiles_per_wg, n_tiles);
    if (tile0 >= n_tiles) return;
    const float2 *base = in + s * in_pitch;
    float2 v[NS];
    auto fetch = [&](int tile) {
        const int first = tile * TO * D;
#pragma unroll
        for (int u = 0; u < NS; u++) { const int k = 256 * u + t; v[u] = (k < W && first + k < input_size) ? base[(size_t)first + k] : make_float2(0.f, 0.f); }
    };
    fetch(tile0);
    for (int k = t; k < PAD + 4 * steps + 4; k += 256) { const int ti = k - PAD; hz[k] = (ti >= 0 && ti < L) ? taps[ti] : 0.f; }
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, i = lane & 15, kk = lane >> 4;      // wave-uniform: the K loop below is a scalar loop
    const int s_lo = wave * steps / 4, s_hi = (wave + 1) * steps / 4;
    const int n = lane & 15, g = n >> 1, part = n & 1;
    const float *ap = hz + PAD + kk - D * i;                          // + 4 step
    // columns of groups g >= NT (NT < 8) are unused: they read group 0's window and are multiplied by zero (no divergence around the MFMAs)
    const float bm = g < NT ? 1.f : 0.f;
    const int b0 = 2 * (16 * D * (g < NT ? g : 0) + kk) + part;       // + 8 step, then the swizzle
    for (int tile = tile0; tile < tile1; tile++) {
        // The window's floats are XOR-swizzled (float address a lives at a ^ ((a >> 5) & 30)): the B operands of the 8 groups sit 16 D samples = a
        // multiple of 32 floats apart, i.e. in ONE bank without it (8-way conflicts on every read); the even mask keeps a (re, im) pair together
#pragma unroll
        for (int u = 0; u < NS; u++) {
            const int k = 256 * u + t;
            if (k < W + 8) { const int a = 2 * k; *reinterpret_cast<float2 *>(xw + (a ^ ((a >> 5) & 30))) = v[u]; }
        }
        __syncthreads();
        if (tile + 1 < tile1) fetch(tile + 1);                        // in flight during the product
        f32x4_mfma acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        int st = s_lo;
        for (; st + 8 <= s_hi; st += 8) {                             // 16 LDS reads in flight, then 8 MFMAs on two alternating accumulators (a dependent
            float av[8], bv[8];                                       // 16x16x4 chain issues every 40 cycles, independent ones every 32)
#pragma unroll
            for (int u = 0; u < 8; u++) { const int a = b0 + 8 * (st + u); av[u] = ap[4 * (st + u)]; bv[u] = xw[a ^ ((a >> 5) & 30)]; }
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], NT == 8 ? bv[u] : bv[u] * bm, acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u + 1], NT == 8 ? bv[u + 1] : bv[u + 1] * bm, acc1, 0, 0, 0);
            }
        }
        for (; st < s_hi; st++) {
            const int a = b0 + 8 * st;
            const float b = xw[a ^ ((a >> 5) & 30)];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * st], NT == 8 ? b : b * bm, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) red[wave * 256 + r * 64 + lane] = acc[r] + acc1[r];
        __syncthreads();
        {   // C layout: column = lane & 15 (n), row = 4 (lane >> 4) + reg (i)
            const int r = t >> 6, ln = t & 63;
            const float sum = red[t] + red[256 + t] + red[512 + t] + red[768 + t];
            const int nn = ln & 15, gg = nn >> 1, pp = nn & 1, ii = 4 * (ln >> 4) + r;
            const int o = tile * TO + 16 * gg + ii;
            if (gg < NT && o < n_out) reinterpret_cast<float *>(out + s * out_pitch + o)[pp] = sum;
        }
        // the next tile overwrites xw / red: everyone has read them (the reads above precede this barrier in program order of every wave)
        __syncthreads();
    }
}

// k_fir_mfma3 (round 5): k_fir_mfma2's product (taps resident in registers, eight waves, hand-scheduled B reads) behind an LDS-DMA staging.  Diagnosis of k_fir_mfma2
// on 64 x 2.4 M samples, 50 / 801 (timing-only builds, profiles/r5_notes.md): products alone 0.231 ms, fetch + staging + reduction alone 0.241 ms, together 0.438 -- the
// phases of ONE resident workgroup (170-200 registers with the next windows in registers) do not overlap at all.  Here the window goes from HBM straight into LDS
// (global_load_lds_dwordx4, 1 KiB per wave instruction, no staging registers: ~110 registers, TWO eight-wave workgroups per CU, one in its products while the other's
// window lands).  The XOR swizzle works on 16-byte granules (mask 28 instead of 30: a DMA piece is linear in LDS, so the permutation is applied to the SOURCE granule of
// each lane, inside its own 128-byte line); lanes past the stream's end re-read its last granule (finite values under zero weights).
#ifndef FIR3_DIAG
#define FIR3_DIAG 0     // timing experiments: 1 = no products, 2 = no window fetch after the first tile
#endif
template <int MAXB, int NW, bool DB>                                 // NW = 8: two workgroups per CU, one window each; NW = 16 (DB): ONE workgroup per CU, four waves per SIMD,
__global__ __launch_bounds__(64 * NW, 4) void k_fir_mfma3(              // two windows -- the next tile lands during this tile's products
const float2 *__restrict__ in, float2 *__restrict__ out, int n_out, int input_size, size_t in_pitch, size_t out_pitch,
                                                      int D, const float *__restrict__ taps, int L, int tiles_per_wg)
{
    extern __shared__ float4 lds_raw[];
    constexpr int NT = 8, NTHR = 64 * NW, TO = 16 * NT, PPW = 64 / NW;      // DMA pieces per wave (of up to 64)
    const int PAD = 15 * D, KT = 15 * D + L, steps = (KT + 3) / 4, nblk = (steps + 3) / 4;
    const int n_pieces = (8 * (16 * D * (NT - 1) + 16 * nblk + 8) + 1023) >> 10;      // 1-KiB DMA pieces of the window (<= 16 nblk + 16 D (NT - 1) + 8 samples)
    float *xw = reinterpret_cast<float *>(lds_raw);
    float *hz = xw + 256 * n_pieces * (DB ? 2 : 1);                   // PAD zeros, the taps, zeros up to 16 nblk + 16 floats
    float *red = hz + PAD + 16 * nblk + 16;                           // NW x 256 partial results
    const size_t s = blockIdx.y;
    const int t = threadIdx.x;
    const int n_tiles = (n_out + TO - 1) / TO, tile0 = blockIdx.x * tiles_per_wg, tile1 = min(tile0 + tiles_per_wg, n_tiles);
    if (tile0 >= n_tiles) return;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, i = lane & 15, kk = lane >> 4;
    const uint8_t *row_base = reinterpret_cast<const uint8_t *>(in + s * in_pitch);
    const uint32_t xw_addr = (uint32_t)(size_t)(__attribute__((address_space(3))) float *)xw;
    // ---- DMA: wave w moves pieces PPW w .. PPW w + PPW - 1; lane l of piece p fills LDS granule Gd = 64 p + l with source granule Gd ^ ((Gd >> 5) & 7)
    uint32_t gsrc[PPW];
#pragma unroll
    for (int r = 0; r < PPW; r++) { const uint32_t gd = 64u * ((uint32_t)PPW * wave + r) + lane; gsrc[r] = gd ^ ((gd >> 5) & 7u); }      // row = gd >> 3; the readers' mask is (row & 28) floats = ((row >> 2) & 7) granules
    auto stage = [&](int tile, uint32_t buf_bytes) {
        const long long first = (long long)tile * TO * D;             // first sample of the window (even: 16-byte granules of the row)
        const long long gmax = ((long long)input_size - first - 2) >> 1;      // last granule that lies inside the stream
        uint32_t vo[PPW];
#pragma unroll
        for (int r = 0; r < PPW; r++) vo[r] = 16u * (uint32_t)min((long long)gsrc[r], gmax > 0 ? gmax : 0LL);
        const uint8_t *sbase = row_base + first * 8;
#pragma unroll
        for (int r = 0; r < PPW; r++) {
            if (PPW * wave + r >= n_pieces) break;                    // (wave uniform)
            const uint32_t la = __builtin_amdgcn_readfirstlane((int)(xw_addr + buf_bytes + 1024u * ((uint32_t)PPW * wave + r)));
            uint32_t keep;
            // (no `nt`: the window's first L - 1 samples were the previous tile's last ones, fetched by this very workgroup a few microseconds ago -- they should come from L2)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(vo[r]), "s"(sbase), "s"(la) : "memory");
        }
    };
    stage(tile0, 0u);
    for (int k = t; k < PAD + 16 * nblk + 16; k += NTHR) { const int ti = k - PAD; hz[k] = (ti >= 0 && ti < L) ? taps[ti] : 0.f; }
    const int b_lo = wave * nblk / NW, b_hi = (wave + 1) * nblk / NW;   // this wave's blocks of four K-steps (wave uniform)
    const int n = lane & 15, g = n >> 1, part = n & 1;
    __syncthreads();                                                  // hz is complete
    float A[4 * MAXB];
    {
        const float *ap = hz + PAD + kk - D * i + 16 * b_lo;
#pragma unroll
        for (int j = 0; j < MAXB; j++)
#pragma unroll
            for (int u = 0; u < 4; u++) A[4 * j + u] = (b_lo + j < b_hi) ? ap[16 * j + 4 * u] : 0.f;
    }
    const int c = 2 * kk + part;
    const int h_last = D * g + b_hi - 1 + (b_hi == b_lo);              // (blocks beyond the wave's range re-read its last one: multiplied by zero)
    for (int tile = tile0; tile < tile1; tile++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's pieces of the window
        __syncthreads();                                              // everybody's; and the previous tile's partial sums have been read
        const uint32_t cur_buf = DB ? (uint32_t)((tile - tile0) & 1) * 1024u * (uint32_t)n_pieces : 0u;
        if (DB && tile + 1 < tile1 && FIR3_DIAG != 2) stage(tile + 1, (1024u * (uint32_t)n_pieces) - cur_buf);      // the other window: every wave has left the tile before last
        f32x4_mfma acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        int hi = D * g + b_lo;
        asm volatile("" : "+v"(hi));                                  // (per tile: otherwise every B address is hoisted out of the tile loop)
        // batches of two blocks (eight B reads in flight, then eight products), the next batch's reads issued before this batch's products; an odd last block on its own
        float bA[8], bB[8];
        auto issue = [&](float (&bv)[8], int blk0, const int nb) {
#pragma unroll
            for (int jj = 0; jj < 2; jj++) {
                if (jj >= nb) break;
                const int h2 = min(blk0 + jj, h_last);
                const int m = h2 & 28;
                const uint32_t a0 = xw_addr + cur_buf + ((uint32_t)h2 << 7) + ((uint32_t)(c ^ m) << 2);
#pragma unroll
                for (int u = 0; u < 4; u++) asm volatile("ds_read_b32 %0, %1" : "=v"(bv[4 * jj + u]) : "v"(a0 ^ (uint32_t)(u << 5)) : "memory");
            }
        };
        constexpr int NBATCH = (MAXB + 1) / 2;                       // batch q covers blocks 2 q, 2 q + 1 (the last one only 2 q when MAXB is odd)
        auto blocks_of = [](int q) { return (2 * q + 1 < MAXB) ? 2 : 1; };
        if (FIR3_DIAG != 1) issue(bA, hi, blocks_of(0));
#pragma unroll
        for (int q = 0; q < (FIR3_DIAG == 1 ? 0 : NBATCH); q++) {
            float (&cur)[8] = (q & 1) ? bB : bA;
            float (&nxt)[8] = (q & 1) ? bA : bB;
            const int nbc = blocks_of(q);
            if (q + 1 < NBATCH) {
                const int nbn = blocks_of(q + 1);
                issue(nxt, hi + 2 * (q + 1), nbn);
                if (nbn == 2) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]), "+v"(cur[6]), "+v"(cur[7]));
                else asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]), "+v"(cur[6]), "+v"(cur[7]));
            } else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]), "+v"(cur[6]), "+v"(cur[7]));
#pragma unroll
            for (int u = 0; u < 4 * nbc; u += 2) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[8 * q + u], cur[u], acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[8 * q + u + 1], cur[u + 1], acc1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) red[wave * 256 + r * 64 + lane] = acc[r] + acc1[r];
        __syncthreads();                                              // every wave has left the window: the next one may land
        if (!DB && tile + 1 < tile1 && FIR3_DIAG != 2) stage(tile + 1, 0u);
        if (t < 256) {   // C layout: column = lane & 15 (n), row = 4 (lane >> 4) + reg (i)
            const int r = t >> 6, ln = t & 63;
            float sum = red[t];
#pragma unroll
            for (int wv = 1; wv < NW; wv++) sum += red[256 * wv + t];
            const int nn = ln & 15, gg = nn >> 1, pp = nn & 1, ii = 4 * (ln >> 4) + r;
            const int o = tile * TO + 16 * gg + ii;
            if (o < n_out) reinterpret_cast<float *>(out + s * out_pitch + o)[pp] = sum;
        }
    }
}

} // namespace

static int pick_tile(int D, int ntaps, int floats_per_sample, int n_out)
{
    // largest tile (<= 1024 outputs) whose window fits in 64 KiB of LDS, at least 1
    int to = 1024;
    while (to > 1 && ((size_t)(to - 1) * D + ntaps) * floats_per_sample * 4 > 64 * 1024) to /= 2;
    if (to > n_out) to = n_out;
    return to < 1 ? 1 : to;
}

static thread_local const char *g_fir_last_kernel = "";

extern "C" {

/* which kernel the calling thread's last csdr_amd_fir_decimate_cc launched (bench_fir.py's roofline.kernel; "" before the first call) */
const char *csdr_amd_fir_last_kernel(void) { return g_fir_last_kernel; }

int csdr_amd_fir_decimate_cc(csdr_amd_ctx *c, const csdr_complexf *in, csdr_complexf *out, int n_streams, int input_size,
                             size_t in_pitch, size_t out_pitch, int decimation, const float *taps, int taps_length)
{
    if (decimation <= 0 || taps_length <= 0) return fail_msg(-3, "fir_decimate_cc: bad decimation/taps_length");
    if (input_size < taps_length || n_streams <= 0) return 0;
    const int n_out = (input_size - taps_length) / decimation + 1;     // libcsdr.c:536-538 loop bound
    PolyCfg g;
    // (the FIR entry points are plain functions: their A/B switches are read once per process)
    static const bool force_generic = getenv("CSDR_AMD_FIR_GENERIC") != nullptr;      // (A/B and the generic kernel's own parity test)
    // (Short filters -- config 1's 10 / 79 -- stay on k_fir_poly: the matrix-core construction of the long-filter kernel was ported to that shape in round 5 (k_fir_mfma5) and ran
    //  exactly as fast: experiments/fir_mfma5.hip.)
    if (!force_generic && poly_cfg(decimation, taps_length, 8, g)) {
        launch_poly<float2>(c, g, (const float2 *)in, (float2 *)out, n_out, n_streams, in_pitch, out_pitch, decimation, taps, taps_length);
        CSDR_LAUNCH_CHECK();
        g_fir_last_kernel = "k_fir_poly";
        return n_out;
    }
    {   // long filters: banded product on the fp32 matrix cores (NT groups of 16 outputs per workgroup, window <= ~64 KiB)
        for (int nt = 8; nt >= 1; nt >>= 1) {
            const int W = (16 * nt - 1) * decimation + taps_length, steps = (15 * decimation + taps_length + 3) / 4;
            const size_t lds = sizeof(float) * (2 * (size_t)(W + 8) + 32 + 15 * (size_t)decimation + 4 * (size_t)steps + 4 + 1024);
            if (lds > 76 * 1024) continue;                           // two workgroups per CU
            const int n_tiles = cdiv(n_out, 16 * nt), ns = cdiv(W + 8, 256);
            if (ns > 32) continue;                                   // register staging: at most 32 float2 per thread
            int tpw = (int)(((long)n_tiles * n_streams + 4095) / 4096); if (tpw < 1) tpw = 1; if (tpw > 16) tpw = 16;      // ~4096 workgroups, each a run of consecutive tiles
            const dim3 grid(cdiv(n_tiles, tpw), (unsigned)n_streams);
#define FIR_MFMA(NTV, NSV) do { if (lds > 64 * 1024) { const int arc = lds_attr_once((const void *)k_fir_mfma<NTV, NSV>, lds); if (arc) return arc; }                 \
            hipLaunchKernelGGL((k_fir_mfma<NTV, NSV>), grid, dim3(256), lds, c->stream, (const float2 *)in, (float2 *)out, n_out, input_size, in_pitch, out_pitch,   \
                               decimation, taps, taps_length, tpw); } while (0)
#define FIR_MFMA_NS(NTV) do { if (ns <= 8) FIR_MFMA(NTV, 8); else if (ns <= 16) FIR_MFMA(NTV, 16); else FIR_MFMA(NTV, 32); } while (0)
            const int nblk = (steps + 3) / 4;
            // k_fir_mfma3: the window by LDS-DMA.  Needs: 8 outputs groups, <= 14 blocks per wave, the window + slack inside 64 KiB, 16-byte aligned stream rows
            {
                const int per_wave8 = (nblk + 7) / 8 + 1;      // (<= 14 blocks per wave)
                const long win_samples = 16L * decimation * 7 + 16L * nblk + 8;
                const size_t lds3 = (((size_t)8 * win_samples + 1023) & ~(size_t)1023) + sizeof(float) * (15 * (size_t)decimation + 16 * (size_t)nblk + 16 + 2048);
                if (nt == 8 && per_wave8 <= 15 && win_samples <= 8192 && (in_pitch % 2) == 0 && (input_size % 2) == 0 && (((uintptr_t)in) & 15) == 0 && lds3 <= 80 * 1024) {
                    // two eight-wave workgroups per CU, one window each.  (Round 5 also measured ONE sixteen-wave workgroup with two windows -- k_fir_mfma3<.., 16, true> --:
                    // slower, profiles/r5_notes.md; it is no longer instantiated.)
#define FIR_MFMA3(MB, NWV, DBV, LDSV) do { const int arc = lds_attr_once((const void *)k_fir_mfma3<MB, NWV, DBV>, LDSV); if (arc) return arc;                                                      \
                    hipLaunchKernelGGL((k_fir_mfma3<MB, NWV, DBV>), grid, dim3(64 * NWV), LDSV, c->stream, (const float2 *)in, (float2 *)out, n_out, input_size, in_pitch, out_pitch,        \
                                       decimation, taps, taps_length, tpw); } while (0)
                    const int need = (nblk + 7) / 8;                 // the largest wave share: wave w takes blocks [w nblk / 8, (w + 1) nblk / 8)
                    if (need <= 8) FIR_MFMA3(8, 8, false, lds3); else if (need <= 10) FIR_MFMA3(10, 8, false, lds3); else if (need <= 12) FIR_MFMA3(12, 8, false, lds3); else if (need <= 13) FIR_MFMA3(13, 8, false, lds3); else FIR_MFMA3(14, 8, false, lds3);
#undef FIR_MFMA3
                    CSDR_LAUNCH_CHECK();
                    g_fir_last_kernel = "k_fir_mfma3";
                    return n_out;
                }
            }
            if (nt == 8) FIR_MFMA_NS(8); else if (nt == 4) FIR_MFMA_NS(4); else if (nt == 2) FIR_MFMA_NS(2); else FIR_MFMA_NS(1);
#undef FIR_MFMA_NS
#undef FIR_MFMA
            CSDR_LAUNCH_CHECK();
            g_fir_last_kernel = "k_fir_mfma";
            return n_out;
        }
    }
    const int to = pick_tile(decimation, taps_length, 2, n_out);
    const size_t win_bytes = ((size_t)(to - 1) * decimation + taps_length) * 8;
    if (win_bytes > 160 * 1024 - 256) return fail_msg(-3, "fir_decimate_cc: %d taps exceed the LDS window (use the FFT path)", taps_length);
    if (win_bytes > 64 * 1024) { const int arc = lds_attr_once((const void *)k_fir_generic<true>, win_bytes + 16); if (arc) return arc; }
    dim3 grid(cdiv(n_out, to), (unsigned)n_streams);
    hipLaunchKernelGGL((k_fir_generic<true>), grid, dim3(256), win_bytes + 16, c->stream, (const float *)in, (float *)out, n_out, to,
                       in_pitch, out_pitch, decimation, taps, taps_length);
    CSDR_LAUNCH_CHECK();
    g_fir_last_kernel = "k_fir_generic";
    return n_out;
}

int csdr_amd_fir_ff(csdr_amd_ctx *c, const float *in, float *out, int n_streams, int input_size, size_t in_pitch, size_t out_pitch,
                    const float *taps, int taps_length)
{
    if (taps_length <= 0) return 0;
    const int n_out = input_size - taps_length;                         // libcsdr.c:1121: i < input_size - taps_length
    if (n_out <= 0 || n_streams <= 0) return 0;
    PolyCfg g;
    static const bool force_generic = getenv("CSDR_AMD_FIR_GENERIC") != nullptr;
    if (!force_generic && poly_cfg(1, taps_length, 4, g)) {
        launch_poly<float>(c, g, in, out, n_out, n_streams, in_pitch, out_pitch, 1, taps, taps_length);
        CSDR_LAUNCH_CHECK();
        return n_out;
    }
    // window for n outputs of the generic kernel is (outs-1)*1 + ntaps, exactly what those outputs read
    const int to = pick_tile(1, taps_length, 1, n_out);
    const size_t win_bytes = ((size_t)(to - 1) + taps_length) * 4;
    dim3 grid(cdiv(n_out, to), (unsigned)n_streams);
    hipLaunchKernelGGL((k_fir_generic<false>), grid, dim3(256), win_bytes + 16, c->stream, in, out, n_out, to, in_pitch, out_pitch, 1, taps, taps_length);
    CSDR_LAUNCH_CHECK();
    return n_out;
}

} // extern "C"
