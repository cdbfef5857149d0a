// experiments/fftfilt_team32.hpp -- NOT BUILT.  The 4096-point window of the one-pass FFT filter as a team of TWO waves with 32 points per lane (round 6): correct (2.2e-7
// from the oracle on the bench's verify rows) and SLOWER than k_fftfilt_wave -- 0.330 against 0.288 ms per 64 x 16 blocks at 1023 taps, 0.285 against 0.256 at 63 --
// although it keeps four waves per SIMD: a third more vector work per window (the radix-4 step across lanes), eight team barriers, and the same ~48 % of the vector ALU's
// time in use (profiles/r6_notes.md).  Kept for its parts: fw_pk_dft32 (below, as it was in fftfilt_wave.hpp), the (jp, jp + 16) row pairing of the 16-byte loads.
// To try it again: include it from fftfilt_lds.hip behind fftfilt_team.hpp, make fw_pk_dft16 / fw_pk_twiddle templates on the array length, add tables + launch (git
// history of this file's commit shows the wiring).
//
// k_fftfilt_wave (one wave, 64 points per lane) holds a window in 128 of a wave's 256 registers: two waves per SIMD, and what a wave waits for -- its samples behind its own
// stores, the spectrum, its transposes -- only ONE other wave can fill (vector ALU 45 % busy).  Here: N = 4096 = 32 x 128, thread t of 128 holds x[t + 128 j], j = 0..31
// (64 registers of <= 128: FOUR waves per SIMD, eight windows per CU in flight), and the steps of fftfilt_team.hpp with radix 32 in registers:
//   pass 0   radix 32 over j -> k_a                  x W_4096^(t k_a)
//   E1       exchange through LDS (17 KiB per window, halves)          thread (k_a, m) gets t = 4 i + m, i = 0..31
//   pass 1   radix 32 over i -> k_c                  x W_128^(m k_c)
//   R        radix 4 over m across the four neighbouring lanes (DPP)   -> X[k_a + 32 (k_c + 32 k_d)],  x taps spectrum,  R inverse,  x conj W_128^(m k_c)
//   pass 2   inverse radix 32 over k_c -> i;  E2 back;  x conj W_4096^(t k_a);  pass 3: inverse radix 32 over k_a -> y[t + 128 j]
// 16-byte loads pair the rows (jp, jp + 16) -- what the radix-32 pass's first step (radix 2 over n, n + 16) and its two radix-16 halves (even / odd outputs) work on.
#pragma once

namespace {

struct Fq {                                                              // geometry: 128 threads, 32 rows
    static constexpr int T = 128, R = 32, N = 4096, M = 4, P = T + M, TWE = 10, ROWB = 16 * T;
    static constexpr size_t LDS_BYTES = (size_t)R * P * sizeof(float);
};
FFL_HD constexpr size_t fq_h_index(int kc, int p) { return ((size_t)(kc >> 1) * Fq::T + p) * 2 + (kc & 1); }

#ifdef __HIPCC__
template <bool IM, bool BACK> __device__ __forceinline__ void fq_exchange_half(fw_pk2 (&v)[32], float *a, float *b)
{
    ffl_barrier();                                                      // the previous half's / exchange's reads are done
#pragma unroll
    for (int r = 0; r < 32; r++) { const float x = IM ? v[r].y : v[r].x; if (BACK) b[Fq::M * r] = x; else a[r * Fq::P] = x; }
    ffl_barrier();
#pragma unroll
    for (int r = 0; r < 32; r++) { const float x = BACK ? a[r * Fq::P] : b[Fq::M * r]; if (IM) v[r].y = x; else v[r].x = x; }
}
template <bool BACK> __device__ __forceinline__ void fq_exchange(fw_pk2 (&v)[32], float *a, float *b) { fq_exchange_half<false, BACK>(v, a, b); fq_exchange_half<true, BACK>(v, a, b); }

__device__ __forceinline__ void fq_tw_issue(fw_pk2 (&tw)[Fq::TWE], ffl_i32x4 rt, int voff, int stride)
{
#pragma unroll
    for (int e = 0; e < Fq::TWE; e++) asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(tw[e]) : "v"(voff), "s"(rt), "s"(e * stride) : "memory");
}
__device__ __forceinline__ void fq_tw_ready(fw_pk2 (&tw)[Fq::TWE])
{
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(tw[0]), "+v"(tw[1]), "+v"(tw[2]), "+v"(tw[3]), "+v"(tw[4]), "+v"(tw[5]), "+v"(tw[6]), "+v"(tw[7]), "+v"(tw[8]), "+v"(tw[9]));
}
// registers 8 c .. 8 c + 7: radix 4 across the lanes, x spectrum, inverse radix 4
__device__ __forceinline__ void fq_centre_chunk(fw_pk2 (&v)[32], const ffl_f32x4 (&h)[4], int c, float sig2, float sig1, bool lane3)
{
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int r = 8 * c + 2 * q + b;
            fw_pk2 x = ft_radix_lanes<4, false>(v[r], sig2, sig1, lane3);
            x = fw_pk_cmul<false>(x, b ? fw_pk2{h[q].z, h[q].w} : fw_pk2{h[q].x, h[q].y});          // libcsdr.c:826-830 (and 836-839: the 1/N is in the table)
            v[r] = ft_radix_lanes<4, true>(x, sig2, sig1, lane3);
        }
    }
}
// rows (jp, jp + 16) of a window, jp = h, h + 2, ...: samples 2 (lane & 31), 2 (lane & 31) + 1 of the wave's 64 columns, row jp in lanes < 32 and row jp + 16 in lanes >= 32
__device__ __forceinline__ void fq_load_half(ffl_f32x4 (&nx)[16], const FwRows &x, int h)
{
#pragma unroll
    for (int m = 0; m < 8; m++) nx[2 * m + h] = ffl_buf_load4(x.r, x.v0 + (Fq::ROWB / 2) * (2 * m + h), 0, 0);
}
__device__ __forceinline__ void fq_rows_to_regs(fw_pk2 (&v)[32], const ffl_f32x4 (&nx)[16])
{
#pragma unroll
    for (int jp = 0; jp < 16; jp++) {
        fw_pk2 P = {nx[jp].x, nx[jp].y}, Q = {nx[jp].z, nx[jp].w};
        fw_swap_halves(P, Q);
        v[jp] = P; v[jp + 16] = Q;
    }
}
// the radix-16 half H of pass 3 (outputs 2 k' + H), its stores -- rows (r, r + 16), r = 2 a + H -- and the same rows of the next window
template <int H> __device__ __forceinline__ void fq_pass3_half(fw_pk2 (&v)[32], const FwRows &y, ffl_f32x4 (&nx)[16], const FwRows &xn)
{
    fw_pk_dft16<16 * H, true>(v);                                       // v[16 H + k'] = y row 2 k' + H
#pragma unroll
    for (int a = 0; a < 4; a++) {                                       // rows r = 2 a + H (k' = a) and r + 16 (k' = a + 8)
        fw_pk2 P = v[16 * H + a], Q = v[16 * H + a + 8];
        fw_swap_halves(P, Q);
        const ffl_f32x4 r = {P.x, P.y, Q.x, Q.y};
        ffl_buf_store4(r, y.r, y.v0 + (Fq::ROWB / 2) * (2 * a + H), 0, 0);
    }
#pragma unroll
    for (int a = 4; a < 8; a++) {
        fw_pk2 P = v[16 * H + a], Q = v[16 * H + a + 8];
        fw_swap_halves(P, Q);
        const ffl_f32x4 r = {P.x, P.y, Q.x, Q.y};
        ffl_buf_store4(r, y.r, y.v0 + (Fq::ROWB / 2) * (2 * a + H), 0, 0);
    }
    fq_load_half(nx, xn, H);
}

// tw1[e T + p]: the bases of W_4096^(t k_a) for thread p (t = ft_logical(p)); tw2[e 4 + m]: those of W_128^(m k_c); hw: the spectrum in fq_h_index order
__global__ __launch_bounds__(128, 4) void k_fftfilt_team32(const float2 *__restrict__ in, size_t in_pitch, const float2 *__restrict__ hist, int k1p, int m_new,
                                                           int n_chunks, int n_windows, float2 *__restrict__ out, size_t out_pitch, const float2 *__restrict__ hw,
                                                           const float2 *__restrict__ g_tw1, const float2 *__restrict__ g_tw2)
{
    constexpr int T = Fq::T, N = Fq::N;
    extern __shared__ float4 ffl_raw[];
    float *L = reinterpret_cast<float *>(ffl_raw);
    const int p = threadIdx.x, lane = p & 63, t = ft_logical(p), m = p & 3;
    float *xa = L + t, *xb = L + (p >> 2) * Fq::P + m;
    const float sig2 = (m & 2) ? -1.f : 1.f, sig1 = (m & 1) ? -1.f : 1.f;
    const bool lane3 = m == 3;
    const unsigned long long b1 = (unsigned long long)g_tw1, b2 = (unsigned long long)g_tw2, bh = (unsigned long long)hw;
    const ffl_i32x4 rt1 = {(int)(unsigned)b1, (int)((b1 >> 32) & 0xffffu), Fq::TWE * T * 8, 0x00020000}, rt2 = {(int)(unsigned)b2, (int)((b2 >> 32) & 0xffffu), Fq::TWE * 4 * 8, 0x00020000};
    const ffl_i32x4 rh = {(int)(unsigned)bh, (int)((bh >> 32) & 0xffffu), N * 8, 0x00020000};
    const int V = N - k1p;
    const int per_xcd = (n_windows + 7) >> 3, xcd = blockIdx.x & 7, stride = gridDim.x >> 3;      // gridDim.x is a multiple of 8
    const int w_end = min(n_windows, (xcd + 1) * per_xcd);
    int w = xcd * per_xcd + (blockIdx.x >> 3);
    if (w >= w_end) return;
    // the lane's 16 bytes of a row pair: samples 2 (lane & 31), + 1 of the wave's 64 columns, in row jp (lanes < 32) / row jp + 16 (lanes >= 32)
    const int n_lane = (p & ~63) + 2 * (lane & 31) + 16 * T * (lane >> 5);
    auto rows_in = [&](int win) {
        const int s = win / n_chunks, c = win - s * n_chunks;
        const unsigned long long bx = (unsigned long long)(in + (size_t)s * in_pitch);
        return FwRows{ffl_i32x4{(int)(unsigned)bx, (int)((bx >> 32) & 0xffffu), m_new * 8, 0x00020000}, (c * V - k1p + n_lane) * 8};
    };
    ffl_f32x4 nx[16];
    {
        const FwRows x0 = rows_in(w);
        fq_load_half(nx, x0, 0); fq_load_half(nx, x0, 1);
    }
    for (;;) {
        const int s = w / n_chunks, c = w - s * n_chunks, w0 = c * V - k1p;
        fw_pk2 v[32];
        fq_rows_to_regs(v, nx);
        if (w0 < 0) {                                                   // uniform: the stream's first window
            const unsigned long long bhs = (unsigned long long)(hist + (size_t)s * k1p);
            const ffl_i32x4 rhs = {(int)(unsigned)bhs, (int)((bhs >> 32) & 0xffffu), k1p * 8, 0x00020000};
            const int vh = (k1p + w0 + t) * 8;
#pragma unroll
            for (int j = 0; j < 32; j++) { const ffl_f32x2 r = ffl_buf_load(rhs, vh + T * 8 * j, 0, 0); v[j].x += r.x; v[j].y += r.y; }
        }
        {
            fw_pk2 tw[Fq::TWE];
            fq_tw_issue(tw, rt1, p * 8, T * 8);
            fw_pk_dft32<false>(v);
            fq_tw_ready(tw);
            fw_pk_twiddle<false>(v, tw);
        }
        fq_exchange<false>(v, xa, xb);
        {
            ffl_f32x4 ha[4], hb[4];
            fw_pk2 tw[Fq::TWE];
            fq_tw_issue(tw, rt2, m * 8, 4 * 8);
            ft_h_issue(ha, rh, p * 16, 0, Fq::ROWB);
            fw_pk_dft32<false>(v);
            fq_tw_ready(tw);
            fw_pk_twiddle<false>(v, tw);
            ft_h_issue(hb, rh, p * 16, 1, Fq::ROWB);
            ft_h_ready<4>(ha); fq_centre_chunk(v, ha, 0, sig2, sig1, lane3); ft_h_issue(ha, rh, p * 16, 2, Fq::ROWB);
            ft_h_ready<4>(hb); fq_centre_chunk(v, hb, 1, sig2, sig1, lane3); ft_h_issue(hb, rh, p * 16, 3, Fq::ROWB);
            ft_h_ready<4>(ha); fq_centre_chunk(v, ha, 2, sig2, sig1, lane3);
            ft_h_ready<0>(hb); fq_centre_chunk(v, hb, 3, sig2, sig1, lane3);
            fw_pk_twiddle<true>(v, tw);
        }
        {
            fw_pk2 tw[Fq::TWE];
            fq_tw_issue(tw, rt1, p * 8, T * 8);
            fw_pk_dft32<true>(v);
            fq_exchange<true>(v, xa, xb);
            fq_tw_ready(tw);
            fw_pk_twiddle<true>(v, tw);
        }
        // results n = k1p .. N-1 of the window are outputs c V + (n - k1p): descriptor based at output c V, range = what is left of the call
        const unsigned long long by = (unsigned long long)(out + (size_t)s * out_pitch + (size_t)c * V);
        const FwRows y = {ffl_i32x4{(int)(unsigned)by, (int)((by >> 32) & 0xffffu), (m_new - c * V) * 8, 0x00020000}, (n_lane - k1p) * 8};
        const int wn = w + stride;
        const bool more = wn < w_end;
        const FwRows xn = rows_in(more ? wn : w);                       // (the last window: this one again, ignored -- no branch around values in flight)
        fw_pk_dft32_head<true>(v);
        fq_pass3_half<0>(v, y, nx, xn); fq_pass3_half<1>(v, y, nx, xn);
        if (!more) break;
        w = wn;
    }
}
#endif

} // namespace
