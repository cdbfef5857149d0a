// fftfilt_team.hpp -- the 8192- and 16384-point windows of the one-pass FFT filter (fftfilt_lds.hip) as a TEAM of M = 2 / 4 waves per window (round 6), built from the
// wave-per-window kernel's parts (fftfilt_wave.hpp: 64 points per lane, radix 64 in registers on packed f32).  Included by fftfilt_lds.hip only.
//
// N = 64 T points, T = 64 M threads.  Thread t holds x[t + T j], j = 0..63:
//   pass 0   radix 64 over j in registers -> k_a            x W_N^(t k_a)
//   E1       exchange through LDS (the whole team)          thread (k_a, m) gets t = M i + m, i = 0..63
//   pass 1   radix 64 over i in registers -> k_c            x W_T^(m k_c)
//   R        radix M over m ACROSS the M neighbouring lanes (DPP)   -> X[k_a + 64 (k_c + 64 k_d)] in register k_c of thread (k_a, slot of k_d)
//            x taps spectrum; R inverse; x conj W_T^(m k_c)
//   pass 2   inverse radix 64 over k_c -> i
//   E2       exchange back                                  thread t gets k_a = 0..63
//   pass 3   x conj W_N^(t k_a), inverse radix 64 over k_a -> y[t + T j]
// The 1024-/512-thread kernels these replace do five or six workgroup-wide exchanges per window with ONE workgroup on a CU: load, transform and store follow each other
// (0.49 ms per 64 x 16 blocks at 4095 taps, 0.26 of the roofline).  Here two exchanges (real and imaginary halves one after the other: 65 KiB of LDS per 16384-point
// window), and two independent teams per CU whose memory phases and butterflies overlap.
#pragma once

namespace {

template <int M> struct FtGeom {
    static constexpr int T = 64 * M, N = 64 * T, LOGM = M == 4 ? 2 : 1;
    static constexpr int P = T + M;                                     // LDS row pitch in floats: the readers' 32 lanes (32 / M rows x M columns) hit 32 banks
    static constexpr size_t LDS_BYTES = (size_t)64 * P * sizeof(float);
    static constexpr int ROWB = 16 * T;                                  // bytes between two row PAIRS of a window (2 T samples)
};
FFL_HD constexpr int ft_slot_kd(int m, int M) { return M == 4 ? ((m & 1) << 1) | (m >> 1) : m; }      // the k_d lane m of a quad ends up with (radix 4: bit reversed)
// logical thread of physical thread p (the rows 16-byte loads + fw_swap_halves leave in lane l of wave a): t = 64 a + fw_pi(l)
FFL_HD constexpr int ft_logical(int p) { return (p & ~63) + fw_pi(p & 63); }
// taps spectrum where thread p finds the pair k_c = 2 q, 2 q + 1: 16 bytes at ((q T + p) * 2)
template <int M> FFL_HD constexpr size_t ft_h_index(int kc, int p) { return ((size_t)(kc >> 1) * FtGeom<M>::T + p) * 2 + (kc & 1); }

#ifdef __HIPCC__
template <int CTRL> __device__ __forceinline__ float ft_dpp(float x)
{
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), CTRL, 0xf, 0xf, true));
}
// butterfly with the lane DIST away inside a quad: x <- partner + sigma x (sigma = +1 in the lower lane of the pair, -1 in the upper one)
template <int DIST> __device__ __forceinline__ fw_pk2 ft_bfly(fw_pk2 x, float sigma)
{
    constexpr int ctrl = DIST == 2 ? 0x4E : 0xB1;                      // quad_perm [2,3,0,1] / [1,0,3,2]
    const fw_pk2 xs = x * sigma;
    return fw_pk2{ft_dpp<ctrl>(x.x) + xs.x, ft_dpp<ctrl>(x.y) + xs.y};
}
// the radix-M step over the M lanes of a group (forward: lane 3's odd difference x -i between the two stages; inverse: the stages in reverse order, x +i)
template <int M, bool INV> __device__ __forceinline__ fw_pk2 ft_radix_lanes(fw_pk2 x, float sig2, float sig1, bool lane3)
{
    if constexpr (M == 2) return ft_bfly<1>(x, sig1);
    else if constexpr (!INV) {
        fw_pk2 y = ft_bfly<2>(x, sig2);
        y = lane3 ? fw_pk2{y.y, -y.x} : y;
        return ft_bfly<1>(y, sig1);
    } else {
        fw_pk2 y = ft_bfly<1>(x, sig1);
        y = lane3 ? fw_pk2{-y.y, y.x} : y;
        return ft_bfly<2>(y, sig2);
    }
}

// one half (real / imaginary parts) of an exchange: every thread writes its 64 values at wr[k P], the team meets, every thread reads rd[M i]
template <int M, bool IM, bool BACK> __device__ __forceinline__ void ft_exchange_half(fw_pk2 (&v)[64], float *a, float *b)
{
    constexpr int P = FtGeom<M>::P;
    // forward (E1): write a[k_a P] (a = L + t), read b[M i] (b = L + k_a P + m); back (E2): write b[M i], read a[k_a P]
    ffl_barrier();                                                      // the previous half's / exchange's reads are done
#pragma unroll
    for (int r = 0; r < 64; r++) { const float x = IM ? v[r].y : v[r].x; if (BACK) b[M * r] = x; else a[r * P] = x; }
    ffl_barrier();
#pragma unroll
    for (int r = 0; r < 64; r++) { const float x = BACK ? a[r * P] : b[M * r]; if (IM) v[r].y = x; else v[r].x = x; }
}
template <int M, bool BACK> __device__ __forceinline__ void ft_exchange(fw_pk2 (&v)[64], float *a, float *b)
{
    ft_exchange_half<M, false, BACK>(v, a, b); ft_exchange_half<M, true, BACK>(v, a, b);
}

__device__ __forceinline__ void ft_h_issue(ffl_f32x4 (&h)[4], ffl_i32x4 rh, int voff, int chunk, int rowb)
{
#pragma unroll
    for (int q = 0; q < 4; q++) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(h[q]) : "v"(voff), "s"(rh), "s"((chunk * 4 + q) * rowb) : "memory");
}
template <int INFLIGHT> __device__ __forceinline__ void ft_h_ready(ffl_f32x4 (&h)[4])
{
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]) : "n"(INFLIGHT));
}
// registers 8 c .. 8 c + 7: radix M across the lanes, x spectrum, inverse radix M
template <int M> __device__ __forceinline__ void ft_centre_chunk(fw_pk2 (&v)[64], const ffl_f32x4 (&h)[4], int c, float sig2, float sig1, bool lane3)
{
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int r = 8 * c + 2 * q + b;
            fw_pk2 x = ft_radix_lanes<M, false>(v[r], sig2, sig1, lane3);
            x = fw_pk_cmul<false>(x, b ? fw_pk2{h[q].z, h[q].w} : fw_pk2{h[q].x, h[q].y});          // libcsdr.c:826-830 (and 836-839: the 1/N is in the table)
            v[r] = ft_radix_lanes<M, true>(x, sig2, sig1, lane3);
        }
    }
}

// the 18 twiddle bases of a thread, requested by hand in one batch (as plain loads the compiler -- no register to spare -- fetches them one by one: load, wait, multiply,
// 18 round trips to the L2 in front of each of the three twiddle steps of a window, half of the first version's time) and waited for behind the radix-64 pass in between
template <int STRIDE> __device__ __forceinline__ void ft_tw_issue(fw_pk2 (&tw)[FW_TWE], ffl_i32x4 rt, int voff)
{
#pragma unroll
    for (int e = 0; e < FW_TWE; e++) asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(tw[e]) : "v"(voff), "s"(rt), "s"(e * STRIDE) : "memory");
}
__device__ __forceinline__ void ft_tw_ready(fw_pk2 (&tw)[FW_TWE])
{
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(tw[0]), "+v"(tw[1]), "+v"(tw[2]), "+v"(tw[3]), "+v"(tw[4]), "+v"(tw[5]), "+v"(tw[6]), "+v"(tw[7]), "+v"(tw[8]),
                                        "+v"(tw[9]), "+v"(tw[10]), "+v"(tw[11]), "+v"(tw[12]), "+v"(tw[13]), "+v"(tw[14]), "+v"(tw[15]), "+v"(tw[16]), "+v"(tw[17]));
}

template <int ROWB> __device__ __forceinline__ void ft_load_half(ffl_f32x4 (&nx)[32], const FwRows &x, int h)
{
#pragma unroll
    for (int m = 0; m < 16; m++) nx[2 * m + h] = ffl_buf_load4(x.r, x.v0 + ROWB * (2 * m + h), 0, 0);
}
template <int ROWB, int H> __device__ __forceinline__ void ft_pass3_half(fw_pk2 (&v)[64], const FwRows &y, ffl_f32x4 (&nx)[32], const FwRows &xn)
{
    fw_pk_dft16<32 * H, true>(v); fw_pk_dft16<32 * H + 16, true>(v);
#pragma unroll
    for (int m = 0; m < 16; m++) {                                      // rows 4 m + 2 H (group 2 H, k2 = m) and 4 m + 2 H + 1 (group 2 H + 1, k2 = m)
        fw_pk2 P = v[16 * (2 * H) + m], Q = v[16 * (2 * H + 1) + m];
        fw_swap_halves(P, Q);
        const ffl_f32x4 r = {P.x, P.y, Q.x, Q.y};
        ffl_buf_store4(r, y.r, y.v0 + ROWB * (2 * m + H), 0, 0);
    }
    ft_load_half<ROWB>(nx, xn, H);
}

// One team per window; windows dealt like the other kernels' (every XCD a contiguous range, consecutive teams consecutive windows).  Edges by the buffer range check.
// tw1[e T + p]: the bases of W_N^(t k_a) for thread p (t = ft_logical(p)); tw2[e M + m]: those of W_T^(m k_c); hw: the spectrum in ft_h_index order.
template <int M>
__global__ __launch_bounds__(64 * M, 2) void k_fftfilt_team(const float2 *__restrict__ in, size_t in_pitch, const float2 *__restrict__ hist, int k1p, int m_new,
                                                            int n_chunks, int n_windows, float2 *__restrict__ out, size_t out_pitch, const float2 *__restrict__ hw,
                                                            const float2 *__restrict__ g_tw1, const float2 *__restrict__ g_tw2)
{
    using G = FtGeom<M>;
    constexpr int T = G::T, N = G::N, ROWB = G::ROWB;
    extern __shared__ float4 ffl_raw[];
    float *L = reinterpret_cast<float *>(ffl_raw);
    const int p = threadIdx.x, lane = p & 63, t = ft_logical(p), m = p & (M - 1);
    float *xa = L + t, *xb = L + (p >> G::LOGM) * G::P + m;
    const float sig2 = (m & 2) ? -1.f : 1.f, sig1 = (m & 1) ? -1.f : 1.f;
    const bool lane3 = m == 3;
    const unsigned long long b1 = (unsigned long long)g_tw1, b2 = (unsigned long long)g_tw2;
    const ffl_i32x4 rt1 = {(int)(unsigned)b1, (int)((b1 >> 32) & 0xffffu), FW_TWE * T * 8, 0x00020000}, rt2 = {(int)(unsigned)b2, (int)((b2 >> 32) & 0xffffu), FW_TWE * M * 8, 0x00020000};
    const int V = N - k1p;
    const int per_xcd = (n_windows + 7) >> 3, xcd = blockIdx.x & 7, stride = gridDim.x >> 3;      // gridDim.x is a multiple of 8
    const int w_end = min(n_windows, (xcd + 1) * per_xcd);
    int w = xcd * per_xcd + (blockIdx.x >> 3);
    if (w >= w_end) return;
    // sample 2 (lane & 31) of the wave's 64 columns, in row 0 (lanes < 32) / row 1 (lanes >= 32) of a row pair: what fw_swap_halves turns into rows (2 jp, 2 jp + 1) of t
    const int n_lane = (p & ~63) + 2 * (lane & 31) + T * (lane >> 5);
    auto rows_in = [&](int win) {
        const int s = win / n_chunks, c = win - s * n_chunks;
        const unsigned long long bx = (unsigned long long)(in + (size_t)s * in_pitch);
        return FwRows{ffl_i32x4{(int)(unsigned)bx, (int)((bx >> 32) & 0xffffu), m_new * 8, 0x00020000}, (c * V - k1p + n_lane) * 8};
    };
    const unsigned long long bh = (unsigned long long)hw;
    const ffl_i32x4 rh = {(int)(unsigned)bh, (int)((bh >> 32) & 0xffffu), N * 8, 0x00020000};
    ffl_f32x4 nx[32];
    {
        const FwRows x0 = rows_in(w);
        ft_load_half<ROWB>(nx, x0, 0); ft_load_half<ROWB>(nx, x0, 1);
    }
    for (;;) {
        const int s = w / n_chunks, c = w - s * n_chunks, w0 = c * V - k1p;
        fw_pk2 v[64];
        fw_rows_to_regs(v, nx);
        if (w0 < 0) {                                                   // uniform: the stream's first window
            const unsigned long long bhs = (unsigned long long)(hist + (size_t)s * k1p);
            const ffl_i32x4 rhs = {(int)(unsigned)bhs, (int)((bhs >> 32) & 0xffffu), k1p * 8, 0x00020000};
            const int vh = (k1p + w0 + t) * 8;
#pragma unroll
            for (int j = 0; j < 64; j++) { const ffl_f32x2 r = ffl_buf_load(rhs, vh + T * 8 * j, 0, 0); v[j].x += r.x; v[j].y += r.y; }
        }
        {
            fw_pk2 tw[FW_TWE];
            ft_tw_issue<T * 8>(tw, rt1, p * 8);
            fw_pk_dft64<false>(v);
            ft_tw_ready(tw);
            fw_pk_twiddle<false>(v, tw);
        }
        ft_exchange<M, false>(v, xa, xb);
        ffl_f32x4 ha[4], hb[4];
        {
            fw_pk2 tw[FW_TWE];
            ft_tw_issue<M * 8>(tw, rt2, m * 8);
            ft_h_issue(ha, rh, p * 16, 0, ROWB);
            fw_pk_dft64<false>(v);
            ft_tw_ready(tw);
            fw_pk_twiddle<false>(v, tw);
            ft_h_issue(hb, rh, p * 16, 1, ROWB);
            ft_h_ready<4>(ha); ft_centre_chunk<M>(v, ha, 0, sig2, sig1, lane3); ft_h_issue(ha, rh, p * 16, 2, ROWB);
            ft_h_ready<4>(hb); ft_centre_chunk<M>(v, hb, 1, sig2, sig1, lane3); ft_h_issue(hb, rh, p * 16, 3, ROWB);
            ft_h_ready<4>(ha); ft_centre_chunk<M>(v, ha, 2, sig2, sig1, lane3); ft_h_issue(ha, rh, p * 16, 4, ROWB);
            ft_h_ready<4>(hb); ft_centre_chunk<M>(v, hb, 3, sig2, sig1, lane3); ft_h_issue(hb, rh, p * 16, 5, ROWB);
            ft_h_ready<4>(ha); ft_centre_chunk<M>(v, ha, 4, sig2, sig1, lane3); ft_h_issue(ha, rh, p * 16, 6, ROWB);
            ft_h_ready<4>(hb); ft_centre_chunk<M>(v, hb, 5, sig2, sig1, lane3); ft_h_issue(hb, rh, p * 16, 7, ROWB);
            ft_h_ready<4>(ha); ft_centre_chunk<M>(v, ha, 6, sig2, sig1, lane3);
            ft_h_ready<0>(hb); ft_centre_chunk<M>(v, hb, 7, sig2, sig1, lane3);
            fw_pk_twiddle<true>(v, tw);
        }
        {
            fw_pk2 tw[FW_TWE];
            ft_tw_issue<T * 8>(tw, rt1, p * 8);
            fw_pk_dft64<true>(v);
            ft_exchange<M, true>(v, xa, xb);
            ft_tw_ready(tw);
            fw_pk_twiddle<true>(v, tw);
        }
        // results n = k1p .. N-1 of the window are outputs c V + (n - k1p): descriptor based at output c V, range = what is left of the call
        const unsigned long long by = (unsigned long long)(out + (size_t)s * out_pitch + (size_t)c * V);
        const FwRows y = {ffl_i32x4{(int)(unsigned)by, (int)((by >> 32) & 0xffffu), (m_new - c * V) * 8, 0x00020000}, (n_lane - k1p) * 8};
        const int wn = w + stride;
        const bool more = wn < w_end;
        const FwRows xn = rows_in(more ? wn : w);                       // (the last window: this one again, ignored -- no branch around values in flight)
        fw_pk_dft64_head<true>(v);
        ft_pass3_half<ROWB, 0>(v, y, nx, xn); ft_pass3_half<ROWB, 1>(v, y, nx, xn);
        if (!more) break;
        w = wn;
    }
}
#endif

} // namespace
