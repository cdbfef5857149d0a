// fftfilt_wave.hpp -- the 4096-point window of the one-pass FFT filter (fftfilt_lds.hip) with ONE WAVE PER WINDOW (round 6).  Included by fftfilt_lds.hip only.
//
// The 256-thread form (16 points per thread, three radix-16 stages, four exchanges through LDS with a workgroup barrier each) spends a window's life waiting: its
// memory phase and its butterflies add up instead of overlapping (0.20 + 0.11 ms per 64 x 16 blocks at 1023 taps), and no resource is busy more than half of the time
// (vector ALU 42 %, LDS ~35 %, HBM 55 %).  Here a window is 4096 = 64 x 64: lane t holds the 64 points x[t + 64 j], so
//
//   pass 0  radix 64 over j in registers            (lane = n1, register = n2 -> k2),  x W_4096^(n1 k2)
//   T1      64 x 64 transpose through LDS           (lane = k2, register = n1)         wave-private: no barrier anywhere in the kernel
//   pass 1  radix 64 over n1 in registers           -> X[64 k1 + k2] in register k1 of lane k2,  x taps spectrum (1/N folded in)
//   pass 2  inverse radix 64 over k1                -> register n1, x conj W_4096^(n1 k2)
//   T2      transpose back                          (lane = n1, register = k2)
//   pass 3  inverse radix 64 over k2                -> y[n1 + 64 n2] in register n2 of lane n1
//
// Two exchanges instead of four, 2.5 x less LDS traffic, and a wave never waits for another one: the second wave of the SIMD (another window, in whatever phase it
// happens to be) fills every wait for HBM.  128 of the 256 registers hold the window; the real and imaginary halves go through the transposes one after the other,
// so a wave needs 16.25 KiB of LDS (eight waves per CU = two per SIMD).
#pragma once
#include <type_traits>

namespace {

constexpr int FW_N = 4096;
constexpr int FW_LP = 65;                                  // LDS row pitch in floats (64 + 1: the transposing writes of 32 lanes hit 32 banks)
constexpr int FW_WAVES = 4;                                // waves per workgroup (two workgroups per CU)
constexpr int FW_TWE = 18;                                 // per-lane twiddle table entries: w^1..w^3, (w^4)^1..(w^4)^15 with w = W_4096^lane

// cos(2 pi m / 64), m = 0..16
constexpr float FW_C64[17] = {1.0f, 0.99518472667219688624f, 0.98078528040323044913f, 0.95694033573220886494f, 0.92387953251128675613f, 0.88192126434835502971f,
                              0.83146961230254523708f, 0.77301045336273696081f, 0.70710678118654752440f, 0.63439328416364549822f, 0.55557023301960222474f,
                              0.47139673682599764856f, 0.38268343236508977173f, 0.29028467725446236764f, 0.19509032201612826785f, 0.09801714032956060199f, 0.0f};
FFL_HD constexpr float fw_cos64(int m) { const int q = (m & 63) >> 4, r = m & 15; return q == 0 ? FW_C64[r] : q == 1 ? -FW_C64[16 - r] : q == 2 ? -FW_C64[r] : FW_C64[16 - r]; }
FFL_HD constexpr float fw_sin64(int m) { const int q = (m & 63) >> 4, r = m & 15; return q == 0 ? FW_C64[16 - r] : q == 1 ? FW_C64[r] : q == 2 ? -FW_C64[16 - r] : -FW_C64[r]; }

// a x W_64^m (forward: exp(-2 pi i m / 64), inverse: the conjugate), m a compile-time constant after unrolling
template <bool INV> FFL_HD float2 fw_twid64(float2 a, int m)
{
    m &= 63;
    if (m == 0) return a;
    if (m == 16) return rot90<INV>(a);
    if (m == 32) return make_float2(-a.x, -a.y);
    if (m == 48) return rot90<!INV>(a);
    const float c = fw_cos64(m), s = INV ? fw_sin64(m) : -fw_sin64(m);
    return make_float2(a.x * c - a.y * s, a.x * s + a.y * c);
}

// 64-point DFT in registers, natural order in and out: n = 16 n1 + n2, k = k1 + 4 k2
template <bool INV> FFL_HD void dft64(float2 (&v)[64])
{
#pragma unroll
    for (int n2 = 0; n2 < 16; n2++) dft4<INV>(v[n2], v[16 + n2], v[32 + n2], v[48 + n2]);            // over n1: v[16 k1 + n2]
#pragma unroll
    for (int k1 = 1; k1 < 4; k1++)
#pragma unroll
        for (int n2 = 1; n2 < 16; n2++) v[16 * k1 + n2] = fw_twid64<INV>(v[16 * k1 + n2], n2 * k1);
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) dft16<INV>(*reinterpret_cast<float2 (*)[16]>(&v[16 * k1]));       // over n2: v[16 k1 + k2] = X[k1 + 4 k2]
    float2 o[64];
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++)
#pragma unroll
        for (int k2 = 0; k2 < 16; k2++) o[k1 + 4 * k2] = v[16 * k1 + k2];
#pragma unroll
    for (int k = 0; k < 64; k++) v[k] = o[k];
}

FFL_HD float2 fw_cmul_conj(float2 a, float2 b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }      // a x conj(b)

// v[k] x w^k (CONJ: x conj(w)^k) with w^k = w^(k & 3) (w^4)^(k >> 2); tw: the lane's FW_TWE entries
template <bool CONJ> FFL_HD void fw_twiddle(float2 (&v)[64], const float2 (&tw)[FW_TWE])
{
#pragma unroll
    for (int k = 1; k < 64; k++) {
        float2 a = v[k];
        if (k >> 2) a = CONJ ? fw_cmul_conj(a, tw[2 + (k >> 2)]) : cmul(a, tw[2 + (k >> 2)]);
        if (k & 3) a = CONJ ? fw_cmul_conj(a, tw[(k & 3) - 1]) : cmul(a, tw[(k & 3) - 1]);
        v[k] = a;
    }
}

FFL_HD void fw_pass0(float2 (&v)[64], const float2 (&tw)[FW_TWE]) { dft64<false>(v); fw_twiddle<false>(v, tw); }
// h: the taps spectrum of this lane, h[k1 * hs] = H[64 k1 + lane] / N
FFL_HD void fw_pass1(float2 (&v)[64], const float2 *h, int hs)
{
    dft64<false>(v);
#pragma unroll
    for (int k = 0; k < 64; k++) v[k] = cmul(v[k], h[(size_t)k * hs]);                               // libcsdr.c:826-830 (and 836-839: the 1/N is in the table)
}
FFL_HD void fw_pass2(float2 (&v)[64], const float2 (&tw)[FW_TWE]) { dft64<true>(v); fw_twiddle<true>(v, tw); }
FFL_HD void fw_pass3(float2 (&v)[64]) { dft64<true>(v); }

#ifdef __HIPCC__
// ---- the same butterflies on packed f32 (device only).  A complex number is one 64-bit register pair; v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 work on both halves,
// pick either half of every source for either half of the result (op_sel / op_sel_hi) and negate per half (neg_lo / neg_hi) for free: a complex add is ONE instruction,
// a + (-+ i) b one, a product with a twiddle two, and every twiddle W_64^m comes out of the nine base pairs (cos, sin)(2 pi b / 64), b <= 8, by modifiers alone.
// On gfx950 a packed operation occupies the vector ALU twice as long as a plain one (tools/probes/pk_rate.hip: 2.0 against 1.1 ns per wave instruction on a saturated
// SIMD) -- no gain where four waves per SIMD keep the ALU busy (the 256-thread kernel: Makefile, FLAGS_fftfilt_lds) --, but ONE wave issues one instruction every
// 2.1-2.4 ns of either kind, and this kernel has two waves per SIMD: it is bound by what a wave can issue, and packed it issues half as much.  (The SLP vectoriser cannot
// be asked to do this: it packs scalars that do not sit in register pairs and surrounds every operation with moves -- 3126 v_mov and 732 spilled registers when tried.)
typedef float fw_pk2 __attribute__((ext_vector_type(2)));

template <int I, int N, class F> __device__ __forceinline__ void fw_static_for(F &&f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); fw_static_for<I + 1, N>(f); }
}

__device__ __forceinline__ fw_pk2 fw_pk_add_mi(fw_pk2 a, fw_pk2 b)     // a - i b = (a.x + b.y, a.y - b.x)
{ fw_pk2 r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ fw_pk2 fw_pk_add_pi(fw_pk2 a, fw_pk2 b)     // a + i b = (a.x - b.y, a.y + b.x)
{ fw_pk2 r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ fw_pk2 fw_pk_mul_mi(fw_pk2 a)               // -i a = (a.y, -a.x)
{ fw_pk2 r; asm("v_pk_add_f32 %0, 0, %1 op_sel:[0,1] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(a)); return r; }
__device__ __forceinline__ fw_pk2 fw_pk_mul_pi(fw_pk2 a)               // +i a = (-a.y, a.x)
{ fw_pk2 r; asm("v_pk_add_f32 %0, 0, %1 op_sel:[0,1] op_sel_hi:[0,0] neg_lo:[0,1]" : "=v"(r) : "v"(a)); return r; }

// a x w (CONJ: a x conj(w)), w in registers
template <bool CONJ> __device__ __forceinline__ fw_pk2 fw_pk_cmul(fw_pk2 a, fw_pk2 w)
{
    fw_pk2 t, r;
    if constexpr (!CONJ) {
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));                                              // (a.x w.x, a.x w.y)
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));              // (-a.y w.y + t.x, a.y w.x + t.y)
    } else {
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));                                // (a.x w.x, -a.x w.y)
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));                             // (a.y w.y + t.x, a.y w.x + t.y)
    }
    return r;
}

// a x (A + i B) with A = (NA ? - : +) base[IA], B = (NB ? - : +) base[1 - IA], base = (cos, sin)(2 pi b / 64) in scalar registers:
//   t = (a.x A, a.x B);  r = (-a.y B + t.x, a.y A + t.y)
#define FW_TW_CASE(IA, IB, NA, NB, NBF)                                                                                                                               \
    if constexpr (CODE == ((IA) | ((NA) << 1) | ((NB) << 2))) {                                                                                                       \
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0," #IA "] op_sel_hi:[0," #IB "] neg_lo:[0," #NA "] neg_hi:[0," #NB "]" : "=v"(t) : "v"(a), "s"(base));                  \
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1," #IB ",0] op_sel_hi:[1," #IA ",1] neg_lo:[0," #NBF ",0] neg_hi:[0," #NA ",0]" : "=v"(r) : "v"(a), "s"(base), "v"(t)); \
    }
template <int CODE> __device__ __forceinline__ fw_pk2 fw_pk_tw_base(fw_pk2 a, fw_pk2 base)
{
    fw_pk2 t, r;
    FW_TW_CASE(0, 1, 0, 0, 1) FW_TW_CASE(1, 0, 0, 0, 1) FW_TW_CASE(0, 1, 1, 0, 1) FW_TW_CASE(1, 0, 1, 0, 1)
    FW_TW_CASE(0, 1, 0, 1, 0) FW_TW_CASE(1, 0, 0, 1, 0) FW_TW_CASE(0, 1, 1, 1, 0) FW_TW_CASE(1, 0, 1, 1, 0)
    return r;
}
#undef FW_TW_CASE

// a x W_64^M (forward: exp(-2 pi i M / 64); INV: the conjugate)
template <int M, bool INV> __device__ __forceinline__ fw_pk2 fw_pk_twid64(fw_pk2 a)
{
    constexpr int q = M & 63, k = q >> 4, r = q & 15;
    if constexpr (r == 0) {
        if constexpr (k == 0) return a;
        else if constexpr (k == 2) return -a;
        else if constexpr ((k == 1) == INV) return fw_pk_mul_pi(a);
        else return fw_pk_mul_mi(a);
    } else {
        constexpr int b = r <= 8 ? r : 16 - r;
        constexpr int ci = r <= 8 ? 0 : 1;                              // where cos(r) sits in the base pair (sin(r): the other half)
        constexpr int ia = (k & 1) ? 1 - ci : ci;                       // quadrant k: (c, s), (-s, c), (-c, -s), (s, -c)
        constexpr int na = (k == 1 || k == 2) ? 1 : 0;
        constexpr int nb = ((k == 2 || k == 3) ? 1 : 0) ^ (INV ? 0 : 1);
        const fw_pk2 base = {FW_C64[b], FW_C64[16 - b]};
        return fw_pk_tw_base<ia | (na << 1) | (nb << 2)>(a, base);
    }
}

template <bool INV> __device__ __forceinline__ void fw_pk_dft4(fw_pk2 &x0, fw_pk2 &x1, fw_pk2 &x2, fw_pk2 &x3)
{
    const fw_pk2 s02 = x0 + x2, d02 = x0 - x2, s13 = x1 + x3, d13 = x1 - x3;
    x0 = s02 + s13; x2 = s02 - s13;
    x1 = INV ? fw_pk_add_pi(d02, d13) : fw_pk_add_mi(d02, d13);
    x3 = INV ? fw_pk_add_mi(d02, d13) : fw_pk_add_pi(d02, d13);
}

// 16-point DFT of v[B .. B + 15], natural order in and out (dft16 of fft_butterflies.hpp: n = 4 n1 + n2, k = k1 + 4 k2)
template <int B, bool INV> __device__ __forceinline__ void fw_pk_dft16(fw_pk2 (&v)[64])
{
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++) fw_pk_dft4<INV>(v[B + n2], v[B + 4 + n2], v[B + 8 + n2], v[B + 12 + n2]);
    fw_static_for<1, 4>([&](auto k1) { fw_static_for<1, 4>([&](auto n2) {
        constexpr int K1 = decltype(k1)::value, N2 = decltype(n2)::value;
        v[B + 4 * K1 + N2] = fw_pk_twid64<4 * K1 * N2, INV>(v[B + 4 * K1 + N2]); }); });
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) fw_pk_dft4<INV>(v[B + 4 * k1], v[B + 4 * k1 + 1], v[B + 4 * k1 + 2], v[B + 4 * k1 + 3]);
    fw_pk2 t;
    t = v[B + 1]; v[B + 1] = v[B + 4]; v[B + 4] = t;   t = v[B + 2]; v[B + 2] = v[B + 8]; v[B + 8] = t;   t = v[B + 3]; v[B + 3] = v[B + 12]; v[B + 12] = t;
    t = v[B + 6]; v[B + 6] = v[B + 9]; v[B + 9] = t;   t = v[B + 7]; v[B + 7] = v[B + 13]; v[B + 13] = t; t = v[B + 11]; v[B + 11] = v[B + 14]; v[B + 14] = t;
}

// dft64 up to its radix-16 groups (v[16 k1 + n2], group k1 still to be transformed over n2), the groups one by one, and the reordering v[16 k1 + k2] = X[k1 + 4 k2] -> natural
template <bool INV> __device__ __forceinline__ void fw_pk_dft64_head(fw_pk2 (&v)[64])
{
#pragma unroll
    for (int n2 = 0; n2 < 16; n2++) fw_pk_dft4<INV>(v[n2], v[16 + n2], v[32 + n2], v[48 + n2]);
    fw_static_for<1, 4>([&](auto k1) { fw_static_for<1, 16>([&](auto n2) {
        constexpr int K1 = decltype(k1)::value, N2 = decltype(n2)::value;
        v[16 * K1 + N2] = fw_pk_twid64<N2 * K1, INV>(v[16 * K1 + N2]); }); });
}
__device__ __forceinline__ void fw_pk_dft64_tail(fw_pk2 (&v)[64])
{
    fw_pk2 o[64];
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++)
#pragma unroll
        for (int k2 = 0; k2 < 16; k2++) o[k1 + 4 * k2] = v[16 * k1 + k2];
#pragma unroll
    for (int k = 0; k < 64; k++) v[k] = o[k];
}
template <bool INV> __device__ __forceinline__ void fw_pk_dft64(fw_pk2 (&v)[64])
{
    fw_pk_dft64_head<INV>(v);
    fw_pk_dft16<0, INV>(v); fw_pk_dft16<16, INV>(v); fw_pk_dft16<32, INV>(v); fw_pk_dft16<48, INV>(v);
    fw_pk_dft64_tail(v);
}
// v[k] x w^k (CONJ: x conj(w)^k), as fw_twiddle
template <bool CONJ> __device__ __forceinline__ void fw_pk_twiddle(fw_pk2 (&v)[64], const fw_pk2 (&tw)[FW_TWE])
{
#pragma unroll
    for (int k = 1; k < 64; k++) {
        fw_pk2 a = v[k];
        if (k >> 2) a = fw_pk_cmul<CONJ>(a, tw[2 + (k >> 2)]);
        if (k & 3) a = fw_pk_cmul<CONJ>(a, tw[(k & 3) - 1]);
        v[k] = a;
    }
}
#endif

} // namespace
