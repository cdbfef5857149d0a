// experiments/fft64q.hip -- NOT built, NOT linked into libcsdr_amd.so (round 6 prune).  The literal 65536-point block of apply_fir_fft_cc (libcsdr.c:814-849) in TWO passes over HBM
// (radix-4 step in the time domain, four 16384-point transforms with the bin product in LDS, a four-point combine pass: 32 B per sample instead of the 48 B of the three-pass
// transform of csdr_amd/csrc/fft64k.hip).  Measured SLOWER than the three passes (0.83 against 0.70 ms per 64 x 16 blocks, profiles/r4_notes.md: the block transform is bound by
// fp32 vector adds, and one 1024-thread workgroup per CU exposes its loads and stores).  It lived in fftfilt_lds.hip (using that file's stage functions FflMidPhases / ffl_first /
// ffl_last) behind CSDR_AMD_FFT64Q=1 at commit 576ffd3, with a CPU twin (csdr_amd_debug_fft64q) and its parity tests.

// ================================================================================================ the LITERAL 65536-point block in two passes ("64q")
// apply_fir_fft_cc at fft_size 65536 (libcsdr.c:814-849; csdr.c:1810-1886) for taps too long for the windows above, or when the caller wants the reference's own
// block transform.  65536 = 4 x 16384, decimation in frequency with the radix-4 step FIRST, in the time domain, on the block's four contiguous quarters:
//     X[4k + q] = FFT_16384 over n of  v_q[n],   v_q[n] = W_N^(n q) * sum_r x[n + 16384 r] W_4^(r q)
//     y[n + 16384 r] = sum_q W_4^(-r q) u_q[n],  u_q[n] = W_N^(-n q) * IFFT_16384 over k of ( X[4k + q] H[4k + q] / N )
// Pass A (k_f64q_main), one workgroup per (block, q): reads the four quarters (coalesced; the other three workgroups of the block run on the same XCD at the same
// time, so HBM sees the block once), forms v_q, and then IS the one-pass kernel above -- 16384-point transform, bin product with H[4k + q] in the transform's own
// slot order, inverse transform, all in LDS -- and stores u_q.  Pass B (k_f64q_combine): four-point butterflies across the u_q, elementwise and coalesced, writing
// the block's own samples to `out` and its tail to the overlap buffer like k_f64_cols_inv_oa.  32 B per sample over HBM (8 + 8 + 8 + 8; the 8 + 8 in the middle
// stay in the Infinity Cache when the caller works through the batch in groups) against 48 B for the 256 x 256 three-pass form of fft64k.hip.
constexpr int Q_N = 65536, Q_M = 16384, Q_T = Q_M / 16;

// element n = t + 1024 j of v_q from the four quarter samples; bq = (W_N^t)^q, cq[j] = W_64^(j q)
FFL_HD float2 f64q_pick(float2 x0, float2 x1, float2 x2, float2 x3, int q)
{
    dft4<false>(x0, x1, x2, x3);
    return q == 0 ? x0 : q == 1 ? x1 : q == 2 ? x2 : x3;
}

// the phases between load and store: exactly the one-pass kernel's (sync = barrier on the device)
#define F64Q_PHASES(SYNC)                                                                                   \
    FflMidPhases<Q_M>::template run<0>(lds, tws, hq, t); SYNC;                                              \
    FflMidPhases<Q_M>::template run<1>(lds, tws, hq, t); SYNC;                                              \
    FflMidPhases<Q_M>::template run<2>(lds, tws, hq, t); SYNC;                                              \
    FflMidPhases<Q_M>::template run<3>(lds, tws, hq, t); SYNC;                                              \
    FflMidPhases<Q_M>::template run<4>(lds, tws, hq, t); SYNC;

// grid: 32 ids per group of 8 blocks: id = 32 g + 8 q + x  <->  block 8 g + x, quarter-frequency q: the four workgroups of a block get ids that are equal mod 8,
// i.e. the same XCD (ids are dealt round robin), and are dispatched together
__global__ __launch_bounds__(Q_T, 1) void k_f64q_main(const float2 *__restrict__ in, size_t in_pitch, int inp, int n_blocks, int batch, float2 *__restrict__ work,
                                                      const float2 *__restrict__ g_hq, const float2 *__restrict__ g_tw1, const float2 *__restrict__ g_tws,
                                                      const float2 *__restrict__ g_twn)
{
    using G = FflGeom<Q_M>;
    extern __shared__ float4 ffl_raw[];
    float2 *lds = reinterpret_cast<float2 *>(ffl_raw), *tws = lds + G::DATA;
    const int t = threadIdx.x;
    const int id = blockIdx.x, q = __builtin_amdgcn_readfirstlane((id >> 3) & 3), blk = (id >> 5) * 8 + (id & 7);
    if (blk >= batch) return;
    for (int i = t; i < G::TWN; i += G::T) tws[i] = g_tws[i];
    const float2 w1 = g_tw1[t];
    const float2 *hq = g_hq + (size_t)q * Q_M;
    const int s = blk / n_blocks, b = blk - s * n_blocks;
    const float2 *x = in + (size_t)s * in_pitch + (size_t)b * inp;
    const unsigned long long bx = (unsigned long long)x;
    const ffl_i32x4 rx = {(int)(unsigned)bx, (int)((bx >> 32) & 0xffffu), inp * 8, 0x00020000};      // samples behind input_size read as zero: the block's padding (csdr.c:1864)
    // W_N^(n q) from a table per q (64 KiB each, L2 resident; a power tree in registers -- 32 more live registers at 128 per thread -- spilled 35 of them);
    // the loads in groups of four j = 16 + 4 in flight (all 64 at once would need every register of the thread)
    const float2 *twq = g_twn + (size_t)q * Q_M;
    float2 v[16];
#pragma unroll
    for (int j0 = 0; j0 < 16; j0 += 4) {
#pragma unroll
        for (int j = j0; j < j0 + 4; j++) {
            const int n = t + Q_T * j;
            const ffl_f32x2 a0 = ffl_buf_load(rx, n * 8, 0, 0), a1 = ffl_buf_load(rx, (n + Q_M) * 8, 0, 0), a2 = ffl_buf_load(rx, (n + 2 * Q_M) * 8, 0, 0),
                            a3 = ffl_buf_load(rx, (n + 3 * Q_M) * 8, 0, 0);
            const float2 a = f64q_pick(make_float2(a0.x, a0.y), make_float2(a1.x, a1.y), make_float2(a2.x, a2.y), make_float2(a3.x, a3.y), q);
            v[j] = q ? cmul(a, twq[n]) : a;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    ffl_first<Q_M>(v, lds, w1, t);
    __syncthreads();
    F64Q_PHASES(__syncthreads())
    ffl_last<Q_M>(v, lds, w1, t);
    float2 *dst = work + ((size_t)blk * 4 + q) * Q_M;
#pragma unroll
    for (int j = 0; j < 16; j++) dst[t + Q_T * j] = q ? cmul(v[j], cconj(twq[t + Q_T * j])) : v[j];
}

// y[n + M r] = sum_q W_4^(-r q) u_q[n]; samples < input_size to the output row, the rest (the block's tail, taps - 1 samples) to tails[batch][ovl]
__global__ __launch_bounds__(256) void k_f64q_combine(const float2 *__restrict__ work, float2 *__restrict__ out, size_t out_pitch, float2 *__restrict__ tails,
                                                      int inp, int ovl, int n_blocks)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    const size_t blk = blockIdx.y, s = blk / n_blocks, b = blk % n_blocks;
    const float2 *u = work + blk * 4 * Q_M + n;
    float2 y0 = u[0], y1 = u[Q_M], y2 = u[2 * Q_M], y3 = u[3 * Q_M];
    dft4<true>(y0, y1, y2, y3);
    float2 *o = out + s * out_pitch + b * (size_t)inp, *tl = tails + blk * (size_t)ovl;
    const float2 y[4] = {y0, y1, y2, y3};
#pragma unroll
    for (int r = 0; r < 4; r++) { const int m = n + Q_M * r; if (m < inp) o[m] = y[r]; else tl[m - inp] = y[r]; }
}

// tables: H[4 f + q] / N at the slot of frequency f of the 16384-point plan, q-major; the plan's own twiddles; W_N^(n q), q-major
void f64q_host_tables(const cf32 *taps, int taps_len, std::vector<float2> &hq, std::vector<float2> &tw1, std::vector<float2> &tws, std::vector<float2> &twn)
{
    using G = FflGeom<Q_M>;
    std::vector<double> re(Q_N, 0.0), im(Q_N, 0.0);
    for (int k = 0; k < taps_len; k++) { re[k] = taps[k].i; im[k] = taps[k].q; }
    host_dft_pow2(re, im);
    hq.resize((size_t)4 * Q_M);
    for (int q = 0; q < 4; q++)
        for (int slot = 0; slot < 16; slot++)
            for (int t = 0; t < G::T; t++) {
                const int f = ffl_freq_of_position<Q_M>(ffl_slot_position<Q_M>(slot, t));
                hq[(size_t)q * Q_M + (size_t)slot * G::T + t] = make_float2((float)(re[4 * f + q] / Q_N), (float)(im[4 * f + q] / Q_N));
            }
    tw1.resize(G::T); tws.resize(G::TWN); twn.resize((size_t)4 * Q_M);
    for (int t = 0; t < G::T; t++) { const double a = -2.0 * M_PI * t / Q_M; tw1[t] = make_float2((float)cos(a), (float)sin(a)); }
    for (int e = 0; e < G::TWN; e++) { const double a = -2.0 * M_PI * e / G::TWN; tws[e] = make_float2((float)cos(a), (float)sin(a)); }
    for (int q = 0; q < 4; q++)
        for (int n = 0; n < Q_M; n++) { const double a = -2.0 * M_PI * (double)n * q / Q_N; twn[(size_t)q * Q_M + n] = make_float2((float)cos(a), (float)sin(a)); }
}

// both passes on the CPU for ONE block: the same functions, one "thread" after the other (tests/test_abi_cpu.py); y: 65536 samples (own part and tail)
void f64q_host_run(const cf32 *taps, int taps_len, const cf32 *xin, int inp, cf32 *y)
{
    using G = FflGeom<Q_M>;
    std::vector<float2> hqa, tw1, tws_v, twn; f64q_host_tables(taps, taps_len, hqa, tw1, tws_v, twn);
    std::vector<float2> lds_v(G::DATA), work((size_t)4 * Q_M);
    float2 *lds = lds_v.data(); const float2 *tws = tws_v.data();
    auto xs = [&](int m) { return m < inp ? make_float2(xin[m].i, xin[m].q) : make_float2(0.f, 0.f); };
    for (int q = 0; q < 4; q++) {
        const float2 *hq = hqa.data() + (size_t)q * Q_M;
        const float2 *twq = twn.data() + (size_t)q * Q_M;
        for (int t = 0; t < G::T; t++) {
            float2 v[16];
            for (int j = 0; j < 16; j++) {
                const int n = t + Q_T * j;
                const float2 a = f64q_pick(xs(n), xs(n + Q_M), xs(n + 2 * Q_M), xs(n + 3 * Q_M), q);
                v[j] = q ? cmul(a, twq[n]) : a;
            }
            ffl_first<Q_M>(v, lds, tw1[t], t);
        }
        for (int t = 0; t < G::T; t++) FflMidPhases<Q_M>::template run<0>(lds, tws, hq, t);
        for (int t = 0; t < G::T; t++) FflMidPhases<Q_M>::template run<1>(lds, tws, hq, t);
        for (int t = 0; t < G::T; t++) FflMidPhases<Q_M>::template run<2>(lds, tws, hq, t);
        for (int t = 0; t < G::T; t++) FflMidPhases<Q_M>::template run<3>(lds, tws, hq, t);
        for (int t = 0; t < G::T; t++) FflMidPhases<Q_M>::template run<4>(lds, tws, hq, t);
        for (int t = 0; t < G::T; t++) {
            float2 v[16];
            ffl_last<Q_M>(v, lds, tw1[t], t);
            for (int j = 0; j < 16; j++) work[(size_t)q * Q_M + t + Q_T * j] = q ? cmul(v[j], cconj(twq[t + Q_T * j])) : v[j];
        }
    }
    for (int n = 0; n < Q_M; n++) {
        float2 y0 = work[n], y1 = work[Q_M + n], y2 = work[2 * Q_M + n], y3 = work[3 * Q_M + n];
        dft4<true>(y0, y1, y2, y3);
        y[n] = cf32{y0.x, y0.y}; y[n + Q_M] = cf32{y1.x, y1.y}; y[n + 2 * Q_M] = cf32{y2.x, y2.y}; y[n + 3 * Q_M] = cf32{y3.x, y3.y};
    }
}


// ---- the two-pass 65536-point block filter (see "64q" above)
struct Fft64q { float2 *d_hq, *d_tw1, *d_tws, *d_twn; };

void fft64q_destroy(Fft64q *p)
{
    if (!p) return;
    (void)hipFree(p->d_hq); (void)hipFree(p->d_tw1); (void)hipFree(p->d_tws); (void)hipFree(p->d_twn);
    delete p;
}
int fft64q_set_taps(Fft64q *p, hipStream_t st, const cf32 *taps, int taps_len)
{
    std::vector<float2> hq, tw1, tws, twn; f64q_host_tables(taps, taps_len, hq, tw1, tws, twn);
    CSDR_HIP(hipStreamSynchronize(st));
    CSDR_HIP(hipMemcpy(p->d_hq, hq.data(), sizeof(float2) * hq.size(), hipMemcpyHostToDevice));
    CSDR_HIP(hipMemcpy(p->d_tw1, tw1.data(), sizeof(float2) * tw1.size(), hipMemcpyHostToDevice));
    CSDR_HIP(hipMemcpy(p->d_tws, tws.data(), sizeof(float2) * tws.size(), hipMemcpyHostToDevice));
    CSDR_HIP(hipMemcpy(p->d_twn, twn.data(), sizeof(float2) * twn.size(), hipMemcpyHostToDevice));
    return 0;
}
Fft64q *fft64q_create(hipStream_t st, const cf32 *taps, int taps_len)
{
    Fft64q *p = new Fft64q();
    p->d_hq = p->d_tw1 = p->d_tws = p->d_twn = nullptr;
    hipError_t e = hipMalloc((void **)&p->d_hq, sizeof(float2) * 4 * Q_M);
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_tw1, sizeof(float2) * Q_T);
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_tws, sizeof(float2) * FflGeom<Q_M>::TWN);
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_twn, sizeof(float2) * 4 * Q_M);
    if (e != hipSuccess) { fail(e, "hipMalloc(fft64q)", __FILE__, __LINE__); fft64q_destroy(p); return nullptr; }
    if (fft64q_set_taps(p, st, taps, taps_len)) { fft64q_destroy(p); return nullptr; }
    return p;
}
// passes A and B for `ns` streams x n_blocks blocks: out gets every block's own input_size samples, d_tails [ns * n_blocks][ovl] the tails (the caller adds them)
int fft64q_filter(Fft64q *p, hipStream_t st, const cf32 *in, size_t in_pitch, int inp, int ovl, int n_blocks, int ns, cf32 *d_work, cf32 *d_tails, cf32 *out, size_t out_pitch)
{
    using G = FflGeom<Q_M>;
    int rc = lds_attr_once((const void *)k_f64q_main, G::LDS_BYTES); if (rc) return rc;
    const int batch = n_blocks * ns;
    const unsigned grid = (unsigned)((batch + 7) / 8) * 32u;
    hipLaunchKernelGGL(k_f64q_main, dim3(grid), dim3(Q_T), G::LDS_BYTES, st, (const float2 *)in, in_pitch, inp, n_blocks, batch, (float2 *)d_work, (const float2 *)p->d_hq,
                       (const float2 *)p->d_tw1, (const float2 *)p->d_tws, (const float2 *)p->d_twn);
    CSDR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_f64q_combine, dim3(Q_M / 256, batch), dim3(256), 0, st, (const float2 *)d_work, (float2 *)out, out_pitch, (float2 *)d_tails, inp, ovl, n_blocks);
    CSDR_LAUNCH_CHECK();
    return 0;
}


// Test hook (CPU, no device): the two-pass 65536-point block (k_f64q_main + k_f64q_combine) on ONE block of `inp` samples; y: 65536 samples = the block's circular
// convolution with the taps (own part [0, inp), tail behind it)
extern "C" int csdr_amd_debug_fft64q(const float *taps_iq, int taps_len, const float *x_iq, int inp, float *y_iq)
{
    if (taps_len < 1 || inp < 1 || inp > Q_N) return -3;
    f64q_host_run(reinterpret_cast<const cf32 *>(taps_iq), taps_len, reinterpret_cast<const cf32 *>(x_iq), inp, reinterpret_cast<cf32 *>(y_iq));
    return 0;
}
