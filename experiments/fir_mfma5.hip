// experiments/fir_mfma5.hip -- NOT built, NOT linked into libcsdr_amd.so (round 6 prune).  k_fir_mfma5: k_fir_mfma3's construction (resident taps operand, LDS-DMA window) on the
// SHORT filter of BASELINE config 1 (fir_decimate_cc 10 / 79).  Measured: exactly as fast as k_fir_poly on every process (0.971 vs 0.964 ms, 1.091 vs 1.081 on a slow one:
// profiles/r5_notes.md, profiles/r5_fir_c1_bimodal_runs.txt) -- config 1 is bound by the memory system for this access pattern, not by the kernel.  It compiled inside fir.hip's
// anonymous namespace at commit 576ffd3; its dispatch lived in csdr_amd_fir_decimate_cc behind CSDR_AMD_FIR_MFMA5=1.

// k_fir_mfma5 (round 5): SHORT filters on complexf (15 D + taps <= 256: fir_decimate_cc 10 / 79, BASELINE config 1) with k_fir_mfma3's means.  A tile of 128 outputs
// reads only ~11 KiB here, so the K range is NOT split: each of the four waves owns a whole tile of its own (all 15-16 blocks of four K-steps, the taps operand -- the
// same for every wave -- resident in 64 registers, no reduction through LDS), a workgroup step is FOUR consecutive tiles behind one window of ~43 KiB that arrives by
// LDS-DMA (three workgroups per CU: while one multiplies, the others' windows land), two barriers per four tiles.  The band is 34 % dense: three times the flops of the
// scalar kernel, on a pipe that has the room (k_fir_poly: 18 of 157 TFLOP/s).  Outputs leave as float2 (re from the even column's lane, im from its odd neighbour).
template <int MAXB>
__global__ __launch_bounds__(256, 3) void k_fir_mfma5(const float2 *__restrict__ in, float2 *__restrict__ out, int n_out, int input_size, size_t in_pitch, size_t out_pitch,
                                                      int D, const float *__restrict__ taps, int L, int steps_per_wg)
{
    extern __shared__ float4 lds_raw[];
    constexpr int NT = 8, NW = 4, NTHR = 256, TO = 16 * NT, TPI = 4;   // tiles per workgroup step
    const int PAD = 15 * D, KT = 15 * D + L, ksteps = (KT + 3) / 4, nblk = (ksteps + 3) / 4;
    const int n_pieces = (8 * (TO * D * (TPI - 1) + 16 * D * (NT - 1) + 16 * nblk + 8) + 1023) >> 10;      // 1-KiB DMA pieces of a step's window
    const int ppw = (n_pieces + NW - 1) / NW;                          // pieces per wave (<= 16)
    float *xw = reinterpret_cast<float *>(lds_raw);
    float *hz = xw + 256 * n_pieces;                                  // PAD zeros, the taps, zeros up to 16 nblk + 16 floats
    const size_t s = blockIdx.y;
    const int t = threadIdx.x;
    const int n_tiles = (n_out + TO - 1) / TO, n_steps = (n_tiles + TPI - 1) / TPI;
    const int step0 = blockIdx.x * steps_per_wg, step1 = min(step0 + steps_per_wg, n_steps);
    if (step0 >= n_steps) return;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, i = lane & 15, kk = lane >> 4;
    const uint8_t *row_base = reinterpret_cast<const uint8_t *>(in + s * in_pitch);
    const uint32_t xw_addr = (uint32_t)(size_t)(__attribute__((address_space(3))) float *)xw;
    auto stage = [&](int step) {
        const long long first = (long long)step * TPI * TO * D;       // first sample of the window (even)
        long long gmax = ((long long)input_size - first - 2) >> 1;    // last granule inside the stream
        if (gmax < 0) gmax = 0;
        const uint8_t *sbase = row_base + first * 8;
        for (int r = 0; r < ppw; r++) {
            const int piece = ppw * wave + r;
            if (piece >= n_pieces) break;                             // (wave uniform)
            const uint32_t gd = 64u * (uint32_t)piece + lane;
            const uint32_t gs = gd ^ ((gd >> 5) & 7u);                // the readers' swizzle, on 16-byte granules, applied to the source
            const uint32_t vo = 16u * (uint32_t)min((long long)gs, gmax);
            const uint32_t la = __builtin_amdgcn_readfirstlane((int)(xw_addr + 1024u * (uint32_t)piece));
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(vo), "s"(sbase), "s"(la) : "memory");
        }
    };
    stage(step0);
    for (int k = t; k < PAD + 16 * nblk + 16; k += NTHR) { const int ti = k - PAD; hz[k] = (ti >= 0 && ti < L) ? taps[ti] : 0.f; }
    const int n = lane & 15, g = n >> 1, part = n & 1;
    __syncthreads();                                                  // hz is complete
    float A[4 * MAXB];
    {
        const float *ap = hz + PAD + kk - D * i;
#pragma unroll
        for (int j = 0; j < MAXB; j++)
#pragma unroll
            for (int u = 0; u < 4; u++) A[4 * j + u] = (j < nblk) ? ap[16 * j + 4 * u] : 0.f;
    }
    const int c = 2 * kk + part;
    const int h0 = D * (NT * wave + g);                               // a >> 5 of this wave's tile, group g, block 0
    const int h_last = h0 + nblk - 1;
    typedef float e_v2f __attribute__((ext_vector_type(2))); typedef __attribute__((address_space(1))) e_v2f *gp_f2;
#ifndef FIR5_DIAG
#define FIR5_DIAG 0     // timing experiment: 1 = every tile's outputs go to the stream's first tile (stores that never leave L2): what the output stream costs
#endif
#ifndef FIR5_HOLD
#define FIR5_HOLD 4
#endif
#ifndef FIR5_NT
#define FIR5_NT 0
#endif
    gp_f2 obase = (gp_f2)(out + s * out_pitch);
    e_v2f hold[FIR5_HOLD][2];
    for (int step = step0; step < step1; step++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's pieces (and its stores of the previous step)
        __syncthreads();
        const int tile = step * TPI + wave;
        f32x4_mfma acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        if (tile < n_tiles) {                                          // (wave uniform)
            int hi = h0;
            asm volatile("" : "+v"(hi));
            float bA[8], bB[8];
            auto issue = [&](float (&bv)[8], int blk0, const int nb) {
#pragma unroll
                for (int jj = 0; jj < 2; jj++) {
                    if (jj >= nb) break;
                    const int h2 = min(blk0 + jj, h_last);
                    const int m = h2 & 28;
                    const uint32_t a0 = xw_addr + ((uint32_t)h2 << 7) + ((uint32_t)(c ^ m) << 2);
#pragma unroll
                    for (int u = 0; u < 4; u++) asm volatile("ds_read_b32 %0, %1" : "=v"(bv[4 * jj + u]) : "v"(a0 ^ (uint32_t)(u << 5)) : "memory");
                }
            };
            constexpr int NBATCH = (MAXB + 1) / 2;
            auto blocks_of = [](int q) { return (2 * q + 1 < MAXB) ? 2 : 1; };
            issue(bA, hi, blocks_of(0));
#pragma unroll
            for (int q = 0; q < NBATCH; q++) {
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
        }
        __syncthreads();                                              // every wave has left the window: the next one may land
        if (step + 1 < step1) stage(step + 1);
        // C layout: column = lane & 15 = (g, part), row = 4 kk + r: output 16 g + 4 kk + r of the tile.  Even lanes take rows 0, 1, odd lanes rows 2, 3, as float2.
        // The outputs of FIR5_HOLD consecutive steps wait in registers and leave together: beside a saturated read stream the memory charges a thin stream of stores by the
        // store EVENT (config 1 without its output stream: 0.89 ms instead of 1.08, for 9 % of the bytes) -- four steps = 16 KiB contiguous per workgroup burst.
        {
            float v[4], o[4];
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = acc[r] + acc1[r];
#pragma unroll
            for (int r = 0; r < 4; r++) o[r] = __shfl_xor(v[r], 1);   // the other part of the same output
            const int hs = (step - step0) % FIR5_HOLD;
#pragma unroll
            for (int q = 0; q < FIR5_HOLD; q++) if (q == hs) {
                hold[q][0] = part ? e_v2f{o[2], v[2]} : e_v2f{v[0], o[0]};
                hold[q][1] = part ? e_v2f{o[3], v[3]} : e_v2f{v[1], o[1]};
            }
            if (hs == FIR5_HOLD - 1 || step + 1 == step1) {
#pragma unroll
                for (int q = 0; q < FIR5_HOLD; q++) {
                    if (q > hs) break;
                    const int tq = (step - hs + q) * TPI + wave;
                    const int ob = (FIR5_DIAG == 1 ? 0 : tq) * TO + 16 * g + 4 * kk + (part ? 2 : 0);
                    if (tq < n_tiles) {
                        if (ob < n_out) { if (FIR5_NT) __builtin_nontemporal_store(hold[q][0], &obase[ob]); else obase[ob] = hold[q][0]; }
                        if (ob + 1 < n_out) { if (FIR5_NT) __builtin_nontemporal_store(hold[q][1], &obase[ob + 1]); else obase[ob + 1] = hold[q][1]; }
                    }
                }
            }
        }
    }
}

