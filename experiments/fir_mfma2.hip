// experiments/fir_mfma2.hip -- NOT built, NOT linked into libcsdr_amd.so (round 6 prune).  k_fir_mfma2: the register-resident taps operand of the long-filter FIR without the
// LDS-DMA window; measured slower than k_fir_mfma3 (its successor, csdr_amd/csrc/fir.hip) and than the round-2 kernel k_fir_mfma: profiles/r5_notes.md, profiles/r5_fir50_*.
// Kept as the text of the experiment; it compiled inside fir.hip's anonymous namespace at commit 576ffd3.

// k_fir_mfma2 (round 5): the same product with the TAPS OPERAND RESIDENT IN REGISTERS.  A[i][k] = h[k - D i] depends on the lane and the K-step only -- not on the
// tile, not on the workgroup -- yet k_fir_mfma read it from LDS for every product (two 4-byte reads + ~4 address instructions per v_mfma), and the compiler, short of
// registers beside the 64 staging registers, funnelled the eight A values of a batch through ONE register pair: three `s_waitcnt lgkmcnt(0)` in the middle of every
// eight products (ISA of k_fir_mfma<8, 32>): ~680 cycles per eight products where the matrix pipe needs 256.  Here a wave loads its MAXS A values once per workgroup
// (a workgroup walks up to 16 tiles), the K loop is fully unrolled (register-indexed A), a batch is eight B reads in flight followed by eight products on two
// alternating accumulators, and the waves' K-ranges are whole blocks of four steps, so that the XOR swizzle of a block's four B addresses is one mask.
// Zero padding makes the surplus steps of the last wave exact zeros.  Same sums per (wave, accumulator) order as k_fir_mfma up to the split points: fp32 rounding noise.
template <int NT, int NS, int MAXB, int NW, int PD>                  // MAXB: blocks of four K-steps per wave (upper bound, compile time); NW waves per workgroup (the K split);
__global__ __launch_bounds__(64 * NW, (NW == 8 && PD == 0) ? 4 : 2) void k_fir_mfma2(      // PD: how many tiles ahead the window is fetched into registers (0: after the tile, 1, 2)
const float2 *__restrict__ in, float2 *__restrict__ out, int n_out, int input_size, size_t in_pitch, size_t out_pitch,
                                                   int D, const float *__restrict__ taps, int L, int tiles_per_wg)
{
    extern __shared__ float4 lds_raw[];
    constexpr int NTHR = 64 * NW;
    const int TO = 16 * NT, W = (TO - 1) * D + L, PAD = 15 * D, KT = 15 * D + L, steps = (KT + 3) / 4, nblk = (steps + 3) / 4;
    const int XW = (2 * (W + 8) + 63) & ~31;                           // window floats incl. the slack the surplus steps of the last block may read (zeros)
    float *xw = reinterpret_cast<float *>(lds_raw);
    float *hz = xw + XW;                                              // PAD zeros, the taps, zeros up to 16 nblk + 16 floats
    float *red = hz + PAD + 16 * nblk + 16;                           // NW x 256 partial results
    const size_t s = blockIdx.y;
    const int t = threadIdx.x;
    const int n_tiles = (n_out + TO - 1) / TO, tile0 = blockIdx.x * tiles_per_wg, tile1 = min(tile0 + tiles_per_wg, n_tiles);
    if (tile0 >= n_tiles) return;
    const float2 *base = in + s * in_pitch;
    // two register sets: the windows of the next TWO tiles are in flight (PREF; with the taps resident a tile's products take ~3600 cycles, less than a fetch under load)
    float2 v0[NS], v1[PD == 2 ? NS : 1];
    auto fetch = [&](float2 (&v)[NS], int tile) {
        const int first = tile * TO * D;
#pragma unroll
        for (int u = 0; u < NS; u++) { const int k = NTHR * u + t; v[u] = (k < W && first + k < input_size) ? base[(size_t)first + k] : make_float2(0.f, 0.f); }
    };
    fetch(v0, tile0);
    if constexpr (PD == 2) { if (tile0 + 1 < tile1) fetch(reinterpret_cast<float2 (&)[NS]>(v1), tile0 + 1); }
    for (int k = t; k < PAD + 16 * nblk + 16; k += NTHR) { const int ti = k - PAD; hz[k] = (ti >= 0 && ti < L) ? taps[ti] : 0.f; }
    for (int k = W + 8 + t; 2 * k < XW; k += NTHR) { const int a = 2 * k; *reinterpret_cast<float2 *>(xw + (a ^ ((a >> 5) & 30))) = make_float2(0.f, 0.f); }      // (the same swizzle as the staging, which stops at W + 8 samples: never written again)
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, i = lane & 15, kk = lane >> 4;
    const int b_lo = wave * nblk / NW, b_hi = (wave + 1) * nblk / NW;   // this wave's blocks of four steps (wave uniform)
    const int n = lane & 15, g = n >> 1, part = n & 1;
    const float bm = g < NT ? 1.f : 0.f;
    __syncthreads();                                                  // hz is complete
    // ---- the resident A operand: step 4 (b_lo + j) + u of this wave, zero beyond its range (and beyond the band: hz is zero padded)
    float A[4 * MAXB];
    {
        const float *ap = hz + PAD + kk - D * i + 16 * b_lo;
#pragma unroll
        for (int j = 0; j < MAXB; j++)
#pragma unroll
            for (int u = 0; u < 4; u++) A[4 * j + u] = (b_lo + j < b_hi) ? ap[16 * j + 4 * u] : 0.f;
    }
    // ---- B addresses: float address a = 32 (D g' + blk) + (c + 8 u), c = 2 kk + part < 8, swizzled a ^ ((a >> 5) & 30): inside a block of four steps a >> 5 is one value
    const int gq = g < NT ? g : 0;
    const int c = 2 * kk + part;
    const int h_last = D * gq + b_hi - 1 + (b_hi == b_lo);             // (blocks beyond the wave's range re-read its last one: multiplied by zero)
    const uint32_t xw_addr = (uint32_t)(size_t)(__attribute__((address_space(3))) float *)xw;
    static_assert(MAXB % 2 == 0, "batches of two blocks");
#ifndef FIR2_DIAG
#define FIR2_DIAG 0     // timing experiments: 1 = no products, 2 = the window staged for the first tile only, 3 = no global fetch after the first
#endif
    auto one_tile = [&](const int tile, float2 (&v)[NS]) {
#pragma unroll
        for (int u = 0; u < NS; u++) {
            if (FIR2_DIAG == 2 && tile != tile0) break;
            const int k = NTHR * u + t;
            if (k < W + 8) { const int a = 2 * k; *reinterpret_cast<float2 *>(xw + (a ^ ((a >> 5) & 30))) = v[u]; }
        }
        __syncthreads();
        if (PD > 0 && tile + PD < tile1 && FIR2_DIAG != 3) fetch(v, tile + PD);        // in flight during the products (PD = 2: this tile's and the next tile's)
        f32x4_mfma acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        int hi = D * gq + b_lo;                                        // a >> 5 of the block
        asm volatile("" : "+v"(hi));                                  // (per tile: otherwise every B address is hoisted out of the tile loop -- 386 registers, one wave per SIMD)
        // A batch = two blocks = eight B reads in flight, then eight products; the NEXT batch's reads are issued before this batch's products (two register sets).
        // The reads are inline asm with the wait counted by hand: written as plain loads the compiler, minimising live ranges, serialised them
        // (ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma, one LDS round trip per product).
        float bA[8], bB[8];
        auto issue = [&](float (&bv)[8], int blk0) {
#pragma unroll
            for (int jj = 0; jj < 2; jj++) {
                const int h2 = min(blk0 + jj, h_last);
                const int m = h2 & 30;
                const uint32_t a0 = xw_addr + ((uint32_t)h2 << 7) + ((uint32_t)(c ^ m) << 2);      // byte address of step u = 0; step u: the (u ^ (m >> 3))-th quarter of the row
#pragma unroll
                for (int u = 0; u < 4; u++) asm volatile("ds_read_b32 %0, %1" : "=v"(bv[4 * jj + u]) : "v"(a0 ^ (uint32_t)(u << 5)) : "memory");
            }
        };
        if (FIR2_DIAG != 1) issue(bA, hi);
#pragma unroll
        for (int j = 0; j < (FIR2_DIAG == 1 ? 0 : MAXB); j += 4) {  // two batches per trip
            if (j + 2 < MAXB) issue(bB, hi + j + 2);
            if (j + 2 < MAXB) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(bA[0]), "+v"(bA[1]), "+v"(bA[2]), "+v"(bA[3]), "+v"(bA[4]), "+v"(bA[5]), "+v"(bA[6]), "+v"(bA[7]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bA[0]), "+v"(bA[1]), "+v"(bA[2]), "+v"(bA[3]), "+v"(bA[4]), "+v"(bA[5]), "+v"(bA[6]), "+v"(bA[7]));
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[4 * j + u], NT == 8 ? bA[u] : bA[u] * bm, acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[4 * j + u + 1], NT == 8 ? bA[u + 1] : bA[u + 1] * bm, acc1, 0, 0, 0);
            }
            if (j + 2 < MAXB) {
                if (j + 4 < MAXB) issue(bA, hi + j + 4);
                if (j + 4 < MAXB) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(bB[0]), "+v"(bB[1]), "+v"(bB[2]), "+v"(bB[3]), "+v"(bB[4]), "+v"(bB[5]), "+v"(bB[6]), "+v"(bB[7]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bB[0]), "+v"(bB[1]), "+v"(bB[2]), "+v"(bB[3]), "+v"(bB[4]), "+v"(bB[5]), "+v"(bB[6]), "+v"(bB[7]));
#pragma unroll
                for (int u = 0; u < 8; u += 2) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[4 * (j + 2) + u], NT == 8 ? bB[u] : bB[u] * bm, acc, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[4 * (j + 2) + u + 1], NT == 8 ? bB[u + 1] : bB[u + 1] * bm, acc1, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) red[wave * 256 + r * 64 + lane] = acc[r] + acc1[r];
        __syncthreads();
        if (t < 256) {   // C layout: column = lane & 15 (n), row = 4 (lane >> 4) + reg (i)
            const int r = t >> 6, ln = t & 63;
            float sum = red[t];
#pragma unroll
            for (int wv = 1; wv < NW; wv++) sum += red[256 * wv + t];
            const int nn = ln & 15, gg = nn >> 1, pp = nn & 1, ii = 4 * (ln >> 4) + r;
            const int o = tile * TO + 16 * gg + ii;
            if (gg < NT && o < n_out) reinterpret_cast<float *>(out + s * out_pitch + o)[pp] = sum;
        }
        __syncthreads();
        if (PD == 0 && tile + 1 < tile1) fetch(v, tile + 1);
    };
    if constexpr (PD == 2) {
        for (int tile = tile0; tile < tile1; tile += 2) {
            one_tile(tile, v0);
            if (tile + 1 < tile1) one_tile(tile + 1, reinterpret_cast<float2 (&)[NS]>(v1));
        }
    } else {
        for (int tile = tile0; tile < tile1; tile++) one_tile(tile, v0);
    }
}

