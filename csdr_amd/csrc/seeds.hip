// seeds.hip -- per-stream chunk-seed tables (see seeds.hpp): the float phase bookkeeping of `csdr shift_addition_cc` (csdr.c:896-923 around libcsdr_gpl.c:27-52)
// for N streams with individual rates, one lane per stream, on a side stream that runs a few calls ahead of the data.
//
// Two tables take turns.  Table entry k of stream s = chunk (tab_first + k): its float starting phase ph[k][s], the seed c[k][s] = (cos, sin)(ph) as floats of the
// double-precision functions (what the reference's shift_addition_cc computes from its float argument), and -- for the streams whose rate makes the float phasor
// recurrence drift (ddc_mfma.hip) -- 32 correction factors per chunk.  While the data kernels read table A, the side stream fills table B from A's last phases;
// the context's stream only ever waits for an event that fired long ago (and then the wait is not even queued).  A retune or a call of unexpected size drops
// the prepared table and regenerates from the phases of the current one.
#include "seeds.hpp"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
using namespace csdr_amd;

namespace {

// one lane per stream: entries [0, cap) of the new table.  old == nullptr: stream start (phase 0 in front of chunk 0, and the "chunk" in front of it).
// Otherwise entries 0 and 1 are the old table's idx and idx + 1 (the history chunk and the first chunk of the call: both were advanced under the rate that was
// valid then), everything behind advances by the current rate.
template <bool RAISED>
__global__ __launch_bounds__(1024) void k_seed_phases(const float *__restrict__ rates, const float *__restrict__ old_ph, long idx, float *__restrict__ ph, size_t pitch, int k0, int k1, int n_streams)
{
    // entries [k0, k1) of the new table (k0 == 2: the table's start, entries 0 and 1 included; a later slice continues from entry k0 - 1, which the slice in front of it
    // on the same stream has written)
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    const float inc = (rates[s] * 2) * PI_F;                          // shift_addition_init libcsdr_gpl.c:83-86 (as ddc_mfma.hip / wfm.hip)
    const float step = inc * (float)1024;
    float p;
    if (k0 == 2) {
        float p0 = 0.f, p1 = 0.f;
        if (old_ph) { p0 = old_ph[(size_t)idx * pitch + s]; p1 = old_ph[(size_t)(idx + 1) * pitch + s]; }
        ph[s] = p0; ph[pitch + s] = p1;
        p = p1;
    } else
        p = ph[(size_t)(k0 - 1) * pitch + s];
    WrapPlan w; wrap_plan_init(w, step);
    if (RAISED) __builtin_amdgcn_s_setprio(3);                        // eight waves beside the data kernels' thousands: a chain of dependent operations, let it issue
    for (int k = k0; k < k1; k++) {
        p = wrap_plan_apply(w, p + step);                             // libcsdr_gpl.c:48-51, exactly (seeds.hpp)
        ph[(size_t)k * pitch + s] = p;
    }
}

__global__ __launch_bounds__(256) void k_seed_cossin(const float *__restrict__ ph, float2 *__restrict__ c, size_t count)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const double p = (double)ph[i];
    c[i] = make_float2((float)cos(p), (float)sin(p));               // libcsdr_gpl.c:33-34: cos / sin of the float phase, in double, stored as floats
}

__device__ __forceinline__ float2 cmulf(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// drift corrections (k_ddc_corr of ddc_mfma.hip with a rate per row): corr[(row * cap + k) * 32 + j] = (float recurrence started at the chunk's seed, after
// 32 j + 16 steps) / (seed * D^(32 j + 16)).  One lane per (row, chunk).
__global__ __launch_bounds__(64) void k_seed_corr(const float2 *__restrict__ c, size_t pitch, int cap, const int *__restrict__ row_stream, const float *__restrict__ rates,
                                                  const float2 *__restrict__ dtab, size_t dtab_stride, float2 *__restrict__ corr)
{
    const int k = blockIdx.x * 64 + threadIdx.x, row = blockIdx.y;
    if (k >= cap) return;
    const int s = row_stream[row];
    if (s < 0) return;
    const float inc = (rates[s] * 2) * PI_F;
    const float cd = (float)cos((double)inc), sd = (float)sin((double)inc);
    const float2 *dt = dtab + (size_t)s * dtab_stride;
    const float2 C = c[(size_t)k * pitch + s];
    float cc = C.x, ss = C.y;
    float2 *dst = corr + ((size_t)row * cap + k) * 32;
    for (int j = 0; j < 32; j++) {
#pragma unroll
        for (int i = 0; i < 16; i++) { const float c2 = cc * cd - ss * sd, s2 = ss * cd + cc * sd; cc = c2; ss = s2; }      // libcsdr_gpl.c:44-45
        {
            const float2 ref = cmulf(C, dt[32 * j + 16 + 2048]);
            const float inv = 1.0f / (ref.x * ref.x + ref.y * ref.y);
            dst[j] = make_float2((cc * ref.x + ss * ref.y) * inv, (ss * ref.x - cc * ref.y) * inv);
        }
#pragma unroll
        for (int i = 0; i < 16; i++) { const float c2 = cc * cd - ss * sd, s2 = ss * cd + cc * sd; cc = c2; ss = s2; }
    }
}

__global__ void k_seed_set(float *rates, int *corr_row, int *row_stream, int s, float r, int row, int old_row)
{
    rates[s] = r; corr_row[s] = row;
    if (old_row >= 0 && old_row != row) row_stream[old_row] = -1;
    if (row >= 0) row_stream[row] = s;
}

} // namespace

struct csdr_amd::SeedTables {
    csdr_amd_ctx *ctx;
    int n; size_t pitch; int cap; size_t per_call;
    hipStream_t side;
    float *d_rates; std::vector<float> rates;
    float *d_ph[2]; float2 *d_c[2]; float2 *d_corr[2];
    int *d_corr_row, *d_row_stream; std::vector<int> corr_row, row_stream; int corr_rows;      // rows allocated
    const float2 *d_dtab; size_t dtab_stride;
    long long first[2]; bool valid[2];
    hipEvent_t ev_ready[2], ev_free[2]; bool free_pending[2];
    int cur;
    bool fresh, dirty;
    // few streams (the CLI's one): the phase chain runs on the HOST (the same wrap_plan code, host build: ~25 ns per chunk step against ~130 ns for a lane of
    // k_seed_phases, whose one wave leaves 63 lanes idle) into pinned shadows of the two tables, uploaded on the side stream
    bool host_chain; float *h_ph[2]; bool h_used[2];
};

namespace {

int alloc_corr(SeedTables *t, int rows)
{
    for (int b = 0; b < 2; b++) {
        if (t->d_corr[b]) { CSDR_HIP(hipFree(t->d_corr[b])); t->d_corr[b] = nullptr; }
        if (rows > 0) CSDR_HIP(hipMalloc((void **)&t->d_corr[b], sizeof(float2) * 32 * (size_t)rows * t->cap));
    }
    if (t->d_row_stream) { CSDR_HIP(hipFree(t->d_row_stream)); t->d_row_stream = nullptr; }
    if (rows > 0) CSDR_HIP(hipMalloc((void **)&t->d_row_stream, sizeof(int) * rows));
    t->corr_rows = rows;
    t->row_stream.assign(rows, -1);
    return 0;
}

// queue the generation of table `dst` on the side stream: from table `src` at entry idx (src < 0: stream start), first chunk `first`
int generate(SeedTables *t, int dst, int src, long idx, long long first)
{
    hipStream_t ss = t->side;
    if (t->free_pending[dst]) { CSDR_HIP(hipStreamWaitEvent(ss, t->ev_free[dst], 0)); t->free_pending[dst] = false; }      // the data kernels that read it have finished
    if (t->host_chain) {
        if (t->h_used[dst]) CSDR_HIP(hipEventSynchronize(t->ev_ready[dst]));       // the previous upload out of this shadow has run (long ago)
        float *h = t->h_ph[dst]; const float *o = src >= 0 ? t->h_ph[src] : nullptr;
        for (int s = 0; s < t->n; s++) {
            const float inc = (t->rates[s] * 2) * PI_F, step = inc * (float)1024;  // as k_seed_phases
            const float p0 = o ? o[(size_t)idx * t->pitch + s] : 0.f, p1 = o ? o[(size_t)(idx + 1) * t->pitch + s] : 0.f;
            h[s] = p0; h[t->pitch + s] = p1;
            WrapPlan w; wrap_plan_init(w, step);
            float p = p1;
            for (int k = 2; k < t->cap; k++) { p = wrap_plan_apply(w, p + step); h[(size_t)k * t->pitch + s] = p; }
        }
        CSDR_HIP(hipMemcpyAsync(t->d_ph[dst], h, sizeof(float) * t->pitch * (size_t)t->cap, hipMemcpyHostToDevice, ss));
        t->h_used[dst] = true;
    }
    // One launch of 64-lane workgroups (a wave per CU on sixteen CUs at 1024 streams), wave priority raised: a wave of this kernel is a chain of dependent operations that
    // occupies its SIMD for the whole table (milliseconds), and a CU that holds one cannot take a workgroup of the per-stream WFM kernel (2 x 256 registers per SIMD).
    // Round 5 measured the alternatives (larger workgroups on fewer CUs, the table in slices, normal priority, the generator switched off as a ceiling: profiles/r5_notes.md);
    // none paid, their switches are gone.
    if (!t->host_chain) hipLaunchKernelGGL(k_seed_phases<true>, dim3(cdiv(t->n, 64)), dim3(64), 0, ss, t->d_rates, src >= 0 ? t->d_ph[src] : nullptr, idx, t->d_ph[dst], t->pitch, 2, t->cap, t->n);
    CSDR_LAUNCH_CHECK();
    const size_t count = t->pitch * (size_t)t->cap;
    hipLaunchKernelGGL(k_seed_cossin, dim3(cdiv(count, 256)), dim3(256), 0, ss, t->d_ph[dst], t->d_c[dst], count);
    CSDR_LAUNCH_CHECK();
    if (t->corr_rows > 0 && t->d_dtab) {
        hipLaunchKernelGGL(k_seed_corr, dim3(cdiv(t->cap, 64), t->corr_rows), dim3(64), 0, ss, t->d_c[dst], t->pitch, t->cap, t->d_row_stream, t->d_rates, t->d_dtab, t->dtab_stride, t->d_corr[dst]);
        CSDR_LAUNCH_CHECK();
    }
    CSDR_HIP(hipEventRecord(t->ev_ready[dst], ss));
    t->first[dst] = first; t->valid[dst] = true;
    return 0;
}

} // namespace

namespace csdr_amd {

SeedTables *seeds_create(csdr_amd_ctx *ctx, int n_streams, const float *rates, const float2 *d_dtab, size_t dtab_stride, size_t max_block_samples)
{
    SeedTables *t = new SeedTables();
    t->ctx = ctx; t->n = n_streams; t->pitch = ((size_t)n_streams + 63) & ~(size_t)63;
    t->per_call = max_block_samples / 1024 + 8;
    // calls ahead: up to 8, within ~96 MB for the two tables
    size_t k = (size_t)96 << 20; k /= 2 * t->per_call * t->pitch * 12; if (k > 8) k = 8; if (k < 2) k = 2;
    t->cap = (int)(k * t->per_call);
    t->side = nullptr; t->d_rates = nullptr; t->d_corr_row = nullptr; t->d_row_stream = nullptr; t->corr_rows = 0;
    t->d_dtab = d_dtab; t->dtab_stride = dtab_stride;
    for (int b = 0; b < 2; b++) { t->d_ph[b] = nullptr; t->d_c[b] = nullptr; t->d_corr[b] = nullptr; t->ev_ready[b] = nullptr; t->ev_free[b] = nullptr; t->valid[b] = false; t->free_pending[b] = false; t->first[b] = 0; }
    t->rates.assign(rates, rates + n_streams); t->corr_row.assign(n_streams, -1);
    t->cur = 0; t->fresh = true; t->dirty = false;
    t->h_ph[0] = t->h_ph[1] = nullptr; t->h_used[0] = t->h_used[1] = false;
    { const char *eh = getenv("CSDR_AMD_SEED_HOST"); const int max_host = eh ? atoi(eh) : 2; t->host_chain = n_streams <= max_host; }      // (0: always the device chain)
    hipError_t e = hipStreamCreateWithFlags(&t->side, hipStreamNonBlocking);
    for (int b = 0; b < 2 && e == hipSuccess && t->host_chain; b++) {
        e = hipHostMalloc((void **)&t->h_ph[b], sizeof(float) * t->pitch * t->cap, hipHostMallocDefault);
        if (e == hipSuccess) memset(t->h_ph[b], 0, sizeof(float) * t->pitch * t->cap);
    }
    if (e == hipSuccess) e = hipMalloc((void **)&t->d_rates, sizeof(float) * t->pitch);
    if (e == hipSuccess) e = hipMalloc((void **)&t->d_corr_row, sizeof(int) * t->pitch);
    for (int b = 0; b < 2 && e == hipSuccess; b++) {
        e = hipMalloc((void **)&t->d_ph[b], sizeof(float) * t->pitch * t->cap);
        if (e == hipSuccess) e = hipMalloc((void **)&t->d_c[b], sizeof(float2) * t->pitch * t->cap);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&t->ev_ready[b], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&t->ev_free[b], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipMemset(t->d_rates, 0, sizeof(float) * t->pitch);
    if (e == hipSuccess) e = hipMemset(t->d_corr_row, 0xff, sizeof(int) * t->pitch);
    for (int b = 0; b < 2 && e == hipSuccess; b++) e = hipMemset(t->d_ph[b], 0, sizeof(float) * t->pitch * t->cap);
    if (e == hipSuccess) e = hipMemcpy(t->d_rates, rates, sizeof(float) * n_streams, hipMemcpyHostToDevice);
    if (e != hipSuccess) { fail(e, "seeds_create", __FILE__, __LINE__); seeds_destroy(t); return nullptr; }
    return t;
}

void seeds_destroy(SeedTables *t)
{
    if (!t) return;
    if (t->side) (void)hipStreamSynchronize(t->side);
    (void)hipStreamSynchronize(t->ctx->stream);
    for (int b = 0; b < 2; b++) {
        (void)hipFree(t->d_ph[b]); (void)hipFree(t->d_c[b]); (void)hipFree(t->d_corr[b]);
        if (t->ev_ready[b]) (void)hipEventDestroy(t->ev_ready[b]);
        if (t->ev_free[b]) (void)hipEventDestroy(t->ev_free[b]);
    }
    (void)hipFree(t->d_rates); (void)hipFree(t->d_corr_row); (void)hipFree(t->d_row_stream);
    for (int b = 0; b < 2; b++) if (t->h_ph[b]) (void)hipHostFree(t->h_ph[b]);
    if (t->side) (void)hipStreamDestroy(t->side);
    delete t;
}

// the drift rows of all streams at once (create / reset time and when the rows run out): everything is synchronised
int seeds_set_drift(SeedTables *t, const std::vector<char> &drift)
{
    CSDR_HIP(hipStreamSynchronize(t->side)); CSDR_HIP(hipStreamSynchronize(t->ctx->stream));
    int rows = 0;
    for (int s = 0; s < t->n; s++) rows += drift[s] ? 1 : 0;
    if (rows > t->corr_rows) { const int rc = alloc_corr(t, rows + 8); if (rc) return rc; }
    t->row_stream.assign(t->corr_rows, -1);
    int r = 0;
    for (int s = 0; s < t->n; s++) { t->corr_row[s] = drift[s] ? r : -1; if (drift[s]) t->row_stream[r++] = s; }
    CSDR_HIP(hipMemcpy(t->d_corr_row, t->corr_row.data(), sizeof(int) * t->n, hipMemcpyHostToDevice));
    if (t->corr_rows > 0) CSDR_HIP(hipMemcpy(t->d_row_stream, t->row_stream.data(), sizeof(int) * t->corr_rows, hipMemcpyHostToDevice));
    t->dirty = true;
    return 0;
}

int seeds_reset(SeedTables *t)
{
    CSDR_HIP(hipStreamSynchronize(t->side));
    t->valid[0] = t->valid[1] = false; t->fresh = true; t->dirty = false; t->cur = 0;
    return 0;
}

int seeds_set_rate(SeedTables *t, int stream, float rate, bool drift)
{
    if (stream < 0 || stream >= t->n) return fail_msg(-3, "set_rate: stream %d out of range", stream);
    const int old_row = t->corr_row[stream];
    int row = -1;
    if (drift && t->d_dtab) {
        row = old_row;
        if (row < 0) {
            for (int r = 0; r < t->corr_rows && row < 0; r++) if (t->row_stream[r] < 0) row = r;
            if (row < 0) {                                            // no row left: grow (rare: everything is synchronised and the tables are rebuilt)
                std::vector<char> dr(t->n, 0);
                for (int s = 0; s < t->n; s++) dr[s] = t->corr_row[s] >= 0;
                dr[stream] = 1;
                t->rates[stream] = rate;
                CSDR_HIP(hipStreamSynchronize(t->side)); CSDR_HIP(hipStreamSynchronize(t->ctx->stream));
                CSDR_HIP(hipMemcpy(t->d_rates + stream, &rate, sizeof(float), hipMemcpyHostToDevice));
                // the current table's phases must survive: only the correction buffers are reallocated
                return seeds_set_drift(t, dr);
            }
        }
    }
    if (old_row >= 0 && old_row != row) t->row_stream[old_row] = -1;
    if (row >= 0) t->row_stream[row] = stream;
    t->corr_row[stream] = row; t->rates[stream] = rate;
    // in order behind whatever the side stream is still generating (that table is dropped below), in front of the regeneration
    hipLaunchKernelGGL(k_seed_set, dim3(1), dim3(1), 0, t->side, t->d_rates, t->d_corr_row, t->d_row_stream, stream, rate, row, old_row);
    CSDR_LAUNCH_CHECK();
    t->dirty = true;
    return 0;
}

const float2 *seeds_corr_entry(SeedTables *t, int stream, long long chunk)
{
    if (stream < 0 || stream >= t->n || t->corr_row[stream] < 0 || !t->d_dtab || t->corr_rows <= 0) return nullptr;
    const int c = t->cur;
    if (t->fresh || !t->valid[c] || chunk < t->first[c] || chunk >= t->first[c] + t->cap) return nullptr;
    return t->d_corr[c] + ((size_t)t->corr_row[stream] * t->cap + (size_t)(chunk - t->first[c])) * 32;
}

int seeds_acquire(SeedTables *t, long long first, size_t n, size_t n_next_hint, SeedView *v)
{
    hipStream_t st = t->ctx->stream;
    if ((long long)n + 2 > t->cap) return fail_msg(-3, "seed table: %zu chunks per call exceed the table (%d)", n, t->cap);
    auto covers = [&](int b) { return t->valid[b] && first >= t->first[b] && first + (long long)n <= t->first[b] + t->cap; };
    bool switched = false;
    if (t->fresh) {
        t->valid[0] = t->valid[1] = false; t->cur = 0;
        CSDR_HIP(hipEventRecord(t->ev_free[0], st)); t->free_pending[0] = true;      // (a reset in mid-stream: earlier calls may still be reading)
        CSDR_HIP(hipEventRecord(t->ev_free[1], st)); t->free_pending[1] = true;
        int rc = generate(t, 0, -1, 0, first); if (rc) return rc;
        t->fresh = false; t->dirty = false; switched = true;
    } else if (t->dirty || !covers(t->cur)) {
        const int o = t->cur ^ 1;
        if (!t->dirty && covers(o)) {                                 // the prepared table takes over
            CSDR_HIP(hipEventRecord(t->ev_free[t->cur], st)); t->free_pending[t->cur] = true;
            t->cur = o;
        } else {
            // regenerate from the phases some table holds for chunks first, first + 1 (a retune, or a call the prepared table does not fit): the history chunk and
            // the call's first chunk were reached under the rate that was valid then
            int src = -1;
            for (int b : {t->cur, o}) if (src < 0 && t->valid[b] && first >= t->first[b] && first + 1 < t->first[b] + t->cap) src = b;
            if (src < 0) return fail_msg(-3, "seed table: cannot continue at chunk %lld", first);
            const int dst = src ^ 1;
            CSDR_HIP(hipEventRecord(t->ev_free[dst], st)); t->free_pending[dst] = true;      // whatever was queued so far may still read it
            t->valid[dst] = false;
            int rc = generate(t, dst, src, (long)(first - t->first[src]), first); if (rc) return rc;
            if (dst != t->cur) { CSDR_HIP(hipEventRecord(t->ev_free[t->cur], st)); t->free_pending[t->cur] = true; }
            t->cur = dst;
        }
        t->dirty = false; switched = true;
    }
    const int c = t->cur, o = c ^ 1;
    // (a wait on an event that has fired is not queued at all: each costs ~10 us of bubble on the stream)
    if (switched && hipEventQuery(t->ev_ready[c]) != hipSuccess) CSDR_HIP(hipStreamWaitEvent(st, t->ev_ready[c], 0));
    // prepare the table behind this one on the side stream: it starts at the first call that will not fit, assuming calls that advance by n_next_hint chunks
    if (!(t->valid[o] && t->first[o] > t->first[c]) && n_next_hint > 0) {
        long long f = first;
        while (f + (long long)n <= t->first[c] + t->cap) f += (long long)n_next_hint;
        const long idx = (long)(f - t->first[c]);
        if (idx >= 0 && idx + 1 < t->cap) { t->valid[o] = false; int rc = generate(t, o, c, idx, f); if (rc) return rc; }
    }
    v->pitch = t->pitch;
    v->ctab = t->d_c[c] + (size_t)(first - t->first[c]) * t->pitch;
    v->n_entries = (int)(t->first[c] + t->cap - first);
    v->corr = t->corr_rows > 0 && t->d_dtab ? t->d_corr[c] + (size_t)(first - t->first[c]) * 32 : nullptr;
    v->corr_row = t->d_corr_row; v->corr_chunks = (size_t)t->cap;
    return 0;
}

} // namespace csdr_amd

// Test hook: n chunk advances of the phase bookkeeping at `rate` from phase ph0 through the kernels' plan (host build); out[k] = phase after k + 1 chunks
extern "C" void csdr_amd_debug_phase_chain(float rate, float ph0, int n, float *out)
{
    const float inc = (rate * 2) * csdr_amd::PI_F, step = inc * (float)1024;
    csdr_amd::WrapPlan w; csdr_amd::wrap_plan_init(w, step);
    float p = ph0;
    for (int k = 0; k < n; k++) { p = csdr_amd::wrap_plan_apply(w, p + step); out[k] = p; }
}
