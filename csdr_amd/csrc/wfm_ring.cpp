// wfm_ring.cpp -- the RESIDENT form of the fused WFM chain (csdr_amd_wfm_ring_*): host side of k_wfm_mfma_seq<false, true> (wfm_mfma.hip).
//
// north_star: "a persistent-kernel ring buffer so the stdin->stdout pipe never round-trips to host between stages".  The reference's unit of work is one
// the_bufsize block (16384 samples) per loop iteration of every stage (csdr.c:189-193, 232-247, 330-392; the shift stage re-reads its rate between two
// blocks, csdr.c:881-923).  A launch per such block leaves the chain kernel at a quarter of the memory rate (the launch gap, ~7 us between kernel entry and
// a workgroup's first step, the de-emphasis warm-up of the three time segments a short call must be cut into).  Here ONE grid stays resident:
//
//   input ring  (device)  [n_slots][n_streams][in_pitch]  u8 IQ: block k of every stream lies in slot k mod n_slots; the producer (a copy engine, another kernel,
//                                                          a peer process through HIP IPC) writes it, THEN the host posts the block;
//   descriptors (host)    [n_slots] x tagged 64-byte lines: the block's chunk seeds (the reference's float phase bookkeeping, libcsdr_gpl.c:33-34, 48-51, stays on
//                                                          the host: 16 values per block) -- a workgroup polls its next block's lines, s_sleep between polls;
//   output ring (device)  [n_slots][n_streams][out_pitch]  s16 audio;  done lines (host) [n_slots]: tag, audio samples per stream, device clock at start and end.
//
// A work item is (block k, 16-stream group sb); workgroup w of G takes the items k n_wsb + sb = w (mod G), so consecutive blocks of a stream group run on
// different workgroups at the same time (1024 streams: 64 groups x 4 blocks in flight fill 256 CUs).  No state passes from block to block on the device -- each
// block warms its de-emphasis up over the 48 audio samples in front of it, read from the previous slot (the filter forgets as 0.706^k: 5e-8 after 48) -- so
// the grid can leave at any block boundary and a new launch continues: it leaves when told to (stop word), when nothing arrived for idle_us, or when it is
// older than life_ms.  A stalled host, or a bug in the walk, therefore cannot hold the GPU; a killed host process loses its queues to the driver like any other.
// submit() / wait() relaunch on demand (hipEventQuery on the event behind the launch).
#include "common.hpp"
#include "wfm_mfma.hpp"
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <time.h>
#include <vector>
using namespace csdr_amd;

namespace {
double now_s() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
inline uint32_t ring_tag(long long k) { return (uint32_t)((unsigned long long)k % 0xfffffffeull) + 1u; }      // = res_tag (wfm_mfma.hip); never 0
}

struct csdr_amd_wfm_ring {
    csdr_amd_ctx *ctx;
    int S, D, L, F, T, N, nch, lines, grid, audio_rate;
    float rate, tau, alpha;
    std::vector<float> taps;
    WfmMfmaDevice mfma;
    float *d_taps; float2 *d_dtab_old, *d_lead_seeds; float *d_lead_d, *d_warm, *d_state; int *d_list;
    size_t in_pitch, out_pitch;
    uint8_t *d_in; int16_t *d_out;
    uint32_t *h_block;                 // host-coherent: [desc: N * lines * 16][ctrl: 16][done: N * 16]
    uint32_t *h_desc, *h_ctrl, *h_done;
    unsigned *d_cnt, *d_exiting; unsigned long long *d_tfirst, *d_next, *d_stats; int fence_mode;
    hipStream_t rs; hipEvent_t ev_exit;
    bool launched;
    long long submitted;               // blocks posted so far (the next block's sequence number)
    float phase; float2 hist[4];       // the shifter's phase in front of the next block; seeds of the four chunks before it
    bool pending_lead;                 // the next block is the first after a retune
    long long idle_ticks, life_ticks; int clock_khz;
    long launches;
};

extern "C" {

static int ring_stop(csdr_amd_wfm_ring *r)
{
    if (!r->launched) return 0;
    __atomic_store_n(r->h_ctrl, 1u, __ATOMIC_RELEASE);
    CSDR_HIP(hipStreamSynchronize(r->rs));
    __atomic_store_n(r->h_ctrl, 0u, __ATOMIC_RELEASE);
    r->launched = false;
    return 0;
}

static int ring_ensure_running(csdr_amd_wfm_ring *r)
{
    if (r->launched) {
        const hipError_t q = hipEventQuery(r->ev_exit);
        if (q == hipErrorNotReady) return 0;
        if (q != hipSuccess) return fail(q, "hipEventQuery(resident grid)", __FILE__, __LINE__);
        r->launched = false;
    }
    WfmResident rv; memset((void *)&rv, 0, sizeof rv);
    rv.desc = r->h_desc; rv.ctrl = r->h_ctrl; rv.done = r->h_done; rv.cnt = r->d_cnt; rv.t_first = r->d_tfirst; rv.next_item = r->d_next; rv.exiting = r->d_exiting;
    rv.in_ring = r->d_in; rv.out_ring = r->d_out; rv.in_slot_bytes = (size_t)r->S * r->in_pitch; rv.out_slot_elems = (size_t)r->S * r->out_pitch;
    rv.n_slots = r->N; rv.desc_lines = r->lines; rv.T = r->T; rv.D = r->D; rv.L = r->L; rv.F = r->F;
    rv.idle_ticks = r->idle_ticks; rv.life_ticks = r->life_ticks; rv.lead_d = r->d_lead_d; rv.lead_stride = wfm_lead_max(r->D, r->L, r->F); rv.lead_state = r->d_state; rv.stats = r->d_stats; rv.fence_mode = r->fence_mode;
    CSDR_HIP(hipMemsetAsync(r->d_exiting, 0, sizeof(unsigned), r->rs));
    const int rc = wfm_mfma_launch_resident(r->rs, r->ev_exit, r->mfma, r->S, r->in_pitch, r->alpha, r->out_pitch, rv, r->grid);
    if (rc) return rc;
    r->launched = true; r->launches++;
    return 0;
}

static bool ring_done(const csdr_amd_wfm_ring *r, long long k)
{
    if (k < 0) return true;
    if (k >= r->submitted) return false;
    if (k + r->N < r->submitted) return true;                                  // its slot has been reused since: it was waited for then
    return __atomic_load_n(r->h_done + (size_t)(k % r->N) * 16, __ATOMIC_ACQUIRE) == ring_tag(k);
}

static int ring_wait_block(csdr_amd_wfm_ring *r, long long k, double timeout_s)
{
    if (ring_done(r, k)) return 0;
    const double t_end = now_s() + (timeout_s > 0 ? timeout_s : 10.0);
    for (unsigned spin = 0;; spin++) {
        if (ring_done(r, k)) return 0;
        if ((spin & 63) == 0) {                                                // the grid may have left (idle, age) with this block still posted
            const int rc = ring_ensure_running(r); if (rc) return rc;
            if (now_s() > t_end) return fail_msg(-4, "wfm ring: block %lld not finished after %.1f s (resident grid %s)", k, timeout_s > 0 ? timeout_s : 10.0, r->launched ? "alive" : "gone");
        }
        __builtin_ia32_pause();
    }
}

static int ring_upload_tables(csdr_amd_wfm_ring *r, float rate)
{
    WfmMfmaTable t;
    wfm_mfma_build_table(r->D, r->L, r->F, rate, r->taps.data(), t);
    r->mfma.tile_stride_bytes = t.tile_stride_bytes; r->mfma.win_off_bytes = t.win_off_bytes; r->mfma.seq_scale = t.seq_scale;
    if (!r->mfma.d_seq_frags) {
        CSDR_HIP(hipMalloc(&r->mfma.d_seq_frags, t.seq_frags.size()));
        CSDR_HIP(hipMalloc((void **)&r->mfma.d_seq_cum, t.seq_cum.size() * sizeof(float)));
        CSDR_HIP(hipMalloc((void **)&r->mfma.d_dtab, t.dtab.size() * sizeof(float2)));
        CSDR_HIP(hipMalloc((void **)&r->d_dtab_old, t.dtab.size() * sizeof(float2)));
    }
    CSDR_HIP(hipMemcpy(r->mfma.d_seq_frags, t.seq_frags.data(), t.seq_frags.size(), hipMemcpyHostToDevice));
    CSDR_HIP(hipMemcpy(r->mfma.d_seq_cum, t.seq_cum.data(), t.seq_cum.size() * sizeof(float), hipMemcpyHostToDevice));
    CSDR_HIP(hipMemcpy(r->mfma.d_dtab, t.dtab.data(), t.dtab.size() * sizeof(float2), hipMemcpyHostToDevice));
    return 0;
}

void csdr_amd_wfm_ring_destroy(csdr_amd_wfm_ring *r)
{
    if (!r) return;
    (void)hipSetDevice(r->ctx->device);
    if (r->rs) { (void)ring_stop(r); (void)hipStreamSynchronize(r->rs); }
    (void)hipFree(r->mfma.d_seq_frags); (void)hipFree(r->mfma.d_seq_cum); (void)hipFree(r->mfma.d_dtab); (void)hipFree(r->d_dtab_old);
    (void)hipFree(r->d_taps); (void)hipFree(r->d_lead_seeds); (void)hipFree(r->d_lead_d); (void)hipFree(r->d_warm); (void)hipFree(r->d_state); (void)hipFree(r->d_list);
    (void)hipFree(r->d_in); (void)hipFree(r->d_out); (void)hipFree(r->d_cnt); (void)hipFree(r->d_exiting); (void)hipFree(r->d_tfirst); (void)hipFree(r->d_next); (void)hipFree(r->d_stats);
    if (r->h_block) (void)hipHostFree(r->h_block);
    if (r->ev_exit) (void)hipEventDestroy(r->ev_exit);
    if (r->rs) (void)hipStreamDestroy(r->rs);
    delete r;
}

int csdr_amd_wfm_ring_reset(csdr_amd_wfm_ring *r)
{
    int rc = ring_stop(r); if (rc) return rc;
    CSDR_HIP(hipStreamSynchronize(r->rs));
    memset(r->h_block, 0, sizeof(uint32_t) * ((size_t)r->N * r->lines * 16 + 16 + (size_t)r->N * 16));
    CSDR_HIP(hipMemset(r->d_cnt, 0, sizeof(unsigned) * r->N));
    CSDR_HIP(hipMemset(r->d_tfirst, 0xff, sizeof(unsigned long long) * r->N));
    CSDR_HIP(hipMemset(r->d_stats, 0, sizeof(unsigned long long) * 4 * r->grid));
    std::vector<unsigned long long> ni(r->grid);
    for (int w = 0; w < r->grid; w++) ni[w] = (unsigned long long)w;
    CSDR_HIP(hipMemcpy(r->d_next, ni.data(), sizeof(unsigned long long) * r->grid, hipMemcpyHostToDevice));
    r->submitted = 0; r->phase = 0.f; r->pending_lead = false;
    for (int i = 0; i < 4; i++) r->hist[i] = make_float2(1.f, 0.f);
    return 0;
}

csdr_amd_wfm_ring *csdr_amd_wfm_ring_create(csdr_amd_ctx *ctx, int n_streams, float shift_rate, int decimation, const float *host_taps, int taps_length, int frac_rate,
                                            float tau, int audio_rate, size_t block_samples, int n_slots)
{
    if (n_streams <= 0 || decimation <= 0 || taps_length <= 0 || frac_rate <= 1) { fail_msg(-3, "wfm ring: bad parameters"); return nullptr; }
    if (!wfm_mfma_supported(decimation, taps_length, frac_rate)) { fail_msg(-3, "wfm ring: decimation %d / %d taps / audio decimation %d is outside the matrix-core chain kernel's shapes", decimation, taps_length, frac_rate); return nullptr; }
    if (block_samples % 1024 || block_samples < 4096 || block_samples > 65536) { fail_msg(-3, "wfm ring: blocks are whole 1024-sample chunks, 4096 .. 65536 samples (got %zu)", block_samples); return nullptr; }
    if (n_slots < 3 || n_slots > 64) { fail_msg(-3, "wfm ring: 3 .. 64 slots (got %d)", n_slots); return nullptr; }
    // the two warm-up steps (12 tiles) and the first window in front of a block must lie inside the previous block
    if ((size_t)(12 * 4 * decimation * frac_rate + 2 * WFM_HIST + 1024) > block_samples) { fail_msg(-3, "wfm ring: block of %zu samples shorter than the warm-up reach", block_samples); return nullptr; }
    (void)hipSetDevice(ctx->device);
    csdr_amd_wfm_ring *r = new csdr_amd_wfm_ring();
    memset((void *)&r->mfma, 0, sizeof r->mfma);
    r->ctx = ctx; r->S = n_streams; r->D = decimation; r->L = taps_length; r->F = frac_rate; r->T = (int)block_samples; r->N = n_slots; r->audio_rate = audio_rate;
    r->rate = shift_rate; r->tau = tau;
    const float dt = (float)(1.0 / audio_rate); r->alpha = dt / (tau + dt);                  // libcsdr.c:1090-1091
    r->taps.assign(host_taps, host_taps + taps_length);
    r->nch = r->T / 1024; r->lines = (r->nch + 6 + 6) / 7;
    r->in_pitch = ((size_t)2 * r->T + 127) & ~(size_t)127;
    r->out_pitch = ((size_t)r->T / ((size_t)decimation * frac_rate) + 2 + 63) & ~(size_t)63;
    const int n_wsb = (n_streams + 15) / 16;
    long long want = (long long)n_wsb * (n_slots - 2);                                       // items that can be in flight
    r->grid = (int)(want < wfm_resident_max_grid() ? want : wfm_resident_max_grid());
    { const char *g = getenv("CSDR_AMD_RING_GRID"); if (g && atoi(g) > 0 && atoi(g) < r->grid) r->grid = atoi(g); }
    r->d_taps = nullptr; r->d_dtab_old = nullptr; r->d_lead_seeds = nullptr; r->d_lead_d = nullptr; r->d_warm = nullptr; r->d_state = nullptr; r->d_list = nullptr; r->d_in = nullptr; r->d_out = nullptr;
    r->h_block = nullptr; r->d_cnt = nullptr; r->d_exiting = nullptr; r->d_tfirst = nullptr; r->d_next = nullptr; r->d_stats = nullptr; r->fence_mode = getenv("CSDR_AMD_RING_FENCE") ? atoi(getenv("CSDR_AMD_RING_FENCE")) : 0; r->rs = nullptr; r->ev_exit = nullptr;
    r->launched = false; r->launches = 0; r->submitted = 0; r->pending_lead = false;
    int khz = 100000; if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device) != hipSuccess || khz <= 0) khz = 100000;
    r->clock_khz = khz;
    r->idle_ticks = (long long)khz * 200 / 1000;                                             // 200 us
    r->life_ticks = (long long)khz * 250;                                                    // 250 ms
    hipError_t e = hipSuccess;
    auto alloc = [&](void **p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
    alloc((void **)&r->d_taps, sizeof(float) * taps_length);
    alloc((void **)&r->d_lead_seeds, sizeof(float2) * 8);
    alloc((void **)&r->d_lead_d, sizeof(float) * (size_t)wfm_lead_max(decimation, taps_length, frac_rate) * n_streams);
    alloc((void **)&r->d_warm, sizeof(float) * WFM_RES_WARM * n_streams);
    alloc((void **)&r->d_state, sizeof(float) * n_streams);
    alloc((void **)&r->d_list, sizeof(int) * n_streams);
    alloc((void **)&r->d_in, (size_t)n_slots * n_streams * r->in_pitch);
    alloc((void **)&r->d_out, sizeof(int16_t) * (size_t)n_slots * n_streams * r->out_pitch);
    alloc((void **)&r->d_cnt, sizeof(unsigned) * n_slots);
    alloc((void **)&r->d_exiting, sizeof(unsigned));
    alloc((void **)&r->d_tfirst, sizeof(unsigned long long) * n_slots);
    alloc((void **)&r->d_next, sizeof(unsigned long long) * r->grid);
    alloc((void **)&r->d_stats, sizeof(unsigned long long) * 4 * r->grid);
    const size_t hwords = (size_t)n_slots * r->lines * 16 + 16 + (size_t)n_slots * 16;
    if (e == hipSuccess) e = hipHostMalloc((void **)&r->h_block, sizeof(uint32_t) * hwords, hipHostMallocCoherent | hipHostMallocMapped);
    // The grid's stream gets a PRIORITY of its own: HIP multiplexes a process's streams onto a few hardware queues (four by default), and whatever shares a queue with the
    // resident grid waits behind it until it leaves -- a block's input copy then takes idle_us instead of microseconds (seen in the test suite once earlier tests had used up
    // the queues: 35 launches for 40 blocks).  Queues are per priority level, so a high-priority stream never shares one with the default-priority streams of the caller.
    if (e == hipSuccess) {
        int lo = 0, hi = 0;
        const char *pe = getenv("CSDR_AMD_RING_PRIO");                      // (A/B: 0 = a default-priority stream)
        if (!(pe && atoi(pe) == 0) && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo) e = hipStreamCreateWithPriority(&r->rs, hipStreamNonBlocking, hi);
        else e = hipStreamCreateWithFlags(&r->rs, hipStreamNonBlocking);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->ev_exit, hipEventDisableTiming);
    if (e != hipSuccess) { fail(e, "wfm ring: allocation", __FILE__, __LINE__); csdr_amd_wfm_ring_destroy(r); return nullptr; }
    r->h_desc = r->h_block; r->h_ctrl = r->h_desc + (size_t)n_slots * r->lines * 16; r->h_done = r->h_ctrl + 16;
    std::vector<int> list(n_streams); for (int s = 0; s < n_streams; s++) list[s] = s;
    if (hipMemcpy(r->d_taps, host_taps, sizeof(float) * taps_length, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(r->d_list, list.data(), sizeof(int) * n_streams, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(r->d_in, 0x80, (size_t)n_slots * n_streams * r->in_pitch) != hipSuccess ||
        ring_upload_tables(r, shift_rate) || csdr_amd_wfm_ring_reset(r)) { csdr_amd_wfm_ring_destroy(r); return nullptr; }
    return r;
}

int csdr_amd_wfm_ring_slots(const csdr_amd_wfm_ring *r) { return r->N; }
int csdr_amd_wfm_ring_grid(const csdr_amd_wfm_ring *r) { return r->grid; }
long csdr_amd_wfm_ring_launches(const csdr_amd_wfm_ring *r) { return r->launches; }
long long csdr_amd_wfm_ring_submitted(const csdr_amd_wfm_ring *r) { return r->submitted; }

int csdr_amd_wfm_ring_set_timeouts(csdr_amd_wfm_ring *r, double idle_us, double life_ms)
{
    if (idle_us < 1 || life_ms < 0.01 || life_ms > 60000) return fail_msg(-3, "wfm ring: idle time >= 1 us, life 0.01 .. 60000 ms");
    int rc = ring_stop(r); if (rc) return rc;
    r->idle_ticks = (long long)(r->clock_khz * idle_us / 1000.0); r->life_ticks = (long long)(r->clock_khz * life_ms);
    if (r->idle_ticks < 1) r->idle_ticks = 1;
    return 0;
}

int csdr_amd_wfm_ring_resident(csdr_amd_wfm_ring *r)
{
    if (!r->launched) return 0;
    if (hipEventQuery(r->ev_exit) == hipErrorNotReady) return 1;
    r->launched = false;
    return 0;
}

int csdr_amd_wfm_ring_stop(csdr_amd_wfm_ring *r) { return ring_stop(r); }

int csdr_amd_wfm_ring_acquire(csdr_amd_wfm_ring *r, long long seq, double timeout_s)
{
    if (seq < r->submitted) return fail_msg(-3, "wfm ring: block %lld has been posted already", seq);
    if (seq > r->submitted + r->N - 2) return fail_msg(-3, "wfm ring: block %lld is more than %d blocks ahead of the last posted one", seq, r->N - 2);
    // slot seq mod N held block seq - N (its output may still be wanted by the caller -- that is the caller's business -- and its input is the history of block seq - N + 1)
    int rc = ring_wait_block(r, seq - r->N, timeout_s); if (rc) return rc;
    return ring_wait_block(r, seq - r->N + 1, timeout_s);
}

uint8_t *csdr_amd_wfm_ring_input(csdr_amd_wfm_ring *r, long long seq, size_t *pitch)
{
    if (pitch) *pitch = r->in_pitch;
    return r->d_in + (size_t)(seq % r->N) * r->S * r->in_pitch;
}

const int16_t *csdr_amd_wfm_ring_output(csdr_amd_wfm_ring *r, long long seq, size_t *pitch)
{
    if (pitch) *pitch = r->out_pitch;
    return r->d_out + (size_t)(seq % r->N) * r->S * r->out_pitch;
}

// audio samples of block k (host mirror of the kernel's arithmetic and of csdr_amd_wfm_process)
static void ring_geometry(const csdr_amd_wfm_ring *r, long long k, long long *j_first, int *n_audio)
{
    auto j_hi = [&](long long kk) -> long long {
        if (kk < 0) return -1;
        const long long avail_last = (kk + 1) * r->T - 1;
        if (avail_last - (r->L - 1) < 0) return -1;
        const long long k_max = (avail_last - (r->L - 1)) / r->D;
        return k_max >= 10 ? (k_max - 10) / r->F : -1;
    };
    const long long jp = j_hi(k - 1), jn = j_hi(k);
    *j_first = jp + 1; *n_audio = (int)(jn - jp);
}

long long csdr_amd_wfm_ring_submit(csdr_amd_wfm_ring *r)
{
    const long long k = r->submitted;
    int rc = csdr_amd_wfm_ring_acquire(r, k, 0); if (rc) return rc;
    const int slot = (int)(k % r->N);
    // chunk seeds C_m = (cos, sin)(starting_phase_m): the reference's float phase bookkeeping (libcsdr_gpl.c:33-34, 48-51) in 1024-sample chunks (csdr.c:911-918),
    // for chunks first - 4 (the warm-up's reach) .. first + nch + 1
    const int ns = r->nch + 6;
    std::vector<float2> seeds(ns);
    for (int i = 0; i < 4; i++) seeds[i] = r->hist[i];
    const float inc = (r->rate * 2) * PI_F;
    float ph = r->phase, ph_next = r->phase;
    for (int m = 0; m < r->nch + 2; m++) {
        seeds[4 + m] = make_float2((float)cos((double)ph), (float)sin((double)ph));
        float nx = ph + inc * (float)1024;
        while (nx > PI_F) nx -= 2 * PI_F;
        while (nx < -PI_F) nx += 2 * PI_F;
        ph = nx;
        if (m + 1 == r->nch) ph_next = ph;
    }
    int n_lead = 0; uint32_t retuned = 0;
    if (r->pending_lead) {
        // the first block behind a retune (csdr.c:881-923: the new rate from this block's first sample, the phase carried): its first audio samples have windows that
        // reach into samples rotated at the old rate -- evaluated with both tables in front of the grid's (re)launch
        long long j_first; int n_audio; ring_geometry(r, k, &j_first, &n_audio);
        const long long lim = k * r->T - 1 - 9LL * r->D;
        long nl = 0;
        if (lim >= 0) nl = (long)(lim / ((long long)r->D * r->F) - j_first + 1);
        if (nl < 0) nl = 0;
        if (nl > wfm_lead_max(r->D, r->L, r->F)) nl = wfm_lead_max(r->D, r->L, r->F);
        if (nl > n_audio) nl = n_audio;
        n_lead = (int)nl; retuned = 1;
        {
            rc = ring_stop(r); if (rc) return rc;
            CSDR_HIP(hipMemcpy(r->d_lead_seeds, seeds.data(), sizeof(float2) * 8, hipMemcpyHostToDevice));          // [0] = chunk first - 4
            const uint8_t *in = r->d_in + (size_t)slot * r->S * r->in_pitch, *prev = r->d_in + (size_t)((slot + r->N - 1) % r->N) * r->S * r->in_pitch;
            rc = wfm_mfma_lead_shared(r->rs, in, r->in_pitch, prev, (size_t)2 * r->T, r->d_taps, r->d_lead_seeds, r->mfma.d_dtab, r->d_dtab_old, r->d_list, r->S, r->d_lead_d, wfm_lead_max(r->D, r->L, r->F),
                                      r->d_warm, r->d_state, r->alpha, r->D, r->L, r->F, k * r->T, j_first, n_lead);
            if (rc) return rc;
            CSDR_HIP(hipStreamSynchronize(r->rs));
        }
        r->pending_lead = false;
    }
    uint32_t *dl = r->h_desc + (size_t)slot * r->lines * 16;
    for (int ln = 0; ln < r->lines; ln++) {
        uint32_t *w = dl + ln * 16;
        w[1] = ln == 0 ? ((uint32_t)n_lead | (retuned << 8)) : 0u;
        for (int i = 0; i < 7; i++) {
            const int si = ln * 7 + i;
            const float2 v = si < ns ? seeds[si] : make_float2(1.f, 0.f);
            memcpy(w + 2 + 2 * i, &v.x, 4); memcpy(w + 3 + 2 * i, &v.y, 4);
        }
    }
    for (int ln = 0; ln < r->lines; ln++) __atomic_store_n(dl + ln * 16, ring_tag(k), __ATOMIC_RELEASE);      // the tag last: a line that shows it is complete (x86 stores are seen in order)
    // the state in front of the next block
    for (int i = 0; i < 4; i++) r->hist[i] = seeds[r->nch + i];
    r->phase = ph_next;
    r->submitted = k + 1;
    rc = ring_ensure_running(r); if (rc) return rc;
    return k;
}

long csdr_amd_wfm_ring_wait(csdr_amd_wfm_ring *r, long long seq, double timeout_s)
{
    if (seq < 0 || seq >= r->submitted) return fail_msg(-3, "wfm ring: block %lld has not been posted", seq);
    if (seq + r->N < r->submitted) return fail_msg(-3, "wfm ring: block %lld's slot has been reused", seq);
    const int rc = ring_wait_block(r, seq, timeout_s); if (rc) return rc;
    return (long)r->h_done[(size_t)(seq % r->N) * 16 + 1];
}

int csdr_amd_wfm_ring_block_times(csdr_amd_wfm_ring *r, long long seq, double *t_first_us, double *t_done_us)
{
    if (!ring_done(r, seq) || seq + r->N < r->submitted) return fail_msg(-3, "wfm ring: block %lld is not finished or its slot has been reused", seq);
    const uint32_t *dn = r->h_done + (size_t)(seq % r->N) * 16;
    const unsigned long long tf = (unsigned long long)dn[2] | ((unsigned long long)dn[3] << 32), td = (unsigned long long)dn[4] | ((unsigned long long)dn[5] << 32);
    if (t_first_us) *t_first_us = (double)tf * 1000.0 / r->clock_khz;
    if (t_done_us) *t_done_us = (double)td * 1000.0 / r->clock_khz;
    return 0;
}

// Benchmark / soak aid: posts n_blocks blocks whose inputs are what lies in the ring's slots (resident synthetic data: consecutive blocks of a stream as far as the
// arithmetic is concerned), as fast as the ring takes them, and waits for the last one.  t_first_us / t_done_us (may be null; n_blocks doubles each) receive every
// block's start and completion on the device clock.
int csdr_amd_wfm_ring_replay(csdr_amd_wfm_ring *r, long n_blocks, double *t_first_us, double *t_done_us)
{
    if (n_blocks <= 0) return 0;
    const long long k0 = r->submitted;
    long long rec = k0;
    auto record = [&](long long b) { return csdr_amd_wfm_ring_block_times(r, b, t_first_us ? t_first_us + (b - k0) : nullptr, t_done_us ? t_done_us + (b - k0) : nullptr); };
    for (long i = 0; i < n_blocks; i++) {
        const long long k = csdr_amd_wfm_ring_submit(r);
        if (k < 0) return (int)k;
        while (rec <= k - r->N + 1) { if (t_first_us || t_done_us) { const int rc = record(rec); if (rc) return rc; } rec++; }      // finished for certain, their done lines still intact
    }
    for (; rec < k0 + n_blocks; rec++) {
        int rc = ring_wait_block(r, rec, 0); if (rc) return rc;
        if (t_first_us || t_done_us) { rc = record(rec); if (rc) return rc; }
    }
    return 0;
}

// `csdr shift_addition_cc --fifo` (csdr.c:881-923): the new rate from the NEXT posted block's first sample on, the phase carried.  Drains the ring, stops the grid,
// rebuilds the weight set (a few milliseconds on a host core), and the next submit evaluates the straddling audio samples with both tables.
int csdr_amd_wfm_ring_set_rate(csdr_amd_wfm_ring *r, float shift_rate)
{
    if (shift_rate == r->rate) return 0;
    int rc;
    for (long long k = r->submitted - r->N; k < r->submitted; k++) if (k >= 0) { rc = ring_wait_block(r, k, 0); if (rc) return rc; }
    rc = ring_stop(r); if (rc) return rc;
    if (!r->pending_lead) CSDR_HIP(hipMemcpy(r->d_dtab_old, r->mfma.d_dtab, sizeof(float2) * 3072, hipMemcpyDeviceToDevice));      // (two retunes without a block between them: the samples in front still carry the first rate)
    rc = ring_upload_tables(r, shift_rate); if (rc) return rc;
    r->rate = shift_rate;
    r->pending_lead = r->submitted > 0;
    return 0;
}

float csdr_amd_wfm_ring_get_rate(const csdr_amd_wfm_ring *r) { return r->rate; }

// Where the grid's time went since the last reset (stops the grid to read its counters): microseconds per work item spent waiting for a block, in the chain's body,
// in the completion (write-back, counting in), averaged over all workgroups; out[3] = items.
int csdr_amd_wfm_ring_stats(csdr_amd_wfm_ring *r, double out[4])
{
    int rc = ring_stop(r); if (rc) return rc;
    std::vector<unsigned long long> st((size_t)4 * r->grid);
    CSDR_HIP(hipMemcpy(st.data(), r->d_stats, sizeof(unsigned long long) * st.size(), hipMemcpyDeviceToHost));
    unsigned long long sum[4] = {0, 0, 0, 0};
    for (int w = 0; w < r->grid; w++) for (int i = 0; i < 4; i++) sum[i] += st[(size_t)4 * w + i];
    for (int i = 0; i < 3; i++) out[i] = sum[3] ? (double)sum[i] * 1000.0 / r->clock_khz / (double)sum[3] : 0.0;
    out[3] = (double)sum[3];
    return 0;
}

} // extern "C"
