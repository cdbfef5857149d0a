// csdr_cli.cpp -- `csdr <function> <args>` for the hot-path commands, served by libcsdr_amd.so (SURVEY.md section 8b "CLI", f1).
//
// Same argv grammar, same raw native-endian sample streams on stdin/stdout as the reference CLI (csdr.c:56-181 usage string;
// per-command loops cited below), so a shell pipeline keeps working when `csdr` is replaced by this binary.  What differs, on purpose:
//   * each process moves whatever has arrived -- at least the reference's the_bufsize (1024 / 16384 samples, csdr.c:189-193, 332), at most
//     CSDR_AMD_BLOCK elements (default 4194304; 65536 when a control channel is open) -- through the GPU per iteration: a live stream sees the
//     reference's latency, a file or a fast producer large blocks (CSDR_AMD_MIN_READ overrides the minimum); the sample VALUES follow the reference's block semantics
//     exactly where they are observable (shift_* re-seed every 1024 samples like csdr.c:785,836,911-918; fastagc_ff works on its own
//     block size; decimating_shift_addition_cc restarts its recurrence every the_bufsize samples) and the stream models verified against
//     the reference (fir_decimate_cc refeed, fractional_decimator_ff refeed, overlap-add, fastddc);
//   * EOF is clean: every complete input sample is processed once; the reference's stale extra block at EOF is not emitted.
// Wire protocol either side of every command (csdr.c:325-419): CSDR_FIXED_BUFSIZE and CSDR_PRINT_BUFSIZES are honoured,
// CSDR_DYNAMIC_BUFSIZE_ON=1 makes every command consume the 8-byte "csdr"+int preamble from stdin and send its own (with the per-command
// size rule of the reference: /decimation, /rate, fft_size, ...) before its data; `setbuf N` starts such a chain.
// Live retune (csdr.c:252-323): `--fifo <path>` / `--fd <n>` in place of the rate arguments of shift_addition_cc, bandpass_fir_fft_cc and
// fastddc_inv_cc; the newest complete line is applied between two blocks.
// Fusion (the part of f1 that a process-per-command shell pipeline cannot give): `csdr chain "<cmd> <args> | <cmd> <args> | ..."` runs
// the listed hot-path commands in ONE process with every intermediate stream resident in HBM (PCIe carries only the first input and the
// last output), and replaces the README.md:66 WFM pattern by the fused matrix-core kernel.
// Device hand-off between ADJACENT csdr processes of an unchanged shell pipeline (`csdr a | csdr b`): see "device hand-off" below -- the samples stay in
// HBM, the pipe between the two processes carries nothing but the preamble.  Negotiated out of band, so a peer that is not this binary sees plain bytes.
// There is no CPU fallback: without a gfx950 device the process exits with status 3 and the reason on stderr.
#include "../../include/csdr_amd.h"
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <errno.h>
#include <fcntl.h>
#include <signal.h>
#include <stdarg.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <poll.h>
#include <pthread.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <time.h>
#include <string>
#include <vector>
#include <atomic>

namespace {

const char *g_cmd = "csdr";
int badsyntax(const char *why) { fprintf(stderr, "csdr %s: %s\n", g_cmd, why); return -1; }              // csdr.c:209-218
[[noreturn]] void die(const char *what) { fprintf(stderr, "csdr %s: %s: %s\n", g_cmd, what, csdr_amd_last_error()); exit(3); }
#define MUST(x) do { long rc__ = (long)(x); if (rc__ < 0) die(#x); } while (0)

size_t block_elems()
{
    const char *e = getenv("CSDR_AMD_BLOCK");
    long v = e ? atol(e) : 4194304;
    if (v < 4096) v = 4096;
    return (size_t)(v / 1024 * 1024);
}

int window_from(const char *s)
{   // libcsdr.c:57-63
    if (!strcmp(s, "BOXCAR")) return CSDR_WINDOW_BOXCAR;
    if (!strcmp(s, "BLACKMAN")) return CSDR_WINDOW_BLACKMAN;
    return CSDR_WINDOW_HAMMING;
}

// One streaming operator: consumes in_elem-byte elements, produces out_elem-byte elements.
struct Stage {
    size_t in_elem = 4, out_elem = 4;
    size_t min_block = 0;          // run() uses blocks of at least 4x this many elements (operators with a long history)
    size_t granule = 1;            // process() is only called with n_in a multiple of this (except at EOF when flush_partial)
    bool flush_partial = true;     // at EOF, a final n_in % granule != 0 call is allowed
    virtual ~Stage() {}
    // returns elements written; *consumed = input elements that need not be presented again
    virtual long process(csdr_amd_ctx *c, const void *d_in, size_t n_in, void *d_out, size_t out_cap, size_t *consumed) = 0;
    virtual size_t out_capacity(size_t n_in) { return n_in + 16; }
    virtual int next_bufsize(int b) { return b; }                   // what the reference passes to sendbufsize() for this command
    virtual const char *ctl_format() { return nullptr; }            // scanf format of a control line, if the command has a control channel
    virtual void retune(csdr_amd_ctx *, float, float) {}
    struct Control *ctl = nullptr;                                  // its open control channel (--fifo / --fd), polled in front of every pass (also inside `chain`)
};

struct Convert : Stage {
    int kind; int bigendian = 0;
    Convert(int k, size_t ie, size_t oe) : kind(k) { in_elem = ie; out_elem = oe; }
    long process(csdr_amd_ctx *c, const void *i, size_t n, void *o, size_t, size_t *cons) override
    {
        *cons = n;
        switch (kind) {
            case 0: MUST(csdr_amd_convert_u8_f(c, (const uint8_t *)i, (float *)o, n)); break;       // csdr.c:534-545
            case 1: MUST(csdr_amd_convert_f_u8(c, (const float *)i, (uint8_t *)o, n)); break;       // :546-557
            case 2: MUST(csdr_amd_convert_s8_f(c, (const int8_t *)i, (float *)o, n)); break;
            case 3: MUST(csdr_amd_convert_f_s8(c, (const float *)i, (int8_t *)o, n)); break;
            case 4: MUST(csdr_amd_convert_f_s16(c, (const float *)i, (int16_t *)o, n)); break;      // :582-593
            case 5: MUST(csdr_amd_convert_s16_f(c, (const int16_t *)i, (float *)o, n)); break;      // :594-605
            case 6: MUST(csdr_amd_convert_f_s24(c, (const float *)i, (uint8_t *)o, n, bigendian)); break;   // :606-619
            case 7: MUST(csdr_amd_convert_s24_f(c, (const uint8_t *)i, (float *)o, n, bigendian)); break;   // :620-633
        }
        return (long)n;
    }
};

struct Shift : Stage {   // csdr.c:703-925
    int variant; float rate; float phase = 0; int aux; bool real_in = false; csdr_complexf *rot = nullptr; size_t rot_cap = 0;
    Shift(int v, float r, int a) : variant(v), rate(r), aux(a) { in_elem = 8; out_elem = 8; granule = 1024; }
    const char *ctl_format() override { return (variant == CSDR_SHIFT_ADDITION || variant == CSDR_SHIFT_ADDFAST || variant == CSDR_SHIFT_UNROLL) ? "%g\n" : nullptr; }   // csdr.c:757-792, 808-843, 881-923, 3373-3407
    void retune(csdr_amd_ctx *, float r, float) override { rate = r; fprintf(stderr, "csdr %s: reinitialized to %g\n", g_cmd, r); }   // phase carries on (csdr.c:896-921)
    long process(csdr_amd_ctx *c, const void *i, size_t n, void *o, size_t, size_t *cons) override
    {
        *cons = n;
        if (real_in) {   // shift_addition_fc csdr.c:927-980
            if (!rot) { rot_cap = n + 8192; rot = (csdr_complexf *)csdr_amd_malloc(c, 8 * rot_cap); }
            else if (n + 16 > rot_cap) { csdr_amd_free(c, rot); rot_cap = n + 8192; rot = (csdr_complexf *)csdr_amd_malloc(c, 8 * rot_cap); }
            MUST(csdr_amd_rotator_generate(c, CSDR_SHIFT_ADDITION, rate, &phase, rot, n, 1024, 0));
            MUST(csdr_amd_mix_fc(c, (const float *)i, (csdr_complexf *)o, rot, 1, n, n, n));
        } else MUST(csdr_amd_shift_cc(c, variant, rate, &phase, (const csdr_complexf *)i, (csdr_complexf *)o, 1, n, n, n, 1024, aux));
        return (long)n;
    }
};

struct FirDecimate : Stage {   // csdr.c:1114-1177
    int D, ntaps; float *d_taps;
    FirDecimate(csdr_amd_ctx *c, int factor, float tbw, int window) : D(factor)
    {
        in_elem = 8; out_elem = 8;
        ntaps = csdr_amd_firdes_filter_len(tbw); min_block = (size_t)ntaps + factor;
        fprintf(stderr, "fir_decimate_cc: taps_length = %d\n", ntaps);
        std::vector<float> t(ntaps);
        csdr_amd_firdes_lowpass_f(t.data(), ntaps, 0.5f / (float)factor, window);
        d_taps = (float *)csdr_amd_malloc(c, 4 * ntaps);
        MUST(csdr_amd_h2d(c, d_taps, t.data(), 4 * ntaps));
    }
    size_t out_capacity(size_t n) override { return n / D + 16; }
    int next_bufsize(int b) override { return b / D; }               // csdr.c:1140
    long process(csdr_amd_ctx *c, const void *i, size_t n, void *o, size_t cap, size_t *cons) override
    {
        long no = csdr_amd_fir_decimate_cc(c, (const csdr_complexf *)i, (csdr_complexf *)o, 1, (int)n, n, cap, D, d_taps, ntaps);
        MUST(no);
        *cons = (size_t)no * D;                                    // the rest is re-presented (csdr.c:1172-1174)
        return no;
    }
};

struct Fmdemod : Stage {   // csdr.c:984-1012
    csdr_complexf *d_last;
    Fmdemod(csdr_amd_ctx *c) { in_elem = 8; out_elem = 4; d_last = (csdr_complexf *)csdr_amd_malloc(c, 8); MUST(csdr_amd_memset(c, d_last, 0, 8)); }
    long process(csdr_amd_ctx *c, const void *i, size_t n, void *o, size_t, size_t *cons) override
    { *cons = n; MUST(csdr_amd_fmdemod_quadri_cf(c, (const csdr_complexf *)i, (float *)o, 1, n, n, n, d_last)); return (long)n; }
};

struct Limit : Stage {   // csdr.c:673-686
    float m; Limit(float mm) : m(mm) { granule = 4; }
    long process(csdr_amd_ctx *c, const void *i, size_t n, void *o, size_t, size_t *cons) override
    { *cons = n; MUST(csdr_amd_limit_ff(c, (const float *)i, (float *)o, n, m)); return (long)n; }
};

struct DeemphWfm : Stage {   // csdr.c:1014-1032
    float tau; int rate; float *d_last;
    DeemphWfm(csdr_amd_ctx *c, int r, float t) : tau(t), rate(r) { d_last = (float *)csdr_amd_malloc(c, 4); MUST(csdr_amd_memset(c, d_last, 0, 4)); }
    long process(csdr_amd_ctx *c, const void *i, size_t n, void *o, size_t, size_t *cons) override
    { *cons = n; MUST(csdr_amd_deemphasis_wfm_ff(c, (const float *)i, (float *)o, 1, n, n, n, tau, rate, d_last)); return (long)n; }
};

struct DeemphNfm : Stage {   // csdr.c:1068-1087
    // The reference's loop runs its FIR over the freshly allocated input buffer BEFORE it reads anything (`processed` starts at 0, so the
    // first fread is empty, csdr.c:1076-1081): its output is the FIR of  the_bufsize zeros ++ stream.  `pre` = zeros not yet consumed.
    int ntaps; float *d_taps; size_t pre; float *d_tmp; size_t tmp_cap;
    DeemphNfm(csdr_amd_ctx *c, int rate, int the_bufsize) : pre((size_t)the_bufsize), d_tmp(nullptr), tmp_cap(0)
    {
        const float *t = nullptr; ntaps = csdr_amd_nfm_deemph_taps(rate, &t); min_block = ntaps;
        if (!ntaps) { badsyntax("deemphasis_nfm_ff: invalid sample rate (this function works only with specific sample rates)."); exit(255); }
        d_taps = (float *)csdr_amd_malloc(c, 4 * ntaps); MUST(csdr_amd_h2d(c, d_taps, t, 4 * ntaps));
    }
    size_t out_capacity(size_t n) override { return n + pre + 16; }
    long process(csdr_amd_ctx *c, const void *i, size_t n, void *o, size_t cap, size_t *cons) override
    {
        if (!pre) {
            long no = csdr_amd_fir_ff(c, (const float *)i, (float *)o, 1, (int)n, n, cap, d_taps, ntaps);
            MUST(no); *cons = (size_t)no; return no;
        }
        const size_t tot = pre + n;
        if (tot > tmp_cap) { if (d_tmp) csdr_amd_free(c, d_tmp); tmp_cap = tot + 64; d_tmp = (float *)csdr_amd_malloc(c, 4 * tmp_cap); if (!d_tmp) die("malloc"); }
        MUST(csdr_amd_memset(c, d_tmp, 0, 4 * pre));
        if (n) MUST(csdr_amd_d2d(c, d_tmp + pre, i, 4 * n));
        long no = csdr_amd_fir_ff(c, d_tmp, (float *)o, 1, (int)tot, tot, cap, d_taps, ntaps);
        MUST(no);
        const size_t from_zeros = (size_t)no < pre ? (size_t)no : pre;
        pre -= from_zeros; *cons = (size_t)no - from_zeros;
        return no;
    }
};

struct FastAgc : Stage {   // csdr.c:1377-1406
    int block; float ref; float *d_state;
    FastAgc(csdr_amd_ctx *c, int b, float r) : block(b), ref(r)
    {
        granule = b; flush_partial = false; init_state(c, b);
    }
    int next_bufsize(int) override { return block; }                 // csdr.c:1386
    void init_state(csdr_amd_ctx *c, int b)
    {
        d_state = (float *)csdr_amd_malloc(c, 4 * (2 * (size_t)b + 4)); MUST(csdr_amd_memset(c, d_state, 0, 4 * (2 * (size_t)b + 4)));
    }
    long process(csdr_amd_ctx *c, const void *i, size_t n, void *o, size_t, size_t *cons) override
    {
        const int nb = (int)(n / block); *cons = (size_t)nb * block;
        if (nb) MUST(csdr_amd_fastagc_ff(c, (const float *)i, (float *)o, 1, nb, block, n, n, ref, d_state));
        return (long)nb * block;
    }
};

struct FracDec : Stage {   // csdr.c:1465-1525
    csdr_amd_fracdec *d; float rate;
    FracDec(float r, int points, const float *taps, int ntaps, int the_bufsize) : rate(r)
    {
        d = csdr_amd_fracdec_create(r, points, taps, ntaps); if (!d) { badsyntax(csdr_amd_last_error()); exit(255); }
        csdr_amd_fracdec_set_cli_bufsize(d, the_bufsize);                           // the reference's window loop: positions of inexact rates depend on it
        min_block = (size_t)the_bufsize;
    }
    size_t out_capacity(size_t n) override { return (size_t)(n / rate) + 64; }
    int next_bufsize(int b) override { return (int)(b / rate); }     // csdr.c:1497
    long process(csdr_amd_ctx *c, const void *i, size_t n, void *o, size_t cap, size_t *cons) override
    {
        int processed = 0;
        long no = csdr_amd_fractional_decimator_ff(c, d, (const float *)i, (float *)o, 1, (int)n, n, cap, &processed);
        MUST(no); *cons = processed > 0 ? (size_t)processed : 0; return no;
    }
};

struct Bandpass : Stage {   // csdr.c:1810-1886
    csdr_amd_fftfilt *f; int inp; int n_taps, win;
    const char *ctl_format() override { return "%g %g\n"; }
    void retune(csdr_amd_ctx *, float lo, float hi) override       // new band edges, the overlap carries on (csdr.c:1862-1880)
    {
        fprintf(stderr, "csdr bandpass_fir_fft_cc: filter initialized, low_cut = %g, high_cut = %g\n", lo, hi);
        std::vector<csdr_complexf> t(n_taps);
        csdr_amd_firdes_bandpass_c(t.data(), n_taps, lo, hi, win);
        MUST(csdr_amd_fftfilt_set_taps(f, t.data(), n_taps));
    }
    Bandpass(csdr_amd_ctx *c, float lo, float hi, float tbw, int window, size_t block)
    {
        in_elem = 8; out_elem = 8; flush_partial = false;
        const int ntaps = csdr_amd_firdes_filter_len(tbw); n_taps = ntaps; win = window;
        int fft = csdr_amd_next_pow2(ntaps);
        if (fft - ntaps < 200) fft <<= 1;                                            // csdr.c:1834-1836
        inp = fft - ntaps + 1;
        fprintf(stderr, "csdr bandpass_fir_fft_cc: (fft_size = %d) = (taps_length = %d) + (input_size = %d) - 1\n(overlap_length = %d) = taps_length - 1\n", fft, ntaps, inp, ntaps - 1);
        std::vector<csdr_complexf> t(ntaps);
        csdr_amd_firdes_bandpass_c(t.data(), ntaps, lo, hi, window);
        f = csdr_amd_fftfilt_create(c, fft, t.data(), ntaps, 1, (int)(block / inp + 2));
        if (!f) die("fftfilt_create");
        granule = inp;
    }
    long process(csdr_amd_ctx *, const void *i, size_t n, void *o, size_t, size_t *cons) override
    {
        const int nb = (int)(n / inp); *cons = (size_t)nb * inp;
        if (nb) MUST(csdr_amd_fftfilt_process(f, (const csdr_complexf *)i, (csdr_complexf *)o, nb, n, n));
        return (long)nb * inp;
    }
};

struct DdcFwd : Stage {   // csdr.c:2255-2300
    csdr_amd_fastddc_fwd *f; csdr_fastddc_t ddc;
    DdcFwd(csdr_amd_ctx *c, int D, float tbw, size_t block)
    {
        in_elem = 8; out_elem = 8; flush_partial = false;
        if (csdr_amd_fastddc_init(&ddc, tbw, D, 0)) { badsyntax("error in fastddc_init()"); exit(1); }
        f = csdr_amd_fastddc_fwd_create(c, &ddc, (int)(block / ddc.input_size + 2)); if (!f) die("fastddc_fwd_create");
        granule = ddc.input_size;
    }
    size_t out_capacity(size_t n) override { return (n / ddc.input_size + 1) * (size_t)ddc.fft_size; }
    int next_bufsize(int) override { return ddc.fft_size; }          // csdr.c:2274
    long process(csdr_amd_ctx *, const void *i, size_t n, void *o, size_t, size_t *cons) override
    {
        const int nb = (int)(n / ddc.input_size); *cons = (size_t)nb * ddc.input_size;
        if (nb) MUST(csdr_amd_fastddc_fwd_process(f, (const csdr_complexf *)i, (csdr_complexf *)o, nb));
        return (long)nb * ddc.fft_size;
    }
};

struct DdcInv : Stage {   // csdr.c:2302-2378
    csdr_amd_fastddc_inv *f = nullptr; csdr_fastddc_t ddc; int maxb, dec, win; float tbw_;
    void build(csdr_amd_ctx *c, float shift)                         // the reference rebuilds everything on a retune, status included (csdr.c:2329-2376)
    {
        if (f) csdr_amd_fastddc_inv_destroy(f);
        if (csdr_amd_fastddc_init(&ddc, tbw_, dec, shift)) { badsyntax("error in fastddc_init()"); exit(1); }
        f = csdr_amd_fastddc_inv_create(c, tbw_, dec, &shift, 1, win, maxb); if (!f) die("fastddc_inv_create");
    }
    const char *ctl_format() override { return "%g\n"; }
    void retune(csdr_amd_ctx *c, float shift, float) override { build(c, shift); }
    int next_bufsize(int) override { return ddc.post_input_size / ddc.post_decimation; }   // csdr.c:2339
    DdcInv(csdr_amd_ctx *c, float shift, int D, float tbw, int window, size_t block) : dec(D), win(window), tbw_(tbw)
    {
        in_elem = 8; out_elem = 8; flush_partial = false;
        if (csdr_amd_fastddc_init(&ddc, tbw, D, shift)) { badsyntax("error in fastddc_init()"); exit(1); }
        maxb = (int)(block / ddc.fft_size + 2);
        build(c, shift);
        granule = ddc.fft_size;
    }
    size_t out_capacity(size_t n) override { return (n / ddc.fft_size + 1) * (size_t)(ddc.post_input_size / ddc.post_decimation + 2) + 16; }
    long process(csdr_amd_ctx *, const void *i, size_t n, void *o, size_t cap, size_t *cons) override
    {
        const int nb = (int)(n / ddc.fft_size); *cons = (size_t)nb * ddc.fft_size;
        int count = 0;
        if (nb) MUST(csdr_amd_fastddc_inv_process(f, (const csdr_complexf *)i, nb, (csdr_complexf *)o, cap, &count));
        return count;
    }
};

struct WfmChain : Stage {   // the fused README.md:66 chain as ONE command (extension: not in the reference's command list)
    csdr_amd_wfm *w; bool retunable;
    const char *ctl_format() override { return retunable ? "%g\n" : nullptr; }       // `wfm_chain_u8_s16 --fifo <path>`: the shift stage's control channel (csdr.c:881-923)
    void retune(csdr_amd_ctx *, float r, float) override { MUST(csdr_amd_wfm_set_rate(w, 0, r)); fprintf(stderr, "csdr %s: reinitialized to %g\n", g_cmd, r); }
    WfmChain(csdr_amd_ctx *c, float shift, size_t block, bool with_ctl) : retunable(with_ctl)
    {
        in_elem = 2; out_elem = 2; granule = 1024;
        std::vector<float> t(79);
        const int nt = csdr_amd_firdes_filter_len(0.05f);
        t.resize(nt); csdr_amd_firdes_lowpass_f(t.data(), nt, 0.05f, CSDR_WINDOW_HAMMING);
        // The rate-per-stream object, always: its one stream can be retuned between two calls (control channel), and its kernel spreads ONE stream over the 16 columns of
        // a tile (16 time segments) where the shared-rate kernel fills one of 16 -- a 4 M-sample block took the latter 0.41 ms, 10 GS/s before a byte was read
        // (CSDR_AMD_CLI_TIMING, round 5).  CSDR_AMD_CLI_SHARED=1: the shared-rate object as before.
        // Blocks under 1 Mi samples stay on the shared-rate object (256 Ki: 3.8 against 2.5 GS/s: per call the rate-per-stream object also looks its seeds up).
        w = (with_ctl || (block >= (1u << 20) && !getenv("CSDR_AMD_CLI_SHARED"))) ? csdr_amd_wfm_create_rates(c, 1, &shift, 10, t.data(), nt, 5, 50e-6f, 48000, block + 1024)
                                                         : csdr_amd_wfm_create(c, 1, shift, 10, t.data(), nt, 5, 50e-6f, 48000, block + 1024);
        if (!w) die("wfm_create");
        if (csdr_amd_wfm_fallback(w)) fprintf(stderr, "csdr %s: note: this shape runs on the fallback kernels (k_wfm_front + k_wfm_back), not on the matrix-core chain kernel\n", g_cmd);
    }
    size_t out_capacity(size_t n) override { return n / 50 + 64; }
    int next_bufsize(int b) override { return b / 50; }
    long process(csdr_amd_ctx *, const void *i, size_t n, void *o, size_t cap, size_t *cons) override
    {   // (one stream: the pitch only has to satisfy the 16-byte rule -- a stream's last block can have any length)
        *cons = n; long na = csdr_amd_wfm_process(w, (const uint8_t *)i, (2 * n + 127) & ~(size_t)127, n, (int16_t *)o, nullptr, cap); MUST(na); return na; }
};

// `CSDR_AMD_RESIDENT=1 csdr wfm_chain_u8_s16 <shift_rate>`: the same chain through the RESIDENT form (csdr_amd_wfm_ring_*): one persistent grid walks a ring of
// the reference's own blocks -- 16384 samples per read, csdr.c:189-193, 330-392 -- , no kernel launch per block; a live stream (a block every 6.8 ms at 2.4 MS/s) keeps
// the grid on the GPU between blocks (idle time 20 ms), a stalled one lets it go.  Whole blocks only: what is left of the stream behind its last whole block is dropped at
// EOF, as the reference's stages drop a partial the_bufsize read (csdr.c:232-247).
struct WfmRingStage : Stage {
    csdr_amd_wfm_ring *r; size_t T; bool retunable;
    const char *ctl_format() override { return retunable ? "%g\n" : nullptr; }
    void retune(csdr_amd_ctx *, float rt, float) override { MUST(csdr_amd_wfm_ring_set_rate(r, rt)); fprintf(stderr, "csdr %s: reinitialized to %g\n", g_cmd, rt); }
    WfmRingStage(csdr_amd_ctx *c, float shift, bool with_ctl) : T(16384), retunable(with_ctl)
    {
        in_elem = 2; out_elem = 2; granule = T; flush_partial = false; min_block = T / 4;
        const int nt = csdr_amd_firdes_filter_len(0.05f);
        std::vector<float> t(nt); csdr_amd_firdes_lowpass_f(t.data(), nt, 0.05f, CSDR_WINDOW_HAMMING);
        r = csdr_amd_wfm_ring_create(c, 1, shift, 10, t.data(), nt, 5, 50e-6f, 48000, T, 8);
        if (!r) die("wfm_ring_create");
        MUST(csdr_amd_wfm_ring_set_timeouts(r, 20000.0, 1000.0));
        fprintf(stderr, "csdr %s: resident grid (%d workgroups), ring of %d blocks of %zu samples\n", g_cmd, csdr_amd_wfm_ring_grid(r), csdr_amd_wfm_ring_slots(r), T);
    }
    ~WfmRingStage() { csdr_amd_wfm_ring_destroy(r); }
    size_t out_capacity(size_t n) override { return n / 50 + 64; }
    int next_bufsize(int b) override { return b / 50; }
    long process(csdr_amd_ctx *c, const void *i, size_t n, void *o, size_t cap, size_t *cons) override
    {
        const size_t nb = n / T; *cons = nb * T;
        const int depth = csdr_amd_wfm_ring_slots(r) - 2;
        long total = 0;
        for (size_t b0 = 0; b0 < nb; b0 += depth) {                  // groups of as many blocks as the ring holds in flight: inputs in, posted, collected in order
            const size_t nbk = nb - b0 < (size_t)depth ? nb - b0 : (size_t)depth;
            const long long s0 = csdr_amd_wfm_ring_submitted(r);
            for (size_t b = 0; b < nbk; b++) {
                MUST(csdr_amd_wfm_ring_acquire(r, s0 + (long long)b, 0));
                size_t pitch; uint8_t *slot = csdr_amd_wfm_ring_input(r, s0 + (long long)b, &pitch);
                MUST(csdr_amd_d2d(c, slot, (const uint8_t *)i + 2 * T * (b0 + b), 2 * T));
            }
            MUST(csdr_amd_ctx_sync(c));                              // the blocks lie in their slots before they are posted
            for (size_t b = 0; b < nbk; b++) { const long long k = csdr_amd_wfm_ring_submit(r); MUST((int)(k < 0 ? k : 0)); }
            for (size_t b = 0; b < nbk; b++) {
                const long na = csdr_amd_wfm_ring_wait(r, s0 + (long long)b, 0); MUST((int)(na < 0 ? na : 0));
                if ((size_t)(total + na) > cap) die("wfm ring: output buffer too small");
                size_t op; const int16_t *out = csdr_amd_wfm_ring_output(r, s0 + (long long)b, &op);
                MUST(csdr_amd_d2d(c, (int16_t *)o + total, out, 2 * (size_t)na));
                total += na;
            }
        }
        return total;
    }
};

struct DdcFront : Stage {   // convert_u8_f | shift_addition_cc r | fir_decimate_cc D tbw window as ONE command (extension): the head of the NFM / AM / SSB chains
    csdr_amd_ddc *d; int dec; bool retunable;
    const char *ctl_format() override { return retunable ? "%g\n" : nullptr; }
    void retune(csdr_amd_ctx *, float r, float) override { MUST(csdr_amd_ddc_set_rate(d, 0, r)); fprintf(stderr, "csdr %s: reinitialized to %g\n", g_cmd, r); }
    DdcFront(csdr_amd_ctx *c, float shift, int D, float tbw, int window, size_t block, bool with_ctl) : dec(D), retunable(with_ctl)
    {
        in_elem = 2; out_elem = 8; granule = 1024;
        const int nt = csdr_amd_firdes_filter_len(tbw);
        std::vector<float> t(nt); csdr_amd_firdes_lowpass_f(t.data(), nt, 0.5f / (float)D, window);       // csdr.c:1144-1158
        d = with_ctl ? csdr_amd_ddc_create_rates(c, 1, &shift, D, t.data(), nt, block + 1024) : csdr_amd_ddc_create(c, 1, shift, D, t.data(), nt, block + 1024);
        if (!d) die("ddc_create");
    }
    size_t out_capacity(size_t n) override { return n / dec + 64; }
    int next_bufsize(int b) override { return b / dec; }
    long process(csdr_amd_ctx *, const void *i, size_t n, void *o, size_t cap, size_t *cons) override
    {
        *cons = n; long no = csdr_amd_ddc_process(d, (const uint8_t *)i, (2 * n + 127) & ~(size_t)127, n, (csdr_complexf *)o, cap); MUST(no);
        if (!noted && n >= 4096 && csdr_amd_ddc_fallback(d)) { noted = true; fprintf(stderr, "csdr %s: note: this shape runs on the plain kernel (k_ddc_direct), not on the matrix-core front end\n", g_cmd); }
        return no;
    }
    bool noted = false;
};

struct NfmChain : Stage {   // the README.md:87 chain as ONE command (extension)
    csdr_amd_nfm *w; int dec; bool retunable;
    const char *ctl_format() override { return retunable ? "%g\n" : nullptr; }
    void retune(csdr_amd_ctx *, float r, float) override { MUST(csdr_amd_nfm_set_rate(w, 0, r)); fprintf(stderr, "csdr %s: reinitialized to %g\n", g_cmd, r); }
    NfmChain(csdr_amd_ctx *c, float shift, int D, float tbw, size_t block, bool with_ctl) : dec(D), retunable(with_ctl)
    {
        in_elem = 2; out_elem = 2; granule = 1024;
        const int nt = csdr_amd_firdes_filter_len(tbw);
        std::vector<float> t(nt); csdr_amd_firdes_lowpass_f(t.data(), nt, 0.5f / (float)D, CSDR_WINDOW_HAMMING);
        w = (with_ctl || (block >= (1u << 20) && !getenv("CSDR_AMD_CLI_SHARED"))) ? csdr_amd_nfm_create_rates(c, 1, &shift, D, t.data(), nt, 48000, 1024, 1.0f, 1.0f, block + 1024)      // (as WfmChain)
                     : csdr_amd_nfm_create(c, 1, shift, D, t.data(), nt, 48000, 1024, 1.0f, 1.0f, block + 1024);      // fastagc_ff defaults csdr.c:1379-1391
        if (!w) die("nfm_create");
    }
    size_t out_capacity(size_t n) override { return n / dec + 4096; }
    int next_bufsize(int b) override { return b / dec; }
    long process(csdr_amd_ctx *, const void *i, size_t n, void *o, size_t cap, size_t *cons) override
    {
        *cons = n; long na = csdr_amd_nfm_process(w, (const uint8_t *)i, (2 * n + 127) & ~(size_t)127, n, (int16_t *)o, nullptr, cap); MUST(na);
        if (!noted && n >= 4096 && csdr_amd_ddc_fallback(csdr_amd_nfm_front_end(w))) { noted = true; fprintf(stderr, "csdr %s: note: the front end of this shape runs on the plain kernel (k_ddc_direct), not on the matrix-core kernel\n", g_cmd); }
        return na;
    }
    bool noted = false;
};

struct DecimatingShift : Stage {   // csdr.c:851-875: one libcsdr call per the_bufsize samples, status carried between calls
    int dec, bufsize; float dsa[3]; void *d_dsa, *d_status;
    DecimatingShift(csdr_amd_ctx *c, float rate, int decimation, int the_bufsize) : dec(decimation), bufsize(the_bufsize)
    {
        in_elem = 8; out_elem = 8; granule = the_bufsize; flush_partial = true;
        csdr_amd_shift_addition_init(rate * (float)decimation, dsa);        // decimating_shift_addition_init libcsdr_gpl.c:126-129
        d_dsa = csdr_amd_malloc(c, 12); d_status = csdr_amd_malloc(c, 12);
        MUST(csdr_amd_h2d(c, d_dsa, dsa, 12)); MUST(csdr_amd_memset(c, d_status, 0, 12));
    }
    size_t out_capacity(size_t n) override { return n / dec + n / bufsize + 16; }
    int next_bufsize(int b) override { return b / dec; }             // csdr.c:861
    long process(csdr_amd_ctx *c, const void *i, size_t n, void *o, size_t, size_t *cons) override
    {
        *cons = n;
        long total = 0;
        for (size_t at = 0; at < n; at += bufsize) {
            const int m = (int)((n - at < (size_t)bufsize) ? n - at : bufsize);
            MUST(csdr_amd_decimating_shift_addition_cc(c, (const csdr_complexf *)i + at, (csdr_complexf *)o + total, 1, m, m, m, d_dsa, dec, d_status));
            int st[3]; MUST(csdr_amd_d2h(c, st, d_status, 12));
            total += st[2];
        }
        return total;
    }
};


// ------------------------------------------------------------------ f2 commands (csdr.c:634-672, 927-983, 1088-1112, 1338-1375, 1569-1661)
struct CfToF : Stage {   // amdemod_cf / amdemod_estimator_cf / realpart_cf / logpower_cf
    int op; float p0;
    CfToF(int o, float a) : op(o), p0(a) { in_elem = 8; out_elem = 4; }
    long process(csdr_amd_ctx *c, const void *i, size_t n, void *o, size_t, size_t *cons) override
    {
        *cons = n;
        const csdr_complexf *x = (const csdr_complexf *)i; float *y = (float *)o;
        switch (op) {
            case 0: MUST(csdr_amd_amdemod_cf(c, x, y, n)); break;
            case 1: MUST(csdr_amd_amdemod_estimator_cf(c, x, y, n, 0.f, 0.f)); break;            // csdr.c:1108
            case 2: MUST(csdr_amd_realpart_cf(c, x, y, n)); break;
            default: MUST(csdr_amd_logpower_cf(c, x, y, n, p0)); break;
        }
        return (long)n;
    }
};
struct Gain : Stage {    // csdr.c:658-672
    float g; Gain(float gg) : g(gg) { granule = 4; }
    long process(csdr_amd_ctx *c, const void *i, size_t n, void *o, size_t, size_t *cons) override
    { *cons = n; MUST(csdr_amd_gain_ff(c, (const float *)i, (float *)o, n, g)); return (long)n; }
};
struct FmdemodAtan : Stage {   // csdr.c:962-977
    float *d_last;
    FmdemodAtan(csdr_amd_ctx *c) { in_elem = 8; out_elem = 4; d_last = (float *)csdr_amd_malloc(c, 4); MUST(csdr_amd_memset(c, d_last, 0, 4)); }
    long process(csdr_amd_ctx *c, const void *i, size_t n, void *o, size_t, size_t *cons) override
    { *cons = n; MUST(csdr_amd_fmdemod_atan_cf(c, (const csdr_complexf *)i, (float *)o, 1, n, n, n, d_last)); return (long)n; }
};
struct DcBlock : Stage {       // csdr.c:927-939 (a = 0 selects 0.999)
    float *d_state;
    DcBlock(csdr_amd_ctx *c) { d_state = (float *)csdr_amd_malloc(c, 8); MUST(csdr_amd_memset(c, d_state, 0, 8)); }
    long process(csdr_amd_ctx *c, const void *i, size_t n, void *o, size_t, size_t *cons) override
    { *cons = n; MUST(csdr_amd_dcblock_ff(c, (const float *)i, (float *)o, 1, n, n, n, 0.f, d_state)); return (long)n; }
};
struct FastDcBlock : Stage {   // csdr.c:941-960
    int block; float *d_last;
    FastDcBlock(csdr_amd_ctx *c, int b) : block(b) { granule = b; flush_partial = false; d_last = (float *)csdr_amd_malloc(c, 4); MUST(csdr_amd_memset(c, d_last, 0, 4)); }
    int next_bufsize(int) override { return block; }
    long process(csdr_amd_ctx *c, const void *i, size_t n, void *o, size_t, size_t *cons) override
    {
        const int nb = (int)(n / block); *cons = (size_t)nb * block;
        if (nb) MUST(csdr_amd_fastdcblock_ff(c, (const float *)i, (float *)o, 1, nb, block, n, n, d_last));
        return (long)nb * block;
    }
};
struct Agc : Stage {           // csdr.c:1338-1375: one agc_ff call per the_bufsize samples
    short hang, wait; float ref, attack, decay, maxg, alpha; int bufsize; float *d_gain;
    Agc(csdr_amd_ctx *c, int the_bufsize) : bufsize(the_bufsize)
    {
        granule = the_bufsize;
        d_gain = (float *)csdr_amd_malloc(c, 4); const float one = 1.0f; MUST(csdr_amd_h2d(c, d_gain, &one, 4));
    }
    long process(csdr_amd_ctx *c, const void *i, size_t n, void *o, size_t, size_t *cons) override
    { *cons = n; MUST(csdr_amd_agc_ff(c, (const float *)i, (float *)o, 1, n, bufsize, n, n, ref, attack, decay, maxg, hang, wait, alpha, d_gain)); return (long)n; }
};
struct FftCc : Stage {         // csdr.c:1569-1641 (binary output; --octave text mode is not offered)
    csdr_amd_fftcc *f; int fft, every;
    FftCc(csdr_amd_ctx *c, int fft_size, int every_n, int window, size_t block) : fft(fft_size), every(every_n)
    {
        in_elem = 8; out_elem = 8; granule = every_n; flush_partial = false;
        f = csdr_amd_fftcc_create(c, fft_size, every_n, window, (int)(block / every_n + 2)); if (!f) die("fftcc_create");
    }
    size_t out_capacity(size_t n) override { return (n / every + 1) * (size_t)fft; }
    int next_bufsize(int) override { return fft; }                   // csdr.c:1596
    long process(csdr_amd_ctx *, const void *i, size_t n, void *o, size_t, size_t *cons) override
    { size_t used = 0; int nf = csdr_amd_fftcc_process(f, (const csdr_complexf *)i, n, (csdr_complexf *)o, &used); MUST(nf); *cons = used; return (long)nf * fft; }
};


// ------------------------------------------------------------------ f3 commands (csdr.c:1745-1768, 1891-1919)
struct AdpcmEnc : Stage {
    int *d_state;
    AdpcmEnc(csdr_amd_ctx *c) { in_elem = 2; out_elem = 1; granule = 2; flush_partial = false; d_state = (int *)csdr_amd_malloc(c, 8); MUST(csdr_amd_memset(c, d_state, 0, 8)); }
    int next_bufsize(int b) override { return b / 2; }               // csdr.c:1893
    size_t out_capacity(size_t n) override { return n / 2 + 16; }
    long process(csdr_amd_ctx *c, const void *i, size_t n, void *o, size_t, size_t *cons) override
    { n &= ~(size_t)1; *cons = n; MUST(csdr_amd_encode_ima_adpcm_i16_u8(c, (const int16_t *)i, (uint8_t *)o, 1, n, n, n / 2, d_state)); return (long)(n / 2); }
};
struct AdpcmDec : Stage {
    int *d_state;
    AdpcmDec(csdr_amd_ctx *c) { in_elem = 1; out_elem = 2; d_state = (int *)csdr_amd_malloc(c, 8); MUST(csdr_amd_memset(c, d_state, 0, 8)); }
    int next_bufsize(int b) override { return b * 2; }               // csdr.c:1910
    size_t out_capacity(size_t n) override { return 2 * n + 16; }
    long process(csdr_amd_ctx *c, const void *i, size_t n, void *o, size_t, size_t *cons) override
    { *cons = n; MUST(csdr_amd_decode_ima_adpcm_u8_i16(c, (const uint8_t *)i, (int16_t *)o, 1, n, n, 2 * n, d_state)); return (long)(2 * n); }
};
struct CompressFft : Stage {
    int fft;
    CompressFft(int f) : fft(f) { in_elem = 4; out_elem = 1; granule = f; flush_partial = false; }
    int next_bufsize(int) override { return fft + 10; }              // csdr.c:1752
    size_t out_capacity(size_t n) override { return (n / fft + 1) * (size_t)((fft + 10) / 2) + 16; }
    long process(csdr_amd_ctx *c, const void *i, size_t n, void *o, size_t, size_t *cons) override
    {
        const int nb = (int)(n / fft); *cons = (size_t)nb * fft;
        if (nb) MUST(csdr_amd_compress_fft_adpcm_f_u8(c, (const float *)i, (uint8_t *)o, nb, fft));
        return (long)nb * ((fft + 10) / 2);
    }
};

// ------------------------------------------------------------------ wire protocol (csdr.c:325-419)
int g_dynamic = 0, g_fixed = 1024, g_fixed_big = 16384, g_print = 0;
void parse_env()
{   // csdr.c:393-419
    if (const char *e = getenv("CSDR_DYNAMIC_BUFSIZE_ON")) { g_dynamic = !!atoi(e); g_fixed = 0; }
    else if (const char *f = getenv("CSDR_FIXED_BUFSIZE")) g_fixed = g_fixed_big = atoi(f);
    if (const char *e = getenv("CSDR_PRINT_BUFSIZES")) g_print = atoi(e);
}
int unitround(int what) { return what <= 0 ? 4 : ((what - 1) & ~3) + 4; }   // csdr.c:352-358

bool read_full(void *buf, size_t bytes, size_t *got)
{   // blocking read until `bytes` or EOF; returns false on EOF (with *got possibly > 0)
    size_t have = 0;
    while (have < bytes) {
        ssize_t r = read(STDIN_FILENO, (char *)buf + have, bytes - have);
        if (r < 0) { if (errno == EINTR) continue; fprintf(stderr, "csdr %s: read error on stdin (%s), treating it as the end of the stream\n", g_cmd, strerror(errno)); *got = have; return false; }
        if (r == 0) { *got = have; return false; }
        have += (size_t)r;
    }
    *got = have; return true;
}
void write_full(const void *buf, size_t bytes)
{
    size_t done = 0;
    while (done < bytes) {
        ssize_t r = write(STDOUT_FILENO, (const char *)buf + done, bytes - done);
        if (r < 0) { if (errno == EINTR) continue; exit(0); }     // downstream closed: end quietly like SIGPIPE would
        done += (size_t)r;
    }
}
int get_bufsize(bool big)
{   // csdr.c:330-341: in dynamic mode the first 8 bytes of stdin are "csdr" + int
    if (!g_dynamic) return unitround(big ? g_fixed_big : g_fixed);
    int first[2] = {0, 0}; size_t got = 0;
    read_full(first, 8, &got);
    if (got != 8 || memcmp(first, "csdr", 4) != 0) {
        badsyntax("warning! Did not match preamble on the beginning of the stream. You should put \"csdr setbuf <buffer size>\" at the beginning of the chain! Falling back to default buffer size: 1024");
        return 1024;
    }
    if (first[1] <= 0) { badsyntax("warning! Invalid buffer size."); exit(254); }
    if (g_print) fprintf(stderr, "csdr %s: buffer size set to %d\n", g_cmd, unitround(first[1]));
    return unitround(first[1]);
}
void send_bufsize(int size)
{   // csdr.c:375-391
    if (!g_dynamic) return;
    if (g_print) fprintf(stderr, "csdr %s: next process proposed input buffer size is %d\n", g_cmd, size);
    int first[2]; memcpy(first, "csdr", 4); first[1] = size;
    write_full(first, 8);
}

// ------------------------------------------------------------------ control channel (csdr.c:252-323)
struct Control {
    int fd = 0; char buf[1024]; int fill = 0;
    bool open_from(int argc, char **argv)
    {
        if (argc < 4) return false;
        if (!strcmp(argv[2], "--fifo")) { fprintf(stderr, "csdr %s: fifo control mode on\n", g_cmd); fd = open(argv[3], O_RDONLY); }
        else if (!strcmp(argv[2], "--fd")) { if (sscanf(argv[3], "%d", &fd) <= 0) return false; fprintf(stderr, "csdr %s: fd control mode on, fd=%d\n", g_cmd, fd); }
        else return false;
        if (fd <= 0) { fd = 0; return false; }
        fcntl(fd, F_SETFL, fcntl(fd, F_GETFL, 0) | O_NONBLOCK);
        return true;
    }
    // newest complete line, parsed with the command's scanf format; non-blocking
    bool poll(const char *fmt, float *a, float *b)
    {
        if (!fd) return false;
        const ssize_t r = read(fd, buf + fill, sizeof(buf) - 1 - fill);
        if (r <= 0) return false;
        const int end = fill + (int)r;
        int prev = 0, last = 0;
        for (int i = 0; i < end; i++) if (buf[i] == '\n') { prev = last; last = i + 1; }
        if (!last) { fill = end; return false; }
        buf[end] = 0;
        float x = 0, y = 0;
        const int n = sscanf(buf + prev, fmt, &x, &y);
        memmove(buf, buf + last, end - last); fill = end - last;
        if (n < 1) return false;
        *a = x; *b = y; return true;
    }
    void wait_first(const char *fmt, float *a, float *b) { while (!poll(fmt, a, b)) usleep(10000); }
};


// ------------------------------------------------------------------ device hand-off between adjacent csdr processes
// north_star: "the stdin->stdout pipe never round-trips to host between stages ... existing shell pipelines drop in unchanged".  In `csdr a | csdr b` both
// ends of the pipe are this binary, both talk to the same GPU, and the samples a produces are already in HBM: writing them to the pipe costs a D2H copy, two
// pipe copies and an H2D copy per stage.  Instead:
//   * b (the consumer), first thing in main(), listens on an abstract unix socket named after the PIPE it reads (st_dev:st_ino of fd 0 -- both ends of a pipe
//     report the same inode);
//   * a (the producer), just before it would write its first byte, tries to connect to the socket named after fd 1.  No listener (the consumer is some other
//     program, or stdout is no pipe): plain bytes, as ever -- a foreign peer never sees anything but the reference's wire format.  Connected: a sends HELLO with
//     the HIP IPC handle of a ring of NBUF output slots in its device memory; b maps it (hipIpcOpenMemHandle) and answers ACK, or NAK (another device, IPC not
//     available in this container, ...) after which both fall back to bytes;
//   * b decides with one poll() on {stdin, listener}: a connection means hand-off, bytes (or EOF) on stdin mean a producer that writes bytes;
//   * per block a writes its result into a free slot, waits for the kernels (stream-ordered event), sends the token {slot, bytes} over the socket; b copies
//     the slot into its own input buffer device-to-device on its stream and returns the slot as a credit once that copy has run.  End of stream = the socket
//     closes.  The 8-byte "csdr"+int preamble of CSDR_DYNAMIC_BUFSIZE_ON still travels through the pipe itself.
// CSDR_AMD_IPC=0 switches the whole mechanism off; CSDR_AMD_IPC_WAIT_MS (default 250) is how long a producer keeps trying to find a listener that is not there
// yet (a consumer of ours listens within a millisecond of its exec, long before the producer's HIP start-up is over); CSDR_AMD_IPC_VERBOSE=1 prints one line
// per link on stderr.
struct IpcHello { char magic[8]; int version, device; char bus_id[32]; hipIpcMemHandle_t mem; unsigned n_slots, reserved; unsigned long long slot_bytes; };
struct IpcToken { unsigned slot, reserved; unsigned long long bytes; };
const char IPC_MAGIC[8] = {'c', 's', 'd', 'r', 'H', 'B', 'M', '1'};
int g_ipc_listen = -1;                       // consumer side: the listening socket named after stdin's pipe
bool ipc_enabled() { const char *e = getenv("CSDR_AMD_IPC"); return !e || atoi(e) != 0; }
bool ipc_verbose() { const char *e = getenv("CSDR_AMD_IPC_VERBOSE"); return e && atoi(e) != 0; }
bool ipc_pipe_name(int fd, struct sockaddr_un *sa, socklen_t *len)
{
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISFIFO(st.st_mode)) return false;
    memset(sa, 0, sizeof *sa); sa->sun_family = AF_UNIX;
    const int n = snprintf(sa->sun_path + 1, sizeof sa->sun_path - 1, "csdr_amd.pipe.%llx.%llx", (unsigned long long)st.st_dev, (unsigned long long)st.st_ino);
    *len = (socklen_t)(offsetof(struct sockaddr_un, sun_path) + 1 + n);                 // abstract name: leading NUL, no file system entry to clean up
    return true;
}
void ipc_listen_on_stdin()
{
    struct sockaddr_un sa; socklen_t len;
    if (!ipc_enabled() || !ipc_pipe_name(STDIN_FILENO, &sa, &len)) return;
    const int fd = socket(AF_UNIX, SOCK_SEQPACKET | SOCK_CLOEXEC, 0);
    if (fd < 0) return;
    if (bind(fd, (struct sockaddr *)&sa, len) != 0 || listen(fd, 1) != 0) { close(fd); return; }
    g_ipc_listen = fd;
}
// the abstract socket name is predictable (dev:ino of the pipe): only a peer of OUR uid gets the IPC handle / is believed (SO_PEERCRED)
bool ipc_peer_is_ours(int fd)
{
    struct ucred cr; socklen_t len = sizeof cr;
    return getsockopt(fd, SOL_SOCKET, SO_PEERCRED, &cr, &len) == 0 && len == sizeof cr && cr.uid == geteuid();
}
void ipc_device_id(int device, char bus_id[32]) { memset(bus_id, 0, 32); if (hipDeviceGetPCIBusId(bus_id, 32, device) != hipSuccess) bus_id[0] = 0; }

// consumer: blocks until the producer has connected (-> the connected socket, ring mapped) or has started to write bytes / closed the pipe (-> -1).  Once.
struct IpcSource { int fd = -1; char *ring = nullptr; unsigned n_slots = 0; size_t slot_bytes = 0; };
bool g_ipc_source_decided = false; IpcSource g_ipc_source;
IpcSource *ipc_source_decide(int device)
{
    if (g_ipc_source_decided) return g_ipc_source.fd >= 0 ? &g_ipc_source : nullptr;
    g_ipc_source_decided = true;
    if (g_ipc_listen < 0) return nullptr;
    int conn = -1;
    for (;;) {
        struct pollfd pf[2] = {{g_ipc_listen, POLLIN, 0}, {STDIN_FILENO, POLLIN, 0}};
        if (poll(pf, 2, -1) < 0) { if (errno == EINTR) continue; break; }
        if (pf[0].revents & POLLIN) { conn = accept4(g_ipc_listen, nullptr, nullptr, SOCK_CLOEXEC); break; }      // (checked first: a producer of ours connects BEFORE it writes the preamble)
        if (pf[1].revents) break;                                                                                  // bytes, EOF or an error on stdin: a producer that writes bytes
    }
    close(g_ipc_listen); g_ipc_listen = -1;
    if (conn < 0) return nullptr;
    if (!ipc_peer_is_ours(conn)) { close(conn); return nullptr; }
    IpcHello h; int ack = 0;
    char mine[32]; ipc_device_id(device, mine);
    void *ring = nullptr;
    const bool test_nak = getenv("CSDR_AMD_IPC_TEST_NAK") != nullptr;  // (tests: refuse as if the handle could not be opened -- the fallback to bytes after a connection)
    if (recv(conn, &h, sizeof h, 0) == (ssize_t)sizeof h && !test_nak && !memcmp(h.magic, IPC_MAGIC, 8) && h.version == 1 && mine[0] && !strncmp(h.bus_id, mine, 32) &&
        hipIpcOpenMemHandle(&ring, h.mem, hipIpcMemLazyEnablePeerAccess) == hipSuccess) ack = 1;
    else (void)hipGetLastError();
    if (send(conn, &ack, sizeof ack, MSG_NOSIGNAL) != (ssize_t)sizeof ack) ack = 0;
    if (!ack) { if (ipc_verbose()) fprintf(stderr, "csdr %s: device hand-off from the previous process refused (another device, or HIP IPC is not available): bytes through the pipe\n", g_cmd); close(conn); return nullptr; }
    g_ipc_source.fd = conn; g_ipc_source.ring = (char *)ring; g_ipc_source.n_slots = h.n_slots; g_ipc_source.slot_bytes = (size_t)h.slot_bytes;
    if (ipc_verbose()) fprintf(stderr, "csdr %s: input arrives by device hand-off (%u slots of %zu bytes in the previous process's HBM ring)\n", g_cmd, h.n_slots, g_ipc_source.slot_bytes);
    return &g_ipc_source;
}

// producer: -> connected socket after HELLO / ACK, or -1 (bytes).  `ring`: n_slots * slot_bytes of device memory from hipMalloc (the base of the allocation)
int ipc_sink_connect(int device, void *ring, unsigned n_slots, size_t slot_bytes)
{
    struct sockaddr_un sa; socklen_t len;
    if (!ipc_enabled() || !ipc_pipe_name(STDOUT_FILENO, &sa, &len)) return -1;
    long wait_ms = 250; if (const char *e = getenv("CSDR_AMD_IPC_WAIT_MS")) wait_ms = atol(e);
    int fd = -1;
    struct timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;) {
        fd = socket(AF_UNIX, SOCK_SEQPACKET | SOCK_CLOEXEC, 0);
        if (fd < 0) return -1;
        if (connect(fd, (struct sockaddr *)&sa, len) == 0) break;
        close(fd); fd = -1;
        struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
        if ((t1.tv_sec - t0.tv_sec) * 1000 + (t1.tv_nsec - t0.tv_nsec) / 1000000 >= wait_ms) return -1;
        usleep(2000);
    }
    if (!ipc_peer_is_ours(fd)) { close(fd); return -1; }
    IpcHello h; memset(&h, 0, sizeof h);
    memcpy(h.magic, IPC_MAGIC, 8); h.version = 1; h.device = device; ipc_device_id(device, h.bus_id); h.n_slots = n_slots; h.slot_bytes = slot_bytes;
    int ack = 0;
    if (hipIpcGetMemHandle(&h.mem, ring) != hipSuccess) { (void)hipGetLastError(); memset(h.magic, 0, 8); }      // (an invalid HELLO: the consumer answers NAK)
    if (send(fd, &h, sizeof h, MSG_NOSIGNAL) != (ssize_t)sizeof h || recv(fd, &ack, sizeof ack, 0) != (ssize_t)sizeof ack || !ack) {
        if (ipc_verbose()) fprintf(stderr, "csdr %s: device hand-off to the next process refused: bytes through the pipe\n", g_cmd);
        close(fd); return -1;
    }
    if (ipc_verbose()) fprintf(stderr, "csdr %s: output leaves by device hand-off (%u slots of %zu bytes)\n", g_cmd, n_slots, slot_bytes);
    return fd;
}

// ------------------------------------------------------------------ the streaming loop: one or more stages, intermediates in HBM
// Host side = three threads around the GPU work so that read(), PCIe and write() overlap (the reference overlaps them with one process per
// command): a READER fills pinned buffers from stdin, the main thread queues H2D -> kernels -> D2H on the context's stream without waiting,
// a WRITER waits for each block's completion event and writes it to stdout.
// Latency: the reader hands a block on as soon as `min_elems` elements have arrived (the reference's the_bufsize, csdr.c:232-247, 332, rounded up to
// the operator's granule) and only takes more when more is ALREADY waiting in the pipe, up to CSDR_AMD_BLOCK elements: a live 2.4 MS/s or 48 kS/s
// stream moves in the reference's own block sizes (6.8 ms / 21 ms), a file or a fast producer in large blocks.
struct HostBuf { char *p = nullptr; size_t cap = 0, bytes = 0; bool eof = false; hipEvent_t ev = nullptr; bool pending = false;
                 const char *dev = nullptr; int slot = -1; };       // device hand-off: where the block lies in the producer's ring; the slot to give back once it is copied (-1: not the slot's last piece)
struct BufQueue {
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER; pthread_cond_t cv = PTHREAD_COND_INITIALIZER; std::vector<HostBuf *> q;
    void push(HostBuf *b) { pthread_mutex_lock(&mu); q.push_back(b); pthread_cond_signal(&cv); pthread_mutex_unlock(&mu); }
    HostBuf *pop() { pthread_mutex_lock(&mu); while (q.empty()) pthread_cond_wait(&cv, &mu); HostBuf *b = q.front(); q.erase(q.begin()); pthread_mutex_unlock(&mu); return b; }
};
struct IoThreads {
    int device = 0; size_t min_bytes = 0, max_bytes = 0;
    BufQueue free_in, full_in, free_out, full_out;
    IpcSource *src = nullptr; BufQueue copied_in;                    // device hand-off, consumer side: blocks whose device copy is queued (their slots go back as credits)
    int sink_fd = -1; HostBuf *sink_bufs = nullptr;                  // producer side: tokens out, credits back; sink_bufs[slot]
    int n_in_bufs = 0;                                               // how many input buffers circulate (reader -> main -> credit thread -> reader)
    std::atomic<bool> sink_closing{false};                           // the writer has sent its last token and shut the socket down: EOF on the credit side is then the normal end
};

// stdin is a regular file (`csdr ... < file`): one thread's read() copies ~14 GB/s out of the page cache into a pinned buffer -- 7 GS/s of u8 IQ, a third of what
// `cat` and the PCIe link manage (profiles/r4_cli_bench.txt).  The block's bytes are then fetched by CSDR_AMD_READERS threads (default 4) with pread() on disjoint
// slices; the file offset is advanced by hand.  Pipes, sockets and ttys keep the single reader (the kernel serialises them anyway).
struct FileInput { bool regular = false; off_t pos = 0; int threads = 1; };
FileInput g_file_in;
void file_input_init()
{
    struct stat st;
    if (fstat(STDIN_FILENO, &st) != 0 || !S_ISREG(st.st_mode)) return;
    const off_t at = lseek(STDIN_FILENO, 0, SEEK_CUR);
    if (at < 0) return;
    int k = 4; if (const char *e = getenv("CSDR_AMD_READERS")) k = atoi(e);
    if (k < 1) k = 1; if (k > 16) k = 16;
    g_file_in.regular = k > 1; g_file_in.pos = at; g_file_in.threads = k;
}
struct PreadJob { char *dst; size_t len; off_t off; size_t got; int err; };
void pread_all(PreadJob *j)
{
    j->got = 0; j->err = 0;
    while (j->got < j->len) {
        const ssize_t r = pread(STDIN_FILENO, j->dst + j->got, j->len - j->got, j->off + (off_t)j->got);
        if (r < 0) { if (errno == EINTR) continue; j->err = errno; break; }      // an I/O error is not the end of the file: reported by read_file_parallel
        if (r == 0) break;
        j->got += (size_t)r;
    }
}
// helper threads that live as long as the process (started at the first parallel read: a thread per block and slice cost more than it saved at 2-MiB blocks)
struct PreadPool {
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER; pthread_cond_t cv_go = PTHREAD_COND_INITIALIZER, cv_done = PTHREAD_COND_INITIALIZER;
    PreadJob *jobs = nullptr; int n_jobs = 0, next = 0, done = 0; unsigned long gen = 0; int n_threads = 0;
} g_pool;
void *pread_pool_main(void *)
{
    unsigned long seen = 0;
    pthread_mutex_lock(&g_pool.mu);
    for (;;) {
        while (g_pool.gen == seen || g_pool.next >= g_pool.n_jobs) { if (g_pool.gen != seen && g_pool.next >= g_pool.n_jobs) seen = g_pool.gen; pthread_cond_wait(&g_pool.cv_go, &g_pool.mu); }
        PreadJob *j = &g_pool.jobs[g_pool.next++];
        pthread_mutex_unlock(&g_pool.mu);
        pread_all(j);
        pthread_mutex_lock(&g_pool.mu);
        if (++g_pool.done == g_pool.n_jobs) pthread_cond_signal(&g_pool.cv_done);
    }
    return nullptr;
}
// up to max_bytes from the file at g_file_in.pos with several threads; a short slice = the end of the file (what lies behind it in later slices is not used)
void read_file_parallel(char *buf, size_t max_bytes, size_t *got, bool *eof)
{
    const int K = g_file_in.threads;
    PreadJob jobs[16];
    size_t slice = (max_bytes / K + 4095) & ~(size_t)4095; if (slice == 0) slice = max_bytes;
    int n = 0;
    for (size_t at = 0; at < max_bytes && n < K; at += slice, n++) jobs[n] = {buf + at, at + slice <= max_bytes ? slice : max_bytes - at, g_file_in.pos + (off_t)at, 0};
    if (g_pool.n_threads == 0) {                                      // (only this thread starts the pool)
        for (int i = 1; i < K; i++) { pthread_t t; if (pthread_create(&t, nullptr, pread_pool_main, nullptr) == 0) { pthread_detach(t); g_pool.n_threads++; } }
        if (g_pool.n_threads == 0) g_pool.n_threads = -1;             // no helpers: this thread reads every slice
    }
    if (g_pool.n_threads > 0 && n > 1) {
        pthread_mutex_lock(&g_pool.mu);
        g_pool.jobs = jobs; g_pool.n_jobs = n; g_pool.next = 1; g_pool.done = 1; g_pool.gen++;      // slice 0 is this thread's
        pthread_cond_broadcast(&g_pool.cv_go);
        pthread_mutex_unlock(&g_pool.mu);
        pread_all(&jobs[0]);
        pthread_mutex_lock(&g_pool.mu);
        while (g_pool.next < g_pool.n_jobs) {                         // (fewer helpers than slices, or none awake yet: take what is left)
            PreadJob *j = &g_pool.jobs[g_pool.next++];
            pthread_mutex_unlock(&g_pool.mu); pread_all(j); pthread_mutex_lock(&g_pool.mu);
            g_pool.done++;
        }
        while (g_pool.done < g_pool.n_jobs) pthread_cond_wait(&g_pool.cv_done, &g_pool.mu);
        g_pool.n_jobs = 0; g_pool.jobs = nullptr;
        pthread_mutex_unlock(&g_pool.mu);
    } else for (int i = 0; i < n; i++) pread_all(&jobs[i]);
    size_t have = 0; bool end = false;
    for (int i = 0; i < n && !end; i++) {
        have += jobs[i].got;
        if (jobs[i].err) {                                            // not an end of file: say so and fail, instead of a silently truncated stream with exit status 0
            fprintf(stderr, "csdr %s: read error on stdin at offset %lld (%s)\n", g_cmd, (long long)(jobs[i].off + (off_t)jobs[i].got), strerror(jobs[i].err));
            exit(5);
        }
        if (jobs[i].got < jobs[i].len) end = true;
    }
    g_file_in.pos += (off_t)have;
    (void)lseek(STDIN_FILENO, g_file_in.pos, SEEK_SET);               // (a later plain read() -- another command sharing the descriptor -- continues behind what was taken)
    *got = have; *eof = end;
}

// blocks until min_bytes have arrived (or EOF / error), then keeps reading only while more is immediately available
void read_some(char *buf, size_t min_bytes, size_t max_bytes, size_t *got, bool *eof)
{
    if (g_file_in.regular && max_bytes >= ((size_t)4 << 20)) { read_file_parallel(buf, max_bytes, got, eof); return; }
    size_t have = 0; *eof = false;
    while (have < max_bytes) {
        if (have >= min_bytes) {
            struct pollfd pf = {STDIN_FILENO, POLLIN, 0};
            if (poll(&pf, 1, 0) <= 0 || !(pf.revents & (POLLIN | POLLHUP))) break;
        }
        ssize_t r = read(STDIN_FILENO, buf + have, max_bytes - have);
        if (r < 0) { if (errno == EINTR) continue; fprintf(stderr, "csdr %s: read error on stdin (%s), treating it as the end of the stream\n", g_cmd, strerror(errno)); *eof = true; break; }
        if (r == 0) { *eof = true; break; }
        have += (size_t)r;
    }
    if (g_file_in.regular) g_file_in.pos += (off_t)have;              // (blocks below 1 MiB take this path on a regular file too)
    *got = have;
}
void *reader_main(void *arg)
{
    IoThreads *io = (IoThreads *)arg;
    (void)hipSetDevice(io->device);
    for (;;) {
        HostBuf *b = io->free_in.pop();
        if (b->pending) { (void)hipEventSynchronize(b->ev); b->pending = false; }      // its previous upload has left the buffer
        read_some(b->p, io->min_bytes, io->max_bytes, &b->bytes, &b->eof);
        const bool eof = b->eof;
        io->full_in.push(b);
        if (eof) return nullptr;
    }
}
void *writer_main(void *arg)
{
    IoThreads *io = (IoThreads *)arg;
    (void)hipSetDevice(io->device);
    for (;;) {
        HostBuf *b = io->full_out.pop();
        if (b->eof) return nullptr;
        if (b->pending) { (void)hipEventSynchronize(b->ev); b->pending = false; }
        size_t done = 0;
        while (done < b->bytes) {
            ssize_t r = write(STDOUT_FILENO, b->p + done, b->bytes - done);
            if (r < 0) { if (errno == EINTR) continue; _exit(128 + SIGPIPE); }      // downstream closed: end quietly, with the status a SIGPIPE death reports to the shell (141)
            done += (size_t)r;
        }
        io->free_out.push(b);
    }
}

// ---- device hand-off variants of the two I/O threads (+ one thread per direction for the credits)
void *reader_ipc_main(void *arg)
{
    IoThreads *io = (IoThreads *)arg;
    for (;;) {
        IpcToken t;
        const ssize_t r = recv(io->src->fd, &t, sizeof t, 0);
        const bool eof = r != (ssize_t)sizeof t || t.slot >= io->src->n_slots || t.bytes > io->src->slot_bytes;
        if (eof) {
            // The handing-over producer is done.  Whatever any OTHER or later writer of the same pipe sends ( `(csdr a; csdr a2) | csdr b`: a2 finds no listener and
            // writes bytes ) is still part of the stream: carry on in byte mode until stdin itself ends (pinned buffers are only allocated now).
            (void)hipSetDevice(io->device);
            {   // Every block of the ring has been copied out once all input buffers are back (credit_out_main returns a buffer only behind its copy's event):
                // tell the producer so NOW -- it stays alive until we close our side (the ring lives in its memory), and in `(csdr a; csdr a2) | csdr b` the
                // second writer only starts once the first has exited: waiting for stdin's EOF with the socket open would be a deadlock.
                std::vector<HostBuf *> all;
                for (int k = 0; k < io->n_in_bufs; k++) all.push_back(io->free_in.pop());
                shutdown(io->src->fd, SHUT_WR);
                for (HostBuf *b : all) io->free_in.push(b);
            }
            for (;;) {
                HostBuf *b = io->free_in.pop();
                if (b->pending) { (void)hipEventSynchronize(b->ev); b->pending = false; }
                b->dev = nullptr; b->slot = -1;
                if (!b->p && hipHostMalloc((void **)&b->p, b->cap, hipHostMallocDefault) != hipSuccess) { b->p = nullptr; b->bytes = 0; b->eof = true; io->full_in.push(b); return nullptr; }
                read_some(b->p, io->min_bytes, io->max_bytes, &b->bytes, &b->eof);
                const bool end = b->eof;
                io->full_in.push(b);
                if (end) return nullptr;
            }
        }
        size_t off = 0;
        do {                                                         // at most max_bytes per pass, like the byte reader: a token may be cut into several blocks
            HostBuf *b = io->free_in.pop();
            const size_t k = (size_t)t.bytes - off < io->max_bytes ? (size_t)t.bytes - off : io->max_bytes;
            b->dev = io->src->ring + (size_t)t.slot * io->src->slot_bytes + off; b->bytes = k; b->eof = false;
            off += k; b->slot = off == t.bytes ? (int)t.slot : -1;
            io->full_in.push(b);
        } while (off < t.bytes);
    }
}
void *credit_out_main(void *arg)
{   // consumer: a block's device copy has run -> its slot goes back to the producer
    IoThreads *io = (IoThreads *)arg;
    (void)hipSetDevice(io->device);
    for (;;) {
        HostBuf *b = io->copied_in.pop();
        if (b->eof) return nullptr;
        if (b->pending) { (void)hipEventSynchronize(b->ev); b->pending = false; }
        if (b->slot >= 0) { const int s = b->slot; (void)send(io->src->fd, &s, sizeof s, MSG_NOSIGNAL); }
        io->free_in.push(b);
    }
}
void *writer_ipc_main(void *arg)
{
    IoThreads *io = (IoThreads *)arg;
    (void)hipSetDevice(io->device);
    for (;;) {
        HostBuf *b = io->full_out.pop();
        if (b->eof) {
            // the last token is out: close OUR end of the pipe too, now -- the consumer carries on reading bytes from stdin after the hand-off's end (another writer of the
            // same pipe may follow), and without this it saw stdin's EOF only when this process had torn its HIP context down: the seven tear-downs of the README.md:66
            // pipeline ran one after the other (+0.3 s per run, tools/bench_cli.sh)
            // (descriptor 1 stays OCCUPIED -- /dev/null dup2'ed onto it: a bare close() would hand the number to the next open() / socket() of the runtime's tear-down,
            //  and a late write to stdout would land in an unrelated file: ADVICE r5)
            io->sink_closing.store(true); shutdown(io->sink_fd, SHUT_WR);
            { const int nul = open("/dev/null", O_WRONLY); if (nul >= 0) { (void)dup2(nul, STDOUT_FILENO); if (nul != STDOUT_FILENO) (void)close(nul); } else (void)close(STDOUT_FILENO); }
            return nullptr;
        }
        if (b->pending) { (void)hipEventSynchronize(b->ev); b->pending = false; }          // the kernels that filled the slot have run
        IpcToken t = {(unsigned)b->slot, 0u, (unsigned long long)b->bytes};
        if (send(io->sink_fd, &t, sizeof t, MSG_NOSIGNAL) != (ssize_t)sizeof t) _exit(128 + SIGPIPE);   // downstream closed: end quietly, with SIGPIPE's status
    }
}
void *credit_in_main(void *arg)
{   // producer: slots the consumer has copied out
    IoThreads *io = (IoThreads *)arg;
    for (;;) {
        int s = -1;
        const ssize_t r = recv(io->sink_fd, &s, sizeof s, 0);
        if (r < 0 && errno == EINTR) continue;
        // EOF after our own shutdown: the consumer is done.  EOF BEFORE it: the consumer died or left mid-stream (`| head`, its own error exit) -- with every ring
        // slot in flight the main thread sits in free_out.pop() and the writer has no token left to send, so nobody would ever see EPIPE: end quietly,
        // as SIGPIPE ends a producer that writes bytes.
        if (r == 0 && io->sink_closing.load()) return nullptr;
        if (r != (ssize_t)sizeof s || s < 0) _exit(128 + SIGPIPE);      // (a truncated pipeline does not report success: ADVICE r5)
        io->free_out.push(&io->sink_bufs[s]);
    }
}

struct Link { Stage *s; char *d_in[2] = {nullptr, nullptr}; char *d_stage = nullptr; int cur = 0; size_t cap_b = 0, have_b = 0; };   // byte counts: a pipe carries bytes,
                                                                                                               // the reader picks the element size
int run(csdr_amd_ctx *c, std::vector<Stage *> &stages, std::vector<size_t> &caps, int in_bufsize, int out_bufsize, int device)
{
    const size_t n_st = stages.size();
    file_input_init();                                               // (behind whatever the protocol's preamble has taken from stdin)
    std::vector<Link> L(n_st);
    Stage *first = stages[0], *last = stages[n_st - 1];
    const size_t block = caps[0];
    for (size_t k = 0; k < n_st; k++) {
        L[k].s = stages[k]; L[k].cap_b = (k == 0 ? 2 * block + 64 : caps[k]) * stages[k]->in_elem;      // stage 0: unconsumed tail + one new block
        for (int b = 0; b < 2; b++) { L[k].d_in[b] = (char *)csdr_amd_malloc(c, L[k].cap_b + 256); if (!L[k].d_in[b]) die("device buffers"); }
        if (k) { L[k].d_stage = (char *)csdr_amd_malloc(c, L[k].cap_b + 256); if (!L[k].d_stage) die("device buffers"); }
    }
    const size_t cap_out = last->out_capacity(caps[n_st - 1]) + 64;
    hipStream_t st = (hipStream_t)csdr_amd_ctx_stream(c);
    enum { NBUF = 3 };
    HostBuf hin[NBUF], hout[NBUF + 1];
    IoThreads io; io.device = device;
    size_t min_elems = (size_t)(in_bufsize > 0 ? in_bufsize : 1024);
    if (const char *e = getenv("CSDR_AMD_MIN_READ")) { long v = atol(e); if (v > 0) min_elems = (size_t)v; }
    if (min_elems % first->granule) min_elems += first->granule - min_elems % first->granule;
    if (min_elems > block) min_elems = block;
    io.min_bytes = min_elems * first->in_elem; io.max_bytes = block * first->in_elem;
    // device hand-off with the neighbours in the shell pipeline, where they are this binary too (see above): input side decided by now or here, output side offered here
    io.src = ipc_source_decide(device);
    char *ring_out = nullptr;
    const size_t slot_bytes = (cap_out * last->out_elem + 255) & ~(size_t)255;
    {
        struct stat so;
        if (ipc_enabled() && fstat(STDOUT_FILENO, &so) == 0 && S_ISFIFO(so.st_mode) && hipMalloc((void **)&ring_out, NBUF * slot_bytes) == hipSuccess) {
            io.sink_fd = ipc_sink_connect(device, ring_out, NBUF, slot_bytes);
            if (io.sink_fd < 0) { (void)hipFree(ring_out); ring_out = nullptr; }
        }
    }
    io.sink_bufs = hout;
    send_bufsize(out_bufsize);                                       // csdr.c:375-391, through the pipe itself in either mode
    for (int k = 0; k < NBUF; k++) {
        hin[k].cap = io.max_bytes + 64; hout[k].cap = cap_out * last->out_elem; hout[k].slot = k;
        if ((!io.src && hipHostMalloc((void **)&hin[k].p, hin[k].cap, hipHostMallocDefault) != hipSuccess) ||
            (io.sink_fd < 0 && hipHostMalloc((void **)&hout[k].p, hout[k].cap, hipHostMallocDefault) != hipSuccess) ||
            hipEventCreateWithFlags(&hin[k].ev, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&hout[k].ev, hipEventDisableTiming) != hipSuccess) {
            fprintf(stderr, "csdr %s: cannot allocate pinned host buffers\n", g_cmd); exit(3);
        }
        io.free_in.push(&hin[k]); io.free_out.push(&hout[k]);
    }
    io.n_in_bufs = NBUF;
#ifdef F_SETPIPE_SZ
    (void)fcntl(STDIN_FILENO, F_SETPIPE_SZ, 1 << 20); (void)fcntl(STDOUT_FILENO, F_SETPIPE_SZ, 1 << 20);      // fewer, larger pipe transfers (ignored for files)
#endif
    pthread_t th_r, th_w, th_ci, th_co;
    if (pthread_create(&th_r, nullptr, io.src ? reader_ipc_main : reader_main, &io) || pthread_create(&th_w, nullptr, io.sink_fd >= 0 ? writer_ipc_main : writer_main, &io) ||
        (io.src && pthread_create(&th_co, nullptr, credit_out_main, &io)) || (io.sink_fd >= 0 && pthread_create(&th_ci, nullptr, credit_in_main, &io))) {
        fprintf(stderr, "csdr %s: cannot start the I/O threads\n", g_cmd); exit(3);
    }
    void *d_out = csdr_amd_malloc(c, cap_out * last->out_elem + 256);
    if (!d_out) die("device buffers");
    // After EOF the pass is repeated (a few times at most) while some stage still consumes input: an operator that works through its input in
    // windows (fractional_decimator_ff) leaves a tail shorter than its window, which only the next call takes as the end of the stream.
    int extra_passes = 0, rc = 0;
    bool failed = false;
    // CSDR_AMD_CLI_TIMING=1: wall time per stage (the stream drained around every call: a diagnostic, it serialises the process), printed at the end
    const bool timing = getenv("CSDR_AMD_CLI_TIMING") != nullptr;
    std::vector<double> t_stage(n_st, 0.0); std::vector<size_t> n_stage(n_st, 0);
    for (bool eof = false, again = true; again && !failed;) {
        // a new input buffer only when the first operator cannot take a whole block from what it already holds: an operator that leaves a tail per pass
        // (fir_decimate_cc, fractional_decimator_ff, deemphasis_nfm_ff) otherwise let the carry grow by that tail every pass while full blocks kept arriving
        if (!eof && L[0].have_b < block * first->in_elem) {
            HostBuf *b = io.full_in.pop();
            eof = b->eof;
            Link &l0 = L[0];
            if (l0.have_b + b->bytes > l0.cap_b) { fprintf(stderr, "csdr %s: block of %zu elements is too small for this operator (raise CSDR_AMD_BLOCK)\n", g_cmd, block); rc = 1; break; }
            if (b->bytes) {
                if (hipMemcpyAsync(l0.d_in[l0.cur] + l0.have_b, b->dev ? (const void *)b->dev : (const void *)b->p, b->bytes, b->dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st) != hipSuccess ||
                    hipEventRecord(b->ev, st) != hipSuccess) die("upload");
                b->pending = true; l0.have_b += b->bytes;
            }
            if (!eof) (io.src ? io.copied_in : io.free_in).push(b);      // (hand-off: the slot goes back once the copy has run)
        }
        bool progressed = false;
        for (Stage *s : stages) if (s->ctl && s->ctl->fd && s->ctl_format()) { float a, b2; if (s->ctl->poll(s->ctl_format(), &a, &b2)) s->retune(c, a, b2); }
        // a pass hands the first operator at most `block` elements (what the operators were sized for); what is left waits for the next pass
        size_t n_in = L[0].have_b / first->in_elem;
        const bool all_of_it = n_in <= block;
        if (!all_of_it) n_in = block;                                 // block is a multiple of the granule
        if (!(eof && all_of_it && first->flush_partial)) n_in -= n_in % first->granule;
        if (n_in == 0 && !(eof && n_st > 1)) { again = !eof; continue; }   // at EOF a chain still flushes what its later stages carry
        const char *d_src = L[0].d_in[L[0].cur];
        long n_out = 0;
        for (size_t k = 0; k < n_st; k++) {
            Stage *s = stages[k];
            void *dst = d_out; size_t dst_cap = cap_out;
            if (k + 1 < n_st) {
                Link &nx = L[k + 1];
                // several element-wise kernels want 16-byte aligned pointers: when the consumer's carry is not a multiple of 16 bytes the
                // producer writes to an aligned staging buffer and the result is appended behind the carry by a device copy
                dst = (nx.have_b % 16 == 0) ? nx.d_in[nx.cur] + nx.have_b : nx.d_stage;
                dst_cap = (nx.cap_b - nx.have_b) / s->out_elem;
                if (nx.have_b % s->out_elem) { fprintf(stderr, "csdr chain: element sizes of \"%s\" and its consumer do not line up\n", g_cmd); failed = true; break; }
            }
            size_t consumed = 0;
            struct timespec tq0, tq1;
            if (timing) { (void)hipStreamSynchronize(st); clock_gettime(CLOCK_MONOTONIC, &tq0); }
            n_out = n_in ? s->process(c, d_src, n_in, dst, dst_cap, &consumed) : 0;
            if (timing) { (void)hipStreamSynchronize(st); clock_gettime(CLOCK_MONOTONIC, &tq1); t_stage[k] += (tq1.tv_sec - tq0.tv_sec) + 1e-9 * (tq1.tv_nsec - tq0.tv_nsec); n_stage[k] += n_in; }
            if (consumed) progressed = true;
            {
                Link &lk = L[k];
                if (consumed * s->in_elem > lk.have_b) consumed = lk.have_b / s->in_elem;
                if (k == 0 && consumed == 0 && lk.have_b >= block * s->in_elem && !eof) {
                    fprintf(stderr, "csdr %s: block of %zu elements is too small for this operator (raise CSDR_AMD_BLOCK)\n", g_cmd, block); failed = true; break; }
                const size_t rest_b = lk.have_b - consumed * s->in_elem;
                if (consumed) {
                    if (rest_b) MUST(csdr_amd_d2d(c, lk.d_in[lk.cur ^ 1], lk.d_in[lk.cur] + consumed * s->in_elem, rest_b));
                    lk.cur ^= 1; lk.have_b = rest_b;
                }
            }
            if (k + 1 == n_st) break;
            Link &nx = L[k + 1];
            if (n_out > 0 && nx.have_b % 16 != 0) MUST(csdr_amd_d2d(c, nx.d_in[nx.cur] + nx.have_b, nx.d_stage, (size_t)n_out * s->out_elem));
            nx.have_b += (n_out > 0 ? (size_t)n_out : 0) * s->out_elem;
            n_in = nx.have_b / stages[k + 1]->in_elem;
            if (!(eof && stages[k + 1]->flush_partial)) n_in -= n_in % stages[k + 1]->granule;
            d_src = nx.d_in[nx.cur];
        }
        if (failed) break;
        if (n_out > 0) {
            HostBuf *ob = io.free_out.pop();
            ob->bytes = (size_t)n_out * last->out_elem;
            if (ob->bytes > ob->cap) { fprintf(stderr, "csdr %s: output block larger than its staging buffer\n", g_cmd); io.free_out.push(ob); failed = true; break; }
            if (hipMemcpyAsync(ring_out ? (void *)(ring_out + (size_t)ob->slot * slot_bytes) : (void *)ob->p, d_out, ob->bytes, ring_out ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipEventRecord(ob->ev, st) != hipSuccess) die("download");
            ob->pending = true;
            // d_out is reused by the next pass: the download is ordered before the next kernels on the same stream
            io.full_out.push(ob);
        }
        bool leftover = false;
        for (size_t k = 0; k < n_st; k++) if (L[k].have_b) leftover = true;
        again = !eof || (progressed && leftover && extra_passes++ < 4);
    }
    // every exit goes through here: what was queued for the writer is written before the process ends
    if (failed) rc = 1;
    if (timing) for (size_t k = 0; k < n_st; k++) fprintf(stderr, "csdr %s: stage %zu: %.3f s for %zu input elements (%.1f M elements/s)\n", g_cmd, k, t_stage[k], n_stage[k], t_stage[k] > 0 ? n_stage[k] / t_stage[k] / 1e6 : 0.0);
    hout[NBUF].eof = true; io.full_out.push(&hout[NBUF]);
    pthread_join(th_w, nullptr);
    (void)hipStreamSynchronize(st);
    if (io.src) { HostBuf end; end.eof = true; io.copied_in.push(&end); pthread_join(th_co, nullptr); close(io.src->fd); }
    if (io.sink_fd >= 0) pthread_join(th_ci, nullptr);               // the ring lives in this process: stay until the consumer has copied the last slot out (it closes the socket then)
    return rc;
}

int passthrough(bool read_preamble, int send_size)
{   // setbuf / clone / through: plumbing commands pipelines use around the hot path (csdr.c:429-451, 2046-2082)
    int b = read_preamble ? get_bufsize(false) : 0;
    send_bufsize(send_size > 0 ? send_size : b);
    std::vector<char> buf(1 << 20);
    for (;;) { size_t got = 0; bool more = read_full(buf.data(), buf.size(), &got); if (got) write_full(buf.data(), got); if (!more) return 0; }
}


// ------------------------------------------------------------------ f4: the ddcd topology in one process
// ddcd runs `csdr fastddc_fwd_cc D | nmux` once and one `csdr fastddc_inv_cc --fd <ctl> D` per client (ddcd_old.cpp:238-252, 474-492):
// N processes re-reading the same spectrum.  Here: one forward transform per block and ONE multi-channel inverse call for all clients;
//   csdr fastddc_bank_cc <decimation> <transition_bw> <window> <ctl | -> <out_0> <shift_rate_0> [<out_1> <shift_rate_1> ...]
// out_k: a path (file or fifo) or fd:<n>;  ctl: a fifo path / fd:<n> carrying lines "<channel> <shift_rate>\n" (newest line per poll), or "-".
// Several GPUs (ddcd_old.cpp:238-252): start the SAME command line once per GPU with CSDR_AMD_RANK / CSDR_AMD_WORLD (and CSDR_AMD_DEVICE) set and
// CSDR_AMD_COMM_FILE naming a path all ranks can reach: rank 0 creates the library's RCCL communicator id there, reads the wideband stream from stdin and the control
// channel; every rank owns a block of the channels (csdr_amd_fastddc_bank_create_sharded) and writes only those outputs.  Per batch rank 0 broadcasts a small header
// (blocks, end of stream, retunes) so that all ranks make the same calls.
int run_bank(csdr_amd_ctx *c, int argc, char **argv, size_t block)
{
    if (argc < 8 || (argc - 6) % 2) return badsyntax("usage: fastddc_bank_cc <decimation> <transition_bw> <window> <ctl|-> <out_0> <rate_0> [<out_k> <rate_k> ...]");
    int D = 0; float tbw = 0.05f; sscanf(argv[2], "%d", &D); sscanf(argv[3], "%g", &tbw);
    const int window = window_from(argv[4]);
    const int n_ch = (argc - 6) / 2;
    int rank = 0, world = 1;
    if (const char *e = getenv("CSDR_AMD_WORLD")) world = atoi(e);
    if (const char *e = getenv("CSDR_AMD_RANK")) rank = atoi(e);
    const bool multi = getenv("CSDR_AMD_WORLD") != nullptr;            // (a world of 1 still goes through the communicator: the single-GPU test of this path)
    if (world < 1 || rank < 0 || rank >= world) return badsyntax("CSDR_AMD_RANK / CSDR_AMD_WORLD out of range");
    auto open_fd = [](const char *spec, int flags) { int fd = -1; if (!strncmp(spec, "fd:", 3)) sscanf(spec + 3, "%d", &fd); else fd = open(spec, flags, 0644); return fd; };
    Control ctl;
    if (rank == 0 && strcmp(argv[5], "-")) { ctl.fd = open_fd(argv[5], O_RDONLY | O_NONBLOCK); if (ctl.fd <= 0) return badsyntax("cannot open the control channel"); fcntl(ctl.fd, F_SETFL, fcntl(ctl.fd, F_GETFL, 0) | O_NONBLOCK); }
    std::vector<float> rates(n_ch);
    for (int k = 0; k < n_ch; k++) sscanf(argv[7 + 2 * k], "%g", &rates[k]);
    csdr_fastddc_t ddc;
    if (csdr_amd_fastddc_init(&ddc, tbw, D, 0)) return badsyntax("error in fastddc_init()");
    int nb_max = (int)(block / ddc.input_size); if (nb_max < 1) nb_max = 1;
    csdr_amd_comm *comm = nullptr; csdr_amd_fastddc_bank *bank = nullptr;
    int first = 0, count = n_ch;
    if (multi) {
        const char *cf = getenv("CSDR_AMD_COMM_FILE");
        if (!cf && world > 1) return badsyntax("CSDR_AMD_COMM_FILE must name a file every rank can reach");
        // CSDR_AMD_COMM=ipc: the ranks are processes on ONE box joined by unix sockets named after CSDR_AMD_COMM_FILE and HIP IPC (csdr_amd_comm_create_ipc) -- RCCL refuses
        // two ranks per device, so this is how the per-rank bootstrap of this command is exercised on a single GPU (tests/test_cli_gpu.py); default: RCCL over xGMI
        const char *ct = getenv("CSDR_AMD_COMM");
        const bool use_ipc = ct && !strcmp(ct, "ipc");
        char id[128];
        if (use_ipc) {
            if (!cf) return badsyntax("CSDR_AMD_COMM=ipc needs CSDR_AMD_COMM_FILE (the sockets' path prefix)");
            comm = csdr_amd_comm_create_ipc(c, cf, rank, world);
            if (!comm) die("communicator (ipc)");
        } else
        if (rank == 0) {
            if (csdr_amd_comm_unique_id(id)) die("communicator id");
            if (cf) { std::string tmp = std::string(cf) + ".tmp"; FILE *f = fopen(tmp.c_str(), "wb"); if (!f || fwrite(id, 1, 128, f) != 128) die("cannot write CSDR_AMD_COMM_FILE"); fclose(f); if (rename(tmp.c_str(), cf)) die("rename CSDR_AMD_COMM_FILE"); }
        } else {
            bool ok = false;
            for (int tries = 0; tries < 6000 && !ok; tries++) { FILE *f = fopen(cf, "rb"); if (f) { ok = fread(id, 1, 128, f) == 128; fclose(f); } if (!ok) usleep(10000); }
            if (!ok) die("timed out waiting for CSDR_AMD_COMM_FILE");
        }
        if (!use_ipc) comm = csdr_amd_comm_create(c, id, rank, world);
        if (!comm) die("communicator");
        // the schedule: the library's choice for this world size (channel shards up to two ranks, time slices beyond), or CSDR_AMD_SHARD=channels|blocks
        const char *sh = getenv("CSDR_AMD_SHARD");
        const int mode = (sh && !strcmp(sh, "blocks")) ? CSDR_AMD_SHARD_BLOCKS : (sh && !strcmp(sh, "channels")) ? CSDR_AMD_SHARD_CHANNELS : csdr_amd_fastddc_bank_default_shard_mode(world);
        bank = csdr_amd_fastddc_bank_create_sharded_by(c, tbw, D, rates.data(), n_ch, window, nb_max, comm, mode);
        if (bank) fprintf(stderr, "csdr fastddc_bank_cc: rank %d of %d, %s transport, schedule: %s\n", rank, world, use_ipc ? "ipc" : "rccl", mode == CSDR_AMD_SHARD_BLOCKS ? "time slices" : "channel shards");
        if (bank) csdr_amd_fastddc_bank_channel_slice(bank, &first, &count);
    } else bank = csdr_amd_fastddc_bank_create(c, tbw, D, rates.data(), n_ch, window, nb_max);
    if (!bank) die("fastddc_bank create");
    std::vector<int> out_fd(n_ch, -1);
    for (int k = first; k < first + count; k++) {                       // this rank's clients only
        out_fd[k] = open_fd(argv[6 + 2 * k], O_WRONLY | O_CREAT | O_TRUNC);
        if (out_fd[k] < 0) { fprintf(stderr, "csdr fastddc_bank_cc: cannot open output %s\n", argv[6 + 2 * k]); return -1; }
    }
    const size_t pitch = (size_t)csdr_amd_fastddc_bank_max_output(bank, nb_max) + 8;
    const size_t in_elems = (size_t)nb_max * ddc.input_size;
    csdr_complexf *h_in = nullptr, *h_out = nullptr;
    if (hipHostMalloc((void **)&h_in, in_elems * 8, hipHostMallocDefault) != hipSuccess || hipHostMalloc((void **)&h_out, (size_t)count * pitch * 8, hipHostMallocDefault) != hipSuccess) die("pinned buffers");
    csdr_complexf *d_in = (csdr_complexf *)csdr_amd_malloc(c, in_elems * 8 + 64);
    csdr_complexf *d_out = (csdr_complexf *)csdr_amd_malloc(c, (size_t)count * pitch * 8 + 64);
    // batch header, rank 0 -> all: {blocks, end of stream, retunes, (channel, rate bits) x up to 16}
    enum { HDR_INTS = 3 + 2 * 16 };
    int *h_hdr = nullptr; if (hipHostMalloc((void **)&h_hdr, HDR_INTS * sizeof(int), hipHostMallocDefault) != hipSuccess) die("pinned header");
    int *d_hdr = (int *)csdr_amd_malloc(c, HDR_INTS * sizeof(int) + 64);
    if (!d_in || !d_out || !d_hdr) die("device buffers");
    std::vector<int> counts(count);
    fprintf(stderr, "csdr fastddc_bank_cc: %d channels%s, fft_size = %d, input_size = %d, %d blocks per call\n", n_ch, multi ? " (sharded)" : "", ddc.fft_size, ddc.input_size, nb_max);
    if (multi) fprintf(stderr, "csdr fastddc_bank_cc: rank %d of %d serves channels %d .. %d\n", rank, world, first, first + count - 1);
    size_t have = 0;
    for (bool eof = false; !eof;) {
        int nb = 0, n_ret = 0; int ret_ch[16]; float ret_rate[16];
        if (rank == 0) {
            size_t got = 0;
            if (!read_full((char *)h_in + have * 8, (in_elems - have) * 8, &got)) eof = true;
            have += got / 8;
            if (ctl.fd) {
                // every complete line since the last poll is applied (several clients may retune between two blocks)
                const ssize_t r = read(ctl.fd, ctl.buf + ctl.fill, sizeof(ctl.buf) - 1 - ctl.fill);
                if (r > 0) ctl.fill += (int)r;
                // at most 16 retunes travel in one batch header: further complete lines stay in the buffer for the next batch (none is dropped)
                int start = 0;
                for (int i = 0; i < ctl.fill && n_ret < 16; i++) if (ctl.buf[i] == '\n') {
                    ctl.buf[i] = 0; int ch = -1; float rate = 0;
                    if (sscanf(ctl.buf + start, "%d %g", &ch, &rate) == 2 && ch >= 0 && ch < n_ch) { ret_ch[n_ret] = ch; ret_rate[n_ret] = rate; n_ret++; }
                    start = i + 1;
                }
                if (start) { memmove(ctl.buf, ctl.buf + start, ctl.fill - start); ctl.fill -= start; }
                else if (ctl.fill >= (int)sizeof(ctl.buf) - 1) ctl.fill = 0;      // an over-long line without a newline: discard it
            }
            nb = (int)(have / ddc.input_size);
        }
        if (multi) {
            if (rank == 0) {
                h_hdr[0] = nb; h_hdr[1] = eof ? 1 : 0; h_hdr[2] = n_ret;
                for (int i = 0; i < n_ret; i++) { h_hdr[3 + 2 * i] = ret_ch[i]; memcpy(&h_hdr[4 + 2 * i], &ret_rate[i], 4); }
                MUST(csdr_amd_h2d(c, d_hdr, h_hdr, HDR_INTS * sizeof(int)));
            }
            MUST(csdr_amd_comm_broadcast(comm, d_hdr, HDR_INTS * sizeof(int), 0));
            MUST(csdr_amd_d2h(c, h_hdr, d_hdr, HDR_INTS * sizeof(int)));
            nb = h_hdr[0]; eof = h_hdr[1] != 0; n_ret = h_hdr[2];
            for (int i = 0; i < n_ret; i++) { ret_ch[i] = h_hdr[3 + 2 * i]; memcpy(&ret_rate[i], &h_hdr[4 + 2 * i], 4); }
        }
        for (int i = 0; i < n_ret; i++) {
            MUST(csdr_amd_fastddc_bank_set_rate_global(bank, ret_ch[i], ret_rate[i]));      // every rank makes the call; a rank applies it to what it computes
            if (ret_ch[i] >= first && ret_ch[i] < first + count) fprintf(stderr, "csdr fastddc_bank_cc: channel %d retuned to %g\n", ret_ch[i], ret_rate[i]);
        }
        if (nb == 0) continue;
        const size_t used = (size_t)nb * ddc.input_size;
        if (rank == 0) MUST(csdr_amd_h2d(c, d_in, h_in, used * 8));
        MUST(csdr_amd_fastddc_bank_process(bank, d_in, nb, d_out, pitch, counts.data()));
        MUST(csdr_amd_d2h(c, h_out, d_out, (size_t)count * pitch * 8));
        for (int k = 0; k < count; k++) {
            size_t done = 0; const size_t bytes = (size_t)counts[k] * 8; const char *src = (const char *)(h_out + (size_t)k * pitch);
            while (done < bytes) { ssize_t r = write(out_fd[first + k], src + done, bytes - done); if (r < 0) { if (errno == EINTR) continue; break; } done += (size_t)r; }
        }
        if (rank == 0) { memmove(h_in, h_in + used, (have - used) * 8); have -= used; }
    }
    for (int k = first; k < first + count; k++) close(out_fd[k]);
    csdr_amd_fastddc_bank_destroy(bank);
    if (comm) csdr_amd_comm_destroy(comm);
    return 0;
}

// ------------------------------------------------------------------ f4, first half: N-stream host ingest into the batch API
// nmux / ddcd fan one source out to N clients, each client = one `csdr ... | csdr ...` pipeline of processes (nmux.cpp:177-283, ddcd_old.cpp:474-492).
// The device batch API wants the opposite shape: N streams side by side in ONE call.  These commands are that producer:
//   csdr wfm_bank_u8_s16 <shift_rate> <in_0> <out_0> [<in_1> <out_1> ...]      N u8 IQ streams -> N s16 audio streams through ONE fused WFM chain object
//   csdr nfm_bank_u8_s16 <shift_rate> <in_0> <out_0> [<in_1> <out_1> ...]      the same through the NFM chain object (README.md:87 defaults)
// in_k / out_k: a path (file or fifo) or fd:<n>.  Every pass reads one block of CSDR_AMD_BANK_BLOCK samples (default 262144, a multiple of 1024) from
// EVERY input (the streams advance in lockstep, like the clients of one nmux), uploads them as the rows of one batch, runs the chain once and writes
// each row's audio to its output.  The pass in which the first stream ends is the last one (lockstep streams end together).
//   <shift_rate> may be a comma-separated list, one rate per stream (ddcd tunes every client on its own: ddcd_old.h:51-61);
//   --ctl <fifo | fd:<n>> in front of it: control lines "<stream> <rate>\n", applied between two passes exactly as `shift_addition_cc --fifo` applies a new rate
//   between two reads (csdr.c:881-923: the phase carries over).
int run_stream_bank(csdr_amd_ctx *c, int argc, char **argv, bool nfm)
{
    int ctl_fd = -1;
    if (argc > 3 && !strcmp(argv[2], "--ctl")) {
        if (!strncmp(argv[3], "fd:", 3)) sscanf(argv[3] + 3, "%d", &ctl_fd); else ctl_fd = open(argv[3], O_RDONLY | O_NONBLOCK);
        if (ctl_fd < 0) { fprintf(stderr, "csdr %s: cannot open the control channel %s\n", g_cmd, argv[3]); return -1; }
        fcntl(ctl_fd, F_SETFL, fcntl(ctl_fd, F_GETFL, 0) | O_NONBLOCK);
        argv += 2; argc -= 2;
    }
    if (argc < 5 || (argc - 3) % 2) return badsyntax("usage: [--ctl <fifo|fd:n>] <shift_rate[,rate_1,...]> <in_0> <out_0> [<in_k> <out_k> ...]   (paths, fifos or fd:<n>)");
    const int S = (argc - 3) / 2;
    std::vector<float> rates;
    for (const char *q = argv[2]; *q;) { char *end = nullptr; const float v = strtof(q, &end); if (end == q) return badsyntax("shift_rate must be a number or a comma-separated list"); rates.push_back(v); q = *end == ',' ? end + 1 : end; if (*end && *end != ',') return badsyntax("shift_rate must be a number or a comma-separated list"); }
    if (rates.size() != 1 && (int)rates.size() != S) return badsyntax("as many shift rates as streams (or one for all)");
    // (fewer than 16 streams: the rate-per-stream object also when they share one rate -- its kernel fills all 16 columns of a tile with time segments of ONE stream,
    // the shared-rate kernel needs 16 streams to fill them)
    const bool per_stream = rates.size() > 1 || ctl_fd >= 0 || (S < 16 && !getenv("CSDR_AMD_CLI_SHARED"));
    if (per_stream && rates.size() == 1) rates.assign(S, rates[0]);
    const float shift = rates[0];
    auto open_fd = [](const char *spec, int flags) { int fd = -1; if (!strncmp(spec, "fd:", 3)) sscanf(spec + 3, "%d", &fd); else fd = open(spec, flags, 0644); return fd; };
    std::vector<int> in_fd(S), out_fd(S);
    for (int k = 0; k < S; k++) {
        in_fd[k] = open_fd(argv[3 + 2 * k], O_RDONLY); out_fd[k] = open_fd(argv[4 + 2 * k], O_WRONLY | O_CREAT | O_TRUNC);
        if (in_fd[k] < 0 || out_fd[k] < 0) { fprintf(stderr, "csdr %s: cannot open %s / %s\n", g_cmd, argv[3 + 2 * k], argv[4 + 2 * k]); return -1; }
    }
    size_t T = 262144; if (const char *e = getenv("CSDR_AMD_BANK_BLOCK")) { long v = atol(e); if (v >= 1024) T = (size_t)v; }
    T -= T % 1024;
    const int D = nfm ? 50 : 10; const float tbw = nfm ? 0.005f : 0.05f;
    const int nt = csdr_amd_firdes_filter_len(tbw);
    std::vector<float> taps(nt); csdr_amd_firdes_lowpass_f(taps.data(), nt, 0.5f / (float)D, CSDR_WINDOW_HAMMING);
    csdr_amd_wfm *w = nullptr; csdr_amd_nfm *n = nullptr;
    if (nfm && per_stream) n = csdr_amd_nfm_create_rates(c, S, rates.data(), D, taps.data(), nt, 48000, 1024, 1.0f, 1.0f, T);
    else if (nfm) n = csdr_amd_nfm_create(c, S, shift, D, taps.data(), nt, 48000, 1024, 1.0f, 1.0f, T);
    else if (per_stream) w = csdr_amd_wfm_create_rates(c, S, rates.data(), D, taps.data(), nt, 5, 50e-6f, 48000, T);
    else w = csdr_amd_wfm_create(c, S, shift, D, taps.data(), nt, 5, 50e-6f, 48000, T);
    std::string ctl_buf;
    if (!w && !n) die("bank create");
    const size_t in_pitch = 2 * T, out_pitch = ((T / 50 + 4096 + 63) / 64) * 64;
    uint8_t *h_in = nullptr; int16_t *h_out = nullptr;
    if (hipHostMalloc((void **)&h_in, (size_t)S * in_pitch, hipHostMallocDefault) != hipSuccess || hipHostMalloc((void **)&h_out, (size_t)S * out_pitch * 2, hipHostMallocDefault) != hipSuccess) die("pinned buffers");
    uint8_t *d_in = (uint8_t *)csdr_amd_malloc(c, (size_t)S * in_pitch + 256); int16_t *d_out = (int16_t *)csdr_amd_malloc(c, (size_t)S * out_pitch * 2 + 256);
    if (!d_in || !d_out) die("device buffers");
    fprintf(stderr, "csdr %s: %d streams, %zu samples per stream and pass\n", g_cmd, S, T);
    std::vector<bool> alive(S, true);
    for (int n_alive = S; n_alive > 0;) {
        size_t got_min = T; bool any = false;
        for (int k = 0; k < S; k++) {
            if (!alive[k]) { memset(h_in + (size_t)k * in_pitch, 0x80, in_pitch); continue; }
            size_t have = 0;
            while (have < in_pitch) { ssize_t r = read(in_fd[k], h_in + (size_t)k * in_pitch + have, in_pitch - have); if (r < 0 && errno == EINTR) continue; if (r <= 0) break; have += (size_t)r; }
            const size_t samples = have / 2;
            if (samples < T) { alive[k] = false; n_alive--; memset(h_in + (size_t)k * in_pitch + have, 0x80, in_pitch - have); }
            if (samples) any = true;
            if (samples && samples < got_min) got_min = samples;
        }
        if (!any) break;
        // retunes that have arrived: complete lines only, the rest waits for the next pass
        if (ctl_fd >= 0) {
            char tmp[1024]; ssize_t r;
            while ((r = read(ctl_fd, tmp, sizeof tmp)) > 0) ctl_buf.append(tmp, (size_t)r);
            size_t nl;
            while ((nl = ctl_buf.find('\n')) != std::string::npos) {
                int st = -1; float rv = 0;
                if (sscanf(ctl_buf.c_str(), "%d %g", &st, &rv) == 2 && st >= 0 && st < S) { MUST(nfm ? csdr_amd_nfm_set_rate(n, st, rv) : csdr_amd_wfm_set_rate(w, st, rv)); fprintf(stderr, "csdr %s: stream %d reinitialized to %g\n", g_cmd, st, rv); }
                ctl_buf.erase(0, nl + 1);
            }
        }
        // a short final block: whole 1024-sample chunks of the shortest live stream (the chain objects take a ragged LAST block only)
        size_t nproc = got_min < T ? got_min : T;
        MUST(csdr_amd_h2d(c, d_in, h_in, (size_t)S * in_pitch));
        long na = nfm ? csdr_amd_nfm_process(n, d_in, in_pitch, nproc, d_out, nullptr, out_pitch) : csdr_amd_wfm_process(w, d_in, in_pitch, nproc, d_out, nullptr, out_pitch);
        MUST(na);
        if (na > 0) {
            MUST(csdr_amd_d2h(c, h_out, d_out, (size_t)S * out_pitch * 2));
            for (int k = 0; k < S; k++) {
                if (out_fd[k] < 0) continue;
                size_t done = 0; const size_t bytes = (size_t)na * 2; const char *src = (const char *)(h_out + (size_t)k * out_pitch);
                while (done < bytes) { ssize_t r = write(out_fd[k], src + done, bytes - done); if (r < 0) { if (errno == EINTR) continue; close(out_fd[k]); out_fd[k] = -1; break; } done += (size_t)r; }
            }
        }
        if (nproc < T) break;                                        // ragged block = the end of the lockstep streams
    }
    for (int k = 0; k < S; k++) { if (out_fd[k] >= 0) close(out_fd[k]); close(in_fd[k]); }
    return 0;
}

// Build the operator for one command line.  `block` = the largest input this stage will be handed in one call.
// ctl: opened when the command line carries --fifo/--fd (single-command mode only).  Returns nullptr after printing why.
Stage *make_stage(csdr_amd_ctx *c, int argc, char **argv, size_t block, Control *ctl, int the_bufsize)
{
    g_cmd = argv[1];
    const std::string cmd = argv[1];
    const bool has_ctl = ctl && ctl->open_from(argc, argv);
    if (cmd == "convert_u8_f") return new Convert(0, 1, 4);
    if (cmd == "convert_f_u8") return new Convert(1, 4, 1);
    if (cmd == "convert_s8_f") return new Convert(2, 1, 4);
    if (cmd == "convert_f_s8") return new Convert(3, 4, 1);
    if (cmd == "convert_f_s16" || cmd == "convert_f_i16") return new Convert(4, 4, 2);
    if (cmd == "convert_s16_f" || cmd == "convert_i16_f") return new Convert(5, 2, 4);
    if (cmd == "convert_f_s24") { Convert *cv = new Convert(6, 4, 3); cv->bigendian = argc > 2 && !strcmp(argv[2], "--bigendian"); cv->granule = 4; return cv; }
    if (cmd == "convert_s24_f") { Convert *cv = new Convert(7, 3, 4); cv->bigendian = argc > 2 && !strcmp(argv[2], "--bigendian"); cv->granule = 4; return cv; }
    if (cmd == "shift_math_cc" || cmd == "shift_addition_cc" || cmd == "shift_table_cc" || cmd == "shift_addfast_cc" || cmd == "shift_unroll_cc" || cmd == "shift_addition_fc") {
        float rate = 0;
        const bool ctl_cmd = cmd == "shift_addition_cc" || cmd == "shift_addition_fc" || cmd == "shift_addfast_cc" || cmd == "shift_unroll_cc";
        if (has_ctl && ctl_cmd) { float d; ctl->wait_first("%g\n", &rate, &d); }
        else { if (argc <= 2) { badsyntax("need required parameter (rate)"); return nullptr; } sscanf(argv[2], "%g", &rate); }
        int variant = CSDR_SHIFT_ADDITION, aux = 0;
        if (cmd == "shift_math_cc") variant = CSDR_SHIFT_MATH;
        else if (cmd == "shift_table_cc") { variant = CSDR_SHIFT_TABLE; aux = 65536; if (argc > 3) sscanf(argv[3], "%d", &aux); }       // csdr.c:731
        else if (cmd == "shift_addfast_cc") variant = CSDR_SHIFT_ADDFAST;
        else if (cmd == "shift_unroll_cc") { variant = CSDR_SHIFT_UNROLL; aux = 1024; }                                                  // csdr.c:821
        Shift *sh = new Shift(variant, rate, aux);
        if (cmd == "shift_addition_fc") { sh->real_in = true; sh->in_elem = 4; }
        return sh;
    }
    if (cmd == "decimating_shift_addition_cc") {
        if (argc <= 2) { badsyntax("need required parameter (rate)"); return nullptr; }
        float rate; int dec = 1; sscanf(argv[2], "%g", &rate); if (argc > 3) sscanf(argv[3], "%d", &dec);
        if (dec < 1) { badsyntax("decimation must be >= 1"); return nullptr; }
        return new DecimatingShift(c, rate, dec, the_bufsize);
    }
    if (cmd == "fir_decimate_cc") {
        if (argc <= 2) { badsyntax("need required parameter (decimation factor)"); return nullptr; }
        int factor = 0;
        if (sscanf(argv[2], "%d", &factor) != 1 || factor < 1) { badsyntax("decimation factor must be an integer >= 1"); return nullptr; }
        float tbw = 0.05f; if (argc >= 4) sscanf(argv[3], "%g", &tbw);
        if (!(tbw > 0)) { badsyntax("transition_bw must be positive"); return nullptr; }
        int window = CSDR_WINDOW_HAMMING; if (argc >= 5) window = window_from(argv[4]); else fprintf(stderr, "csdr fir_decimate_cc: window = HAMMING\n");
        return new FirDecimate(c, factor, tbw, window);
    }
    if (cmd == "fmdemod_quadri_cf" || cmd == "fmdemod_quadri_novect_cf") return new Fmdemod(c);
    if (cmd == "limit_ff") { float m = 1.0f; if (argc >= 3) sscanf(argv[2], "%g", &m); return new Limit(m); }
    if (cmd == "deemphasis_wfm_ff") {
        if (argc <= 3) { badsyntax("need required parameters (sample rate, tau)"); return nullptr; }
        int rate; float tau; sscanf(argv[2], "%d", &rate); sscanf(argv[3], "%g", &tau);
        fprintf(stderr, "csdr deemphasis_wfm_ff: tau = %g, sample_rate = %d\n", tau, rate);
        return new DeemphWfm(c, rate, tau);
    }
    if (cmd == "deemphasis_nfm_ff") { if (argc <= 2) { badsyntax("need required parameter (sample rate)"); return nullptr; } int rate; sscanf(argv[2], "%d", &rate); return new DeemphNfm(c, rate, g_dynamic ? the_bufsize : unitround(g_fixed)); }   // without the preamble protocol every reference process has its default buffer
    if (cmd == "fastagc_ff") { int b = 1024; float ref = 1.0f; if (argc >= 3) sscanf(argv[2], "%d", &b); if (argc >= 4) sscanf(argv[3], "%g", &ref); if (b <= 0) { badsyntax("block size must be positive"); return nullptr; } return new FastAgc(c, b, ref); }
    if (cmd == "fractional_decimator_ff") {
        if (argc <= 2) { badsyntax("need required parameters (rate)"); return nullptr; }
        float rate; sscanf(argv[2], "%g", &rate);
        int points = 12; if (argc >= 4) sscanf(argv[3], "%d", &points);
        if (points & 1) { badsyntax("num_poly_points should be even"); return nullptr; }
        if (points < 2) { badsyntax("num_poly_points should be >= 2"); return nullptr; }
        std::vector<float> taps;
        if (argc >= 5 && !strcmp(argv[4], "--prefilter")) {                          // csdr.c:1481-1486, 1499-1507: only --prefilter enables it
            const float tbw = 0.03f;
            const int nt = csdr_amd_firdes_filter_len(tbw); taps.resize(nt);
            csdr_amd_firdes_lowpass_f(taps.data(), nt, 0.5f / (rate - tbw), CSDR_WINDOW_HAMMING);
        }
        return new FracDec(rate, points, taps.empty() ? nullptr : taps.data(), (int)taps.size(), g_dynamic ? the_bufsize : unitround(g_fixed));
    }
    if (cmd == "bandpass_fir_fft_cc") {
        float lo = 0, hi = 0, tbw = 0;
        if (has_ctl) { ctl->wait_first("%g %g\n", &lo, &hi); if (argc <= 4) { badsyntax("need more required parameters (transition_bw)"); return nullptr; } }
        else { if (argc <= 4) { badsyntax("need required parameters (low_cut, high_cut, transition_bw)"); return nullptr; } sscanf(argv[2], "%g", &lo); sscanf(argv[3], "%g", &hi); }
        sscanf(argv[4], "%g", &tbw);
        return new Bandpass(c, lo, hi, tbw, argc >= 6 ? window_from(argv[5]) : CSDR_WINDOW_HAMMING, block);
    }
    if (cmd == "fastddc_fwd_cc") {
        if (argc <= 2) { badsyntax("need required parameter (decimation)"); return nullptr; }
        int D; sscanf(argv[2], "%d", &D); float tbw = 0.05f; if (argc > 3) sscanf(argv[3], "%g", &tbw);
        return new DdcFwd(c, D, tbw, block);
    }
    if (cmd == "fastddc_inv_cc") {
        float shift = 0; int plus = 0;
        if (has_ctl) { float d; ctl->wait_first("%g\n", &shift, &d); plus = 1; }
        else { if (argc <= 2) { badsyntax("need required parameter (rate)"); return nullptr; } sscanf(argv[2], "%g", &shift); }
        if (argc <= 3 + plus) { badsyntax("need required parameter (decimation)"); return nullptr; }
        int D; sscanf(argv[3 + plus], "%d", &D);
        float tbw = 0.05f; if (argc > 4 + plus) sscanf(argv[4 + plus], "%g", &tbw);
        return new DdcInv(c, shift, D, tbw, argc > 5 + plus ? window_from(argv[5 + plus]) : CSDR_WINDOW_HAMMING, block);
    }
    if (cmd == "amdemod_cf") return new CfToF(0, 0);
    if (cmd == "amdemod_estimator_cf") return new CfToF(1, 0);
    if (cmd == "realpart_cf") return new CfToF(2, 0);
    if (cmd == "logpower_cf") { float add_db = 0; if (argc >= 3) sscanf(argv[2], "%g", &add_db); return new CfToF(3, add_db); }
    if (cmd == "gain_ff") { if (argc <= 2) { badsyntax("need required parameter (gain)"); return nullptr; } float g; sscanf(argv[2], "%g", &g); return new Gain(g); }
    if (cmd == "fmdemod_atan_cf") return new FmdemodAtan(c);
    if (cmd == "dcblock_ff") return new DcBlock(c);
    if (cmd == "fastdcblock_ff") { int b = 1024; if (argc >= 3) sscanf(argv[2], "%d", &b); if (b <= 0) { badsyntax("block size must be positive"); return nullptr; } return new FastDcBlock(c, b); }
    if (cmd == "agc_ff") {   // defaults csdr.c:1343-1361
        Agc *a = new Agc(c, the_bufsize);
        a->hang = 200; a->ref = 0.2f; a->attack = 0.01f; a->decay = 0.0001f; a->maxg = 65536; a->wait = 0; a->alpha = 0.999f;
        if (argc >= 3) sscanf(argv[2], "%hd", &a->hang);
        if (argc >= 4) sscanf(argv[3], "%g", &a->ref);
        if (argc >= 5) sscanf(argv[4], "%g", &a->attack);
        if (argc >= 6) sscanf(argv[5], "%g", &a->decay);
        if (argc >= 7) sscanf(argv[6], "%g", &a->maxg);
        if (argc >= 8) sscanf(argv[7], "%hd", &a->wait);
        if (argc >= 9) sscanf(argv[8], "%g", &a->alpha);
        return a;
    }
    if (cmd == "fft_cc") {
        if (argc <= 3) { badsyntax("need required parameters (fft_size, out_of_every_n_samples)"); return nullptr; }
        int fft, every; sscanf(argv[2], "%d", &fft); sscanf(argv[3], "%d", &every);
        if (csdr_amd_log2n(fft) == -1) { badsyntax("fft_size should be power of 2"); return nullptr; }
        if (every <= 0) { badsyntax("out_of_every_n_samples must be positive"); return nullptr; }
        if (argc >= 6 && !strcmp(argv[5], "--octave")) { badsyntax("--octave text output is not offered by the MI355X back end"); return nullptr; }
        return new FftCc(c, fft, every, argc >= 5 ? window_from(argv[4]) : CSDR_WINDOW_HAMMING, block);
    }
    if (cmd == "encode_ima_adpcm_i16_u8" || cmd == "encode_ima_adpcm_s16_u8") return new AdpcmEnc(c);
    if (cmd == "decode_ima_adpcm_u8_i16" || cmd == "decode_ima_adpcm_u8_s16") return new AdpcmDec(c);
    if (cmd == "compress_fft_adpcm_f_u8") {
        if (argc <= 2) { badsyntax("need required parameters (fft_size)"); return nullptr; }
        int fft; sscanf(argv[2], "%d", &fft);
        if (fft <= 0 || (fft & 1)) { badsyntax("fft_size must be positive and even"); return nullptr; }
        return new CompressFft(fft);
    }
    // the fused commands: `--fifo <path>` / `--fd <n>` stand where the shift rate stands, as in shift_addition_cc (csdr.c:881-893); the first rate is waited for
    if (cmd == "ddc_u8_cc" || cmd == "nfm_chain_u8_s16" || cmd == "wfm_chain_u8_s16") {
        float shift = 0;
        int a = 3;                                                   // argv index of the first argument behind the rate
        if (has_ctl) { float d; ctl->wait_first("%g\n", &shift, &d); a = 4; }
        else if (argc > 2) sscanf(argv[2], "%g", &shift);
        else if (cmd == "ddc_u8_cc") { badsyntax("need required parameters (shift rate, decimation factor)"); return nullptr; }
        if (cmd == "ddc_u8_cc") {
            if (argc <= a) { badsyntax("need required parameters (shift rate, decimation factor)"); return nullptr; }
            float tbw = 0.05f; int factor = 0; sscanf(argv[a], "%d", &factor);
            if (factor < 1) { badsyntax("decimation factor must be >= 1"); return nullptr; }
            if (argc > a + 1) sscanf(argv[a + 1], "%g", &tbw);
            const int window = argc > a + 2 ? window_from(argv[a + 2]) : CSDR_WINDOW_HAMMING;
            return new DdcFront(c, shift, factor, tbw, window, block, has_ctl);
        }
        if (cmd == "nfm_chain_u8_s16") {
            float tbw = 0.005f; int factor = 50;
            if (argc > a) sscanf(argv[a], "%d", &factor);
            if (argc > a + 1) sscanf(argv[a + 1], "%g", &tbw);
            return new NfmChain(c, shift, factor, tbw, block, has_ctl);
        }
        { const char *rs = getenv("CSDR_AMD_RESIDENT"); if (rs && atoi(rs)) return new WfmRingStage(c, shift, has_ctl); }
        return new WfmChain(c, shift, block, has_ctl);
    }
    fprintf(stderr, "csdr: function \"%s\" is not part of the MI355X hot path (see --help)\n", argv[1]);
    return nullptr;
}

// "a b c | d e" -> {{"csdr","a","b","c"},{"csdr","d","e"}}
std::vector<std::vector<std::string>> split_chain(const char *spec)
{
    std::vector<std::vector<std::string>> out(1, std::vector<std::string>(1, "csdr"));
    std::string tok;
    auto flush = [&]() { if (!tok.empty()) { if (tok != "csdr" || out.back().size() > 1) out.back().push_back(tok); tok.clear(); } };
    for (const char *p = spec; *p; p++) {
        if (*p == '|') { flush(); out.push_back(std::vector<std::string>(1, "csdr")); }
        else if (*p == ' ' || *p == '\t' || *p == '\n') flush();
        else tok.push_back(*p);
    }
    flush();
    return out;
}

// shift_addition_cc <rate> | shift_addition_cc --fifo <path> | shift_addition_cc --fd <n>: the tokens that stand for the rate (kept as they are in the fused command)
bool shift_rate_args(const std::vector<std::string> &cmd, std::vector<std::string> *rate_args)
{
    if (cmd.size() < 3 || cmd[1] != "shift_addition_cc") return false;
    if (cmd.size() == 4 && (cmd[2] == "--fifo" || cmd[2] == "--fd")) { rate_args->assign(cmd.begin() + 2, cmd.end()); return true; }
    float r;
    if (cmd.size() == 3 && sscanf(cmd[2].c_str(), "%g", &r) == 1) { char sh[64]; snprintf(sh, sizeof sh, "%.9g", r); rate_args->assign(1, sh); return true; }
    return false;
}

bool is_wfm_pattern(const std::vector<std::vector<std::string>> &cmds, std::vector<std::string> *rate_args)
{   // README.md:66 exactly: the shape the fused matrix-core kernel implements
    if (cmds.size() != 7) return false;
    auto is = [&](size_t k, std::initializer_list<const char *> want) {
        if (cmds[k].size() != want.size() + 1) return false;
        size_t j = 1; for (const char *w : want) { if (w[0] != '*' && cmds[k][j] != w) return false; j++; }
        return true;
    };
    if (!is(0, {"convert_u8_f"}) || !is(2, {"fir_decimate_cc", "10", "0.05", "HAMMING"}) || !is(3, {"fmdemod_quadri_cf"}) ||
        !is(4, {"fractional_decimator_ff", "5"}) || !is(5, {"deemphasis_wfm_ff", "48000", "50e-6"}) || !is(6, {"convert_f_s16"})) return false;
    return shift_rate_args(cmds[1], rate_args);
}

bool is_nfm_pattern(const std::vector<std::vector<std::string>> &cmds, std::vector<std::string> *rate_args)
{   // README.md:87 exactly: the shape csdr_amd_nfm implements
    if (cmds.size() != 8) return false;
    auto is = [&](size_t k, std::initializer_list<const char *> want) {
        if (cmds[k].size() != want.size() + 1) return false;
        size_t j = 1; for (const char *w : want) { if (w[0] != '*' && cmds[k][j] != w) return false; j++; }
        return true;
    };
    if (!is(0, {"convert_u8_f"}) || !is(2, {"fir_decimate_cc", "50", "0.005", "HAMMING"}) || !is(3, {"fmdemod_quadri_cf"}) ||
        !is(4, {"limit_ff"}) || !is(5, {"deemphasis_nfm_ff", "48000"}) || !is(6, {"fastagc_ff"}) || !is(7, {"convert_f_s16"})) return false;
    return shift_rate_args(cmds[1], rate_args);
}

// convert_u8_f | shift_addition_cc r | fir_decimate_cc D [tbw [window]] at the head of a chain -> one ddc_u8_cc command
bool fuse_front_end(std::vector<std::vector<std::string>> &cmds)
{
    if (cmds.size() < 3 || cmds[0].size() != 2 || cmds[0][1] != "convert_u8_f") return false;
    std::vector<std::string> rate_args;
    if (!shift_rate_args(cmds[1], &rate_args)) return false;
    if (cmds[2].size() < 3 || cmds[2].size() > 5 || cmds[2][1] != "fir_decimate_cc") return false;
    int d;
    if (sscanf(cmds[2][2].c_str(), "%d", &d) != 1 || d < 1) return false;
    std::vector<std::string> fused = {"csdr", "ddc_u8_cc"};
    fused.insert(fused.end(), rate_args.begin(), rate_args.end());
    fused.push_back(cmds[2][2]);
    for (size_t k = 3; k < cmds[2].size(); k++) fused.push_back(cmds[2][k]);
    cmds.erase(cmds.begin(), cmds.begin() + 3);
    cmds.insert(cmds.begin(), fused);
    return true;
}

} // namespace

int main(int argc, char **argv)
{
    parse_env();
    if (argc <= 1 || !strcmp(argv[1], "--help")) {
        fprintf(stderr, "csdr (MI355X back end): convert_u8_f convert_f_u8 convert_s8_f convert_f_s8 convert_f_s16 convert_s16_f convert_f_i16 convert_i16_f "
                        "convert_f_s24 convert_s24_f shift_math_cc shift_addition_cc shift_addition_fc shift_table_cc shift_addfast_cc shift_unroll_cc "
                        "decimating_shift_addition_cc fir_decimate_cc fmdemod_quadri_cf fmdemod_quadri_novect_cf fractional_decimator_ff deemphasis_wfm_ff "
                        "deemphasis_nfm_ff limit_ff fastagc_ff bandpass_fir_fft_cc fastddc_fwd_cc fastddc_inv_cc firdes_lowpass_f firdes_bandpass_c "
                        "amdemod_cf amdemod_estimator_cf fmdemod_atan_cf dcblock_ff fastdcblock_ff agc_ff gain_ff realpart_cf logpower_cf fft_cc encode_ima_adpcm_i16_u8 decode_ima_adpcm_u8_i16 compress_fft_adpcm_f_u8 "
                        "setbuf clone through | extensions: wfm_chain_u8_s16 <shift_rate>, nfm_chain_u8_s16 <shift_rate> [decimation [transition_bw]], ddc_u8_cc <shift_rate> <decimation> [transition_bw [window]], fastddc_bank_cc <decimation> <tbw> <window> <ctl|-> <out_0> <rate_0> ..., wfm_bank_u8_s16 / nfm_bank_u8_s16 <shift_rate> <in_0> <out_0> [<in_k> <out_k> ...], chain \"<cmd> <args> | <cmd> <args> ...\"\n");
        return -1;
    }
    g_cmd = argv[1];
    const std::string cmd = argv[1];
    if (cmd == "setbuf") {   // csdr.c:429-438
        if (argc <= 2) return badsyntax("need required parameter (buffer size)");
        int b = 0; sscanf(argv[2], "%d", &b);
        if (b <= 0) return badsyntax("buffer size <= 0 is invalid");
        return passthrough(false, b);
    }
    if (cmd == "clone" || cmd == "REM" || cmd == "through") return passthrough(true, 0);
    if (cmd == "firdes_lowpass_f" || cmd == "firdes_bandpass_c") {   // csdr.c:1251-1335: print the designed taps ("%g " each), --octave wraps them in a plot script
        const bool bp = cmd == "firdes_bandpass_c";
        const int a0 = bp ? 5 : 4;                                    // argv index of the optional window
        if (argc < a0) return badsyntax(bp ? "need required parameters (low_cut, high_cut, length)" : "need required parameters (cutoff_rate, length)");
        float f1 = 0, f2 = 0; int length = 0;
        sscanf(argv[2], "%g", &f1);
        if (bp) sscanf(argv[3], "%g", &f2);
        sscanf(argv[a0 - 1], "%d", &length);
        if (length <= 0 || length % 2 == 0) return badsyntax("number of symmetric FIR filter taps should be odd");
        int window = CSDR_WINDOW_HAMMING;
        if (argc > a0) window = window_from(argv[a0]); else fprintf(stderr, "csdr %s: window = HAMMING\n", g_cmd);
        const bool octave = argc > a0 + 1 && !strcmp(argv[a0 + 1], "--octave");
        if (octave) printf("taps=[");
        if (bp) {
            std::vector<csdr_complexf> t(length); csdr_amd_firdes_bandpass_c(t.data(), length, f1, f2, window);
            for (int i = 0; i < length; i++) printf("(%g)+(%g)*i ", t[i].i, t[i].q);
            if (octave) printf("];spec=fftshift(abs(fft([taps,zeros(1,%d)])).^2);subplot(2,1,1);plot(spec);subplot(2,1,2);plot(arg(fft(taps)));\n", 4 * csdr_amd_next_pow2(length) - length);
        } else {
            std::vector<float> t(length); csdr_amd_firdes_lowpass_f(t.data(), length, f1, window);
            for (int i = 0; i < length; i++) printf("%g ", t[i]);
            if (octave) printf("];plot(taps);figure(2);freqz(taps);\n");
        }
        if (octave) { fflush(stdout); getchar(); }                   // keep octave's window open until the user closes the pipe
        return 0;
    }
    if (cmd == "fractional_decimator_ff" && argc > 2) { float r = 0; sscanf(argv[2], "%g", &r); if (r == 1) return passthrough(true, 0); }   // csdr.c:1494
    // device hand-off from the previous process of the shell pipeline (the streaming commands only): listen before anything slow -- the producer looks for this
    // socket when its first block is ready
    if (cmd != "fastddc_bank_cc" && cmd != "wfm_bank_u8_s16" && cmd != "nfm_bank_u8_s16") ipc_listen_on_stdin();
    const char *dev = getenv("CSDR_AMD_DEVICE");
    csdr_amd_ctx *c = csdr_amd_ctx_create(dev ? atoi(dev) : 0, nullptr);
    if (!c) { fprintf(stderr, "csdr %s: %s\n", g_cmd, csdr_amd_last_error()); return 3; }
    size_t block = block_elems();
    if (cmd == "fastddc_bank_cc") return run_bank(c, argc, argv, block);
    if (cmd == "wfm_bank_u8_s16" || cmd == "nfm_bank_u8_s16") return run_stream_bank(c, argc, argv, cmd[0] == 'n');
    std::vector<Stage *> stages; std::vector<size_t> caps;
    std::vector<std::vector<std::string>> cmds;
    if (cmd == "chain") {
        if (argc <= 2) return badsyntax("need the pipeline as one argument: \"<cmd> <args> | <cmd> <args> ...\"");
        cmds = split_chain(argv[2]);
        std::vector<std::string> rate_args;                            // the shift rate, or --fifo <path> / --fd <n> in its place: fusion AND retune
        if (is_wfm_pattern(cmds, &rate_args)) {
            fprintf(stderr, "csdr chain: WFM receive pattern recognised -> fused matrix-core kernel\n");
            std::vector<std::string> fused = {"csdr", "wfm_chain_u8_s16"}; fused.insert(fused.end(), rate_args.begin(), rate_args.end());
            cmds.assign(1, fused);
        } else if (is_nfm_pattern(cmds, &rate_args) && !g_dynamic && unitround(g_fixed) == 1024) {    // (the chain object models the pipeline at the default buffer size)
            fprintf(stderr, "csdr chain: NFM receive pattern recognised -> fused chain (matrix-core front end and de-emphasis)\n");
            std::vector<std::string> fused = {"csdr", "nfm_chain_u8_s16"}; fused.insert(fused.end(), rate_args.begin(), rate_args.end());
            cmds.assign(1, fused);
        } else if (fuse_front_end(cmds)) {
            fprintf(stderr, "csdr chain: convert_u8_f | shift_addition_cc | fir_decimate_cc recognised -> fused matrix-core front end\n");
        }
    } else {
        cmds.assign(1, std::vector<std::string>(argv, argv + argc));
    }
    if (g_dynamic) ipc_source_decide(dev ? atoi(dev) : 0);           // (before the preamble is read: a producer of ours connects first, then writes it)
    const int in_bufsize = get_bufsize(cmds[0].size() > 1 && (cmds[0][1] == "shift_addition_cc" || cmds[0][1] == "decimating_shift_addition_cc" || cmds[0][1] == "shift_addition_fc"));
    int out_bufsize = in_bufsize;
    // every command may have its control channel, also inside `chain` (fusion and retune together): the newest complete line is applied in front of a pass
    std::vector<Control> ctls(cmds.size());
    bool any_ctl = false;
    for (auto &cm : cmds) for (auto &t : cm) if (t == "--fifo" || t == "--fd") any_ctl = true;
    if (any_ctl && block > 65536 && !getenv("CSDR_AMD_BLOCK")) block = 65536;                               // retune latency
    size_t cap = block; bool cap_is_bytes = false;
    for (size_t k = 0; k < cmds.size(); k++) {
        std::vector<char *> av; for (auto &t : cmds[k]) av.push_back(const_cast<char *>(t.c_str()));
        if (av.size() < 2) return badsyntax("empty command in chain");
        // element size of the next command is only known once it is built; size its block for the worst case (1-byte elements) first
        Stage *s = make_stage(c, (int)av.size(), av.data(), cap, &ctls[k], out_bufsize);
        if (!s) return -1;
        if (ctls[k].fd) s->ctl = &ctls[k];
        if (cap_is_bytes) cap = cap / s->in_elem;
        if (cap < 4 * s->min_block) cap = 4 * s->min_block;
        if (cap < 2 * s->granule) cap = 2 * s->granule;
        if (k == 0) { cap -= cap % s->granule; block = cap; }
        stages.push_back(s); caps.push_back(cap + (k ? 64 : 0));
        out_bufsize = s->next_bufsize(out_bufsize);
        cap = s->out_capacity(caps.back()) * s->out_elem + 8 * (size_t)65536 * 8;    // BYTES the next stage may be handed: this stage's output plus its own carry
        cap_is_bytes = true;
    }
    g_cmd = argv[1];
    const int rc = run(c, stages, caps, in_bufsize, out_bufsize, dev ? atoi(dev) : 0);      // (sends the preamble: behind the hand-off offer to the next process)
    (void)csdr_amd_ctx_sync(c);
    return rc;
}
