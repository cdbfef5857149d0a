/* oracle/fftw_shim.c -- TEST INFRASTRUCTURE, not product code.
 *
 * A small CPU DFT behind the seven fftwf_* symbols the reference links against
 * (fft_fftw.c:9-45).  FFTW3 itself is absent from this image and from
 * /root/reference (un-vendored dependency, pinned only as "fftw-3.3.3" for the
 * Emscripten build, Makefile:44).  The DFT is mathematically defined:
 *     X[k] = sum_n x[n] * exp(sign * 2*pi*i * n*k / N)      (unnormalised)
 * so any correct transform is a valid oracle at the 1e-5 tolerance; this one
 * works in double precision internally (error ~1e-16) and rounds once to float,
 * i.e. it is *more* accurate than FFTW's float codelets.
 *
 * Power-of-two sizes: iterative radix-2 with a precomputed double twiddle table.
 * Other sizes: direct O(N^2) evaluation (the hot path only uses powers of two).
 */
#include "fftw3.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

enum { KIND_C2C = 0, KIND_R2C = 1, KIND_C2R = 2 };

struct oracle_fftwf_plan_s {
    int n, sign, kind, pow2;
    void *in, *out;
    double *wr, *wi;      /* twiddles exp(sign*2*pi*i*k/n), k < n/2 (pow2) or k < n */
    double *re, *im;      /* work arrays */
    int *rev;             /* bit-reversal permutation (pow2) */
};

static fftwf_plan mkplan(int n, void *in, void *out, int sign, int kind)
{
    fftwf_plan p = (fftwf_plan)calloc(1, sizeof(*p));
    p->n = n; p->sign = sign; p->kind = kind; p->in = in; p->out = out;
    p->pow2 = n > 0 && (n & (n - 1)) == 0;
    int nt = p->pow2 ? (n / 2 > 0 ? n / 2 : 1) : n;
    p->wr = (double *)malloc(sizeof(double) * nt);
    p->wi = (double *)malloc(sizeof(double) * nt);
    for (int k = 0; k < nt; k++) {
        double a = sign * 2.0 * M_PI * (double)k / (double)n;
        p->wr[k] = cos(a); p->wi[k] = sin(a);
    }
    p->re = (double *)malloc(sizeof(double) * n);
    p->im = (double *)malloc(sizeof(double) * n);
    if (p->pow2) {
        p->rev = (int *)malloc(sizeof(int) * n);
        int bits = 0; while ((1 << bits) < n) bits++;
        for (int i = 0; i < n; i++) {
            int r = 0;
            for (int b = 0; b < bits; b++) if (i & (1 << b)) r |= 1 << (bits - 1 - b);
            p->rev[i] = r;
        }
    }
    return p;
}

/* in-place transform of p->re/p->im (natural order in, natural order out) */
static void run(fftwf_plan p)
{
    int n = p->n;
    double *re = p->re, *im = p->im;
    if (p->pow2) {
        for (int i = 0; i < n; i++) {
            int r = p->rev[i];
            if (r > i) { double t = re[i]; re[i] = re[r]; re[r] = t; t = im[i]; im[i] = im[r]; im[r] = t; }
        }
        for (int len = 2; len <= n; len <<= 1) {
            int half = len >> 1, step = n / len;
            for (int base = 0; base < n; base += len)
                for (int j = 0; j < half; j++) {
                    double wr = p->wr[j * step], wi = p->wi[j * step];
                    int a = base + j, b = a + half;
                    double tr = re[b] * wr - im[b] * wi, ti = re[b] * wi + im[b] * wr;
                    re[b] = re[a] - tr; im[b] = im[a] - ti;
                    re[a] += tr; im[a] += ti;
                }
        }
    } else {
        double *or_ = (double *)malloc(sizeof(double) * n), *oi = (double *)malloc(sizeof(double) * n);
        for (int k = 0; k < n; k++) {
            double sr = 0, si = 0;
            for (int j = 0; j < n; j++) {
                int idx = (int)(((long long)j * k) % n);
                sr += re[j] * p->wr[idx] - im[j] * p->wi[idx];
                si += re[j] * p->wi[idx] + im[j] * p->wr[idx];
            }
            or_[k] = sr; oi[k] = si;
        }
        memcpy(re, or_, sizeof(double) * n); memcpy(im, oi, sizeof(double) * n);
        free(or_); free(oi);
    }
}

fftwf_plan fftwf_plan_dft_1d(int n, fftwf_complex *in, fftwf_complex *out, int sign, unsigned flags)
{ (void)flags; return mkplan(n, in, out, sign, KIND_C2C); }

fftwf_plan fftwf_plan_dft_r2c_1d(int n, float *in, fftwf_complex *out, unsigned flags)
{ (void)flags; return mkplan(n, in, out, FFTW_FORWARD, KIND_R2C); }

fftwf_plan fftwf_plan_dft_c2r_1d(int n, fftwf_complex *in, float *out, unsigned flags)
{ (void)flags; return mkplan(n, in, out, FFTW_BACKWARD, KIND_C2R); }

void fftwf_execute(const fftwf_plan p)
{
    int n = p->n;
    if (p->kind == KIND_C2C) {
        const float *x = (const float *)p->in; float *y = (float *)p->out;
        for (int i = 0; i < n; i++) { p->re[i] = x[2 * i]; p->im[i] = x[2 * i + 1]; }
        run(p);
        for (int i = 0; i < n; i++) { y[2 * i] = (float)p->re[i]; y[2 * i + 1] = (float)p->im[i]; }
    } else if (p->kind == KIND_R2C) {
        const float *x = (const float *)p->in; float *y = (float *)p->out;
        for (int i = 0; i < n; i++) { p->re[i] = x[i]; p->im[i] = 0; }
        run(p);
        for (int i = 0; i <= n / 2; i++) { y[2 * i] = (float)p->re[i]; y[2 * i + 1] = (float)p->im[i]; }
    } else {
        const float *x = (const float *)p->in; float *y = (float *)p->out;
        for (int i = 0; i <= n / 2; i++) { p->re[i] = x[2 * i]; p->im[i] = x[2 * i + 1]; }
        for (int i = n / 2 + 1; i < n; i++) { p->re[i] = p->re[n - i]; p->im[i] = -p->im[n - i]; }
        p->im[0] = 0; if (n % 2 == 0) p->im[n / 2] = 0;
        run(p);
        for (int i = 0; i < n; i++) y[i] = (float)p->re[i];
    }
}

void fftwf_destroy_plan(fftwf_plan p)
{
    if (!p) return;
    free(p->wr); free(p->wi); free(p->re); free(p->im); free(p->rev); free(p);
}

void *fftwf_malloc(size_t n) { void *p = NULL; if (posix_memalign(&p, 64, n ? n : 64)) return NULL; return p; }
void fftwf_free(void *p) { free(p); }
