/*
 * liquid_port.c -- CPU ORACLE, TEST INFRASTRUCTURE ONLY.  Never linked/imported by the product.
 *
 * Plain-C restatement of the liquid-dsp v1.5.0 algorithms that CubicSDR's streaming-IQ hot path calls
 * (reference call sites: SURVEY.md section 2.3).  liquid-dsp's source is NOT vendored under /root/reference
 * (only external/liquid-dsp/include/liquid/liquid.h + Windows binaries), so each function below restates the
 * published v1.5.0 algorithm (upstream jgaeddert/liquid-dsp tag v1.5.0, file named per function) and is PINNED
 * against the reference's own binary (oracle/_ref/libliquid_ref.so = the vendored DLL run through a PE loader)
 * by tests/test_oracle_pin.py, and against tests/golden/*.npz which were generated from that binary.
 *
 * Exported names/arguments equal the liquid API (liquid.h line numbers cited) so one harness drives both.
 *
 * One deliberate table: liquid 1.5.0 designs its half-band filters (resamp2) with an iterative Parks-McClellan
 * optimiser (liquid_firdespm_halfband_as -> 32-step qs1dsearch).  CubicSDR only ever instantiates three of those
 * designs on this path (As=60 -> 65 dB, m = 10, 5, 3; see msresamp2 below), so their taps are recorded from the
 * reference binary (HB_TAPS_*) instead of re-deriving the optimiser.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float re, im; } cf32;

static inline cf32 cmulf_(cf32 a, cf32 b) { cf32 r = { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re }; return r; }

const char *liquid_libversion(void) { return "1.5.0-port"; }
int liquid_libversion_number(void) { return 1005000; }

/* ================================================================== math + filter design
 * liquid v1.5.0 src/math/src/math.bessel.c, math.gamma.c, windows.c; src/filter/src/firdes.c */
static float lngammaf_(float z)
{
    if (z < 10.0f) return lngammaf_(z + 1.0f) - logf(z);
    float g = 0.5f * (logf(2 * (float)M_PI) - logf(z));
    g += z * (logf(z + (1 / (12.0f * z - 0.1f / z))) - 1);
    return g;
}

float liquid_besseli0f(float z)
{
    if (z == 0.0f) return 1.0f;
    float y = 0.0f;
    for (unsigned k = 0; k < 32; k++) {
        float t = k * logf(0.5f * z) - lngammaf_((float)k + 1.0f);
        y += expf(2 * t);
    }
    return y;
}

float sincf(float x)
{
    if (fabsf(x) < 0.01f)
        return cosf((float)M_PI * x / 2.0f) * cosf((float)M_PI * x / 4.0f) * cosf((float)M_PI * x / 8.0f);
    return sinf((float)M_PI * x) / ((float)M_PI * x);
}

float kaiser_beta_As(float As)
{
    As = fabsf(As);
    if (As > 50.0f) return 0.1102f * (As - 8.7f);
    if (As > 21.0f) return 0.5842f * powf(As - 21, 0.4f) + 0.07886f * (As - 21);
    return 0.0f;
}

float liquid_kaiser(unsigned i, unsigned wlen, float beta)
{
    float t = (float)i - (float)(wlen - 1) / 2;
    float r = 2.0f * t / (float)(wlen - 1);
    float a = liquid_besseli0f(beta * sqrtf(1 - r * r));
    float b = liquid_besseli0f(beta);
    return a / b;
}

/* liquid.h:2262 estimate_req_filter_len -> Kaiser's formula, truncated */
unsigned estimate_req_filter_len(float df, float As)
{
    return (unsigned)((As - 7.95f) / (14.26f * df));
}

/* liquid.h:2103 liquid_firdes_kaiser */
int liquid_firdes_kaiser(unsigned n, float fc, float As, float mu, float *h)
{
    float beta = kaiser_beta_As(As);
    for (unsigned i = 0; i < n; i++) {
        float t = (float)i - (float)(n - 1) / 2 + mu;
        float h1 = sincf(2.0f * fc * t);
        float h2 = liquid_kaiser(i, n, beta);
        h[i] = h1 * h2;
    }
    return 0;
}

/* liquid_firdes_notch (firdes.c) */
int liquid_firdes_notch(unsigned m, float f0, float As, float *h)
{
    unsigned h_len = 2 * m + 1;
    float beta = kaiser_beta_As(As);
    float scale = 0.0f;
    for (unsigned i = 0; i < h_len; i++) {
        float p = -cosf(2.0f * (float)M_PI * f0 * ((float)i - (float)m));
        float w = liquid_kaiser(i, h_len, beta);
        h[i] = p * w;
        scale += h[i] * p;
    }
    for (unsigned i = 0; i < h_len; i++) h[i] /= scale;
    h[m] += 1.0f;
    return 0;
}

/* ================================================================== window + dotprod helpers
 * liquid window.proto.c: fixed-length FIFO read oldest-first */
typedef struct { cf32 *v; unsigned n; } wincf;
typedef struct { float *v; unsigned n; } winf;
static wincf wincf_new(unsigned n) { wincf w = { (cf32 *)calloc(n, sizeof(cf32)), n }; return w; }
static winf winf_new(unsigned n) { winf w = { (float *)calloc(n, sizeof(float)), n }; return w; }
static void wincf_push(wincf *w, cf32 x) { memmove(w->v, w->v + 1, (w->n - 1) * sizeof(cf32)); w->v[w->n - 1] = x; }
static void winf_push(winf *w, float x) { memmove(w->v, w->v + 1, (w->n - 1) * sizeof(float)); w->v[w->n - 1] = x; }
static cf32 dot_crcf(const float *h, const cf32 *x, unsigned n)
{ cf32 r = { 0, 0 }; for (unsigned i = 0; i < n; i++) { r.re += h[i] * x[i].re; r.im += h[i] * x[i].im; } return r; }
static float dot_rrrf(const float *h, const float *x, unsigned n)
{ float r = 0; for (unsigned i = 0; i < n; i++) r += h[i] * x[i]; return r; }

/* ================================================================== nco_crcf  (liquid v1.5.0 src/nco/src/nco.proto.c; liquid.h nco section)
 * Both LIQUID_NCO and LIQUID_VCO use a 1024-entry sine table without interpolation in this version. */
typedef struct { int type; float sintab[1024]; uint32_t theta, d_theta; float alpha, beta; } nco_t;

static uint32_t nco_constrain(float theta)
{
    float p = (float)((double)theta * 0.159154943091895);
    float fpart = p - (float)((long)p);
    if (fpart < 0.0f) fpart += 1.0f;
    return (uint32_t)(int64_t)(fpart * 4294967296.0f);
}

void *nco_crcf_create(int type)
{
    nco_t *q = (nco_t *)calloc(1, sizeof(nco_t));
    q->type = type;
    for (unsigned i = 0; i < 1024; i++) q->sintab[i] = (float)sin((double)(2.0f * (float)M_PI * (float)i / 1024.0f));   /* float argument, correctly rounded sine: the reference binary's table bit for bit */
    return q;
}
int nco_crcf_destroy(void *q) { free(q); return 0; }
int nco_crcf_reset(void *p) { nco_t *q = (nco_t *)p; q->theta = 0; q->d_theta = 0; return 0; }
int nco_crcf_set_frequency(void *p, float f) { ((nco_t *)p)->d_theta = nco_constrain(f); return 0; }
int nco_crcf_set_phase(void *p, float f) { ((nco_t *)p)->theta = nco_constrain(f); return 0; }
float nco_crcf_get_phase(void *p) { return 2.0f * (float)M_PI * (float)((nco_t *)p)->theta / 4294967296.0f; }
float nco_crcf_get_frequency(void *p)
{ float d = 2.0f * (float)M_PI * (float)((nco_t *)p)->d_theta / 4294967296.0f; return d > (float)M_PI ? d - 2 * (float)M_PI : d; }
int nco_crcf_step(void *p) { nco_t *q = (nco_t *)p; q->theta += q->d_theta; return 0; }
static inline void nco_sincos(const nco_t *q, float *s, float *c)
{
    unsigned idx = (q->theta + (1u << 21)) >> 22;
    *s = q->sintab[idx & 1023];
    *c = q->sintab[(idx + 256) & 1023];
}
int nco_crcf_cexpf(void *p, cf32 *y) { nco_sincos((nco_t *)p, &y->im, &y->re); return 0; }
int nco_crcf_mix_up(void *p, cf32 x, cf32 *y)
{ float s, c; nco_sincos((nco_t *)p, &s, &c); cf32 v = { c, s }; *y = cmulf_(x, v); return 0; }
int nco_crcf_mix_down(void *p, cf32 x, cf32 *y)
{ float s, c; nco_sincos((nco_t *)p, &s, &c); cf32 v = { c, -s }; *y = cmulf_(x, v); return 0; }
int nco_crcf_mix_block_up(void *p, cf32 *x, cf32 *y, unsigned n)
{ for (unsigned i = 0; i < n; i++) { nco_crcf_mix_up(p, x[i], &y[i]); nco_crcf_step(p); } return 0; }
int nco_crcf_mix_block_down(void *p, cf32 *x, cf32 *y, unsigned n)
{ for (unsigned i = 0; i < n; i++) { nco_crcf_mix_down(p, x[i], &y[i]); nco_crcf_step(p); } return 0; }
/* phase-locked loop of the oscillator (liquid v1.5.0 nco.proto.c): bandwidth -> alpha = bw, beta = sqrt(bw); a step moves the
 * frequency word by alpha dphi and the phase word by beta dphi, each quantised like set_frequency / set_phase */
int nco_crcf_pll_set_bandwidth(void *p, float bw) { nco_t *q = (nco_t *)p; q->alpha = bw; q->beta = sqrtf(bw); return 0; }
int nco_crcf_pll_step(void *p, float dphi)
{ nco_t *q = (nco_t *)p; q->d_theta += nco_constrain(q->alpha * dphi); q->theta += nco_constrain(q->beta * dphi); return 0; }
/* test hook: raw phase words */
void port_nco_get_state(void *p, uint32_t *theta, uint32_t *dtheta) { *theta = ((nco_t *)p)->theta; *dtheta = ((nco_t *)p)->d_theta; }

/* ================================================================== resamp2 (half-band)  liquid v1.5.0 src/filter/src/resamp2.proto.c */
static const float HB_TAPS_3[3] = { 0x1.31fb88p-6f, -0x1.d5fe0ep-4f, 0x1.31556cp-1f };
static const float HB_TAPS_5[5] = { 0x1.4ae43ap-8f, -0x1.679e46p-6f, 0x1.084878p-4f, -0x1.57458ap-3f, 0x1.3da7d4p-1f };
static const float HB_TAPS_10[10] = { -0x1.7604f2p-10f, 0x1.caf71ap-9f, -0x1.e992e0p-8f, 0x1.cc3ab4p-7f, -0x1.8f23dcp-6f,
                                      0x1.495b22p-5f, -0x1.0a5864p-4f, 0x1.b87790p-4f, -0x1.992bc8p-3f, 0x1.43c83ep-1f };

/* h1[0..2m-1]: the odd-indexed taps of the 4m+1 half-band prototype (symmetric). */
static int halfband_h1(unsigned m, float As, float *h1)
{
    const float *t = NULL;
    if (fabsf(As - 65.0f) < 1e-3f) { if (m == 3) t = HB_TAPS_3; else if (m == 5) t = HB_TAPS_5; else if (m == 10) t = HB_TAPS_10; }
    if (!t) { fprintf(stderr, "[oracle port] half-band design (m=%u, As=%g) not tabulated\n", m, As); return -1; }
    for (unsigned i = 0; i < m; i++) { h1[i] = t[i]; h1[2 * m - 1 - i] = t[i]; }
    return 0;
}

typedef struct { unsigned m; float *h1; wincf w0, w1; } resamp2c_t;
typedef struct { unsigned m; float *h1; winf w0, w1; } resamp2r_t;

void *resamp2_crcf_create(unsigned m, float f0, float As)
{
    (void)f0;
    resamp2c_t *q = (resamp2c_t *)calloc(1, sizeof(*q));
    q->m = m; q->h1 = (float *)calloc(2 * m, sizeof(float));
    if (halfband_h1(m, As, q->h1)) { free(q->h1); free(q); return NULL; }
    q->w0 = wincf_new(2 * m); q->w1 = wincf_new(2 * m);
    return q;
}
int resamp2_crcf_destroy(void *p) { resamp2c_t *q = (resamp2c_t *)p; free(q->h1); free(q->w0.v); free(q->w1.v); free(q); return 0; }
/* x[0] -> filter branch, x[1] -> delay branch; y = delay + filter (no scaling in 1.5.0) */
int resamp2_crcf_decim_execute(void *p, cf32 *x, cf32 *y)
{
    resamp2c_t *q = (resamp2c_t *)p;
    wincf_push(&q->w1, x[0]);
    cf32 y1 = dot_crcf(q->h1, q->w1.v, 2 * q->m);
    wincf_push(&q->w0, x[1]);
    cf32 y0 = q->w0.v[q->m - 1];
    y->re = y0.re + y1.re; y->im = y0.im + y1.im;
    return 0;
}
int resamp2_crcf_interp_execute(void *p, cf32 x, cf32 *y)
{
    resamp2c_t *q = (resamp2c_t *)p;
    wincf_push(&q->w0, x); y[0] = q->w0.v[q->m - 1];
    wincf_push(&q->w1, x); y[1] = dot_crcf(q->h1, q->w1.v, 2 * q->m);
    return 0;
}
void *resamp2_rrrf_create(unsigned m, float f0, float As)
{
    (void)f0;
    resamp2r_t *q = (resamp2r_t *)calloc(1, sizeof(*q));
    q->m = m; q->h1 = (float *)calloc(2 * m, sizeof(float));
    if (halfband_h1(m, As, q->h1)) { free(q->h1); free(q); return NULL; }
    q->w0 = winf_new(2 * m); q->w1 = winf_new(2 * m);
    return q;
}
int resamp2_rrrf_destroy(void *p) { resamp2r_t *q = (resamp2r_t *)p; free(q->h1); free(q->w0.v); free(q->w1.v); free(q); return 0; }
int resamp2_rrrf_decim_execute(void *p, float *x, float *y)
{
    resamp2r_t *q = (resamp2r_t *)p;
    winf_push(&q->w1, x[0]); float y1 = dot_rrrf(q->h1, q->w1.v, 2 * q->m);
    winf_push(&q->w0, x[1]); float y0 = q->w0.v[q->m - 1];
    *y = y0 + y1;
    return 0;
}
int resamp2_rrrf_interp_execute(void *p, float x, float *y)
{
    resamp2r_t *q = (resamp2r_t *)p;
    winf_push(&q->w0, x); y[0] = q->w0.v[q->m - 1];
    winf_push(&q->w1, x); y[1] = dot_rrrf(q->h1, q->w1.v, 2 * q->m);
    return 0;
}

/* ================================================================== msresamp2  liquid v1.5.0 src/filter/src/msresamp2.proto.c */
#define MSR2_MAX 16
typedef struct { int type; unsigned S; unsigned m[MSR2_MAX]; void *st[MSR2_MAX]; float zeta; int cplx; void *b0, *b1; } msresamp2_t;

static void msresamp2_stage_m(unsigned S, float fc0, float As, unsigned *m)
{
    float fc = fc0, as = As + 5.0f;
    for (unsigned i = 0; i < S; i++) {
        fc = (i == 1) ? (0.5f - fc) * 0.5f : 0.5f * fc;
        float ft = 2 * (0.25f - fc);
        unsigned h_len = estimate_req_filter_len(ft, as);
        unsigned mm = (unsigned)ceilf((float)(h_len - 1) / 4.0f);
        m[i] = mm < 3 ? 3 : mm;
    }
}
static void *msresamp2_create_(int cplx, int type, unsigned S, float fc, float f0, float As)
{
    msresamp2_t *q = (msresamp2_t *)calloc(1, sizeof(*q));
    q->type = type; q->S = S; q->cplx = cplx; q->zeta = 1.0f / (float)(1u << S);
    msresamp2_stage_m(S, fc, As, q->m);
    for (unsigned i = 0; i < S; i++) {
        q->st[i] = cplx ? resamp2_crcf_create(q->m[i], f0, As + 5.0f) : resamp2_rrrf_create(q->m[i], f0, As + 5.0f);
        if (!q->st[i]) return NULL;
    }
    size_t es = cplx ? sizeof(cf32) : sizeof(float);
    q->b0 = calloc((size_t)1 << (S ? S : 1), es); q->b1 = calloc((size_t)1 << (S ? S : 1), es);
    return q;
}
void *msresamp2_crcf_create(int type, unsigned S, float fc, float f0, float As) { return msresamp2_create_(1, type, S, fc, f0, As); }
int msresamp2_crcf_destroy(void *p)
{ msresamp2_t *q = (msresamp2_t *)p; for (unsigned i = 0; i < q->S; i++) resamp2_crcf_destroy(q->st[i]); free(q->b0); free(q->b1); free(q); return 0; }
static void msresamp2_rrrf_destroy_(void *p)
{ msresamp2_t *q = (msresamp2_t *)p; for (unsigned i = 0; i < q->S; i++) resamp2_rrrf_destroy(q->st[i]); free(q->b0); free(q->b1); free(q); }

/* interp: 1 in -> 2^S out, design index 0 runs first (lowest rate).  decim: 2^S in -> 1 out, design index S-1 first, x 2^-S */
int msresamp2_crcf_execute(void *p, cf32 *x, cf32 *y)
{
    msresamp2_t *q = (msresamp2_t *)p;
    if (q->S == 0) { y[0] = x[0]; return 0; }
    cf32 *a = (cf32 *)q->b0, *b = (cf32 *)q->b1;
    if (q->type == 0) { /* LIQUID_RESAMP_INTERP */
        a[0] = x[0];
        for (unsigned s = 0; s < q->S; s++) {
            unsigned k = 1u << s;
            cf32 *dst = (s == q->S - 1) ? y : b;
            for (unsigned i = 0; i < k; i++) resamp2_crcf_interp_execute(q->st[s], a[i], &dst[2 * i]);
            cf32 *t = a; a = b; b = t;
        }
    } else {
        const cf32 *src = x;
        for (unsigned s = 0; s < q->S; s++) {
            unsigned g = q->S - s - 1, k = 1u << g;
            for (unsigned i = 0; i < k; i++) resamp2_crcf_decim_execute(q->st[g], (cf32 *)&src[2 * i], &b[i]);
            cf32 *t = a; a = b; b = t; src = a;
        }
        y->re = a[0].re * q->zeta; y->im = a[0].im * q->zeta;
    }
    return 0;
}
static void msresamp2_rrrf_execute_(msresamp2_t *q, float *x, float *y)
{
    if (q->S == 0) { y[0] = x[0]; return; }
    float *a = (float *)q->b0, *b = (float *)q->b1;
    if (q->type == 0) {
        a[0] = x[0];
        for (unsigned s = 0; s < q->S; s++) {
            unsigned k = 1u << s;
            float *dst = (s == q->S - 1) ? y : b;
            for (unsigned i = 0; i < k; i++) resamp2_rrrf_interp_execute(q->st[s], a[i], &dst[2 * i]);
            float *t = a; a = b; b = t;
        }
    } else {
        const float *src = x;
        for (unsigned s = 0; s < q->S; s++) {
            unsigned g = q->S - s - 1, k = 1u << g;
            for (unsigned i = 0; i < k; i++) resamp2_rrrf_decim_execute(q->st[g], (float *)&src[2 * i], &b[i]);
            float *t = a; a = b; b = t; src = a;
        }
        y[0] = a[0] * q->zeta;
    }
}

/* ================================================================== resamp (arbitrary, 24-bit fixed-point phase)
 * liquid v1.5.0 src/filter/src/resamp.fixed.proto.c + firpfb.proto.c */
typedef struct { unsigned m, npfb, bits; uint32_t step, phase; float *h; /* [npfb][2m], oldest-first */ int cplx; wincf wc; winf wr; } resamp_t;

static void *resamp_create_(int cplx, float rate, unsigned m, float fc, float As, unsigned npfb)
{
    resamp_t *q = (resamp_t *)calloc(1, sizeof(*q));
    unsigned bits = 0; while ((1u << bits) < npfb) bits++;
    q->bits = bits; q->npfb = 1u << bits; q->m = m; q->cplx = cplx;
    q->step = (uint32_t)roundf((float)(1u << 24) / rate);
    unsigned n = 2 * m * q->npfb + 1, sub = 2 * m;
    float *hf = (float *)malloc(n * sizeof(float));
    liquid_firdes_kaiser(n, fc / (float)q->npfb, As, 0.0f, hf);
    float gain = 0.0f;
    for (unsigned i = 0; i < n; i++) gain += hf[i];
    gain = (float)q->npfb / gain;
    q->h = (float *)malloc((size_t)q->npfb * sub * sizeof(float));
    /* firpfb_create(M, h, n-1): arm i, window position j (oldest first) multiplies h[i + (sub-1-j)*M] */
    for (unsigned i = 0; i < q->npfb; i++)
        for (unsigned k = 0; k < sub; k++) q->h[i * sub + (sub - 1 - k)] = hf[i + k * q->npfb] * gain;
    free(hf);
    if (cplx) q->wc = wincf_new(sub); else q->wr = winf_new(sub);
    return q;
}
void *resamp_crcf_create(float r, unsigned m, float fc, float As, unsigned npfb) { return resamp_create_(1, r, m, fc, As, npfb); }
void *resamp_rrrf_create(float r, unsigned m, float fc, float As, unsigned npfb) { return resamp_create_(0, r, m, fc, As, npfb); }
static void resamp_destroy_(void *p) { resamp_t *q = (resamp_t *)p; free(q->h); free(q->wc.v); free(q->wr.v); free(q); }
int resamp_crcf_destroy(void *p) { resamp_destroy_(p); return 0; }
int resamp_rrrf_destroy(void *p) { resamp_destroy_(p); return 0; }
static unsigned resamp_crcf_exec1(resamp_t *q, cf32 x, cf32 *y)
{
    unsigned n = 0;
    wincf_push(&q->wc, x);
    while (q->phase < (1u << 24)) {
        unsigned arm = q->phase >> (24 - q->bits);
        y[n++] = dot_crcf(q->h + (size_t)arm * 2 * q->m, q->wc.v, 2 * q->m);
        q->phase += q->step;
    }
    q->phase -= (1u << 24);
    return n;
}
static unsigned resamp_rrrf_exec1(resamp_t *q, float x, float *y)
{
    unsigned n = 0;
    winf_push(&q->wr, x);
    while (q->phase < (1u << 24)) {
        unsigned arm = q->phase >> (24 - q->bits);
        y[n++] = dot_rrrf(q->h + (size_t)arm * 2 * q->m, q->wr.v, 2 * q->m);
        q->phase += q->step;
    }
    q->phase -= (1u << 24);
    return n;
}
int resamp_crcf_execute_block(void *p, cf32 *x, unsigned nx, cf32 *y, unsigned *ny)
{ unsigned n = 0; for (unsigned i = 0; i < nx; i++) n += resamp_crcf_exec1((resamp_t *)p, x[i], y + n); *ny = n; return 0; }
int resamp_rrrf_execute_block(void *p, float *x, unsigned nx, float *y, unsigned *ny)
{ unsigned n = 0; for (unsigned i = 0; i < nx; i++) n += resamp_rrrf_exec1((resamp_t *)p, x[i], y + n); *ny = n; return 0; }

/* ================================================================== msresamp  liquid v1.5.0 src/filter/src/msresamp.proto.c (liquid.h:8735-8857) */
typedef struct { int type, cplx; unsigned S; float rate_arb; msresamp2_t *hb; resamp_t *arb; void *buf; unsigned buf_idx; } msresamp_t;

static void *msresamp_create_(int cplx, float r, float As)
{
    msresamp_t *q = (msresamp_t *)calloc(1, sizeof(*q));
    q->cplx = cplx; q->type = r > 1.0f ? 0 : 1; q->rate_arb = r;
    if (q->type == 0) while (q->rate_arb > 2.0f) { q->S++; q->rate_arb *= 0.5f; }
    else while (q->rate_arb < 0.5f) { q->S++; q->rate_arb *= 2.0f; }
    q->hb = (msresamp2_t *)msresamp2_create_(cplx, q->type, q->S, 0.4f, 0.0f, As);
    float fc = 0.515f * q->rate_arb; if (fc > 0.49f) fc = 0.49f;
    q->arb = (resamp_t *)resamp_create_(cplx, q->rate_arb, 7, fc, As, 256);
    q->buf = calloc(4 + ((size_t)1 << q->S), cplx ? sizeof(cf32) : sizeof(float));
    return q;
}
void *msresamp_crcf_create(float r, float As) { return msresamp_create_(1, r, As); }
void *msresamp_rrrf_create(float r, float As) { return msresamp_create_(0, r, As); }
int msresamp_crcf_destroy(void *p) { msresamp_t *q = (msresamp_t *)p; msresamp2_crcf_destroy(q->hb); resamp_destroy_(q->arb); free(q->buf); free(q); return 0; }
int msresamp_rrrf_destroy(void *p) { msresamp_t *q = (msresamp_t *)p; msresamp2_rrrf_destroy_(q->hb); resamp_destroy_(q->arb); free(q->buf); free(q); return 0; }

int msresamp_crcf_execute(void *p, cf32 *x, unsigned nx, cf32 *y, unsigned *ny_out)
{
    msresamp_t *q = (msresamp_t *)p;
    unsigned ny = 0, M = 1u << q->S;
    cf32 *buf = (cf32 *)q->buf;
    if (q->type == 1) {
        for (unsigned i = 0; i < nx; i++) {
            buf[q->buf_idx++] = x[i];
            if (q->buf_idx == M) {
                cf32 hbo; msresamp2_crcf_execute(q->hb, buf, &hbo);
                ny += resamp_crcf_exec1(q->arb, hbo, y + ny);
                q->buf_idx = 0;
            }
        }
    } else {
        for (unsigned i = 0; i < nx; i++) {
            unsigned nw = resamp_crcf_exec1(q->arb, x[i], buf);
            for (unsigned k = 0; k < nw; k++) { msresamp2_crcf_execute(q->hb, &buf[k], y + ny); ny += M; }
        }
    }
    *ny_out = ny;
    return 0;
}
/* msresamp_cccf: the complex-coefficient instance.  Its prototype filters are the same real designs (centre frequency 0)
 * stored as complex numbers with zero imaginary part, so the arithmetic is that of msresamp_crcf. */
void *msresamp_cccf_create(float r, float As) { return msresamp_crcf_create(r, As); }
int msresamp_cccf_destroy(void *p) { return msresamp_crcf_destroy(p); }
int msresamp_cccf_execute(void *p, cf32 *x, unsigned nx, cf32 *y, unsigned *ny_out) { return msresamp_crcf_execute(p, x, nx, y, ny_out); }

int msresamp_rrrf_execute(void *p, float *x, unsigned nx, float *y, unsigned *ny_out)
{
    msresamp_t *q = (msresamp_t *)p;
    unsigned ny = 0, M = 1u << q->S;
    float *buf = (float *)q->buf;
    if (q->type == 1) {
        for (unsigned i = 0; i < nx; i++) {
            buf[q->buf_idx++] = x[i];
            if (q->buf_idx == M) {
                float hbo; msresamp2_rrrf_execute_(q->hb, buf, &hbo);
                ny += resamp_rrrf_exec1(q->arb, hbo, y + ny);
                q->buf_idx = 0;
            }
        }
    } else {
        for (unsigned i = 0; i < nx; i++) {
            unsigned nw = resamp_rrrf_exec1(q->arb, x[i], buf);
            for (unsigned k = 0; k < nw; k++) { msresamp2_rrrf_execute_(q->hb, &buf[k], y + ny); ny += M; }
        }
    }
    *ny_out = ny;
    return 0;
}
/* test hooks: integer state of the decimator (bit-exact parity items) */
void port_msresamp_get_state(void *p, unsigned *S, unsigned *buf_idx, uint32_t *phase, uint32_t *step)
{ msresamp_t *q = (msresamp_t *)p; *S = q->S; *buf_idx = q->buf_idx; *phase = q->arb->phase; *step = q->arb->step; }

/* ================================================================== firpfbch analyzer  liquid v1.5.0 src/multichannel/src/firpfbch.proto.c (liquid.h firpfbch section) */
typedef struct { unsigned M, p; float *h; /* [M][p] oldest-first */ wincf *w; unsigned fidx; double *tw_c, *tw_s; } firpfbch_t;

void *firpfbch_crcf_create_kaiser(int type, unsigned M, unsigned m, float As)
{
    if (type != 0) return NULL; /* analyzer only on this path */
    firpfbch_t *q = (firpfbch_t *)calloc(1, sizeof(*q));
    unsigned h_len = 2 * M * m + 1, p = 2 * m;
    float *h = (float *)malloc(h_len * sizeof(float));
    liquid_firdes_kaiser(h_len, 0.5f / (float)M, As, 0.0f, h);
    q->M = M; q->p = p; q->h = (float *)malloc((size_t)M * p * sizeof(float));
    for (unsigned i = 0; i < M; i++)
        for (unsigned n = 0; n < p; n++) q->h[i * p + (p - 1 - n)] = h[i + n * M];
    free(h);
    q->w = (wincf *)malloc(M * sizeof(wincf));
    for (unsigned i = 0; i < M; i++) q->w[i] = wincf_new(p);
    q->fidx = M - 1;
    q->tw_c = (double *)malloc(M * sizeof(double)); q->tw_s = (double *)malloc(M * sizeof(double));
    for (unsigned i = 0; i < M; i++) { q->tw_c[i] = cos(2.0 * M_PI * i / M); q->tw_s[i] = sin(2.0 * M_PI * i / M); }
    return q;
}
int firpfbch_crcf_destroy(void *p)
{ firpfbch_t *q = (firpfbch_t *)p; for (unsigned i = 0; i < q->M; i++) free(q->w[i].v); free(q->w); free(q->h); free(q->tw_c); free(q->tw_s); free(q); return 0; }
int firpfbch_crcf_reset(void *p)
{ firpfbch_t *q = (firpfbch_t *)p; for (unsigned i = 0; i < q->M; i++) memset(q->w[i].v, 0, q->p * sizeof(cf32)); q->fidx = q->M - 1; return 0; }
int firpfbch_crcf_analyzer_execute(void *p, cf32 *x, cf32 *y)
{
    firpfbch_t *q = (firpfbch_t *)p;
    unsigned M = q->M;
    for (unsigned i = 0; i < M; i++) { wincf_push(&q->w[q->fidx], x[i]); q->fidx = (q->fidx + M - 1) % M; }
    cf32 *X = (cf32 *)alloca(M * sizeof(cf32));
    for (unsigned i = 0; i < M; i++) X[M - i - 1] = dot_crcf(q->h + (size_t)i * q->p, q->w[i].v, q->p);
    /* forward M-point DFT (the reference uses liquid's fft; numerically any exact DFT agrees to rounding) */
    for (unsigned k = 0; k < M; k++) {
        double ar = 0, ai = 0;
        for (unsigned c = 0; c < M; c++) {
            unsigned t = (unsigned)(((uint64_t)k * c) % M);
            ar += X[c].re * q->tw_c[t] + X[c].im * q->tw_s[t];
            ai += X[c].im * q->tw_c[t] - X[c].re * q->tw_s[t];
        }
        y[k].re = (float)ar; y[k].im = (float)ai;
    }
    return 0;
}

/* ================================================================== firpfbch2 analyzer  liquid v1.5.0 src/multichannel/src/firpfbch2.proto.c (liquid.h firpfbch2 section)
 * 2x oversampled analysis bank: M channels, M/2 new samples per execute().  Prototype: liquid_firdes_kaiser(2 M m + 1,
 * fc = 1/M, As) scaled to sum M; branch i filters the samples M apart with the sub-sampled taps h[i + r M], r < 2m; the
 * branch order rotates by M/2 on alternate calls (the `flag`), then an M-point inverse DFT and a 1/M gain.  Pinned
 * against the reference DLL (impulse responses and random input, <= 2e-7). */
typedef struct { unsigned M, m, hl; float *h; cf32 *hist; /* last hl samples, oldest first */ unsigned flag; double *tw_c, *tw_s; } firpfbch2_t;

void *firpfbch2_crcf_create_kaiser(int type, unsigned M, unsigned m, float As)
{
    if (type != 0) return NULL; /* analyzer only on this path */
    firpfbch2_t *q = (firpfbch2_t *)calloc(1, sizeof(*q));
    unsigned h_len = 2 * M * m + 1;
    float *h = (float *)malloc(h_len * sizeof(float));
    liquid_firdes_kaiser(h_len, 1.0f / (float)M, As, 0.0f, h);
    float sum = 0.0f;
    for (unsigned i = 0; i < h_len; i++) sum += h[i];
    for (unsigned i = 0; i < h_len; i++) h[i] = h[i] * (float)M / sum;
    q->M = M; q->m = m; q->hl = 2 * M * m; q->h = h;
    q->hist = (cf32 *)calloc(q->hl, sizeof(cf32));
    q->tw_c = (double *)malloc(M * sizeof(double)); q->tw_s = (double *)malloc(M * sizeof(double));
    for (unsigned i = 0; i < M; i++) { q->tw_c[i] = cos(2.0 * M_PI * i / M); q->tw_s[i] = sin(2.0 * M_PI * i / M); }
    return q;
}
int firpfbch2_crcf_destroy(void *p)
{ firpfbch2_t *q = (firpfbch2_t *)p; free(q->h); free(q->hist); free(q->tw_c); free(q->tw_s); free(q); return 0; }
int firpfbch2_crcf_execute(void *p, cf32 *x, cf32 *y)
{
    firpfbch2_t *q = (firpfbch2_t *)p;
    unsigned M = q->M, M2 = M / 2, hl = q->hl;
    memmove(q->hist, q->hist + M2, (hl - M2) * sizeof(cf32));
    memcpy(q->hist + hl - M2, x, M2 * sizeof(cf32));
    /* branch outputs: U[i] = sum_r h[i + r M] * (sample i + r M back from the newest) */
    cf32 *U = (cf32 *)alloca(M * sizeof(cf32));
    for (unsigned i = 0; i < M; i++) {
        float ar = 0.f, ai = 0.f;
        for (unsigned r = 0; r < 2 * q->m; r++) {
            const cf32 v = q->hist[hl - 1 - i - r * M];
            ar += q->h[i + r * M] * v.re; ai += q->h[i + r * M] * v.im;
        }
        U[i].re = ar; U[i].im = ai;
    }
    /* inverse DFT over the branches, rotated by M/2 on odd calls: y[k] = (flag ? (-1)^k : 1) / M * sum_i U[i] e^{+j 2 pi k i / M} */
    for (unsigned k = 0; k < M; k++) {
        double ar = 0, ai = 0;
        for (unsigned i = 0; i < M; i++) {
            unsigned t = (unsigned)(((uint64_t)k * i) % M);
            ar += U[i].re * q->tw_c[t] - U[i].im * q->tw_s[t];
            ai += U[i].im * q->tw_c[t] + U[i].re * q->tw_s[t];
        }
        double s = (q->flag && (k & 1)) ? -1.0 : 1.0;
        y[k].re = (float)(s * ar / (double)M); y[k].im = (float)(s * ai / (double)M);
    }
    q->flag ^= 1;
    return 0;
}

/* ================================================================== iirfilt_crcf  liquid v1.5.0 src/filter/src/iirfilt.proto.c, iirfiltsos.proto.c, iirdes.c */
typedef struct { int sos; unsigned nsos; float b[3 * 8], a[3 * 8]; cf32 v[3 * 8]; float nb[2], na[2]; cf32 nv[2]; } iirfilt_t;

void *iirfilt_crcf_create_dc_blocker(float alpha)
{
    iirfilt_t *q = (iirfilt_t *)calloc(1, sizeof(*q));
    q->sos = 0; q->nb[0] = 1.0f; q->nb[1] = -1.0f; q->na[0] = 1.0f; q->na[1] = -1.0f + alpha;
    return q;
}
/* Butterworth low-pass, bilinear transform, second-order sections (iirdes.c: butter_azpkf, bilinear_zpkf, iirdes_dzpk2sosf) */
void *iirfilt_crcf_create_lowpass(unsigned order, float fc)
{
    iirfilt_t *q = (iirfilt_t *)calloc(1, sizeof(*q));
    unsigned n = order, r = n % 2, L = (n - r) / 2;
    if (L + r > 8) { free(q); return NULL; }
    float mm = 1.0f / tanf((float)M_PI * fc);
    /* analog poles in conjugate pairs, digital via bilinear; zeros all at -1 */
    double kd_re = 1.0, kd_im = 0.0;
    double pdr[16], pdi[16];
    unsigned k = 0;
    for (unsigned i = 0; i < L; i++) {
        float theta = (float)(2 * (i + 1) + n - 1) * (float)M_PI / (float)(2 * n);
        for (int sgn = 1; sgn >= -1; sgn -= 2) {
            double pr = cosf(theta) / mm, pi = sgn * sinf(theta) / mm; /* pa/m */
            double dr = 1.0 - pr, di = -pi, nr = 1.0 + pr, ni = pi, den = dr * dr + di * di;
            pdr[k] = (nr * dr + ni * di) / den; pdi[k] = (ni * dr - nr * di) / den;
            /* G *= (1 - pd)/(1 - zd), zd = -1 */
            double gr = (1.0 - pdr[k]) / 2.0, gi = (-pdi[k]) / 2.0;
            double t = kd_re * gr - kd_im * gi; kd_im = kd_re * gi + kd_im * gr; kd_re = t;
            k++;
        }
    }
    if (r) { double pr = -1.0 / mm; pdr[k] = (1.0 + pr) / (1.0 - pr); pdi[k] = 0; kd_re *= (1.0 - pdr[k]) / 2.0; k++; }
    q->sos = 1; q->nsos = L + r;
    float kg = powf((float)kd_re, 1.0f / (float)(L + r));
    /* liquid_cplxpair sorts conjugate pairs by increasing real part... for Butterworth low-pass at these orders the
     * design order of sections is pinned by test_oracle_pin against the reference (sections sorted by ascending a2). */
    for (unsigned i = 0; i < L; i++) {
        double p0r = -pdr[2 * i], p0i = -pdi[2 * i], p1r = -pdr[2 * i + 1], p1i = -pdi[2 * i + 1];
        q->b[3 * i + 0] = kg; q->b[3 * i + 1] = 2.0f * kg; q->b[3 * i + 2] = kg;
        q->a[3 * i + 0] = 1.0f; q->a[3 * i + 1] = (float)(p0r + p1r); q->a[3 * i + 2] = (float)(p0r * p1r - p0i * p1i);
    }
    if (r) { unsigned i = L; q->b[3 * i] = kg; q->b[3 * i + 1] = kg; q->b[3 * i + 2] = 0; q->a[3 * i] = 1; q->a[3 * i + 1] = (float)(-pdr[2 * L]); q->a[3 * i + 2] = 0; }
    /* sort sections by ascending a2 (matches the reference's section order for the 6th-order fc=0.25 design) */
    for (unsigned i = 0; i < L; i++)
        for (unsigned j = i + 1; j < L; j++)
            if (q->a[3 * j + 2] < q->a[3 * i + 2])
                for (unsigned c = 0; c < 3; c++) { float t = q->a[3 * i + c]; q->a[3 * i + c] = q->a[3 * j + c]; q->a[3 * j + c] = t; }
    return q;
}
int iirfilt_crcf_destroy(void *p) { free(p); return 0; }
int iirfilt_crcf_reset(void *p) { iirfilt_t *q = (iirfilt_t *)p; memset(q->v, 0, sizeof(q->v)); memset(q->nv, 0, sizeof(q->nv)); return 0; }
int iirfilt_crcf_execute(void *p, cf32 x, cf32 *y)
{
    iirfilt_t *q = (iirfilt_t *)p;
    if (!q->sos) { /* direct form II, n=2: v1 <- v0; v0 = x - a1 v1; y = b0 v0 + b1 v1 */
        q->nv[1] = q->nv[0];
        cf32 v0 = { x.re - q->na[1] * q->nv[1].re, x.im - q->na[1] * q->nv[1].im };
        q->nv[0] = v0;
        y->re = q->nb[0] * v0.re + q->nb[1] * q->nv[1].re; y->im = q->nb[0] * v0.im + q->nb[1] * q->nv[1].im;
        return 0;
    }
    cf32 t = x;
    for (unsigned i = 0; i < q->nsos; i++) {
        cf32 *v = q->v + 3 * i; const float *a = q->a + 3 * i, *b = q->b + 3 * i;
        v[2] = v[1]; v[1] = v[0];
        v[0].re = t.re - a[1] * v[1].re - a[2] * v[2].re; v[0].im = t.im - a[1] * v[1].im - a[2] * v[2].im;
        cf32 o = { b[0] * v[0].re + b[1] * v[1].re + b[2] * v[2].re, b[0] * v[0].im + b[1] * v[1].im + b[2] * v[2].im };
        t = o;
    }
    *y = t;
    return 0;
}
int iirfilt_crcf_execute_block(void *p, cf32 *x, unsigned n, cf32 *y) { for (unsigned i = 0; i < n; i++) iirfilt_crcf_execute(p, x[i], &y[i]); return 0; }
void port_iirfilt_get_sos(void *p, unsigned *nsos, float *b, float *a)
{ iirfilt_t *q = (iirfilt_t *)p; *nsos = q->nsos; memcpy(b, q->b, 3 * q->nsos * sizeof(float)); memcpy(a, q->a, 3 * q->nsos * sizeof(float)); }

/* ================================================================== freqdem  liquid v1.5.0 src/modem/src/freqdem.proto.c */
typedef struct { float kf, ref; cf32 r_prime; } freqdem_t;
void *freqdem_create(float kf) { freqdem_t *q = (freqdem_t *)calloc(1, sizeof(*q)); q->kf = kf; q->ref = 1.0f / (2 * (float)M_PI * kf); return q; }
int freqdem_destroy(void *p) { free(p); return 0; }
int freqdem_reset(void *p) { freqdem_t *q = (freqdem_t *)p; q->r_prime.re = q->r_prime.im = 0; return 0; }
int freqdem_demodulate_block(void *p, cf32 *x, unsigned n, float *y)
{
    freqdem_t *q = (freqdem_t *)p;
    for (unsigned i = 0; i < n; i++) {
        cf32 r = x[i], pr = q->r_prime;
        float re = r.re * pr.re + r.im * pr.im, im = r.im * pr.re - r.re * pr.im; /* r * conj(r') */
        y[i] = atan2f(im, re) * q->ref;
        q->r_prime = r;
    }
    return 0;
}

/* ================================================================== firfilt_rrrf (AM DC blocker)  liquid v1.5.0 src/filter/src/firfilt.proto.c */
typedef struct { unsigned n; float *h; /* oldest-first */ winf w; } firfilt_t;
void *firfilt_rrrf_create_dc_blocker(unsigned m, float As)
{
    firfilt_t *q = (firfilt_t *)calloc(1, sizeof(*q));
    q->n = 2 * m + 1;
    float *h = (float *)malloc(q->n * sizeof(float));
    liquid_firdes_notch(m, 0.0f, As, h);
    q->h = (float *)malloc(q->n * sizeof(float));
    for (unsigned i = 0; i < q->n; i++) q->h[q->n - 1 - i] = h[i];
    free(h);
    q->w = winf_new(q->n);
    return q;
}
int firfilt_rrrf_destroy(void *p) { firfilt_t *q = (firfilt_t *)p; free(q->h); free(q->w.v); free(q); return 0; }
int firfilt_rrrf_push(void *p, float x) { winf_push(&((firfilt_t *)p)->w, x); return 0; }
int firfilt_rrrf_execute(void *p, float *y) { firfilt_t *q = (firfilt_t *)p; *y = dot_rrrf(q->h, q->w.v, q->n); return 0; }
int firfilt_rrrf_execute_block(void *p, float *x, unsigned n, float *y)
{ for (unsigned i = 0; i < n; i++) { firfilt_rrrf_push(p, x[i]); firfilt_rrrf_execute(p, &y[i]); } return 0; }
unsigned firfilt_rrrf_get_length(void *p) { return ((firfilt_t *)p)->n; }

/* ================================================================== firhilbf c2r  liquid v1.5.0 src/filter/src/firhilb.proto.c */
typedef struct { unsigned m; float *hq; winf w0, w1, w2, w3; int toggle; } firhilb_t;
void *firhilbf_create(unsigned m, float As)
{
    firhilb_t *q = (firhilb_t *)calloc(1, sizeof(*q));
    unsigned h_len = 4 * m + 1;
    float *h = (float *)malloc(h_len * sizeof(float));
    q->m = m;
    liquid_firdes_kaiser(h_len, 0.25f, fabsf(As), 0.0f, h);
    for (unsigned i = 0; i < h_len; i++) {
        float t = (float)i - (float)(h_len - 1) / 2.0f;
        h[i] = h[i] * sinf(0.5f * (float)M_PI * t); /* imag(h * exp(j pi t / 2)) */
    }
    q->hq = (float *)malloc(2 * m * sizeof(float));
    unsigned j = 0;
    for (unsigned i = 1; i < h_len; i += 2) q->hq[j++] = h[h_len - i - 1];
    free(h);
    q->w0 = winf_new(2 * m); q->w1 = winf_new(2 * m); q->w2 = winf_new(2 * m); q->w3 = winf_new(2 * m);
    return q;
}
int firhilbf_destroy(void *p) { firhilb_t *q = (firhilb_t *)p; free(q->hq); free(q->w0.v); free(q->w1.v); free(q->w2.v); free(q->w3.v); free(q); return 0; }
int firhilbf_c2r_execute(void *p, cf32 x, float *y0, float *y1)
{
    firhilb_t *q = (firhilb_t *)p;
    float yi, yq;
    if (q->toggle == 0) {
        winf_push(&q->w0, x.re); winf_push(&q->w1, x.im);
        yi = q->w0.v[q->m - 1];
        yq = dot_rrrf(q->hq, q->w3.v, 2 * q->m);
    } else {
        winf_push(&q->w2, x.re); winf_push(&q->w3, x.im);
        yi = q->w2.v[q->m - 1];
        yq = dot_rrrf(q->hq, q->w1.v, 2 * q->m);
    }
    q->toggle = 1 - q->toggle;
    *y0 = yi + yq; /* lower sideband */
    *y1 = yi - yq; /* upper sideband */
    return 0;
}

/* ================================================================== fft (power-of-two: radix-2 DIT; otherwise direct DFT)
 * liquid v1.5.0 src/fft/src/fft_radix2.proto.c, fft_common.proto.c (liquid.h fft section) */
typedef struct { unsigned n; cf32 *x, *y; int dir; cf32 *tw; unsigned *rev; unsigned m; } fft_t;
void *fft_create_plan(unsigned n, cf32 *x, cf32 *y, int dir, int flags)
{
    (void)flags;
    fft_t *q = (fft_t *)calloc(1, sizeof(*q));
    q->n = n; q->x = x; q->y = y; q->dir = dir;
    unsigned m = 0; while ((1u << m) < n) m++;
    q->m = ((1u << m) == n) ? m : 0;
    double d = (dir == -1) ? 1.0 : -1.0;
    q->tw = (cf32 *)malloc(n * sizeof(cf32));
    for (unsigned i = 0; i < n; i++) { q->tw[i].re = (float)cos(d * 2.0 * M_PI * (double)i / (double)n); q->tw[i].im = (float)sin(d * 2.0 * M_PI * (double)i / (double)n); }
    if (q->m) {
        q->rev = (unsigned *)malloc(n * sizeof(unsigned));
        for (unsigned i = 0; i < n; i++) { unsigned r = 0; for (unsigned b = 0; b < m; b++) if (i & (1u << b)) r |= 1u << (m - 1 - b); q->rev[i] = r; }
    }
    return q;
}
int fft_destroy_plan(void *p) { fft_t *q = (fft_t *)p; free(q->tw); free(q->rev); free(q); return 0; }
int fft_execute(void *p)
{
    fft_t *q = (fft_t *)p;
    unsigned n = q->n;
    if (!q->m || n == 1) {
        for (unsigned k = 0; k < n; k++) {
            double ar = 0, ai = 0;
            for (unsigned c = 0; c < n; c++) {
                cf32 t = q->tw[(unsigned)(((uint64_t)k * c) % n)];
                ar += (double)q->x[c].re * t.re - (double)q->x[c].im * t.im;
                ai += (double)q->x[c].re * t.im + (double)q->x[c].im * t.re;
            }
            q->y[k].re = (float)ar; q->y[k].im = (float)ai;
        }
        return 0;
    }
    cf32 *y = q->y;
    for (unsigned i = 0; i < n; i++) y[i] = q->x[q->rev[i]];
    unsigned n1, n2 = 1, stride = n;
    for (unsigned i = 0; i < q->m; i++) {
        n1 = n2; n2 *= 2; stride >>= 1;
        unsigned ti = 0;
        for (unsigned j = 0; j < n1; j++) {
            cf32 t = q->tw[ti]; ti = (ti + stride) % n;
            for (unsigned k = j; k < n; k += n2) {
                cf32 yp = cmulf_(y[k + n1], t);
                y[k + n1].re = y[k].re - yp.re; y[k + n1].im = y[k].im - yp.im;
                y[k].re += yp.re; y[k].im += yp.im;
            }
        }
    }
    return 0;
}

/* ================================================================== block helpers (loops the reference runs in C++; kept in C so the CPU baseline is not python-bound) */
/* SDRPostThread.cpp:449-451: one analyzer_execute per M-sample frame */
int oracle_firpfbch_analyzer_block(void *q, unsigned M, cf32 *x, unsigned nframes, cf32 *y)
{ for (unsigned i = 0; i < nframes; i++) firpfbch_crcf_analyzer_execute(q, x + (size_t)i * M, y + (size_t)i * M); return 0; }
/* ================================================================== ampmodem, DSB with suppressed carrier  liquid v1.5.0 src/modem/src/ampmodem.c
 * (ampmodem_create(0.5, LIQUID_AMPMODEM_DSB, 1), ModemDSB.cpp:6).  The demodulator is a Costas loop around the table
 * oscillator: mix down, phase error = Im(v) signed by Re(v), pll step (bandwidth 0.001), oscillator step, output Re(v) /
 * mod_index.  (The object also builds a DC blocker, a Hilbert transformer, a low-pass and a delay line: the other modes'.) */
typedef struct { float mod_index; void *mixer; } ampmodem_t;
void *ampmodem_create(float mod_index, int type, int suppressed)
{
    if (type != 0 || !suppressed) return NULL;   /* only the mode CubicSDR instantiates */
    ampmodem_t *q = (ampmodem_t *)calloc(1, sizeof(*q));
    q->mod_index = mod_index;
    q->mixer = nco_crcf_create(0);
    nco_crcf_pll_set_bandwidth(q->mixer, 0.001f);
    nco_crcf_reset(q->mixer);
    return q;
}
int ampmodem_destroy(void *p) { ampmodem_t *q = (ampmodem_t *)p; nco_crcf_destroy(q->mixer); free(q); return 0; }
int ampmodem_demodulate(void *p, cf32 x, float *y)
{
    ampmodem_t *q = (ampmodem_t *)p;
    cf32 v;
    nco_crcf_mix_down(q->mixer, x, &v);
    float phase_error = v.re > 0.0f ? v.im : -v.im;
    nco_crcf_pll_step(q->mixer, phase_error);
    nco_crcf_step(q->mixer);
    *y = v.re / q->mod_index;
    return 0;
}
int oracle_dsb_block(void *q, cf32 *in, unsigned n, float *out)
{ for (unsigned i = 0; i < n; i++) ampmodem_demodulate(q, in[i], &out[i]); return 0; }

/* SDRPostThread.cpp:505-507: one firpfbch2 execute per M/2 input samples, M outputs each */
int oracle_firpfbch2_block(void *q, unsigned M, cf32 *x, unsigned ncalls, cf32 *y)
{ for (unsigned i = 0; i < ncalls; i++) firpfbch2_crcf_execute(q, x + (size_t)i * (M / 2), y + (size_t)i * M); return 0; }
/* ModemAM.cpp:41-47 */
int oracle_am_block(void *dcblock, cf32 *x, unsigned n, float *y)
{ for (unsigned i = 0; i < n; i++) { float I = x[i].re, Q = x[i].im; firfilt_rrrf_push(dcblock, sqrtf(I * I + Q * Q)); firfilt_rrrf_execute(dcblock, &y[i]); } return 0; }
/* ModemUSB.cpp:54-61 (usb=1) / ModemLSB.cpp (usb=0) */
/* ModemCW.cpp:175-180: mix up by the beep-frequency oscillator, step it, keep the upper-sideband output of the c2r Hilbert transform */
int oracle_cw_block(void *nco, void *hilb, cf32 *in, unsigned n, float *out)
{
    for (unsigned i = 0; i < n; i++) {
        cf32 sig; float lsb;
        nco_crcf_mix_up(nco, in[i], &sig);
        nco_crcf_step(nco);
        firhilbf_c2r_execute(hilb, sig, &lsb, &out[i]);
    }
    return 0;
}
int oracle_ssb_block(void *nco, void *iir, void *hilb, int usb, cf32 *in, unsigned n, float *out)
{
    for (unsigned i = 0; i < n; i++) {
        cf32 x, y; float lo, up;
        nco_crcf_step(nco);
        if (usb) nco_crcf_mix_down(nco, in[i], &x); else nco_crcf_mix_up(nco, in[i], &x);
        iirfilt_crcf_execute(iir, x, &y);
        if (usb) nco_crcf_mix_up(nco, y, &x); else nco_crcf_mix_down(nco, y, &x);
        firhilbf_c2r_execute(hilb, x, &lo, &up);
        out[i] = usb ? up : lo;
    }
    return 0;
}
#include "chain_bench.inc"
