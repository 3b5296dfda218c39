/*
 * pe_loader.c -- TEST INFRASTRUCTURE ONLY (oracle/). Never linked into the product.
 *
 * Minimal PE32+ image loader that maps the reference's own vendored liquid-dsp 1.5.0
 * Windows binary (reference: external/liquid-dsp/gcc/64/libliquid.dll, built from
 * external/liquid-dsp/makefile.mingw64:74 with -O3 -msse4.2 -ffast-math) into this
 * Linux process so that the reference's arithmetic can be executed as the parity oracle.
 *
 * Recipe: mmap SizeOfImage, copy headers + sections, apply DIR64
 * base relocations, bind the KERNEL32/msvcrt imports to ms_abi shims over glibc, resolve
 * exports by name.  DllMain / TLS callbacks / CRT init are deliberately NOT run.
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <errno.h>
#include <ctype.h>
#include <wchar.h>
#include <locale.h>
#include <sys/mman.h>

#define MS __attribute__((ms_abi))

typedef struct {
    uint8_t *img;
    uint32_t size;
    uint32_t exp_rva;
} pe_image;

static pe_image g_img;

/* ------------------------------------------------------------------ msvcrt / kernel32 shims */
static char g_iob[3 * 48]; /* msvcrt FILE is 48 bytes on x64; [0]=stdin [1]=stdout [2]=stderr */

static FILE *map_file(void *f)
{
    char *p = (char *)f;
    if (p >= g_iob && p < g_iob + sizeof(g_iob)) {
        long idx = (p - g_iob) / 48;
        return idx == 0 ? stdin : (idx == 1 ? stdout : stderr);
    }
    return (FILE *)f; /* fopen()ed through our shim => a real glibc FILE* */
}

static MS void *s_iob_func(void) { return g_iob; }
static MS void *s_malloc(size_t n) { return malloc(n); }
static MS void *s_calloc(size_t a, size_t b) { return calloc(a, b); }
static MS void *s_realloc(void *p, size_t n) { return realloc(p, n); }
static MS void s_free(void *p) { free(p); }
static MS void *s_memcpy(void *d, const void *s, size_t n) { return memcpy(d, s, n); }
static MS void *s_memmove(void *d, const void *s, size_t n) { return memmove(d, s, n); }
static MS void *s_memset(void *d, int c, size_t n) { return memset(d, c, n); }
static MS size_t s_strlen(const char *s) { return strlen(s); }
static MS int s_strcmp(const char *a, const char *b) { return strcmp(a, b); }
static MS int s_strncmp(const char *a, const char *b, size_t n) { return strncmp(a, b, n); }
static MS char *s_strerror(int e) { return strerror(e); }
static MS long s_strtol(const char *s, char **e, int b) { return strtol(s, e, b); }
static MS unsigned long s_strtoul(const char *s, char **e, int b) { return strtoul(s, e, b); }
static MS long long s_strtoi64(const char *s, char **e, int b) { return strtoll(s, e, b); }
static MS unsigned long long s_strtoui64(const char *s, char **e, int b) { return strtoull(s, e, b); }
static MS int s_isspace(int c) { return isspace(c); }
static MS int s_isxdigit(int c) { return isxdigit(c); }
static MS int s_tolower(int c) { return tolower(c); }
static MS size_t s_wcslen(const uint16_t *s) { size_t n = 0; while (s[n]) n++; return n; }
static MS int *s_errno(void) { return &errno; }
static MS int s_rand(void) { return rand(); }
static MS void s_abort(void) { fprintf(stderr, "[pe_loader] abort() called from DLL\n"); abort(); }
static MS void s_assert(const char *m, const char *f, unsigned l)
{ fprintf(stderr, "[pe_loader] assert: %s (%s:%u)\n", m, f, l); abort(); }
static MS void s_amsg_exit(int c) { fprintf(stderr, "[pe_loader] _amsg_exit(%d)\n", c); abort(); }
static MS void s_lock(int n) { (void)n; }
static MS void s_unlock(int n) { (void)n; }
static MS void s_initterm(void *a, void *b) { (void)a; (void)b; }
static MS void s_setusermatherr(void *p) { (void)p; }
static MS unsigned s_lc_codepage(void) { return 0; }
static MS int s_mb_cur_max(void) { return 1; }
static MS void *s_localeconv(void) { return localeconv(); }
static MS void s_qsort(void *b, size_t n, size_t s, int(MS *cmp)(const void *, const void *))
{
    /* insertion sort: sizes used by liquid are tiny; avoids an ABI-thunk for the comparator */
    char *a = (char *)b, *tmp = (char *)malloc(s);
    for (size_t i = 1; i < n; i++) {
        memcpy(tmp, a + i * s, s);
        size_t j = i;
        while (j > 0 && cmp(a + (j - 1) * s, tmp) > 0) { memcpy(a + j * s, a + (j - 1) * s, s); j--; }
        memcpy(a + j * s, tmp, s);
    }
    free(tmp);
}
typedef struct { int quot, rem; } ms_div_t;
static MS ms_div_t s_div(int a, int b) { ms_div_t r = { a / b, a % b }; return r; }

/* stdio */
static MS size_t s_fwrite(const void *p, size_t s, size_t n, void *f) { return fwrite(p, s, n, map_file(f)); }
static MS int s_fputc(int c, void *f) { return fputc(c, map_file(f)); }
static MS void *s_fopen(const char *p, const char *m) { return fopen(p, m); }
static MS int s_fclose(void *f) { return fclose(map_file(f)); }
static MS int s_feof(void *f) { return feof(map_file(f)); }
static MS int s_getc(void *f) { return getc(map_file(f)); }
static MS int s_ungetc(int c, void *f) { return ungetc(c, map_file(f)); }
static MS int s_vfprintf(void *f, const char *fmt, void *ap)
{ (void)ap; return fputs(fmt, map_file(f)); } /* ms va_list is not glibc's; print the raw format */

/* math: resolved to glibc (msvcrt is not available); <=1 ulp differences are acceptable at 1e-5 */
static MS float s_sinf(float x) { return sinf(x); }
static MS float s_cosf(float x) { return cosf(x); }
static MS float s_tanf(float x) { return tanf(x); }
static MS float s_expf(float x) { return expf(x); }
static MS float s_logf(float x) { return logf(x); }
static MS float s_log10f(float x) { return log10f(x); }
static MS float s_sinhf(float x) { return sinhf(x); }
static MS float s_coshf(float x) { return coshf(x); }
static MS float s_atan2f(float y, float x) { return atan2f(y, x); }
static MS double s_sin(double x) { return sin(x); }
static MS double s_cos(double x) { return cos(x); }
static MS double s_tan(double x) { return tan(x); }
static MS double s_exp(double x) { return exp(x); }
static MS double s_log(double x) { return log(x); }
static MS double s_log10(double x) { return log10(x); }
static MS double s_sinh(double x) { return sinh(x); }
static MS double s_cosh(double x) { return cosh(x); }
static MS double s_tanh(double x) { return tanh(x); }
static MS double s_pow(double x, double y) { return pow(x, y); }
static MS double s_hypot(double x, double y) { return hypot(x, y); }

/* kernel32 */
static MS void s_cs(void *p) { (void)p; }
static MS uint32_t s_GetLastError(void) { return 0; }
static MS int s_IsDBCSLeadByteEx(unsigned cp, uint8_t c) { (void)cp; (void)c; return 0; }
static MS int s_MultiByteToWideChar(unsigned cp, uint32_t fl, const char *s, int n, uint16_t *w, int wn)
{ (void)cp; (void)fl; int i; if (n < 0) n = (int)strlen(s) + 1; if (!wn) return n;
  for (i = 0; i < n && i < wn; i++) w[i] = (uint8_t)s[i]; return i; }
static MS int s_WideCharToMultiByte(unsigned cp, uint32_t fl, const uint16_t *w, int wn, char *s, int n,
                                    const char *d, int *u)
{ (void)cp; (void)fl; (void)d; if (u) *u = 0; int i; if (wn < 0) { wn = 0; while (w[wn]) wn++; wn++; }
  if (!n) return wn; for (i = 0; i < wn && i < n; i++) s[i] = (char)w[i]; return i; }
static MS void s_Sleep(uint32_t ms) { (void)ms; }
static MS void *s_TlsGetValue(uint32_t i) { (void)i; return NULL; }
static MS int s_VirtualProtect(void *a, size_t n, uint32_t p, uint32_t *o) { (void)a; (void)n; (void)p; if (o) *o = 0x40; return 1; }
static MS size_t s_VirtualQuery(void *a, void *b, size_t n) { (void)a; (void)b; (void)n; return 0; }

static const char *g_unresolved_name = "?";
static MS void s_unresolved(void)
{ fprintf(stderr, "[pe_loader] call into unresolved import (last bound: %s)\n", g_unresolved_name); abort(); }

typedef struct { const char *name; void *fn; } shim;
#define S(n, f) { n, (void *)f }
static const shim g_shims[] = {
    S("__iob_func", s_iob_func), S("malloc", s_malloc), S("calloc", s_calloc), S("realloc", s_realloc),
    S("free", s_free), S("memcpy", s_memcpy), S("memmove", s_memmove), S("memset", s_memset),
    S("strlen", s_strlen), S("strcmp", s_strcmp), S("strncmp", s_strncmp), S("strerror", s_strerror),
    S("strtol", s_strtol), S("strtoul", s_strtoul), S("_strtoi64", s_strtoi64), S("_strtoui64", s_strtoui64),
    S("isspace", s_isspace), S("isxdigit", s_isxdigit), S("tolower", s_tolower), S("wcslen", s_wcslen),
    S("_errno", s_errno), S("rand", s_rand), S("abort", s_abort), S("_assert", s_assert),
    S("_amsg_exit", s_amsg_exit), S("_lock", s_lock), S("_unlock", s_unlock), S("_initterm", s_initterm),
    S("__setusermatherr", s_setusermatherr), S("___lc_codepage_func", s_lc_codepage),
    S("___mb_cur_max_func", s_mb_cur_max), S("localeconv", s_localeconv), S("qsort", s_qsort), S("div", s_div),
    S("fwrite", s_fwrite), S("fputc", s_fputc), S("fopen", s_fopen), S("fclose", s_fclose), S("feof", s_feof),
    S("getc", s_getc), S("ungetc", s_ungetc), S("vfprintf", s_vfprintf),
    S("sinf", s_sinf), S("cosf", s_cosf), S("tanf", s_tanf), S("expf", s_expf), S("logf", s_logf),
    S("log10f", s_log10f), S("sinhf", s_sinhf), S("coshf", s_coshf), S("atan2f", s_atan2f),
    S("sin", s_sin), S("cos", s_cos), S("tan", s_tan), S("exp", s_exp), S("log", s_log), S("log10", s_log10),
    S("sinh", s_sinh), S("cosh", s_cosh), S("tanh", s_tanh), S("pow", s_pow), S("_hypot", s_hypot),
    S("DeleteCriticalSection", s_cs), S("EnterCriticalSection", s_cs), S("InitializeCriticalSection", s_cs),
    S("LeaveCriticalSection", s_cs), S("GetLastError", s_GetLastError), S("IsDBCSLeadByteEx", s_IsDBCSLeadByteEx),
    S("MultiByteToWideChar", s_MultiByteToWideChar), S("WideCharToMultiByte", s_WideCharToMultiByte),
    S("Sleep", s_Sleep), S("TlsGetValue", s_TlsGetValue), S("VirtualProtect", s_VirtualProtect),
    S("VirtualQuery", s_VirtualQuery),
};

static void *find_shim(const char *name)
{
    for (size_t i = 0; i < sizeof(g_shims) / sizeof(g_shims[0]); i++)
        if (!strcmp(g_shims[i].name, name)) return g_shims[i].fn;
    return NULL;
}

/* ------------------------------------------------------------------ loader */
#define RD16(p) (*(const uint16_t *)(p))
#define RD32(p) (*(const uint32_t *)(p))
#define RD64(p) (*(const uint64_t *)(p))

int pe_load(const char *path)
{
    if (g_img.img) return 0;
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "[pe_loader] cannot open %s\n", path); return -1; }
    fseek(f, 0, SEEK_END);
    long fsz = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *file = (uint8_t *)malloc((size_t)fsz);
    if (fread(file, 1, (size_t)fsz, f) != (size_t)fsz) { fclose(f); free(file); return -2; }
    fclose(f);

    if (RD16(file) != 0x5a4d) { free(file); return -3; }
    const uint8_t *nt = file + RD32(file + 0x3c);
    if (RD32(nt) != 0x00004550) { free(file); return -4; }
    uint16_t nsec = RD16(nt + 6), optsz = RD16(nt + 20);
    const uint8_t *opt = nt + 24;
    if (RD16(opt) != 0x20b) { free(file); return -5; } /* PE32+ only */
    uint64_t image_base = RD64(opt + 24);
    uint32_t size_image = RD32(opt + 56), size_hdr = RD32(opt + 60);
    const uint8_t *dd = opt + 112;
    uint32_t exp_rva = RD32(dd + 0), imp_rva = RD32(dd + 8), rel_rva = RD32(dd + 40), rel_sz = RD32(dd + 44);

    uint8_t *img = (uint8_t *)mmap(NULL, size_image, PROT_READ | PROT_WRITE | PROT_EXEC,
                                   MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (img == MAP_FAILED) { free(file); return -6; }
    memcpy(img, file, size_hdr);
    const uint8_t *sh = opt + optsz;
    for (unsigned i = 0; i < nsec; i++, sh += 40) {
        uint32_t va = RD32(sh + 12), rawsz = RD32(sh + 16), raw = RD32(sh + 20);
        if (rawsz && raw) memcpy(img + va, file + raw, rawsz);
    }
    /* base relocations */
    int64_t delta = (int64_t)((uint64_t)img - image_base);
    for (uint32_t off = 0; off + 8 <= rel_sz;) {
        const uint8_t *blk = img + rel_rva + off;
        uint32_t page = RD32(blk), bsz = RD32(blk + 4);
        if (!bsz) break;
        for (uint32_t e = 8; e + 2 <= bsz; e += 2) {
            uint16_t ent = RD16(blk + e);
            if ((ent >> 12) == 10) *(uint64_t *)(img + page + (ent & 0xfff)) += (uint64_t)delta;
        }
        off += bsz;
    }
    /* imports */
    for (const uint8_t *d = img + imp_rva; RD32(d + 12); d += 20) {
        uint32_t oft = RD32(d), ft = RD32(d + 16);
        const uint64_t *lut = (const uint64_t *)(img + (oft ? oft : ft));
        uint64_t *iat = (uint64_t *)(img + ft);
        for (; *lut; lut++, iat++) {
            if (*lut >> 63) { *iat = (uint64_t)(void *)s_unresolved; continue; }
            const char *name = (const char *)(img + (uint32_t)*lut + 2);
            void *fn = find_shim(name);
            if (!fn) { g_unresolved_name = name; fn = (void *)s_unresolved; }
            *iat = (uint64_t)fn;
        }
    }
    free(file);
    g_img.img = img; g_img.size = size_image; g_img.exp_rva = exp_rva;
    return 0;
}

void *pe_sym(const char *name)
{
    if (!g_img.img) return NULL;
    const uint8_t *img = g_img.img, *ed = img + g_img.exp_rva;
    uint32_t nnames = RD32(ed + 24);
    const uint32_t *funcs = (const uint32_t *)(img + RD32(ed + 28));
    const uint32_t *names = (const uint32_t *)(img + RD32(ed + 32));
    const uint16_t *ords = (const uint16_t *)(img + RD32(ed + 36));
    /* names are sorted: binary search */
    uint32_t lo = 0, hi = nnames;
    while (lo < hi) {
        uint32_t mid = (lo + hi) / 2;
        int c = strcmp((const char *)(img + names[mid]), name);
        if (!c) return (void *)(img + funcs[ords[mid]]);
        if (c < 0) lo = mid + 1; else hi = mid;
    }
    return NULL;
}
