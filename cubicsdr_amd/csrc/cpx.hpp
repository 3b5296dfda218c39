// cpx.hpp -- complex float32 arithmetic on the packed fp32 pipe of gfx950.
//
// A complex sample is ONE 64-bit register pair (float ext_vector_type(2)): a complex add / subtract is one v_pk_add_f32, a complex product is
// v_pk_mul_f32 + v_pk_fma_f32 (the operand swizzles and the sign of the -im x im term ride on op_sel / neg_lo), a multiplication by -i / +i is
// folded into the add that consumes it (op_sel swap + neg).  The transforms of the spectrum chain are bound by vector-ALU ISSUE:
// with float2 structs the compiler packed about a quarter of their arithmetic (profiles/isa_stats.py: 230 v_pk_* of ~760 arithmetic
// instructions in spec_fft_rows4096); written on this type every butterfly is packed.
//
// A host compiler (the host-executing test build) has no vector swizzles: there the same functions are plain float arithmetic.
#pragma once
#include "common.hpp"

namespace csdr {

#if defined(__AMDGCN__)
typedef float cpx __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cpx cpx_make(float re, float im) { return cpx{re, im}; }
// a * w, w in registers (a twiddle that was loaded): (a.x w.x - a.y w.y, a.x w.y + a.y w.x)
__device__ __forceinline__ cpx cpx_mul(cpx a, cpx w) {
    cpx r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"                                        // (a.x w.x, a.x w.y)
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"        // + (a.y (-w.y), a.y w.x)
        : "=&v"(r) : "v"(a), "v"(w));
    return r;
}
// a * w, w a compile-time constant: left to the compiler (the constant and its rotated / negated copy become scalar operands)
__device__ __forceinline__ cpx cpx_mulc(cpx a, float wr, float wi) { return a.xx * cpx{wr, wi} + a.yy * cpx{-wi, wr}; }
// a - i b and a + i b (the +-i rotation of a radix-4 / radix-2 butterfly, free)
__device__ __forceinline__ cpx cpx_sub_ib(cpx a, cpx b) {
    cpx r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));     // (a.x + b.y, a.y - b.x)
    return r;
}
__device__ __forceinline__ cpx cpx_add_ib(cpx a, cpx b) {
    cpx r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));     // (a.x - b.y, a.y + b.x)
    return r;
}
__device__ __forceinline__ float cpx_abs(cpx v) { const cpx q = v * v; return __builtin_amdgcn_sqrtf(q.x + q.y); }      // v_sqrt_f32: one ulp (cabs_f, kernels_spec.hpp)
// acc + s v, both components fused (one v_pk_fma_f32; s may be wave-uniform: a scalar operand): a real tap on a complex / two-stream sample
__device__ __forceinline__ cpx cpx_fma_s(float s, cpx v, cpx acc) { return __builtin_elementwise_fma(cpx{s, s}, v, acc); }
#else
struct cpx { float x, y; };
static inline cpx cpx_make(float re, float im) { return cpx{re, im}; }
static inline cpx operator+(cpx a, cpx b) { return cpx{a.x + b.x, a.y + b.y}; }
static inline cpx operator-(cpx a, cpx b) { return cpx{a.x - b.x, a.y - b.y}; }
static inline cpx cpx_mul(cpx a, cpx w) { return cpx{fmaf(a.x, w.x, -(a.y * w.y)), fmaf(a.y, w.x, a.x * w.y)}; }
static inline cpx cpx_mulc(cpx a, float wr, float wi) { return cpx_mul(a, cpx{wr, wi}); }
static inline cpx cpx_sub_ib(cpx a, cpx b) { return cpx{a.x + b.y, a.y - b.x}; }
static inline cpx cpx_add_ib(cpx a, cpx b) { return cpx{a.x - b.y, a.y + b.x}; }
static inline float cpx_abs(cpx v) { return sqrtf(v.x * v.x + v.y * v.y); }
static inline cpx cpx_fma_s(float s, cpx v, cpx acc) { return cpx{fmaf(s, v.x, acc.x), fmaf(s, v.y, acc.y)}; }
#endif
__device__ __forceinline__ cpx cpx_from(float2 v) { return cpx_make(v.x, v.y); }
__device__ __forceinline__ float2 cpx_to(cpx v) { return make_float2(v.x, v.y); }

// exp(-2 pi i k / 16), k = 0 .. 7 (constants: the arguments of cpx_mulc)
__device__ __forceinline__ constexpr float cw16_re(int k) {
    constexpr float c[8] = {1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f, 0.0f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f};
    return c[k];
}
__device__ __forceinline__ constexpr float cw16_im(int k) {
    constexpr float s[8] = {0.0f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f, -1.0f, -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f};
    return s[k];
}
template <int R> __device__ inline constexpr int cpx_bitrev(int i) { int r = 0; for (int b = 1; b < R; b <<= 1) { r = (r << 1) | (i & 1); i >>= 1; } return r; }

// in-register forward DFT of R = 2, 4, 8 or 16 points, natural order in and out: radix-2 decimation in time; the twiddles of a stage are
// constants, exp(-2 pi i / 4) is the free rotation
template <int R>
__device__ __forceinline__ void cpx_dft(cpx (&v)[R]) {
    cpx t[R];
#pragma unroll
    for (int i = 0; i < R; ++i) t[cpx_bitrev<R>(i)] = v[i];
#pragma unroll
    for (int len = 2; len <= R; len <<= 1) {
#pragma unroll
        for (int i = 0; i < R; i += len) {
#pragma unroll
            for (int j = 0; j < len / 2; ++j) {
                const int tk = j * (16 / len);                  // twiddle exponent on the 16-point circle, 0 .. 7
                const cpx a = t[i + j], b = t[i + j + len / 2];
                if (tk == 0) { t[i + j] = a + b; t[i + j + len / 2] = a - b; }
                else if (tk == 4) { t[i + j] = cpx_sub_ib(a, b); t[i + j + len / 2] = cpx_add_ib(a, b); }       // b (-i)
                else { const cpx bw = cpx_mulc(b, cw16_re(tk), cw16_im(tk)); t[i + j] = a + bw; t[i + j + len / 2] = a - bw; }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) v[i] = t[i];
}

}  // namespace csdr
