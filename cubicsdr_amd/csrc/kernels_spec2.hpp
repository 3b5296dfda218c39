// kernels_spec2.hpp -- spec_cols512: the 512-point column pass in front of the 4096-point rows of a 2^21-point frame (BASELINE config 5:
// fftSize 1 048 576).  One pass through LDS instead of a radix-32 and a radix-16 pass through HBM -- 16 instead of 32 B/sample
// (C5: 0.34 -> 0.18 ms per batch, and the rows / averaging kernels gain from the one-level row layout: 35.0 -> 40.9 GS/s; against a float64
// transform the 2^21-point display values are 9.5e-6 off where the two-pass form was 2.2e-5 and the reference's own class is 1.8e-5).
//
// Replaces (reference file:line): the first half of fft_execute, SpectrumVisualProcessor.cpp:439.
//
// (Round 4 also built the headline size N = 2^17 as 512 x 256 with the averaging fused into a 256-point row pass -- one workgroup per row pair
// walking every frame in order, four doubles of averager state per thread: exact, parity-green over 953 frames, 30 instead of 38 B/sample, and
// slower (0.44 + 0.42 ms against 0.35 + 0.27 + 0.18 ms): those kernels are bound by vector-ALU issue, not by bytes.  DESIGN 12.3 has the
// numbers; the code is in the history at "spec_cols512: the exchange in two halves".)
#pragma once
#include "kernels_spec.hpp"

namespace csdr {

constexpr int kC512 = 512, kC512Cols = 16;
constexpr int kC512KaPitch = 16 * 16 + 16;                       // [k_a][v][col] with 16 float2 of padding per k_a: two k_a of a half-wave hit different bank halves
constexpr size_t kC512Lds = (size_t)16 * kC512KaPitch * sizeof(float2);      // the exchange runs in two halves of sixteen k_a: 35 KB, four workgroups per CU by LDS

// pass 1.  Frame f, columns n2 in [16 blockIdx.x, + 16): X[k1][n2] = W_N^(n2 k1) sum_n1 x[n1 R + n2] W_512^(n1 k1), n1 = 16 u + v, k1 = k_a + 32 k_b.
// thread = (v = tid >> 4, col = tid & 15): radix 32 over u in registers, times W_512^(v k_a); exchange; thread = (k_a = tid >> 4 and + 16, col):
// radix 16 over v, times W_N^(n2 k1); stores Z[f][k1][n2] (16 columns = 128 contiguous bytes per k1).
CSDR_KERNEL __launch_bounds__(kFftThreads) void spec_cols512(FrameSrc fs, int N, const float2 *__restrict__ tw4096, const float2 *__restrict__ tw_hi,
                                                                const float2 *__restrict__ tw_lo, float2 *__restrict__ dst) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *s_x = reinterpret_cast<float2 *>(smem);
    const int R = N / kC512;
    const int f = blockIdx.y, tid = threadIdx.x, col = tid & 15, hi = tid >> 4;
    const int n2 = blockIdx.x * kC512Cols + col;
    const float2 *xb = frame_ptr(fs, f);
    float2 a[32];
    {
        const int v = hi;
        if (f == 0 && fs.split < N) {                                // (block-uniform) the one frame that lies in two pieces
#pragma unroll
            for (int u = 0; u < 32; ++u) a[u] = frame_at(fs, f, xb, (int64_t)(16 * u + v) * R + n2);
        } else {
            const float2 *xc = xb + (int64_t)v * R + n2;
#pragma unroll
            for (int u = 0; u < 32; ++u) a[u] = xc[(int64_t)(16 * u) * R];
        }
        dft_reg<32>(a);
        float2 leaf[5];
#pragma unroll
        for (int l = 0; l < 5; ++l) leaf[l] = tw4096[(8 * v) << l];              // W_512^(v 2^l) = exp(-2 pi i 8 v 2^l / 4096), 8 * 15 * 16 < 4096
        twiddle_powers<32>(a, leaf);
    }
    float2 *o = dst + (int64_t)f * N + n2;
    // W_N^(n2 (ka + 32 kb)) = W_N^(n2 ka) (W_N^(32 n2))^kb
    float2 leaf[5];
#pragma unroll
    for (int l = 0; l < 4; ++l) leaf[l] = tw_split(tw_hi, tw_lo, (unsigned)(32 * n2) << l);      // 32 n2 8 < N (n2 < R = N / 512)
    leaf[4] = leaf[3];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h) __syncthreads();                                      // the first half has been picked up
#pragma unroll
        for (int ka = 0; ka < 16; ++ka) s_x[ka * kC512KaPitch + hi * 16 + col] = a[16 * h + ka];      // (hi = v here)
        __syncthreads();
        const int ka = hi + 16 * h;
        float2 b[16];
#pragma unroll
        for (int v = 0; v < 16; ++v) b[v] = s_x[hi * kC512KaPitch + v * 16 + col];
        dft_reg<16>(b);
        twiddle_powers<16>(b, leaf);
        const float2 w0 = tw_split(tw_hi, tw_lo, (unsigned)(n2 * ka));
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) st_stream(o + (int64_t)(ka + 32 * kb) * R, cmul(b[kb], w0));
    }
}

}  // namespace csdr
