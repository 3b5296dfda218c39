#!/bin/bash
# channel counts with a prime factor >= 97: the chirp-z pass of chan_analyze_fft against the two-factor direct-DFT kernel (lab build of the post unit: CSDR_CHAN_FFT=0)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
exec > gpurun_out/r06_blue.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "channelizer_fft_sizes" 2>&1 | tail -4
cp _ab/post_lab.so cubicsdr_amd/libcsdr_hip.so
CHAN_BENCH_ITERS=100 CHAN_BENCH_VARIANTS="tf8t512:CSDR_CHANFFT_TF=8,CSDR_CHANFFT_THREADS=512;tf16t1024:CSDR_CHANFFT_TF=16,CSDR_CHANFFT_THREADS=1024;tf8t256:CSDR_CHANFFT_TF=8,CSDR_CHANFFT_THREADS=256;tf16t512:CSDR_CHANFFT_TF=16,CSDR_CHANFFT_THREADS=512" python profiles/chan_bench.py M194 M202 M254 M298 M326 M388 M398 2>/dev/null
cp _ab/new.so cubicsdr_amd/libcsdr_hip.so
