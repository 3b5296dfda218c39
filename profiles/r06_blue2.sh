#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
exec > gpurun_out/r06_blue2.txt 2>&1
cp _ab/post_lab.so cubicsdr_amd/libcsdr_hip.so
CHAN_BENCH_BASE=0 CHAN_BENCH_ITERS=100 CHAN_BENCH_VARIANTS="tf8:CSDR_CHANFFT_TF=8;tf16:CSDR_CHANFFT_TF=16;tf32:CSDR_CHANFFT_TF=32;tf8t512:CSDR_CHANFFT_TF=8,CSDR_CHANFFT_THREADS=512;tf16t1024:CSDR_CHANFFT_TF=16,CSDR_CHANFFT_THREADS=1024" python profiles/chan_bench.py M116 M134 M202 M290 M398 2>/dev/null
cp _ab/new.so cubicsdr_amd/libcsdr_hip.so
