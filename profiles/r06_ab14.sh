#!/bin/bash
# the one-wave modem / audio kernels compiled for more waves per SIMD (register budget): C3 bench, kernel times
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
exec > gpurun_out/r06_ab14.txt 2>&1
bash profiles/ab_so.sh C3 _ab/base.so _ab/au_a5.so _ab/au_a6.so _ab/au_a8.so _ab/au_a6m1.so
