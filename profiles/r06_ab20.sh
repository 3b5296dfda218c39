#!/bin/bash
# spec_cols512p: which (column tile, frame group) a workgroup takes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
exec > gpurun_out/r06_ab20.txt 2>&1
bash profiles/ab_so.sh C3 _ab/p1m0.so _ab/p1m1.so _ab/p1m2.so
