#!/bin/bash
# front-end wave priorities on the three-interval schedule (stage / tail; shipped: 1 / 2)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
exec > gpurun_out/r06_ab23.txt 2>&1
bash profiles/ab_so.sh C3 _ab/sched3.so _ab/fe_p00.so _ab/fe_p01.so _ab/fe_p11.so _ab/fe_p13.so _ab/fe_p23.so
