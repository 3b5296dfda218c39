#!/bin/bash
cd "$(dirname "$0")"
mkdir -p ../../gpurun_out
exec > ../../gpurun_out/xcd_team_probe.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o xcd_team_probe xcd_team_probe.hip 2>/dev/null
for m in 2 3; do timeout 60 ./xcd_team_probe 200 $m 256; done
timeout 100 ./xcd_team_probe 4 0 256
