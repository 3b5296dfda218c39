#!/bin/bash
# A/B helper (runs here, no GPU): profiles/mkvariant.sh NAME UNIT "FLAGS"  ->  _ab/NAME.so = the current library with unit UNIT (csdr_post | csdr_bank | csdr_spec ...)
# recompiled with extra FLAGS (e.g. -DCSDR_P2_MIRROR=0).  The variants are compared on one box with profiles/ab_so.sh.
set -e
cd "$(dirname "$0")/.."
name=$1; unit=$2; flags=$3
python -m cubicsdr_amd.build >/dev/null
mkdir -p _ab/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $flags -c cubicsdr_amd/csrc/$unit.hip -o _ab/obj/$name.o
objs=""
for u in csdr_ctx csdr_post csdr_bank csdr_spec csdr_io csdr_comm; do
  if [ "$u" = "$unit" ]; then objs="$objs _ab/obj/$name.o"; else objs="$objs cubicsdr_amd/csrc/_obj/$u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -ldl -o _ab/$name.so
echo built _ab/$name.so
