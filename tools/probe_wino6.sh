#!/bin/bash
# Probe build of csrc/conv_wino6.hip, csrc/conv_wino6s.hip (python tools/probe_wino6.py --split) and csrc/conv_wgrad_wino6.hip (-DRAMNET_PROBE: shader-clock stamps at the phase boundaries of every workgroup) ->
# rpg_ramnet_amd/abl/lib_probe6.so (git-ignored, travels with gpurun); run:  RAMNET_HIP_LIB=rpg_ramnet_amd/abl/lib_probe6.so python tools/probe_wino6.py
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/rpg_ramnet_amd
mkdir -p $PKG/abl
python -c "from rpg_ramnet_amd import build; build.build()" > /dev/null
OBJS=$(ls $PKG/build/*.o | grep -v "conv_wino6.o\|conv_wgrad_wino6.o\|conv_wino6s.o")
for F in conv_wino6 conv_wgrad_wino6 conv_wino6s; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DRAMNET_PROBE -c $PKG/csrc/$F.hip -o /tmp/${F}_probe.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/abl/lib_probe6.so $OBJS /tmp/conv_wino6_probe.o /tmp/conv_wgrad_wino6_probe.o /tmp/conv_wino6s_probe.o
ls -la $PKG/abl/lib_probe6.so
