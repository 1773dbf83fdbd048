#!/bin/bash
# Builds the instrumented library (phase probe in the CloudCrop kernel) -> profiles/probe/libgraspnet_hip_probe.so
# Run in the build container: the .so travels to the GPU box with the snapshot.
set -e
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(dirname "$HERE"); SRC=$ROOT/graspnet-baseline_amd/csrc
mkdir -p $HERE/probe/obj
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I$ROOT/include -I$SRC -DGN_PHASE_PROBE $PROBE_DEFS"   # PROBE_DEFS: e.g. -DGN_DBG_NOPOOL
for f in $SRC/*.hip $SRC/*.cpp; do
  o=$HERE/probe/obj/$(basename $f).o
  X=""
  case $(basename $f) in mlp_split.hip|mlp_point_split.hip) X="-mllvm -amdgpu-mfma-vgpr-form";; fps.hip) X="-mllvm -amdgpu-atomic-optimizer-strategy=None";; esac   # as csrc/Makefile
  if [ ! -f $o ] || [ $f -nt $o ] || [ $SRC/mlp_core.h -nt $o ] || [ $SRC/mlp_split.h -nt $o ]; then /opt/rocm/bin/hipcc $FLAGS $X -x hip -c $f -o $o & fi
done
# the evaluation-policy copies of the index-deciding kernels (csrc/Makefile: -DGN_FP=1 / 2), without the probe marks
for f in fps fps_cell query grid knn group; do
  for p in 1 2; do
    o=$HERE/probe/obj/$f.fp$p.o
    X=""; [ $f = fps ] && X="-mllvm -amdgpu-atomic-optimizer-strategy=None"
    if [ ! -f $o ] || [ $SRC/$f.hip -nt $o ] || [ $SRC/gn_common.h -nt $o ]; then /opt/rocm/bin/hipcc ${FLAGS/-DGN_PHASE_PROBE/} $X -DGN_FP=$p -x hip -c $SRC/$f.hip -o $o & fi
  done
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $HERE/probe/libgraspnet_hip_probe.so $HERE/probe/obj/*.o
echo built $HERE/probe/libgraspnet_hip_probe.so
