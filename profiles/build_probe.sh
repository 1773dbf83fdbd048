#!/bin/bash
# Builds the instrumented library (phase probe in the CloudCrop kernel) -> profiles/probe/libgraspnet_hip_probe.so
# Run in the build container: the .so travels to the GPU box with the snapshot.
set -e
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(dirname "$HERE"); SRC=$ROOT/graspnet-baseline_amd/csrc
mkdir -p $HERE/probe/obj
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I$ROOT/include -I$SRC -DGN_PHASE_PROBE"
for f in $SRC/*.hip $SRC/*.cpp; do
  o=$HERE/probe/obj/$(basename $f).o
  if [ ! -f $o ] || [ $f -nt $o ] || [ $SRC/mlp_core.h -nt $o ]; then /opt/rocm/bin/hipcc $FLAGS -x hip -c $f -o $o & fi
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $HERE/probe/libgraspnet_hip_probe.so $HERE/probe/obj/*.o
echo built $HERE/probe/libgraspnet_hip_probe.so
