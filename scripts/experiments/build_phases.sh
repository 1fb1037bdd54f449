#!/bin/bash
# (re)build scripts/experiments/libdrn_hip_phases.so: the library with -DDRN_NT_PHASES (per-workgroup phase stamps of the GEMM kernels);
# objects cached in /tmp/ph, only the files named on the command line (default: all) are recompiled.  Also prints the AGPR / scratch census
# of the product build of those files.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd $ROOT/drn_amd/csrc
mkdir -p /tmp/ph
FILES=${@:-$(ls *.hip)}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-function $EXTRA_FLAGS"
for f in $FILES; do
  /opt/rocm/bin/hipcc $FLAGS -DDRN_NT_PHASES -c $f -o /tmp/ph/${f%.hip}.o 2>&1 | grep -i " error" -A3 || true
  /opt/rocm/bin/hipcc $FLAGS --cuda-device-only -S -o /tmp/ph/${f%.hip}.s $f 2>&1 | grep -i " error" -A3 || true
  echo "$f: compiler-made AGPR accesses $(awk '/#ASMSTART/{a=1} /#ASMEND/{a=0} !a && /v_accvgpr_/' /tmp/ph/${f%.hip}.s | wc -l), scratch $(grep 'ScratchSize' /tmp/ph/${f%.hip}.s | awk '{print $3}' | tr '\n' ' ')"
done
for f in $(ls *.hip); do [ -f /tmp/ph/${f%.hip}.o ] || /opt/rocm/bin/hipcc $FLAGS -DDRN_NT_PHASES -c $f -o /tmp/ph/${f%.hip}.o; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/ph/*.o -o $ROOT/scripts/experiments/libdrn_hip_phases.so
ls -la $ROOT/scripts/experiments/libdrn_hip_phases.so
