#!/bin/bash
# A/B builds: tools/build_variant.sh NAME [-DMACRO ...]  ->  tools/ab/libNAME.so (same sources, extra flags);
# run with SPBLAS_GFX950_LIB=$PWD/tools/ab/libNAME.so.  Measurement scaffolding only.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=tools/ab/obj_$name; mkdir -p $out
for f in handle spmv spmv_sliced spmv_hot spmm spgemm transpose sptrsv multigpu; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-function \
    -I include -I spblas-reference_amd/csrc "$@" -c spblas-reference_amd/csrc/$f.hip -o $out/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/lib$name.so $out/*.o
rm -rf $out
echo tools/ab/lib$name.so
