#!/bin/bash
# Build the C-ABI library for gfx950 (cross-compiles without a GPU).
# -ffp-contract=off: fp64 IoU must not be fused into fma (bit parity with the
# CPU result); no fast-math anywhere.  One object per translation unit, built
# side by side, then one link.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=../libtao_amodal_hip.so
OBJ=../../build/obj
mkdir -p $OBJ
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result"
pids=()
for f in api iou_match track_iou flatten sort accumulate exchange rle_iou json_ingest; do
    extra=""
    # sort.hip: the register bitonic networks are unrolled in full (up to 66
    # layers x 32 registers), beyond the default size limit of #pragma unroll
    [ $f = sort ] && extra="-mllvm -pragma-unroll-threshold=262144"
    $HIPCC $FLAGS $extra "$@" -c $f.hip -o $OBJ/$f.o &
    pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -fPIC -shared $OBJ/api.o $OBJ/iou_match.o $OBJ/track_iou.o $OBJ/flatten.o \
    $OBJ/sort.o $OBJ/accumulate.o $OBJ/exchange.o $OBJ/rle_iou.o $OBJ/json_ingest.o -o $OUT
echo "built $(realpath $OUT)"
# host-only: columnar JSON ingest + writer, run-length masks (no GPU code)
g++ -O3 -std=c++17 -fPIC -shared -fopenmp -Wall -ffp-contract=off ingest.cpp rle.cpp jsonwrite.cpp -o ../libtao_amodal_ingest.so
echo "built $(realpath ../libtao_amodal_ingest.so)"
