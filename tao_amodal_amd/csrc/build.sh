#!/bin/bash
# Build the C-ABI library for gfx950 (cross-compiles without a GPU).
# -ffp-contract=off: fp64 IoU must not be fused into fma (bit parity with the
# CPU result); no fast-math anywhere.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=../libtao_amodal_hip.so
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
    -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result \
    api.hip iou_match.hip track_iou.hip flatten.hip sort.hip accumulate.hip exchange.hip rle_iou.hip -o $OUT "$@"
echo "built $(realpath $OUT)"
# host-only: columnar JSON ingest + writer, run-length masks (no GPU code)
g++ -O3 -std=c++17 -fPIC -shared -fopenmp -Wall -ffp-contract=off ingest.cpp rle.cpp jsonwrite.cpp -o ../libtao_amodal_ingest.so
echo "built $(realpath ../libtao_amodal_ingest.so)"
