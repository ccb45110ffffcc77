#!/bin/bash
# Builds libcriminisi_hip.so for gfx950 next to the package (image-processing_amd/lib/).
# -ffp-contract=off: the fp64 confidence / priority / cost arithmetic must not be fused.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="${CRIM_LIB_OUT:-$HERE/../lib}"   # CRIM_LIB_OUT: alternative output directory (diagnostic builds)
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 \
    -Wall -Wno-unused-result $EXTRA_HIPCC_FLAGS \
    -o "$OUT/libcriminisi_hip.so" "$HERE/crim_engine.hip"
echo "built $OUT/libcriminisi_hip.so"
# the hand-counted wait of the search kernels' select prologue is checked against the generated code
python3 "$HERE/../../tools/check_select_prologue.py" "$OUT/libcriminisi_hip.so"
