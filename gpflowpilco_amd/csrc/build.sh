#!/usr/bin/env bash
# Build the gfx950 shared library in-tree (hipcc cross-compiles without a GPU).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="${here}/../libgpflowpilco_mm.so"
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared \
  -Wno-unused-result \
  "${here}/mm_kernels.hip" "${here}/mm_mfma.hip" "${here}/mm_f64.hip" "${here}/mm_pathwise.hip" "${here}/mm_backward.hip" -o "${out}" "$@"
echo "built ${out}"
