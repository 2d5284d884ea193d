#!/usr/bin/env bash
# Lists every kernel instantiation that uses scratch memory (register spills or dynamically indexed
# register arrays) or more than 256 VGPRs (AGPR spills).  The hot instantiations (d <= 16) must not
# appear: a spilled inner loop cost 8x on the f32 reduce ("Measured and rejected").
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
tmp="$(mktemp -d)"
for f in mm_kernels mm_mfma mm_f64 mm_backward mm_pathwise; do
  extra=()
  [[ $f == mm_mfma ]] && extra=(-fno-honor-nans)
  [[ $f == mm_pathwise ]] && extra=(-fno-slp-vectorize)
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -w "${extra[@]}" \
    "${here}/gpflowpilco_amd/csrc/${f}.hip" -o "${tmp}/${f}.s"
  grep -E "^_Z[0-9]+k_[a-z0-9_]+|ScratchSize:|TotalNumVgprs" "${tmp}/${f}.s" | paste - - - \
    | awk -v f="$f" '{ if ($NF + 0 > 0 || $(NF-3) + 0 > 256) print f ": " substr($1, 1, 70), "vgprs", $(NF-3), "scratch", $NF }'
done
rm -rf "${tmp}"
