#!/usr/bin/env bash
# CPU build of csrc/mm_adjoint.h for tests/test_adjoint_host.py (test infrastructure; see mm_adjoint_host.hip).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
hipcc -O2 -std=c++17 --offload-arch=gfx950 -fPIC -shared "${here}/mm_adjoint_host.hip" -o "${here}/libmm_adjoint_host.so"
echo "built ${here}/libmm_adjoint_host.so"
