#!/usr/bin/env bash
# Collect the hardware counters bench.py's roofline block reads, on the GPU box.
#
#   tools/collect_pmc.sh <tag> [bench.py args...]        e.g.  tools/collect_pmc.sh c3 --config c3
#
# One rocprofv3 run per counter group (<= 8 SQ counters, FETCH_SIZE and WRITE_SIZE in separate passes --
# MI355X_MICROARCH.md "rocprofv3 PMC slots"), each with --pmc ONLY (no trace domains), the program directly
# after "--".  Raw output: gpurun_out/<round>/pmc_<tag>/pass*/ ; summary (what bench.py reads, tagged with the
# hash of the kernel sources it was measured on): profiles/<round>_pmc_<tag>.json  via tools/pmc_summary.py.
set -euo pipefail
tag="$1"; shift
round="${ROUND:-r05}"
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="${root}/gpurun_out/${round}/pmc_${tag}"
rm -rf "${out}"; mkdir -p "${out}"
export TMPDIR=/tmp
groups=(
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
  "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32"
  "SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"
  "FETCH_SIZE"
  "WRITE_SIZE"
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
)
cd /tmp
i=0
for g in "${groups[@]}"; do
  i=$((i + 1))
  # shellcheck disable=SC2086
  # only this library's kernels are counted (the model set-up launches thousands of rocBLAS / rocSOLVER kernels:
  # counting those makes a pass take many minutes)
  rocprofv3 --pmc ${g} --kernel-include-regex "^(void )?(k_|mm_)" -d "${out}/pass${i}" -o pmc --output-format csv -- python3 "${root}/bench.py" "$@" --steps "${PMC_STEPS:-4}" --warmup 1 --no-cpu-baseline --pmc-run \
    > "${out}/pass${i}.log" 2>&1 || { echo "pass ${i} failed (see ${out}/pass${i}.log)"; tail -5 "${out}/pass${i}.log"; exit 1; }
  echo "pass ${i} done: ${g}"
done
python3 "${root}/tools/pmc_summary.py" "${out}" --tag "${tag}" -o "${root}/gpurun_out/${round}/${round}_pmc_${tag}.json" --bench-args "$*"
# the raw per-dispatch CSVs are large (gpurun merges at most 64 MiB back): keep the summary and the logs only
find "${out}" -name "*.csv" -delete; find "${out}" -name "*.db" -delete
echo "summary: gpurun_out/${round}/${round}_pmc_${tag}.json  (copy to profiles/ to commit)"
