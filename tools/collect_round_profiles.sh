#!/usr/bin/env bash
# Everything under profiles/<round>_* from ONE command on the GPU box (about 12 minutes):
#
#   ROUND=r05 bash tools/collect_round_profiles.sh          # writes gpurun_out/<round>/final/ and gpurun_out/<round>/<round>_pmc_*.json
#
#   1. the driver's bench command verbatim (line + bench_detail.json);
#   2. rocprofv3 --kernel-trace --stats of the same workload, one recipe at a time (per-kernel durations);
#   3. hardware counters (tools/collect_pmc.sh: --pmc only, one pass per group) for every config, copied into profiles/ ON THE BOX so
#      that step 4's bench lines carry roofline.traffic / issue_frac from counters of this very tree (src_hash);
#   4. the bench line of every other config, the two-rank gloo rehearsal, the pathwise Jacobian pass's counters;
#   5. the GPU test suite and smoke().
# Copy what is to be judged from gpurun_out/<round>/ into profiles/ afterwards (gpurun_out/ is scratch).
set -u
round="${ROUND:-r05}"
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; O="$R/gpurun_out/$round/final"; mkdir -p "$O"
cd "$R"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_c3_driver_line.json" 2> "$O/bench_c3_driver.err" && cp bench_detail.json "$O/bench_c3_driver.json"
export TMPDIR=/tmp
for rec in baseline pilco worst; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d "$O/kt_$rec" -o t --output-format csv -- python3 "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --recipe $rec --no-cpu-baseline > "$O/bench_c3_${rec}_under_rocprof_line.json" 2> "$O/kt_$rec.err")
  cp "$R/bench_detail.json" "$O/bench_c3_${rec}_under_rocprof.json"; cp "$O/kt_$rec/t_kernel_stats.csv" "$O/bench_c3_${rec}_kernel_stats.csv"; rm -rf "$O/kt_$rec"
done
for t in "c3_baseline --config c3 --recipe baseline" "c3_pilco --config c3 --recipe pilco" "c3_worst --config c3 --recipe worst" "c2_pilco --config c2" \
         "c3_grad_baseline --config c3_grad" "c5 --config c5" "c4 --config c4 --batch 32"; do
  set -- $t; tag=$1; shift
  ROUND=$round bash tools/collect_pmc.sh "$tag" "$@" > "$O/pmc_$tag.log" 2>&1; tail -1 "$O/pmc_$tag.log"
done
cp "$R/gpurun_out/$round/${round}"_pmc_*.json "$R/profiles/"
run() { name=$1; shift; python3 bench.py "$@" > "$O/bench_${name}_line.json" 2> "$O/bench_${name}.err"; cp bench_detail.json "$O/bench_${name}.json"; }
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_c3_driver_line.json" 2> "$O/bench_c3_driver.err" && cp bench_detail.json "$O/bench_c3_driver.json"
run c1 --config c1; run c1_closed --config c1_closed; run c2 --config c2; run c3_grad --config c3_grad; run c5 --config c5
run c5_shard --config c5 --batch 8192; run c4_shard --config c4 --batch 32 --steps 10 --warmup 2
run c3_rehearse_gloo2 --gpus 2 --rehearse-gloo --steps 10 --warmup 2
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$O/pwj_kt" -o t --output-format csv -- python3 "$R/tools/pathwise_jac_run.py" > "$O/pwj_kt.log" 2>&1)
cp "$O/pwj_kt/t_kernel_stats.csv" "$O/pathwise_jac_kernel_stats.csv"; rm -rf "$O/pwj_kt"
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$O/qs" -o t --output-format csv -- python3 "$R/tools/q_stage_kernels.py" --reps 10 > "$O/qs.log" 2>&1)
cp "$O/qs/t_kernel_stats.csv" "$O/q_stage_standalone_kernel_stats.csv"; rm -rf "$O/qs"
timeout -k 10 800 python3 -m pytest tests -m gpu -q --timeout 400 > "$O/pytest_gpu.log" 2>&1; tail -2 "$O/pytest_gpu.log"
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
