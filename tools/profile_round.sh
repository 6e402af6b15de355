#!/bin/bash
# tools/profile_round.sh TAG CONFIG STEPS
# On the GPU box (under gpurun): three rocprofv3 runs of the SAME bench.py command for one BASELINE config --
#   1. --kernel-trace --stats                 -> gpurun_out/prof/TAG_cfgN_stats/
#   2. --kernel-trace --pmc FETCH_SIZE        -> gpurun_out/prof/TAG_cfgN_fetch/
#   3. --kernel-trace --pmc WRITE_SIZE        -> gpurun_out/prof/TAG_cfgN_write/
# Counters are collected for the engine's own kernels only (names k_*): bench.py's synthetic-data generator issues
# thousands of small torch kernels, and counting each of them would serialise the run for minutes.
# Counter passes are separate runs with kernel tracing only (MI355X_MICROARCH.md, HBM section); the program after `--`
# is python3 itself.  Steps are chained: nothing runs after a failure.
set -o pipefail
tag="$1"; cfg="$2"; steps="${3:-10}"
root="$(pwd)"
out="$root/gpurun_out/prof"
mkdir -p "$out"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
cmd=(python3 "$root/bench.py" --config "$cfg" --steps "$steps" --warmup 2 --no-cpu-baseline)
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_cfg${cfg}_stats" -o run -- "${cmd[@]}" > "$out/${tag}_cfg${cfg}_stats.log" 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex '^(void )?k_' --output-format csv -d "$out/${tag}_cfg${cfg}_fetch" -o run -- "${cmd[@]}" > "$out/${tag}_cfg${cfg}_fetch.log" 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex '^(void )?k_' --output-format csv -d "$out/${tag}_cfg${cfg}_write" -o run -- "${cmd[@]}" > "$out/${tag}_cfg${cfg}_write.log" 2>&1
rc=$?
cd "$root"
# keep the merge-back small (gpurun returns at most 64 MiB): drop the per-dispatch traces, keep the summaries and the counter tables
find "$out" -name '*kernel_trace.csv' -delete 2>/dev/null
tail -n 3 "$out/${tag}_cfg${cfg}_stats.log" "$out/${tag}_cfg${cfg}_fetch.log" "$out/${tag}_cfg${cfg}_write.log" 2>/dev/null | cut -c1-400
echo "profile_round $tag cfg$cfg: exit $rc"
exit $rc
