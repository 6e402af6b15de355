#!/bin/bash
# tools/profile_r03.sh CONFIG : the round's records for one BASELINE config, on the GPU box (under gpurun):
#   bench line (with CPU baseline), rocprofv3 --kernel-trace --stats of the same command, FETCH_SIZE / WRITE_SIZE passes,
#   -> gpurun_out/r03/ ; tools/collect_r03.sh copies the summaries into profiles/.
set -o pipefail
cfg="$1"; root="$(pwd)"; out="$root/gpurun_out/r03"; mkdir -p "$out"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python3 bench.py --config $cfg --steps 20 --warmup 3 > "$out/bench_cfg$cfg.log" 2>&1 || { tail -5 "$out/bench_cfg$cfg.log"; exit 1; }
grep '^{"metric' "$out/bench_cfg$cfg.log" > "$out/bench_cfg$cfg.json"
cmd=(python3 "$root/bench.py" --config "$cfg" --steps 10 --warmup 2 --no-cpu-baseline)
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/cfg${cfg}_stats" -o run -- "${cmd[@]}" > "$out/cfg${cfg}_stats.log" 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex '^(void )?k_' --output-format csv -d "$out/cfg${cfg}_fetch" -o run -- "${cmd[@]}" > "$out/cfg${cfg}_fetch.log" 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex '^(void )?k_' --output-format csv -d "$out/cfg${cfg}_write" -o run -- "${cmd[@]}" > "$out/cfg${cfg}_write.log" 2>&1
rc=$?
cd "$root"
find "$out" -name '*kernel_trace.csv' -delete 2>/dev/null
grep '^{"metric' "$out/cfg${cfg}_stats.log" > "$out/cfg${cfg}_bench_under_rocprof.json"
echo "profile_r03 cfg$cfg: exit $rc"; head -c 600 "$out/bench_cfg$cfg.json"; echo
exit $rc
