#!/bin/bash
# tools/profile_stats.sh TAG CONFIG [extra bench.py args]: rocprofv3 --kernel-trace --stats of a short bench.py run,
# engine kernels listed on stdout (see profile_round.sh for the full three-pass version)
set -o pipefail
tag="$1"; cfg="$2"; shift 2
root="$(pwd)"; out="$root/gpurun_out/prof"; mkdir -p "$out"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_cfg${cfg}_stats" -o run -- python3 "$root/bench.py" --config "$cfg" --steps 5 --warmup 2 --no-cpu-baseline "$@" > "$out/${tag}_cfg${cfg}_stats.log" 2>&1
rc=$?
cd "$root"
find "$out" -name '*kernel_trace.csv' -delete 2>/dev/null
grep -E '^"(void )?k_' "$out/${tag}_cfg${cfg}_stats/run_kernel_stats.csv" | awk -F'",' '{print substr($1,2,60) " | " $2}' | head -8
echo "profile_stats $tag cfg$cfg: exit $rc"
exit $rc
