#!/bin/bash
# tools/profile_sq.sh TAG CONFIG "COUNTERS" [extra bench.py args]
# One rocprofv3 counter pass (kernel tracing only, engine kernels only) of a short bench.py run; see profile_round.sh.
set -o pipefail
tag="$1"; cfg="$2"; ctr="$3"; shift 3
root="$(pwd)"; out="$root/gpurun_out/prof"; mkdir -p "$out"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc $ctr --kernel-include-regex '^(void )?k_' --output-format csv -d "$out/${tag}_cfg${cfg}_sq" -o run -- python3 "$root/bench.py" --config "$cfg" --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$out/${tag}_cfg${cfg}_sq.log" 2>&1
rc=$?
cd "$root"
find "$out" -name '*kernel_trace.csv' -delete 2>/dev/null
tail -n 2 "$out/${tag}_cfg${cfg}_sq.log" | cut -c1-300
echo "profile_sq $tag cfg$cfg: exit $rc"
exit $rc
