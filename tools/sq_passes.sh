#!/bin/bash
# tools/sq_passes.sh TAG SITES "PASS1 counters" "PASS2 counters" ... : rocprofv3 counter passes (kernel tracing only, the
# engine's ll kernels only) of tools/profile_ll.py on BASELINE config $CFG (default 3).  A pass that fails because a counter
# name is unknown does not stop the later ones; a pass that is killed at its limit does.
tag="$1"; sites="$2"; shift 2
root="$(pwd)"; out="$root/gpurun_out/prof"; mkdir -p "$out"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
cfg="${CFG:-3}"; what="${WHAT:-ll}"; regex="${KREGEX:-^(void )?k_}"
i=0
for ctr in "$@"; do
  i=$((i+1))
  cd /tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr --kernel-include-regex "$regex" --output-format csv -d "$out/${tag}_p$i" -o run -- python3 "$root/tools/profile_ll.py" --config "$cfg" --sites "$sites" --steps 3 --what "$what" $EXTRA > "$out/${tag}_p$i.log" 2>&1
  rc=$?
  cd "$root"
  echo "pass $i ($ctr): exit $rc"; tail -n 2 "$out/${tag}_p$i.log" | cut -c1-200
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass killed at its limit: stopping"; exit $rc; fi
done
find "$out" -name '*kernel_trace.csv' -delete 2>/dev/null
python3 "$root/tools/sq_summary.py" "$out" "$tag" > "$root/gpurun_out/${tag}_counters.json"
cat "$root/gpurun_out/${tag}_counters.json" | head -c 3000
