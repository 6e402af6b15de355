#!/bin/bash
# tools/step.sh NAME [--timeout SECONDS] -- COMMAND...
# One step of a GPU-box session (run under gpurun): stdout and stderr of COMMAND go, together and unbuffered, to
# gpurun_out/NAME.log (always kept, also when the step fails or is killed), the exit status to gpurun_out/NAME.rc,
# and the last lines of the log to the terminal.  Steps are chained with && so that nothing runs after a failure.
name="$1"; shift
limit=1100
if [ "$1" = "--timeout" ]; then limit="$2"; shift 2; fi
[ "$1" = "--" ] && shift
mkdir -p gpurun_out
log="gpurun_out/$name.log"
echo "== step $name: $* (limit ${limit}s)" | tee "$log"
export PYTHONUNBUFFERED=1 AMD_LOG_LEVEL=${AMD_LOG_LEVEL:-1}     # level 1 keeps the runtime's own error lines (e.g. "Memory access fault by GPU ...")
timeout -k 10 "$limit" "$@" >> "$log" 2>&1
rc=$?
echo "$rc" > "gpurun_out/$name.rc"
echo "== step $name: exit $rc" >> "$log"
tail -n 15 "$log"
if [ $rc -ne 0 ]; then cp "$log" "gpurun_out/$name.FAILED.log"; fi
exit $rc
