set -o pipefail
for s in 1250000 10000000; do
python3 bench.py --sites $s --steps 20 --warmup 3 --no-cpu-baseline --deriv-steps 0 --dist 2>gpurun_out/s14_err_$s.log | grep '^{"metric' > gpurun_out/s14_native_$s.json || { tail -5 gpurun_out/s14_err_$s.log; exit 1; }
python3 bench.py --sites $s --steps 20 --warmup 3 --no-cpu-baseline --deriv-steps 0 --dist --torch-allreduce 2>/dev/null | grep '^{"metric' > gpurun_out/s14_torch_$s.json || exit 1
python3 bench.py --sites $s --steps 20 --warmup 3 --no-cpu-baseline --deriv-steps 0 2>/dev/null | grep '^{"metric' > gpurun_out/s14_none_$s.json || exit 1
done
tail -3 gpurun_out/s14_err_1250000.log
