set -o pipefail
tools/step.sh s19_tests --timeout 900 -- python -m pytest tests/test_gpu_ll.py tests/test_gpu_expect.py tests/test_gpu_deriv_marginal.py -x -q || exit 1
python3 bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/s19_cfg5.json || exit 1
python3 bench.py --config 4 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/s19_cfg4.json || exit 1
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/s19_stats -o run -- python3 $OLDPWD/tools/profile_ll.py --config 5 --sites 100000 --what em > $OLDPWD/gpurun_out/s19_em5.log 2>&1; cd $OLDPWD
grep -E '^"(void )?k_' gpurun_out/s19_stats/run_kernel_stats.csv | cut -c1-110 | head -8
