set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_deriv_marginal.py tests/test_gpu_differential.py tests/test_gpu_hess.py -x -q -k "not 600_taxon" 2>&1 | tail -4 || exit 1
for v in 2 6; do
  ARBPLF_UP_NODES=$v timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/rb_cfg3_$v.json || exit 1
  ARBPLF_UP_NODES=$v timeout -k 10 200 python bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/rb_cfg2_$v.json || exit 1
done
