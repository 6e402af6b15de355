set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_fused_asm.py tests/test_gpu_ll.py tests/test_gpu_shard.py tests/test_gpu_fullsize.py tests/test_gpu_group.py -x -q 2>&1 | tail -3 || exit 1
for c in 3 2; do timeout -k 10 100 python bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline --deriv-steps 0 2>/dev/null | grep '^{"metric' > gpurun_out/fs_cfg$c.json || exit 1; done
timeout -k 10 100 python bench.py --sites 1250000 --steps 30 --warmup 5 --no-cpu-baseline --deriv-steps 0 --dist 2>/dev/null | grep '^{"metric' > gpurun_out/fs_proxy.json || exit 1
