set -o pipefail
mkdir -p gpurun_out
(nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; lscpu | head -20; free -g | head -2) > gpurun_out/s1_host.txt 2>&1
tools/step.sh s1_shard --timeout 500 -- python -m pytest tests/test_gpu_shard.py -x -q &&
tools/step.sh s1_bench10M --timeout 300 -- python bench.py --steps 20 --warmup 3 &&
tools/step.sh s1_bench5M --timeout 200 -- python bench.py --steps 20 --warmup 3 --sites 5000000 --no-cpu-baseline --deriv-steps 0 &&
tools/step.sh s1_bench2p5M --timeout 200 -- python bench.py --steps 20 --warmup 3 --sites 2500000 --no-cpu-baseline --deriv-steps 0 &&
tools/step.sh s1_bench1p25M --timeout 200 -- python bench.py --steps 20 --warmup 3 --sites 1250000 --no-cpu-baseline --deriv-steps 0 &&
tools/step.sh s1_dist1 --timeout 200 -- python bench.py --steps 20 --warmup 3 --sites 1250000 --no-cpu-baseline --deriv-steps 0 --dist
