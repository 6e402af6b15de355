set -o pipefail
root=$(pwd)
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/hess_stats -o run -- python3 $root/tools/time_queries.py --config 3 --sites 100000 --hess-sites 50000 > $root/gpurun_out/hess_prof.log 2>&1
cd $root; find gpurun_out/hess_stats -name '*kernel_trace.csv' -delete; tail -2 gpurun_out/hess_prof.log
