set -o pipefail
mkdir -p gpurun_out/r03
tools/profile_r03.sh 3 || exit 1
for s in 5000000 2500000 1250000; do
  python3 bench.py --sites $s --steps 20 --warmup 3 --no-cpu-baseline --deriv-steps 0 2>/dev/null | grep '^{"metric' > gpurun_out/r03/proxy_$s.json || exit 1
  python3 bench.py --sites $s --steps 20 --warmup 3 --no-cpu-baseline --deriv-steps 0 --dist 2>/dev/null | grep '^{"metric' > gpurun_out/r03/proxy_${s}_dist_world1.json || exit 1
  python3 bench.py --sites $s --steps 20 --warmup 3 --no-cpu-baseline --deriv-steps 0 --dist --sync-allreduce 2>/dev/null | grep '^{"metric' > gpurun_out/r03/proxy_${s}_dist_world1_sync.json || exit 1
done
for c in 2 4 5; do tools/profile_r03.sh $c || exit 1; done
# Frechet K1 (122 x 122 block exponentials) and one em-update step at config 5, marginal queries of configs 2..5
root=$(pwd)
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r03/em5_stats -o run -- python3 $root/tools/profile_ll.py --config 5 --sites 100000 --what em > $root/gpurun_out/r03/em5.log 2>&1) || exit 1
find gpurun_out/r03/em5_stats -name '*kernel_trace.csv' -delete 2>/dev/null
: > gpurun_out/r03/query_times.jsonl
for c in 2 3 4 5; do hs=0; [ $c = 2 ] && hs=1000000; [ $c = 3 ] && hs=500000; python3 tools/time_queries.py --config $c --hess-sites $hs 2>/dev/null | grep '^{' >> gpurun_out/r03/query_times.jsonl || exit 1; done
