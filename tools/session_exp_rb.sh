set -o pipefail
tools/profile_r03.sh 4 || exit 1
: > gpurun_out/r03/query_times.jsonl
for c in 2 3 4 5; do hs=0; [ $c = 2 ] && hs=1000000; [ $c = 3 ] && hs=500000; python3 tools/time_queries.py --config $c --hess-sites $hs 2>/dev/null | grep '^{' >> gpurun_out/r03/query_times.jsonl || exit 1; done
