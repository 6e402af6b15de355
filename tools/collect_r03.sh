#!/bin/bash
# tools/collect_r03.sh : copies the summaries of tools/session_prof.sh (gpurun_out/r03/) into profiles/ and derives the
# PMC traffic files bench.py reads (profiles/traffic_cfgN.json, traffic_cfgN_deriv.json) with the source hashes of this tree.
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"; in="$root/gpurun_out/r03"; out="$root/profiles"
declare -A LLK=([2]=k_ll_fused4_v4 [3]=k_ll_fused4_v4 [4]=k_ll_vec [5]=k_ll_mfma)
declare -A LLKEY=([2]=k_ll_fused4 [3]=k_ll_fused4 [4]=k_ll_vec [5]=k_ll_mfma)
declare -A DK1=([2]=k_down_fused4 [3]=k_down_fused4 [4]=k_down_vec [5]=k_down_fused_mfma)
declare -A DK2=([2]=k_up4_nodes [3]=k_up4_nodes [4]=k_up_vec [5]=k_up_mfma)
declare -A DKEY=([2]=deriv4 [3]=deriv4 [4]=deriv_vec [5]=deriv_mfma)
for c in 2 3 4 5; do
  [ -f "$in/bench_cfg$c.json" ] || continue
  cp "$in/bench_cfg$c.json" "$out/r03_bench_cfg$c.json"
  cp "$in/cfg${c}_bench_under_rocprof.json" "$out/r03_cfg${c}_bench_under_rocprof.json"
  cp "$(ls "$in/cfg${c}_stats"/*kernel_stats.csv | head -1)" "$out/r03_cfg${c}_kernel_stats.csv"
  S=$(python3 -c "import json;print(json.load(open('$in/bench_cfg$c.json'))['config']['sites_per_gpu'])")
  SD=$(python3 -c "import json;print(json.load(open('$in/bench_cfg$c.json'))['deriv']['sites_per_gpu'])")
  python3 "$root/tools/pmc_traffic.py" --kernel "${LLK[$c]}" --source-key "${LLKEY[$c]}" --sites "$S" --fetch "$in/cfg${c}_fetch" --write "$in/cfg${c}_write" \
      --config "BASELINE config $c, bench.py --config $c --steps 10" --out "$out/traffic_cfg$c.json" | tail -1
  python3 "$root/tools/pmc_traffic.py" --kernel "${DK1[$c]}" --kernel "${DK2[$c]}" --source-key "${DKEY[$c]}" --sites "$SD" --fetch "$in/cfg${c}_fetch" --write "$in/cfg${c}_write" \
      --config "BASELINE config $c deriv leg, $SD sites" --out "$out/traffic_cfg${c}_deriv.json" | tail -1
done
for s in 5000000 2500000 1250000; do
  [ -f "$in/proxy_$s.json" ] || continue
  cp "$in/proxy_$s.json" "$out/r03_bench_cfg3_proxy_$s.json"
  cp "$in/proxy_${s}_dist_world1.json" "$out/r03_bench_cfg3_proxy_${s}_rccl_native_world1.json"
  cp "$in/proxy_${s}_dist_world1_sync.json" "$out/r03_bench_cfg3_proxy_${s}_rccl_native_world1_sync.json"
done
[ -f "$in/query_times.jsonl" ] && cp "$in/query_times.jsonl" "$out/r03_query_times.jsonl"
[ -d "$in/em5_stats" ] && cp "$(ls "$in/em5_stats"/*kernel_stats.csv | head -1)" "$out/r03_cfg5_em_update_100k_kernel_stats.csv"
echo "collected into profiles/"
