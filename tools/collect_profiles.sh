#!/bin/bash
# Copy the summaries of tools/profile.sh runs (gpurun_out/final, gpurun_out/final_fused) into profiles/.
# usage: tools/collect_profiles.sh <prefix>     e.g. r01_final
set -eu
P=${1:-r01_final}
cd "$(dirname "$0")/.."
for m in "" "_fused"; do
  S=gpurun_out/final$m
  cp $S/kernel_summary.txt profiles/${P}${m}_kernel_summary.txt
  cp $S/kernel_stats.csv profiles/${P}${m}_rocprofv3_kernel_stats.csv
  cp $S/traffic_raw.json profiles/${P}${m}_traffic_raw.json
  cat $S/pmc_sq1.txt $S/pmc_sq2.txt $S/pmc_tcc.txt $S/pmc_tcp.txt $S/pmc_fetch.txt $S/pmc_write.txt $S/pmc_grbm.txt > profiles/${P}${m}_pmc.txt
done
python tools/make_traffic.py gpurun_out/final/traffic_raw.json gpurun_out/final_fused/traffic_raw.json profiles/traffic.json
cp gpurun_out/bench_final.json profiles/${P}_bench.json 2>/dev/null || true
cp gpurun_out/bench_final_fused.json profiles/${P}_fused_bench.json 2>/dev/null || true
