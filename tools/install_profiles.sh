#!/bin/bash
# Copy the outputs of tools/final_runs.sh <P> (merged back into gpurun_out/) into profiles/ (tracked).  usage: tools/install_profiles.sh r04
set -eu
P=${1:-r06}; O=gpurun_out/final_$P
cd "$(dirname "$0")/.."
cp $O/pmc.json profiles/pmc.json
cp gpurun_out/$P/kernel_summary.txt profiles/${P}_kernel_summary.txt; cp gpurun_out/$P/kernel_stats.csv profiles/${P}_rocprofv3_kernel_stats.csv; cp gpurun_out/$P/traffic_raw.json profiles/${P}_traffic_raw.json
cp gpurun_out/${P}_serial/kernel_summary.txt profiles/${P}_serial_kernel_summary.txt; cp gpurun_out/${P}_serial/kernel_stats.csv profiles/${P}_serial_rocprofv3_kernel_stats.csv
cp gpurun_out/${P}_tight/kernel_summary.txt profiles/${P}_tight_kernel_summary.txt; cp gpurun_out/${P}_tight/traffic_raw.json profiles/${P}_tight_traffic_raw.json; cp gpurun_out/${P}_tight_serial/kernel_summary.txt profiles/${P}_tight_serial_kernel_summary.txt
for T in $P ${P}_tight; do cat gpurun_out/$T/pmc_sq1.txt gpurun_out/$T/pmc_sq2.txt gpurun_out/$T/pmc_tcc.txt gpurun_out/$T/pmc_tcp.txt gpurun_out/$T/pmc_fetch.txt gpurun_out/$T/pmc_write.txt gpurun_out/$T/pmc_grbm.txt | python -c "
import sys
keep = False
for line in sys.stdin:
    if not line.startswith(' '):
        keep = line.startswith(('k_', 'void k_', '__amd_rocclr_fill'))
    if keep:
        sys.stdout.write(line)
" > profiles/${T}_pmc.txt; done
cp $O/bench.json profiles/${P}_bench.json; cp $O/bench_fused.json profiles/${P}_fused_bench.json; cp $O/configs.txt profiles/${P}_configs.txt; cp $O/scaling_proxy.txt profiles/${P}_scaling_proxy.txt; cp $O/modes.txt profiles/${P}_modes.txt; cp $O/iter_bench.txt profiles/${P}_iter_bench.txt; cp $O/remesh_probe.txt profiles/${P}_remesh_probe.txt; cp $O/recon_trend_gpu.txt profiles/${P}_recon_trend_gpu_remesh.txt; cp $O/recon_trend_host.txt profiles/${P}_recon_trend_host_remesh.txt; cp $O/recon_monkey_144views.txt profiles/${P}_recon_monkey_144views.txt; cp $O/bench_2rank_gloo.json profiles/${P}_bench_2rank_gloo_one_gpu.json; cp $O/trace_repeat.txt profiles/${P}_trace_repeat.txt; cp $O/step_timeline_tight.txt profiles/${P}_step_timeline_tight.txt; cp $O/bind_det.txt profiles/${P}_bind_det.txt; cp $O/bench_8rank_gloo.json profiles/${P}_bench_8rank_gloo_one_gpu.json
grep -v 'amdgpu.ids' $O/gputest.log | tail -n 40 > profiles/${P}_gputest.log
cp $O/soak.txt profiles/${P}_soak.txt
cp $O/remesh_kernels.txt profiles/${P}_remesh_kernels.txt; cp $O/recon_error_map.txt profiles/${P}_recon_error_map.txt
