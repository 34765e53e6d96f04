#!/bin/bash
# Copy the summaries of tools/profile.sh runs into profiles/ (tracked).
# usage: tools/collect_profiles.sh <prefix> [<tag of the drop-in run> [<tag of the serialised run>]]     e.g. r02 r02 r02_serial
set -eu
P=${1:-r02}; T=${2:-$P}; S=${3:-}
cd "$(dirname "$0")/.."
cp gpurun_out/$T/kernel_summary.txt profiles/${P}_kernel_summary.txt
cp gpurun_out/$T/kernel_stats.csv profiles/${P}_rocprofv3_kernel_stats.csv
cp gpurun_out/$T/traffic_raw.json profiles/${P}_traffic_raw.json
cat gpurun_out/$T/pmc_sq1.txt gpurun_out/$T/pmc_sq2.txt gpurun_out/$T/pmc_tcc.txt gpurun_out/$T/pmc_tcp.txt gpurun_out/$T/pmc_fetch.txt gpurun_out/$T/pmc_write.txt gpurun_out/$T/pmc_grbm.txt \
  | python -c "
import sys
keep = False
for line in sys.stdin:
    if not line.startswith(' '):
        keep = line.startswith(('k_', 'void k_', '__amd_rocclr_fill'))
    if keep:
        sys.stdout.write(line)
" > profiles/${P}_pmc.txt
python tools/make_pmc_json.py gpurun_out/$T profiles/pmc.json dropin
if [ -n "$S" ]; then
  cp gpurun_out/$S/kernel_summary.txt profiles/${P}_serial_kernel_summary.txt
  cp gpurun_out/$S/kernel_stats.csv profiles/${P}_serial_rocprofv3_kernel_stats.csv
fi
