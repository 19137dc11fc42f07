#!/bin/bash
# Copy the summaries of a tools/gpu_round.sh visit from gpurun_out/ (scratch) into profiles/ (tracked).  usage: tools/collect_profiles.sh <tag>
TAG=${1:-r02}
cp gpurun_out/${TAG}_bench.json profiles/${TAG}_bench.json
cp gpurun_out/${TAG}_kernel_stats.csv profiles/${TAG}_kernel_stats.csv
grep -vE "^/opt/amdgpu" gpurun_out/${TAG}_pytest_gpu.log | cut -c1-400 > profiles/${TAG}_pytest_gpu.log
cp gpurun_out/${TAG}_smoke.log profiles/${TAG}_smoke.log
cp gpurun_out/${TAG}_pmc_hbm.json profiles/${TAG}_pmc_hbm.json
cp gpurun_out/pmcsq_${TAG}.json profiles/${TAG}_pmc_sq.json
cp gpurun_out/${TAG}_icub_bench.json profiles/${TAG}_icub_bench.json
cp gpurun_out/${TAG}_hands_bench.json profiles/${TAG}_hands_bench.json
for f in icub_steady.json icub_push_soak.json icub_kernel_trace_tail.txt icub_kernel_stats.csv tail_probe.json; do [ -f gpurun_out/${TAG}_$f ] && cp gpurun_out/${TAG}_$f profiles/${TAG}_$f; done
[ -f gpurun_out/pmc_icub_${TAG}.json ] && cp gpurun_out/pmc_icub_${TAG}.json profiles/${TAG}_pmc_icub.json
[ -f gpurun_out/pmc_icub_hbm_${TAG}.json ] && cp gpurun_out/pmc_icub_hbm_${TAG}.json profiles/${TAG}_pmc_icub_hbm.json
cp gpurun_out/${TAG}_bench2.json profiles/${TAG}_bench_2ranks_one_device.json
[ -f gpurun_out/complex_breakdown.json ] && cp gpurun_out/complex_breakdown.json profiles/${TAG}_complex_breakdown.json
[ -f gpurun_out/r02a_parity_report.json ] && cp gpurun_out/r02a_parity_report.json profiles/${TAG}_parity_report_hip.json
ls -la profiles | grep ${TAG}
