#!/bin/bash
# Copy the summaries of a tools/gpu_visit5b.sh visit from gpurun_out/ (scratch) into profiles/ (tracked).  usage: tools/collect_profiles4.sh <tag>
TAG=${1:-r05}
c() { [ -f "$1" ] && cp "$1" "$2"; }
c gpurun_out/${TAG}_bench.json profiles/${TAG}_bench.json
c gpurun_out/${TAG}_kernel_stats.csv profiles/${TAG}_kernel_stats.csv
[ -f gpurun_out/${TAG}_pytest_gpu.log ] && grep -vE "^/opt/amdgpu" gpurun_out/${TAG}_pytest_gpu.log | cut -c1-400 > profiles/${TAG}_pytest_gpu.log
c gpurun_out/${TAG}_smoke.log profiles/${TAG}_smoke.log
c gpurun_out/${TAG}_pmc_hbm.json profiles/${TAG}_pmc_hbm.json
c gpurun_out/${TAG}_pmc_sq.json profiles/${TAG}_pmc_sq.json
c gpurun_out/${TAG}_small_shards.json profiles/${TAG}_small_shards.json
c gpurun_out/${TAG}_shard_sweep.json profiles/${TAG}_shard_sweep.json
c gpurun_out/${TAG}_step_kernels.txt profiles/${TAG}_step_kernels.txt
c gpurun_out/${TAG}_reset_time.json profiles/${TAG}_reset_time.json
c gpurun_out/${TAG}_16384_pmc_hbm.json profiles/${TAG}_pmc_hbm_16384.json
c gpurun_out/${TAG}_16384_pmc_sq.json profiles/${TAG}_pmc_sq_16384.json
c gpurun_out/${TAG}_kernel_stats_stationary_16384.csv profiles/${TAG}_kernel_stats_stationary_16384.csv
c gpurun_out/${TAG}_tail_probe.json profiles/${TAG}_tail_probe.json
c gpurun_out/${TAG}_icub_bench.json profiles/${TAG}_icub_bench.json
c gpurun_out/${TAG}_icub_steady.json profiles/${TAG}_icub_steady.json
c gpurun_out/${TAG}_icub_steady_kernels.json profiles/${TAG}_icub_steady_kernels.json
c gpurun_out/pmc_icub_${TAG}.json profiles/${TAG}_pmc_icub.json
c gpurun_out/pmc_icub_hbm_${TAG}.json profiles/${TAG}_pmc_icub_hbm.json
c gpurun_out/pmc_hands_${TAG}.json profiles/${TAG}_pmc_hands.json
c gpurun_out/${TAG}_hands_bench.json profiles/${TAG}_hands_bench.json
c gpurun_out/${TAG}_bench2.json profiles/${TAG}_bench_2ranks_one_device.json
c gpurun_out/${TAG}_bench8.json profiles/${TAG}_bench_8ranks_one_device.json
ls -la profiles | grep ${TAG}
