#!/bin/bash
# Regenerates this round's committed measurements on the GPU box (run from the
# repo root): kernel-trace stats + bench line per workload, PMC traffic, sweeps.
set -e
R=${1:-r01}
for wl in c4 c2 c3 c1 c5shard; do
  extra=""; [ $wl = c4 ] || extra="--no-cpu-baseline"
  steps=50; [ $wl = c3 ] && steps=10
  profiles/run_prof.sh $wl --workload $wl --steps $steps --warmup 5 $extra > gpurun_out/prof_$wl.txt
  cp gpurun_out/prof/${wl}_kernel_stats.csv gpurun_out/${R}_${wl}_kernel_stats.csv
  timeout -k 10 300 python3 bench.py --workload $wl --steps $steps --warmup 5 $extra > gpurun_out/${R}_${wl}_bench.json
  echo "$wl done"
done
timeout -k 10 600 python3 profiles/collect_pmc_traffic.py gpurun_out/${R}_pmc_traffic.json
SWEEP_BATCHES=1,2,3,4,8,9,16,32,33,64,65,96,128,129,192,256,384,512,1024,2048,4096 timeout -k 10 200 python3 profiles/sweep_batch.py 1000000 1024 bf16 10 > gpurun_out/${R}_sweep_bf16_1Mx1024.json
SWEEP_BATCHES=1,2,8,32,33,64,65,96,128,129,192,256,384,512,1024 timeout -k 10 300 python3 profiles/sweep_batch.py 10000000 1024 bf16 10 > gpurun_out/${R}_sweep_bf16_10Mx1024.json
SWEEP_BATCHES=1,32,64,96,128,192,256 timeout -k 10 300 python3 profiles/sweep_batch.py 12500000 768 bf16 10 > gpurun_out/${R}_sweep_bf16_12p5Mx768.json
rm -f gpurun_out/${R}_query_data_dependence.jsonl
for b in 32 64 128; do timeout -k 10 300 python3 profiles/query_data_dependence.py medical-rag_amd/libmedtutor_retrieval.so 10000000 1024 $b >> gpurun_out/${R}_query_data_dependence.jsonl; done
timeout -k 10 200 python3 profiles/sweep_batch.py 1000000 768 fp32 10 > gpurun_out/${R}_sweep_fp32_1Mx768.json
timeout -k 10 300 python3 profiles/host_io_latency.py > gpurun_out/${R}_host_io_c4.json
timeout -k 10 100 python3 profiles/host_io_latency.py 4032 4096 fp32 1 1 > gpurun_out/${R}_host_io_c1_top1.json
timeout -k 10 100 python3 profiles/host_io_latency.py 4032 4096 fp32 34 10 > gpurun_out/${R}_host_io_c1_b34.json
echo all done
