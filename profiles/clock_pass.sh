#!/bin/bash
# One rocprofv3 --pmc GRBM_GUI_ACTIVE pass per batch size over profiles/query_data_dependence.py (run from the repo root on
# the GPU box); profiles/clock_under_load.py turns busy cycles and durations into the clock held per query kind.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for spec in "128 score_topk_mfma_kernel" "32 score_topk_mfma_kernel" "1 scan_stream_kernel"; do
  set -- $spec
  out=gpurun_out/pmc/clock_b$1
  rm -rf $out; mkdir -p $out
  timeout -s KILL 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $out -- python3 profiles/query_data_dependence.py medical-rag_amd/libmedtutor_retrieval.so 10000000 1024 $1 > $out.out 2>&1
  csv=$(find $out -name "*counter_collection.csv" | head -1)
  head -1 $csv | cut -c1-400
  python3 profiles/clock_under_load.py $csv $2 | tee -a gpurun_out/r03_clock_under_load.jsonl
  rm -rf $out
done
