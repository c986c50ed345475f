#!/bin/bash
# usage: profiles/pmc_pass.sh TAG "COUNTER COUNTER ..." KERNEL_SUBSTRING bench-args...
# One rocprofv3 --pmc pass (its own run, no trace domains) over one bench.py
# run; prints the per-dispatch means of the counters for one kernel.
tag=$1; counters=$2; kern=$3; shift 3
out=gpurun_out/pmc/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc $counters --output-format csv -d "$out" -- python3 bench.py "$@" > "$out.out" 2>&1
csv=$(find "$out" -name "*counter_collection.csv" | head -1)
[ -n "$csv" ] || { echo "no counter file"; tail -5 "$out.out"; exit 1; }
python3 profiles/pmc_summary.py "$csv" "$kern" | tee "gpurun_out/pmc/${tag}.json"
rm -rf "$out"   # the raw counter dump is large; only the summary travels back
echo
