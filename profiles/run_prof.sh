#!/bin/bash
# usage: profiles/run_prof.sh TAG bench-args...   (run on the GPU box from the repo root)
# rocprofv3 kernel trace + stats of one bench.py run; prints this library's kernels only.
tag=$1; shift
out=gpurun_out/prof/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 bench.py "$@" > "$out.out" 2>&1 || { echo "rocprof run failed"; tail -5 "$out.out"; exit 1; }
stats=$(find "$out" -name "*kernel_stats.csv" | head -1)
python3 profiles/trim_stats.py "$stats" "gpurun_out/prof/${tag}_kernel_stats.csv"
rm -rf "$out"   # the raw trace is large; only the trimmed summary travels back
grep -E "^\"?Name|mtr::|rocclr" "gpurun_out/prof/${tag}_kernel_stats.csv" | cut -c1-230
grep -E "^\{" "$out.out" | cut -c1-1500
