#!/bin/bash
# usage: profiles/dbg_run.sh "ENV=VAL ..." bench-args...   (GPU box, repo root)
# One bench.py run under timing-experiment switches; prints kernel span and step time.
# The planted-row check is expected to fail under switches that drop candidates.
envs=$1; shift
echo "== $envs $*"
env $envs timeout -k 10 150 python3 bench.py "$@" --no-cpu-baseline > gpurun_out/dbg_run.out 2> gpurun_out/dbg_run.err
rc=$?
[ $rc -ge 124 ] && { echo "bench run ended with $rc"; tail -3 gpurun_out/dbg_run.err; exit $rc; }
python3 profiles/kms.py < gpurun_out/dbg_run.out || { tail -3 gpurun_out/dbg_run.err; }
exit 0
