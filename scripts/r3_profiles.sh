#!/bin/bash
# GPU box: the rocprofv3 evidence of the four single-GPU configurations (scripts/profile_config.sh each), then the
# plain bench lines of the same commands. Everything under gpurun_out/prof_c<N>/ and gpurun_out/bench_c<N>.json.
cd $GRAFT_REPO_ROOT
for c in 2 3 5 1; do
  steps=3; [ $c = 5 ] && steps=20; [ $c = 1 ] && steps=50
  timeout -k 10 900 bash scripts/profile_config.sh c$c $c $steps > gpurun_out/prof_c$c.log 2>&1 || { echo "profile of config $c failed"; tail -5 gpurun_out/prof_c$c.log; exit 1; }
  echo "config $c profiled"
  timeout -k 10 300 python bench.py --config $c --steps $steps --warmup 1 > gpurun_out/bench_c$c.json 2> gpurun_out/bench_c$c.err || { echo "bench of config $c failed"; tail -5 gpurun_out/bench_c$c.err; exit 1; }
  tail -c 600 gpurun_out/bench_c$c.json; echo
done
