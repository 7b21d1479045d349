#!/bin/bash
# GPU box: tests with observation log, then the static-scratch footprint of the previous and the current library
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/observed_parity.jsonl
timeout -k 10 800 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t3.log 2>&1; rc=$?
echo "pytest rc $rc" >> gpurun_out/t3.log; tail -8 gpurun_out/t3.log
[ $rc -eq 0 ] || exit $rc
{ timeout -k 10 100 python scripts/static_footprint.py ray_tracing_amd/libmirt_r2.so && timeout -k 10 100 python scripts/static_footprint.py; } > gpurun_out/static_footprint.txt 2>&1
cat gpurun_out/static_footprint.txt
