#!/bin/bash
# GPU box: what 100 more scalar instructions, 100 more vector instructions of the fast class (v_add_u32) and 100 of the
# slow class (v_max_u32) per loop iteration cost the persistent kernel (builds with -DMIRT_PROBE_*).
out=gpurun_out/probe_ab.txt
: > $out
for lib in libmirt.so libmirt_psalu.so libmirt_pvfast.so libmirt_pvslow.so libmirt.so; do
  MIRT_LIB=$PWD/ray_tracing_amd/$lib timeout -k 10 300 python scripts/ab.py --reps 2 ${CASES:-cornell random final million4} >> $out 2>&1 || exit 1
done
cat $out
