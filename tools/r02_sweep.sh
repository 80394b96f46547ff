#!/bin/bash
# Frames-per-launch sweep of the two fused kernels at 1080p and 4K (HIP-event launch times via bench.py).
set -e
out=gpurun_out/sweep
mkdir -p $out
for g in ${GROUPS_HD:-1 2 4 8 16}; do
  timeout -k 10 120 python3 bench.py --steps 5 --warmup 2 --prewarm 1 --batch 64 --group $g --no-cpu-baseline > $out/hd_g$g.json 2> $out/hd_g$g.err
  echo "1080p group $g: $(grep -o 'avg launch[^,]*' $out/hd_g$g.json) $(grep -o '"value": [0-9.]*' $out/hd_g$g.json)"
done
for g in ${GROUPS_4K:-1 2 4 8}; do
  timeout -k 10 120 python3 bench.py --steps 5 --warmup 2 --prewarm 1 --batch 32 --group $g --height 2160 --width 3840 --qp 32 --no-cpu-baseline > $out/4k_g$g.json 2> $out/4k_g$g.err
  echo "4K group $g: $(grep -o 'avg launch[^,]*' $out/4k_g$g.json) $(grep -o '"value": [0-9.]*' $out/4k_g$g.json)"
done
