#!/bin/bash
# old against fused line filter, rows then columns: digests side by side (must agree), time per call
set -o pipefail
for axis in 0 1; do
  NF_LF_FUSED=0 NF_LF_FUSED_COLS=0 timeout -k 10 120 python tools/linefilt_ab.py $axis > gpurun_out/lf_old.txt 2> gpurun_out/lf_old.err || { tail -5 gpurun_out/lf_old.err; exit 1; }
  NF_LF_FUSED=1 NF_LF_FUSED_COLS=1 timeout -k 10 120 python tools/linefilt_ab.py $axis > gpurun_out/lf_new.txt 2> gpurun_out/lf_new.err || { tail -5 gpurun_out/lf_new.err; exit 1; }
  echo "axis $axis (old, fused)"
  paste -d'\n' gpurun_out/lf_old.txt gpurun_out/lf_new.txt
done
