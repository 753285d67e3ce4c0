#!/bin/bash
out=gpurun_out/sweep3.log; : > $out
run() { echo "== $*" >> $out; env "${@:1:$#-1}" timeout -k 10 60 ./tools/fftbench ${!#} 1 5 2 >> $out 2>&1 || exit 1; }
run NF_X=1 "4096 4096 0"
for c in 1 2 4; do for th in 256 512 1024; do run NF_FFT_COLS_PER_WG=$c NF_FFT_COL_THREADS=$th "4096 4096 0"; done; done
run NF_X=1 "4096 4096 1"
for c in 1 2; do for th in 256 512 1024; do run NF_FFT_COLS_PER_WG=$c NF_FFT_COL_THREADS=$th "4096 4096 1"; done; done
echo done >> $out
