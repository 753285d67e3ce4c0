#!/bin/bash
out=$1
run() { echo "== $*" >> $out; env "${@:1:$#-1}" timeout -k 10 60 ./tools/fftbench ${!#} 4 5 >> $out 2>&1 || exit 1; }
for c in 2 4; do for th in 512 1024; do run NF_FFT_COLS_PER_WG=$c NF_FFT_COL_THREADS=$th "4096 4096 0"; done; done
for c in 1 2; do for th in 512 1024; do run NF_FFT_COLS_PER_WG=$c NF_FFT_COL_THREADS=$th "4096 4096 1"; done; done
run NF_FFT_ROW_TB=512 "4096 4096 0"
run NF_FFT_ROW_TB=512 "4096 4096 1"
run NF_FFT_ROW_TB=256 "4096 4096 1"
echo done >> $out
