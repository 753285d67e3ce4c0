#!/bin/bash
# column-tile geometry sweep at 4096^2 with the packed DC/Nyquist column (half-stored real transforms)
out=$1
run() { echo "== $*" >> $out; env "${@:1:$#-1}" timeout -k 10 60 ./tools/fftbench ${!#} 10 2 2>&1 | grep -E "cols|maxerr" >> $out || exit 1; }
for c in 1 2 4; do for th in 256 512 1024; do run NF_FFT_COLS_PER_WG=$c NF_FFT_COL_THREADS=$th "4096 4096 1 1"; done; done
for c in 2 4 8; do for th in 256 512 1024; do run NF_FFT_COLS_PER_WG=$c NF_FFT_COL_THREADS=$th "4096 4096 0 1"; done; done
echo done >> $out
