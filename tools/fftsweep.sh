#!/bin/bash
# sweep of the FFT geometry overrides (NF_FFT_* environment variables read by Plan2D::init)
out=$1
run() { echo "== $*" >> $out; env "${@:1:$#-1}" timeout -k 10 60 ./tools/fftbench ${!#} 4 5 >> $out 2>&1 || exit 1; }
for kb in 40 64 128; do for tb in 256 512; do run NF_FFT_ROW_LDS_KB=$kb NF_FFT_ROW_TB=$tb "4096 4096 0"; done; done
for kb in 64 128; do for tb in 256 512; do run NF_FFT_ROW_LDS_KB=$kb NF_FFT_ROW_TB=$tb "4096 4096 1"; done; done
for c in 2 4 8; do for th in 256 512; do run NF_FFT_COLS_PER_WG=$c NF_FFT_COL_THREADS=$th "4096 4096 0"; done; done
for c in 1 2; do for th in 256 512; do run NF_FFT_COLS_PER_WG=$c NF_FFT_COL_THREADS=$th "4096 4096 1"; done; done
for c in 4 8 16; do for th in 128 256; do run NF_FFT_COLS_PER_WG=$c NF_FFT_COL_THREADS=$th "840 1440 0"; done; done
for c in 2 4 8; do for th in 128 256; do run NF_FFT_COLS_PER_WG=$c NF_FFT_COL_THREADS=$th "840 1440 1"; done; done
for kb in 16 24 40 64; do run NF_FFT_ROW_LDS_KB=$kb "840 1440 0"; run NF_FFT_ROW_LDS_KB=$kb "840 1440 1"; done
echo done >> $out
