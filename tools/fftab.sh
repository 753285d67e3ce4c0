#!/bin/bash
# A/B of the packed DC/Nyquist column (NF_FFT_PACK_NYQ) on half-stored real transforms
out=$1
for reg in 0 1; do for spec in "4096 4096 1 1" "4096 4096 0 2" "4096 4096 0 1" "2048 2048 1 1" "840 1440 1 2" "1031 1032 1 1"; do
  echo "== NF_FFT_PACK_NYQ=$reg $spec" >> $out
  NF_FFT_PACK_NYQ=$reg timeout -k 10 60 ./tools/fftbench $spec 20 2 >> $out 2>&1 || exit 1
done; done
echo done >> $out
