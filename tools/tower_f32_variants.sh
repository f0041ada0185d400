#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/tower_f32_variants.log; : > $out
for v in base g1 g2 g4 g8 g16 g31 base; do
  if [ $v == base ]; then unset DFM_LIB_PATH; else export DFM_LIB_PATH=$PWD/deepfm_amd/lib/variants/lib_$v.so; fi
  echo -n "[$v] " >> $out
  timeout -k 10 120 python tools/time_tower_f32.py 100 2>&1 | grep -v amdgpu.ids >> $out || { echo "FAILED $v" >> $out; cat $out; exit 1; }
done
cat $out
