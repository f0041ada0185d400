#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/tower_variants.log; : > $out
for v in base x6a1 x6a2 x6a4 x6a8 x6a15 base; do
  if [ $v == base ]; then unset DFM_LIB_PATH; else export DFM_LIB_PATH=$PWD/deepfm_amd/lib/variants/lib_$v.so; fi
  echo -n "[$v] " >> $out
  timeout -k 10 120 python tools/time_tower_kernels.py 200 2>&1 | grep -v amdgpu.ids >> $out || { echo "FAILED $v" >> $out; cat $out; exit 1; }
done
cat $out
