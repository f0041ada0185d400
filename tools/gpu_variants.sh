#!/bin/bash
# usage: tools/gpu_variants.sh "<what> <iters> <mode>" variant1 variant2 ...   ('base' = the product library)
mkdir -p gpurun_out
args=$1; shift
out=gpurun_out/variants.log; : > $out
for round in 1 2; do
for v in "$@"; do
  if [ $v == base ]; then unset DFM_LIB_PATH; else export DFM_LIB_PATH=$PWD/deepfm_amd/lib/variants/lib_$v.so; fi
  echo -n "[$v] " >> $out
  timeout -k 10 120 python tools/time_layers.py $args 2>&1 | grep -v amdgpu.ids >> $out || { echo "FAILED $v" >> $out; cat $out; exit 1; }
done; done
cat $out
