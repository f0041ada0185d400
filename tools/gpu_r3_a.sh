#!/bin/bash
# round 3, call A: new parity tests, then baseline vs no-SLP CIN timings
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "train_golden or fullsize or cin_cfg3" -s > gpurun_out/t1.log 2>&1
echo "pytest rc $?" >> gpurun_out/t1.log
grep -E "passed|failed|fullsize|Error" gpurun_out/t1.log | tail -30
for v in base noslp; do
  if [ $v == base ]; then unset DFM_LIB_PATH; else export DFM_LIB_PATH=$PWD/deepfm_amd/lib/variants/lib_$v.so; fi
  for i in 1 2; do timeout -k 10 120 python tools/time_layers.py cin 20 split >> gpurun_out/cin_variants.log 2>&1 || exit 1; done
  echo "^^^ $v" >> gpurun_out/cin_variants.log
done
cat gpurun_out/cin_variants.log
