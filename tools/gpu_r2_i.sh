#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd $root
timeout -k 10 1000 python3 -m pytest tests/test_gpu_fused_tower.py tests/test_gpu_models_step.py tests/test_gpu_train_golden.py tests/test_gpu_fullsize.py tests/test_gpu_dp_rehearsal.py tests/test_gpu_packed_pipeline.py tests/test_gpu_checkpoint.py -q -x > $out/r2i_pytest.log 2>&1
rc=$?
tail -8 $out/r2i_pytest.log
if [ $rc -ne 0 ]; then grep -E "^E " $out/r2i_pytest.log | head -20; fi
if [ $rc -ge 124 ]; then echo "pytest timed out"; exit $rc; fi
cd /tmp && export TMPDIR=/tmp
python3 $root/bench.py --steps 300 --no-extra-configs > $out/bench_i.json 2> $out/bench_i.err || { tail -20 $out/bench_i.err; exit 1; }
python3 -c "
import json
d=json.loads(open('$out/bench_i.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['final_loss']); print(json.dumps(d['roofline'])[:900])"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_i_t -o run -- python3 $root/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extra-configs > $out/bench_i_t.json 2> $out/bench_i_t.err
python3 $root/tools/kstats.py $out/bench_i_t 120 12
python3 -c "
import json
d=json.loads(open('$out/bench_i_t.json').read().strip().splitlines()[-1])
print('under rocprof: events avg', d['roofline']['avg_launch_us'], 'min', d['roofline']['min_launch_us'])"
