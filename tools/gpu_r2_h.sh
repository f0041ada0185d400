#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd $root
timeout -k 10 600 python3 -m pytest tests/test_gpu_attention.py tests/test_gpu_cin.py -q -x > $out/r2h_pytest.log 2>&1
rc=$?
tail -6 $out/r2h_pytest.log
if [ $rc -ge 124 ]; then echo "pytest timed out"; exit $rc; fi
cd /tmp && export TMPDIR=/tmp
time python3 $root/bench.py > $out/bench_h.json 2> $out/bench_h.err || { tail -20 $out/bench_h.err; exit 1; }
python3 -c "
import json
d=json.loads(open('$out/bench_h.json').read().strip().splitlines()[-1])
print(json.dumps({k:v for k,v in d.items() if k not in ('extra_configs','roofline')}, indent=0)[:1500])
print(json.dumps(d['roofline'], indent=0)[:1200])
for e in d.get('extra_configs',[]): print(json.dumps(e, indent=0))
"
