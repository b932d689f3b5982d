#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "TRS_PRESORT_GRID_CAP=512" "TRS_PRESORT_GRID_CAP=4096" "TRS_PRESORT_GRID_CAP=256" "TRS_PRESORT_GRID_CAP=512" "TRS_PRESORT_GRID_CAP=4096"; do
  echo "== $v"
  env $v python3 $ROOT/bench.py --gpus 1 --steps 1024 --warmup 32 --no-cpu-baseline --no-pass --no-mlp --no-kernel-events | python3 -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('long us/step %.2f'%(d['ms_per_step']*1e3))
"
  env $v python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pass --no-mlp --no-kernel-events | python3 -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('short us/step %.2f'%(d['ms_per_step']*1e3))
"
done
for v in "TRS_PRESORT_GRID_CAP=512" "TRS_PRESORT_GRID_CAP=4096"; do
  echo "== rocprof $v"
  rm -rf /tmp/tr_ab
  env $v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_ab -- python3 $ROOT/bench.py --steps 1024 --warmup 32 --no-cpu-baseline --no-mlp --no-pass > /dev/null 2>&1
  f=$(find /tmp/tr_ab -name "*kernel_stats.csv" | head -1)
  grep -E "fwd_stage|epoch_refs|batch_flags" $f | cut -c1-200
done
