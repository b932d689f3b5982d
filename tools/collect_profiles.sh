#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/prof/ (copied to profiles/ afterwards):
#   bench lines (driver command, long window, c2 / c3 / c5), rocprofv3 --kernel-trace --stats summaries, PMC traffic
# usage: tools/collect_profiles.sh r03
R=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py"
echo "== bench lines"
$B --gpus 1 --steps 20 --warmup 5 > $O/${R}_bench_c4_driver_cmd.json 2> $O/err.log || exit 1
$B --gpus 1 --no-mlp > $O/${R}_bench_c4.json 2>> $O/err.log
$B --config c2 --no-mlp > $O/${R}_bench_c2.json 2>> $O/err.log
$B --config c5 --steps 64 --warmup 8 > $O/${R}_bench_c5.json 2>> $O/err.log
$B --config c3 --steps 32 --warmup 4 > $O/${R}_bench_c3.json 2>> $O/err.log
$B --config c1 --steps 256 --warmup 16 > $O/${R}_bench_c1.json 2>> $O/err.log
echo "== kernel stats"
for cfg in c4 c2 c5 c3; do
  steps=1024; warm=32
  [ $cfg = c5 ] && { steps=64; warm=8; }
  [ $cfg = c3 ] && { steps=32; warm=4; }
  rm -rf $O/trace_$cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$cfg -- python3 $ROOT/bench.py --config $cfg --steps $steps --warmup $warm --no-cpu-baseline --no-mlp > $O/${R}_bench_${cfg}_under_rocprof.json 2>> $O/err.log
  f=$(find $O/trace_$cfg -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" $O/${R}_bench_${cfg}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if ("trs::" in r["Name"] or "_GLOBAL__N" in r["Name"] or "kernel" in r["Name"].lower()) and "at::native" not in r["Name"]
        and "ROCPRIM_400001" not in r["Name"] and "fillBuffer" not in r["Name"] and "copyBuffer" not in r["Name"]]
if keep:
    w = csv.DictWriter(open(sys.argv[2], "w"), fieldnames=list(keep[0].keys()))
    w.writeheader(); w.writerows(keep)
PY
done
echo "== PMC traffic (separate passes)"
for cfg in c4 c2; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_${cfg}_$ctr
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_${cfg}_$ctr -- python3 $ROOT/bench.py --config $cfg --steps 256 --warmup 16 --no-cpu-baseline --no-kernel-events --no-pass --no-mlp > /dev/null 2>> $O/err.log
  done
  ff=$(find $O/pmc_${cfg}_FETCH_SIZE -name "*counter_collection.csv" | head -1)
  fw=$(find $O/pmc_${cfg}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  [ -n "$ff" ] && [ -n "$fw" ] && python3 $ROOT/tools/pmc_traffic.py --fetch $ff --write $fw --out $O/${R}_pmc_traffic_$cfg.json \
     --source "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on python bench.py --config $cfg --steps 256 --warmup 16 --no-cpu-baseline --no-kernel-events --no-pass --no-mlp, MI355X"
done
# keep the merged directory small: the raw traces stay on the box
rm -rf $O/trace_* $O/pmc_*
ls -la $O
