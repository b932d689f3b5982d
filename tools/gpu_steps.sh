#!/bin/bash
# Runs the given steps ("name|seconds|command" arguments) one after another on the GPU box, each under its own
# `timeout -k 10`, output into gpurun_out/$OUT/<name>.log.  A failing step (assertion, non-zero exit) does not stop the
# sequence; a step that was KILLED (timeout 124 / 137, or died on a signal) does: nothing further touches the GPU.
OUT=${OUT:-misc}
mkdir -p gpurun_out/$OUT
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; secs=${rest%%|*}; cmd=${rest#*|}
  echo "=== $name (limit ${secs}s): $cmd"
  start=$(date +%s)
  timeout -k 10 "$secs" bash -c "$cmd" > gpurun_out/$OUT/$name.log 2>&1
  rc=$?
  echo "=== $name rc=$rc in $(( $(date +%s) - start ))s"
  tail -n 6 gpurun_out/$OUT/$name.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -ge 128 ]; then
    echo "=== $name was killed: stopping here"; exit $rc
  fi
done
exit 0
