#!/bin/bash
# Run GPU steps one after another on the GPU box; a step that was KILLED at its time limit (rc 124 / 137) ends the call - no
# further GPU step is started after a hang.  usage: tools/gpu_steps.sh <out-prefix> "<secs> <name> <command...>" ...
prefix=$1; shift
mkdir -p "$(dirname "$prefix")"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for spec in "$@"; do
  secs=${spec%% *}; rest=${spec#* }; name=${rest%% *}; cmd=${rest#* }
  echo "== $name: $cmd" | tee -a "${prefix}_progress.log"
  start=$(date +%s)
  timeout -k 10 "$secs" bash -c "$cmd" > "${prefix}_${name}.out" 2> "${prefix}_${name}.err"
  rc=$?
  echo "== $name rc=$rc $(( $(date +%s) - start )) s" | tee -a "${prefix}_progress.log"
  tail -n 6 "${prefix}_${name}.out"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name was killed at its limit: stopping" | tee -a "${prefix}_progress.log"; exit $rc; fi
done
exit 0
