#!/bin/bash
# Run GPU steps one after another on the gpurun box; a step that TIMES OUT or is KILLED ends the call (no further GPU step is
# started behind a hung one); an ordinary non-zero exit (a failed assertion) is recorded and the next step runs.
#   bash tools/gpu_steps.sh OUTDIR "SECONDS|name|command" ...
out=$1; shift
mkdir -p "$out"
for spec in "$@"; do
  secs=${spec%%|*}; rest=${spec#*|}; name=${rest%%|*}; cmd=${rest#*|}
  echo "== $name (limit ${secs}s): $cmd" | tee -a "$out/steps.log"
  timeout -k 10 "$secs" bash -c "$cmd" > "$out/$name.txt" 2>&1
  rc=$?
  echo "== $name rc=$rc" | tee -a "$out/steps.log"
  tail -n 4 "$out/$name.txt"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "== $name timed out / was killed: stopping here" | tee -a "$out/steps.log"; exit $rc; fi
done
exit 0
