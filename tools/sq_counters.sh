#!/bin/bash
# SQ / LDS / clock counters of the bench step's kernels (diagnosis aid):  bash tools/sq_counters.sh r2_a [bench.py args]   (GPU box)
tag=${1:-rX}
shift || true
extra="$@"      # further bench.py arguments, e.g. --math f32
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 -L > $out/counters_list.txt 2>&1 || true
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/sq$i -- python3 bench.py --traffic-child --steps 3 --warmup 1 $extra > /dev/null 2> $out/sq$i.err || echo "pass $i failed" >> $out/sq.err
done
python tools/pmc_summary.py $out/sq1 $out/sq2 > $out/${tag}_sq_counters.txt 2>> $out/sq.err || true
rm -rf $out/sq1 $out/sq2
