#!/bin/bash
# L2 behaviour of the configs[4] GEMMs: FETCH_SIZE and TCC hit / miss counters per kernel (separate rocprofv3 passes).
#   bash tools/exp/c5_pmc.sh OUTDIR     (GPU box)
out=$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
i=0
for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 bench.py --traffic-child --steps 1 --warmup 1 --workload c5 > /dev/null 2> $out/p$i.err || echo "pass $i ($set) failed" >> $out/pmc.err
done
python tools/pmc_summary.py $out/p1 $out/p2 $out/p3 > $out/c5_pmc.txt 2>> $out/pmc.err
rm -rf $out/p1 $out/p2 $out/p3
head -60 $out/c5_pmc.txt
