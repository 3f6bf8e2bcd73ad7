#!/bin/bash
# Produces the files committed under profiles/ for one round tag:  bash tools/profile_round.sh r4_a   (on the GPU box)
# Every rocprofv3 pass is its own run (counters never share a run with --stats; FETCH_SIZE / WRITE_SIZE never share a pass).
set -e
tag=${1:-rX}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_driver_shape.json 2> $out/bench.err      # the driver's own command line
python bench.py --no-cpu-baseline --no-c5 > $out/${tag}_bench.json 2>> $out/bench.err              # 200 steps
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline --no-traffic --no-secondary > $out/bench_under_rocprof.json 2> $out/stats.err
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/${tag}_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmcF -- python3 bench.py --traffic-child --steps 5 --warmup 1 > /dev/null 2> $out/pmcF.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmcW -- python3 bench.py --traffic-child --steps 5 --warmup 1 > /dev/null 2> $out/pmcW.err
python tools/pmc_traffic.py $out/pmcF $out/pmcW > $out/${tag}_traffic.json
rm -rf $out/stats $out/pmcF $out/pmcW
# forward only (stash-less: the fused front end) -- kernel stats and HBM traffic, fused and unfused
for f in 1 0; do
  export WGNN_FUSED_FWD=$f
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/fs$f -- python3 tools/fwd_only.py 20 > /dev/null 2> $out/fs$f.err
  cp $(ls $out/fs$f/*/*kernel_stats.csv | head -1) $out/${tag}_fwd_only_fused${f}_kernel_stats.csv
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fF$f -- python3 tools/fwd_only.py 5 > /dev/null 2> $out/fF$f.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/fW$f -- python3 tools/fwd_only.py 5 > /dev/null 2> $out/fW$f.err
  python tools/pmc_traffic.py $out/fF$f $out/fW$f > $out/${tag}_fwd_only_fused${f}_traffic.json
  rm -rf $out/fs$f $out/fF$f $out/fW$f
done
unset WGNN_FUSED_FWD
# BASELINE configs[4] (4096 stations, CSR, H = 12288, B = 128): kernel stats of 3 steps, strict and mixed
for m in f16x3 f16x3g; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/c5$m -- python3 bench.py --workload c5 --math $m --steps 3 --warmup 2 --no-traffic --no-cpu-baseline > $out/${tag}_bench_c5_$m.json 2> $out/c5$m.err
  cp $(ls $out/c5$m/*/*kernel_stats.csv | head -1) $out/${tag}_c5_${m}_kernel_stats.csv
  rm -rf $out/c5$m
done
for spec in "f16x3 fp32" "f16x3g fp32" "f32 fp32" "f16 bf16"; do
  set -- $spec
  MATH=$1 IO=$2 python tools/kernel_times.py > $out/${tag}_kernel_times_$1_$2.txt 2>/dev/null
done
ls -la $out
