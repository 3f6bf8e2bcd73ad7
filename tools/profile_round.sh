#!/bin/bash
# Produces the files committed under profiles/ for one round tag:  bash tools/profile_round.sh r2_a   (on the GPU box)
# Every rocprofv3 pass is its own run (counters never share a run with --stats; FETCH_SIZE / WRITE_SIZE never share a pass).
set -e
tag=${1:-rX}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_driver_shape.json 2> $out/bench.err      # the driver's own command line
python bench.py --no-cpu-baseline > $out/${tag}_bench.json 2>> $out/bench.err                        # 200 steps
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline --no-traffic --no-secondary > $out/bench_under_rocprof.json 2> $out/stats.err
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/${tag}_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmcF -- python3 bench.py --traffic-child --steps 5 --warmup 1 > /dev/null 2> $out/pmcF.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmcW -- python3 bench.py --traffic-child --steps 5 --warmup 1 > /dev/null 2> $out/pmcW.err
python tools/pmc_traffic.py $out/pmcF $out/pmcW > $out/${tag}_traffic.json
rm -rf $out/stats $out/pmcF $out/pmcW
ls -la $out
