#!/bin/bash
# Produces the files committed under profiles/ for one round tag:  bash tools/profile_round.sh r1_d   (on the GPU box)
set -e
tag=${1:-rX}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python bench.py > $out/${tag}_bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/stats.err
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/${tag}_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmcF -- python bench.py --no-cpu-baseline --steps 5 > /dev/null 2> $out/pmcF.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmcW -- python bench.py --no-cpu-baseline --steps 5 > /dev/null 2> $out/pmcW.err
python tools/pmc_traffic.py $out/pmcF $out/pmcW > $out/${tag}_traffic.json
python bench.py --math f16 --no-cpu-baseline > $out/${tag}_bench_f16.json 2>> $out/bench.err
python bench.py --math f32 --batch 256 --no-cpu-baseline > $out/${tag}_bench_f32_b256.json 2>> $out/bench.err
python bench.py --batch 256 --no-cpu-baseline > $out/${tag}_bench_f16x3_b256.json 2>> $out/bench.err
python bench.py --workload c5 --steps 3 --warmup 1 > $out/${tag}_bench_c5_b128.json 2>> $out/bench.err
rm -rf $out/stats $out/pmcF $out/pmcW
ls -la $out
