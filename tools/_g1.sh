set -e
bash tools/profile_round.sh r2_c > /dev/null
out=gpurun_out/r2_c
python bench.py --math f16 --no-traffic > $out/r2_c_bench_f16.json 2>> $out/bench.err
python bench.py --math f16 --io bf16 --no-traffic > $out/r2_c_bench_f16_bf16io.json 2>> $out/bench.err
python bench.py --math f32 --batch 256 --no-traffic > $out/r2_c_bench_f32_b256.json 2>> $out/bench.err
python bench.py --math f32 --no-traffic --no-secondary --steps 50 > $out/r2_c_bench_f32_b4096.json 2>> $out/bench.err
bash tools/sq_counters.sh r2_c_f32 --math f32
cp gpurun_out/r2_c_f32/r2_c_f32_sq_counters.txt $out/r2_c_sq_counters_f32.txt
ls $out
