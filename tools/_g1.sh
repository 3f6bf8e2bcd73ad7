mkdir -p gpurun_out/g22
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "f32 or golden or big_tile or 32row or sweep" > gpurun_out/g22/t.log 2>&1; tail -2 gpurun_out/g22/t.log
timeout -k 10 300 python bench.py --math f32 --no-traffic --no-cpu-baseline --no-secondary --steps 50 > gpurun_out/g22/b4096.json 2>gpurun_out/g22/b.err
python - <<P
import json
d=json.loads(open("gpurun_out/g22/b4096.json").read().strip().splitlines()[-1])
print(d["value"],d["ms_per_step"],[(k["name"],k["avg_us"]) for k in d["kernels"] if "gru" in k["name"]])
P
