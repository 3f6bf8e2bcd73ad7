import os, sys, json, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
r = subprocess.run([sys.executable, "bench.py", "--workload", "c5", "--math", "f16x3g", "--steps", "3", "--warmup", "2", "--no-traffic",
                    "--no-cpu-baseline"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
print(sys.argv[1] if len(sys.argv) > 1 else "", "step %.1f ms" % d["ms_per_step"],
      [(k["name"], round(k["avg_us"] / 1e3, 2)) for k in d["kernels"] if k["name"].startswith("finish")])
