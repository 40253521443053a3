"""Run bench.py under a set of environment variants and print ms/step + the profiled kernel's mean launch time (A/B tool)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variants = [v.split(",") for v in sys.argv[1].split(";")]          # "A=1,B=2;A=0"
kernels = sys.argv[2].split(",")
for var in variants:
    env = dict(os.environ)
    for kv in var:
        if kv:
            k, v = kv.split("=")
            env[k] = v
    for kern in kernels:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--no-cpu-baseline",
                              "--roofline-kernel", kern], env=env, capture_output=True, text=True).stdout.strip().splitlines()
        d = json.loads(out[-1])
        print(var, kern, "ms/step", d["ms_per_step"], "kernel us", d["roofline"]["avg_launch_us"], "TF", d["roofline"]["achieved"], flush=True)
