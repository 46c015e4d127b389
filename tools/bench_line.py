"""Condense a bench.py JSON line (stdin) into one row: fps, ms, per-kernel (TFLOP/s, ms/frame), ray-march ms."""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else ""
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
ks = {k[-12:]: (round(v["tflops"], 1), round(v["ms_per_frame"], 2)) for k, v in d.get("kernels", {}).items()}
print(tag, round(d["value"], 1), round(d["ms_per_step"], 2), ks, round(d.get("raymarch", {}).get("ms_per_frame", 0), 3))
