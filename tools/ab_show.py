"""stdin: one bench.py JSON line -> `TAG form kernel_us  other_form kernel_us ...  wall_us` (tools/ab_trees.sh)"""
import json
import sys

d = json.loads(sys.stdin.read())
r = d["roofline"]
rest = " ".join(f"{k} {v['kernel_us']:.3f}" for k, v in r.items() if isinstance(v, dict) and "kernel_us" in v)
print(sys.argv[1] if len(sys.argv) > 1 else "-", r["form"], f"{r['kernel_us']:.3f}", rest, "wall_us", f"{d['ms_per_step'] * 1000:.3f}", flush=True)
