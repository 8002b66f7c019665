#!/usr/bin/env python3
"""one-screen digest of a bench.py JSON line: python scripts/show_bench.py gpurun_out/bench_x.json"""
import json
import sys

d = json.load(open(sys.argv[1]))
print(f"{d['value']:.0f} {d['unit']}  {d['ms_per_step']:.4f} ms/step  host enqueue {d.get('host_enqueue_ms_per_step', 0):.4f} ms/step  n_gpus={d['n_gpus']}")
if "roofline" in d:
    r = d["roofline"]
    print(f"roofline: {r.get('kernel')} {r['achieved']:.0f}/{r['peak']:.0f} {r['unit']} frac {r['frac']:.3f} traffic {r['traffic']}")
print({k: round(v["avg_us"], 1) for k, v in d.get("kernels", {}).items()})
if "cpu_baseline" in d:
    print("cpu:", d["cpu_baseline"]["value"], d["cpu_baseline"]["unit"])
if "local_ba" in d:
    print("ba:", {k: v for k, v in d["local_ba"].items() if not isinstance(v, dict)})
