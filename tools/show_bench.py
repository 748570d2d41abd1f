#!/usr/bin/env python3
"""Print the headline fields of a bench.py JSON line: python tools/show_bench.py gpurun_out/b.log"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["unit"], d["config"].get("last_step"))
r = d.get("roofline") or {}
print({k: (v["launches"], round(v["ms"], 1)) for k, v in (r.get("kernels") or {}).items()})
print({k: r.get(k) for k in ("kernel", "bound", "achieved", "peak", "frac", "traffic")})
if d.get("cpu_baseline"): print(d["cpu_baseline"])
