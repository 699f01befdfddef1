#!/usr/bin/env python3
"""One-line digest of a bench.py JSON line (stdin): per-side kernel times."""
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
o = json.loads(sys.stdin.read())
r = o.get("roofline", {})
print(tag, "ms/step", round(o["ms_per_step"], 2), "x", round(r.get("x_side_ms", 0), 2), "theta", round(r.get("theta_side_ms", 0), 2),
      "reduce_x", round(r.get("reduce_kernel_ms_x_side", 0), 3), "rmse", o.get("rmse"))
