#!/usr/bin/env python3
"""Condense bench.py's JSON line: value, ms/step, kernel us, roofline fraction."""
import json, sys
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1])
r = d.get("roofline") or {}
print("value %.3e agent-steps/s  ms/step %.5f  kernel_us %s  frac %s" % (d["value"], d["ms_per_step"], r.get("kernel_us"), r.get("frac")))
