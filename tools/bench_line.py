#!/usr/bin/env python3
"""Condense bench.py's JSON line (STDIN): headline, every configuration leg, predictor mode, CPU baseline."""
import json, sys
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1])
r = d.get("roofline") or {}
print("headline  value %.3e agent-steps/s  ms/step %.5f  kernel_us %s (samples %s)  frac_kernel %s  step_us %s  frac_step_rate %s  frac of device copy %s  traffic %s" % (
    d["value"], d["ms_per_step"], r.get("kernel_us"), r.get("kernel_samples"), r.get("frac"), r.get("step_us"), r.get("frac_step_rate"), r.get("frac_of_device_copy"), r.get("traffic")))
for k, c in (d.get("configs") or {}).items():
    if k == "beyond_l3":
        for e, cc in c.items():
            rr = cc.get("roofline") or {}
            print("beyond_l3 %8s envs  value %.3e  ms/step %.5f  kernel_us %s  frac %s" % (e, cc["value"], cc["ms_per_step"], rr.get("kernel_us"), rr.get("frac")))
        continue
    rr = c.get("roofline") or {}
    line = "%-16s value %.3e  ms/step %.5f  kernel_us %s  frac_kernel %s  frac_step_rate %s" % (k, c["value"], c["ms_per_step"], rr.get("kernel_us"), rr.get("frac"), rr.get("frac_step_rate"))
    if "generator_ms_per_episode" in c:
        line += "\n                 generator ms per episode %s; steady batch %s ms; incl. generator at 800-step episodes %.3e (steady %.3e = %.2f of stepping)" % (
            c["generator_ms_per_episode"], c["generator_ms_task_batch_steady"], c["value_incl_generator_at_800_step_episodes"],
            c["value_incl_generator_steady_at_800_step_episodes"], c["value_incl_generator_steady_at_800_step_episodes"] / c["value"])
    print(line)
t = d.get("tp_mode")
if t:
    print("tp_mode   value %.3e  ms/step %.5f  observe_us %s  step_kernel_us %s  mfma frac %s" % (t["value"], t["ms_per_step"], t["observe_us"], t["step_kernel_us"], t["roofline"]["frac"]))
    rd = t.get("reference_default_batch")
    if rd:
        print("          at the reference's default batch (2 048 envs): value %.3e  ms/step %.5f  observe_us %s" % (rd["value"], rd["ms_per_step"], rd["observe_us"]))
c = d.get("cpu_baseline")
if c:
    print("cpu_baseline %.3e agent-steps/s on %d threads (1 thread %.3e)" % (c["value"], c["cores"], c["one_core_value"]))
    if c.get("cfg2"):
        print("cpu_baseline cfg2 (4 096 envs) %.3e on %d threads (1 thread %.3e)" % (c["cfg2"]["value"], c["cfg2"]["cores"], c["cfg2"]["one_core_value"]))
a = d.get("abi_rate")
if a:
    print("abi_rate  %.3e  ms/step %.5f" % (a["value"], a["ms_per_step"]))
