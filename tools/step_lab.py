#!/usr/bin/env python3
"""A/B bench of step-kernel builds on one GPU box (run through gpurun).

    python tools/step_lab.py name=path/to/libhns_x.so[:LAB_FLAGS] ...

Every variant runs in its own process (HNS_LIBRARY selects the build, HNS_LAB_FLAGS the ablation switches of a
-DHNS_LAB build), on the same seeded 65 536-env 3v1 / 8-cylinder batch: R rounds x N steps, per-launch kernel time from dispatch-bound hipEvents (hns_enable_timing) and wall time per step.
Prints one line per variant: median / min kernel us, wall us per step, and a digest of every state and output
buffer after a fixed 40-step prologue — equal digests = bit-identical results.
"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(args):
    sys.path.insert(0, ROOT)
    import ctypes as C
    import time
    import torch
    import hns_amd  # noqa: F401
    from hns_amd import config
    from hns_amd.env import HideAndSeek
    E, A, Cn = int(args.get("envs", 65536)), int(args.get("agents", 3)), int(args.get("cyl", 8))
    NT = int(args.get("targets", 1))
    cfg = config.make_cfg({"num_agents": A, "num_targets": NT, "cylinder": {"max_num": Cn, "min_num": Cn},
                           "env": {"num_envs": E, "max_episode_length": 800}})
    saved = os.environ.pop("HNS_LAB_FLAGS", None)        # reset + prologue always run the full kernel
    env = HideAndSeek(cfg, headless=True)
    env.set_seed(0)
    env.reset()
    lib, h = env._lib, env._env
    dev = env.device
    gen = torch.Generator(device=dev).manual_seed(1234)
    acts = [torch.randn(E, A, 4, generator=gen, device=dev) for _ in range(8)]
    sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ap = [C.c_void_p(a.data_ptr()) for a in acts]
    for i in range(40):
        assert lib.hns_step(h, ap[i % 8], sp) == 0
    torch.cuda.synchronize()
    dig = hashlib.sha256()
    for k in sorted(env._bufs):
        v = env._bufs[k]
        if v is not None and k != "state_drones":
            dig.update(v.cpu().numpy().tobytes())
    if saved is not None:
        os.environ["HNS_LAB_FLAGS"] = saved
    rounds, steps = int(args.get("rounds", 5)), int(args.get("steps", 300))
    ks, ws = [], []
    for _ in range(rounds):
        for i in range(30):
            lib.hns_step(h, ap[i % 8], sp)
        torch.cuda.synchronize()
        lib.hns_enable_timing(h, 4)
        t0 = time.perf_counter()
        for i in range(steps):
            lib.hns_step(h, ap[i % 8], sp)
        torch.cuda.synchronize()
        ws.append((time.perf_counter() - t0) / steps * 1e6)
        lib.hns_enable_timing(h, 0)
        ks.append(env.kernel_ms()[0] * 1e3)
    ks.sort(); ws.sort()
    print(json.dumps({"kernel_us_med": round(ks[len(ks) // 2], 2), "kernel_us_min": round(ks[0], 2),
                      "wall_us_med": round(ws[len(ws) // 2], 2), "wall_us_min": round(ws[0], 2), "digest": dig.hexdigest()[:16]}))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(dict(kv.split("=", 1) for kv in sys.argv[2:]))
        return
    common = [a[2:] for a in sys.argv[1:] if a.startswith("--")]
    variants = [a for a in sys.argv[1:] if not a.startswith("--")]
    rows = []
    for v in variants:
        name, spec = v.split("=", 1)
        parts = spec.split(":")
        path, flags = parts[0], (parts[1] if len(parts) > 1 else "")
        envv = dict(os.environ)
        for kv in (parts[2].split(",") if len(parts) > 2 and parts[2] else []):
            k, _, val = kv.partition("=")
            envv[k] = val
        if path:
            envv["HNS_LIBRARY"] = os.path.join(ROOT, path)
        envv.pop("HNS_LAB_FLAGS", None)
        if flags:
            envv["HNS_LAB_FLAGS"] = flags
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", *common], env=envv, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            print(f"{name:28s} FAILED rc={r.returncode} {r.stderr[-400:]}")
            continue
        d = json.loads(line[-1])
        rows.append((name, d))
        print(f"{name:28s} kernel med {d['kernel_us_med']:7.2f} min {d['kernel_us_min']:7.2f} us   wall med {d['wall_us_med']:7.2f} min {d['wall_us_min']:7.2f} us   digest {d['digest']}", flush=True)
    if rows:
        ref = rows[0][1]["digest"]
        print("bit-identical to %s: %s" % (rows[0][0], ", ".join(n for n, d in rows if d["digest"] == ref)))


if __name__ == "__main__":
    main()
