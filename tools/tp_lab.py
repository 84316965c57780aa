#!/usr/bin/env python3
"""A/B bench of trajectory-predictor builds on one GPU box (run through gpurun).

    python tools/tp_lab.py [--rounds=5] [--agents=3] [--obst=0] name=path/to/libhns_x.so ...

Every variant runs in its own process (HNS_LIBRARY selects the build).  Per variant: `hns_tp_observe` alone, back to back,
bracketed by events on its stream (R rounds x N calls, median / min us), step + predictor wall time per step, and the
largest deviation of the observation rows / predictions from the CPU oracle's fp32 LSTM over 14 steps of a 512-env batch
with parameters at 1x and 3x the reference's initialisation (the parity gate is 1e-5).
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(args):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import time
    import numpy as np
    import torch
    import hns_amd  # noqa: F401
    from hns_amd import abi, config
    from hns_amd.env import HideAndSeek
    import hns_oracle as O
    A = int(args.get("agents", 3))
    obst = int(args.get("obst", 0))
    C = int(args.get("cyl", 8))

    def make(E, scale=1.0):
        cfg = config.make_cfg({"num_agents": A, "use_obstacles": obst, "cylinder": {"max_num": C, "min_num": C},
                               "env": {"num_envs": E, "max_episode_length": 800}}, algo={"use_TP_net": 1})
        env = HideAndSeek(cfg, headless=True)
        if scale != 1.0:
            with torch.no_grad():
                for p in env.TP.parameters():
                    p.mul_(scale)
        env.set_seed(0)
        env.reset()
        return env

    # ---- accuracy against the oracle ----
    errs = {}
    for scale in (1.0, 3.0):
        env = make(512, scale)
        g = torch.Generator(device="cpu").manual_seed(3)
        host = env.export_state()
        tpa = {k: v.cpu().numpy().copy() for k, v in env._tp_bufs.items() if k != "packed"}
        tpa["packed"] = np.zeros(16, np.uint8)
        for f, key in abi.TP_STATE_DICT_KEYS.items():
            tpa[f] = env.TP.state_dict()[key].detach().cpu().numpy().copy()
        tpa["history"][:] = 0
        O.tp_observe(env.hcfg, host, tpa, fill=True)
        e_rows = e_pred = 0.0
        for t in range(14):
            env.step(env.rand_step_input(torch.randn(512, A, 4, generator=g).to(env.device)))
            O.tp_observe(env.hcfg, env.export_state(), tpa, fill=False)
            e_rows = max(e_rows, float(np.abs(env._tp_bufs["obs_self"].cpu().numpy() - tpa["obs_self"]).max()))
            e_pred = max(e_pred, float(np.abs(env._tp_bufs["pred"].cpu().numpy() - tpa["pred"]).max()))
        errs[f"x{scale:g}"] = {"rows": e_rows, "pred": e_pred, "window": bool(np.array_equal(env._tp_bufs["history"].cpu().numpy(), tpa["history"]))}
        del env

    # ---- time ----
    E = int(args.get("envs", 65536))
    env = make(E)
    dev = env.device
    gen = torch.Generator(device=dev).manual_seed(1234)
    tds = [env.rand_step_input(torch.randn(E, A, 4, generator=gen, device=dev)) for _ in range(4)]
    for i in range(20):
        env.step(tds[i % 4])
    rounds, n = int(args.get("rounds", 5)), int(args.get("calls", 100))
    obs, wall = [], []
    for _ in range(rounds):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            env._tp_observe()
        ev0.record()
        for _ in range(n):
            env._tp_observe()
        ev1.record()
        torch.cuda.synchronize()
        obs.append(ev0.elapsed_time(ev1) / n * 1e3)
        t0 = time.perf_counter()
        for i in range(n):
            env.step(tds[i % 4])
        torch.cuda.synchronize()
        wall.append((time.perf_counter() - t0) / n * 1e6)
    obs.sort(); wall.sort()
    print(json.dumps({"observe_us_med": round(obs[len(obs) // 2], 2), "observe_us_min": round(obs[0], 2),
                      "step_plus_observe_us_med": round(wall[len(wall) // 2], 2), "err": errs}))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(dict(kv.split("=", 1) for kv in sys.argv[2:]))
        return
    common = [a[2:] for a in sys.argv[1:] if a.startswith("--")]
    variants = [a for a in sys.argv[1:] if not a.startswith("--")]
    for v in variants:
        name, path = v.split("=", 1)
        env = dict(os.environ, HNS_LIBRARY=os.path.abspath(path))
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", *common], env=env, capture_output=True, text=True)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        if out.returncode != 0 or not line:
            print(f"{name:14s} FAILED rc={out.returncode}\n{out.stdout[-1500:]}\n{out.stderr[-3000:]}")
            continue
        r = json.loads(line[-1])
        e = r["err"]
        print(f"{name:14s} observe {r['observe_us_med']:7.2f} us (min {r['observe_us_min']:7.2f})  step+observe {r['step_plus_observe_us_med']:7.2f} us  "
              f"err rows/pred x1 {e['x1']['rows']:.1e}/{e['x1']['pred']:.1e}  x3 {e['x3']['rows']:.1e}/{e['x3']['pred']:.1e}"
              + ("" if e['x1']['window'] and e['x3']['window'] else "  WINDOW DIFFERS FROM THE ORACLE"), flush=True)


if __name__ == "__main__":
    main()
