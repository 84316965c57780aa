#!/usr/bin/env python3
"""Column tiles per workgroup of the weight-stationary predictor (csrc/hns_tp.hip: ws_envs, HNS_TP_TILES) by batch size, on one GPU box.

    python tools/tp_tiles.py [--envs=2048,4096,...] [--lib=path/to/libhns.so]

Per batch size and tiles in (auto, 1, 2, 4), one process each: `hns_tp_observe` alone (events on its stream, R rounds x N calls, median / min us), step + predictor
wall time per step, and a digest of the predictor's outputs (predictions, observation rows, window) after 12 steps of a seeded action stream — the digests of
every tile count must agree (the tile arithmetic does not depend on how many tiles a workgroup serves)."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(E, agents, obst=0, cyl=5):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import time
    import torch
    import hns_amd  # noqa: F401
    from hns_amd import config
    from hns_amd.env import HideAndSeek
    torch.manual_seed(0)                      # the predictor's parameters are drawn at construction
    cfg = config.make_cfg({"num_agents": agents, "use_obstacles": obst, "cylinder": {"max_num": cyl, "min_num": cyl}, "env": {"num_envs": E, "max_episode_length": 800}}, algo={"use_TP_net": 1})
    env = HideAndSeek(cfg, headless=True)
    env.set_seed(0)
    env.reset()
    dev = env.device
    gen = torch.Generator(device=dev).manual_seed(1234)
    tds = [env.rand_step_input(torch.randn(E, agents, 4, generator=gen, device=dev)) for _ in range(4)]
    for i in range(12):
        env.step(tds[i % 4])
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for k in ("pred", "obs_self", "history"):
        h.update(env._tp_bufs[k].cpu().numpy().tobytes())
    if os.environ.get("HNS_TP_TILES_DIGEST_ONLY"):          # tests/test_hip_tp.py: the digest alone
        print(json.dumps({"digest": h.hexdigest()[:16]}))
        return
    for i in range(20):
        env.step(tds[i % 4])
    obs, wall = [], []
    for _ in range(5):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            env._tp_observe()
        ev0.record()
        for _ in range(200):
            env._tp_observe()
        ev1.record()
        torch.cuda.synchronize()
        obs.append(ev0.elapsed_time(ev1) / 200 * 1e3)
        t0 = time.perf_counter()
        for i in range(200):
            env.step(tds[i % 4])
        torch.cuda.synchronize()
        wall.append((time.perf_counter() - t0) / 200 * 1e6)
    obs.sort(); wall.sort()
    print(json.dumps({"observe_us": round(obs[2], 2), "observe_us_min": round(obs[0], 2), "step_plus_observe_us": round(wall[2], 2), "digest": h.hexdigest()[:16]}))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(*(int(x) for x in sys.argv[2:6]))
        return
    args = dict(a[2:].split("=", 1) for a in sys.argv[1:] if a.startswith("--"))
    envs = [int(x) for x in args.get("envs", "2048,4096,8192,16384,32768,65536").split(",")]
    agents, obst, cyl = int(args.get("agents", 3)), int(args.get("obst", 0)), int(args.get("cyl", 5))
    frame = 7 + 3 * agents + (3 * cyl if obst else 0)
    print(f"# envs  tiles  observe us (min)  step+observe us  digest      ({agents} pursuers, frames of {frame} values = {(frame + 15) // 16} chunk(s))")
    for E in envs:
        digests = set()
        for tiles in ("auto", "1", "2", "4"):
            env = dict(os.environ)
            env.pop("HNS_TP_TILES", None)
            if tiles != "auto":
                env["HNS_TP_TILES"] = tiles
            if "lib" in args:
                env["HNS_LIBRARY"] = os.path.abspath(args["lib"])
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(E), str(agents), str(obst), str(cyl)], env=env, capture_output=True, text=True)
            line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
            if out.returncode != 0 or not line:
                print(f"{E:6d}  {tiles:>5s}  FAILED rc={out.returncode}\n{out.stdout[-800:]}\n{out.stderr[-2000:]}")
                continue
            r = json.loads(line[-1])
            digests.add(r["digest"])
            print(f"{E:6d}  {tiles:>5s}  {r['observe_us']:8.2f} ({r['observe_us_min']:7.2f})  {r['step_plus_observe_us']:8.2f}         {r['digest']}", flush=True)
        print(f"#        digests {'AGREE' if len(digests) == 1 else 'DIFFER: ' + str(sorted(digests))}", flush=True)


if __name__ == "__main__":
    main()
