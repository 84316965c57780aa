#!/usr/bin/env python3
"""A collector-shaped loop over the env, the way scripts/train.py drives the reference
(SyncDataCollector with return_same_td=True, frames_per_batch = num_envs * train_every): a stand-in
policy reads the same keys MAPPOPolicy reads, the rollout is stacked into pre-allocated [T, E, ...]
storage, and the advantage normalisation runs in its data-parallel form (one all-gather per rollout).

    python examples/rollout.py --envs 65536 --tp            # one GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/rollout.py

The policy is a fixed random linear map (there is no learner in this repository)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hns_amd  # noqa: E402,F401
from hns_amd import config, sharding  # noqa: E402
from hns_amd.env import HideAndSeek  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=16384, help="envs per GPU")
    ap.add_argument("--rollouts", type=int, default=5)
    ap.add_argument("--train-every", type=int, default=64)
    ap.add_argument("--tp", action="store_true", help="algo.use_TP_net: 1 (the reference's default)")
    args = ap.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(os.environ.get("HNS_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    torch.cuda.set_device(local)
    cfg = config.make_cfg({"cylinder": {"max_num": 8, "min_num": 8}, "env": {"num_envs": args.envs},
                           "sim": {"device": f"cuda:{local}"}}, algo={"use_TP_net": int(args.tp)})
    env = HideAndSeek(cfg, env_index_offset=rank * args.envs)
    env.set_seed(0)
    E, A, T = env.num_envs, env.num_agents, args.train_every
    td = env.reset()
    D = td[("agents", "observation", "state_self")].shape[-1]
    n_in = D + 3 * (A - 1) + 5 * env.obs_max_cylinder
    W = torch.randn(n_in, 4, device=env.device) * 0.3          # the stand-in policy
    obs_buf = torch.empty(T, E, A, n_in, device=env.device)
    rew_buf = torch.empty(T, E, A, device=env.device)
    done_buf = torch.empty(T, E, dtype=torch.bool, device=env.device)

    def flat_obs(t):
        o = t[("agents", "observation")]
        return torch.cat([o["state_self"].reshape(E, A, -1), o["state_others"].reshape(E, A, -1), o["cylinders"].reshape(E, A, -1)], dim=-1)

    cur = td
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(args.rollouts):
        for t in range(T):
            x = flat_obs(cur)
            obs_buf[t] = x
            step_td = env.rand_step_input(torch.tanh(x @ W))
            nxt = env.step(step_td)["next"]
            rew_buf[t] = nxt[("agents", "reward")].squeeze(-1)
            done_buf[t] = nxt["done"].squeeze(-1)
            if bool(done_buf[t].any()):                          # episodes are lock-step: reset exactly the done envs
                rtd = env.rand_step_input()
                rtd.set("_reset", done_buf[t])
                cur = env.reset(rtd)
            else:
                cur = nxt
        adv = rew_buf - rew_buf.mean()                          # placeholder for GAE: what matters here is the global normalisation
        adv_n, success = sharding.normalise_advantages(adv, env.stats["success"])
        if rank == 0:
            print(f"rollout {r}: reward mean {float(rew_buf.mean()):+.3f}  |adv| mean {float(adv_n.abs().mean()):.3f}  "
                  f"global success {success:.3f}")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(f"{world} x {E} envs, {args.rollouts} rollouts of {T} steps incl. policy + storage: "
              f"{world * E * A * T * args.rollouts / dt:.3e} agent-steps/s")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
