#!/usr/bin/env python3
"""Step + predictor (algo.use_TP_net: 1) with the batch as shards on two HIP streams: does the HBM-bound step kernel of one shard hide
under the issue-bound predictor of the other?  (VERDICT r3 #2b.)  Every variant steps the same 65 536 envs:

  whole        one env, one stream: hns_step, then hns_tp_observe                        (what env.step does)
  halves       two shards, two streams, nothing between them                               (both steps first, then both predictors)
  staggered    shard B's step waits for shard A's step: step(B) runs beside tp(A)
  quarters     four shards on two streams, staggered the same way (A0 B0 A1 B1)

usage: python tools/tp_overlap_lab.py [envs] [steps]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import hns_amd  # noqa: E402,F401
from hns_amd import config  # noqa: E402
from hns_amd.env import HideAndSeek  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")


def make(n, offset):
    cfg = config.make_cfg({"num_agents": 3, "cylinder": {"max_num": 8, "min_num": 8}, "env": {"num_envs": n, "max_episode_length": 50000},
                           "sim": {"device": "cuda:0"}}, algo={"use_TP_net": 1})
    env = HideAndSeek(cfg, env_index_offset=offset)
    env.set_seed(0)
    env.reset()
    return env


def run(name, shards, streams, stagger):
    envs = [make(E // shards, i * (E // shards)) for i in range(shards)]
    acts = [torch.randn(E // shards, 3, 4, device=dev) for _ in range(shards)]
    st = [torch.cuda.Stream(dev) for _ in range(streams)] if streams > 1 else [torch.cuda.current_stream(dev)]
    evs = [torch.cuda.Event() for _ in range(shards)]
    torch.cuda.synchronize()

    def one():
        # shard i runs on stream i % streams; with `stagger` its step waits for the step of shard i - 1
        for i, env in enumerate(envs):
            s = st[i % len(st)]
            with torch.cuda.stream(s):
                if stagger and i > 0:
                    s.wait_event(evs[i - 1])
                rc = env._lib.hns_step(env._env, acts[i].data_ptr(), C.c_void_p(s.cuda_stream))
                assert rc == 0
                evs[i].record(s)
                env._tp_observe()
    for _ in range(20):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        one()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / N * 1e6
    print(f"{name:10s} {shards} shard(s) on {len(st)} stream(s){' staggered' if stagger else '':10s}: {us:7.1f} us per step of {E} envs  = {E * 3 / us * 1e6:.3e} agent-steps/s", flush=True)
    for env in envs:
        assert env.check_finite()
    del envs


run("whole", 1, 1, False)
run("halves", 2, 2, False)
run("staggered", 2, 2, True)
run("quarters", 4, 2, True)
run("eighths", 8, 2, True)
