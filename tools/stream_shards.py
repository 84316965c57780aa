#!/usr/bin/env python3
"""The 65 536-env batch as 1 / 2 / 4 independent shards on separate HIP streams of one GPU (no join between steps):
per-step time and aggregate throughput."""
import os, sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import torch, hns_amd
from hns_amd import config
from hns_amd.env import HideAndSeek
dev = torch.device("cuda", 0)
E, A, Cn = 65536, 3, 8
def mk(e, off):
    cfg = config.make_cfg({"num_agents": A, "cylinder": {"max_num": Cn, "min_num": Cn}, "env": {"num_envs": e, "max_episode_length": 800}})
    env = HideAndSeek(cfg, headless=True, env_index_offset=off, write_critic_state=False)
    env.set_seed(0); env.reset(); return env
for G in (1, 2, 4):
    envs = [mk(E // G, g * (E // G)) for g in range(G)]
    streams = [torch.cuda.Stream(dev) for _ in range(G)]
    acts = [[torch.randn(E // G, A, 4, device=dev) for _ in range(4)] for _ in range(G)]
    torch.cuda.synchronize()
    lib = envs[0]._lib
    def run(n):
        for i in range(n):
            for g in range(G):
                rc = lib.hns_step(envs[g]._env, C.c_void_p(acts[g][i % 4].data_ptr()), C.c_void_p(streams[g].cuda_stream))
                assert rc == 0
    run(200); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(2000); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"groups {G}: {dt/2000*1e6:.2f} us per 65536-env step, {E*A*2000/dt:.3e} agent-steps/s")
