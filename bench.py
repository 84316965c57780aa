#!/usr/bin/env python3
"""bench.py — env agent-steps/s of the fused HideAndSeek step on MI355X.

Workload = BASELINE.json configs[2]: HideAndSeek 3 pursuers / 1 evader, 8 cylinders
(k-nearest + line-of-sight sensing), 65 536 envs per GPU, synthetic N(0,1) policy outputs
resident in HBM.  A "step" is one `hns_step` over the whole env batch (+ the `hns_reset` launch
at the natural 1/800 episode boundary).  Multi-GPU: one process per GPU (torchrun), contiguous
env-index shards, weak scaling; the only collective is one RCCL all-gather of 5 fp64 values per
64-step rollout (the advantage-normalisation moments named by north_star).

Prints ONE JSON line (rank 0).  `roofline` is measured live: every 32nd step launch inside the
timed region is bracketed by hipEvents on the launch stream (hns_enable_timing).
`cpu_baseline` (N=1 only) times the CPU oracle — test infrastructure, never the product — on a
bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
NUM_STATS = 24


def algorithmic_bytes_per_env(A, C, k, S=NUM_STATS, NT=1):
    """mandatory fp32 traffic of one env-step, one read + one write, no temporaries.
    Two-evader extension (NT=2): + the second evader's position read (12 B), position + velocity written (24 B)
    and 4 more values in every state_self row (16 B per pursuer)."""
    base = A * (232 + 4 * (20 + 3 * (A - 1) + 5 * k) + 4) + 4 * (3 + 3 * C + 1 + S) + 4 * (7 + S) + 1
    return base + (NT - 1) * (12 + 24 + 16 * A)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--agents", type=int, default=3)
    ap.add_argument("--cylinders", type=int, default=8)
    ap.add_argument("--targets", type=int, default=1, help="evaders per env: 1 = the reference, 2 = BASELINE config 5's extension")
    ap.add_argument("--episode", type=int, default=800)
    ap.add_argument("--critic-state", action="store_true", help="also write the [E,A,20] centralised-critic state (critic_input: state)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=100)
    ap.add_argument("--time-every", type=int, default=32)
    ap.add_argument("--stream-groups", type=int, default=0,
                    help="optional extra leg (e.g. 2): the env batch as shards on separate HIP streams of one GPU, reported as "
                         "`stream_shards`; off by default so that a profile of the default command holds whole-batch launches only")
    ap.add_argument("--group-steps", type=int, default=1000)
    ap.add_argument("--tp-steps", type=int, default=300,
                    help="extra untimed-in-`value` leg: steps with the trajectory predictor in the observation "
                         "(algo.use_TP_net: 1, the reference's default config), reported as `tp_mode`; 0 = skip")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm; HNS_DIST_BACKEND=gloo only for smoke-testing the N>1 path on a 1-GPU box
        dist.init_process_group(os.environ.get("HNS_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    import hns_amd  # noqa: F401
    from hns_amd import abi, config
    if not os.path.exists(abi.library_path()):            # fresh checkout: the .so files are git-ignored
        if rank == 0:
            import __graft_entry__
            __graft_entry__.build()
        if world > 1:
            dist.barrier()
    from hns_amd.env import HideAndSeek

    E, A, C, K = args.envs, args.agents, args.cylinders, 3
    cfg = config.make_cfg({"num_agents": A, "num_targets": args.targets, "cylinder": {"max_num": C, "min_num": C, "obs_max_cylinder": K},
                           "env": {"num_envs": E, "max_episode_length": args.episode},
                           "sim": {"device": f"cuda:{local_rank}"}})
    env = HideAndSeek(cfg, headless=True, env_index_offset=rank * E, write_critic_state=args.critic_state)
    env.set_seed(0)
    env.reset()
    lib, henv = env._lib, env._env
    import ctypes as Cx
    stream = torch.cuda.current_stream(device)
    sptr = Cx.c_void_p(stream.cuda_stream)

    # synthetic policy outputs: a ring of pre-generated N(0,1) action batches resident in HBM
    R = 8
    gen = torch.Generator(device=device).manual_seed(1000 + rank)
    actions = [torch.randn(E, A, 4, generator=gen, device=device) for _ in range(R)]
    aptr = [Cx.c_void_p(a.data_ptr()) for a in actions]
    done_ptr = Cx.c_void_p(env._bufs["done"].data_ptr())
    reward = env._bufs["reward"]
    success = env.stats["success"]
    from hns_amd import sharding
    rollout = int(cfg.algo.get("train_every", 64))
    progress = {"t": 0}

    def run(n):
        for _ in range(n):
            i = progress["t"]
            rc = lib.hns_step(henv, aptr[i % R], sptr)
            assert rc == 0, lib.hns_last_error()
            progress["t"] = i + 1
            if (i + 1) % args.episode == 0:          # lock-step episodes: every env is done now
                rc = lib.hns_reset(henv, done_ptr, Cx.c_uint64(env.seed), sptr)
                assert rc == 0, lib.hns_last_error()
            if world > 1 and (i + 1) % rollout == 0:
                # per-rollout moments for advantage normalisation (learning/mappo.py:391-396 made
                # data-parallel) + the success rate of the curriculum (hideandseek.py:1012-1015):
                # ONE all-gather of 5 fp64 values per rank over RCCL/xGMI
                sharding.allgather_moments(sharding.local_moments(reward, success))

    def sync():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    run(args.warmup)
    sync()
    lib.hns_enable_timing(henv, args.time_every)
    t0 = time.perf_counter()
    run(args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    lib.hns_enable_timing(henv, 0)
    kernel_ms, n_samples = env.kernel_ms()
    if world > 1:
        tt = torch.tensor([elapsed], device=device if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    assert torch.isfinite(env._bufs["reward"]).all(), "non-finite reward"
    total_agent_steps = world * E * A * args.steps
    value = total_agent_steps / elapsed
    b_env = algorithmic_bytes_per_env(A, C, K, NT=args.targets)
    roofline = None
    traffic = None
    try:   # HBM bytes per launch from the PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), see profiles/
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        key = f"hns_step_kernel<{A}>|E{E}|C{C}|k{K}|critic_state_{'on' if args.critic_state else 'off'}"
        traffic = tj.get(key, {}).get("traffic_bytes_per_launch")
    except Exception:  # noqa: BLE001
        pass
    # achievable HBM bandwidth on this box ("measure achievable with a device copy kernel and report both")
    copy_gbs = None
    try:
        src = torch.empty(256 * 1024 * 1024 // 4, dtype=torch.float32, device=device).normal_()
        dst = torch.empty_like(src)
        for _ in range(3):
            dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize(device)
        copy_gbs = round(20 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
        del src, dst
    except Exception:  # noqa: BLE001
        pass
    if kernel_ms > 0:
        achieved = b_env * E / (kernel_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "kernel": f"hns_step_kernel<{A}>", "kernel_us": round(kernel_ms * 1e3, 2), "samples": n_samples,
                    "bytes_per_launch": b_env * E, "device_copy_GBs": copy_gbs,
                    "frac_of_device_copy": round(achieved / copy_gbs, 4) if copy_gbs else None}

    # secondary leg ("use_TP_net=1 reported separately"): step + hns_tp_observe
    tp_mode = None
    if args.tp_steps > 0 and world == 1 and args.targets == 1:
        cfg_tp = config.make_cfg({"num_agents": A, "cylinder": {"max_num": C, "min_num": C, "obs_max_cylinder": K},
                                  "env": {"num_envs": E, "max_episode_length": args.episode},
                                  "sim": {"device": f"cuda:{local_rank}"}}, algo={"use_TP_net": 1})
        env_tp = HideAndSeek(cfg_tp, headless=True, write_critic_state=args.critic_state)
        env_tp.set_seed(0)
        env_tp.reset()                              # binds + packs the predictor's parameters
        h2 = env_tp._env

        def run_tp(n):
            for i in range(n):
                assert lib.hns_step(h2, aptr[i % R], sptr) == 0, lib.hns_last_error()
                assert lib.hns_tp_observe(h2, 0, sptr) == 0, lib.hns_last_error()
        run_tp(20)
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        run_tp(args.tp_steps)
        torch.cuda.synchronize(device)
        dt_tp = time.perf_counter() - t1
        tp_mode = {"value": round(E * A * args.tp_steps / dt_tp, 1), "unit": "agent-steps/s", "steps": args.tp_steps,
                   "ms_per_step": round(dt_tp / args.tp_steps * 1e3, 5),
                   "what": "hns_step + hns_tp_observe (window shift, LSTM(16->64)x10 + FC on the matrix cores, 35-value rows)"}
        del env_tp

    # secondary leg: the same 65 536 envs as G shards on G HIP streams of this GPU (the multi-GPU sharding applied
    # inside one GPU).  Shards are independent, so the tail of one shard's launch — workgroups draining their stores —
    # overlaps the load burst and the arithmetic of the others; an asynchronous (double-buffered) collector gets this
    # rate.  Not the headline: the per-launch roofline above needs serial launches to mean anything.
    streams_mode = None
    if args.stream_groups > 1 and world == 1 and args.targets == 1 and E % args.stream_groups == 0:
        G, Eg = args.stream_groups, E // args.stream_groups
        cfg_g = config.make_cfg({"num_agents": A, "cylinder": {"max_num": C, "min_num": C, "obs_max_cylinder": K},
                                 "env": {"num_envs": Eg, "max_episode_length": args.episode},
                                 "sim": {"device": f"cuda:{local_rank}"}})
        shards = []
        for gidx in range(G):
            sh = HideAndSeek(cfg_g, headless=True, env_index_offset=gidx * Eg, write_critic_state=args.critic_state)
            sh.set_seed(0)
            sh.reset()
            shards.append(sh)
        gstreams = [torch.cuda.Stream(device) for _ in range(G)]
        gptr = [Cx.c_void_p(st.cuda_stream) for st in gstreams]
        gact = [[Cx.c_void_p(a[gidx * Eg:(gidx + 1) * Eg].data_ptr()) for a in actions] for gidx in range(G)]
        torch.cuda.synchronize(device)

        def run_groups(n):
            for i in range(n):
                for gidx in range(G):
                    assert lib.hns_step(shards[gidx]._env, gact[gidx][i % R], gptr[gidx]) == 0, lib.hns_last_error()
        run_groups(100)
        torch.cuda.synchronize(device)
        t2 = time.perf_counter()
        run_groups(args.group_steps)
        torch.cuda.synchronize(device)
        dt_g = time.perf_counter() - t2
        ms_g = dt_g / args.group_steps * 1e3
        streams_mode = {"groups": G, "value": round(E * A * args.group_steps / dt_g, 1), "unit": "agent-steps/s", "steps": args.group_steps,
                        "ms_per_step": round(ms_g, 5), "hbm_frac": round(b_env * E / (ms_g * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "what": f"the same {E} envs as {G} shards of {Eg} on {G} HIP streams (no join between steps)"}
        del shards

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import numpy as np
        import hns_oracle as O
        host = O.alloc_buffers(env.hcfg)
        O.reset(env.hcfg, host, None, 0, 0)
        act = np.random.default_rng(0).standard_normal((E, A, 4)).astype(np.float32)
        O.step(env.hcfg, host, act)
        c0 = time.perf_counter()
        for _ in range(args.cpu_steps):
            O.step(env.hcfg, host, act)
        cdt = time.perf_counter() - c0
        one_core = E * A * args.cpu_steps / cdt
        # the same sample on the host cores (envs are independent: OpenMP over the env loop).  os.cpu_count()
        # can exceed what the container may use, so the thread count is the best of a short probe
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        best_n, best_rate = 1, one_core
        for n in [2, 4, 8, 16, 32, 64, 128, 256]:
            if n > avail:
                break
            O.set_threads(n)
            O.step(env.hcfg, host, act)
            p0 = time.perf_counter()
            for _ in range(6):
                O.step(env.hcfg, host, act)
            rate = E * A * 6 / (time.perf_counter() - p0)
            if rate > best_rate:
                best_n, best_rate = n, rate
        ncores = best_n
        O.set_threads(ncores)
        m0 = time.perf_counter()
        for _ in range(args.cpu_steps):
            O.step(env.hcfg, host, act)
        mdt = time.perf_counter() - m0
        O.set_threads(1)
        cpu_baseline = {"value": round(E * A * args.cpu_steps / mdt, 1), "unit": "agent-steps/s", "cores": ncores,
                        "kind": "port", "one_core_value": round(one_core, 1),
                        "sample": f"{args.cpu_steps} steps of the same {E}-env workload with the C oracle "
                                  f"(oracle/hns_oracle.c): {ncores} threads (best of a probe up to {avail}) {mdt:.1f} s, 1 thread {cdt:.1f} s"}

    if rank == 0:
        out = {
            "metric": "env agent-steps/sec at 65536 envs, HideAndSeek 3v1; 1/2/4/8 GPU",
            "value": round(value, 1), "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"HideAndSeek {A}v{args.targets}, {C} random cylinders + LOS/k-nearest sensing, "
                                   f"{E} envs per GPU (BASELINE configs[2])",
                       "num_envs_per_gpu": E, "num_agents": A, "num_targets": args.targets, "num_cylinders": C, "obs_max_cylinder": K,
                       "episode_length": args.episode, "critic_state_output": args.critic_state,
                       "sharding": f"contiguous env slices x{world}",
                       "collective": "1 all-gather of 5 fp64 per 64-step rollout" if world > 1 else "none"},
            "env_frames_per_s": round(value / A, 1),
            "roofline": roofline, "cpu_baseline": cpu_baseline, "tp_mode": tp_mode, "stream_shards": streams_mode,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
