#!/usr/bin/env python3
"""bench.py — env agent-steps/s of the fused HideAndSeek step on MI355X.

Workload = BASELINE.json configs[2]: HideAndSeek 3 pursuers / 1 evader, 8 cylinders (k-nearest + line-of-sight
sensing), 65 536 envs per GPU, synthetic N(0,1) policy outputs resident in HBM.  A "step" is one `env.step(td)` of the
Python class ("time env.step") = one `hns_step` launch over the whole env batch, plus the `reset` at the
natural 1/800 episode boundary.  Multi-GPU: one process per GPU, contiguous env-index shards, weak scaling; the only
collective is one RCCL all-gather of 8 fp64 values per 64-step rollout (the advantage-normalisation moments named by
north_star).  `python bench.py --gpus N` launches its own N ranks when it was not started by torchrun.

Prints ONE JSON line (rank 0):
  * `value` / `ms_per_step`: the timed `env.step` loop; `abi_rate`: the same steps through the bare C ABI;
  * `roofline`: the step kernel, measured live.  `frac` = `frac_kernel` = algorithmic bytes per launch / `kernel_us`, the kernel's average launch duration:
    HIP events around blocks of 64 consecutive plain launches on the step stream, right after the timed region (back to back every dispatch starts
    the instant its predecessor ends — profiles/r05_launch_overlap.txt — so this is the duration rocprofv3 reports; a `profiles/r05_*.txt` header
    reproduces it); `frac_step_rate` = the same bytes / `step_us`, the whole timed region per step (ONE hipEvent pair around it, never less than its
    wall time; resets and host stalls included).  Per-rank kernel times in `kernel_us_by_rank`; `traffic` from two live rocprofv3 PMC passes
    (N = 1), else a look-up of the committed ones; `device_copy_GBs` is the library's float4 copy kernel on the same box;
  * `configs`: the other BASELINE configurations (cfg2 4 096 envs / no cylinders, cfg4 envgen with the generator's cost
    per episode, cfg5_shard 6v2 / 16 cylinders — one GPU's shard), each with its own ms_per_step and roofline fraction, and
    `beyond_l3`: the headline shape at 262 144 and 1 048 576 envs (0.4 / 1.6 GB touched per step: past the 256 MiB Infinity Cache);
  * N > 1: each rank stops its clock when ITS K steps are complete on its device, ahead of the closing barrier (`closing_barrier_us`); `value` = all ranks' units / the MAX
    over ranks of those times;
  * `n_gpus`: distinct (host, device) pairs the ranks ran on (`config.ranks` = ranks); the RCCL backend refuses ranks > devices;
  * `tp_mode`: step + trajectory predictor (the reference's default `use_TP_net: 1`);
  * `cpu_baseline` (N = 1 only): the CPU oracle — test infrastructure, never the product — on a bounded sample.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F16_PEAK_TFLOPS = 2500.0
NUM_STATS = 24


def algorithmic_bytes_per_env(A, C, k, S=NUM_STATS, NT=1):
    """mandatory fp32 traffic of one env-step, one read + one write, no temporaries.
    Two-evader extension (NT=2): + the second evader's position read (12 B), position + velocity written (24 B)
    and 4 more values in every state_self row (16 B per pursuer)."""
    base = A * (232 + 4 * (20 + 3 * (A - 1) + 5 * k) + 4) + 4 * (3 + 3 * C + 1 + S) + 4 * (7 + S) + 1
    return base + (NT - 1) * (12 + 24 + 16 * A)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--settle-steps", type=int, default=1500,
                    help="pre-region phase of FIXED length, independent of --warmup (round 6): the env's first-use launches (masked reset, event-bracketed dispatch) and this "
                         "many plain steps, followed by a full reset, before the W warm-up steps.  A region that starts 0.4 ms into the process's life runs on clocks that "
                         "are still moving (the line's clock_mhz_*); 1 500 steps = 25 ms, where the committed rocprofv3 summaries start averaging too")
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--agents", type=int, default=3)
    ap.add_argument("--cylinders", type=int, default=8)
    ap.add_argument("--targets", type=int, default=1, help="evaders per env: 1 = the reference, 2 = BASELINE config 5's extension")
    ap.add_argument("--episode", type=int, default=800)
    ap.add_argument("--critic-state", action="store_true", help="also write the [E,A,20] centralised-critic state (critic_input: state)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--traffic-live", action=argparse.BooleanOptionalAction, default=True,
                    help="measure roofline.traffic in THIS run (default since round 4, N = 1 only): two extra rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, "
                         "separate, kernel-trace only) of a short inner run of the same workload, ~20 s each, outside every timed region; --no-traffic-live (or a "
                         "failing rocprofv3) falls back to the look-up of the committed passes (profiles/traffic.json)")
    ap.add_argument("--cpu-steps", type=int, default=100)
    ap.add_argument("--time-every", type=int, default=0, help="N > 0: every Nth launch of the timed region is an event-bracketed dispatch (off by default: each one "
                    "costs the region ~11 us of device idle time)")
    ap.add_argument("--abi-steps", type=int, default=500, help="secondary leg: the same steps through the bare C ABI (0 = skip)")
    ap.add_argument("--config-steps", type=int, default=2000,
                    help="steps of each extra BASELINE configuration leg (0 = skip the legs); regions of a few hundred steps read 3-4 %% slower than the "
                         "steady rate (the first ~100 steps after a leg's set-up run slower), as the predictor leg showed")
    ap.add_argument("--envgen-episodes", type=int, default=7, help="cfg4 leg: episodes (of --envgen-episode-length steps) incl. the generator")
    ap.add_argument("--envgen-episode-length", type=int, default=800, help="the reference's max_episode_length")
    ap.add_argument("--stream-groups", type=int, default=0,
                    help="optional extra leg (e.g. 2): the env batch as shards on separate HIP streams of one GPU, reported as "
                         "`stream_shards`; off by default so that a profile of the default command holds whole-batch launches only")
    ap.add_argument("--group-steps", type=int, default=1000)
    ap.add_argument("--state-digest", type=int, default=0,
                    help="check-out for the multi-GPU plumbing (tests/test_bench_contract.py): S > 0 = the job's env batch is stepped with ONE global action stream "
                         "(every rank takes its slice) and the line carries sha256 digests of the state after the timed region, one per S equal env slices of the "
                         "GLOBAL batch, so a W-rank run and a 1-rank run of the same global batch can be compared slice by slice")
    ap.add_argument("--tp-steps", type=int, default=2000,
                    help="extra leg: steps with the trajectory predictor in the observation (algo.use_TP_net: 1, the reference's "
                         "default config), reported as `tp_mode`; 0 = skip")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one per GPU, RCCL) and relay rank 0's line."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    if out.returncode != 0 or len(lines) != 1:
        sys.stderr.write(out.stdout[-2000:] + "\n" + out.stderr[-4000:] + "\n")
        raise SystemExit(out.returncode or 1)
    sys.stderr.write(out.stderr[-2000:])
    print(lines[0])


def live_traffic(args, kernel_substr):
    """HBM bytes per launch of the step kernel from two rocprofv3 PMC passes (MI355X_MICROARCH.md: counters in their own runs, kernel-trace
    only; FETCH_SIZE in KiB counts half the bytes on gfx950) of a short inner run of this workload.  None when rocprofv3 is not usable."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    inner = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "60", "--warmup", "10", "--envs", str(args.envs), "--agents", str(args.agents),
             "--cylinders", str(args.cylinders), "--targets", str(args.targets), "--no-cpu-baseline", "--tp-steps", "0", "--config-steps", "0", "--abi-steps", "0",
             "--stream-groups", "0", "--no-traffic-live"] + (["--critic-state"] if args.critic_state else [])
    out = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="hns_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE"):
                env.pop(k, None)
            r = subprocess.run([prof, "--kernel-trace", "--pmc", counter, "-d", d, "--", *inner], cwd="/tmp", env=env, capture_output=True, text=True, timeout=120)
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return None
            cur = sqlite3.connect(dbs[0]).cursor()
            rows = list(cur.execute(
                "select count(*), avg(e.value), count(distinct d.id) from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id=p.id "
                "join rocpd_kernel_dispatch d on d.event_id=e.event_id join rocpd_info_kernel_symbol s on d.kernel_id=s.id "
                "where s.kernel_name like ? and p.name = ?", (f"%{kernel_substr}%", counter)))
            n, avg, nd = rows[0]
            if not n or not nd:
                return None
            out[counter] = float(avg) * (n // nd)             # KiB per dispatch, summed over the counter's instances
            out[counter + "_dispatches"] = int(nd)
        except Exception:  # noqa: BLE001
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out["traffic_bytes_per_launch"] = int(round(out["FETCH_SIZE"] * 1024 * 2 + out["WRITE_SIZE"] * 1024))
    return out


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args)

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # HNS_BENCH_FORCE_DIST=1: take the distributed branch at world size 1 too (tests/test_bench_contract.py: the RCCL path on a 1-GPU box)
    force_dist = os.environ.get("HNS_BENCH_FORCE_DIST") == "1" and "WORLD_SIZE" in os.environ
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm; HNS_DIST_BACKEND=gloo only for smoke-testing the N>1 path on a 1-GPU box
        dist.init_process_group(os.environ.get("HNS_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    n_dev = max(torch.cuda.device_count(), 1)
    backend = dist.get_backend() if dist is not None else None
    if world > 1 and backend == "nccl" and int(os.environ.get("LOCAL_WORLD_SIZE", world)) > n_dev:
        # RCCL needs one device per rank; folding ranks onto one GPU would also overstate n_gpus
        raise SystemExit(f"[bench] {world} ranks but only {n_dev} GPU(s) visible: refuse to fold ranks onto one device with the nccl/RCCL "
                         f"backend (HNS_DIST_BACKEND=gloo runs the N>1 path on fewer devices as a smoke test; n_gpus then reports the devices)")
    local_rank = local_rank % n_dev
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    coll_dev = device if (dist is not None and dist.get_backend() == "nccl") else "cpu"

    import hns_amd  # noqa: F401
    from hns_amd import abi, config, sharding
    if not os.path.exists(abi.library_path()):            # fresh checkout: the .so files are git-ignored
        if rank == 0:
            import __graft_entry__
            __graft_entry__.build()
        if dist is not None:
            dist.barrier()
    from hns_amd.env import HideAndSeek
    from hns_amd.tensordict_shim import TensorDict
    import ctypes as Cx

    def sync():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(device)

    def make_env(E, A, C, NT=1, K=3, task=None, algo=None, cls=HideAndSeek, offset=0, episode=None):
        t = {"num_agents": A, "num_targets": NT, "cylinder": {"max_num": C, "min_num": C, "obs_max_cylinder": K},
             "env": {"num_envs": E, "max_episode_length": episode or args.episode}, "sim": {"device": f"cuda:{local_rank}"}}
        if task:
            cyl = dict(t["cylinder"], **task.pop("cylinder", {}))
            t.update(task)
            t["cylinder"] = cyl
        env = cls(config.make_cfg(t, algo=algo or {}), headless=True, env_index_offset=offset, write_critic_state=args.critic_state)
        env.set_seed(0)
        env.reset()
        return env

    def action_ring(E, A, seed, R=8):
        gen = torch.Generator(device=device).manual_seed(seed)
        acts = [torch.randn(E, A, 4, generator=gen, device=device) for _ in range(R)]
        return acts, [TensorDict({"agents": {"action": a}}, [E]) for a in acts]

    def roofline_obj(kernel_ms, E, A, C, NT=1, K=3, step_ms=None, samples=None):
        """Roofline object of the step kernel.  TWO durations, never mixed:
          * `kernel_us` = the kernel's average launch duration: HIP events around BLOCKS of consecutive plain launches on the step stream (no reset, no
            event-bracketed dispatch, no host sync inside a block), device time / launches.  profiles/r05_launch_overlap.txt shows why that is the
            kernel's duration as rocprofv3 reports it: in a back-to-back stream every dispatch starts the instant its predecessor ends (gap 0 for the
            median pair, no overlap), so period = duration.  `frac` = `frac_kernel` = algorithmic bytes per launch / that;
          * `step_us` = the whole timed region per step (one event pair around it, never less than its wall time: resets, host stalls included);
            `frac_step_rate` = bytes / that."""
        if kernel_ms is None or kernel_ms <= 0:
            return None
        b_env = algorithmic_bytes_per_env(A, C, K, NT=NT)
        achieved = b_env * E / (kernel_ms * 1e-3) / 1e9
        out = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
               "frac_kernel": round(achieved / HBM_PEAK_GBS, 4), "kernel_us": round(kernel_ms * 1e3, 2),
               "kernel_us_source": "HIP events around blocks of consecutive plain launches of the step kernel on the step stream (device time / launches), taken at steady state",
               "kernel_samples": samples, "bytes_per_launch": b_env * E, "bytes_per_env": b_env}
        if step_ms is not None and step_ms > 0:
            out["step_us"] = round(step_ms * 1e3, 2)
            out["frac_step_rate"] = round(b_env * E / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            out["step_us_source"] = "max(device time between ONE hipEvent pair around the timed region on the step stream, wall time of that region) / steps"
        return out

    def timed_steps(env, tds, n, warm, reset_every=None, timing=0):
        """n `env.step` calls (with the episode-boundary reset when given); returns (wall seconds, device ms between one event pair around them)."""
        done_td = TensorDict({}, [env.num_envs])
        for i in range(warm):
            env.step(tds[i % len(tds)])
        torch.cuda.synchronize(device)
        if timing:
            env.enable_kernel_timing(timing)
        env.region_begin()
        t0 = time.perf_counter()
        for i in range(n):
            env.step(tds[i % len(tds)])
            if reset_every and (warm + i + 1) % reset_every == 0:      # (the warm-up steps count towards the episode)
                done_td.set("_reset", env._bufs["done"])
                env.reset(done_td)
        env.region_end()
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        if timing:
            env.enable_kernel_timing(0)
        return dt, env.region_ms()

    def kernel_blocks(env, tds, blocks=8, per=64, lead=16, step=None):
        """Average launch duration of the step kernel (ms) from `blocks` blocks of `per` consecutive `env.step` launches, each between one event pair
        on the step stream; `lead` untimed launches before every block keep the queue full (the host issues a launch in a fraction of a kernel's
        duration at the bench's batch sizes), nothing is synchronised until all blocks are queued.  Returns (mean ms per launch, launches timed,
        [ms per launch of every block])."""
        evs = []
        step = step or env.step
        torch.cuda.synchronize(device)
        k = 0
        for _ in range(blocks):
            for _ in range(lead):
                step(tds[k % len(tds)])
                k += 1
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(per):
                step(tds[k % len(tds)])
                k += 1
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize(device)
        per_block = [a.elapsed_time(b) / per for a, b in evs]
        return sum(per_block) / len(per_block), blocks * per, per_block

    def leg_roofline(env, tds, n, dt, rms, E_, A_, C_, NT=1):
        """Roofline of a configuration leg: the region of `n` steps just timed gives the step rate, blocks of consecutive launches right after it (same
        env, same state stream) the kernel's own duration."""
        kms, kn, _ = kernel_blocks(env, tds, blocks=4, per=64)
        return roofline_obj(kms, E_, A_, C_, NT=NT, step_ms=max(rms, dt * 1e3) / n, samples=kn)

    def cfg2_leg(n):
        # cfg2: 4 096 envs, the default 5 cylinder slots all inactive (BASELINE configs[1]; bytes per env 1 497)
        e2 = make_env(4096, 3, 5, task={"cylinder": {"fixed_num": 0, "min_num": 0}})
        _, td2 = action_ring(4096, 3, 7)
        dt, rms = timed_steps(e2, td2, n, 100)
        r2 = leg_roofline(e2, td2, n, dt, rms, 4096, 3, 5)
        out = {"workload": "HideAndSeek 3v1, 5 cylinder slots all inactive, 4 096 envs", "value": round(4096 * 3 * n / dt, 1), "unit": "agent-steps/s",
               "ms_per_step": round(dt / n * 1e3, 5), "steps": n, "roofline": r2,
               "note": "64 workgroups on 256 CUs: one launch is a single workgroup's latency, not a bandwidth figure"}
        del e2
        return out

    # ======================= headline: cfg3 through env.step =====================================================
    E, A, C, K = args.envs, args.agents, args.cylinders, 3
    env = make_env(E, A, C, NT=args.targets, offset=rank * E)
    lib, henv = env._lib, env._env
    stream = torch.cuda.current_stream(device)
    sptr = Cx.c_void_p(stream.cuda_stream)
    if args.state_digest:
        # one action stream for the global batch: generated whole on every rank (same seed), sliced by rank
        gen = torch.Generator(device=device).manual_seed(1000)
        actions = [torch.randn(world * E, A, 4, generator=gen, device=device)[rank * E:(rank + 1) * E].contiguous() for _ in range(8)]
        tds = [TensorDict({"agents": {"action": a}}, [E]) for a in actions]
    else:
        actions, tds = action_ring(E, A, 1000 + rank)
    aptr = [Cx.c_void_p(a.data_ptr()) for a in actions]
    R = len(actions)
    reward, success = env._bufs["reward"], env.stats["success"]
    rollout = int(env.cfg.algo.get("train_every", 64))
    reset_td = TensorDict({}, [E])
    progress = {"t": 0}
    rate_hook = sharding.GlobalSuccessRate()
    env.success_rate_fn = rate_hook

    coll_events, coll_host_us = [], []

    def run(n):
        for _ in range(n):
            i = progress["t"]
            env.step(tds[i % R])
            progress["t"] = i + 1
            if (i + 1) % args.episode == 0:          # lock-step episodes: every env is done now
                reset_td.set("_reset", env._bufs["done"])
                env.reset(reset_td)
            if dist is not None and (i + 1) % rollout == 0:
                # per-rollout moments for advantage normalisation (learning/mappo.py:391-396 made data-parallel) + the success
                # rate of the curriculum (hideandseek.py:1012-1015) + ValueNorm1's batch moments (valuenorm.py:83-91): ONE all-gather of sharding.MOMENT_DIM = 8 fp64
                # values per rank over RCCL/xGMI (the reward buffer stands in for the learner's advantages and returns: same shape, same launch).
                # Its cost is timed where it is paid: device time between two events on the step stream for RCCL (the collective
                # runs on RCCL's stream, the step stream waits for it), host time of the call for gloo (host tensors)
                loc = sharding.local_moments(reward, success, reward)
                if coll_dev == "cpu":
                    c0 = time.perf_counter()
                    table = sharding._allgather(loc)          # (the collective itself, also with one rank)
                    coll_host_us.append((time.perf_counter() - c0) * 1e6)
                else:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record()
                    table = sharding._allgather(loc)
                    ev[1].record()
                    coll_events.append(ev)
                rate_hook.update(table)

    # the episode-boundary path (masked reset: mask conversion, statistics clone, one read-back of max(progress)) runs a handful of torch
    # kernels whose FIRST launch in a process loads their code objects — 15-20 ms on a fresh box, which landed inside the timed region
    # (first boundary at step 800 - warmup) and read as 26-29 us per step beside a 19-20 us kernel on two boxes of round 3.  One empty
    # masked reset (no env is done yet: the state is untouched) before the warm-up pays that once, outside the timed region.
    reset_td.set("_reset", env._bufs["done"])
    env.reset(reset_td)
    # likewise the event-bracketed dispatch (hipExtLaunchKernelGGL with start / stop events): launched once here, so that its first-use cost is not part of
    # the timed region either.  Round 6: these first-use launches and `--settle-steps` plain steps form ONE pre-region phase whose length does not depend on
    # --warmup; a full reset ends it, so the W warm-up steps and the K timed steps meet the episodes a fresh env would.
    env.enable_kernel_timing(1)
    env.step(tds[0])
    env.enable_kernel_timing(0)
    env.kernel_ms()
    for i in range(max(0, args.settle_steps)):
        env.step(tds[i % R])
    env.reset()
    progress["t"] = 0
    run(args.warmup)
    sync()
    clock_probe_before = env.clock_probe()        # one wave for 20 us on the step stream, finished before the region's start event is recorded
    torch.cuda.synchronize(device)                # (the host clock below must not start while the probe still runs)
    # The timed region: ONE hipEvent pair around it on the stream the steps are launched on (hns_region_begin / hns_region_end) beside the wall clock.
    # Nothing rides on the launches inside it by default (an event-bracketed dispatch — `--time-every N`, off at 0 — leaves the device idle for
    # ~11 us around itself, profiles/r05_launch_overlap.txt).  The kernel's own duration is measured right AFTER the region (`kernel_blocks`).
    time_every = max(0, args.time_every)
    in_region_events = time_every > 0 and args.steps >= 8 * time_every
    if in_region_events:
        env.enable_kernel_timing(time_every)
    done_ev = torch.cuda.Event()                  # (made here: creating it is not part of the K steps)
    env.region_begin()
    t0 = time.perf_counter()
    run(args.steps)
    env.region_end()
    # (the host polls the region's last event before it synchronises: a blocking wait wakes the host 20-30 us after the device is done — invisible in a
    #  2 000-step region, 1-1.5 us per step in the driver's 20-step one; the barrier + synchronize below still bracket the region)
    done_ev.record()
    while not done_ev.query():
        pass
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0            # this rank's K steps, complete on its device; the MAX over ranks below is the job's time
    # N > 1: the closing barrier brackets the region like the opening one, but its own latency (an RCCL all-reduce + the host's wake-up: tens of microseconds
    # against a 20-step region of 0.35 ms) is not part of the K steps — at N = 1 there is no barrier at all, so a clock stopped behind it would charge the
    # multi-GPU lines a fixed cost the single-GPU line does not carry.  Reported beside the line as `closing_barrier_us`.
    sync()
    closing_barrier_us = (time.perf_counter() - t0 - elapsed) * 1e6
    clock_probe_after = env.clock_probe()         # behind the region and its wall clock: not part of any timed quantity
    torch.cuda.synchronize(device)
    clock_mhz = {"before_region": env.clock_mhz(clock_probe_before), "after_region": env.clock_mhz(clock_probe_after)}
    env.enable_kernel_timing(0)
    region_ms = env.region_ms()
    in_ms, in_n = env.kernel_ms() if in_region_events else (-1.0, 0)
    state_digest = None
    if args.state_digest:
        import hashlib
        S = args.state_digest
        assert S % world == 0 and E % (S // world) == 0, "--state-digest: the slices must tile every rank's shard"
        per, n = S // world, E // (S // world)
        st = env.export_state()
        mine = []
        for i in range(per):
            h = hashlib.sha256()
            for k in sorted(st):
                if k == "nonfinite":
                    continue
                v = st[k]
                h.update(v[:, i * n:(i + 1) * n].tobytes() if k == "stats" else v[i * n:(i + 1) * n].tobytes())
            mine.append(h.hexdigest())
        if dist is not None:
            allv = [None] * world
            dist.all_gather_object(allv, mine)
            state_digest = [d for r in allv for d in r]
        else:
            state_digest = mine
    # the kernel alone: blocks of consecutive launches right after the region (same env, same state stream, outside every timed quantity) ...
    #     — at steady state: a short timed region (the driver's --steps 20 --warmup 5) ends a few hundred microseconds into the device's life in this
    #     process, where blocks read 16.7-19.4 us on a box whose steady launches take 16.0 (clocks still settling); so the blocks start no earlier than
    #     1 500 launches in, which is also what the committed rocprofv3 summaries (thousands of launches) average over
    for i in range(max(0, 1500 - args.settle_steps - args.warmup - args.steps)):
        env.step(tds[i % R])
    blk_ms, blk_n, blk_each = kernel_blocks(env, tds, blocks=8, per=64)
    clock_probe_blocks = env.clock_probe()
    torch.cuda.synchronize(device)
    clock_mhz["after_kernel_blocks"] = env.clock_mhz(clock_probe_blocks)
    # ... and, as a separately named field, 16 ISOLATED dispatches with start / stop events bound to each (hns_enable_timing): such a dispatch sits between
    # two idle gaps and its events take in more than the kernel (rocprofv3 reads 15.8 us for the dispatches these events read 18-20 us for)
    env.enable_kernel_timing(1)
    for i in range(16):
        env.step(tds[i % R])
    torch.cuda.synchronize(device)
    env.enable_kernel_timing(0)
    post_ms, post_n = env.kernel_ms()
    step_ms = max(region_ms, elapsed * 1e3) / args.steps if region_ms > 0 else elapsed * 1e3 / args.steps
    roofline = roofline_obj(blk_ms, E, A, C, NT=args.targets, K=K, step_ms=step_ms, samples=blk_n)
    if roofline is not None:
        roofline["kernel_us_blocks"] = [round(x * 1e3, 2) for x in blk_each]
        roofline["region_ms"] = round(region_ms, 4)
        roofline["region_wall_ms"] = round(elapsed * 1e3, 4)
        # what a region costs beyond its launches: first-launch latency + the host noticing the end.  ~40 us — 2 us per step of the driver's 20-step
        # region, 0.02 us per step of the default 2 000-step one; with clock_mhz_* at the maximum it is what separates ms_per_step from kernel_us there
        roofline["region_fixed_cost_us"] = round(elapsed * 1e6 - args.steps * blk_ms * 1e3, 1)
        roofline["kernel_us_isolated_dispatch_events"] = round(post_ms * 1e3, 2) if post_n > 0 else None
        roofline["isolated_dispatch_samples"] = post_n
        if in_n > 0:
            roofline["kernel_us_dispatch_events_in_region"] = round(in_ms * 1e3, 2)
            roofline["dispatch_event_samples_in_region"] = in_n
    kernel_ms = blk_ms

    coll_us = coll_host_us[-(args.steps // rollout):] if coll_host_us else [a.elapsed_time(b) * 1e3 for a, b in coll_events[-(args.steps // rollout):]]
    collective = None
    if coll_us:
        collective = {"per_rollout_us_mean": round(sum(coll_us) / len(coll_us), 1), "per_rollout_us_max": round(max(coll_us), 1), "rollouts": len(coll_us),
                      "rollout_steps": rollout, "what": "local moments (ONE launch: hns_rollout_moments) + ONE all-gather of 8 fp64 per rank; " +
                      ("host time of the call (gloo, host tensors)" if coll_host_us else "device time between two events on the step stream (RCCL)")}
    # ranks that actually took part (an all-reduce of ones), slowest rank's wall time, per-rank kernel time
    n_ranks, n_devices, kernel_by_rank = 1, 1, None
    if dist is not None:
        ones = torch.ones(1, device=coll_dev, dtype=torch.float64)
        dist.all_reduce(ones)
        n_ranks = int(round(float(ones.item())))
        where = [None] * world                      # n_gpus = distinct (host, device) pairs, not ranks
        dist.all_gather_object(where, (socket.gethostname(), local_rank))
        n_devices = len(set(where))
        tt = torch.tensor([elapsed], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        kk = [torch.zeros(1, device=coll_dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(kk, torch.tensor([kernel_ms * 1e3], device=coll_dev, dtype=torch.float64))
        kernel_by_rank = [round(float(x.item()), 2) for x in kk]
    assert torch.isfinite(env._bufs["reward"]).all(), "non-finite reward"
    assert env.check_finite(), "non-finite state"
    value = n_ranks * E * A * args.steps / elapsed

    if roofline is not None:
        traffic = None
        try:   # HBM bytes per launch measured in the committed PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            key = f"hns_step_kernel<{A}>|E{E}|C{C}|k{K}|critic_state_{'on' if args.critic_state else 'off'}"
            traffic = tj.get(key, {}).get("traffic_bytes_per_launch")
            roofline["traffic_source"] = tj.get(key, {}).get("source", "profiles/traffic.json") + " (static look-up of the committed rocprofv3 PMC passes, not measured in this run)"
        except Exception:  # noqa: BLE001
            pass
        if args.traffic_live and world == 1:
            live = live_traffic(args, "hns_step_v4_kernel")
            if live is not None:
                traffic = live["traffic_bytes_per_launch"]
                roofline["traffic_source"] = "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, two separate passes of this run's workload (FETCH_SIZE KiB x 2 on gfx950 + WRITE_SIZE KiB)"
                roofline["traffic_detail"] = live
        roofline["traffic"] = traffic
        roofline["kernel"] = "hns_step_v4_kernel<%d,%d,%s,4,false,%d>" % (A, args.targets, "false" if E % 64 == 0 else "true", C if (E % 64 == 0 and C in (5, 8, 16)) else 0)
        if kernel_by_rank:
            roofline["kernel_us_by_rank"] = {"min": min(kernel_by_rank), "max": max(kernel_by_rank), "all": kernel_by_rank}
        # achievable bandwidth on this box ("measure achievable with a device copy kernel and report both"): the library's float4
        # copy kernel (hns_copy_f4; MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy) at two footprints — 2 x 48 MiB, about what one
        # step of the headline batch touches (the 256 MiB Infinity Cache holds it between launches, as it holds the step's buffers), and
        # 2 x 1 GiB (every byte to / from HBM)
        try:
            like, hbm = round(env.device_copy_GBs(48, 40), 1), round(env.device_copy_GBs(1024, 10), 1)
            roofline["device_copy_GBs"] = hbm
            roofline["device_copy_GBs_at_step_footprint"] = like
            roofline["device_copy_kernel"] = "hns_copy_f4_kernel (one float4 per thread): 2 x 1 GiB per pass (HBM only) / 2 x 48 MiB per pass (Infinity-Cache resident, like the step's 105 MB)"
            roofline["frac_of_device_copy"] = round(roofline["achieved"] / (like if E * algorithmic_bytes_per_env(A, C, K, NT=args.targets) < 200e6 else hbm), 4)
            roofline["frac_of_device_copy_what"] = "achieved (kernel alone) / the copy kernel's rate at the same footprint"
        except Exception as ex:  # noqa: BLE001
            roofline["device_copy_error"] = str(ex)[:200]

    single = world == 1
    # secondary: the same steps through the bare C ABI (what the Python class adds is the difference)
    abi_rate = None
    if args.abi_steps > 0 and single:
        for i in range(50):
            lib.hns_step(henv, aptr[i % R], sptr)
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        for i in range(args.abi_steps):
            assert lib.hns_step(henv, aptr[i % R], sptr) == 0, lib.hns_last_error()
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t1
        abi_rate = {"value": round(E * A * args.abi_steps / dt, 1), "unit": "agent-steps/s", "steps": args.abi_steps,
                    "ms_per_step": round(dt / args.abi_steps * 1e3, 5), "what": "hns_step through ctypes, no Python class, no resets"}

    # ======================= the other BASELINE configurations ================================================
    configs = {}
    if args.config_steps > 0 and single and args.targets == 1:
        n = args.config_steps
        configs["cfg2"] = cfg2_leg(n)
        # cfg5's per-GPU shard: 6 pursuers / 2 evaders / 16 cylinders / 65 536 envs (the two-evader extension)
        e5 = make_env(E, 6, 16, NT=2)
        _, td5 = action_ring(E, 6, 9)
        dt, rms = timed_steps(e5, td5, n, 100)
        r5 = leg_roofline(e5, td5, n, dt, rms, E, 6, 16, NT=2)
        assert e5.check_finite()
        if r5 is not None:
            try:
                tj5 = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(f"hns_step_kernel<6>|E{E}|C16|k3|critic_state_off|NT2", {})
                r5["traffic"] = tj5.get("traffic_bytes_per_launch")
                r5["traffic_source"] = tj5.get("source", "profiles/traffic.json") + " (static look-up of the committed rocprofv3 PMC passes)"
            except Exception:  # noqa: BLE001
                pass
        configs["cfg5_shard"] = {"workload": f"HideAndSeek 6v2 (extension), 16 cylinders, {E} envs = one GPU's shard of the 8 x 65 536 job",
                                 "value": round(E * 6 * n / dt, 1), "unit": "agent-steps/s", "ms_per_step": round(dt / n * 1e3, 5), "steps": n, "roofline": r5}
        del e5
        # the two-evader extension at the reference's own pursuer count (3v2, 8 cylinders): the shape `num_targets: 2` meets most often outside BASELINE's
        # configuration 5; its instantiations spilled registers until round 6 (VERDICT r5 #5 asked for a number)
        e32 = make_env(E, 3, 8, NT=2)
        _, td32 = action_ring(E, 3, 11)
        dt, rms = timed_steps(e32, td32, n, 100)
        r32 = leg_roofline(e32, td32, n, dt, rms, E, 3, 8, NT=2)
        assert e32.check_finite()
        configs["ext_3v2"] = {"workload": f"HideAndSeek 3v2 (two-evader extension), 8 cylinders, {E} envs",
                              "value": round(E * 3 * n / dt, 1), "unit": "agent-steps/s", "ms_per_step": round(dt / n * 1e3, 5), "steps": n, "roofline": r32}
        del e32
        # beyond the 256 MiB Infinity Cache: the headline batch touches ~105 MB per step, which the last-level cache can hold between
        # launches; these two batches touch 0.42 GB and 1.7 GB per step, so every byte comes from / goes to HBM
        beyond = {}
        for eb in ((262144, 1048576) if E >= 65536 else ()):          # (not at the reduced sizes of the contract test)
            nb = max(40, n // 4)
            ebv = make_env(eb, A, C)
            _, tdb = action_ring(eb, A, 13, R=2)
            dtb, rmsb = timed_steps(ebv, tdb, nb, 20)
            rb = leg_roofline(ebv, tdb, nb, dtb, rmsb, eb, A, C)
            assert ebv.check_finite()
            beyond[str(eb)] = {"workload": f"HideAndSeek {A}v1, {C} cylinders, {eb} envs ({algorithmic_bytes_per_env(A, C, K) * eb / 1e6:.0f} MB algorithmic per step)",
                               "value": round(eb * A * nb / dtb, 1), "unit": "agent-steps/s", "ms_per_step": round(dtb / nb * 1e3, 5), "steps": nb, "roofline": rb}
            del ebv, tdb
            torch.cuda.empty_cache()
        if beyond:
            configs["beyond_l3"] = beyond
        # cfg4: HideAndSeek_envgen — steps + the Adaptive Environment Generator at every episode boundary, in the reference's order
        # (trim, then draw the next batch from the trimmed history).  Running the trim on a side stream beside the next episode's
        # steps was measured in round 3 and dropped: the steps slow down by what the trim saves (DESIGN.md §3.3,
        # profiles/r03_bench_async_trim_experiment.json)
        from hns_amd.envgen import HideAndSeek_envgen
        L, EP = args.envgen_episode_length, args.envgen_episodes

        def envgen_leg(R_min=0.0, R_max=1.0):
            t0c = time.perf_counter()
            e4 = make_env(E, 3, 8, cls=HideAndSeek_envgen, episode=L,
                          task={"name": "HideAndSeek_envgen", "use_particle_generator": 1, "ratio_unif": 0.3, "eval_iter": 3, "R_min": R_min, "R_max": R_max})
            torch.cuda.synchronize(device)
            first_reset_ms = (time.perf_counter() - t0c) * 1e3
            _, td4 = action_ring(E, 3, 11)
            gen_ms, ep_ms, step_s = [], [], 0.0
            rtd = TensorDict({}, [E])
            t_all = time.perf_counter()
            for ep in range(EP):
                g0 = e4.generator_seconds
                ts = time.perf_counter()
                for t in range(L):
                    e4.step(td4[t % len(td4)])
                torch.cuda.synchronize(device)
                step_s += time.perf_counter() - ts - (e4.generator_seconds - g0)
                rtd.set("_reset", e4._bufs["done"])
                e4.reset(rtd)
                torch.cuda.synchronize(device)
                gen_ms.append((e4.generator_seconds - g0) * 1e3)
                ep_ms.append((time.perf_counter() - ts) * 1e3)
            total = time.perf_counter() - t_all
            # (the step kernel alone, through the base class's `_step`: these launches run past the episode's end, where the generator's hook has nothing to update)
            k4_ms, k4_n, _ = kernel_blocks(e4, td4, blocks=4, per=64, step=lambda td: HideAndSeek._step(e4, td))
            r4 = roofline_obj(k4_ms, E, 3, 8, step_ms=step_s / (L * EP) * 1e3, samples=k4_n)
            steady = sorted(gen_ms[1:])
            # steady state = the last three episodes (one task batch: history full, everything warm); an episode of the reference's
            # 800 steps costs 800 x the measured step time + that episode's measured non-step time
            nonstep_last3 = sum(ep_ms[-3:]) * 1e-3 - 3 * L * (step_s / (L * EP))
            what = ("R_min 0.0 / R_max 1.0 (the reference's 0.5 / 0.9 admits no task under this bench's random policy: every task enters the history here, so the "
                    "trim runs at its full 5000 + E size - the generator's worst case)" if (R_min, R_max) == (0.0, 1.0) else
                    f"R_min {R_min} / R_max {R_max} (BASELINE's values; under this bench's random policy almost no task's success rate falls inside, "
                    "so the history stays small and the trim is short - the generator's cost is the task reset)")
            out = {"workload": f"HideAndSeek_envgen, 3v1, 8 cylinders, {E} envs, episodes of {L} steps (the reference: 800), new task batch every 3 episodes, " + what,
                   "value": round(E * 3 * L * EP / step_s, 1), "unit": "agent-steps/s (stepping only)", "ms_per_step": round(step_s / (L * EP) * 1e3, 5),
                   "roofline": r4, "episodes": EP, "generator_ms_per_episode": [round(x, 2) for x in gen_ms],
                   "episode_wall_ms": [round(x, 2) for x in ep_ms],
                   "generator_ms_per_episode_steady_median": round(steady[len(steady) // 2], 2) if steady else None,
                   "generator_ms_task_batch_max": round(max(gen_ms), 2), "construction_and_first_reset_ms": round(first_reset_ms, 1),
                   "value_incl_generator": round(E * 3 * L * EP / total, 1),
                   "value_incl_generator_at_800_step_episodes": round(E * 3 * 800 / (800 * step_s / (L * EP) + sum(gen_ms[1:]) / max(EP - 1, 1) * 1e-3), 1),
                   "generator_ms_task_batch_steady": round(max(gen_ms[-3:]), 2),            # the last batch: history full, everything warm
                   "value_incl_generator_steady_at_800_step_episodes": round(E * 3 * 800 * 3 / (3 * 800 * step_s / (L * EP) + max(nonstep_last3, 0.0)), 1),
                   "history_size": len(e4.gen_buffer)}
            del e4
            return out
        if args.envgen_episodes > 0:
            configs["cfg4"] = envgen_leg()
            configs["cfg4_baseline_R"] = envgen_leg(0.5, 0.9)       # BASELINE's R_min / R_max beside the worst case

    # secondary leg ("use_TP_net=1 reported separately"): step + hns_tp_observe
    tp_mode = None
    if args.tp_steps > 0 and single and args.targets == 1:
        env_tp = make_env(E, A, C, algo={"use_TP_net": 1})
        n = args.tp_steps
        # (a region of 200-300 steps measured 110-114 us per step where 1 000 steps give 102 and 3 000 give 100: the first ~100 steps after the
        #  set-up run slower — device clocks, first launches — and a short region is mostly those)
        dt_tp, _ = timed_steps(env_tp, tds, n, 100, reset_every=args.episode)      # (lock-step episodes, reset at the boundary, as the headline region)
        timed_steps(env_tp, tds, 64, 4, timing=4)                 # the step kernel's own duration in this mode (events on its dispatch)
        tp_step_us = env_tp.kernel_ms()[0] * 1e3
        T, F, I = env_tp.tp_history_step, env_tp.tp_future_step, env_tp.tp_frame_dim
        flop_env = 2.0 * (T * 4 * 64 * (I + 64) + 64 * 3 * F)          # useful FLOP of LSTM(I->64) x T + Linear(64->3F), per env and step
        # the predictor alone (frame append + LSTM + rows: hns_tp_observe), bracketed by events on the stream it is launched on
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(20, n // 3)
        for _ in range(5):
            env_tp._tp_observe()
        ev0.record()
        for _ in range(reps):
            env_tp._tp_observe()
        ev1.record()
        torch.cuda.synchronize(device)
        obs_us = ev0.elapsed_time(ev1) / reps * 1e3
        tflops = flop_env * E / (obs_us * 1e-6) / 1e12
        tp_mode = {"value": round(E * A * n / dt_tp, 1), "unit": "agent-steps/s", "steps": n, "ms_per_step": round(dt_tp / n * 1e3, 5),
                   "useful_mflop_per_env": round(flop_env / 1e6, 4), "observe_us": round(obs_us, 2), "step_kernel_us": round(tp_step_us, 2),
                   "roofline": {"bound": "mfma", "achieved": round(tflops, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tflops / 2500.0, 4),
                                "note": "useful FLOP of LSTM + FC per launch of hns_tp_observe / its duration / dense f16 peak; every product is issued as three "
                                        "f16 MFMAs (hi*hi, hi*lo, lo*hi of the split operands) to hold the 1e-5 parity, so the matrix pipe does 3x this"},
                   "halves_on_two_streams": env_tp._halves is not None,
                   "what": "env.step with algo.use_TP_net=1: hns_step + hns_tp_observe (window shift, LSTM(16->64)x10 + FC on the matrix cores, 35-value rows)"
                           + ("; the batch as two half batches on two streams (task.tp_overlap): step_kernel_us is ONE half's step kernel, observe_us the "
                              "predictor over the whole batch on one stream" if env_tp._halves is not None else "")}
        del env_tp
        # ... and at the reference's OWN default batch (cfg/task/HideAndSeek.yaml: 2 048 envs, 3 pursuers, 5 cylinder slots; cfg/algo/mappo.yaml: use_TP_net 1):
        # 16 four-tile workgroups would leave 240 CUs idle, so the predictor serves such batches with one column tile per workgroup (csrc/hns_tp.hip: ws_envs)
        e_rd = make_env(2048, 3, 5, task={"cylinder": {"min_num": 4}}, algo={"use_TP_net": 1})
        _, td_rd = action_ring(2048, 3, 13)
        n_rd = min(n, 1000)
        dt_rd, _ = timed_steps(e_rd, td_rd, n_rd, 100, reset_every=args.episode)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            e_rd._tp_observe()
        ev0.record()
        for _ in range(200):
            e_rd._tp_observe()
        ev1.record()
        torch.cuda.synchronize(device)
        tp_mode["reference_default_batch"] = {
            "workload": "HideAndSeek 3v1, 5 cylinder slots (4-5 active), 2 048 envs, use_TP_net 1: the reference's own task / algo defaults",
            "value": round(2048 * 3 * n_rd / dt_rd, 1), "unit": "agent-steps/s", "steps": n_rd, "ms_per_step": round(dt_rd / n_rd * 1e3, 5),
            "observe_us": round(ev0.elapsed_time(ev1) / 200 * 1e3, 2)}
        del e_rd

    # secondary leg: the same envs as G shards on G HIP streams of this GPU (the multi-GPU sharding applied inside one GPU)
    streams_mode = None
    if args.stream_groups > 1 and single and args.targets == 1 and E % args.stream_groups == 0:
        G, Eg = args.stream_groups, E // args.stream_groups
        shards = [make_env(Eg, A, C, offset=g * Eg) for g in range(G)]
        gstreams = [torch.cuda.Stream(device) for _ in range(G)]
        gptr = [Cx.c_void_p(st.cuda_stream) for st in gstreams]
        gact = [[Cx.c_void_p(a[g * Eg:(g + 1) * Eg].data_ptr()) for a in actions] for g in range(G)]
        torch.cuda.synchronize(device)

        def run_groups(n):
            for i in range(n):
                for g in range(G):
                    assert lib.hns_step(shards[g]._env, gact[g][i % R], gptr[g]) == 0, lib.hns_last_error()
        run_groups(100)
        torch.cuda.synchronize(device)
        t2 = time.perf_counter()
        run_groups(args.group_steps)
        torch.cuda.synchronize(device)
        ms_g = (time.perf_counter() - t2) / args.group_steps * 1e3
        b_env = algorithmic_bytes_per_env(A, C, K)
        streams_mode = {"groups": G, "value": round(E * A / (ms_g * 1e-3), 1), "unit": "agent-steps/s", "steps": args.group_steps,
                        "ms_per_step": round(ms_g, 5), "hbm_frac": round(b_env * E / (ms_g * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "what": f"the same {E} envs as {G} shards of {Eg} on {G} HIP streams (no join between steps)"}
        del shards

    cpu_baseline = None
    if rank == 0 and single and not args.no_cpu_baseline:
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
        O.set_threads(best_n)
        m0 = time.perf_counter()
        for _ in range(args.cpu_steps):
            O.step(env.hcfg, host, act)
        mdt = time.perf_counter() - m0
        O.set_threads(1)
        # SURVEY §8(d) also names E = 4 096 (BASELINE configs[1]: 5 cylinder slots, all inactive): the same oracle, one thread and the best thread count
        c2 = config.resolve_hns_cfg(config.make_cfg({"num_agents": 3, "cylinder": {"max_num": 5, "min_num": 0, "fixed_num": 0}, "env": {"num_envs": 4096, "max_episode_length": args.episode}}))
        h2 = O.alloc_buffers(c2)
        O.reset(c2, h2, None, 0, 0)
        a2 = np.random.default_rng(1).standard_normal((4096, 3, 4)).astype(np.float32)
        cfg2_cpu = {}
        for threads in (1, best_n):
            O.set_threads(threads)
            O.step(c2, h2, a2)
            n2 = 200 if threads == 1 else 1000
            q0 = time.perf_counter()
            for _ in range(n2):
                O.step(c2, h2, a2)
            cfg2_cpu[threads] = 4096 * 3 * n2 / (time.perf_counter() - q0)
        O.set_threads(1)
        cpu_baseline = {"value": round(E * A * args.cpu_steps / mdt, 1), "unit": "agent-steps/s", "cores": best_n,
                        "cfg2": {"value": round(max(cfg2_cpu.values()), 1), "unit": "agent-steps/s", "cores": max(cfg2_cpu, key=cfg2_cpu.get),
                                 "one_core_value": round(cfg2_cpu[1], 1), "sample": "200 / 1000 steps of HideAndSeek 3v1, 5 inactive cylinder slots, 4 096 envs with the C oracle"},
                        "kind": "port", "one_core_value": round(one_core, 1),
                        "sample": f"{args.cpu_steps} steps of the same {E}-env workload with the C oracle "
                                  f"(oracle/hns_oracle.c): {best_n} threads (best of a probe up to {avail}) {mdt:.1f} s, 1 thread {cdt:.1f} s"}

    if rank == 0:
        import hashlib
        lib_sha16 = hashlib.sha256(open(abi.library_path(), "rb").read()).hexdigest()[:16]     # the same digest heads the profiles/*.txt of this build
        out = {
            "metric": "env agent-steps/sec at 65536 envs, HideAndSeek 3v1; 1/2/4/8 GPU",
            "value": round(value, 1), "unit": "agent-steps/s", "n_gpus": n_devices, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"HideAndSeek {A}v{args.targets}, {C} random cylinders + LOS/k-nearest sensing, "
                                   f"{E} envs per GPU (BASELINE configs[2]), timed through env.step(td)",
                       "num_envs_per_gpu": E, "num_agents": A, "num_targets": args.targets, "num_cylinders": C, "obs_max_cylinder": K,
                       "episode_length": args.episode, "critic_state_output": args.critic_state,
                       "sharding": f"contiguous env slices x{n_ranks}", "world_size_launched": world, "ranks": n_ranks,
                       "dist_backend": backend, "library_sha16": lib_sha16,
                       "collective": "1 all-gather of 8 fp64 per 64-step rollout" if dist is not None else "none",
                       "pre_region": {"settle_steps": args.settle_steps, "what": "first-use launches + settle_steps plain steps + one full reset, then the W warm-up steps "
                                                                                 "(fixed length, independent of --warmup)"}},
            "clock_mhz_before_region": round(clock_mhz["before_region"], 1) if clock_mhz.get("before_region") else None,
            "clock_mhz_after_region": round(clock_mhz["after_region"], 1) if clock_mhz.get("after_region") else None,
            "clock_mhz_after_kernel_blocks": round(clock_mhz["after_kernel_blocks"], 1) if clock_mhz.get("after_kernel_blocks") else None,
            "clock_mhz_what": "shader clock read by a one-wave probe kernel (s_memtime cycles per s_memrealtime 100 MHz tick, 20 us) on the step stream right before the "
                              "region's start event / right behind its stop event / behind the kernel-duration blocks: the chip clocks to its power budget, a nearly idle "
                              "chip reads near the 2.4 GHz maximum",
            "schema": 6, "schema_note": "r05 on: roofline.frac / achieved / kernel_us describe the step kernel ALONE (blocks of plain launches after the region); the region-bounded "
                                      "figure that BENCH_r01-r04 called frac is frac_step_rate.  r06 on: a fixed pre-region phase (config.pre_region) and clock_mhz_* fields",
            "collective_us": collective, "closing_barrier_us": round(closing_barrier_us, 1) if dist is not None else None, "state_digest": state_digest,
            "env_frames_per_s": round(value / A, 1),
            "roofline": roofline, "cpu_baseline": cpu_baseline, "abi_rate": abi_rate, "configs": configs or None,
            "tp_mode": tp_mode, "stream_shards": streams_mode,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
