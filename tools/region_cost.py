import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tools/ -> repo
for p in (ROOT, os.path.join(ROOT, "oracle")): sys.path.insert(0, p)
import torch, hns_amd
from hns_amd import config
from hns_amd.env import HideAndSeek
E=65536
env = HideAndSeek(config.make_cfg({"num_agents":3,"cylinder":{"max_num":8,"min_num":8},"env":{"num_envs":E,"max_episode_length":800}}), headless=True)
env.set_seed(0); env.reset()
tds=[env.rand_step_input(torch.randn(E,3,4,device=env.device)) for _ in range(8)]
for i in range(2000): env.step(tds[i%8])
torch.cuda.synchronize()
# idle synchronize cost
t=[]
for _ in range(50):
    t0=time.perf_counter(); torch.cuda.synchronize(); t.append((time.perf_counter()-t0)*1e6)
t.sort(); print("idle torch.cuda.synchronize: median %.1f us min %.1f"%(t[25],t[0]))
# env.step host cost when the queue is deep (GPU-bound => measure enqueue only for first few)
t=[]
for _ in range(30):
    torch.cuda.synchronize()
    t0=time.perf_counter(); env.step(tds[0]); t.append((time.perf_counter()-t0)*1e6)
t.sort(); print("env.step host call on an idle device: median %.1f us min %.1f"%(t[15],t[0]))
# one step: launch -> completion seen by polling an event
t=[];t2=[]
for _ in range(30):
    torch.cuda.synchronize()
    t0=time.perf_counter(); env.step(tds[0]); ev=torch.cuda.Event(); ev.record()
    while not ev.query(): pass
    t1=time.perf_counter(); torch.cuda.synchronize(); t3=time.perf_counter()
    t.append((t1-t0)*1e6); t2.append((t3-t1)*1e6)
t.sort(); t2.sort(); print("one step, t0 -> event seen by polling: median %.1f us min %.1f (kernel 16); synchronize behind it: median %.1f min %.1f"%(t[15],t[0],t2[15],t2[0]))
for K in (1,5,20):
    t=[]
    for _ in range(20):
        torch.cuda.synchronize()
        t0=time.perf_counter()
        for i in range(K): env.step(tds[i%8])
        ev=torch.cuda.Event(); ev.record()
        while not ev.query(): pass
        t1=time.perf_counter(); t.append((t1-t0)*1e6)
    t.sort(); print("K=%d steps to polled completion: median %.1f us min %.1f -> fixed %.1f"%(K,t[10],t[0],t[10]-K*15.9))

def trial(K, pre=None, region=False, torch_event=True):
    t = []
    for _ in range(12):
        if pre: pre()
        torch.cuda.synchronize()
        if region: env.region_begin()
        t0 = time.perf_counter()
        for i in range(K): env.step(tds[i % 8])
        if region: env.region_end()
        ev = torch.cuda.Event(); ev.record()
        while not ev.query(): pass
        torch.cuda.synchronize()
        t.append(((time.perf_counter() - t0) * 1e6, env.region_ms() * 1e3 if region else 0.0))
    t.sort()
    return t[6][0], t[0][0], t[6][1]

def pre_reset_warm():
    env.reset()
    for i in range(5): env.step(tds[i % 8])
def pre_settle_reset_warm():
    for i in range(1500): env.step(tds[i % 8])
    env.reset()
    for i in range(5): env.step(tds[i % 8])
def pre_probe():
    p = env.clock_probe(); torch.cuda.synchronize()
def pre_all():
    pre_settle_reset_warm(); torch.cuda.synchronize(); pre_probe()
for name, kw in (("plain", {}), ("plain again", {}), ("region events", {"region": True}), ("region events", {"region": True}), ("after reset + 5 steps", {"pre": pre_reset_warm}), ("after 1500 + reset + 5", {"pre": pre_settle_reset_warm}),
                 ("after clock probe", {"pre": pre_probe}), ("bench sequence", {"pre": pre_all, "region": True})):
    m, mn, dev = trial(20, **kw)
    print("K=20 %-28s median %.1f us min %.1f -> per step %.2f   (device time between the region's events %.1f us)" % (name, m, mn, m / 20, dev))
