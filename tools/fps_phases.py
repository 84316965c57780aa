import sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import torch, hns_amd
from hns_amd import abi
lib = abi.load_library()
dev = torch.device("cuda:0")
names = ["update", "cand+top stage1+barrier", "top stage2 (wave 0)", "publish+sweep", "global top-8 (+row loads issued)", "rows->LDS, pairs, accept", "final barrier", "loop top (out_idx)"]
"""Where an exchange of the XCD-local farthest-point kernel spends its time: per-phase stamps of wave 0 of workgroup 0 (a build with -DFPS_PHASES:
tools/build_variant.sh fpsph -DFPS_PHASES; run with HNS_LIBRARY=build/variants/libhns_fpsph.so).  The device-side sums are cumulative: deltas are printed."""
prev = [0] * 9
for n, k, d in ((70536, 5000, 36), (65536, 5000, 36), (10000, 5000, 36)):
    p = torch.rand(n, d, device=dev)
    out = torch.zeros(k, dtype=torch.int32, device=dev)
    scratch = torch.zeros(lib.hns_fps_scratch_bytes(), dtype=torch.uint8, device=dev)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        assert lib.hns_fps(p.data_ptr(), n, d, k, 0, out.data_ptr(), scratch.data_ptr(), s) == 0
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    w = scratch[:64].view(torch.int64).cpu().tolist()
    ph = []
    for i in range(4):
        ph += [w[2 + i] & 0xFFFFFFFF, (w[2 + i] >> 32) & 0xFFFFFFFF]
    cur = ph + [w[6]]
    # two launches since the last read: halve
    dl = [((c - q) & 0xFFFFFFFF) / 2.0 for c, q in zip(cur, prev)]
    prev = cur
    nx = dl[8]
    print(f"n={n}: {dt*1e3:.2f} ms per launch, {nx:.0f} exchanges ({k / max(nx, 1):.2f} samples per exchange), {dt * 1e6 / max(nx, 1):.2f} us per exchange")
    if dl[7] / 100.0 / max(nx, 1) > 100.0:               # the very first stamp of a process measures from an unset clock value: not a duration
        dl[7] = float("nan")
    for nm, t in zip(names, dl[:8]):
        print(f"   {nm:36s} {t / 100.0 / max(nx, 1):7.3f} us per exchange")
    print(f"   sum {sum(x for x in dl[:8] if x == x) / 100.0 / max(nx, 1):.3f} us per exchange")
