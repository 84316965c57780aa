import sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import torch, hns_amd
from hns_amd import abi
lib = abi.load_library()
dev = torch.device("cuda:0")
for n, k, d in ((51000, 5000, 36), (65536, 5000, 36), (70536, 5000, 36), (10000, 5000, 36), (69632, 5000, 27), (69632, 5000, 30)):
    p = torch.rand(n, d, device=dev)
    out = torch.zeros(k, dtype=torch.int32, device=dev)
    scratch = torch.zeros(lib.hns_fps_scratch_bytes(), dtype=torch.uint8, device=dev)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    times = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        assert lib.hns_fps(p.data_ptr(), n, d, k, 0, out.data_ptr(), scratch.data_ptr(), s) == 0
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        times.append(dt * 1e3)
    print(f"hns_fps n={n} d={d} k={k}: {dt*1e3:.1f} ms ({dt/k*1e6:.2f} us/round), calls {[round(t, 1) for t in times]} ms, err word {int(scratch[:8].view(torch.int64)[0])}")
