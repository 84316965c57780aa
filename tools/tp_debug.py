#!/usr/bin/env python3
"""Where does a predictor build deviate from the oracle?  Zeroes parts of the parameters (recurrent weights, input weights, biases)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, torch, hns_amd
from hns_amd import abi, config
from hns_amd.env import HideAndSeek
import hns_oracle as O
E = 256
for name, zero in (("full", ()), ("no recurrence (w_hh = 0)", ("lstm.weight_hh_l0",)), ("no input (w_ih = 0)", ("lstm.weight_ih_l0",)),
                   ("biases only", ("lstm.weight_hh_l0", "lstm.weight_ih_l0")), ("no biases", ("lstm.bias_ih_l0", "lstm.bias_hh_l0", "fc.bias"))):
    cfg = config.make_cfg({"num_agents": 3, "cylinder": {"max_num": 8, "min_num": 8}, "env": {"num_envs": E, "max_episode_length": 800}}, algo={"use_TP_net": 1})
    env = HideAndSeek(cfg, headless=True)
    with torch.no_grad():
        for k, v in env.TP.state_dict().items():
            if k in zero:
                v.zero_()
    env.set_seed(0); env.reset()
    g = torch.Generator(device="cpu").manual_seed(3)
    host = env.export_state()
    tpa = {k: v.cpu().numpy().copy() for k, v in env._tp_bufs.items() if k != "packed"}
    tpa["packed"] = np.zeros(16, np.uint8)
    for f, key in abi.TP_STATE_DICT_KEYS.items():
        tpa[f] = env.TP.state_dict()[key].detach().cpu().numpy().copy()
    tpa["history"][:] = 0
    O.tp_observe(env.hcfg, host, tpa, fill=True)
    errs = []
    for t in range(12):
        env.step(env.rand_step_input(torch.randn(E, 3, 4, generator=g).to(env.device)))
        host_now = env.export_state()
        O.tp_observe(env.hcfg, host_now, tpa, fill=False)
        d = np.abs(env._tp_bufs["pred"].cpu().numpy() - tpa["pred"]).reshape(E, -1).max(1)
        errs.append(float(d.max()))
        if "--comp" in sys.argv and name == "full":
            dd = np.abs(env._tp_bufs["pred"].cpu().numpy() - tpa["pred"]).reshape(E, -1)
            print("   step", t, "max error per prediction component:", " ".join(f"{v:.0e}" for v in dd.max(0)), "| envs above 1e-6:", np.nonzero(dd.max(1) > 1e-6)[0][:16])
        if "--detail" in sys.argv and name == "full" and d.max() > 2e-6:
            bad = np.nonzero(d > 2e-6)[0]
            hist = tpa["history"].reshape(E, 10, -1)
            print(f"   step {t}: {len(bad)} envs above 2e-6: {bad[:12]}; detect of those {host_now['detect'][bad[:12]].ravel()}; |frame| max per bad env {np.abs(hist[bad[:12]]).max((1, 2))}; "
                  f"|frame| max over good envs {np.abs(hist[d <= 2e-6]).max():.3f}; bad env window col0 {hist[bad[0], :, 0]}, cols1-6 of newest {hist[bad[0], -1, 1:7]}")
    if name == "full":          # determinism: the same filled window evaluated five times
        import ctypes as C
        outs = []
        for _ in range(5):
            assert env._lib.hns_tp_observe(env._env, 1, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
            torch.cuda.synchronize()
            outs.append(env._tp_bufs["pred"].cpu().numpy().copy())
        print("   repeatability (fill mode, 5 runs): max difference between runs", max(float(np.abs(o - outs[0]).max()) for o in outs))
        tpf = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in tpa.items()}
        O.tp_observe(env.hcfg, env.export_state(), tpf, fill=True)
        dd = np.abs(outs[0] - tpf["pred"]).reshape(E, -1).max(1)
        print("   fill mode against the oracle: max %.1e, envs above 1e-6: %s" % (dd.max(), np.nonzero(dd > 1e-6)[0][:16]))
        np.set_printoptions(precision=9, linewidth=250)
        hw = env._tp_bufs["history"].cpu().numpy().reshape(E, 10, -1)
        for b in np.nonzero(dd > 1e-6)[0][:4]:
            print("   bad env", b, "frame", hw[b, -1], "pred hip", outs[0].reshape(E, -1)[b][:6], "oracle", tpf["pred"].reshape(E, -1)[b][:6])
        good = np.nonzero(dd < 2e-7)[0][:2]
        for b in good:
            print("   good env", b, "frame", hw[b, -1])
        # windowed mode, repeated: restore the window, observe, compare between repetitions
        win = env._tp_bufs["history"].clone()
        reps = []
        for _ in range(4):
            env._tp_bufs["history"].copy_(win)
            assert env._lib.hns_tp_observe(env._env, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
            torch.cuda.synchronize()
            reps.append(env._tp_bufs["pred"].cpu().numpy().copy())
        tpw = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in tpa.items()}
        tpw["history"] = win.cpu().numpy().copy()
        O.tp_observe(env.hcfg, env.export_state(), tpw, fill=False)
        dd = np.abs(reps[0] - tpw["pred"]).reshape(E, -1).max(1)
        print("   windowed mode, 4 repetitions from the same window: max difference between runs %.1e; against the oracle max %.1e, envs above 1e-6: %s" % (
            max(float(np.abs(o - reps[0]).max()) for o in reps), dd.max(), np.nonzero(dd > 1e-6)[0][:16]))
    print(f"{name:28s} max |pred - oracle| per step: " + " ".join(f"{e:.1e}" for e in errs))
