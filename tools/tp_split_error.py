"""Accuracy of the two-term fp16 split used by hns_tp_lstm_kernel, emulated in numpy on the reference golden weights/window: plain fp32 and the split against fp64."""
import numpy as np
g = np.load('/root/repo/tests/golden/g_tp_obs.npz')
Wih, Whh = g['w_lstm_weight_ih_l0'].astype(np.float32), g['w_lstm_weight_hh_l0'].astype(np.float32)
b = (g['w_lstm_bias_ih_l0'] + g['w_lstm_bias_hh_l0']).astype(np.float32)
Wfc, bfc = g['w_fc_weight'].astype(np.float32), g['w_fc_bias'].astype(np.float32)
X = g['TP_input'][-1].astype(np.float32)            # [48,10,16]
rng = np.random.default_rng(0)

def split(v):
    v = v.astype(np.float32)
    hi = v.astype(np.float16)
    lo = ((v - hi.astype(np.float32)) * np.float32(2048)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)

def mm_split(W, v):      # W [G,K], v [N,K] -> [N,G], fp32 accumulation per 16-wide chunk
    W1, W2 = split(W); v1, v2 = split(v)
    K = W.shape[1]
    acc = np.zeros((v.shape[0], W.shape[0]), np.float32)
    for k0 in range(0, K, 16):
        s = slice(k0, k0 + 16)
        acc = (acc.astype(np.float64) + v2[:, s] @ W1[:, s].T).astype(np.float32)
        acc = (acc.astype(np.float64) + v1[:, s] @ W2[:, s].T).astype(np.float32)
    return acc

def mm_hi(W, v, acc):
    W1, _ = split(W); v1, _ = split(v)
    for k0 in range(0, W.shape[1], 16):
        s = slice(k0, k0 + 16)
        acc = (acc.astype(np.float64) + v1[:, s] @ W1[:, s].T).astype(np.float32)
    return acc

def lstm(X, Wih, Whh, b, Wfc, bfc, mode):
    N, T, I = X.shape
    dt = np.float64 if mode == 'f64' else np.float32
    h = np.zeros((N, 64), dt); c = np.zeros((N, 64), dt)
    sig = lambda z: 1 / (1 + np.exp(-z))
    for t in range(T):
        x = X[:, t].astype(dt)
        if mode == 'split':
            v = np.concatenate([x, h], 1); W = np.concatenate([Wih, Whh], 1)
            lo = mm_split(W, v)
            acc = (lo * np.float32(1 / 2048) + b).astype(np.float32)
            z = mm_hi(W, v, acc)
        else:
            z = (x @ Wih.T.astype(dt) + h @ Whh.T.astype(dt) + b.astype(dt)).astype(dt)
        i, f, gg, o = sig(z[:, :64]), sig(z[:, 64:128]), np.tanh(z[:, 128:192]), sig(z[:, 192:])
        c = (f * c + i * gg).astype(dt); h = (o * np.tanh(c)).astype(dt)
    if mode == 'split':
        lo = mm_split(Wfc, h); acc = (lo * np.float32(1 / 2048) + bfc).astype(np.float32); out = mm_hi(Wfc, h, acc)
    else:
        out = h @ Wfc.T.astype(dt) + bfc.astype(dt)
    return np.tanh(out)

for name, scale, prog in (("ref init", 1.0, None), ("x3 weights", 3.0, None), ("x3, progress~800", 3.0, 790.0), ("x8 weights", 8.0, 790.0)):
    Xs = X.copy()
    if prog is not None: Xs[:, :, 0] = prog + np.arange(10)[None, :]
    a = [w * np.float32(scale) for w in (Wih, Whh, b, Wfc, bfc)]
    r64 = lstm(Xs, *a, 'f64'); r32 = lstm(Xs, *a, 'f32'); rs = lstm(Xs, *a, 'split')
    print("%-18s |f32-f64| %.2e   |split-f64| %.2e   |split-f32| %.2e" % (name, np.abs(r32 - r64).max(), np.abs(rs - r64).max(), np.abs(rs - r32).max()))
