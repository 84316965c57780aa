"""Trajectory-prediction network inside the observation path (`algo.use_TP_net: 1`).

Reference: `TP_net` = LSTM(7+3A -> 64, 1 layer, batch_first) + Linear(64 -> 15) + tanh
(omni_drones/learning/mappo.py:572-589) evaluated INSIDE `HideAndSeek._compute_state_and_obs`
(omni_drones/envs/hide_and_seek/hideandseek.py:805-854) on a 10-frame history of
`[progress, evader pos (masked), evader vel (masked), pursuer positions]`; its 5 predicted evader
positions enter `state_self` as 15 relative coordinates (35-dim rows).

`TPNet` is the parameter container the learner trains (names match the reference, so its
checkpoints' `"TP"` entry loads unchanged).  In the env the forward pass does NOT go through this
module: `hns_tp_observe` (csrc/hns_tp.hip) reads the parameters in place and runs the window
shift, the LSTM on the matrix cores (`v_mfma_f32_32x32x16_f16` on operands split into two fp16 terms, fp32 accumulation:
within 1.5e-7 of the fp32 LSTM), the output layer and the row assembly in two launches
(the MIOpen LSTM this replaces took 2.15 ms per step at 65 536 envs — 70x the step kernel).
The same composition in plain torch — the fp32 reference the tests compare against — lives in
`tests/tp_reference.py`, outside the package: the env has no alternative path.
"""
import torch
import torch.nn as nn


class TPNet(nn.Module):
    def __init__(self, input_dim, output_dim, future_predcition_step, window_step):
        super().__init__()
        self.hidden_dim, self.num_layers = 64, 1
        self.future_predcition_step, self.window_step = future_predcition_step, window_step
        self.lstm = nn.LSTM(input_dim, self.hidden_dim, self.num_layers, batch_first=True)
        self.fc = nn.Linear(self.hidden_dim, output_dim)

    def forward(self, x):
        out, _ = self.lstm(x)                       # zero initial state, as mappo.py:582-585
        return torch.tanh(self.fc(out[:, -1, :]))
