"""Trajectory-prediction network inside the observation path (`algo.use_TP_net: 1`).

Reference: `TP_net` = LSTM(7+3A -> 64, 1 layer, batch_first) + Linear(64 -> 15) + tanh
(omni_drones/learning/mappo.py:572-589) evaluated INSIDE `HideAndSeek._compute_state_and_obs`
(omni_drones/envs/hide_and_seek/hideandseek.py:805-854) on a 10-frame history of
`[progress, evader pos (masked), evader vel (masked), pursuer positions]`; its 5 predicted evader
positions enter `state_self` as 15 relative coordinates (35-dim rows).

`TPNet` is the parameter container the learner trains (names match the reference, so its
checkpoints' `"TP"` entry loads unchanged).  In the env the forward pass does NOT go through this
module: `hns_tp_observe` (csrc/hns_tp.hip) reads the parameters in place and runs the window
shift, the LSTM on the fp32 matrix cores, the output layer and the row assembly in two launches
(the MIOpen LSTM this replaces took 2.15 ms per step at 65 536 envs — 70x the step kernel).
`TPObservation` is the same composition in plain torch, kept as the fp32 reference the tests
compare against.
"""
import collections

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


class TPObservation:
    """History + prediction + 35-dim row assembly in plain torch: the fp32 reference of hns_tp_observe
    (tests only; the env never calls it)."""

    def __init__(self, tp, num_agents, arena_size, max_height, max_episode_length, history_step=10, future_step=5,
                 mask_value=-5.0):
        self.tp, self.A = tp, num_agents
        self.arena_size, self.max_height, self.max_len = arena_size, max_height, max_episode_length
        self.history_step, self.future_step, self.mask_value = history_step, future_step, mask_value
        self.history = collections.deque(maxlen=history_step)   # never reset per env (hideandseek.py:825-830)

    @torch.no_grad()
    def __call__(self, obs_self20, drone_pos, target_pos, target_vel, progress, detect):
        """obs_self20 [E,A,20] (kernel), drone_pos [E,A,3], target_pos/vel [E,3], progress [E], detect [E] bool."""
        E, A = drone_pos.shape[:2]
        det = detect.reshape(E, 1).bool()
        mv = torch.full_like(target_pos, self.mask_value)
        frame = torch.cat([progress.reshape(E, 1), torch.where(det, target_pos, mv), torch.where(det, target_vel, mv),
                           drone_pos.reshape(E, -1)], dim=-1)                                   # :815-820
        if len(self.history) < self.history_step:
            for _ in range(self.history_step):
                self.history.append(frame)
        else:
            self.history.append(frame)
        tp_input = torch.stack(list(self.history), dim=1)                                     # [E,10,7+3A]
        pred = self.tp(tp_input).reshape(E, self.future_step, 3).clone()
        pred[..., :2] = pred[..., :2] * 0.5 * self.arena_size                                 # :835-836
        pred[..., 2] = (pred[..., 2] + 1.0) / 2.0 * self.max_height
        rpos_pred = (drone_pos.unsqueeze(2) - pred.unsqueeze(1)).reshape(E, A, -1)            # :844
        rt_unmasked = drone_pos - target_pos.unsqueeze(1)
        state_self = torch.cat([obs_self20[..., :3], rpos_pred, obs_self20[..., 3:]], dim=-1)  # :846-854
        state_drones = torch.cat([rt_unmasked, rpos_pred, obs_self20[..., 3:]], dim=-1)        # :873-880
        gt = target_pos.clone()                                                                # :839-842
        gt[..., :2] = gt[..., :2] / (0.5 * self.arena_size)
        gt[..., 2] = gt[..., 2] / self.max_height * 2.0 - 1.0
        tp_done = (progress <= (self.max_len - self.future_step)).unsqueeze(-1)                # :838
        return state_self, state_drones, {"TP_input": tp_input, "TP_groundtruth": gt, "TP_done": tp_done}
