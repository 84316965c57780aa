"""Plain-torch restatement of the TP branch of `_compute_state_and_obs` (hideandseek.py:805-854): window,
TP_net forward (torch.nn.LSTM), rescale, 35-value rows.  TEST REFERENCE ONLY — the env runs `hns_tp_observe`."""
import collections

import torch


class TPObservation:
    """History + prediction + 35-dim row assembly in plain torch: the fp32 reference of hns_tp_observe
    (tests only; the env never calls it)."""

    def __init__(self, tp, num_agents, arena_size, max_height, max_episode_length, history_step=10, future_step=5,
                 mask_value=-5.0, cylinder_size=None):
        self.tp, self.A = tp, num_agents
        self.arena_size, self.max_height, self.max_len = arena_size, max_height, max_episode_length
        self.history_step, self.future_step, self.mask_value = history_step, future_step, mask_value
        self.cylinder_size = cylinder_size                      # not None: task.use_obstacles (hideandseek.py:808-816)
        self.history = collections.deque(maxlen=history_step)   # never reset per env (hideandseek.py:825-830)

    @torch.no_grad()
    def __call__(self, obs_self20, drone_pos, target_pos, target_vel, progress, detect, cylinders=None):
        """obs_self20 [E,A,20] (kernel), drone_pos [E,A,3], target_pos/vel [E,3], progress [E], detect [E] bool."""
        E, A = drone_pos.shape[:2]
        det = detect.reshape(E, 1).bool()
        mv = torch.full_like(target_pos, self.mask_value)
        frame = torch.cat([progress.reshape(E, 1), torch.where(det, target_pos, mv), torch.where(det, target_vel, mv),
                           drone_pos.reshape(E, -1)], dim=-1)                                   # :815-820
        if self.cylinder_size is not None:
            frame = torch.cat([frame, torch.cat([cylinders[..., :2], torch.full_like(cylinders[..., :1], self.cylinder_size)],
                                                dim=-1).reshape(E, -1)], dim=-1)                   # :808-816
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
