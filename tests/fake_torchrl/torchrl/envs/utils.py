"""step_mdp of torchrl 0.1.1: the next observation / done become the root of the tensordict the policy sees at t + 1."""
from tensordict import TensorDict


def step_mdp(tensordict, next_tensordict=None, keep_other=True, exclude_reward=True, exclude_done=False, exclude_action=True):
    nxt = tensordict.get("next")
    out = TensorDict({}, tensordict.batch_size, tensordict.device)
    if keep_other:
        for k in tensordict.keys():
            if k == "next" or (exclude_action and k == "action"):
                continue
            out.set(k, tensordict.get(k))
    for k in nxt.keys(True, True):
        key = k if isinstance(k, tuple) else (k,)
        if exclude_reward and key[-1] == "reward":
            continue
        if exclude_done and key[-1] == "done":
            continue
        out.set(k, nxt.get(k))
    if exclude_action:
        for k in list(out.keys(True, True)):
            key = k if isinstance(k, tuple) else (k,)
            if key[-1] == "action":
                out.exclude(k, inplace=True)
    return out
