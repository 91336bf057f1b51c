"""Learning-rate and weight-decay schedules of DB1 pre-training (contract: the class name, constructor arguments, ``step`` / ``get_lr`` /
``get_wd`` / ``state_dict`` / ``load_state_dict`` of src/train_utils/optimizer_param_scheduler.py:22-234; values pinned by
tests/golden/scheduler.npz).

Both schedules are the same object: a RAMP from one value to another over a step interval with a shape ("linear", "cosine" or
"constant"), optionally behind a linear warm-up from zero.  ``lr_at`` / ``wd_at`` are pure functions of the step; the class only keeps the
step counter, writes the two values into the optimizer's parameter groups and (de)serialises the ten numbers of a checkpoint."""
import math

def _rise(x: float, style: str, what: str) -> float:
    """how far a ramp of shape ``style`` has risen at relative position x in [0, 1]: 0 at x = 0, 1 at x = 1"""
    if style == "linear":
        return x
    if style == "cosine":
        return 0.5 * (math.cos(math.pi * (1.0 - x)) + 1.0)
    raise ValueError(f"{style} {what} style is not supported (linear, cosine, constant)")


def lr_at(step: int, max_lr: float, min_lr: float, warmup: int, decay_steps: int, style: str) -> float:
    """linear warm-up 0 -> max_lr over ``warmup`` steps, then max_lr -> min_lr until ``decay_steps`` (optimizer_param_scheduler.py:101-134)"""
    if 0 < warmup and step <= warmup:
        return max_lr * step / warmup
    if style == "constant":
        return max_lr
    if step > decay_steps:
        return min_lr
    x = (step - warmup) / (decay_steps - warmup)
    # a falling ramp is a rising one read backwards; written out so that the rounding is the reference's (cos(pi x), not cos(pi (1 - (1 - x))))
    left = 1.0 - x if style == "linear" else (0.5 * (math.cos(math.pi * x) + 1.0) if style == "cosine" else _rise(x, style, "decay"))
    return min_lr + left * (max_lr - min_lr)


def wd_at(step: int, start_wd: float, end_wd: float, incr_steps: int, style: str) -> float:
    """start_wd -> end_wd over ``incr_steps`` steps (optimizer_param_scheduler.py:73-99)"""
    if step > incr_steps or style == "constant":
        return end_wd
    return start_wd + _rise(step / incr_steps, style, "weight decay increment") * (end_wd - start_wd)


class OptimizerParamScheduler:
    _LR_KEYS = ("max_lr", "min_lr", "lr_warmup_steps", "lr_decay_steps", "lr_decay_style")
    _WD_KEYS = ("start_wd", "end_wd", "wd_incr_steps", "wd_incr_style")

    def __init__(self, optimizer, max_lr, min_lr, lr_warmup_steps, lr_decay_steps, lr_decay_style, start_wd, end_wd, wd_incr_steps,
                 wd_incr_style, use_checkpoint_opt_param_scheduler=True, override_opt_param_scheduler=False):
        if not (0.0 <= min_lr <= max_lr):
            raise ValueError(f"need 0 <= min_lr <= max_lr, got {min_lr}, {max_lr}")
        if not (0 <= lr_warmup_steps < lr_decay_steps):
            raise ValueError(f"need lr_warmup_steps < lr_decay_steps, got {lr_warmup_steps}, {lr_decay_steps}")
        if not (0.0 <= start_wd <= end_wd) or (wd_incr_style == "constant" and start_wd != end_wd):
            raise ValueError(f"need 0 <= start_wd <= end_wd (equal for a constant schedule), got {start_wd}, {end_wd}")
        if override_opt_param_scheduler and use_checkpoint_opt_param_scheduler:
            raise ValueError("override_opt_param_scheduler and use_checkpoint_opt_param_scheduler exclude each other")
        self.optimizer = optimizer
        self.max_lr, self.min_lr = float(max_lr), min_lr
        self.lr_warmup_steps, self.lr_decay_steps, self.lr_decay_style = lr_warmup_steps, lr_decay_steps, lr_decay_style
        self.start_wd, self.end_wd, self.wd_incr_steps, self.wd_incr_style = start_wd, end_wd, wd_incr_steps, wd_incr_style
        self.use_checkpoint_opt_param_scheduler = use_checkpoint_opt_param_scheduler
        self.override_opt_param_scheduler = override_opt_param_scheduler
        self.num_steps = 0
        self.step(0)

    def get_lr(self) -> float:
        return lr_at(self.num_steps, *(getattr(self, k) for k in self._LR_KEYS))

    def get_wd(self) -> float:
        return wd_at(self.num_steps, *(getattr(self, k) for k in self._WD_KEYS))

    def step(self, increment: int):
        """advance by ``increment`` optimizer steps and write the two values into every parameter group (scaled by its lr_mult / wd_mult)"""
        self.num_steps += increment
        lr, wd = self.get_lr(), self.get_wd()
        for grp in self.optimizer.param_groups:
            grp["lr"] = lr * grp.get("lr_mult", 1.0)
            grp["weight_decay"] = wd * grp.get("wd_mult", 1.0)

    def state_dict(self) -> dict:
        sd = {k: getattr(self, k) for k in self._LR_KEYS + self._WD_KEYS}
        sd["num_steps"] = self.num_steps
        return sd

    def load_state_dict(self, sd: dict):
        """checkpointed schedule parameters win unless ``override_opt_param_scheduler``; with neither flag set they must agree"""
        for k in self._LR_KEYS + self._WD_KEYS:
            if k not in sd or self.override_opt_param_scheduler:
                continue
            if not self.use_checkpoint_opt_param_scheduler and getattr(self, k) != sd[k]:
                raise ValueError(f"OptimizerParamScheduler: {k} is {getattr(self, k)} here and {sd[k]} in the checkpoint")
            setattr(self, k, sd[k])
        self.num_steps = 0
        self.step(sd.get("num_steps", 0))
