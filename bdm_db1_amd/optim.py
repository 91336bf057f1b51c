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


def _require(ok: bool, msg: str):
    # the reference guards its constructor and its checkpoint comparison with `assert` (optimizer_param_scheduler.py:43-65,164-169): callers
    # that catch AssertionError keep working; unlike a bare assert this does not vanish under `python -O`
    if not ok:
        raise AssertionError(msg)


class OptimizerParamScheduler:
    _LR_KEYS = ("max_lr", "min_lr", "lr_warmup_steps", "lr_decay_steps", "lr_decay_style")
    _WD_KEYS = ("start_wd", "end_wd", "wd_incr_steps", "wd_incr_style")
    # names older checkpoints use for the same quantities, tried first (optimizer_param_scheduler.py:179-218)
    _LEGACY = {"max_lr": ("start_lr",), "lr_warmup_steps": ("warmup_iter", "warmup_steps"), "lr_decay_steps": ("end_iter", "decay_steps"),
               "lr_decay_style": ("decay_style",), "num_steps": ("num_iters",)}

    def __init__(self, optimizer, max_lr, min_lr, lr_warmup_steps, lr_decay_steps, lr_decay_style, start_wd, end_wd, wd_incr_steps,
                 wd_incr_style, use_checkpoint_opt_param_scheduler=True, override_opt_param_scheduler=False):
        _require(min_lr >= 0.0 and float(max_lr) >= min_lr, f"need 0 <= min_lr <= max_lr, got {min_lr}, {max_lr}")
        _require(lr_decay_steps > 0 and lr_warmup_steps < lr_decay_steps, f"need lr_warmup_steps < lr_decay_steps, 0 < lr_decay_steps; got {lr_warmup_steps}, {lr_decay_steps}")
        _require(start_wd >= 0.0 and end_wd >= start_wd, f"need 0 <= start_wd <= end_wd, got {start_wd}, {end_wd}")
        _require(not (override_opt_param_scheduler and use_checkpoint_opt_param_scheduler), "both override and use-checkpoint are set.")
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
        if self.wd_incr_style == "constant" and self.num_steps <= self.wd_incr_steps:
            _require(self.start_wd == self.end_wd, "a constant weight-decay schedule needs start_wd == end_wd")   # (:78-80: checked when the value is asked for)
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

    def _from_checkpoint(self, sd: dict, key: str):
        """the checkpoint's value of ``key`` under its current or an older name; KeyError (the current name) when it has none"""
        for k in self._LEGACY.get(key, ()) + (key,):
            if k in sd:
                return sd[k]
        raise KeyError(key)

    def _adopt(self, sd: dict, key: str):
        """one schedule parameter from a checkpoint: kept as constructed under ``override_opt_param_scheduler``, otherwise the checkpoint's --
        which must equal the constructed one unless ``use_checkpoint_opt_param_scheduler`` (optimizer_param_scheduler.py:158-171)"""
        theirs = self._from_checkpoint(sd, key)           # (a missing mandatory key raises whatever the flags say, as the reference's lookup does)
        if self.override_opt_param_scheduler:
            return
        if not self.use_checkpoint_opt_param_scheduler:
            _require(getattr(self, key) == theirs, f"OptimizerParamScheduler: {key} is {getattr(self, key)} here and {theirs} in the checkpoint")
        setattr(self, key, theirs)

    def load_state_dict(self, sd: dict):
        """optimizer_param_scheduler.py:173-234, in its order: the learning-rate parameters (older checkpoints' names accepted, every one
        mandatory), then the step counter is ADVANCED by the checkpoint's count -- a scheduler is loaded once, freshly constructed, so this
        is "set"; loading twice adds twice there and here --, then the weight-decay parameters if the checkpoint has them (all four or
        KeyError).  As in the reference, the values written into the parameter groups by that advance still use the weight-decay
        parameters of the constructor; the next step() uses the loaded ones."""
        for key in self._LR_KEYS:
            self._adopt(sd, key)
        self.step(self._from_checkpoint(sd, "num_steps"))
        if "start_wd" in sd:
            for key in self._WD_KEYS:
                self._adopt(sd, key)
