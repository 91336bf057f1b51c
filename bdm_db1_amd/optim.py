"""Learning-rate / weight-decay schedule with the reference's class name, constructor and
``step / get_lr / get_wd / state_dict / load_state_dict`` surface
(src/train_utils/optimizer_param_scheduler.py:22-234).  Host-side scalar math only; the values are
pinned against the reference by tests/golden/scheduler.npz."""
import math


class OptimizerParamScheduler(object):
    def __init__(self, optimizer, max_lr, min_lr, lr_warmup_steps, lr_decay_steps, lr_decay_style,
                 start_wd, end_wd, wd_incr_steps, wd_incr_style,
                 use_checkpoint_opt_param_scheduler=True, override_opt_param_scheduler=False):
        self.optimizer = optimizer
        self.max_lr = float(max_lr)
        self.min_lr = min_lr
        assert self.min_lr >= 0.0 and self.max_lr >= self.min_lr
        self.lr_warmup_steps = lr_warmup_steps
        self.num_steps = 0
        self.lr_decay_steps = lr_decay_steps
        assert self.lr_decay_steps > 0 and self.lr_warmup_steps < self.lr_decay_steps
        self.lr_decay_style = lr_decay_style
        self.start_wd, self.end_wd = start_wd, end_wd
        assert self.start_wd >= 0.0 and self.end_wd >= self.start_wd
        self.wd_incr_steps = wd_incr_steps
        self.wd_incr_style = wd_incr_style
        self.override_opt_param_scheduler = override_opt_param_scheduler
        self.use_checkpoint_opt_param_scheduler = use_checkpoint_opt_param_scheduler
        if self.override_opt_param_scheduler:
            assert not self.use_checkpoint_opt_param_scheduler, "both override and use-checkpoint are set."
        self.step(0)

    def get_wd(self):
        if self.num_steps > self.wd_incr_steps:
            return self.end_wd
        if self.wd_incr_style == "constant":
            assert self.start_wd == self.end_wd
            return self.end_wd
        ratio = float(self.num_steps) / float(self.wd_incr_steps)
        assert 0.0 <= ratio <= 1.0
        if self.wd_incr_style == "linear":
            coeff = ratio
        elif self.wd_incr_style == "cosine":
            coeff = 0.5 * (math.cos(math.pi * (1 - ratio)) + 1.0)
        else:
            raise Exception("{} weight decay increment style is not supported.".format(self.wd_incr_style))
        return self.start_wd + coeff * (self.end_wd - self.start_wd)

    def get_lr(self):
        if self.lr_warmup_steps > 0 and self.num_steps <= self.lr_warmup_steps:
            return self.max_lr * float(self.num_steps) / float(self.lr_warmup_steps)
        if self.lr_decay_style == "constant":
            return self.max_lr
        if self.num_steps > self.lr_decay_steps:
            return self.min_lr
        ratio = float(self.num_steps - self.lr_warmup_steps) / float(self.lr_decay_steps - self.lr_warmup_steps)
        assert 0.0 <= ratio <= 1.0
        if self.lr_decay_style == "linear":
            coeff = 1.0 - ratio
        elif self.lr_decay_style == "cosine":
            coeff = 0.5 * (math.cos(math.pi * ratio) + 1.0)
        else:
            raise Exception("{} decay style is not supported.".format(self.lr_decay_style))
        return self.min_lr + coeff * (self.max_lr - self.min_lr)

    def step(self, increment):
        self.num_steps += increment
        new_lr, new_wd = self.get_lr(), self.get_wd()
        for group in self.optimizer.param_groups:
            group["lr"] = new_lr * group.get("lr_mult", 1.0)
            group["weight_decay"] = new_wd * group.get("wd_mult", 1.0)

    def state_dict(self):
        return {k: getattr(self, k) for k in ("max_lr", "lr_warmup_steps", "num_steps", "lr_decay_style", "lr_decay_steps",
                                              "min_lr", "start_wd", "end_wd", "wd_incr_style", "wd_incr_steps")}

    def _check_and_set(self, cls_value, sd_value, name):
        if self.override_opt_param_scheduler:
            return cls_value
        if not self.use_checkpoint_opt_param_scheduler:
            assert cls_value == sd_value, f"OptimizerParamScheduler: class input value {cls_value} and checkpoint value {sd_value} for {name} do not match"
        return sd_value

    def load_state_dict(self, sd):
        for k in ("max_lr", "min_lr", "lr_warmup_steps", "lr_decay_steps", "lr_decay_style", "start_wd", "end_wd",
                  "wd_incr_steps", "wd_incr_style"):
            if k in sd:
                setattr(self, k, self._check_and_set(getattr(self, k), sd[k], k))
        self.num_steps = 0
        self.step(increment=sd.get("num_steps", 0))
