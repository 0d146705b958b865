"""Per-iteration learning-rate schedules with the reference's interface
(wesep/utils/schedulers.py:99-294): `step(cur_iter)` writes the lr into every param group
before the optimizer step; `state_dict()` is the object's __dict__ minus the optimizer."""
import math


class BaseClass:
    def __init__(self, optimizer, num_epochs, epoch_iter, initial_lr, final_lr, warm_up_epoch=6,
                 scale_ratio=1.0, warm_from_zero=False):
        self.optimizer = optimizer
        self.max_iter = num_epochs * epoch_iter
        self.initial_lr, self.final_lr = initial_lr, final_lr
        self.scale_ratio = scale_ratio
        self.current_iter = 0
        self.warm_up_iter = warm_up_epoch * epoch_iter
        self.warm_from_zero = warm_from_zero

    def get_multi_process_coeff(self):
        coeff = 1.0 * self.scale_ratio
        if self.current_iter < self.warm_up_iter:
            frac = self.current_iter / self.warm_up_iter
            if self.warm_from_zero:
                coeff = self.scale_ratio * frac
            elif self.scale_ratio > 1:
                coeff = (self.scale_ratio - 1) * frac + 1.0
        return coeff

    def get_current_lr(self):
        return 0.0

    def get_lr(self):
        return self.optimizer.param_groups[0]["lr"]

    def set_lr(self):
        lr = self.get_current_lr()
        for group in self.optimizer.param_groups:
            group["lr"] = lr

    def step(self, current_iter=None):
        if current_iter is not None:
            self.current_iter = current_iter
        self.set_lr()
        self.current_iter += 1

    def step_return_lr(self, current_iter=None):
        if current_iter is not None:
            self.current_iter = current_iter
        lr = self.get_current_lr()
        self.current_iter += 1
        return lr

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != "optimizer"}

    def load_state_dict(self, state_dict):
        self.__dict__.update(state_dict)


class ExponentialDecrease(BaseClass):
    def get_current_lr(self):
        decay = math.exp((self.current_iter / self.max_iter) * math.log(self.final_lr / self.initial_lr))
        return self.get_multi_process_coeff() * self.initial_lr * decay


class TriAngular2(BaseClass):
    """Cyclic triangular schedule whose amplitude halves each cycle (schedulers.py:225-294)."""

    def __init__(self, optimizer, num_epochs, epoch_iter, initial_lr, final_lr, warm_up_epoch=6,
                 scale_ratio=1.0, cycle_step=2, reduce_lr_diff_ratio=0.5):
        super().__init__(optimizer, num_epochs, epoch_iter, initial_lr, final_lr, warm_up_epoch, scale_ratio)
        self.reduce_lr_diff_ratio = reduce_lr_diff_ratio
        self.cycle_iter = cycle_step * epoch_iter
        self.step_size = self.cycle_iter // 2
        self.max_lr, self.min_lr = initial_lr, final_lr
        self.gap = self.max_lr - self.min_lr

    def get_current_lr(self):
        coeff = self.get_multi_process_coeff()
        point = self.current_iter % self.cycle_iter
        cycle = self.current_iter // self.cycle_iter
        self.max_lr = self.min_lr + self.gap * self.reduce_lr_diff_ratio ** cycle
        if point <= self.step_size:
            lr = self.min_lr + (self.max_lr - self.min_lr) * point / self.step_size
        else:
            lr = self.max_lr - (self.max_lr - self.min_lr) * (point - self.step_size) / self.step_size
        return coeff * lr
