"""Learning-rate schedules with the names/arguments of furnace/engine/lr_policy.py."""


class BaseLR(object):
    def get_lr(self, cur_iter):
        raise NotImplementedError


class PolyLR(BaseLR):
    """lr0 * (1 - it/T)^power  (lr_policy.py:18-26; used at train.py:93,133)."""

    def __init__(self, start_lr, lr_power, total_iters):
        self.start_lr = start_lr
        self.lr_power = lr_power
        self.total_iters = float(total_iters)

    def get_lr(self, cur_iter):
        frac = 1.0 - float(cur_iter) / self.total_iters
        return self.start_lr * frac ** self.lr_power


class MultiStageLR(BaseLR):
    """Piecewise-constant: lr_stages = [[until_epoch, lr], ...] (lr_policy.py:29-38)."""

    def __init__(self, lr_stages):
        if not isinstance(lr_stages, (list, tuple)) or len(lr_stages[0]) != 2:
            raise AssertionError('lr_stages must be list or tuple, with [iters, lr] format')
        self._stages = lr_stages

    def get_lr(self, epoch):
        for until, lr in self._stages:
            if epoch < until:
                return lr
        return None


class LinearIncreaseLR(BaseLR):
    """Linear warm-up from start_lr to end_lr over warm_iters (lr_policy.py:41-49)."""

    def __init__(self, start_lr, end_lr, warm_iters):
        self._start = start_lr
        self._step = (end_lr - start_lr) / warm_iters

    def get_lr(self, cur_epoch):
        return self._start + cur_epoch * self._step
