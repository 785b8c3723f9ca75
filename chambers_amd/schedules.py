"""chambers.schedules on the host (reference: chambers/schedules.py:5-48).

`LinearWarmup(learning_rate, warmup_steps, ramp=True)`: a learning-rate schedule wrapper, evaluated on the host once per
step (the value is a scalar argument of chb_adamw).  Arithmetic is float32 like the reference's tf.float32 graph:
  ramp=True : step <  warmup_steps -> step * (lr(0) / warmup_steps); else lr(step - warmup_steps)      (:11-13,20-25)
  ramp=False: lr(step) * min(1, step / warmup_steps)                                                       (:26-29)
`learning_rate` may be a float, a zero-argument callable, or another schedule (an object with `__call__(step)` that is
marked as a schedule by deriving from `LearningRateSchedule`), :33-41."""
import numpy as np


class LearningRateSchedule:
    """Marker base class (keras.optimizers.schedules.LearningRateSchedule): `__call__(step)` -> learning rate."""

    def __call__(self, step):
        raise NotImplementedError

    def get_config(self):
        raise NotImplementedError

    @classmethod
    def from_config(cls, config):
        return cls(**config)


class LinearWarmup(LearningRateSchedule):
    def __init__(self, learning_rate, warmup_steps, ramp=True):
        self.learning_rate = learning_rate
        self.warmup_steps = np.float32(warmup_steps)
        self.ramp = ramp
        if ramp:
            self.step_size = np.float32(self._get_learning_rate(0)) / np.float32(warmup_steps)

    def __call__(self, step):
        step = np.float32(step)
        if self.ramp:
            if step < self.warmup_steps:
                return np.float32(step * self.step_size)
            return np.float32(self._get_learning_rate(step - self.warmup_steps))
        lr_mult = np.minimum(np.float32(1.0), step / self.warmup_steps)
        return np.float32(np.float32(self._get_learning_rate(step)) * lr_mult)

    def _get_learning_rate(self, step):
        if isinstance(self.learning_rate, LearningRateSchedule):
            return self.learning_rate(step)
        if callable(self.learning_rate):
            return self.learning_rate()
        return self.learning_rate

    def get_config(self):
        return {"learning_rate": self.learning_rate, "warmup_steps": self.warmup_steps, "ramp": self.ramp}
