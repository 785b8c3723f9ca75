"""chambers.optimizers.AdamW on MI355X (reference: chambers/optimizers.py:10-189 WeightDecayExtension, :372-464 AdamW).

Same constructor arguments and `_is_decay_allowed` regex semantics (decay_include / decay_exclude on variable names,
:169-181).  `apply(engine)` runs the fused HIP update over the engine's flat parameter buffer: per variable, decay first
(`var -= wd * var`, wd NOT multiplied by lr, :147-155), then the keras Adam epsilon-hat update."""
import re


class AdamW:
    def __init__(self, weight_decay, decay_include=None, decay_exclude=None, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-07,
                 amsgrad=False, name="AdamW", **kwargs):
        if decay_include is not None and decay_exclude is not None:
            raise ValueError("Got both `decay_include` and `decay_exclude` arguments. Use only `decay_include` or `decay_exclude`.")
        if amsgrad:
            raise ValueError("amsgrad is not implemented by the fused MI355X update")
        self.weight_decay = weight_decay
        self.decay_include = list(decay_include) if decay_include is not None else None
        self.decay_exclude = list(decay_exclude) if decay_exclude is not None else None
        self.learning_rate, self.beta_1, self.beta_2, self.epsilon, self.amsgrad, self.name = learning_rate, beta_1, beta_2, epsilon, amsgrad, name

    def _is_decay_allowed(self, var_name):
        """optimizers.py:169-181."""
        if self.decay_include is not None:
            return any(re.search(n, var_name) is not None for n in self.decay_include)
        if self.decay_exclude is not None:
            return not any(re.search(n, var_name) is not None for n in self.decay_exclude)
        return True

    def decay_fn(self, cfg=None):
        """Predicate for chambers_amd.engine.build_param_table(decay_fn=...) / ViTEngine(decay_fn=...).  With the model's
        ViTConfig the regexes are matched against the reference's Keras variable names (`var.name`, optimizers.py:169-181)
        via engine.keras_variable_names; without it, against the engine's own tensor names."""
        if cfg is None:
            return self._is_decay_allowed
        from .engine import decay_fn_from_variable_predicate
        return decay_fn_from_variable_predicate(cfg, self._is_decay_allowed)

    def _value(self, v, step):
        """Hyper-parameters may be schedules (keras LearningRateSchedule semantics: called with `iterations`)."""
        return float(v(step)) if callable(v) else float(v)

    def apply(self, engine):
        step = engine.opt_step
        engine.adamw_step(learning_rate=self._value(self.learning_rate, step), beta_1=self.beta_1, beta_2=self.beta_2, epsilon=self.epsilon,
                          weight_decay=self._value(self.weight_decay, step))

    def apply_gradients(self, grads_and_vars, name=None, **kwargs):
        """keras OptimizerV2.apply_gradients for models composed from the stand-alone layers: an iterable of (gradient, variable)
        pairs, variable = chambers_amd._keras_like.Variable (gradient None: taken from `variable.value.grad`, where
        `loss.backward()` left it).  Per variable the fused HIP update of `apply` (chb_adamw): decay first where
        `_is_decay_allowed(variable.name)` (optimizers.py:147-155,169-181), then keras Adam.  Adam moments live in the optimizer,
        keyed by variable, as keras slots do; `iterations` counts calls."""
        import numpy as np
        import torch
        from . import kernels as K
        step = getattr(self, "iterations", 0)
        lr = np.float32(self._value(self.learning_rate, step))
        wd = self._value(self.weight_decay, step)
        t = step + 1
        b1, b2 = np.float32(self.beta_1), np.float32(self.beta_2)
        lr_t = float(lr * np.sqrt(np.float32(1.0) - np.power(b2, np.float32(t))) / (np.float32(1.0) - np.power(b1, np.float32(t))))
        import weakref
        slots = self.__dict__.setdefault("_slots", weakref.WeakKeyDictionary())     # Adam moments per variable, dropped with it
        for grad, var in grads_and_vars:
            g = var.value.grad if grad is None else grad
            if g is None:
                continue
            p = var.value.detach()
            n = p.numel()
            n4 = (n + 3) // 4 * 4
            st = slots.get(var)
            if st is None:
                st = slots[var] = (torch.zeros(n4, dtype=torch.float32, device=p.device), torch.zeros(n4, dtype=torch.float32, device=p.device))
            if n4 != n or not p.is_contiguous():          # odd sizes: update a padded copy (chb_adamw works on float4)
                pp = torch.zeros(n4, dtype=torch.float32, device=p.device)
                pp[:n] = p.reshape(-1)
                gg = torch.zeros(n4, dtype=torch.float32, device=p.device)
                gg[:n] = g.detach().reshape(-1)
            else:
                pp, gg = p.view(-1), g.detach().to(torch.float32).contiguous().view(-1)
            K.adamw(pp, gg, st[0], st[1], None, lr_t, float(self.beta_1), float(self.beta_2), float(self.epsilon),
                    float(wd) if self._is_decay_allowed(var.name) else 0.0)
            if pp.data_ptr() != p.data_ptr():
                with torch.no_grad():
                    var.value.copy_(pp[:n].reshape(p.shape))
        self.iterations = t
        from .layers import autograd as _ag
        _ag.weights_written()        # chb_adamw rewrote the variables through raw pointers: cached bf16 operand images are stale

    def get_config(self):
        return {"name": self.name, "learning_rate": self.learning_rate, "beta_1": self.beta_1, "beta_2": self.beta_2, "epsilon": self.epsilon,
                "amsgrad": self.amsgrad, "weight_decay": self.weight_decay, "decay_include": self.decay_include,
                "decay_exclude": self.decay_exclude}
