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

    def get_config(self):
        return {"name": self.name, "learning_rate": self.learning_rate, "beta_1": self.beta_1, "beta_2": self.beta_2, "epsilon": self.epsilon,
                "amsgrad": self.amsgrad, "weight_decay": self.weight_decay, "decay_include": self.decay_include,
                "decay_exclude": self.decay_exclude}
