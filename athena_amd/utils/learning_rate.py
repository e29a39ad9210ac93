"""Warm-up schedule and Adam (mirror of the reference's athena/utils/learning_rate.py:24-146).

The update itself is one fused HIP kernel over the flat parameter buffer
(``athena_clip_adam_step``): per-tensor clip_by_norm, Adam with Keras' bias correction, and the
schedule evaluated on the device from the step counter, so a whole training step can be replayed
from a HIP graph without host-side state.
"""
import torch

from .. import _lib, ops
from .hparam import register_and_parse_hparams


class WarmUpLearningSchedule:
    """lr = k * decay_rate^(step // decay_steps) * d^-0.5 * min(step^-0.5, step * warmup^-1.5)
    (learning_rate.py:52-58)."""

    def __init__(self, model_dim=512, warmup_steps=4000, k=1.0, decay_steps=99999999, decay_rate=1.0):
        self.model_dim, self.warmup_steps, self.k = float(model_dim), warmup_steps, k
        self.decay_steps, self.decay_rate = float(decay_steps), float(decay_rate)

    def __call__(self, step):
        step = float(step)
        arg1 = float("inf") if step == 0 else step ** -0.5
        arg2 = step * (self.warmup_steps ** -1.5)
        k = self.k * self.decay_rate ** (step // self.decay_steps)
        return k * self.model_dim ** -0.5 * min(arg1, arg2)


class ExponentialDecayLearningRateSchedule:
    """learning_rate.py:89-117"""

    def __init__(self, initial_lr=0.005, decay_steps=10000, decay_rate=0.5, start_decay_steps=30000,
                 final_lr=1e-5):
        self.initial_lr, self.decay_steps, self.decay_rate = initial_lr, float(decay_steps), float(decay_rate)
        self.start_decay_steps, self.final_lr = start_decay_steps, final_lr

    def __call__(self, step):
        factor = 1.0 if step < self.start_decay_steps else \
            self.decay_rate ** ((step - self.start_decay_steps) // self.decay_steps)
        return min(max(self.initial_lr * factor, self.final_lr), self.initial_lr)


class WarmUpAdam:
    """Adam(beta_1=0.9, beta_2=0.999, epsilon=1e-7) driven by WarmUpLearningSchedule
    (learning_rate.py:61-86).  ``iterations`` lives on the device (int64)."""
    default_config = {
        "d_model": 512,
        "warmup_steps": 8000,
        "k": 0.5,
        "decay_steps": 100000,
        "decay_rate": 1.0
    }

    def __init__(self, config=None, beta_1=0.9, beta_2=0.999, epsilon=1e-7, amsgrad=False, name="WarmUpAdam",
                 **kwargs):
        if amsgrad:
            raise NotImplementedError("amsgrad is not built")
        self.hparams = register_and_parse_hparams(self.default_config, config, cls=self.__class__)
        self.beta_1, self.beta_2, self.epsilon, self.name = beta_1, beta_2, epsilon, name
        self.lr = WarmUpLearningSchedule(self.hparams.d_model, self.hparams.warmup_steps, self.hparams.k,
                                         self.hparams.decay_steps, self.hparams.decay_rate)
        self.store = None
        self._host_iterations = 0

    # ---- state ------------------------------------------------------------------------------
    def attach(self, store):
        """Allocate the Adam slots for a model's ParamStore."""
        self.store = store
        dev = store.device
        self.m = torch.zeros(store.n_train, dtype=torch.float32, device=dev)
        self.v = torch.zeros(store.n_train, dtype=torch.float32, device=dev)
        self.step_t = torch.zeros(1, dtype=torch.int64, device=dev)
        self.norms = torch.zeros(len(store.train_slots), dtype=torch.float32, device=dev)
        self.lr_t = torch.zeros(1, dtype=torch.float32, device=dev)

    @property
    def iterations(self):
        """optimizer.iterations (host copy; synchronises)."""
        return int(self.step_t.item()) if self.store is not None else self._host_iterations

    def variables(self):
        return [self.step_t, self.m, self.v]

    def _cfg(self, clip_norm, grad_scale):
        h = self.hparams
        return _lib.AdamCfg(self.beta_1, self.beta_2, self.epsilon, float(clip_norm), float(h.d_model),
                            float(h.warmup_steps), float(h.k), float(h.decay_steps), float(h.decay_rate),
                            float(grad_scale), 0.0, 0.0, 0.0, 0.0)

    def apply_gradients(self, clip_norm=0.0, grad_scale=1.0):
        """clip_by_norm per tensor + Adam on the flat buffers, then mark the bf16 shadows stale."""
        st = self.store
        ops.clip_adam_step(st.params[:st.n_train], st.grads, self.m, self.v, st.seg_offsets, len(st.train_slots),
                           self.norms, self.step_t, self._cfg(clip_norm, grad_scale), self.lr_t)
        st.shadows_dirty = True

    def state_dict(self):
        return {"iterations": self.step_t.cpu(), "m": self.m.cpu(), "v": self.v.cpu()}

    def load_state_dict(self, state):
        self.step_t.copy_(state["iterations"])
        self.m.copy_(state["m"])
        self.v.copy_(state["v"])


class ExponentialDecayAdam(WarmUpAdam):
    """Adam(beta_1=0.9, beta_2=0.999, epsilon=1e-6) driven by ExponentialDecayLearningRateSchedule
    (learning_rate.py:120-146): the same fused clip + Adam kernel with the other schedule."""
    default_config = {
        "initial_lr": 0.005,
        "decay_steps": 10000,
        "decay_rate": 0.5,
        "start_decay_steps": 30000,
        "final_lr": 1e-5
    }

    def __init__(self, config=None, beta_1=0.9, beta_2=0.999, epsilon=1e-6, amsgrad=False, name="WarmUpAdam",
                 **kwargs):
        self.hparams = register_and_parse_hparams(self.default_config, config, cls=self.__class__)
        self.lr = ExponentialDecayLearningRateSchedule(self.hparams.initial_lr, self.hparams.decay_steps,
                                                       self.hparams.decay_rate, self.hparams.start_decay_steps,
                                                       self.hparams.final_lr)
        self.beta_1, self.beta_2, self.epsilon, self.name = beta_1, beta_2, epsilon, name
        self.store = None
        self._host_iterations = 0

    def _cfg(self, clip_norm, grad_scale):
        h = self.hparams
        return _lib.AdamCfg(self.beta_1, self.beta_2, self.epsilon, float(clip_norm), 1.0, 1.0, 1.0,
                            float(h.decay_steps), float(h.decay_rate), float(grad_scale), 1.0, float(h.initial_lr),
                            float(h.start_decay_steps), float(h.final_lr))
