"""ValueNorm with device-resident statistics (reference: harl/common/valuenorm.py:7-92).

State = one fp32 device tensor ``stats = [running_mean, running_mean_sq, debiasing_term]`` that the GAE,
advantage and critic-loss kernels read directly -- the reference's per-step NumPy <-> torch round trips
(3 per time step inside compute_returns) disappear.  ``state_dict()`` keeps the reference's three keys.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, stream


class ValueNorm:
    def __init__(self, input_shape=1, norm_axes=1, beta=0.99999, per_element_update=False, epsilon=1e-5,
                 device=torch.device("cuda:0")):
        if input_shape != 1 or norm_axes != 1 or per_element_update:
            raise NotImplementedError("only the scalar-value configuration used by the on-policy runners")
        _lib.require_gpu(torch.device(device))
        self.device = torch.device(device)
        self.beta = float(beta)
        self.epsilon = float(epsilon)
        self.stats = torch.zeros(3, dtype=torch.float32, device=self.device)
        self._sums = torch.zeros(2, dtype=torch.float64, device=self.device)

    # -- the reference exposes these three tensors by name
    @property
    def running_mean(self):
        return self.stats[0:1]

    @property
    def running_mean_sq(self):
        return self.stats[1:2]

    @property
    def debiasing_term(self):
        return self.stats[2]

    def running_mean_var(self):
        d = self.stats[2].clamp(min=self.epsilon)
        mean = self.stats[0:1] / d
        var = (self.stats[1:2] / d - mean ** 2).clamp(min=1e-2)
        return mean, var

    @torch.no_grad()
    def update(self, input_vector, idx: Optional[torch.Tensor] = None, count: Optional[int] = None, reduce_fn=None,
               local_count: Optional[int] = None):
        """EMA update from a minibatch of returns (valuenorm.py:47-64).  ``reduce_fn`` (data-parallel
        all-reduce of the fp64 {sum, sumsq}) and the global ``count`` are supplied by the sharded critic;
        ``local_count`` = number of local rows when it is not idx.numel() / x.numel() (0: this rank only reduces)."""
        x = _as_dev(input_vector, self.device).reshape(-1)
        m = x.numel() if idx is None else idx.numel()
        if local_count is not None and idx is None:
            m = local_count
        self._sums.zero_()
        if m > 0:
            call("harl_sum_sumsq", ptr(x), ptr(idx), m, ptr(self._sums), stream())
        if reduce_fn is not None:
            reduce_fn(self._sums)
        call("harl_valuenorm_apply", ptr(self.stats), ptr(self._sums), float(count if count is not None else m),
             self.beta, stream())

    def normalize(self, input_vector):
        x = _as_dev(input_vector, self.device)
        mean, var = self.running_mean_var()
        return (x - mean[None]) / torch.sqrt(var)[None]

    def denormalize(self, input_vector):
        """Returns a host NumPy array like the reference (valuenorm.py:78-92); the update path never calls it."""
        x = _as_dev(input_vector, self.device)
        mean, var = self.running_mean_var()
        return (x * torch.sqrt(var)[None] + mean[None]).cpu().numpy()

    def state_dict(self):
        return {"running_mean": self.stats[0:1].clone(), "running_mean_sq": self.stats[1:2].clone(),
                "debiasing_term": self.stats[2].clone()}

    def load_state_dict(self, sd):
        if not sd:  # checkpoints written by the reference on a GPU have an empty ValueNorm state_dict
            return
        self.stats[0] = torch.as_tensor(sd["running_mean"]).reshape(-1)[0]
        self.stats[1] = torch.as_tensor(sd["running_mean_sq"]).reshape(-1)[0]
        self.stats[2] = torch.as_tensor(sd["debiasing_term"]).reshape(-1)[0]

    def to(self, *a, **k):
        return self


def _as_dev(x, device) -> torch.Tensor:
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    elif x.dtype == torch.float32 and x.device == device and x.is_contiguous():
        return x  # already in place (the device-resident buffers): not worth a dispatcher round trip per argument and update
    return x.to(device=device, dtype=torch.float32).contiguous()
