"""On-device mirror-symmetry transforms for SymmetricRL-style trainers.

The reference only *publishes* index sets through `get_mirror_indices()` (env_locomotion.py:224-282, :761-840);
the trainers apply them in numpy on the host.  With observations living on the GPU the same transform is a
gather + sign flip on the device, so the learner's data never leaves HBM.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import torch


class MirrorTransform:
    """obs' = M_obs obs, act' = M_act act where M swaps the right/left index sets and negates the `neg` set."""

    def __init__(self, mirror_indices: Tuple[Sequence[int], ...], obs_dim: int, act_dim: int, device=None):
        neg_obs, right_obs, left_obs, neg_act, right_act, left_act = [torch.as_tensor(list(x), dtype=torch.long)
                                                                      for x in mirror_indices]
        self.obs_perm, self.obs_sign = self._build(obs_dim, neg_obs, right_obs, left_obs, device)
        self.act_perm, self.act_sign = self._build(act_dim, neg_act, right_act, left_act, device)

    @staticmethod
    def _build(dim, neg, right, left, device):
        perm = torch.arange(dim)
        perm[right], perm[left] = left.clone(), right.clone()
        sign = torch.ones(dim)
        sign[neg] = -1.0
        return perm.to(device), sign.to(device)

    def obs(self, x: torch.Tensor) -> torch.Tensor:
        return x[..., self.obs_perm] * self.obs_sign

    def act(self, a: torch.Tensor) -> torch.Tensor:
        return a[..., self.act_perm] * self.act_sign
