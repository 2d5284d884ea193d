"""Minimal bijectors standing in for the tensorflow_probability ones the policy head uses
(``examples/cartpole_swingup/swingup_loops.py:85-91``): Chain[Scale, Shift, NormalCDF]."""
from __future__ import annotations

from typing import Sequence

import torch

from .special import ndtr


class Bijector:
  def __call__(self, x):
    return self.forward(x)


class Shift(Bijector):
  def __init__(self, shift):
    self.shift = shift

  def forward(self, x):
    return x + self.shift


class Scale(Bijector):
  def __init__(self, scale):
    self.scale = scale

  def forward(self, x):
    return x * self.scale


class NormalCDF(Bijector):
  def forward(self, x):
    return ndtr(x)


class Chain(Bijector):
  """tfb.Chain: bijectors are applied right to left."""

  def __init__(self, bijectors: Sequence[Bijector]):
    self.bijectors = list(bijectors)

  def forward(self, x):
    for b in reversed(self.bijectors):
      x = b(x)
    return x
