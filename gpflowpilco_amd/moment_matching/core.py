"""Dispatcher and carrier types of the moment-matching API.

Mirrors ``gpflow_pilco/moment_matching/core.py:18-141``: ``Moments``, ``MomentMatch``,
``Chain``, ``dispatcher`` (type-based multiple dispatch on ``(type(x), type(obj))``),
``moment_matching`` and ``register_type``.  The reference uses
``gpflow.utilities.Dispatcher`` (multipledispatch); that package is not a dependency
here, so a small MRO-distance dispatcher with the same ``register`` / call contract
is provided.
"""
from __future__ import annotations

from dataclasses import dataclass
from functools import partial
from typing import Any, Callable, Dict, Hashable, Iterable, List, Tuple, Type, Union

import torch

__all__ = (
    "ArrayTypes", "Chain", "Dispatcher", "LinearOperatorDiag", "dispatcher", "get_type",
    "moment_matching", "Moments", "MomentMatch", "register_type",
)


class LinearOperatorDiag:
  """Stand-in for ``tf.linalg.LinearOperatorDiag`` (the diagonal covariances returned with
  ``full_output_cov=False``, models.py:293-296, and auto-detected at :302-311)."""

  def __init__(self, diag: torch.Tensor):
    self.diag = diag

  @property
  def dtype(self):
    return self.diag.dtype

  @property
  def shape(self):
    return tuple(self.diag.shape) + (self.diag.shape[-1],)

  def diag_part(self) -> torch.Tensor:
    return self.diag

  def to_dense(self) -> torch.Tensor:
    return torch.diag_embed(self.diag)

  def matmul(self, x: torch.Tensor) -> torch.Tensor:
    return self.diag.unsqueeze(-1) * x

  def __matmul__(self, other):
    if isinstance(other, LinearOperatorDiag):
      return LinearOperatorDiag(self.diag * other.diag)
    return self.matmul(other)

  def __rmatmul__(self, other: torch.Tensor) -> torch.Tensor:
    return other * self.diag.unsqueeze(-2)          # A @ diag(v): scale the columns

  def solve(self, rhs: torch.Tensor) -> torch.Tensor:
    return rhs / self.diag.unsqueeze(-1)

  @classmethod
  def identity_like(cls, mean: torch.Tensor, multiplier=1.0) -> "LinearOperatorDiag":
    """Stand-in for LinearOperatorIdentity / LinearOperatorScaledIdentity (maths.py:43,77)."""
    return cls(torch.ones_like(mean) * multiplier)


ArrayTypes = (torch.Tensor, LinearOperatorDiag)


class Dispatcher:
  """Multiple dispatch on the types of the leading positional arguments."""

  def __init__(self, name: str):
    self.name = name
    self._registry: List[Tuple[Tuple, Callable]] = []

  def register(self, *types):
    sig = tuple(t if isinstance(t, tuple) else (t,) for t in types)

    def decorator(fn):
      self._registry.append((sig, fn))
      return fn
    return decorator

  @staticmethod
  def _distance(cls: Type, candidates: Tuple[Type, ...]):
    best = None
    mro = cls.__mro__
    for cand in candidates:
      if cand is object:
        dist = len(mro)
      elif cand in mro:
        dist = mro.index(cand)
      else:
        continue
      best = dist if best is None else min(best, dist)
    return best

  def dispatch(self, *types: Type):
    best_fn, best_score = None, None
    for sig, fn in self._registry:
      if len(sig) != len(types):
        continue
      score = []
      for cls, cands in zip(types, sig):
        dist = self._distance(cls, cands)
        if dist is None:
          break
        score.append(dist)
      else:
        score = tuple(score)
        if best_score is None or score < best_score:
          best_fn, best_score = fn, score
    return best_fn

  def __call__(self, *args, **kwargs):
    arities = sorted({len(sig) for sig, _ in self._registry}, reverse=True)
    for n in arities:
      if n > len(args):
        continue
      fn = self.dispatch(*(type(a) for a in args[:n]))
      if fn is not None:
        return fn(*args, **kwargs)
    raise NotImplementedError(
        f"Could not find signature for {self.name}: <{', '.join(type(a).__name__ for a in args)}>")


_MomentMatchingCustomTypes: Dict[Hashable, Type] = dict()
dispatcher = Dispatcher("moment_matching")


def get_type(obj: Hashable) -> Type:
  return _MomentMatchingCustomTypes[obj]


def register_type(obj: Hashable, name: str = None, bases: Tuple = tuple(),
                  dict: Dict = None, exist_ok: bool = False) -> Type:
  """core.py:46-66: a dedicated type standing for a plain callable (e.g. ``torch.sin``)."""
  if obj in _MomentMatchingCustomTypes and not exist_ok:
    raise ValueError("Attempted to register a preexisting custom type")
  if name is None:
    name = f"{getattr(obj, '__module__', 'builtins')}.{getattr(obj, '__name__', repr(obj))}"
  new_type = _MomentMatchingCustomTypes[obj] = type(name, bases, dict or {})
  return new_type


@dataclass
class Moments:
  """core.py:69-110."""
  moments: Union[List, Tuple]
  centered: bool

  def __getitem__(self, index):
    return self.moments[index]

  def mean(self):
    return self[0]

  def covariance(self, dense: bool = None):
    m1, m2 = self[:2]
    if self.centered:
      Syy = m2
    elif isinstance(m2, LinearOperatorDiag):
      Syy = m2.to_dense() - m1.unsqueeze(-1) * m1.unsqueeze(-2)
    else:
      Syy = m2 - m1.unsqueeze(-1) * m1.unsqueeze(-2)
    if dense and isinstance(Syy, LinearOperatorDiag):
      Syy = Syy.to_dense()
    return Syy

  @property
  def ndim(self) -> int:
    return self[0].shape[-1]

  @property
  def dtype(self):
    dtype = self[0].dtype
    for moment in self[1:]:
      assert moment.dtype == dtype, ValueError("dtype of moments do not match")
    return dtype


@dataclass
class MomentMatch:
  x: Moments
  y: Moments


class Chain(tuple):
  """core.py:119-126: ops applied right-to-left."""

  def __new__(cls, *ops: Iterable[Callable]):
    return super().__new__(cls, ops)

  def __call__(self, x):
    for op in reversed(self):
      x = op(x)
    return x


@dispatcher.register(Moments, partial)
def _mm_partial(x: Moments, op: partial):
  return moment_matching(x, op.func, *op.args, **op.keywords)


def moment_matching(x: Moments, obj: Any, *args, **kwargs) -> MomentMatch:
  """core.py:134-141."""
  if isinstance(obj, (partial, Chain)):
    return dispatcher(x, obj, *args, **kwargs)
  if isinstance(obj, Hashable) and obj in _MomentMatchingCustomTypes:
    obj = get_type(obj)()
  return dispatcher(x, obj, *args, **kwargs)
