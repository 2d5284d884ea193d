"""``gpflow_pilco.moment_matching`` call surface on torch tensors + HIP kernels
(``gpflow_pilco/moment_matching/__init__.py:4-15``)."""
__all__ = (
    "Chain",
    "GaussianMatch",
    "GaussianMoments",
    "LinearOperatorDiag",
    "Moments",
    "moment_matching",
    "MomentMatch",
    "dispatcher",
    "register_type",
)

from .core import (Chain, LinearOperatorDiag, MomentMatch, Moments, dispatcher,
                   moment_matching, register_type)
from .gaussian import GaussianMatch, GaussianMoments
from . import maths, bijectors, components, models  # registers the handlers
