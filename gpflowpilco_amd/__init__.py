"""MI355X-native moment-matched GP rollout (drop-in for the hot path of GPflowPILCO).

Public surface mirrors the reference modules on that path:
``gpflowpilco_amd.moment_matching`` (dispatcher, GaussianMoments, GaussianMatch, moment_matching),
``gpflowpilco_amd.models`` (SVGP / GPR parameter containers),
``gpflowpilco_amd.dynamics`` (MomentMatchingEuler, rollouts) and the raw kernel front end
``gpflowpilco_amd.ops``.
"""
__version__ = "0.1.0"
