"""Experiment drivers on the HIP backend: the toy-signal regression mesh and a gradient-based sampler
(the protocols of pssgp/experiments/toy_models/{speed_and_stability,mcmc}.py of the reference)."""
