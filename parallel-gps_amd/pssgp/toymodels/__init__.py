"""Toy signals of the reference under their original import path (`from pssgp.toymodels import sinu, obs_noise`;
pssgp/toymodels/data_funcs.py:10-97).  The functions live in pssgp/experiments/toy.py."""
from ..experiments.toy import comp_sinu, obs_noise, rect, sinu

__all__ = ["sinu", "comp_sinu", "rect", "obs_noise"]
