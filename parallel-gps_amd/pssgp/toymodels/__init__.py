"""The reference's import path for its synthetic signals (`from pssgp.toymodels import sinu, obs_noise`, used by its
experiment scripts and notebooks; pssgp/toymodels/data_funcs.py:10-94).  The functions live in
`pssgp.experiments.toy`; this module only keeps the old names importable."""
from ..experiments.toy import comp_sinu, obs_noise, rect, sinu

__all__ = ["sinu", "comp_sinu", "rect", "obs_noise"]
