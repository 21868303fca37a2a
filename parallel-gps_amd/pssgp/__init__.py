"""pssgp on MI355X: the reference package's names for its parallel Kalman hot path.

Mirrors the import surface of the reference package (/root/reference/pssgp/__init__.py is
empty; users import `pssgp.model.StateSpaceGP`, `pssgp.kernels.*`, `pssgp.kalman.*`).
Host code is plain numpy; all filtering/smoothing/discretisation arithmetic runs in the
hand-written HIP library `libpgps.so` (see `pssgp/_backend.py`, `include/pgps.h`).
"""
__version__ = "0.1.0"
