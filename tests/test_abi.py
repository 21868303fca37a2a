"""The C-ABI library loads and exports every symbol include/pgps.h declares; error paths that
need no GPU; the sequential (parallel=False) host mode against the oracle.  No GPU compute."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import np_oracle as O
from tests.conftest import make_times, relerr, sample_series

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from pssgp import _backend
    return _backend.load_library()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "pgps.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(pgps_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 35
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_version_and_strerror(lib):
    assert lib.pgps_version() >= 100
    assert lib.pgps_strerror(0).decode() == "ok"
    assert "dimension" in lib.pgps_strerror(-2).decode()
    for i, name in enumerate(["k_filter_reduce", "k_filter_apply", "k_smoother_reduce", "k_smoother_apply",
                              "k_ll_finalize", "k_discretise"]):
        assert lib.pgps_kernel_name(i).decode() == name


def test_no_silent_cpu_fallback(lib):
    """Without a GPU the parallel path must raise, not compute somewhere else."""
    from pssgp import _backend
    n = ctypes.c_int(-1)
    assert lib.pgps_device_count(ctypes.byref(n)) == 0
    if n.value > 0:
        pytest.skip("a GPU is visible here")
    from pssgp.kalman.parallel import pkf
    from pssgp.kernels import Matern32
    ssm = O.get_ssm(Matern32().get_sde(), make_times(10), 0.1)
    with pytest.raises(_backend.PgpsError):
        pkf(ssm, np.zeros(10))
    h = ctypes.c_void_p()
    assert lib.pgps_create(0, ctypes.byref(h)) == -6          # PGPS_E_NO_DEVICE


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_sequential_mode_matches_oracle(dtype):
    """kf / ks / kfs (parallel=False): host C++ in libpgps vs the numpy oracle."""
    from pssgp.kalman.sequential import kf, kfs
    from pssgp.kernels import Matern52, RBF
    for k in (Matern52(1., 0.6), RBF(1., 0.8, order=8, balancing_iter=10)):
        t = make_times(600, seed=4)
        ssm = O.get_ssm(k.get_sde(), t, 0.1)
        y = sample_series(ssm, seed=4, nan_frac=0.2)
        ssm_t = tuple(np.asarray(a, dtype) for a in ssm)
        fms, fPs, ll, mps, Pps = kf(ssm_t, y[:, None].astype(dtype), True, True)
        sms, sPs = kfs(ssm_t, y.astype(dtype))
        of, oP, oll = O.kf(ssm, y, True)
        os_, osP = O.kfs(ssm, y)
        tol = 1e-10 if dtype == np.float64 else 2e-3
        assert relerr(fms, of) < tol and relerr(fPs, oP) < tol
        assert relerr(sms, os_) < tol and relerr(sPs, osP) < tol
        assert abs(float(ll) - oll) < tol * abs(oll)


def test_generated_asm_header_is_current():
    """csrc/pgps_rc_asm.h (the DPP building blocks of the row-cooperative kernels) is generated: the committed
    file must be what tools/gen_rc_asm.py writes, and every block must follow the hazard rules it documents."""
    import io
    import os
    import re
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    try:
        import gen_rc_asm
    finally:
        sys.path.pop(0)
    buf = io.StringIO()
    gen_rc_asm.emit(buf)
    text = open(os.path.join(root, "parallel-gps_amd", "csrc", "pgps_rc_asm.h")).read()
    assert buf.getvalue() == text
    blocks = re.findall(r"asm volatile\((.*?)\);", text, flags=re.S)
    assert len(blocks) > 180          # fp64 and fp32 sets
    for b in blocks:
        lines = [ln.strip().strip('"') for ln in b.split("\n") if "v_fmac" in ln or "s_nop" in ln]
        assert lines[0].startswith("s_nop 4"), "every block opens with the DPP entry wait states"
        written = set()
        for ln in lines[1:]:
            m = re.match(r"v_fmac_f(?:64|32)_dpp %(\d+), %(\d+), %(\d+) row_newbcast", ln)
            assert m, ln
            dst, src0 = int(m.group(1)), int(m.group(2))
            # a DPP source is never a register an earlier instruction of the block wrote, except the
            # elimination's own accumulator (read and written by the same instruction, after its last DPP use)
            assert src0 not in written or src0 == dst, ln
            written.add(dst)


def test_rccl_and_roctx_are_loaded_on_first_use_not_linked():
    """libpgps.so must load where neither RCCL nor roctx is installed (a single-GPU user), and a process that already
    carries a copy of RCCL (torch.distributed's nccl backend bundles one) must keep using that one: no DT_NEEDED entry
    for either library, and pgps_comm_library says which copy the loader picked."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "parallel-gps_amd", "pssgp", "libpgps.so")
    needed = subprocess.run(["readelf", "-d", so], stdout=subprocess.PIPE, text=True, check=True).stdout
    needed = [ln for ln in needed.splitlines() if "NEEDED" in ln]
    assert needed and not any("rccl" in ln or "roctx" in ln for ln in needed), needed
    code = ("import sys; sys.path.insert(0, %r); from pssgp import _backend as B; print(B.Context.comm_library())"
            % os.path.join(root, "parallel-gps_amd"))
    alone = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True, check=True).stdout.strip()
    assert alone.startswith("librccl") or alone.startswith("/opt/rocm") or alone.startswith("unavailable"), alone
    with_torch = subprocess.run([sys.executable, "-c", "import torch, torch.distributed; " + code], stdout=subprocess.PIPE,
                                text=True, check=True).stdout.strip()
    assert "already in the process" in with_torch, with_torch         # torch's bundled librccl.so.1, not a second copy
