#!/usr/bin/env python3
"""Build gate: no scratch memory in kernels that read other lanes' registers.

The cooperative kernel families (pgps::rc, pgps::rc2, pgps::qc) pass operands between lanes in REGISTERS -- DPP
row_newbcast operands, v_permlane16_swap, ds_bpermute, v_readlane.  A register that the compiler spills to scratch is
stored and reloaded under the EXEC mask of the moment: inside a divergent region (the `if (lv && real)` load blocks, a
predicated store) the inactive lanes' copy is neither saved nor restored, and a later cross-lane read of such a lane
returns whatever the physical register held before -- wrong numbers with no fault (DESIGN.md section 4k has the case this
rule comes from).  So: every kernel of those namespaces must compile to ScratchSize = 0, or sit on the allow-list below
with the reason why its spills are harmless.

Usage: scratch_gate.py build/*.res   (the Makefile writes hipcc's -Rpass-analysis=kernel-resource-usage remarks of every
cooperative unit to build/<unit>.res and runs this after linking; exit status 1 fails the build)."""
import re
import shutil
import subprocess
import sys

GATED = ("pgps::rc::", "pgps::rc2::", "pgps::qc::", "pgps::k_pkfs_resident")

# units that legitimately hold no gated kernel (their .res may be passed by a glob): rc2_32 -- the two-rows kernels end at
# d = 23 in fp64 and d = 31 in fp32, the unit only carries the dispatch stubs
EXPECT_NONE = ("rc2_32.res",)

# (regular expression on the demangled kernel name, reason).  Keep it short and argued.
ALLOW = [
]


def demangle(names):
    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    try:
        out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return [o.strip() for o in out[:len(names)]]
    except Exception:           # noqa: BLE001  (no demangler: mangled names still carry the namespaces as 4pgps2rc...)
        return names


def kernels(path):
    txt = open(path, errors="replace").read()
    for block in re.split(r"remark: Function Name: ", txt)[1:]:
        name = block.split()[0]
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", block)
        v = re.search(r" VGPRs: (\d+)", block)
        a = re.search(r"AGPRs: (\d+)", block)
        o = re.search(r"Occupancy \[waves/SIMD\]: (\d+)", block)
        if m:
            yield name, int(m.group(1)), int(v.group(1)) if v else -1, int(a.group(1)) if a else -1, int(o.group(1)) if o else -1


def main(argv):
    verbose = "-v" in argv
    files = [a for a in argv if not a.startswith("-")]
    rows = []
    for f in files:
        for k in kernels(f):
            rows.append((f,) + k)
    dem = demangle([r[1] for r in rows])
    bad, allowed, seen = [], [], set()
    per_file = {f: 0 for f in files}
    for (f, name, scratch, vgpr, agpr, occ), d in zip(rows, dem):
        gated = any(g in d for g in GATED) or any(t in name for t in ("4pgps2rc", "4pgps3rc2", "4pgps2qc", "4pgps15k_pkfs_resident"))
        if not gated or (name, f) in seen:
            continue
        seen.add((name, f))
        per_file[f] += 1
        if verbose:
            print(f"{scratch:6d} B  vgpr {vgpr:3d} agpr {agpr:3d} occ {occ}  {d[:140]}")
        if scratch > 0:
            reason = next((why for pat, why in ALLOW if re.search(pat, d)), None)
            (allowed if reason else bad).append((scratch, vgpr, agpr, d, f, reason))
    for scratch, vgpr, agpr, d, f, reason in allowed:
        print(f"scratch_gate: allowed {scratch} B/lane in {d[:150]}  [{reason}]")
    for scratch, vgpr, agpr, d, f, _ in bad:
        print(f"scratch_gate: {scratch} B/lane of scratch (vgpr {vgpr}, agpr {agpr}) in {d[:200]}   ({f})", file=sys.stderr)
    print(f"scratch_gate: {len(seen)} cooperative kernels checked, {len(bad)} with unexplained scratch, {len(allowed)} allow-listed")
    # a gate that saw nothing enforces nothing: a changed remark format, an empty .res file or a unit whose remarks are
    # missing must fail the build, not pass it
    empty = [f for f, n in per_file.items() if n == 0 and not f.endswith(EXPECT_NONE)]
    if not files or not seen or empty:
        for f in empty:
            print(f"scratch_gate: no gated kernel found in {f} (remark format changed? unit not compiled with -Rpass-analysis?)", file=sys.stderr)
        if not files or not seen:
            print("scratch_gate: nothing was checked", file=sys.stderr)
        return 1
    if verbose:
        for f, n in sorted(per_file.items()):
            print(f"scratch_gate: {n:4d} kernels in {f}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
