#!/usr/bin/env python3
"""Scan gfx950 assembly (hipcc -S --cuda-device-only) of the cooperative kernels for the hazard behind DESIGN.md section 4k:

    a vector register is (re)written under a PARTIAL EXEC mask by a spill reload (scratch_load ... Folded Reload, or
    v_accvgpr_read of a register parked in an AGPR) -- so the lanes that are switched off keep whatever the physical
    register held before -- and is later READ ACROSS LANES (DPP row_newbcast / row_shr / quad_perm, v_permlane*, v_readlane,
    v_readfirstlane, ds_bpermute, ds_swizzle) with no full-EXEC write of it in between.

The compiler's liveness is per lane: for a lane that was off during the reload the value is "not live", and a plain
per-lane use after the region would be preceded by a reload of its own.  A cross-lane read is invisible to that
reasoning: lane j consumes lane k's copy, and lane k may have been off when the register was refilled.

Linear scan per kernel (EXEC regions: s_and_saveexec_b64 ... s_or_b64 exec, exec, ...); loops are walked once, which is
enough to see a reload inside a region followed by a cross-lane read further down the same iteration.
Usage: exec_hazard_scan.py file.s [kernel-name-substring]"""
import re
import sys

XLANE = re.compile(r"(_dpp\b|row_newbcast|row_shr|row_shl|row_bcast|quad_perm|v_permlane|v_readlane|v_readfirstlane|ds_bpermute|ds_swizzle|v_mov_b32_dpp)")
REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs(tok):
    out = []
    for m in REG.finditer(tok):
        if m.group(1):
            out.append((m.group(1), int(m.group(2))))
        else:
            out.extend((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def scan(name, body):
    partial = 0                     # depth of saveexec regions
    tainted = {}                    # reg -> (line of the partial reload, text)
    findings = []
    for ln, raw in body:
        t = raw.split(";")[0].strip()
        if not t or t.endswith(":") or t.startswith("."):
            continue
        op = t.split()[0]
        if op.startswith("s_and_saveexec") or op.startswith("s_andn2_saveexec") or op.startswith("s_or_saveexec"):
            partial += 1
            continue
        if (op in ("s_or_b64", "s_mov_b64", "s_xor_b64", "s_andn2_b64", "s_and_b64")) and re.match(r"\S+\s+exec\b", t):
            if op == "s_or_b64" or op == "s_mov_b64":
                partial = max(0, partial - 1)
            else:
                partial += 0
            continue
        args = t[len(op):]
        parts = [p.strip() for p in args.split(",")]
        dst = regs(parts[0]) if parts else []
        srcs = [r for p in parts[1:] for r in regs(p)]
        is_store = op.startswith(("scratch_store", "global_store", "buffer_store", "ds_write", "ds_store", "flat_store"))
        if is_store:
            srcs = [r for p in parts for r in regs(p)]
            dst = []
        # cross-lane read of a tainted register?
        if XLANE.search(t):
            # for DPP VALU ops the DPP source is src0; for v_fmac the accumulator (dst) is also read -- per lane, harmless
            x_srcs = srcs if not op.startswith("v_readlane") else srcs
            for r in x_srcs:
                if r in tainted:
                    findings.append((ln, t, r, tainted[r]))
        reload = ("Folded Reload" in raw and op.startswith("scratch_load")) or op.startswith("v_accvgpr_read")
        for r in dst:
            if partial > 0 and reload:
                tainted[r] = (ln, t)
            elif partial > 0:
                # an ordinary write under partial EXEC keeps the taint of the lanes that are off
                pass
            else:
                tainted.pop(r, None)
    return findings


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    lines = open(path, errors="replace").read().split("\n")
    kernels, cur, name = [], None, None
    for i, l in enumerate(lines, 1):
        m = re.match(r"^(_Z\w+):\s", l)
        if m:
            name, cur = m.group(1), []
            kernels.append((name, cur))
        elif cur is not None:
            if l.startswith("\t.end_amdhsa_kernel") or l.strip().startswith("s_endpgm"):
                cur.append((i, l))
                cur = None
            else:
                cur.append((i, l))
    total = 0
    for name, body in kernels:
        if want and want not in name:
            continue
        f = scan(name, body)
        total += len(f)
        print(f"{name}: {len(f)} cross-lane reads of registers refilled under a partial EXEC mask")
        for ln, t, r, (pl, pt) in f[:12]:
            print(f"    line {ln}: {t}\n        {r[0]}{r[1]} refilled at line {pl} under partial EXEC: {pt}")
    return 0 if total == 0 else 2


if __name__ == "__main__":
    sys.exit(main())
