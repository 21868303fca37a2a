#!/usr/bin/env python3
"""Symbolic traces over gfx950 assembly (hipcc -S --cuda-device-only) of one loop of a cooperative kernel, written for the
round-3 'ghost' (profiles/r04_experiments.txt item 1):

    isa_symtrace.py sgpr file.s HEADER_LINE BACKEDGE_LINE [FIRST_LINE]
        every SGPR that the loop body uses before defining it must hold, at the back edge, the value it held at the loop
        header.  Copies (s_mov), spills to VGPR lanes (v_writelane / v_readlane) and everything else (a fresh symbol per
        defining line) are followed through the prologue (FIRST_LINE .. HEADER_LINE) and once through the body; prints the
        registers for which that fails.  (Straight-line reading: conditional regions are taken.)
    isa_symtrace.py dpp file.s HEADER_LINE BACKEDGE_LINE
        for every v_fmac_f64_dpp of the body: the symbolic values of its accumulator, its broadcast operand and its per-lane
        operand, followed through v_mov, AGPR copies, scratch spill slots, v_permlane16_swap and global loads (tagged with
        their offsets), tab-separated -- to check that every term of every product is fed by the operand it should be."""
import re
import sys


def sgpr(path, lo, hi, pre):
    lines = open(path).read().split("\n")
    rng = re.compile(r"s\[(\d+):(\d+)\]|\bs(\d+)\b")

    def sregs(tok):
        out = []
        for m in rng.finditer(tok):
            out += list(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else [int(m.group(3))]
        return out
    val = {i: f"in:s{i}" for i in range(0, 106)}
    lane, used, defined, snap = {}, {}, set(), None

    def use(r, ln):
        if r not in defined and r not in used:
            used[r] = ln
    for i in (list(range(pre, lo)) if pre else []) + list(range(lo, hi + 1)):
        if i == lo:
            snap = dict(val); used.clear(); defined.clear()
        t = lines[i - 1].split(";")[0].strip()
        if not t or t.endswith(":") or t.startswith(".") or t.startswith("v_fmac_f64_dpp"):
            continue
        parts = t.split(None, 1)
        if len(parts) < 2:
            continue
        op, args = parts
        ops = [a.strip() for a in args.split(",")]
        if op.startswith("v_writelane"):
            s = sregs(ops[1])
            if s:
                use(s[0], i); lane[(ops[0], int(ops[2]))] = val[s[0]]
            else:
                lane[(ops[0], int(ops[2]))] = f"imm@{i}"
            continue
        if op.startswith("v_readlane"):
            d = sregs(ops[0])[0]
            val[d] = lane.get((ops[1], int(ops[2])), f"in:{ops[1]}[{ops[2]}]"); defined.add(d)
            continue
        if op in ("s_mov_b64", "s_mov_b32"):
            d, s = sregs(ops[0]), sregs(ops[1])
            if ops[0] == "exec" or not d:
                for r in s:
                    use(r, i)
                continue
            if s and len(s) == len(d):
                for r in s:
                    use(r, i)
                for r, v in zip(d, [val[r] for r in s]):
                    val[r] = v; defined.add(r)
            else:
                for r in d:
                    val[r] = f"{op}({ops[1]})@{i}"; defined.add(r)
            continue
        dsts, srcs = [ops[0]], ops[1:]
        if op.startswith(("v_mad_u64_u32", "v_mad_i64_i32", "v_add_co", "v_sub_co", "v_addc_co", "v_subb_co", "v_div_scale")):
            dsts.append(ops[1]); srcs = ops[2:]
        if op.startswith(("s_cmp", "s_cbranch", "s_waitcnt", "s_nop", "s_branch", "s_barrier", "s_endpgm", "s_bitcmp")):
            for o in ops:
                for r in sregs(o):
                    use(r, i)
            continue
        if op.startswith(("global_store", "scratch_store", "ds_write", "ds_store", "global_load", "scratch_load", "ds_read", "ds_load", "s_load")):
            for o in ops[1:]:
                for r in sregs(o):
                    use(r, i)
            if op.startswith("s_load"):
                for r in sregs(ops[0]):
                    val[r] = f"{op}({','.join(ops[1:])})@{i}"; defined.add(r)
            continue
        for o in srcs:
            for r in sregs(o):
                use(r, i)
        for dst in dsts:
            for r in sregs(dst):
                val[r] = f"{op}@{i}"; defined.add(r)
    bad = 0
    for r, ln in sorted(used.items()):
        if val[r] != snap[r]:
            bad += 1
            print(f"  s{r}: first used at line {ln}; header value {snap[r]}; at the back edge {val[r]}")
    print(f"{bad} of {len(used)} live-in SGPRs differ at the back edge (a loop counter is expected to)")


def dpp(path, lo, hi):
    lines = open(path).read().split("\n")
    R = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")

    def regs(tok):
        out = []
        for m in R.finditer(tok):
            out += [(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)] if m.group(1) else [(m.group(4), int(m.group(5)))]
        return out
    val, slot = {}, {}

    def get(r):
        return val.get(r, f"in:{r[0]}{r[1]}")
    for i in range(lo, hi + 1):
        t = lines[i - 1].split(";")[0].strip()
        if not t or t.endswith(":") or t.startswith("."):
            continue
        parts = t.split(None, 1)
        if len(parts) < 2:
            continue
        op, args = parts
        ops = [a.strip() for a in args.split(",")]
        if op == "v_fmac_f64_dpp":
            d, s0, s1 = regs(ops[0]), regs(ops[1]), regs(ops[2].split()[0])
            k = int(re.search(r"row_newbcast:(\d+)", t).group(1))
            print(i, k, ops[0], get(d[0]), ops[1], get(s0[0]), ops[2].split()[0], get(s1[0]), sep="\t")
            for r in d:
                val[r] = f"acc@{i}"
            continue
        off = re.search(r"offset:(\d+)", t); off = int(off.group(1)) if off else 0
        if op.startswith("scratch_store"):
            for n, r in enumerate(regs(ops[1])):
                slot[off + 4 * n] = get(r)
            continue
        if op.startswith("scratch_load"):
            for n, r in enumerate(regs(ops[0])):
                val[r] = slot.get(off + 4 * n, f"in:slot{off + 4 * n}")
            continue
        if op.startswith(("v_accvgpr_write", "v_accvgpr_read", "v_mov_b32", "v_mov_b64", "v_accvgpr_mov")):
            d, s = regs(ops[0]), regs(ops[1])
            if s and len(s) == len(d):
                for r, v in zip(d, [get(r) for r in s]):
                    val[r] = v
            else:
                for r in d:
                    val[r] = f"const({ops[1]})@{i}"
            continue
        if op.startswith(("global_store", "ds_write", "ds_store", "s_", "v_cmp", "v_writelane", "buffer_store")):
            continue
        if op.startswith("v_permlane16_swap"):
            a, b = regs(ops[0])[0], regs(ops[1])[0]
            va, vb = get(a), get(b)
            val[a] = f"swapA@{i}({va}|{vb})"; val[b] = f"swapB@{i}({va}|{vb})"
            continue
        for r in regs(ops[0]):
            val[r] = f"gld@{i}:{ops[1]}+{off}" if op.startswith("global_load") else f"{op}@{i}"


if __name__ == "__main__":
    if sys.argv[1] == "sgpr":
        sgpr(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]) if len(sys.argv) > 5 else None)
    else:
        dpp(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
