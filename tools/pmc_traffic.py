#!/usr/bin/env python3
"""HBM-side traffic per launch from rocprofv3 PMC passes -> profiles/rNN_traffic.json entries.

  tools/pmc_traffic.py <pmc dir of tools/pmc.sh> <workload key> [<existing json to update>] [<fetch factor>]

The PMC passes are separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md, rocprofv3 PMC
slots); traffic = factor * FETCH_SIZE + WRITE_SIZE in bytes (counters are KiB).  factor = 2 for kernels whose reads are
wide coalesced 16-byte accesses: on gfx950 FETCH_SIZE reports half their bytes (same guide, HBM section; confirmed here
on k_filter_reduce / rc_reduce1, whose reads are exactly Fs, Qs, ys: 0.55-0.57 of the known volume).  Other access
widths are uncalibrated, the guide says, so the lane-chunk kernels at d >= 4 (direct 64 / 100 / 144-byte record
accesses) are calibrated the same way: their k_filter_reduce reads exactly (2 d^2 + 1) w bytes per step and reports
0.82 of that -- factor 1.22.  Kernels are mapped to bench.py's launch slots; a slot that holds
several launches per pass (the Kogge-Stone levels of the row-cooperative family sit in the reduce slots) gets the SUM
over its kernels per pass."""
import collections
import csv
import glob
import json
import sys

SLOT_OF = [("k_pkfs_resident", "k_pkfs_resident"), ("k_filter_reduce", "k_filter_reduce"), ("k_filter_apply", "k_filter_apply"), ("k_filter_single", "k_filter_apply"),
           ("k_smoother_apply", "k_smoother_apply"), ("k_smoother_reduce", "k_smoother_reduce"),
           ("rc_reduce1", "k_filter_reduce"), ("rc_ks_filter", "k_filter_reduce"), ("rc_apply1", "k_filter_apply"),
           ("rc_ks_smoother", "k_smoother_reduce"), ("rc_selem1", "k_smoother_reduce"), ("rc_smooth1", "k_smoother_apply"),
           ("rc_scan_blk_f", "k_filter_reduce"), ("rc_scan_blk_s", "k_smoother_reduce"), ("q_reduce1", "k_filter_reduce"),
           ("q_apply1", "k_filter_apply"), ("q_smooth1", "k_smoother_apply")]


def main():
    root, key = sys.argv[1], sys.argv[2]
    out_path = sys.argv[3] if len(sys.argv) > 3 else None
    factor = float(sys.argv[4]) if len(sys.argv) > 4 else 2.0
    # optional slot renames "from=to,..." (the fused road's resident launch is filed as the bench line's fused_path leg)
    rename = dict(kv.split("=") for kv in sys.argv[5].split(",")) if len(sys.argv) > 5 else {}
    per = collections.defaultdict(lambda: collections.defaultdict(list))    # kernel -> counter -> values per dispatch
    for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    slots = collections.defaultdict(lambda: {"FETCH_SIZE_KiB": 0.0, "WRITE_SIZE_KiB": 0.0, "kernels": []})
    # passes PER COUNTER RUN: FETCH_SIZE and WRITE_SIZE come from separate runs of the bench, and since its re-warm is time
    # based the two runs do not make the same number of passes
    passes = {"FETCH_SIZE": 0, "WRITE_SIZE": 0}
    for kname, ctr in per.items():
        slot = next((s for pat, s in SLOT_OF if pat in kname), None)
        slot = rename.get(slot, slot)
        if slot is None or "FETCH_SIZE" not in ctr or "WRITE_SIZE" not in ctr:
            continue
        short = kname.split("(")[0].replace("void pgps::", "")
        n = len(ctr["FETCH_SIZE"])
        slots[slot]["kernels"].append({"name": short, "dispatches": n, "FETCH_SIZE_KiB_mean": sum(ctr["FETCH_SIZE"]) / n,
                                       "WRITE_SIZE_KiB_mean": sum(ctr["WRITE_SIZE"]) / len(ctr["WRITE_SIZE"])})
        # single-launch kernels fix the number of passes that were profiled
        # (the most frequent one: warm-up / fused-path variants of the same kernel run fewer times)
        if any(p in kname for p in ("rc_apply1", "k_filter_apply", "k_filter_single", "k_pkfs_resident")):
            for c in passes:
                passes[c] = max(passes[c], len(ctr[c]))
        slots[slot]["FETCH_SIZE_KiB"] += sum(ctr["FETCH_SIZE"])
        slots[slot]["WRITE_SIZE_KiB"] += sum(ctr["WRITE_SIZE"])
    assert passes["FETCH_SIZE"] and passes["WRITE_SIZE"], "no filter-apply dispatches found"
    entry = {}
    for slot, v in slots.items():
        f, w = v["FETCH_SIZE_KiB"] / passes["FETCH_SIZE"], v["WRITE_SIZE_KiB"] / passes["WRITE_SIZE"]
        entry[slot] = {"FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1), "fetch_factor": factor,
                       "traffic_bytes": int((factor * f + w) * 1024), "kernels": v["kernels"]}
    doc = {}
    if out_path:
        try:
            doc = json.load(open(out_path))
        except Exception:
            doc = {}
    doc.setdefault("_comment", "HBM-side bytes per pass and launch slot from rocprofv3 PMC (separate --pmc passes, tools/pmc.sh + "
                               "tools/pmc_traffic.py): traffic = fetch_factor*FETCH_SIZE + WRITE_SIZE (KiB -> bytes); FETCH_SIZE "
                               "reports half the bytes of wide coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section) -- "
                               "factor 2; kernels with direct record accesses are calibrated on their reduce kernel's known read "
                               "volume (see tools/pmc_traffic.py)")
    # what the figures were measured on: bench.py reports traffic_stale = true once the kernels have changed
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        from bench import kernel_source_sha
        entry["_kernel_source_sha"] = kernel_source_sha()
    except Exception:                   # noqa: BLE001
        pass
    if rename and key in doc:           # a renamed slot is ADDED to the workload's entry (measured in a run of its own)
        merged = dict(doc[key])
        merged.update({k: v for k, v in entry.items() if k in rename.values()})
        entry = merged
    doc[key] = entry
    text = json.dumps(doc, indent=1)
    if out_path:
        open(out_path, "w").write(text + "\n")
    print(json.dumps({key: {k: v["traffic_bytes"] for k, v in entry.items() if isinstance(v, dict)}}))


if __name__ == "__main__":
    main()
