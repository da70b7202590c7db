#!/usr/bin/env python3
"""Condense rocprofv3 output directories into the small files kept under profiles/.

  python tools/profile_summarize.py stats   <rocprof dir> <out.csv>          # --kernel-trace --stats run
  python tools/profile_summarize.py traffic <fetch dir> <write dir> <tag> [config]  # two --pmc passes (FETCH_SIZE / WRITE_SIZE)
  python tools/profile_summarize.py pmc <config> <out.json> <B> <T> <NA> <pass dir> ...  # round 4: FULL-BATCH launches only

`traffic` writes profiles/traffic_<config>_<kernel>.json per hot kernel: HBM bytes per launch as /opt/skills/guides/
MI355X_MICROARCH.md prescribes -- FETCH_SIZE and WRITE_SIZE are reported in KiB-sized units by rocprofv3 (x1024) and, on
gfx950, FETCH_SIZE tallies 128-B requests at 64 B, so reads are doubled; the raw values are kept next to the corrected one.
"""
import collections
import csv
import glob
import json
import os
import sys

KERNELS = ("k_linearize_all", "k_linearize_full", "k_linearize", "k_backward", "k_rollout", "k_select", "k_calc", "k_squash_out", "k_plant_rk4", "k_pack_rows")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    if "k_linearize_all" in name:  # both bodies in one launch (sweeps with less than half of the batch active)
        return "k_linearize_all"
    if "k_linearize" in name:  # two bodies per sweep: lean (no operational frames) and full
        return "k_linearize_full" if "true>(" in name.replace(" ", "") else "k_linearize"
    for k in KERNELS:
        if k in name:
            return k
    return name[:60]


def stats(src, out):
    files = glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        raise SystemExit("no *kernel_stats.csv under " + src)
    rows = list(csv.DictReader(open(max(files, key=os.path.getmtime))))  # newest run in the directory
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms", "percent", "full_name"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], "%.4f" % (float(r["TotalDurationNs"]) / 1e6),
                        "%.4f" % (float(r["AverageNs"]) / 1e6), "%.4f" % (float(r["MinNs"]) / 1e6),
                        "%.4f" % (float(r["MaxNs"]) / 1e6), r["Percentage"], r["Name"][:160]])
    print(open(out).read())


def counters(src, counter):
    files = glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no *counter_collection.csv under " + src)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
        if r["Counter_Name"] == counter:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg


def traffic(fetch_dir, write_dir, tag, config="displacement"):
    fe = counters(fetch_dir, "FETCH_SIZE")
    wr = counters(write_dir, "WRITE_SIZE")
    # one linearize step of a sweep = both bodies (two launches, or one k_linearize_all launch): total bytes of all
    # linearize launches spread over the sweeps of the run (= backward launches)
    if "k_backward" in fe and "k_backward" in wr:
        ns = len(fe["k_backward"])
        for tab in (fe, wr):
            tot = sum(tab.get("k_linearize", [])) + sum(tab.get("k_linearize_full", [])) + sum(tab.get("k_linearize_all", []))
            tab["k_linearize"] = [tot / ns] * ns
    for k in ("k_linearize", "k_backward", "k_rollout"):
        if k not in fe or k not in wr:
            continue
        # the solve loop launches over a shrinking active set; the first launches (whole batch active) are the ones the
        # bench's per-launch algorithmic bytes describe on average, so report the mean over all launches of the run
        f_raw = sum(fe[k]) / len(fe[k]) * 1024.0
        w_raw = sum(wr[k]) / len(wr[k]) * 1024.0
        out = {"kernel": k, "tag": tag, "config": config, "launches": len(fe[k]),
               "fetch_bytes_raw_per_launch": f_raw, "write_bytes_raw_per_launch": w_raw,
               "hbm_bytes_per_launch": 2.0 * f_raw + w_raw,
               "correction": "FETCH_SIZE x1024 x2 (gfx950 tallies 128-B read requests at 64 B), WRITE_SIZE x1024; 8-B/lane "
                             "accesses are outside the guide's calibrated 16-B/lane pattern, so the absolute value is "
                             "indicative, ratios between builds are exact"}
        name = k.replace("k_", "")
        with open(os.path.join(ROOT, "profiles", "traffic_%s_%s.json" % (config, name)), "w") as f:
            json.dump(out, f, indent=1)
        print(json.dumps(out))


def sq(dirs, out):
    """Per-kernel means of every counter found in the given --pmc passes (one row per kernel)."""
    table = collections.defaultdict(dict)
    for d in dirs:
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            raise SystemExit("no *counter_collection.csv under " + d)
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        meta = {}
        for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size"], r["Grid_Size"])
        for k in agg:
            if k not in KERNELS:
                continue
            table[k].update({c: sum(v) / len(v) for c, v in agg[k].items()})
            table[k].update(dict(zip(("VGPR", "AGPR", "SGPR", "LDS", "WG", "GRID"), meta[k])))
    cols = sorted({c for k in table for c in table[k]})
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel"] + cols)
        for k in table:
            w.writerow([k] + [("%.6g" % table[k][c]) if isinstance(table[k].get(c), float) else table[k].get(c, "") for c in cols])
    for k in table:
        t = table[k]
        print(k, {c: (("%.4g" % t[c]) if isinstance(t[c], float) else t[c]) for c in cols if c in t})


def pmc(config, out, B, T, NA, dirs):
    """profiles/r04_pmc_<config>.json: per hot kernel, means over the launches whose grid is the FULL batch (a stream run keeps
    the slots full until its queue is dry; the drain sweeps at the end have smaller grids and are left out):
      hbm_bytes_per_launch   2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024 (MI355X_MICROARCH.md, HBM section: gfx950 tallies 128-B
                             read requests at 64 B; 8-B-per-lane accesses are outside the guide's calibrated pattern)
      fp64_flop_per_launch   64 lanes x (ADD_F64 + MUL_F64 + TRANS_F64 + 2 FMA_F64) + 512 x MFMA_MOPS_F64: flops ISSUED, idle
                             lanes included -- what the FP64 pipes were asked to do
      valu_active_frac       SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (both in quad-cycles): share of a wavefront's life with a
                             vector instruction in execution; wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES
    linearize = lean launch + full-body launch of one sweep (whichever exist)."""
    rows = collections.defaultdict(lambda: collections.defaultdict(list))  # kernel -> counter -> values of full-batch launches
    grids = collections.defaultdict(int)
    for d in dirs:
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            print("no counter_collection.csv under", d)
            continue
        recs = []
        for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
            k = short(r["Kernel_Name"])
            if k in ("k_linearize", "k_linearize_full", "k_linearize_all", "k_backward", "k_rollout"):
                recs.append((int(r["Dispatch_Id"]), k, int(r["Grid_Size"]), r["Counter_Name"], float(r["Counter_Value"])))
                if k != "k_linearize_all":
                    grids[k] = max(grids[k], int(r["Grid_Size"]))
        # A sweep is linearize (one or two launches) -> backward -> rollout, in dispatch order.  The linearize grid follows the
        # number of live slots (the backward / rollout grids do not: their surplus workgroups return at once), so a sweep counts
        # as FULL when its linearize launch(es) have the largest grid seen; backward and rollout inherit the flag of the
        # linearize launch that precedes them.
        recs.sort()
        full = False
        for disp, k, g, c, v in recs:
            if k.startswith("k_linearize"):
                full = (k != "k_linearize_all") and g == grids[k]
            if full and k != "k_linearize_all":
                rows[k][c].append(v)
    mean = lambda k, c: (sum(rows[k][c]) / len(rows[k][c])) if rows[k].get(c) else None
    # (rollout: units are (trajectory, knot) from round 5 on -- one launch tries NA step lengths on each; bench.py counts the same way)
    units = {"linearize": B * (T + 1), "backward": B * T, "rollout": B * (T + 1)}
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import device_code_id as dci
    res = {"config": config, "B": B, "T": T, "NA": NA, "kernels": {},
           # which kernels these counters describe: bench.py uses the file only when the id equals that of the library it loaded
           "device_code_id": dci.device_code_id(), "commit": os.environ.get("EMPC_COMMIT"), "library": os.environ.get("EMPC_LIB_PATH") or "eagle-mpc_amd/libempc.so",
           "method": "rocprofv3 --kernel-trace --pmc <one counter group per pass> over `bench.py --mode stream`; launches with the full grid only"}
    groups = {"linearize": [k for k in ("k_linearize", "k_linearize_full") if k in rows], "backward": ["k_backward"], "rollout": ["k_rollout"]}
    for name, ks in groups.items():
        ks = [k for k in ks if k in rows]
        if not ks:
            continue
        tot = lambda c: (sum(mean(k, c) for k in ks) if all(mean(k, c) is not None for k in ks) else None)
        e = {"launch_kinds": ks, "grid_sizes": {k: grids[k] for k in ks}, "units_per_launch": units[name],
             "full_launches_seen": {k: max([len(v) for v in rows[k].values()] or [0]) for k in ks}}
        f, w = tot("FETCH_SIZE"), tot("WRITE_SIZE")
        if f is not None and w is not None:
            e.update(fetch_raw_bytes=f * 1024.0, write_raw_bytes=w * 1024.0, hbm_bytes_per_launch=2.0 * f * 1024.0 + w * 1024.0)
        a, m_, t_, fm, mo = (tot("SQ_INSTS_VALU_ADD_F64"), tot("SQ_INSTS_VALU_MUL_F64"), tot("SQ_INSTS_VALU_TRANS_F64"),
                             tot("SQ_INSTS_VALU_FMA_F64"), tot("SQ_INSTS_VALU_MFMA_MOPS_F64"))
        if None not in (a, m_, t_, fm):
            e["fp64_insts_per_launch"] = {"add": a, "mul": m_, "trans": t_, "fma": fm, "mfma_mops": mo, "valu_all": tot("SQ_INSTS_VALU"),
                                          "salu": tot("SQ_INSTS_SALU"), "waves": tot("SQ_WAVES")}
            e["fp64_flop_per_launch"] = 64.0 * (a + m_ + t_ + 2.0 * fm) + 512.0 * (mo or 0.0)
        wc = tot("SQ_WAVE_CYCLES")
        if wc:
            e["valu_active_frac"] = (tot("SQ_ACTIVE_INST_VALU") or 0.0) / wc
            e["wait_frac"] = (tot("SQ_WAIT_ANY") or 0.0) / wc
            e["issue_stall_frac"] = (tot("SQ_WAIT_INST_ANY") or 0.0) / wc
            e["sq_busy_cycles"] = tot("SQ_BUSY_CYCLES")
            e["wave_quad_cycles"] = wc
            e["lds_insts"] = tot("SQ_INSTS_LDS")
            e["lds_bank_conflict_cycles"] = tot("SQ_LDS_BANK_CONFLICT")
        res["kernels"][name] = e
    with open(out, "w") as fjs:
        json.dump(res, fjs, indent=1)
    print(json.dumps(res)[:3000])


if __name__ == "__main__":
    if len(sys.argv) >= 8 and sys.argv[1] == "pmc":
        pmc(sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7:])
    elif len(sys.argv) >= 4 and sys.argv[1] == "sq":
        sq(sys.argv[3:], sys.argv[2])
    elif len(sys.argv) >= 4 and sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif len(sys.argv) >= 5 and sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3], sys.argv[4], *(sys.argv[5:6]))
    else:
        raise SystemExit(__doc__)
