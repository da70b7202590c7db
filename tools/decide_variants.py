#!/usr/bin/env python3
"""After `bash tools/gpurun_r6.sh variants` (results under profiles/r06_variants/): the verdict on every variant library by the rule of
DESIGN.md section 4 -- adopt a switch only if the parity core of the GPU suite is green through its library AND its bench lines are
not slower than the shipped library's on every workload measured (tolerance 1 %); otherwise delete its code.

    python3 tools/decide_variants.py [directory]          (default profiles/r06_variants)
Reads <dir>/pytest_<tag>.log (the last line of the parity core run through libempc_<tag>.so) and <dir>/bench_<config>_<tag>.json
(+ bench_<config>_shipped.json).  Prints one row per variant: suite, speed ratio per workload, kernel times, decision."""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def line(path):
    try:
        return json.loads(open(path).read().strip().splitlines()[-1])
    except Exception:
        return None


def main():
    d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r06_variants")
    shipped = {}
    for f in glob.glob(os.path.join(d, "bench_*_shipped.json")):
        cfg = os.path.basename(f)[len("bench_"):-len("_shipped.json")]
        shipped[cfg] = line(f)
    tags = sorted(set(re.match(r"pytest_(.+)\.log", os.path.basename(f)).group(1) for f in glob.glob(os.path.join(d, "pytest_*.log"))))
    if not tags:
        print("no variant results under", d)
        return 1
    print("%-10s %-22s %s" % ("variant", "parity core", "value / shipped per workload (backward, rollout, linearize ms per launch)   -> decision"))
    for tag in tags:
        tail = open(os.path.join(d, "pytest_%s.log" % tag)).read().strip().splitlines()
        last = tail[-1] if tail else ""
        green = bool(re.search(r"\d+ passed", last)) and not re.search(r"failed|error", last)
        ratios, cells = [], []
        for cfg, base in sorted(shipped.items()):
            v = line(os.path.join(d, "bench_%s_%s.json" % (cfg, tag)))
            if not v or not base:
                cells.append("%s: no line" % cfg)
                continue
            r = v["value"] / base["value"]
            ratios.append(r)
            k = v.get("kernels", {})
            ms = " ".join("%.3f" % k[n]["avg_ms"] for n in ("backward", "rollout", "linearize") if n in k)
            cells.append("%s %.3f (%s)" % (cfg, r, ms))
        fast = bool(ratios) and min(ratios) >= 0.99
        decision = "ADOPT" if (green and fast) else ("delete: suite not green" if not green else "delete: slower (min ratio %.3f)" % (min(ratios) if ratios else 0.0))
        print("%-10s %-22s %s   -> %s" % (tag, last[:22], "; ".join(cells), decision))
    return 0


if __name__ == "__main__":
    sys.exit(main())
