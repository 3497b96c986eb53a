"""Summarise rocprofv3 runs of bench.py into the JSON files committed under profiles/.

  python tools/pmc_summary.py traffic <fetch_dir> <write_dir> <forwards> <out.json>
      per-kernel HBM bytes from two separate --pmc passes (FETCH_SIZE, WRITE_SIZE).  Counter unit: KiB; FETCH_SIZE is
      doubled (gfx950 counts the 128-B requests of 16 B/lane coalesced reads as 64 B: MI355X_MICROARCH.md, HBM section),
      WRITE_SIZE is exact.
  python tools/pmc_summary.py stats <stats_dir> <forwards> <out.csv>
      per-kernel launches / total / average duration from a --kernel-trace --stats run.
"""
import csv, glob, json, os, re, sys
from collections import defaultdict


def short(name: str) -> str:
    """'void (anonymous namespace)::conv3x3_f16x3_row_kernel<7>(atmvfi::Conv3Dev)' -> 'conv3x3_f16x3_row_kernel' (template
    instances of one kernel are summed: they are the same code at different tile widths)."""
    n = name.strip().strip('"')
    n = re.sub(r"^void\s+", "", n)
    n = n.replace("(anonymous namespace)::", "")
    n = re.split(r"[<(]", n, 1)[0]
    return n.split("::")[-1].strip()


def csrc_digest():
    """The build these counters were taken on: bench.py's own digest (kernel sources, C-ABI header, Makefile), so the two cannot drift."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    return bench.csrc_digest()


def counter_sum(d, counter):
    tot, n = defaultdict(float), defaultdict(int)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            tot[k] += float(r["Counter_Value"])
            n[k] += 1
    return tot, n


def traffic(fetch_dir, write_dir, forwards, out):
    fe, nf = counter_sum(fetch_dir, "FETCH_SIZE")
    wr, _ = counter_sum(write_dir, "WRITE_SIZE")
    per = {}
    for k in sorted(fe, key=lambda k: -(2 * fe[k] + wr.get(k, 0.0))):
        launches = nf[k] / forwards
        rd = 2.0 * fe[k] * 1024 / forwards / 1e9
        w = wr.get(k, 0.0) * 1024 / forwards / 1e9
        per[k] = {"launches": round(launches, 1), "read_GB_corrected": round(rd, 3), "write_GB": round(w, 3),
                  "traffic_GB_per_launch": round((rd + w) / launches, 4)}
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --steps 1 --warmup 1`: "
                       f"{forwards} forwards of network_base 1088x1920. Counter unit KiB. FETCH_SIZE is doubled (gfx950 tallies the 128-B "
                       "requests of 16 B/lane coalesced reads at 64 B: MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.",
               "csrc_sha256": csrc_digest(), "per_forward": per}, open(out, "w"), indent=1)
    print(open(out).read()[:3000])


def stats(d, forwards, out):
    files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no kernel_stats.csv under {d}")
    agg = {}
    for r in csv.DictReader(open(files[0])):
        a = agg.setdefault(short(r["Name"]), {"calls": 0, "ns": 0.0, "min": 1e30, "max": 0.0})
        a["calls"] += int(r["Calls"])
        a["ns"] += float(r["TotalDurationNs"])
        a["min"] = min(a["min"], float(r["MinNs"]))
        a["max"] = max(a["max"], float(r["MaxNs"]))
    tot = sum(a["ns"] for a in agg.values())
    with open(out, "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel (template instances summed)", "calls", "calls_per_forward", "total_ms", "ms_per_forward", "avg_us", "min_us", "max_us", "percent"])
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ns"]):
            w.writerow([k, a["calls"], round(a["calls"] / forwards, 1), round(a["ns"] / 1e6, 3), round(a["ns"] / 1e6 / forwards, 3),
                        round(a["ns"] / a["calls"] / 1e3, 2), round(a["min"] / 1e3, 2), round(a["max"] / 1e3, 2), round(100 * a["ns"] / tot, 2)])
    print(open(out).read()[:2500])


if __name__ == "__main__":
    if sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5])
    else:
        stats(sys.argv[2], int(sys.argv[3]), sys.argv[4])
