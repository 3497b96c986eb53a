"""Sum LDS counters per kernel from rocprofv3 --pmc passes: python tools/pmc_lds.py <dir> [<dir> ...]"""
import csv, glob, os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import short
tot = defaultdict(lambda: defaultdict(float))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            tot[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
names = sorted({c for v in tot.values() for c in v})
for k, v in sorted(tot.items(), key=lambda kv: -sum(kv[1].values()))[:int(os.environ.get('PMC_TOP', '24'))]:
    print(k)
    for n in names:
        print(f"    {n:32s} {v.get(n, 0):14.5g}")
