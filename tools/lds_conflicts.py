"""LDS bank-conflict share of every kernel of a step, from a rocprofv3 counter pass:
   rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d <dir> -o p -- python bench.py --steps 6 --warmup 2 --repeats 1 --quick
   python tools/lds_conflicts.py <dir>
conflict = extra LDS cycles, active = all LDS-array cycles (MI355X_MICROARCH.md, LDS section)."""
import collections, csv, glob, sys

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    c, a = d.get("SQ_LDS_BANK_CONFLICT", [0.0]), d.get("SQ_LDS_IDX_ACTIVE", [1.0])
    ca, aa = sum(c) / len(c), sum(a) / len(a)
    if aa > 1000:
        print("%-62s conflict %12.0f  active %12.0f  share %.3f  launches %d" % (k, ca, aa, ca / max(aa, 1.0), len(a)))
