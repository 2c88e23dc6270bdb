"""Per-kernel MEDIANS over the dispatches of rocprofv3 --pmc counters: python tools/pmc_summary.py <dir with *_counter_collection.csv> [...]
(median, not mean: a sampler call launches k_r2 as norms-only, merged ... merged, predictor-only -- the median is the merged launch the
bench line's launch time belongs to; for k_xa the norms and predictor launches differ by a few percent either way)"""
import csv, glob, sys, collections, json, os

out = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            k = "k_xa" if "k_xa" in name else "k_r2" if "k_r2" in name else None
            if k is None:
                continue
            per[(k, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (k, _, c), v in per.items():
            out[k][c].append(v)
res = {k: {c: sorted(v)[len(v) // 2] for c, v in cs.items()} for k, cs in out.items()}
print(json.dumps(res, indent=1))
