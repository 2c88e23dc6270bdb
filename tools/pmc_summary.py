"""Per-kernel MEDIANS over the dispatches of rocprofv3 --pmc counters: python tools/pmc_summary.py <dir with *_counter_collection.csv> [...]
(median, not mean: a sampler call launches k_r2 as norms-only, merged ... merged, predictor-only -- the median is the merged launch the
bench line's launch time belongs to; for k_xa the norms and predictor launches differ by a few percent either way)"""
import csv, glob, sys, collections, json, os

KNAMES = ("k_xa", "k_r2", "k_hf_score", "k_hp_full", "k_gemm_h", "k_gemm_p", "k_langevin_apply", "k_noise_norm", "k_s4_apply", "k_ew1")
out = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            k = next((n for n in KNAMES if n in name), None)     # (k_gemm_p0 counts as k_gemm_p: the same GEMM, narrow instance)
            if k is None:
                continue
            per[(k, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (k, _, c), v in per.items():
            out[k][c].append(v)
res = {k: {c: sorted(v)[len(v) // 2] for c, v in cs.items()} for k, cs in out.items()}
print(json.dumps(res, indent=1))
