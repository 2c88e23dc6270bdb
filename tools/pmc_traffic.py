"""profiles/<round>_pmc.json from the rocprofv3 --pmc passes of tools/pmc_run.sh:  python tools/pmc_traffic.py [--workload W] <round> <tag> [<tag> ...]
Per-launch medians for k_r2 / k_xa (k_r2: the merged launch); HBM bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (gfx950 reports half of a wide coalesced
read in FETCH_SIZE -- MI355X_MICROARCH.md, HBM / rocprofv3 section; WRITE_SIZE as is).  `_meta` records the commit and the
hash of the kernel sources the passes ran on: bench.py replays these counters in its roofline objects and marks them stale
when the sources have changed since."""
import json, os, subprocess, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench

workload = "qm9_CC"
if "--workload" in sys.argv:
    i = sys.argv.index("--workload")
    workload = sys.argv[i + 1]
    del sys.argv[i:i + 2]
rnd, tags = sys.argv[1], sys.argv[2:]
out = {}
for tag in tags:
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_summary.py"), os.path.join(root, "gpurun_out", f"pmc_{tag}")],
                       capture_output=True, text=True, check=True)
    for k, cs in json.loads(r.stdout).items():
        out.setdefault(k, {}).update(cs)
for k, cs in out.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        cs["FETCH_SIZE_KB"] = cs.pop("FETCH_SIZE")
        cs["WRITE_SIZE_KB"] = cs.pop("WRITE_SIZE")
        cs["hbm_bytes_per_launch"] = int(2 * cs["FETCH_SIZE_KB"] * 1024 + cs["WRITE_SIZE_KB"] * 1024)
try:
    commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=root, text=True).strip()
except Exception:
    commit = os.environ.get("CCSD_COMMIT", "unknown")
out["_meta"] = {"workload": workload, "batch": bench.WORKLOADS[workload]["batch"], "commit": commit, "kernel_src_sha16": bench.kernel_source_hash(),
                "note": f"medians over the launches of a 10-step bench run (bench.py --workload {workload} --steps 10 --warmup 2); hbm_bytes = 2*FETCH_SIZE*1024 + "
                        "WRITE_SIZE*1024 (gfx950 FETCH_SIZE half-count correction, MI355X_MICROARCH.md); separate --pmc passes"}
json.dump(out, open(os.path.join(root, "profiles", f"{rnd}_pmc.json" if workload == "qm9_CC" else f"{rnd}_pmc_{workload}.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
