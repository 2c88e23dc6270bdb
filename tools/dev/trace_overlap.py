"""Diagnostic: python tools/dev/trace_overlap.py <rocprofv3 out dir>  -- per kernel: launches, mean duration, and how much of its time it
shared the device with kernels of another queue (kernel-trace timestamps)."""
import csv, glob, os, sys, collections
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    k = next((x for x in ("k_zn_steps", "k_xa", "k_r2", "k_normsum", "k_flagbits", "k_masktab", "k_init_state") if x in n), n[:30])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, r.get("Queue_Id", "?")))
rows.sort()
t0 = rows[0][0]
skip = int(len(rows) * 0.5)
print("first rows after warm-up:")
for s, e, k, q in rows[skip:skip + 24]:
    print(f"  {k:14s} q{q:>3s} start {(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:7.1f} us")
dur = collections.defaultdict(list)
ov = collections.defaultdict(float)
for i, (s, e, k, q) in enumerate(rows):
    dur[k].append(e - s)
    for s2, e2, k2, q2 in rows[max(0, i - 40):i + 40]:
        if q2 != q:
            o = min(e, e2) - max(s, s2)
            if o > 0:
                ov[k] += o
for k, d in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:16s} n {len(d):6d} mean {sum(d) / len(d) / 1e3:8.1f} us  total {sum(d) / 1e6:8.2f} ms  overlapped with other queues {ov[k] / max(sum(d), 1) * 100:5.1f} %")
print(f"span {(rows[-1][1] - rows[skip][0]) / 1e3:.1f} us for the second half of the launches")
