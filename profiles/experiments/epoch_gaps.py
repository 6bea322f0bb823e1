"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV, grouped by (kernel before, kernel after).
Usage: python profiles/experiments/epoch_gaps.py <kernel_trace.csv> [epochs]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
by = collections.defaultdict(list)
busy = sum(e - s for s, e, _ in ev)
for (s0, e0, n0), (s1, e1, n1) in zip(ev, ev[1:]):
    if 0 < s1 - e0 < 2_000_000: by[(short(n0), short(n1))].append(s1 - e0)
tot = sum(sum(v) for v in by.values())
print(f"{len(ev)} kernels, busy {busy / 1e6:.2f} ms, idle between kernels {tot / 1e6:.3f} ms = {tot / 1e3 / epochs:.0f} us per epoch ({epochs} epochs)")
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(f"  {k[0]:44s} -> {k[1]:44s} x{len(v):4d}  median {sorted(v)[len(v) // 2] / 1e3:7.1f} us  avg {sum(v) / len(v) / 1e3:7.1f} us  total {sum(v) / 1e3:8.1f} us")
