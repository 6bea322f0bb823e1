#!/usr/bin/env python3
"""Turns rocprofv3 CSV output (gpurun_out/prof/...) into the small summaries kept
under profiles/ and into profiles/spmm_hbm_traffic.json, which bench.py reports as
roofline.traffic.

  python profiles/summarize.py <tag> <kernel_stats.csv> [--fetch f.csv] [--write w.csv] [--l2 l2.csv]

HBM-side traffic per launch follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
FETCH_SIZE / WRITE_SIZE are in KiB, collected in separate --pmc passes; on gfx950
FETCH_SIZE reports half of the bytes of a wide (16 B/lane) coalesced read, so it is
doubled for kernels whose reads are dwordx4 row gathers (the vec4 SpMM); WRITE_SIZE is
exact for 16-B-per-lane stores.  These counters sit at the L2's fabric side, so
Infinity-Cache hits are included: "traffic" = bytes that left the XCD L2s.
"""
import argparse
import collections
import csv
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from bench import spmm_kernel_sha  # noqa: E402  (content hash of csrc/spmm*.hip: stamps the counters' kernel version)


def per_kernel(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("kernel_stats")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--l2")
    ap.add_argument("--main-kernel", default="spmm_vec4_kernel<32, 8, true>")
    ap.add_argument("--launches-per-unit", type=float, default=1,
                    help="kernel launches that make up ONE mggcn_spmm_csr_f32 call (sweep form: rounds)")
    ap.add_argument("--narrow-kernel", default=None, help="second SpMM kernel (the logits-width form)")
    ap.add_argument("--narrow-launches-per-unit", type=float, default=1)
    ap.add_argument("--no-json", action="store_true", help="summary only: leave spmm_hbm_traffic.json (the headline workload's) alone")
    a = ap.parse_args()

    lines = []
    rows = list(csv.DictReader(open(a.kernel_stats)))
    lines.append(f"# rocprofv3 --kernel-trace --stats summary ({a.tag})\n")
    lines.append("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
    for r in rows[:20]:
        lines.append(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | "
                     f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
    pmc = {}
    for key, path in (("fetch", a.fetch), ("write", a.write), ("l2", a.l2)):
        if path:
            for k, ctrs in per_kernel(path).items():
                for c, v in ctrs.items():
                    pmc.setdefault(short(k), {})[c] = sum(v) / len(v)
    if pmc:
        lines.append("\n# PMC averages per launch (separate --pmc passes)\n")
        lines.append("| kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | TCC_HIT | TCC_MISS | L2 hit % |\n|---|---|---|---|---|---|")
        for k, c in sorted(pmc.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0))[:12]:
            h, m = c.get("TCC_HIT_sum"), c.get("TCC_MISS_sum")
            hit = f"{100 * h / (h + m):.1f}" if h is not None and m else ""
            lines.append(f"| {k} | {c.get('FETCH_SIZE', 0):.0f} | {c.get('WRITE_SIZE', 0):.0f} | "
                         f"{h or 0:.0f} | {m or 0:.0f} | {hit} |")
        mk = pmc.get(a.main_kernel)
        if mk and "FETCH_SIZE" in mk and "WRITE_SIZE" in mk:
            traffic = int((2 * mk["FETCH_SIZE"] + mk["WRITE_SIZE"]) * 1024 * a.launches_per_unit)
            rec = {"kernel": a.main_kernel, "bytes_per_call": traffic, "source": f"profiles/{a.tag}_summary.md",
                   "kernel_source_sha": spmm_kernel_sha(),
                   "kernel_launches_per_spmm_call": a.launches_per_unit,
                   "formula": "(2*FETCH_SIZE + WRITE_SIZE) KiB * 1024 per kernel launch x launches per "
                              "SpMM call (gfx950: FETCH_SIZE halves wide coalesced reads); separate --pmc passes"}
            lines.append(f"\nmain kernel `{a.main_kernel}`: traffic beyond L2 = {traffic / 1e9:.2f} GB per SpMM call "
                         f"({a.launches_per_unit} kernel launch(es) per call)")
            nk = pmc.get(a.narrow_kernel) if a.narrow_kernel else None
            if nk and "FETCH_SIZE" in nk and "WRITE_SIZE" in nk:
                tn = int((2 * nk["FETCH_SIZE"] + nk["WRITE_SIZE"]) * 1024 * a.narrow_launches_per_unit)
                rec["narrow_kernel"] = a.narrow_kernel
                rec["bytes_per_call_narrow"] = tn
                rec["kernel_launches_per_spmm_call_narrow"] = a.narrow_launches_per_unit
                lines.append(f"narrow kernel `{a.narrow_kernel}`: traffic beyond L2 = {tn / 1e9:.2f} GB per SpMM call "
                             f"({a.narrow_launches_per_unit} kernel launch(es) per call)")
            if not a.no_json:
                json.dump(rec, open(os.path.join(HERE, "spmm_hbm_traffic.json"), "w"), indent=1)
    out = os.path.join(HERE, f"{a.tag}_summary.md")
    open(out, "w").write("\n".join(lines) + "\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
