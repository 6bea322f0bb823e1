# round 3: which clock does each kernel of the epoch run at?  GRBM_GUI_ACTIVE (cycles, summed over the
# 8 XCDs) / kernel duration per dispatch, inside the bench's own epochs.  bash profiles/experiments/epoch_clocks.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/epoch_clocks
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/clk -o clk --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench.json 2> $O/err.log
echo rc=$?
python3 - $O/clk/clk_counter_collection.csv <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
        continue
    dur_us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if dur_us < 20:
        continue
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    agg[(name, r["Grid_Size"])].append((dur_us, float(r["Counter_Value"]) / 8 / dur_us / 1e3))
print("| kernel | grid | dispatches | mean us | min us | GHz (GRBM_GUI_ACTIVE / 8 / duration) |\n|---|---|---|---|---|---|")
for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
    print(f"| {k} | {g} | {len(v)} | {sum(x[0] for x in v) / len(v):.1f} | {min(x[0] for x in v):.1f} | {sum(x[1] for x in v) / len(v):.3f} |")
PY
