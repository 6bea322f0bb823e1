# instruction mix of the d = 128 pair kernel, general instance against FAST (one SQ pass each; run through gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_fast_pairs
mkdir -p $O
for V in 0 1; do
  export MGGCN_SPMM_FAST_PAIRS=$V
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d $O/fast$V -o sq --output-format csv -- python3 $R/profiles/experiments/one_spmm.py 128 3 > $O/fast$V.log 2>&1
  echo "fast=$V rc=$?"
done
python3 - <<'PY'
import csv, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_fast_pairs"
for v in (0, 1):
    path = None
    for dp, _, fs in os.walk(f"{root}/fast{v}"):
        for f in fs:
            if f.endswith("counter_collection.csv"):
                path = os.path.join(dp, f)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        if "spmm_sweep_pair_kernel" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        m = {n: sum(x) / len(x) for n, x in c.items()}
        g = m.get("SQ_INSTS_VMEM_RD", 1)
        print(f"FAST={v} {k}: per gather VALU {m['SQ_INSTS_VALU']/g:.2f} SALU {m['SQ_INSTS_SALU']/g:.2f} SMEM {m['SQ_INSTS_SMEM']/g:.3f}; "
              f"VALU busy {100*m['SQ_ACTIVE_INST_VALU']/m['SQ_BUSY_CYCLES']/4:.1f} % scalar busy {100*m['SQ_ACTIVE_INST_SCA']/m['SQ_BUSY_CYCLES']/4:.1f} % "
              f"(launches {len(next(iter(c.values())))})")
PY
