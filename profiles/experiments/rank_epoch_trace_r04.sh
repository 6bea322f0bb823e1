# kernel stats of rank 0's exchange-free epoch at P = 8 (where do the 2.4 ms of SpMM calls go?)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_rank_epoch_p8
mkdir -p $O
export RANK_EPOCH_P=8
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O -o p8 --output-format csv -- python3 $R/profiles/experiments/rank_epoch_r04.py > $O/out.log 2> $O/err.log
grep "^P=" $O/out.log
python3 - <<PY
import pandas as pd
df = pd.read_csv("$O/p8_kernel_stats.csv")
df["Name"] = df["Name"].str.replace("void (anonymous namespace)::","").str.replace("(anonymous namespace)::","").str.slice(0,60)
print(df[["Name","Calls","TotalDurationNs","AverageNs","Percentage"]].head(16).to_string())
PY
