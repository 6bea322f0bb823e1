cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_dist_host_trace
mkdir -p $O
export EXP_CHUNKS=4
timeout -k 10 300 rocprofv3 --kernel-trace -d $O -o k4 --output-format csv -- python3 $R/profiles/experiments/dist_host_time_r04.py > $O/out.log 2> $O/err.log
grep "^n =" $O/out.log
python3 - <<PY
import pandas as pd
df = pd.read_csv("$O/k4_kernel_trace.csv").sort_values("Start_Timestamp")
t0 = df.Start_Timestamp.min()
df["s"] = (df.Start_Timestamp - t0) / 1e3; df["e"] = (df.End_Timestamp - t0) / 1e3
df["k"] = df.Kernel_Name.str.replace("void (anonymous namespace)::","").str.replace("(anonymous namespace)::","").str.slice(0, 34)
sub = df.tail(420).head(140)
prev = None
for _, r in sub.iterrows():
    gap = (r.s - prev) if prev is not None else 0
    print(f"{r.Stream_Id:3d} {r.k:34s} start {r.s:12.1f} dur {r.e - r.s:7.1f} gap_since_prev_end {gap:8.1f}")
    prev = r.e
PY
