# the model again with a delay kernel that reads the clock every 3.5 us (the first one looked every 55 us and overshot every piece)
cd $GRAFT_REPO_ROOT
export EXP_REPS=30 EXP_WARMUP=20
run() { line="P=$1 K=$2:"; for G in $3; do out=$(RANK_EPOCH_P=$1 EXP_CHUNKS=$2 EXP_GBPS=$G timeout -k 10 200 python3 profiles/experiments/rank_epoch_model_r04.py 2>/dev/null | grep "^P=" | sed 's/.*no exchange \([0-9.]*\) ms.*/\1/'); line="$line  ${G}GB/s ${out}ms"; done; echo "$line"; }
for K in 1 2 3 4 6; do run 8 $K "1e9 450 350 250"; done
for K in 1 2 3 4 6; do run 4 $K "1e9 200 150 100"; done
for K in 1 2 3 4; do run 2 $K "1e9 75 50"; done
