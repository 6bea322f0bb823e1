# the exchange model (rank_epoch_model_r04.py) with an UNEQUAL first piece: fraction of the shard in piece 0 (MGGCN_DIST_FIRST_PIECE)
# (ran against a build whose dist.chunk_bounds took the first piece as a fraction of the shard -- MGGCN_DIST_FIRST_PIECE, default 1/P when P > K;
#  the change was not kept: the product cuts equal pieces)
cd $GRAFT_REPO_ROOT
export EXP_REPS=30 EXP_WARMUP=20
run() { line="P=$1 K=$2 first=$3:"; for G in $4; do out=$(MGGCN_DIST_FIRST_PIECE=$3 RANK_EPOCH_P=$1 EXP_CHUNKS=$2 EXP_GBPS=$G timeout -k 10 200 python3 profiles/experiments/rank_epoch_model_r04.py 2>/dev/null | grep "^P=" | sed 's/.*no exchange \([0-9.]*\) ms.*/\1/'); line="$line  ${G}GB/s ${out}ms"; done; echo "$line"; }
for f in 0.25 0.18 0.125 0.08; do run 8 4 $f "1e9 450 350 250"; done
for f in 0.333 0.2 0.125; do run 8 3 $f "1e9 450 350 250"; done
for f in 0.5 0.3 0.2; do run 8 2 $f "350 250"; done
for f in 0.25 0.18 0.12; do run 4 4 $f "200 150"; done
