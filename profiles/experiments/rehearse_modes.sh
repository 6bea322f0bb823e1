cd $GRAFT_REPO_ROOT
export MGGCN_BENCH_REHEARSAL=1
P=29650
for MODE in allgather rounds halo; do for OV in "" "--no-overlap"; do
P=$((P+1))
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $P bench.py --gpus 2 --steps 2 --warmup 1 --scale 0.05 --mode $MODE $OV 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = j['comm']
print('$MODE', '$OV', 'value', j['value'], {k: c.get(k) for k in ('backend','mode','overlap','exchange_ms','exposed_ms','overlap_frac')})"
done; done
