# host-side enqueue cost of the C++ CLI per epoch as a function of -P: a graph so small that the GPU work is negligible
# (n = 2328, nnz = 1.1 M), ranks wrapped over the one card (MGGCN_OVERSUBSCRIBE=1).  What it reads: epoch seconds ~ host time.
cd $GRAFT_REPO_ROOT
D=/tmp/small/permuted/synth
python3 - <<'PY'
import sys, importlib
sys.path.insert(0, '.')
pkg = importlib.import_module('mg-gcn_amd')
(ip, ix, dv), X, Y = pkg.datasets.synth_reddit_like(0.01, seed=3)
pkg.datasets.write_dataset('/tmp/small/permuted/synth', ip, ix, dv, X, Y)
print('n', len(ip) - 1, 'nnz', len(ix))
PY
cd /tmp/small
for P in 1 2 4 8; do
for MODE in allgather rounds; do
echo "== -P $P mode=$MODE"
MGGCN_OVERSUBSCRIBE=1 MGGCN_DIST_MODE=$MODE timeout -k 10 120 $GRAFT_REPO_ROOT/mg-gcn_amd/bin/mg_gcn -P $P -R 1 -E 12 train $D 3 128 128 128 2>&1 | awk 'NF==4 && $1 ~ /^[0-9]+$/ {print $1, $4}' | tail -4 | tr '\n' ';'
echo
done
done
