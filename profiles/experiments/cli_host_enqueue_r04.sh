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
for MODE in allgather rounds; do for TH in 1 0; do
echo "== -P $P mode=$MODE threads=$TH"
MGGCN_TIMING=1 MGGCN_ENQUEUE_THREADS=$TH MGGCN_OVERSUBSCRIBE=1 MGGCN_DIST_MODE=$MODE timeout -k 10 120 $GRAFT_REPO_ROOT/mg-gcn_amd/bin/mg_gcn -P $P -R 1 -E 12 train $D 3 128 128 128 2>&1 | python3 -c "import sys; L=sys.stdin.read().splitlines(); ep=[float(l.split()[3]) for l in L if len(l.split())==4 and l.split()[0].isdigit()][4:]; hi=[float(l.split()[5]) for l in L if \"host-issue-ms\" in l][4:]; print(\"epoch_ms %.3f host_issue_ms %.3f\" % (1e3*sorted(ep)[len(ep)//2], sorted(hi)[len(hi)//2]))"
echo; done
done
done
