# -P 8 wrapped over ONE card at the Reddit shape: which of round 4's changes moved the (meaningless, but visible) epoch time there
cd $GRAFT_REPO_ROOT
D=/tmp/reddit_like/permuted/reddit
python3 - <<'PY'
import sys
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_package()
(ip, ix, dv), X, Y = pkg.datasets.synth_reddit_like(1.0, seed=1)
pkg.datasets.write_dataset('/tmp/reddit_like/permuted/reddit', ip, ix, dv, X, Y)
PY
cd /tmp/reddit_like
run() { echo "== $*"; env "$@" MGGCN_TIMING=1 MGGCN_OVERSUBSCRIBE=1 $GRAFT_REPO_ROOT/mg-gcn_amd/bin/mg_gcn -P 8 -R 1 -E 6 train $D 3 128 128 128 2>&1 | python3 -c "
import sys; L=sys.stdin.read().splitlines(); ep=[float(l.split()[3]) for l in L if len(l.split())==4 and l.split()[0].isdigit()][2:]; hi=[float(l.split()[5]) for l in L if 'host-issue-ms' in l][2:]; print('epoch_ms %.1f host_issue_ms %.1f' % (1e3*sorted(ep)[len(ep)//2], sorted(hi)[len(hi)//2]))"; }
run A=default
run MGGCN_P2P_PEER_STREAMS=0
run MGGCN_ENQUEUE_THREADS=0
run MGGCN_ENQUEUE_THREADS=0 MGGCN_P2P_PEER_STREAMS=0
run MGGCN_P2P_PEER_STREAMS=0 GPU_MAX_HW_QUEUES=8
run MGGCN_DIST_MODE=rounds
run MGGCN_DIST_MODE=rounds MGGCN_P2P_PEER_STREAMS=0
