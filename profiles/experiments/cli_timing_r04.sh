# start-up stages of the CLI at the Reddit shape (MGGCN_TIMING=1), default 20 epochs; single GPU and -P 8 wrapped over one card
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
for i in 1 2; do
  SECONDS=0; s=$(date +%s%N)
  MGGCN_TIMING=1 $GRAFT_REPO_ROOT/mg-gcn_amd/bin/mg_gcn train $D 3 128 128 128 2> /tmp/cli.err
  e=$(date +%s%N); echo "run $i: process wall $(( (e - s) / 1000000 )) ms"
  grep "mggcn timing" /tmp/cli.err | grep -v "host-issue"; grep -E "^(0|1|19) " /tmp/cli.err
done
s=$(date +%s%N)
MGGCN_TIMING=1 MGGCN_OVERSUBSCRIBE=1 $GRAFT_REPO_ROOT/mg-gcn_amd/bin/mg_gcn -P 8 -R 1 -E 4 train $D 3 128 128 128 2> /tmp/cli8.err
e=$(date +%s%N); echo "-P 8: process wall $(( (e - s) / 1000000 )) ms"
grep "mggcn timing" /tmp/cli8.err | grep -v "host-issue"; grep -E "^(0|1|2|3) " /tmp/cli8.err
