cd $GRAFT_REPO_ROOT
export MGGCN_OVERSUBSCRIBE=1
B=mg-gcn_amd/bin
for q in 4 8 24; do
for P in 4 8; do
  SECONDS=0
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 $B/test_dist $P > /dev/null 2>&1
  echo "queues $q P $P rc $? seconds $SECONDS"
done; done
