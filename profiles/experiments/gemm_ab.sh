# usage (on the GPU box): bash profiles/experiments/gemm_ab.sh "<-D flags of variant A>" "<-D flags of variant B>" ...
# Builds profiles/experiments/gemm_timeline.hip (= the product gemm.hip, marks off) once per variant and times the
# epoch's shapes with every variant on the SAME box, 20 back-to-back calls each.
cd $GRAFT_REPO_ROOT
i=0
for flags in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DNO_MARKS $flags -I include -I mg-gcn_amd/csrc profiles/experiments/gemm_timeline.hip \
      -L mg-gcn_amd/lib -lmggcn_hip -Wl,-rpath,$PWD/mg-gcn_amd/lib -o /tmp/gemm_ab_$i 2>/dev/null || { echo "build failed: $flags"; exit 1; }
  i=$((i+1))
done
for round in 1 2; do
  i=0
  for flags in "$@"; do
    echo "== variant $i: ${flags:-default} (round $round)"
    timeout -k 5 60 /tmp/gemm_ab_$i 232968 608 0 40 && timeout -k 5 60 /tmp/gemm_ab_$i 232968 128 0 40 && timeout -k 5 60 /tmp/gemm_ab_$i 608 232968 1 40 && timeout -k 5 60 /tmp/gemm_ab_$i 128 232968 1 40 && timeout -k 5 60 /tmp/gemm_ab_$i 232968 128 0 40 41 && timeout -k 5 60 /tmp/gemm_ab_$i 128 232968 1 40 41 || exit 1
    i=$((i+1))
  done
done
