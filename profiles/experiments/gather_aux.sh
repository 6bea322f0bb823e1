# usage (GPU box): bash profiles/experiments/gather_aux.sh
# Cache-policy bits on the row gathers of spmm_sweep_pair_kernel: rebuilds the library on the box once per variant
# (hipcc is in the image) and times the four SpMM shapes with each.  gfx940+ encoding: 1 = sc0, 2 = nt, 16 = sc1.
cd $GRAFT_REPO_ROOT
BASE="-O3 -std=c++17 -fPIC -fvisibility=hidden -I$PWD/include -Wall -Wextra -Wno-unused-parameter"
for aux in 0 1 2 16 17 3 0; do
  touch mg-gcn_amd/csrc/spmm_sweep.hip
  make -s -C mg-gcn_amd/csrc CXXFLAGS="$BASE -DMGGCN_GATHER_AUX=$aux" > /dev/null 2>&1 || { echo "build failed aux=$aux"; exit 1; }
  python profiles/experiments/spmm_ab.py "gather aux=$aux" 2>&1 | tail -1
done
