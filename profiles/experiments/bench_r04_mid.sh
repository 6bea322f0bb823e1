cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 500 python bench.py > gpurun_out/r04/bench_n1.json 2> gpurun_out/r04/bench_n1.err
echo "n1 rc $?"; cut -c1-700 gpurun_out/r04/bench_n1.json
MGGCN_BENCH_REHEARSAL=1 timeout -k 10 600 python bench.py --gpus 4 --steps 3 --warmup 1 > gpurun_out/r04/bench_rehearsal4.json 2> gpurun_out/r04/bench_rehearsal4.err
echo "rehearsal4 rc $?"; python3 -c "
import json; j=json.load(open('gpurun_out/r04/bench_rehearsal4.json')); print({k:v for k,v in j.items() if k.startswith('cli') or k in ('value','n_gpus','loss_first_last')}); print(j['comm'])"
