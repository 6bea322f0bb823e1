#!/bin/bash
# the push form of the peer-copy transport: the suite's cases, then `mg_gcn -P 8` on ONE card, pull against push (no links here:
# what the protocol itself costs the host and the device; the links are for the first multi-GPU run: bench.py cli_p2p_push_*)
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
LOG=gpurun_out/p2p_push_r04.log
{
  if [ "${SKIP_TESTS:-0}" != 1 ]; then timeout -k 10 500 python3 -m pytest tests/test_gpu_host_cpp.py tests/test_gpu_bench.py -x -q -k "senders_push or rccl_with_one_rank"; fi &&
  python3 - <<'PY'
import os, sys, subprocess, tempfile, re
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
pkg = ge.load_package()
(ip, ix, dv), X, Y = pkg.datasets.synth_reddit_like(1.0, seed=1)
tmp = tempfile.mkdtemp()
d = os.path.join(tmp, "permuted", "bench")
pkg.datasets.write_dataset(d, ip, ix, dv, X, Y)
for name, env in (("pull", {}), ("push", {"MGGCN_P2P_PUSH": "1"}), ("pull", {}), ("push", {"MGGCN_P2P_PUSH": "1"})):
    e = dict(os.environ, MGGCN_OVERSUBSCRIBE="1", MGGCN_TIMING="1", **env)
    r = subprocess.run([os.path.join(os.getcwd(), "mg-gcn_amd", "bin", "mg_gcn"), "-P", "8", "-R", "1", "-E", "10", "train", d, "3", "128", "128", "128"],
                       cwd=tmp, env=e, capture_output=True, text=True, timeout=280)
    ep = [float(l.split()[3]) for l in r.stderr.splitlines() if len(l.split()) == 4 and l.split()[0].isdigit()]
    issue = [float(l.split()[5]) for l in r.stderr.splitlines() if "host-issue-ms" in l]
    tr = re.search(r"transport (\S+)", r.stderr)
    print(f"-P 8 one card, {name} ({tr.group(1) if tr else '?'}): median epoch {sorted(ep[2:])[len(ep[2:]) // 2] * 1e3:.2f} ms, "
          f"host issue {sorted(issue[2:])[len(issue[2:]) // 2]:.2f} ms", flush=True)
PY
} > $LOG 2>&1
echo "exit $?" >> $LOG
tail -25 $LOG
