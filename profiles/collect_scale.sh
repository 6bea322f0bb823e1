#!/bin/bash
# The scaling table in ONE command, for whoever holds an 8-GPU MI355X node (the builder's pool has one-GPU boxes only;
# the driver's SCALE run covers N = 1, 2, 4, 8 in the default mode):
#
#   bash profiles/collect_scale.sh [steps] [warmup]   ->  profiles/scale_<date>.jsonl  (one bench.py line per run)
#
# N = 1, 2, 4, 8  x  exchange schedule {allgather (K pieces, default), rounds (the reference's broadcast pipeline,
# src/cuda_utils.hpp:57-92), halo}  x  overlap {on, off = the reference's -S}.  Every line carries
# comm{backend, world_size, devices[], rccl_version, exchange_ms, exposed_ms, overlap_frac} (bench.py: comm_report).
# The Python ranks exchange over torch.distributed = RCCL; every N > 1 line also carries the drop-in CLI's epoch on the same files in
# three legs (cli_*: one enqueue thread per GPU over RCCL, cli_p2p_*: copy-engine peer copies, cli_serial_*: one host thread), each
# with host_issue_ms -- the table below prints them.  bench.py starts its ranks itself (no launcher needed).
set -u
STEPS=${1:-20}
WARMUP=${2:-3}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/profiles/scale_$(date +%Y%m%d_%H%M).jsonl
export HSA_ENABLE_IPC_MODE_LEGACY=0
PORT=29600
: > "$OUT"
python3 "$ROOT/bench.py" --gpus 1 --steps "$STEPS" --warmup "$WARMUP" --no-extras >> "$OUT" || echo '{"error": "N=1"}' >> "$OUT"
for N in 2 4 8; do
  for MODE in allgather rounds halo; do
    for OV in "" "--no-overlap"; do
      PORT=$((PORT + 1))
      echo "N=$N mode=$MODE $OV" >&2
      timeout -k 10 1500 python3 "$ROOT/bench.py" --gpus "$N" --steps "$STEPS" --warmup "$WARMUP" --mode "$MODE" $OV >> "$OUT" \
        || echo "{\"error\": \"N=$N mode=$MODE $OV\"}" >> "$OUT"
    done
  done
done
# the all-gather exchange in 1 / 2 / 8 pieces at the largest N (default is 4: every piece is one more collective)
for K in 1 2 8; do
  PORT=$((PORT + 1))
  echo "N=8 mode=allgather chunks=$K" >&2
  timeout -k 10 1500 python3 "$ROOT/bench.py" --gpus 8 --steps "$STEPS" --warmup "$WARMUP" --mode allgather --chunks "$K" >> "$OUT" \
    || echo "{\"error\": \"N=8 allgather chunks=$K\"}" >> "$OUT"
done
python3 - "$OUT" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
base = next((r["value"] for r in rows if r.get("n_gpus") == 1), None)
print("| N | mode (pieces) | overlap | epoch ms | speed-up | exchange ms / SpMM | exposed ms | overlap frac | CLI rccl ms (host issue) | CLI p2p ms | CLI one-thread ms (host issue) |\n|---|---|---|---|---|---|---|---|---|---|---|")
for r in rows:
    if "error" in r:
        print("|", r["error"], "| failed |||||||"); continue
    c = r.get("comm") or {}
    print(f"| {r['n_gpus']} | {c.get('mode', '-')} ({c.get('chunks') or 'default'}) | {c.get('overlap', '-')} | {r['value']:.3f} | "
          f"{(base / r['value']) if base else float('nan'):.2f} | {c.get('exchange_ms', '-')} | {c.get('exposed_ms', '-')} | {c.get('overlap_frac', '-')} | "
          f"{r.get('cli_epoch_ms', r.get('cli_error', '-'))} ({r.get('cli_host_issue_ms', '-')}) | {r.get('cli_p2p_epoch_ms', r.get('cli_p2p_error', '-'))} | "
          f"{r.get('cli_serial_epoch_ms', r.get('cli_serial_error', '-'))} ({r.get('cli_serial_host_issue_ms', '-')}) |")
PY
