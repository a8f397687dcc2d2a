#!/bin/bash
# A/B of two builds' peer exchange with the ranks as processes on ONE GPU (no xGMI hop: what it shows is the protocol's own
# cost -- stores, waits, reads -- and that nothing got slower).  usage: tools/ab_p2p_protocol.sh <other lib> [world] [samples]
set -e
cd "$(dirname "$0")/.."
other=$1; world=${2:-3}; total=${3:-300000}
run() {  # $1 = SALNMF_LIB or empty
    SALNMF_LIB=$1 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$world --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) \
        bench.py --gpus $world --steps 20 --warmup 5 --rehearse-one-device --samples-total $total --busy-seconds 1.0 --cpu-steps-sharded 3 2>/dev/null | tail -1 | python -c '
import json, sys
d = json.loads(sys.stdin.readline()); t = d["config"]["exchange"]["timeline_us_per_rank"][0]
print("ms_per_step %.5f | " % d["ms_per_step"] + " ".join("%s %.2f" % (k, t[k]) for k in ("tail_exchange_launch", "local_slab_reduce", "peer_stores_and_flags", "wait_for_peer_flags", "read_and_sum_peer_rows", "w_row_finish")))'
}
for i in 1 2; do
    echo -n "this build : "; run ""
    echo -n "other build: "; run "$PWD/$other"
done
