cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call29; mkdir -p $O
python3 - <<PY
import torch
from witw_amd import parallel
s = parallel.masked_stream(32)
print('masked stream:', s)
if s is not None:
    with torch.cuda.stream(s):
        x = torch.ones(1 << 20, device='cuda') * 2
    s.synchronize(); print('ran on it:', float(x.sum()))
PY
D=$(mktemp -d /tmp/witw_e2e_XXXX)
for cfg in "0 stride" "32 stride" "64 stride" "128 stride" "64 block"; do
set -- $cfg
WITW_STAGING_CUS=$1 WITW_STAGING_CU_PATTERN=$2 timeout -k 10 400 python3 bench.py --mode e2e --precision bf16 --workers 4 --e2e-pairs 8192 --e2e-dir $D --device-entropy all --detail-out $O/e2e_all_$1_$2.json > /dev/null 2> $O/e2e_all_$1_$2.err
python3 -c "
import json; d=json.load(open('$O/e2e_all_$1_$2.json')); print('e2e bf16 all, 4 workers, staging CUs $1 $2:', d['value'], d['steady_state_pairs_per_s'], {k[:12]: v for k, v in d['stage_pairs_per_s'].items()})"
done
rm -rf $D
