cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call48; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_bf16_train_gpu.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 $O/pytest.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
python3 tools/bench_wgrad_bf16.py 2>&1 | grep -v amdgpu.ids | tee $O/wgrad_layers.txt | tail -14
python3 bench.py --mode train --precision bf16 --detail-out $O/d.json 2> /dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 train', d['value'], d['ms_per_step'])"
python3 bench.py --model semantic --mode train --precision bf16 --detail-out $O/d.json 2> /dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('semantic bf16 train', d['value'], d['ms_per_step'])"
