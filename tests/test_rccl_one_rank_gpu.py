"""RCCL on the one GPU of the test box. A second rank cannot share the device under RCCL, so this is the degenerate world of ONE
rank -- but it is RCCL itself (backend 'nccl', init_process_group with device_id, the communicator, its kernels and streams) that
the calls of witw_amd/parallel.py meet here for the first time: the step's three payloads through `all_gather_into_tensor`,
`reduce_scatter_tensor` and the asynchronous bucket `all_reduce` + wait of OverlappedGradReducer (DataParallel semantics of the
reference: model/cvig_baseline.py:339-343, global-batch normaliser model/cvig_fov.py:380), on the stream discipline the training
step uses, with the results checked (one rank: every collective is the identity). Runs in a child process (a process group is
process-wide state)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, json, time
sys.path.insert(0, %r)
import torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', %r)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
from witw_amd import parallel, cvig_fov, synth
out = {'backend': dist.get_backend(), 'rccl': '.'.join(str(v) for v in torch.cuda.nccl.version())}
b, E = 128, 16 * 4 * 64
ov = torch.randn((b, E), device=dev)
got = parallel._all_gather_cat(ov)                                  # all_gather_into_tensor, 2 MiB
out['all_gather_identity'] = bool(torch.equal(got, ov))
g = torch.randn((b, E), device=dev)
rs = torch.empty_like(g)
dist.reduce_scatter_tensor(rs, g, op=dist.ReduceOp.SUM)             # what reduce_scatter_rows issues under RCCL
out['reduce_scatter_identity'] = bool(torch.equal(rs, g))
# the gradient path of one training step: two encoders, buckets, asynchronous all-reduce launched from the backward, wait, Adam
w = synth.fov_dsm_weights(3)
se = cvig_fov.FOV_DSM(False, weights=w).to(dev).train()
oe = cvig_fov.FOV_DSM(True, weights=w).to(dev).train()
opt = cvig_fov.Adam(list(se.parameters()) + list(oe.parameters()), lr=1e-5)
red = parallel.OverlappedGradReducer([se, oe])
real_world = parallel.world
xs = torch.from_numpy(synth.normalized_images(3, 1, (4, 3, 128, 512))).to(dev)
xo = torch.from_numpy(synth.normalized_images(3, 2, (4, 3, 128, 512))).to(dev)
parallel.world = lambda: 2
try:
    su, ovv = se(xs), oe(xo)
    parallel.world = real_world                                     # the loss itself on the single-rank path
    ori, dd = cvig_fov.match(ovv, su)
    loss = cvig_fov.triplet_loss(dd)
    opt.zero_grad()
    parallel.world = lambda: 2
    loss.backward()
    launched = sorted(red.inflight)
    n = red.wait()
finally:
    parallel.world = real_world
out['buckets_all_reduced'] = launched
out['floats_reduced'] = int(n)
ref = [p.grad.clone() for p in list(se.parameters()) + list(oe.parameters()) if p.grad is not None]
out['grads_finite'] = bool(all(torch.isfinite(t).all() for t in ref))
opt.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    dist.all_reduce(red.buckets[0].flat, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()
out['all_reduce_29MB_us'] = round((time.perf_counter() - t0) / 10 * 1e6, 1)
dist.barrier()
dist.destroy_process_group()
print('RCCL1 ' + json.dumps(out))
'''


def test_rccl_world_of_one_runs_the_steps_collectives():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    p = subprocess.run([sys.executable, '-c', CHILD % (ROOT, str(port))], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    import json
    line = [ln for ln in p.stdout.splitlines() if ln.startswith('RCCL1 ')]
    assert line, p.stdout[-2000:]
    out = json.loads(line[0][6:])
    assert out['backend'] == 'nccl' and out['rccl']
    assert out['all_gather_identity'] and out['reduce_scatter_identity']
    assert out['buckets_all_reduced'] == [0, 1] and out['floats_reduced'] == 2 * 7236432 and out['grads_finite']
    assert out['all_reduce_29MB_us'] > 0
