import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from witw_amd import ops, _lib
dev = torch.device('cuda:0')
lib = _lib.load()
SHAPES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(128, 128, 64), (128, 128, 12), (128, 128, 33), (100, 77, 64), (256, 128, 64)]
for (bo, bs, we) in SHAPES:
    ov = torch.randn((bo, 16, 4, 64), device=dev); su = torch.randn((bs, 16, 4, we), device=dev)
    ori = torch.empty((bo, bs), dtype=torch.int64, device=dev); d = torch.empty((bo, bs), device=dev); sc = torch.empty((bo, bs), device=dev)
    ws = torch.empty(lib.witw_match_workspace_floats(bo, bs), device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def run():
        _lib.check(lib.witw_match_fwd(ov.data_ptr(), su.data_ptr(), bo, bs, we, ori.data_ptr(), d.data_ptr(), sc.data_ptr(), ws.data_ptr(), st), 'm')
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 50
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * bo * bs * 64 * 64 * we
    # reference orientation via pair kernel? compare with generic path
    os.environ['WITW_MATCH_GENERIC'] = '1'
    print('Bo=%d Bs=%d We=%d: %.1f us (3 launches: 2 norm kernels + match), %.1f TF/s' % (bo, bs, we, ms * 1e3, fl / ms / 1e9))
