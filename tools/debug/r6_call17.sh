cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call17; mkdir -p $O
python3 - <<'PY' 2>&1 | grep -v amdgpu
import io, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from PIL import Image
from witw_amd import jpeg
sys.path.insert(0, 'tools/debug')
g = np.random.Generator(np.random.Philox(key=[1, 2]))
def picture(h, w):
    small = g.integers(0, 256, size=(h // 8 + 2, w // 8 + 2, 3), dtype=np.uint8)
    img = np.asarray(Image.fromarray(small).resize((w + 16, h + 16), Image.BICUBIC))[8:8 + h, 8:8 + w]
    return np.clip(img.astype(np.int16) + g.integers(-12, 13, size=(h, w, 3)), 0, 255).astype(np.uint8)
dev = torch.device('cuda:0')
for (h, w) in ((512, 512), (224, 224)):
    files = []
    for i in range(16):
        b = io.BytesIO(); Image.fromarray(picture(h, w)).save(b, 'JPEG', quality=90); files.append(b.getvalue())
    items = [jpeg.open_file(files[i % 16]) for i in range(128)]
    t0 = time.perf_counter(); buf, desc, _k = jpeg.pack(items); t_pack = time.perf_counter() - t0
    dbuf = buf.pin_memory().to(dev)
    for _ in range(2): keep, table = jpeg.decode_packed(dbuf, desc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): keep, table = jpeg.decode_packed(dbuf, desc)
    e1.record(); torch.cuda.synchronize()
    print('no restart markers %dx%d: %d files, %.1f KB each, host parse+pack %.2f ms, device decode (self-sync huffman + idct + rgb) %.3f ms per batch, errors %d'
          % (h, w, len(items), len(files[0]) / 1e3, t_pack * 1e3, e0.elapsed_time(e1) / 5, jpeg.entropy_errors()))
PY
D=$(mktemp -d /tmp/witw_e2e_XXXX)
for wk in 4 16; do
timeout -k 10 400 python3 bench.py --mode e2e --precision bf16 --workers $wk --e2e-pairs 8192 --e2e-dir $D --no-decode-scaling --detail-out $O/e2e_w$wk.json > /dev/null 2> $O/e2e_w$wk.err
python3 -c "
import json; d=json.load(open('$O/e2e_w$wk.json')); print('selfsync w$wk', d['value'], d['steady_state_pairs_per_s'], {k[:36]: v for k, v in d['stage_pairs_per_s'].items()}, d['pcie_bytes_per_pair'], d['jpeg_decode'][:60])"
done
WITW_JPEG_DEVICE_ENTROPY=0 timeout -k 10 400 python3 bench.py --mode e2e --precision bf16 --workers 16 --e2e-pairs 8192 --e2e-dir $D --no-decode-scaling --detail-out $O/e2e_host_w16.json > /dev/null 2> $O/e2e_host_w16.err
python3 -c "
import json; d=json.load(open('$O/e2e_host_w16.json')); print('host entropy w16', d['value'], d['steady_state_pairs_per_s'], {k[:36]: v for k, v in d['stage_pairs_per_s'].items()}, d['pcie_bytes_per_pair'])"
rm -rf $D
