"""End-to-end data path of the cvig_fov drivers, measured: image files on disk -> ImagePairDataset in DataLoader worker
processes (decode) -> collate_packed (one byte block per side) -> pinned memory -> PCIe on a copy stream (DevicePrefetcher)
-> batched resize + normalise + polar on the GPU (GpuPreprocess, 3 launches) -> the two FOV_DSM encoders -> embeddings.

The reference runs the transforms inside 12 DataLoader workers at about one second per sample (PolarTransform in numpy,
model/cvig_fov.py:186-209, 393-403); here the workers only decode. `bench()` reports the disk -> embedding rate next to the
rate of every stage on its own, and names the slowest."""
import os
import time

import numpy as np
import torch


def _write_pair(args):
    root, i, seed, restart_rows, restart_blocks = args
    from PIL import Image
    g = np.random.Generator(np.random.Philox(key=[seed, i]))

    def picture(h, w):
        # band-limited noise: decodes like a photograph (pure noise would be the JPEG decoder's worst case, flat colour its best)
        small = g.integers(0, 256, size=(h // 8 + 2, w // 8 + 2, 3), dtype=np.uint8)
        img = np.asarray(Image.fromarray(small).resize((w + 16, h + 16), Image.BICUBIC))[8:8 + h, 8:8 + w]
        fine = g.integers(-12, 13, size=(h, w, 3))
        return np.clip(img.astype(np.int16) + fine, 0, 255).astype(np.uint8)
    for name, (h, w) in (('ov_%05d.jpg' % i, (512, 512)), ('su_%05d.jpg' % i, (224, 224))):
        # written under a temporary name and renamed into place: a writer killed half way (bench.run_child's timeout) never leaves
        # a truncated file under the name the next block would reuse
        tmp = os.path.join(root, '.%s.%d.tmp' % (name, os.getpid()))
        # restart_rows > 0: a restart marker every that many MCU rows (libjpeg's -restart N; 1.3 % larger files at one row) -- such
        # files are entropy-decoded on the GPU, one thread per restart interval (witw_amd/jpeg.py DEVICE_ENTROPY)
        # restart_blocks > 0: a marker every that many MCUs (-restart NB): the shorter an interval, the shorter the sequential chain of
        # the GPU thread that decodes it (8 MCUs: +0.4 % file size)
        kw = {'restart_marker_blocks': restart_blocks} if restart_blocks else {'restart_marker_rows': restart_rows} if restart_rows else {}
        Image.fromarray(picture(h, w)).save(tmp, format='JPEG', quality=90, **kw)
        os.replace(tmp, os.path.join(root, name))
    return i


def make_dataset(root, n_pairs, n_unique=512, seed=77, procs=8, restart_rows=0, restart_blocks=0):
    """A synthetic cvusa-format data set (CSV columns: overhead, surface; model/cvig_fov.py:38-44): n_pairs rows over
    min(n_pairs, n_unique) distinct JPEG pairs (overhead 512x512, ground 224x224: the raw sizes of BASELINE.json)."""
    import json
    import multiprocessing as mp
    os.makedirs(root, exist_ok=True)
    n_unique = min(n_pairs, n_unique)
    # A kept directory (bench.py --e2e-dir, shared by two blocks) is reused only for what its MANIFEST vouches for: the manifest is
    # written LAST, names the generator (seed, sizes, quality) and how many pairs are complete; files of another seed / size, or of a
    # run that died before its manifest, are written again.
    want = {'generator': 'witw_amd.e2e._write_pair/2', 'seed': int(seed), 'overhead': [512, 512], 'ground': [224, 224], 'quality': 90,
            'restart_marker_rows': int(restart_rows), 'restart_marker_blocks': int(restart_blocks)}
    mpath = os.path.join(root, 'manifest.json')
    have = 0
    try:
        m = json.load(open(mpath))
        if all(m.get(k) == v for k, v in want.items()):
            have = int(m.get('pairs_complete', 0))
    except (OSError, ValueError):
        pass
    todo = [(root, i, seed, int(restart_rows), int(restart_blocks)) for i in range(n_unique)
            if i >= have or not (os.path.exists(os.path.join(root, 'ov_%05d.jpg' % i)) and os.path.exists(os.path.join(root, 'su_%05d.jpg' % i)))]
    if todo:
        if os.path.exists(mpath):
            os.remove(mpath)
        with mp.get_context('spawn').Pool(procs) as pool:
            pool.map(_write_pair, todo, chunksize=16)
    if todo or have < n_unique:
        tmp = mpath + '.%d.tmp' % os.getpid()
        with open(tmp, 'w') as f:
            json.dump(dict(want, pairs_complete=max(have, n_unique)), f)
        os.replace(tmp, mpath)
    csv = os.path.join(root, 'pairs.csv')
    with open(csv, 'w') as f:
        for i in range(n_pairs):
            f.write('ov_%05d.jpg,su_%05d.jpg\n' % (i % n_unique, i % n_unique))
    size = sum(os.path.getsize(os.path.join(root, n)) for n in os.listdir(root) if n.endswith('.jpg'))
    return csv, n_unique, size


def _decode_worker(args):
    """entropy-decode (witw_amd/jpeg.py: header parse + Huffman decoding into coefficient blocks) every pair of `files` over and
    over for `seconds`; -> pairs decoded. Host only: never touches the GPU."""
    files, seconds = args
    from . import jpeg
    blobs = [(np.fromfile(o, dtype=np.uint8), np.fromfile(s, dtype=np.uint8)) for o, s in files]
    bufs = {}
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for pair in blobs:
            for b in pair:
                f = jpeg.open_file(b.tobytes())
                k = int(f.info[5])
                if k not in bufs:
                    bufs[k] = (np.empty((k, 64), np.int16), np.empty((int(f.info[2]), 64), np.uint16))
                f.decode_into(*bufs[k])
            n += 1
    return n / (time.perf_counter() - t0)


def decode_scaling(root, n_unique, cores, gpu_stage_pairs_per_s, seconds=1.0):
    """How the host half of the device-JPEG path (entropy decoding, the only per-image work left on the CPU) scales over processes:
    pairs/s at 1, 2, 4, 8, 16 (at most `cores`, at most the box's CPU share of 16) processes, each decoding its own files in a
    loop with no loader, no packing and no GPU; and what that means for a node: cores needed to feed 8 GPUs at the measured GPU
    stage rate."""
    import multiprocessing as mp
    files = [(os.path.join(root, 'ov_%05d.jpg' % i), os.path.join(root, 'su_%05d.jpg' % i)) for i in range(min(n_unique, 64))]
    out = {}
    for k in [c for c in (1, 2, 4, 8, 16) if c <= max(1, min(cores, 16))]:
        with mp.get_context('spawn').Pool(k) as pool:
            pool.map(_decode_worker, [(files[:2], 0.05)] * k)                      # import + first-touch outside the timing
            rates = pool.map(_decode_worker, [(files[i::k] or files, seconds) for i in range(k)])
        out[k] = sum(rates)
    kmax = max(out)
    per_core_1, per_core_max = out[1], out[kmax] / kmax
    need = 8 * gpu_stage_pairs_per_s / max(1e-9, per_core_max)
    return {'pairs_per_s_by_processes': {str(k): round(v, 1) for k, v in out.items()},
            'host_decode_pairs_per_s_per_core': round(per_core_max, 1), 'host_decode_pairs_per_s_one_process': round(per_core_1, 1),
            'scaling_efficiency_at_%d' % kmax: round(per_core_max / max(1e-9, per_core_1), 3),
            'eight_gpu_budget': {'gpu_stage_pairs_per_s_per_gpu': round(gpu_stage_pairs_per_s, 1), 'host_cores_needed_for_8_gpus': round(need, 1),
                                 'host_cores_visible': cores,
                                 'note': 'cores needed = 8 x the GPU stage rate / pairs per second and core at %d processes; a node whose cores do not cover it '
                                         'is host-bound on this path (Huffman decoding is serial per file; restart-marker or '
                                         'self-synchronising device decoders are not built)' % kmax},
            'what': 'one pair = one 512x512 + one 224x224 quality-90 JPEG; header parse + Huffman decode into int16 coefficient blocks, per process in a loop'}


def bench(a, device, n_pairs=None, workers=None, keep_dir=None, decode=None, precision=None):
    """decode: 'device' (workers entropy-decode, the GPU finishes the JPEG: witw_amd/jpeg.py) or 'host' (Pillow in the workers);
    precision: 'fp32' | 'bf16' | 'fp16x3' encoders."""
    import tempfile
    from . import cvig_fov, synth
    decode = decode or getattr(a, 'decode', 'device')
    precision = precision or getattr(a, 'precision', 'fp32')
    n_pairs = n_pairs or a.e2e_pairs
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    workers = workers if workers is not None else max(1, min(a.workers, cores))
    B, fov = a.batch, a.fov
    tmp = tempfile.TemporaryDirectory(prefix='witw_e2e_')
    root = keep_dir or getattr(a, 'e2e_dir', None) or tmp.name
    t0 = time.perf_counter()
    restart_rows, restart_blocks = int(getattr(a, 'jpeg_restart_rows', 0) or 0), int(getattr(a, 'jpeg_restart_blocks', 0) or 0)
    csv, n_unique, nbytes = make_dataset(root, n_pairs, procs=min(16, cores), restart_rows=restart_rows, restart_blocks=restart_blocks)
    from . import jpeg as jpeg_mod
    if getattr(a, 'device_entropy', None):
        jpeg_mod.DEVICE_ENTROPY = False if a.device_entropy == 'off' else a.device_entropy      # (loader workers are forked below: they inherit it)
    dev_entropy = decode == 'device' and bool(jpeg_mod.DEVICE_ENTROPY) and (restart_rows > 0 or restart_blocks > 0 or jpeg_mod.DEVICE_ENTROPY == 'all')
    t_make = time.perf_counter() - t0

    ds = cvig_fov.ImagePairDataset('cvusa', csv, raw='jpeg' if decode == 'device' else True)
    split = cvig_fov.loader_split(B, workers)      # workers decode quarter batches (as test() does): first batch 4x sooner
    # batch blocks are built by the workers directly in page-locked shared memory (ring.PinnedRing): no pickling copy, no pinning
    # thread; --no-ring: torch's own path (shared-memory pickling + pin_memory thread)
    use_ring = bool(workers) and not getattr(a, 'no_ring', False)
    ring = None
    if use_ring:
        import functools
        from . import ring as ring_mod
        per_part = (B // split) * int(1.25 * (3 * 512 * 512 + 3 * 224 * 224 + 4096))
        ring = ring_mod.PinnedRing(slots=workers * 4 + 2 * split + 4, slot_bytes=per_part)
    collate = functools.partial(cvig_fov.collate_packed, ring=ring) if use_ring else cvig_fov.collate_packed
    loader = torch.utils.data.DataLoader(ds, batch_size=B // split, shuffle=False, drop_last=False, num_workers=workers,
                                         collate_fn=collate, pin_memory=not use_ring,
                                         prefetch_factor=4 if workers else None, persistent_workers=bool(workers))
    prep = cvig_fov.GpuPreprocess('cvusa', fov, random_orientation=False, device=device, ring=ring)
    wts = synth.fov_dsm_weights(1234)
    se = cvig_fov.FOV_DSM(False, weights=wts).to(device).eval()
    oe = cvig_fov.FOV_DSM(True, weights=wts).to(device).eval()

    def embed(st):
        with torch.no_grad():
            data = prep(st)
            if precision == 'bf16':
                return se.forward_bf16(data['surface']), oe.forward_bf16(data['polar'])
            if precision == 'fp16x3':
                return se.forward_f16x3(data['surface']), oe.forward_f16x3(data['polar'])
            return se(data['surface']), oe(data['polar'])

    # ---- stage rates on their own
    # (1) the loader alone: decode + pack in the workers, pinning thread; one untimed pass starts the workers / fills the page cache
    def private_copy(raw):        # a batch kept beyond its slot's life: ordinary pinned blocks
        if 'ring' not in raw:
            return raw
        slot, s_off, s_len, o_off, o_len = raw['ring']
        out = {k: v for k, v in raw.items() if k != 'ring'}
        out['surface_bytes'] = ring.view(slot, s_off, s_len).clone().pin_memory()
        out['overhead_bytes'] = ring.view(slot, o_off, o_len).clone().pin_memory()
        ring.release(slot)
        return out
    first = []                     # the parts of the first full batch
    for raw in loader:
        if len(first) < split:
            first.append(private_copy(raw))
        elif 'ring' in raw:
            ring.release(raw['ring'][0])
    t0 = time.perf_counter()
    n = 0
    for raw in loader:
        n += raw['surface_desc'].shape[0]
        if 'ring' in raw:
            ring.release(raw['ring'][0])
    t_load = time.perf_counter() - t0
    # (2) host -> device copy of one packed batch from pinned memory
    blk = sum(f['surface_bytes'].numel() + f['overhead_bytes'].numel() for f in first)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        for f in first:
            f['surface_bytes'].to(device, non_blocking=True)
            f['overhead_bytes'].to(device, non_blocking=True)
    torch.cuda.synchronize()
    t_h2d = (time.perf_counter() - t0) / 5
    # (3) the GPU side AS THE PIPELINE RUNS IT (DevicePrefetcher): batch i+1 is staged on the copy stream -- H2D of the pinned blocks,
    # [device decode: 2 launches], descriptor tables -- while batch i's 2 preprocessing launches + 2 encoders run on the compute stream.
    # The round-3 figure staged and embedded one batch after the other on ONE stream: the synchronous copy sat inside the stage, the
    # steady state came out FASTER than its "limiting stage" (overlap efficiency 1.19). Both are reported; the bound is the overlapped one.
    copy_stream = torch.cuda.Stream()

    def staged():
        with torch.cuda.stream(copy_stream):
            return prep.stage(first if split > 1 else first[0])

    def gpu_pipelined(n_it):
        nxt = staged()
        for _ in range(n_it):
            cur, nxt = nxt, staged()
            embed(cur)
        return cur
    gpu_pipelined(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = gpu_pipelined(8)
    torch.cuda.synchronize()
    t_gpu = (time.perf_counter() - t0) / 8
    nb = st.n

    def gpu_serial():
        st_ = prep.stage(first if split > 1 else first[0])
        embed(st_)
    gpu_serial()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        gpu_serial()
    torch.cuda.synchronize()
    t_gpu_serial = (time.perf_counter() - t0) / 5

    # ---- end to end: every pair from disk to its two embeddings, prefetch one batch ahead
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    su_all, ov_all = [], []
    t_first, n_first = None, 0
    for st in cvig_fov.DevicePrefetcher(loader, prep, group=split):
        su, ov = embed(st)
        su_all.append(su)
        ov_all.append(ov)
        if t_first is None:          # the pipeline is full once the first batch is through: steady state counts from here
            torch.cuda.synchronize()
            t_first, n_first = time.perf_counter(), st.n
    torch.cuda.synchronize()
    t_end = time.perf_counter()
    t_e2e = t_end - t0
    su_all, ov_all = torch.cat(su_all), torch.cat(ov_all)
    assert su_all.shape[0] == n_pairs and ov_all.shape[0] == n_pairs
    del loader

    rates = {'%s (%d DataLoader workers)' % ('header-parse_marker-scan_and_pack (entropy decoding on the GPU)' if dev_entropy else
                                             'entropy-decode_and_pack' if decode == 'device' else 'decode_and_pack', workers): n / t_load,
             'host_to_device copy (pinned, %.1f MB per batch)' % (blk / 1e6): nb / t_h2d,
             'gpu (%sbatched resize+normalise [+polar, fused], 2 %s encoders; the next batch staged on the copy stream meanwhile, as in the pipeline)'
             % (('JPEG on the GPU: Huffman decoding (%s) + dequantise + IDCT + upsample + colour, ' % ('one thread per restart interval' if restart_rows or restart_blocks else 'self-synchronising, one workgroup per file') if dev_entropy else
                 'JPEG back end: dequantise + IDCT + upsample + colour, ') if decode == 'device' else '', precision): nb / t_gpu}
    limiting = min(rates, key=rates.get)
    scaling = decode_scaling(root, n_unique, cores, nb / t_gpu, seconds=getattr(a, 'decode_scaling_seconds', 1.0)) \
        if decode == 'device' and not dev_entropy and not getattr(a, 'no_decode_scaling', False) else None
    entropy_errors = jpeg_mod.entropy_errors() if dev_entropy else None
    out = {'metric': 'image-pairs/sec (disk -> embeddings)', 'value': round(n_pairs / t_e2e, 2), 'unit': 'pairs/s', 'n_gpus': 1,
           'steps': (n_pairs + B - 1) // B, 'warmup': 0, 'ms_per_step': round(t_e2e / ((n_pairs + B - 1) // B) * 1e3, 3),
           'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': {'fp32': 'f32'}.get(precision, precision), 'data': 'synthetic',
           'jpeg_decode': ('device, entropy decoding included (%s: host = header parse + marker scan only); damaged files '
                           'flagged in the last batches: %d' % ('files with a restart marker per %d MCUs, one GPU thread per interval' % restart_blocks if restart_blocks else
                                                                'files with a restart marker per %d MCU row(s), one GPU thread per interval' % restart_rows if restart_rows else
                                                                'files without restart markers, self-synchronising decode, one workgroup per file',
                                                                entropy_errors)) if dev_entropy else
                          'device (host: entropy decoding only)' if decode == 'device' else 'host (Pillow)',
           'pcie_bytes_per_pair': int(blk / max(1, nb)),
           'staging': ('PinnedRing: %d slots x %.1f MB of page-locked shared memory, workers build the batch blocks in place' % (ring.slots, ring.slot_bytes / 1e6))
           if ring is not None else 'torch: shared-memory pickling + pin_memory thread',
           'config': {'workload': 'cvig_fov fov=%d, %d JPEG pairs on disk (%d distinct files pairs, overhead 512x512 + ground 224x224, %.1f MB) -> '
                                  'ImagePairDataset(raw=%s) in %d DataLoader workers (quarter batches) -> collate_packed -> pinned -> copy stream -> %sGpuPreprocess '
                                  '(2 launches per batch) -> 2x FOV_DSM (%s) -> embeddings' % (fov, n_pairs, n_unique, nbytes / 1e6, "'jpeg'" if decode == 'device' else 'True', workers,
                                                                                                'JPEG back end on the GPU (2 launches) -> ' if decode == 'device' else '', precision),
                      'pairs_per_gpu': B, 'host_cores_available': cores, 'dataset_dir': 'tmpfs/disk under %s' % tempfile.gettempdir(),
                      'reference_data_path': 'model/cvig_fov.py:393-403: transforms inside 12 DataLoader workers, ~1 s per sample '
                                             '(SURVEY §6: PolarTransform 0.9-1.2 s)'},
           'stage_pairs_per_s': {k: round(v, 1) for k, v in rates.items()},
           'limiting_stage': limiting,
           'steady_state_pairs_per_s': round((n_pairs - n_first) / max(1e-9, t_end - t_first), 1),
           'pipeline_fill_s': round(t_first - t0, 3),
           'note': 'value = the whole pass including the pipeline fill (a DataLoader worker decodes one whole batch: the first one '
                   'arrives after ~batch x decode time); steady_state = from the first finished batch to the end',
           # steady state over the slowest stage's own rate. The stage rates are short separate measurements (8 pipelined iterations of
           # ONE batch; one loader pass) and read 1-2 % low against the long pass, so the raw quotient can exceed 1: it is reported
           # clamped, the raw value beside it
           'overlap_efficiency_steady_state': round(min(1.0, ((n_pairs - n_first) / max(1e-9, t_end - t_first)) / min(rates.values())), 3),
           'overlap_efficiency_raw': round(((n_pairs - n_first) / max(1e-9, t_end - t_first)) / min(rates.values()), 3),
           'gpu_stage_serialised_pairs_per_s': round(nb / t_gpu_serial, 1),      # staging and compute on one stream (the round-3 way of timing it)
           **({'host_decode_scaling': scaling, 'host_decode_pairs_per_s_per_core': scaling['host_decode_pairs_per_s_per_core']} if scaling else {}),
           'dataset_written_in_s': round(t_make, 1)}
    if ring is not None:
        ring.close()
    tmp.cleanup()
    return out
