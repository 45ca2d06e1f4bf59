"""On-device JPEG decode for the data path: the reference decodes inside its DataLoader workers
(skimage.io.imread -> PIL -> libjpeg, model/cvig_fov.py:88-89, :402). Here a worker only ENTROPY-decodes
(libwitw_jpeg.so, csrc_host/jpeg_coef.cpp: Huffman decoding into quantised DCT coefficient blocks, about 40 % of libjpeg's
decode time); the coefficient blocks of a batch cross PCIe as one block and the GPU does the rest (csrc/jpeg.hip:
dequantisation, integer inverse DCT, fancy chroma upsampling, YCbCr -> RGB) into the same interleaved uint8 image Pillow
produces, byte for byte. Files this decoder leaves alone (progressive, arithmetic-coded, CMYK, 12-bit, unusual sampling) are
decoded by Pillow in the worker as before and travel as bytes in the same batch.

Nothing here touches the GPU at import or in a worker: libwitw_jpeg.so is host-only C++."""
import ctypes
import os

import numpy as np
import torch

from . import build as _build

_HOST = None
KIND_JPEG = 2          # packed-batch kind: coefficient blocks (+ raw uint8 images for the files left to Pillow)
DESC_COLS = 30         # per image: coefficient byte offset, quantisation-table byte offset, 22 info ints, is_raw, channels,
#                        entropy decoding on the device (1: column 0 is then the offset of the FILE BYTES), plan byte offset, file length,
#                        restart intervals
# Baseline / extended-sequential Huffman files are entropy-decoded ON THE DEVICE (round 6; csrc/jpeg.hip): with restart markers one GPU
# thread per restart interval (jpeg_huffman_kernel), without them a self-synchronising decode, one workgroup per file
# (jpeg_selfsync_kernel). The worker only parses the header and scans for markers; the file bytes cross PCIe instead of the
# coefficient blocks, and every launch covers all the files of a batch side (decode_packed_multi).
# Modes: 'all' (default) = every such file; disk -> embeddings with the bf16 encoders and FOUR loader workers: 16.1-16.4 k pairs/s on files
# with a restart marker every 1-2 MCUs, 14.9 k on ordinary files without markers -- 16 workers with host Huffman decoding reach 14.8 k on
# that box (11.5-15.0 k by box), the encoders alone 16.1-16.2 k; fp32 encoders 1,867 against 1,865 (profiles/r06_e2e_device_entropy_dev.json). 'restart' = only files with restart markers go to
# the device decoder, the others are Huffman-decoded in the worker; 'off' / False = host Huffman decoding for all.
# WITW_JPEG_DEVICE_ENTROPY = 0 | off | restart | 1 | all.
_mode = os.environ.get('WITW_JPEG_DEVICE_ENTROPY', 'all').lower()
DEVICE_ENTROPY = False if _mode in ('0', 'off', 'false') else 'restart' if _mode in ('restart', '1') else 'all'
CHECK_ERRORS = True    # decode_packed(host_buf=...) re-decodes files the device flagged as damaged with Pillow (as the host path does)
REPAIRED = [0]         # how many files that happened to
SELFSYNC_MIN_BLOCKS = 96   # a marker-less file of at most this many blocks is decoded by ONE thread of the interval kernel (a 1024-thread
#                            workgroup would find nothing to split)
SELFSYNC_WIDE_BYTES = 48 * 1024   # a launch of the self-synchronising kernel that holds a file this large runs 1024 threads per file, else 512
_ERRORS = []           # error flags (device int32 tensors) of the last batches decoded on the device: entropy_errors() sums them


def host_lib():
    global _HOST
    if _HOST is None:
        path = _build.build_host(verbose=False)
        lib = ctypes.CDLL(path)
        lib.witw_jpeg_info.restype = ctypes.c_int
        lib.witw_jpeg_info.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        lib.witw_jpeg_decode_coef.restype = ctypes.c_int
        lib.witw_jpeg_decode_coef.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
        lib.witw_jpeg_entropy_plan_bytes.restype = ctypes.c_longlong
        lib.witw_jpeg_entropy_plan_bytes.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        lib.witw_jpeg_entropy_plan.restype = ctypes.c_longlong
        lib.witw_jpeg_entropy_plan.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        _HOST = lib
    return _HOST


class JpegCoef(object):
    """The entropy-decoded form of one JPEG file: info (the 22 ints of witw_jpeg_info), coef int16 [blocks, 64] (natural
    order, components one after another), qt uint16 [components, 64]."""
    __slots__ = ('info', 'coef', 'qt')

    def __init__(self, info, coef, qt):
        self.info, self.coef, self.qt = info, coef, qt

    @property
    def shape(self):          # of the decoded image, HWC
        return (int(self.info[0]), int(self.info[1]), int(self.info[2]))


class JpegFile(JpegCoef):
    """A JPEG file whose header has been parsed (info) but whose entropy-coded data is still to be decoded: pack() decodes it
    straight into its place in the batch block (no array of its own, no copy). `coef` / `qt` decode on demand."""
    __slots__ = ('data',)

    def __init__(self, info, data):
        self.info, self.data = info, data

    def decode_into(self, coef, qt):
        """coef int16 [blocks, 64], qt uint16 [components, 64]: views of the destination. -> False if the entropy data is bad."""
        return host_lib().witw_jpeg_decode_coef(self.data.ctypes.data, self.data.size, coef.ctypes.data, qt.ctypes.data) == 0

    def entropy_plan(self):
        """What the DEVICE needs to entropy-decode this file itself (csrc_host/jpeg_coef.cpp witw_jpeg_entropy_plan: header fields,
        Huffman tables, the byte offset of every restart interval) -> (plan uint8 array, qt uint16 [components, 64]), or None for a
        file without restart markers (or one this scheme leaves to the host: more than two tables of a kind, markers out of
        sequence). A byte scan of the file, no Huffman decoding."""
        lib = host_lib()
        n = int(lib.witw_jpeg_entropy_plan_bytes(self.data.ctypes.data, self.data.size))
        if n <= 0:
            return None
        plan = np.empty((n,), dtype=np.uint8)
        qt = np.empty((int(self.info[2]), 64), dtype=np.uint16)
        if lib.witw_jpeg_entropy_plan(self.data.ctypes.data, self.data.size, plan.ctypes.data, n, qt.ctypes.data) != n:
            return None
        return plan, qt

    def pillow(self):
        """The host decoder's opinion of the file (what the reference's imread returns, model/cvig_fov.py:88-89): uint8 HWC.
        pack() takes it when the entropy data does not decode cleanly here -- libjpeg reads many damaged files with a warning."""
        import io
        from PIL import Image
        a = np.asarray(Image.open(io.BytesIO(self.data.tobytes())))
        return a[:, :, None] if a.ndim == 2 else a

    def _decoded(self):
        coef = np.empty((int(self.info[5]), 64), dtype=np.int16)
        qt = np.empty((int(self.info[2]), 64), dtype=np.uint16)
        if not self.decode_into(coef, qt):
            raise ValueError('corrupt JPEG entropy data')
        return coef, qt

    coef = property(lambda self: self._decoded()[0])
    qt = property(lambda self: self._decoded()[1])


def open_file(src):
    """src: a path or the bytes of a JPEG file -> JpegFile (header parsed, entropy decoding deferred to pack()), or None when the
    file is left to the host decoder."""
    data = np.fromfile(src, dtype=np.uint8) if isinstance(src, (str, os.PathLike)) else np.frombuffer(src, dtype=np.uint8)
    info = np.zeros(22, dtype=np.int32)
    if host_lib().witw_jpeg_info(data.ctypes.data, data.size, info.ctypes.data) != 0:
        return None
    return JpegFile(info, data)


def read_coef(src):
    """src: a path or the bytes of a JPEG file -> JpegCoef, or None when the file is left to the host decoder."""
    data = np.fromfile(src, dtype=np.uint8) if isinstance(src, (str, os.PathLike)) else np.frombuffer(src, dtype=np.uint8)
    lib = host_lib()
    info = np.zeros(22, dtype=np.int32)
    if lib.witw_jpeg_info(data.ctypes.data, data.size, info.ctypes.data) != 0:
        return None
    coef = np.empty((int(info[5]), 64), dtype=np.int16)
    qt = np.empty((int(info[2]), 64), dtype=np.uint16)
    if lib.witw_jpeg_decode_coef(data.ctypes.data, data.size, coef.ctypes.data, qt.ctypes.data) != 0:
        return None               # truncated / corrupt entropy data: let the host decoder say what it thinks of the file
    return JpegCoef(info, coef, qt)


def _shared_bytes(n):
    """n bytes of shared memory as a uint8 tensor: a DataLoader worker's batch block is built in place where the parent process
    will read it (torch would otherwise copy every tensor of a batch into shared memory when it pickles it)."""
    return torch.empty((n,), dtype=torch.uint8).share_memory_() if n else torch.empty((0,), dtype=torch.uint8)


def pack(images, shared=False, alloc=None):
    """A list of JpegCoef / JpegFile (and, for files left to Pillow, uint8 HWC arrays) -> (uint8 tensor, int64 [B, DESC_COLS],
    KIND_JPEG): one contiguous block for the pinned copy, images 128-byte aligned, the quantisation tables behind the coefficient
    data. JpegFile entries are entropy-decoded HERE, straight into the block. shared: allocate the block in shared memory;
    alloc(nbytes) -> (offset, uint8 numpy view) or None: build the block in caller-provided memory (ring.PinnedRing), in which
    case the first result is (offset, nbytes) instead of a tensor."""
    desc, off, spans = np.zeros((len(images), DESC_COLS), dtype=np.int64), 0, []
    plans = {}
    for i, a in enumerate(images):
        if DEVICE_ENTROPY and isinstance(a, JpegFile):
            pl = a.entropy_plan()
            # (a plan of ONE interval = a file without restart markers: the self-synchronising kernel, mode 'all' only)
            if pl is not None and (DEVICE_ENTROPY in ('all', True) or int(pl[0][4:8].view(np.int32)[0]) > 1):
                plans[i] = pl
        if i in plans:
            # entropy decoding on the device: the FILE BYTES travel (24 readable bytes behind the end: the kernel reads aligned 8-byte
            # words one ahead of the one it decodes)
            nbytes = int(a.data.size) + 24
            desc[i, 2:24] = a.info
            desc[i, 26], desc[i, 28] = 1, int(a.data.size)
            desc[i, 29] = int(plans[i][0][4:8].view(np.int32)[0])      # restart intervals of the file (grid sizing of the device decoder)
        elif isinstance(a, JpegCoef):
            nbytes = int(a.info[5]) * 128
            desc[i, 2:24] = a.info
        else:
            a = np.ascontiguousarray(a)
            if a.dtype != np.uint8 or a.ndim != 3:
                raise ValueError('jpeg.pack takes JpegCoef / JpegFile objects and uint8 HWC arrays')
            nbytes = a.size
            desc[i, 2], desc[i, 3], desc[i, 24], desc[i, 25] = a.shape[0], a.shape[1], 1, a.shape[2]
        desc[i, 0] = off
        spans.append((off, nbytes))
        # every entry starts on a 128-byte boundary: the device back end addresses coefficient data in 128-byte BLOCKS from the start
        # of the buffer (decode_tables: first block = offset // 128), so a raw image of a file left to Pillow in front of a JPEG
        # entry must not push it off that grid (round 4: it did -- 16-byte alignment -- and such a batch decoded its later JPEG
        # entries from up to 112 bytes too early)
        off += (nbytes + 127) // 128 * 128
    for i, a in enumerate(images):
        if isinstance(a, JpegCoef):
            desc[i, 1] = off
            off += (int(a.info[2]) * 128 + 15) // 16 * 16
    for i in plans:
        desc[i, 27] = off
        off += (plans[i][0].size + 15) // 16 * 16
    if alloc is not None:
        got = alloc(off)
        if got is None:
            return None
        t, buf = (got[0], off), got[1]
    else:
        t = _shared_bytes(off) if shared else torch.empty((off,), dtype=torch.uint8)
        buf = t.numpy()
    for i, a in enumerate(images):
        o, nbytes = spans[i]
        if i in plans:
            plan, qt = plans[i]
            buf[o:o + a.data.size] = a.data
            buf[o + a.data.size:o + nbytes] = 0
            buf[int(desc[i, 27]):int(desc[i, 27]) + plan.size] = plan
            buf[int(desc[i, 1]):int(desc[i, 1]) + qt.size * 2] = qt.reshape(-1).view(np.uint8)
        elif isinstance(a, JpegFile):
            coef = buf[o:o + nbytes].view(np.int16).reshape(-1, 64)
            qt = buf[int(desc[i, 1]):int(desc[i, 1]) + int(a.info[2]) * 128].view(np.uint16).reshape(-1, 64)
            if not a.decode_into(coef, qt):
                # damaged entropy data: the file goes to Pillow after all (as the reference and raw=True read it) and rides in the
                # same span as a raw uint8 image -- the coefficient blocks of an image are never smaller than its pixels
                px = np.ascontiguousarray(a.pillow())
                if px.dtype != np.uint8 or px.size > nbytes:
                    raise ValueError('image %d of the batch: JPEG entropy data is corrupt and the host decoder returned %s %s'
                                     % (i, px.dtype, px.shape))
                buf[o:o + px.size] = px.reshape(-1)
                desc[i, 2:24] = 0
                desc[i, 2], desc[i, 3], desc[i, 24], desc[i, 25] = px.shape[0], px.shape[1], 1, px.shape[2]
        elif isinstance(a, JpegCoef):
            buf[o:o + nbytes] = a.coef.reshape(-1).view(np.uint8)
            buf[int(desc[i, 1]):int(desc[i, 1]) + a.qt.size * 2] = a.qt.reshape(-1).view(np.uint8)
        else:
            buf[o:o + nbytes] = np.ascontiguousarray(a).reshape(-1)
    return t, torch.from_numpy(desc), KIND_JPEG


def decode_tables(d):
    """The launch tables of the device back end from the host descriptor rows of the JPEG entries of a batch (d: int64 [n, DESC_COLS],
    none of them raw) -> (planes int64 [P, 6] = {first coefficient block, quantisation table index, byte offset of the plane in the
    component buffer, blocks wide, blocks high, first block of the plane in the launch}, images int64 [n, 12] = {H, W, components,
    upsampling mode, luma plane offset, luma pitch, Cb offset, Cr offset, chroma pitch, chroma rows, chroma columns, output byte
    offset}, blocks in all, component bytes, output bytes, byte offset of the first quantisation table). Whole-array numpy: a
    per-image Python loop here cost 8 ms per 128 pairs, as much as the bf16 encoders take for them."""
    n = d.shape[0]
    if (d[:, 0] % 128).any():
        raise ValueError('jpeg: a coefficient entry does not start on a 128-byte boundary of the packed block')
    H, W, ncomp, hmax, vmax = (d[:, k] for k in range(2, 7))
    qt_base = int(d[:, 1].min())
    bw, bh = d[:, [10, 14, 18]], d[:, [11, 15, 19]]
    valid = np.arange(3)[None, :] < ncomp[:, None]
    nb = bw * bh * valid                                                  # blocks per (image, component)
    cb = (d[:, 0] // 128)[:, None] + np.cumsum(nb, axis=1) - nb             # coefficient blocks are 128 bytes
    flat = nb[valid]
    blk_off = np.cumsum(flat) - flat                                      # image-major, component-minor
    plane_first = np.zeros((n, 3), dtype=np.int64)
    plane_first[valid] = blk_off
    qti = ((d[:, 1] - qt_base) // 128)[:, None] + np.arange(3)[None, :]
    planes = np.stack([cb[valid], qti[valid], blk_off * 64, bw[valid], bh[valid], blk_off], axis=1).astype(np.int64)
    mode = np.where((hmax == 1) & (vmax == 1), 0, np.where((hmax == 2) & (vmax == 1), 1, np.where((hmax == 2) & (vmax == 2), 2, -1)))
    if (mode < 0).any():
        i = int(np.nonzero(mode < 0)[0][0])
        raise ValueError('jpeg: sampling %dx%d reached the device path' % (int(hmax[i]), int(vmax[i])))
    cw, ch = -(-W // hmax), -(-H // vmax)
    mode = mode + 2 * ((mode > 0) & (cw <= 2))      # libjpeg replicates chroma planes of at most two columns instead of filtering them
    images = np.zeros((n, 12), dtype=np.int64)
    images[:, 0], images[:, 1], images[:, 2], images[:, 3] = H, W, ncomp, mode
    images[:, 4], images[:, 5] = plane_first[:, 0] * 64, bw[:, 0] * 8
    c3 = ncomp == 3
    images[c3, 6], images[c3, 7], images[c3, 8] = plane_first[c3, 1] * 64, plane_first[c3, 2] * 64, bw[c3, 1] * 8
    images[c3, 9], images[c3, 10] = ch[c3], cw[c3]
    ob = (H * W * ncomp + 15) // 16 * 16
    images[:, 11] = np.cumsum(ob) - ob
    return planes, images, int(flat.sum()), int(flat.sum()) * 64, int(ob.sum()), qt_base


def _decode_group(dev, d, coef_ptr, qt_ptr):
    """dequantisation + inverse DCT + upsampling + colour conversion of the JPEG entries d (descriptor rows whose column 0 is the
    byte offset of their coefficient blocks from coef_ptr, column 1 that of their quantisation tables from qt_ptr) -> (tensors to keep
    alive, device address of every decoded image, components per image). Two launches."""
    from . import _lib, ops
    try:
        planes, images, blk, pbytes, obytes, qt_base = decode_tables(d)
    except ValueError as e:
        raise _lib.WitwError(str(e))
    plane_t = torch.from_numpy(planes).pin_memory().to(dev, non_blocking=True)
    image_t = torch.from_numpy(images).pin_memory().to(dev, non_blocking=True)
    comp = torch.empty((pbytes,), dtype=torch.uint8, device=dev)
    rgb = torch.empty((obytes,), dtype=torch.uint8, device=dev)
    lib = _lib.load()
    st = ops._stream()
    _lib.check(lib.witw_jpeg_idct(coef_ptr, qt_ptr + qt_base, plane_t.data_ptr(), planes.shape[0], blk, comp.data_ptr(), st), 'witw_jpeg_idct')
    _lib.check(lib.witw_jpeg_to_rgb(comp.data_ptr(), image_t.data_ptr(), int(d.shape[0]), int((images[:, 0] * images[:, 1]).max()),
                                    rgb.data_ptr(), st), 'witw_jpeg_to_rgb')
    return [plane_t, image_t, comp, rgb], rgb.data_ptr() + images[:, 11], images[:, 2]


def decode_packed_multi(parts, defer=False):
    """parts: [(dbuf, desc, host_buf or None), ...] = the packed blocks of ONE side of a batch, each on the GPU with its HOST descriptor
    table (a batch arrives in pieces when the loader hands out quarter batches: DevicePrefetcher(group=4)) -> (tensors to keep alive,
    int64 host table [sum B, 5] = {device address, H, W, 0, channels} of the decoded uint8 HWC images in part order: the descriptor
    rows of witw_resize_bilinear_normalize_batched / witw_polar_from_raw, kind 1, finish).
    EVERY LAUNCH COVERS THE FILES OF ALL PARTS (the descriptor rows carry addresses, so the blocks need not be one allocation): two
    launches for the entries whose coefficient blocks came from the host; for the entries that travelled as file bytes (DEVICE_ENTROPY)
    one or two more in front entropy-decode them into a coefficient buffer on the device -- a thread's chain of symbols bounds
    jpeg_huffman_kernel, so four launches over 32 files each take four times as long as one over 128.
    finish: None, or (defer=True, some part came with its host block) a callable that reads the damage flags of the device decoder --
    the one host wait of the call -- and lets Pillow re-decode flagged files, patching `table` in place; with defer=False that has
    happened on return."""
    from . import _lib, ops
    dev = parts[0][0].device
    ref = min(int(p[0].data_ptr()) for p in parts)      # offsets of every part are re-based to the lowest block (allocations are 512-byte aligned)
    ds, part_of, off0 = [], [], []
    for k, (dbuf, desc, _hb) in enumerate(parts):
        dk = np.array(desc.numpy() if isinstance(desc, torch.Tensor) else desc, dtype=np.int64, copy=True)
        delta = int(dbuf.data_ptr()) - ref
        off0.append(dk[:, 0].copy())
        dk[:, 0] += delta
        dk[:, 1] += delta
        if dk.shape[1] > 27:
            dk[:, 27] += delta
        ds.append(dk)
        part_of.append(np.full((dk.shape[0],), k, dtype=np.int64))
    d, part_of, off0 = np.concatenate(ds), np.concatenate(part_of), np.concatenate(off0)
    B = d.shape[0]
    table = np.zeros((B, 5), dtype=np.int64)
    keep = []
    finish = None
    is_raw = d[:, 24] != 0
    on_dev = (d[:, 26] != 0) & ~is_raw if d.shape[1] > 26 else np.zeros((B,), dtype=bool)
    table[:, 1], table[:, 2] = d[:, 2], d[:, 3]
    table[is_raw, 0] = ref + d[is_raw, 0]
    table[is_raw, 4] = d[is_raw, 25]
    jp = np.nonzero(~is_raw & ~on_dev)[0]
    if jp.size:
        k, addr, ncomp = _decode_group(dev, d[jp], ref, ref)
        keep += k
        table[jp, 0], table[jp, 4] = addr, ncomp
    dv = np.nonzero(on_dev)[0]
    if dv.size:
        blocks = d[dv, 7]                                        # info[5]: coefficient blocks of the file
        first = np.cumsum(blocks) - blocks
        coef = torch.zeros((int(blocks.sum()) * 64,), dtype=torch.int16, device=dev)
        errors = torch.zeros((int(dv.size),), dtype=torch.int32, device=dev)
        # files with restart markers: one thread per interval; files without (ONE interval of more than SELFSYNC_MIN_BLOCKS blocks):
        # the self-synchronising kernel, one workgroup per file
        sync = (d[dv, 29] == 1) & (blocks > SELFSYNC_MIN_BLOCKS)
        rows = np.stack([ref + d[dv, 0], ref + d[dv, 27], coef.data_ptr() + first * 128, d[dv, 28]], axis=1).astype(np.int64)
        ir = np.nonzero(~sync)[0]
        if ir.size:
            files_t = torch.from_numpy(np.ascontiguousarray(rows[ir])).pin_memory().to(dev, non_blocking=True)
            e_r = errors if ir.size == dv.size else torch.zeros((int(ir.size),), dtype=torch.int32, device=dev)
            _lib.check(_lib.load().witw_jpeg_huffman(files_t.data_ptr(), int(ir.size), int(d[dv[ir], 29].max()), e_r.data_ptr(), ops._stream()),
                       'witw_jpeg_huffman')
            if e_r is not errors:
                errors[torch.from_numpy(ir).to(dev)] = e_r
            keep += [files_t, e_r]
        isy = np.nonzero(sync)[0]
        if isy.size:
            sizes = (d[dv[isy], 28] + 32 + 7) // 8 * 8
            soff = np.cumsum(sizes) - sizes
            scratch = torch.empty((int(sizes.sum()),), dtype=torch.uint8, device=dev)
            srows = np.concatenate([rows[isy], (scratch.data_ptr() + soff)[:, None], np.zeros((isy.size, 1), np.int64)], axis=1).astype(np.int64)
            files_s = torch.from_numpy(np.ascontiguousarray(srows)).pin_memory().to(dev, non_blocking=True)
            e_s = errors if isy.size == dv.size else torch.zeros((int(isy.size),), dtype=torch.int32, device=dev)
            # threads per file: 1024 when the launch holds a large file (shorter subsequences: 113 KB files 1.3 against 1.4 ms per 128 at
            # 512 threads), 512 otherwise (22 KB files: 0.76 against 0.89 ms -- more rounds than the shorter subsequences save)
            env_t = os.environ.get('WITW_SELFSYNC_THREADS')
            threads = (1024 if int(env_t) >= 1024 else 512 if int(env_t) >= 512 else 256) if env_t else \
                (1024 if int(d[dv[isy], 28].max()) >= SELFSYNC_WIDE_BYTES else 512)
            _lib.check(_lib.load().witw_jpeg_huffman_selfsync_threads(files_s.data_ptr(), int(isy.size), threads, e_s.data_ptr(), ops._stream()),
                       'witw_jpeg_huffman_selfsync_threads')
            if e_s is not errors:
                errors[torch.from_numpy(isy).to(dev)] = e_s
            keep += [files_s, e_s, scratch]
        dd = d[dv].copy()
        dd[:, 0] = first * 128                                   # where each file's coefficient blocks sit in `coef`
        k, addr, ncomp = _decode_group(dev, dd, coef.data_ptr(), ref)
        keep += k + [coef, errors]
        table[dv, 0], table[dv, 4] = addr, ncomp
        _ERRORS.append(errors)
        del _ERRORS[:-64]
        if CHECK_ERRORS and any(p[2] is not None for p in parts):
            # A file whose entropy-coded data is damaged is read by the reference through libjpeg, which recovers what it can (with
            # a warning); the host path hands such a file to Pillow inside pack(). Here the damage shows only once the kernel has
            # run: look at the flags (a 4-byte-per-file copy; it waits for THIS stream's staging work, which the drivers run on the
            # copy stream one batch ahead of the encoders) and let Pillow decode the flagged files from the bytes that are still in
            # the host block -- the same image, by the same decoder, as on the host path.
            flags_host = torch.empty((int(dv.size),), dtype=torch.int32).pin_memory()
            flags_host.copy_(errors, non_blocking=True)
            flags_ready = torch.cuda.Event()
            flags_ready.record(torch.cuda.current_stream(dev))

            def finish():
                flags_ready.synchronize()
                for j in np.nonzero(flags_host.numpy())[0]:
                    i = int(dv[j])
                    hb = parts[int(part_of[i])][2]
                    if hb is None:
                        continue
                    hb = hb.numpy() if isinstance(hb, torch.Tensor) else np.asarray(hb)
                    px = np.array(JpegFile(None, hb[int(off0[i]):int(off0[i]) + int(d[i, 28])]).pillow())      # a writable copy (from_numpy warns on Pillow's read-only view)
                    t = torch.from_numpy(px).to(dev)
                    keep.append(t)
                    table[i] = (t.data_ptr(), px.shape[0], px.shape[1], 0, px.shape[2])
                    REPAIRED[0] += 1
    if finish is not None and not defer:
        finish()
        finish = None
    return keep, torch.from_numpy(table), finish


def decode_packed(dbuf, desc, host_buf=None):
    """One packed block: decode_packed_multi([(dbuf, desc, host_buf)]) -> (tensors to keep alive, table); damaged files repaired on
    return."""
    keep, table, _f = decode_packed_multi([(dbuf, desc, host_buf)])
    return keep, table


def entropy_errors():
    """Files of the last (up to 64) device-entropy-decoded batches whose entropy-coded data was damaged (the host path hands such a
    file to Pillow; on the device it decodes with zeros from the damage on and is counted here). Synchronises."""
    n = sum(int((e != 0).sum().item()) for e in _ERRORS)
    return n


def decode(images, device):
    """Decode a list of JpegCoef (or uint8 HWC arrays, passed through) on `device` -> list of uint8 HWC tensors (tests and
    one-off use; the data path goes through pack / decode_packed inside GpuPreprocess)."""
    buf, desc, _k = pack(images)
    dbuf = buf.to(device)
    keep, table = decode_packed(dbuf, desc)
    torch.cuda.synchronize(device)
    out = []
    for i in range(len(images)):
        H, W, C = int(table[i, 1]), int(table[i, 2]), int(table[i, 4])
        src = next(t for t in [dbuf] + keep if t.dtype == torch.uint8 and t.data_ptr() <= int(table[i, 0]) < t.data_ptr() + max(1, t.numel())
                   and int(table[i, 0]) + H * W * C <= t.data_ptr() + t.numel() and (t is dbuf) == bool(desc[i, 24]))
        o = int(table[i, 0]) - src.data_ptr()
        out.append(src[o:o + H * W * C].reshape(H, W, C).clone())
    return out
