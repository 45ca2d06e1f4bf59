"""The JPEG back end on the GPU in isolation: 128 overhead (512 x 512) + 128 ground (224 x 224) files -> decode_packed, us per batch."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    import torch
    from witw_amd import e2e, jpeg
    dev = torch.device('cuda:0')
    with tempfile.TemporaryDirectory() as root:
        for i in range(128):                      # written in this process (no worker pool: this file is a script, not a module)
            e2e._write_pair((root, i, 77))
        for side in ('ov', 'su'):
            files = [jpeg.open_file(os.path.join(root, '%s_%05d.jpg' % (side, i))) for i in range(128)]
            buf, desc, _k = jpeg.pack(files)
            dbuf = buf.to(dev)
            for _ in range(3):
                jpeg.decode_packed(dbuf, desc)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                keep, table = jpeg.decode_packed(dbuf, desc)
            e1.record()
            torch.cuda.synchronize()
            print('%s: %.1f us per 128 images (idct + upsample/colour, 2 launches)' % (side, e0.elapsed_time(e1) / 20 * 1e3), flush=True)


if __name__ == '__main__':
    main()
