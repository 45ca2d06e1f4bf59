"""Host-side wrappers: torch tensors in, C-ABI calls (pointers + sizes + HIP stream) out.

torch is used for device memory and the current stream only; all arithmetic on the hot path
runs in libwitw_hip.so. Every wrapper refuses CPU tensors — there is no CPU fallback.
"""
import torch

from . import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev_f32(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise _lib.WitwError('%s must be a CUDA(HIP) tensor: the WITW hot path has no CPU fallback' % name)
    if t.dtype != torch.float32:
        raise _lib.WitwError('%s must be float32, got %s' % (name, t.dtype))
    if not t.is_contiguous():
        raise _lib.WitwError('%s must be contiguous' % name)
    return t


def _p(t):
    return 0 if t is None else t.data_ptr()


class PackedConv:
    """Weights of one 3x3 conv packed for the MFMA kernel ([n_tile][cin/8][tap][quad][TN][4])
    plus the zero-padded bias. `transpose_flip` packs the dgrad filter of the same weights."""

    def __init__(self, weight, bias, transpose_flip=False):
        lib = _lib.load()
        w = _dev_f32(weight.detach(), 'weight')
        if transpose_flip:
            cin, cout = w.shape[0], w.shape[1]   # packed filter maps grad_out (w.shape[0]) -> grad_in
        else:
            cout, cin = w.shape[0], w.shape[1]
        self.cout, self.cin = cout, cin
        self.cin_pad = (cin + 7) // 8 * 8
        n = lib.witw_conv3x3_packed_floats(cout, cin)
        self.wpk = torch.empty(n, dtype=torch.float32, device=w.device)
        _lib.check(lib.witw_conv3x3_pack_weights(w.data_ptr(), self.wpk.data_ptr(), cout, cin, int(transpose_flip),
                                                 _stream()), 'witw_conv3x3_pack_weights')
        nb = lib.witw_conv3x3_bias_floats(cout)
        self.bias = torch.zeros(nb, dtype=torch.float32, device=w.device)
        if bias is not None and not transpose_flip:
            self.bias[:cout].copy_(bias.detach())


def nchw_to_nhwc8(x):
    lib = _lib.load()
    x = _dev_f32(x, 'x')
    B, C, H, W = x.shape
    y = torch.empty((B, H, W, 8), dtype=torch.float32, device=x.device)
    _lib.check(lib.witw_nchw_to_nhwc8(x.data_ptr(), y.data_ptr(), B, C, H, W, _stream()), 'witw_nchw_to_nhwc8')
    return y


def conv3x3_fwd(x_nhwc, packed, stride_h=1, circular=False, relu=True, pool=False, out_nchw=False, drop_scale=None):
    """x_nhwc [B,H,W,Cin_pad] -> NHWC [B,Hy,Wy,Cout] (or NCHW [B,Cout,Hy,Wy])."""
    lib = _lib.load()
    x = _dev_f32(x_nhwc, 'x')
    B, H, W, C = x.shape
    if C != packed.cin_pad:
        raise _lib.WitwError('conv3x3_fwd: input has %d channels, packed weights expect %d' % (C, packed.cin_pad))
    Ho = (H + 2 - 3) // stride_h + 1
    Hy, Wy = (Ho // 2, W // 2) if pool else (Ho, W)
    shape = (B, packed.cout, Hy, Wy) if out_nchw else (B, Hy, Wy, packed.cout)
    y = torch.empty(shape, dtype=torch.float32, device=x.device)
    if drop_scale is not None:
        drop_scale = _dev_f32(drop_scale, 'drop_scale')
        if tuple(drop_scale.shape) != (B, packed.cout):
            raise _lib.WitwError('drop_scale must be [B,Cout]')
    _lib.check(lib.witw_conv3x3_fwd(x.data_ptr(), packed.wpk.data_ptr(), packed.bias.data_ptr(), _p(drop_scale),
                                    y.data_ptr(), B, H, W, C, packed.cout, stride_h, int(circular), int(relu),
                                    int(pool), int(out_nchw), _stream()), 'witw_conv3x3_fwd')
    return y
