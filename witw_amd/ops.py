"""Host-side wrappers: torch tensors in, C-ABI calls (pointers + sizes + HIP stream) out.

torch is used for device memory and the current stream only; all arithmetic on the hot path
runs in libwitw_hip.so. Every wrapper refuses CPU tensors — there is no CPU fallback.
"""
import torch

from . import _lib


# When set to a list, every conv3x3 launch is bracketed by HIP events on the launch stream and
# (variant, algorithmic FLOPs, start, end) is appended (bench.py's live roofline measurement).
PROFILE = None
PROFILE_BY_KERNEL = None      # bench.py batch sweep: {kernel instantiation (witw_last_kernel_variant): [(flop, start event, end event)]}


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
    plus the zero-padded bias. `transpose_flip` packs the dgrad filter of the same weights. `taps4` packs only the
    2x2 sub-window that is live in a filter whose first tap row/column are zero (cvig_baseline's 4x4/s2 convs over the
    space-to-depth image; with transpose_flip: the dgrad filter, whose LAST row/column are zero) for the 4-tap kernels."""

    def __init__(self, weight, bias, transpose_flip=False, taps4=False, reuse=None):
        """reuse: a PackedConv of the same layer and mode whose device buffers are overwritten in place (re-packing
        after an optimizer step without allocating or zero-filling; stream order keeps earlier launches safe)."""
        lib = _lib.load()
        w = _dev_f32(weight.detach(), 'weight')
        if transpose_flip:
            cin, cout = w.shape[0], w.shape[1]   # packed filter maps grad_out (w.shape[0]) -> grad_in
        else:
            cout, cin = w.shape[0], w.shape[1]
        self.cout, self.cin = cout, cin
        self.cin_pad = (cin + 7) // 8 * 8
        self.taps4 = bool(taps4)
        self.tap_base = 0 if transpose_flip else 1
        n_pk = lib.witw_conv3x3_packed_floats_taps4(cout, cin) if taps4 else lib.witw_conv3x3_packed_floats(cout, cin)
        if reuse is not None and not (reuse.wpk.numel() == n_pk and reuse.wpk.device == w.device and reuse.cout == cout
                                      and reuse.taps4 == self.taps4):
            reuse = None
        if taps4:
            self.wpk = reuse.wpk if reuse is not None else torch.empty(n_pk, dtype=torch.float32, device=w.device)
            _lib.check(lib.witw_conv3x3_pack_weights_taps4(w.data_ptr(), self.wpk.data_ptr(), cout, cin, int(transpose_flip),
                                                           _stream()), 'witw_conv3x3_pack_weights_taps4')
        else:
            self.wpk = reuse.wpk if reuse is not None else torch.empty(n_pk, dtype=torch.float32, device=w.device)
            _lib.check(lib.witw_conv3x3_pack_weights(w.data_ptr(), self.wpk.data_ptr(), cout, cin, int(transpose_flip),
                                                     _stream()), 'witw_conv3x3_pack_weights')
        nb = lib.witw_conv3x3_bias_floats(cout)
        self.bias = reuse.bias if reuse is not None else torch.zeros(nb, dtype=torch.float32, device=w.device)
        if bias is not None and not transpose_flip:
            self.bias[:cout].copy_(bias.detach())


class PackedFirstConv:
    """Filter of the first layer (C<=4 -> 64) packed for witw_conv3x3_first_fwd; bf16=True rounds it to bf16."""

    def __init__(self, weight, bias, bf16=False):
        lib = _lib.load()
        w = _dev_f32(weight.detach(), 'weight')
        if w.shape[0] != 64 or w.shape[1] > (8 if bf16 else 4):
            raise _lib.WitwError('PackedFirstConv: expects a [64, C<=4, 3, 3] filter (C<=8 with bf16=True)')
        self.cin, self.bf16 = w.shape[1], bool(bf16)
        self.wf = torch.empty(2560, dtype=torch.float32, device=w.device)
        _lib.check(lib.witw_conv3x3_first_pack(w.data_ptr(), self.wf.data_ptr(), self.cin, int(self.bf16), _stream()),
                   'witw_conv3x3_first_pack')
        self.bias = bias.detach().to(torch.float32).contiguous().clone()


def conv3x3_first_fwd(x_nchw, packed, circular=False, relu=True, split_f16=False):
    """x NCHW fp32 [B,C<=4,H,W] -> NHWC [B,H,W,64] (fp32, or bf16 when packed.bf16, or split-fp16 [B,H,W,8,2,8] with
    split_f16 and an fp32-packed filter)."""
    lib = _lib.load()
    x = _dev_f32(x_nchw, 'x')
    B, C, H, W = x.shape
    if C != packed.cin:
        raise _lib.WitwError('conv3x3_first_fwd: input has %d channels, filter expects %d' % (C, packed.cin))
    if split_f16 and packed.bf16:
        raise _lib.WitwError('conv3x3_first_fwd: split_f16 needs a filter packed with bf16=False')
    if split_f16:
        y = torch.empty((B, H, W, 8, 2, 8), dtype=torch.float16, device=x.device)
    else:
        y = torch.empty((B, H, W, 64), dtype=torch.bfloat16 if packed.bf16 else torch.float32, device=x.device)
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(lib.witw_conv3x3_first_fwd(x.data_ptr(), packed.wf.data_ptr(), packed.bias.data_ptr(), y.data_ptr(), B, C, H, W,
                                          int(circular), int(relu), 2 if split_f16 else int(packed.bf16), _stream()),
               'witw_conv3x3_first_fwd')
    if prof is not None:
        e1.record()
        prof.append((('first', packed.bf16), 2.0 * C * 64 * 9 * H * W * B, e0, e1))
    return y


def conv_first2_bf16(x_nchw, packed_first, packed_second, circular=False):
    """Layers 0 and 2 fused (bf16 inference): x NCHW fp32 [B,C<=8,H,W] -> NHWC bf16 [B,H/2,W/2,64]; packed_first =
    PackedFirstConv(bf16=True), packed_second = PackedConvBf16 of the 64 -> 64 conv."""
    lib = _lib.load()
    x = _dev_f32(x_nchw, 'x')
    B, C, H, W = x.shape
    if not packed_first.bf16 or C != packed_first.cin or packed_second.cin != 64 or packed_second.cout != 64:
        raise _lib.WitwError('conv_first2_bf16: needs a bf16-packed C -> 64 first filter and a 64 -> 64 second one')
    y = torch.empty((B, H // 2, W // 2, 64), dtype=torch.bfloat16, device=x.device)
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(lib.witw_conv_first2_bf16_fwd(x.data_ptr(), packed_first.wf.data_ptr(), packed_first.bias.data_ptr(),
                                             packed_second.wpk.data_ptr(), packed_second.bias.data_ptr(), y.data_ptr(), B, C, H, W,
                                             int(circular), _stream()), 'witw_conv_first2_bf16_fwd')
    if prof is not None:
        e1.record()
        prof.append((('first2', True), 2.0 * (C + 64) * 64 * 9 * H * W * B, e0, e1))
        if PROFILE_BY_KERNEL is not None:
            PROFILE_BY_KERNEL.setdefault('conv_first2_bf16_kernel', []).append((prof[-1][1], e0, e1))
    return y


def conv_first2_bf16_train(x_nchw, packed_first, packed_second, circular=False):
    """conv_first2_bf16 for a training step whose backward crosses both layers (cvig_semantic: layer 0 trains): -> (y NHWC bf16
    [B,H/2,W/2,64], pool codes uint8 of y's shape, gate_bits uint8 [B,H,W,8]: layer 0's ReLU gate, bit c & 7 of byte c >> 3 = channel
    c of that pixel is > 0). Neither 64-channel activation is written: the backward routes the pooled gradient with the codes
    (maxpool2x2_bwd_bf16), gates layer 2 with y > 0 and layer 0 with the bits (conv3x3_bf16_dgrad_gatebits)."""
    lib = _lib.load()
    x = _dev_f32(x_nchw, 'x')
    B, C, H, W = x.shape
    if not packed_first.bf16 or C != packed_first.cin or packed_second.cin != 64 or packed_second.cout != 64:
        raise _lib.WitwError('conv_first2_bf16_train: needs a bf16-packed C -> 64 first filter and a 64 -> 64 second one')
    if H % 2 or W % 2:
        raise _lib.WitwError('conv_first2_bf16_train: H and W must be even, got %d x %d' % (H, W))
    y = torch.empty((B, H // 2, W // 2, 64), dtype=torch.bfloat16, device=x.device)
    code = torch.empty((B, H // 2, W // 2, 64), dtype=torch.uint8, device=x.device)
    bits = torch.empty((B, H, W, 8), dtype=torch.uint8, device=x.device)
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(lib.witw_conv_first2_bf16_fwd_train(x.data_ptr(), packed_first.wf.data_ptr(), packed_first.bias.data_ptr(),
                                                   packed_second.wpk.data_ptr(), packed_second.bias.data_ptr(), y.data_ptr(),
                                                   code.data_ptr(), bits.data_ptr(), B, C, H, W, int(circular), _stream()),
               'witw_conv_first2_bf16_fwd_train')
    if prof is not None:
        e1.record()
        prof.append((('first2', True), 2.0 * (C + 64) * 64 * 9 * H * W * B, e0, e1))
        if PROFILE_BY_KERNEL is not None:
            PROFILE_BY_KERNEL.setdefault('conv_first2_bf16_kernel<train>', []).append((prof[-1][1], e0, e1))
    return y, code, bits


def gatebits_dgrad_ok(B, H, W, cin, cout):
    """can conv3x3_bf16_dgrad_gatebits run this shape? (the weight-resident kernel must apply and be switched on)"""
    return bool(_lib.load().witw_conv3x3_bf16_gatebits_ok(B, H, W, cin, cout))


def conv3x3_bf16_dgrad_gatebits(dz_nhwc, packed_t, gate_bits, circular=False):
    """Data gradient of a 64-input-channel stride-1 layer with the ReLU gate of the layer in front as one bit per output
    (conv_first2_bf16_train's gate_bits): dz NHWC bf16 [B,H,W,64 (padded cout of the layer)] -> NHWC bf16 [B,H,W,Cout_t], zero
    where the bit is clear. Same bits as conv3x3_bf16_fwd(dz, packed_t, relu=False, gate=<the bf16 activation>)."""
    lib = _lib.load()
    if not (dz_nhwc.is_cuda and dz_nhwc.dtype == torch.bfloat16 and dz_nhwc.is_contiguous() and dz_nhwc.dim() == 4):
        raise _lib.WitwError('conv3x3_bf16_dgrad_gatebits: dz must be a contiguous bfloat16 NHWC GPU tensor')
    B, H, W, C = dz_nhwc.shape
    if C != packed_t.cin_pad:
        raise _lib.WitwError('conv3x3_bf16_dgrad_gatebits: dz has %d channels, the packed filter expects %d' % (C, packed_t.cin_pad))
    if not (gate_bits.is_cuda and gate_bits.dtype == torch.uint8 and gate_bits.is_contiguous()
            and tuple(gate_bits.shape) == (B, H, W, packed_t.cout // 8)):
        raise _lib.WitwError('conv3x3_bf16_dgrad_gatebits: gate_bits must be uint8 [%d,%d,%d,%d]' % (B, H, W, packed_t.cout // 8))
    y = torch.empty((B, H, W, packed_t.cout), dtype=torch.bfloat16, device=dz_nhwc.device)
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(lib.witw_conv3x3_bf16_fwd_gatebits(dz_nhwc.data_ptr(), packed_t.wpk.data_ptr(), packed_t.bias.data_ptr(), gate_bits.data_ptr(),
                                                  y.data_ptr(), B, H, W, C, packed_t.cout, int(circular), 0, _stream()),
               'witw_conv3x3_bf16_fwd_gatebits')
    if prof is not None:
        e1.record()
        prof.append((('bf16_wres', lib.witw_conv3x3_tile_n(packed_t.cout), 1, False), 2.0 * packed_t.cin * packed_t.cout * 9 * H * W * B, e0, e1))
        if PROFILE_BY_KERNEL is not None:
            PROFILE_BY_KERNEL.setdefault(last_kernel_variant(), []).append((prof[-1][1], e0, e1))
    return y


def nchw_to_nhwc8(x):
    lib = _lib.load()
    x = _dev_f32(x, 'x')
    B, C, H, W = x.shape
    y = torch.empty((B, H, W, 8), dtype=torch.float32, device=x.device)
    _lib.check(lib.witw_nchw_to_nhwc8(x.data_ptr(), y.data_ptr(), B, C, H, W, _stream()), 'witw_nchw_to_nhwc8')
    return y


def nchw_to_nhwc(x, cpad):
    lib = _lib.load()
    x = _dev_f32(x, 'x')
    B, C, H, W = x.shape
    y = torch.empty((B, H, W, cpad), dtype=torch.float32, device=x.device)
    _lib.check(lib.witw_nchw_to_nhwc(x.data_ptr(), y.data_ptr(), B, C, H, W, cpad, _stream()), 'witw_nchw_to_nhwc')
    return y


def conv3x3_fwd(x_nhwc, packed, stride_h=1, circular=False, relu=True, pool=False, out_nchw=False, drop_scale=None,
                gate=None, dilate_h=False, out_h=None, lrelu_slope=None, post_scale=None, post_shift=None,
                want_pool_code=False):
    """x_nhwc [B,H,W,Cin_pad] -> NHWC [B,Hy,Wy,Cout] (or NCHW [B,Cout,Hy,Wy])."""
    lib = _lib.load()
    x = _dev_f32(x_nhwc, 'x')
    B, H, W, C = x.shape
    if C != packed.cin_pad:
        raise _lib.WitwError('conv3x3_fwd: input has %d channels, packed weights expect %d' % (C, packed.cin_pad))
    h_phys = H
    if dilate_h:
        if out_h is None or (out_h - 1) // 2 + 1 != H:
            raise _lib.WitwError('conv3x3_fwd: dilate_h needs out_h with (out_h-1)//2+1 == %d physical rows' % H)
        H = out_h          # logical (zero-interleaved) height
    Ho = (H + 2 - 3) // stride_h + 1
    # algorithmic rows of the launch's FLOP count: a zero-interleaved launch is the data gradient of a stride-(2,1) conv, whose
    # multiply-adds are those of that conv's forward (one per real input row), not one per interleaved row
    flop_rows = h_phys if dilate_h else Ho
    Hy, Wy = (Ho // 2, W // 2) if pool else (Ho, W)
    shape = (B, packed.cout, Hy, Wy) if out_nchw else (B, Hy, Wy, packed.cout)
    y = torch.empty(shape, dtype=torch.float32, device=x.device)
    if drop_scale is not None:
        drop_scale = _dev_f32(drop_scale, 'drop_scale')
        if tuple(drop_scale.shape) != (B, packed.cout):
            raise _lib.WitwError('drop_scale must be [B,Cout]')
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if gate is not None:
        gate = _dev_f32(gate, 'gate')
        if tuple(gate.shape) != shape:
            raise _lib.WitwError('gate must have the output shape %s, got %s' % (shape, tuple(gate.shape)))
    act = 2 if lrelu_slope is not None else int(bool(relu))
    if post_scale is not None:
        post_scale, post_shift = _dev_f32(post_scale, 'post_scale'), _dev_f32(post_shift, 'post_shift')
        if post_scale.numel() != packed.cout or post_shift.numel() != packed.cout:
            raise _lib.WitwError('post_scale/post_shift must have Cout entries')
    code = torch.empty(shape, dtype=torch.uint8, device=x.device) if (pool and want_pool_code) else None
    if getattr(packed, 'taps4', False):
        if stride_h != 1 or circular or pool or out_nchw or dilate_h or drop_scale is not None:
            raise _lib.WitwError('conv3x3_fwd: a taps4-packed filter runs stride 1, zero padding, NHWC out, no pool / dropout')
        _lib.check(lib.witw_conv3x3_fwd_taps4(x.data_ptr(), packed.wpk.data_ptr(), packed.bias.data_ptr(), _p(gate),
                                              _p(post_scale), _p(post_shift), y.data_ptr(), B, H, W, C, packed.cout, act,
                                              float(lrelu_slope or 0.), packed.tap_base, _stream()), 'witw_conv3x3_fwd_taps4')
    else:
        _lib.check(lib.witw_conv3x3_fwd_ex(x.data_ptr(), packed.wpk.data_ptr(), packed.bias.data_ptr(), _p(drop_scale),
                                           _p(gate), _p(post_scale), _p(post_shift), y.data_ptr(), _p(code), B, H, W, C,
                                           packed.cout, stride_h, int(circular), act, float(lrelu_slope or 0.), int(pool),
                                           int(out_nchw), int(bool(dilate_h)), _stream()), 'witw_conv3x3_fwd_ex')
    if prof is not None:
        e1.record()
        variant = (lib.witw_conv3x3_tile_n(packed.cout), stride_h, bool(pool),
                   lib.witw_conv3x3_workgroup_waves(B, H, W, packed.cout, stride_h))
        if dilate_h:      # the zero-row-skipping instantiation (GEO = 2) is a launch class of its own: credited its real input rows
            variant = variant + ('dil',)
        prof.append((variant, 2.0 * packed.cin * packed.cout * (4 if getattr(packed, 'taps4', False) else 9) * flop_rows * W * B, e0, e1))
        if PROFILE_BY_KERNEL is not None:
            PROFILE_BY_KERNEL.setdefault(last_kernel_variant(), []).append((prof[-1][1], e0, e1))
    if want_pool_code:
        return y, code
    return y


def _taps4_args(x_nhwc, packed, lrelu_slope, post_scale, post_shift):
    x = _dev_f32(x_nhwc, 'x')
    if not getattr(packed, 'taps4', False):
        raise _lib.WitwError('a taps4-packed filter is required')
    if x.shape[3] != packed.cin_pad:
        raise _lib.WitwError('conv taps4: input has %d channels, packed weights expect %d' % (x.shape[3], packed.cin_pad))
    if post_scale is not None:
        post_scale, post_shift = _dev_f32(post_scale, 'post_scale'), _dev_f32(post_shift, 'post_shift')
        if post_scale.numel() != packed.cout or post_shift.numel() != packed.cout:
            raise _lib.WitwError('post_scale/post_shift must have Cout entries')
    act = 2 if lrelu_slope is not None else 0
    return x, act, post_scale, post_shift


def conv_taps4_s2d(x_nhwc, packed, valid_hw, lrelu_slope=None, post_scale=None, post_shift=None):
    """2x2-tap convolution whose epilogue writes the space-to-depth(2) image of its (vh, vw) valid outputs, zeros elsewhere:
    x [B,H,W,Cin_pad] -> [B, ceil(vh/2), ceil(vw/2), 4*Cout], the input layout of the next Conv2d(k=4, s=2) block
    (model/cvig_baseline.py:236-252) without a separate pass over the activation."""
    lib = _lib.load()
    x, act, post_scale, post_shift = _taps4_args(x_nhwc, packed, lrelu_slope, post_scale, post_shift)
    B, H, W, C = x.shape
    vh, vw = valid_hw
    y = torch.empty((B, (vh + 1) // 2, (vw + 1) // 2, 4 * packed.cout), dtype=torch.float32, device=x.device)
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(lib.witw_conv3x3_fwd_taps4_ex(x.data_ptr(), packed.wpk.data_ptr(), packed.bias.data_ptr(), None, _p(post_scale),
                                             _p(post_shift), y.data_ptr(), B, H, W, C, packed.cout, act, float(lrelu_slope or 0.),
                                             packed.tap_base, 1, 1, vh, vw, _stream()), 'witw_conv3x3_fwd_taps4_ex')
    if prof is not None:
        e1.record()
        prof.append(((lib.witw_conv3x3_tile_n(packed.cout), 1, False, lib.witw_conv3x3_workgroup_waves(B, H, W, packed.cout, 1)),
                     2.0 * packed.cin * packed.cout * 4 * vh * vw * B, e0, e1))
    return y


def conv4x4s2_first(x_nchw, weight, bias, post_scale=None, post_shift=None, normalize=True, lrelu_slope=0.2):
    """First block of a cvig_baseline encoder in one launch: x NCHW fp32 [B,C,H,W] (raw 0..255 values when normalize) ->
    [B, ceil(vh/2), ceil(vw/2), 256], the space-to-depth(2) input of the second block (model/cvig_baseline.py:236-240, 265-268).
    weight [64,C,4,4] (torch layout), bias / post_scale / post_shift [64]."""
    lib = _lib.load()
    x = _dev_f32(x_nchw, 'x')
    w, b = _dev_f32(weight.detach(), 'weight'), _dev_f32(bias.detach(), 'bias')
    B, C, H, W = x.shape
    if tuple(w.shape) != (64, C, 4, 4) or tuple(b.shape) != (64,):
        raise _lib.WitwError('conv4x4s2_first: weight %s / bias %s do not fit a %d-channel input' % (tuple(w.shape), tuple(b.shape), C))
    if (post_scale is None) != (post_shift is None):
        raise _lib.WitwError('conv4x4s2_first: post_scale and post_shift come together')
    if post_scale is not None:
        post_scale, post_shift = _dev_f32(post_scale, 'post_scale'), _dev_f32(post_shift, 'post_shift')
    vh, vw = (H - 4) // 2 + 1, (W - 4) // 2 + 1
    y = torch.empty((B, (vh + 1) // 2, (vw + 1) // 2, 256), dtype=torch.float32, device=x.device)
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(lib.witw_conv4x4s2_first_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), _p(post_scale), _p(post_shift), y.data_ptr(),
                                            B, C, H, W, int(bool(normalize)), float(lrelu_slope), _stream()), 'witw_conv4x4s2_first_fwd')
    if prof is not None:
        e1.record()
        prof.append((('first4x4', 64, 2, False), 2.0 * C * 64 * 16 * vh * vw * B, e0, e1))
    return y


def conv_taps4_splitk(x_mosaic, packed, n_images, g, valid_hw, lrelu_slope=None, post_scale=None, post_shift=None, ksplit=None):
    """Split-K 2x2-tap convolution over a g x g mosaic (g = 1: the plain batch): x [ceil(n/g^2), g*h, g*w, Cin_pad] ->
    y [n_images, vh, vw, Cout] (the valid outputs, after bias / LeakyReLU / affine). ksplit None: the library's choice."""
    lib = _lib.load()
    x, act, post_scale, post_shift = _taps4_args(x_mosaic, packed, lrelu_slope, post_scale, post_shift)
    Bm, H, W, C = x.shape
    if H % g or W % g or Bm != (n_images + g * g - 1) // (g * g):
        raise _lib.WitwError('conv_taps4_splitk: %s is not a %dx%d mosaic of %d images' % (tuple(x.shape), g, g, n_images))
    h, w = H // g, W // g
    vh, vw = valid_hw
    S = ksplit or lib.witw_conv3x3_taps4_ksplit(Bm, H, W, C, packed.cout)
    ws = torch.empty((S, Bm, H, W, packed.cout), dtype=torch.float32, device=x.device)
    y = torch.empty((n_images, vh, vw, packed.cout), dtype=torch.float32, device=x.device)
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if S > 1:
        _lib.check(lib.witw_conv3x3_fwd_taps4_ex(x.data_ptr(), packed.wpk.data_ptr(), packed.bias.data_ptr(), None, None, None,
                                                 ws.data_ptr(), Bm, H, W, C, packed.cout, 0, 0., packed.tap_base, S, 0, 0, 0,
                                                 _stream()), 'witw_conv3x3_fwd_taps4_ex')
        _lib.check(lib.witw_taps4_splitk_finish(ws.data_ptr(), S, packed.bias.data_ptr(), act, float(lrelu_slope or 0.),
                                                _p(post_scale), _p(post_shift), y.data_ptr(), n_images, g, h, w, vh, vw,
                                                packed.cout, _stream()), 'witw_taps4_splitk_finish')
    else:       # one slice: the conv applies its own epilogue, the finish pass only gathers the valid outputs out of the mosaic
        _lib.check(lib.witw_conv3x3_fwd_taps4_ex(x.data_ptr(), packed.wpk.data_ptr(), packed.bias.data_ptr(), None, _p(post_scale),
                                                 _p(post_shift), ws.data_ptr(), Bm, H, W, C, packed.cout, act,
                                                 float(lrelu_slope or 0.), packed.tap_base, 1, 0, 0, 0, _stream()),
                   'witw_conv3x3_fwd_taps4_ex')
        zero = torch.zeros((packed.cout,), dtype=torch.float32, device=x.device)
        _lib.check(lib.witw_taps4_splitk_finish(ws.data_ptr(), 1, zero.data_ptr(), 0, 0., None, None, y.data_ptr(), n_images, g, h,
                                                w, vh, vw, packed.cout, _stream()), 'witw_taps4_splitk_finish')
    if prof is not None:
        e1.record()
        prof.append(((lib.witw_conv3x3_tile_n(packed.cout), 1, False, lib.witw_conv3x3_workgroup_waves(Bm, H, W, packed.cout, 1)),
                     2.0 * packed.cin * packed.cout * 4 * vh * vw * n_images, e0, e1))
    return y


def space_to_depth2_mosaic(y_nhwc, g=1):
    """y [B,H,W,C] -> [ceil(B/g^2), g*ceil(H/2), g*ceil(W/2), 4C]: g x g images per mosaic, each the space-to-depth(2) image."""
    lib = _lib.load()
    y = _dev_f32(y_nhwc, 'y')
    B, H, W, C = y.shape
    out = torch.empty(((B + g * g - 1) // (g * g), g * ((H + 1) // 2), g * ((W + 1) // 2), 4 * C), dtype=torch.float32, device=y.device)
    _lib.check(lib.witw_space_to_depth2_mosaic(y.data_ptr(), out.data_ptr(), B, H, W, C, g, _stream()), 'witw_space_to_depth2_mosaic')
    return out


def maxpool2x2_bwd(dy, code, out_hw):
    """dy/code [B,Hp,Wp,C] -> dx [B,H,W,C] (gradient routed to the recorded arg-max position)."""
    lib = _lib.load()
    dy = _dev_f32(dy, 'dy')
    B, Hp, Wp, C = dy.shape
    H, W = out_hw
    dx = torch.empty((B, H, W, C), dtype=torch.float32, device=dy.device)
    _lib.check(lib.witw_maxpool2x2_bwd(dy.data_ptr(), code.data_ptr(), dx.data_ptr(), B, Hp, Wp, H, W, C, _stream()),
               'witw_maxpool2x2_bwd')
    return dx


def conv3x3_wgrad(x_nhwc, dz_nhwc, cin_real, stride_h=1, circular=False, want_bias=True, taps4=False, out=None):
    """-> (dW [Cout,cin_real,3,3], db [Cout] or None) for one conv layer. taps4: only the taps {1,2}^2 are computed
    (filters whose first tap row/column are structurally zero), the others come back as exact zeros. out = (dW, db): write
    into these contiguous tensors (the .grad views of a parallel.GradBucket) instead of fresh ones."""
    lib = _lib.load()
    x = _dev_f32(x_nhwc, 'x')
    dz = _dev_f32(dz_nhwc, 'dz')
    B, H, W, Cin = x.shape
    Ho = (H + 2 - 3) // stride_h + 1
    Cout = dz.shape[3]
    if tuple(dz.shape[:3]) != (B, Ho, W):
        raise _lib.WitwError('conv3x3_wgrad: dz %s does not match x %s (stride %d)' % (tuple(dz.shape), tuple(x.shape), stride_h))
    if out is not None:
        dw, db = out
        if not (dw.is_contiguous() and tuple(dw.shape) == (Cout, cin_real, 3, 3) and dw.dtype == torch.float32 and dw.is_cuda
                and db is not None and db.is_contiguous() and db.numel() == Cout):
            raise _lib.WitwError('conv3x3_wgrad: out must be contiguous float32 GPU tensors [%d,%d,3,3] and [%d]' % (Cout, cin_real, Cout))
    else:
        dw = torch.empty((Cout, cin_real, 3, 3), dtype=torch.float32, device=x.device)
        db = torch.empty((Cout,), dtype=torch.float32, device=x.device) if want_bias else None
    ws = torch.empty(lib.witw_conv3x3_wgrad_workspace_floats(B, H, W, Cin, Cout, stride_h), dtype=torch.float32,
                     device=x.device)
    if taps4:
        if stride_h != 1 or circular:
            raise _lib.WitwError('conv3x3_wgrad: taps4 runs stride 1 with zero padding')
        _lib.check(lib.witw_conv3x3_wgrad_taps4(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), _p(db), ws.data_ptr(), B, H, W, Cin,
                                                cin_real, Cout, 0, _stream()), 'witw_conv3x3_wgrad_taps4')
    else:
        _lib.check(lib.witw_conv3x3_wgrad(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), _p(db), ws.data_ptr(), B, H, W, Cin,
                                          cin_real, Cout, stride_h, int(circular), 0, _stream()), 'witw_conv3x3_wgrad')
    return dw, db


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8):
    lib = _lib.load()
    for t, n in ((param, 'param'), (grad, 'grad'), (exp_avg, 'exp_avg'), (exp_avg_sq, 'exp_avg_sq')):
        _dev_f32(t, n)
    _lib.check(lib.witw_adam_step(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                  param.numel(), float(lr), float(beta1), float(beta2), float(eps), int(step), _stream()),
               'witw_adam_step')


def adam_step_multi(params, grads, exp_avgs, exp_avg_sqs, steps, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8):
    """adam_step over lists of tensors in ceil(len / 48) launches (same arithmetic per element)."""
    import ctypes
    lib = _lib.load()
    k = len(params)
    if not (k and len(grads) == k and len(exp_avgs) == k and len(exp_avg_sqs) == k and len(steps) == k):
        raise _lib.WitwError('adam_step_multi: lists of different lengths')
    for group, name in ((params, 'param'), (grads, 'grad'), (exp_avgs, 'exp_avg'), (exp_avg_sqs, 'exp_avg_sq')):
        for t in group:
            _dev_f32(t, name)
    for p, g, m, v in zip(params, grads, exp_avgs, exp_avg_sqs):
        if not (g.numel() == p.numel() and m.numel() == p.numel() and v.numel() == p.numel()):
            raise _lib.WitwError('adam_step_multi: tensor sizes differ within a parameter')
    ptrs = [(ctypes.c_void_p * k)(*[t.data_ptr() for t in group]) for group in (params, grads, exp_avgs, exp_avg_sqs)]
    n = (ctypes.c_longlong * k)(*[p.numel() for p in params])
    st = (ctypes.c_int * k)(*[int(s) for s in steps])
    _lib.check(lib.witw_adam_step_multi(ptrs[0], ptrs[1], ptrs[2], ptrs[3], n, st, k, float(lr), float(beta1), float(beta2),
                                        float(eps), _stream()), 'witw_adam_step_multi')


# ----------------------------------------------------------------------------- matching
def match_bwd(overhead_embed, surface_embed, orientation, score, workspace, grad_distance, need_ov=True, need_su=True):
    lib = _lib.load()
    ov = _dev_f32(overhead_embed, 'overhead_embed')
    su = _dev_f32(surface_embed, 'surface_embed')
    gd = _dev_f32(grad_distance, 'grad_distance')
    Bo, Bs, We = ov.shape[0], su.shape[0], su.shape[3]
    gov = torch.empty_like(ov) if need_ov else None
    gsu = torch.empty_like(su) if need_su else None
    n_scratch = lib.witw_match_bwd_scratch_floats(Bo, Bs, We) if need_su else 0
    scratch = torch.empty(n_scratch, dtype=torch.float32, device=ov.device) if n_scratch > 0 else None
    _lib.check(lib.witw_match_bwd(ov.data_ptr(), su.data_ptr(), orientation.data_ptr(), score.data_ptr(),
                                  workspace.data_ptr(), gd.data_ptr(), _p(gov), _p(gsu), _p(scratch), Bo, Bs, We, _stream()),
               'witw_match_bwd')
    return gov, gsu


def match_fwd(overhead_embed, surface_embed, want_score=False, want_workspace=False):
    """Fused correlation -> argmax -> window norm -> chord distance (no crop tensor).
    overhead_embed [Bo,16,4,64], surface_embed [Bs,16,4,We] -> (orientation int64 [Bo,Bs],
    distance f32 [Bo,Bs][, max score f32 [Bo,Bs]])."""
    lib = _lib.load()
    ov = _dev_f32(overhead_embed, 'overhead_embed')
    su = _dev_f32(surface_embed, 'surface_embed')
    if ov.dim() != 4 or su.dim() != 4 or ov.shape[1] * ov.shape[2] != 64 or ov.shape[3] != 64:
        raise _lib.WitwError('match_fwd: overhead embedding must be [Bo,16,4,64], got %s' % (tuple(ov.shape),))
    if su.shape[1] != ov.shape[1] or su.shape[2] != ov.shape[2]:
        raise _lib.WitwError('match_fwd: surface embedding %s does not match overhead %s' % (tuple(su.shape), tuple(ov.shape)))
    Bo, Bs, We = ov.shape[0], su.shape[0], su.shape[3]
    ori = torch.empty((Bo, Bs), dtype=torch.int64, device=ov.device)
    dist = torch.empty((Bo, Bs), dtype=torch.float32, device=ov.device)
    score = torch.empty((Bo, Bs), dtype=torch.float32, device=ov.device) if want_score else None
    ws = torch.empty(lib.witw_match_workspace_floats(Bo, Bs), dtype=torch.float32, device=ov.device)
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(lib.witw_match_fwd(ov.data_ptr(), su.data_ptr(), Bo, Bs, We, ori.data_ptr(), dist.data_ptr(), _p(score),
                                  ws.data_ptr(), _stream()), 'witw_match_fwd')
    if prof is not None:      # the launch = two small norm kernels + the match kernel
        e1.record()
        prof.append((('match', We), 2.0 * 64 * (64 * We) * Bo * Bs, e0, e1))
    if want_workspace:
        return ori, dist, score, ws
    return (ori, dist, score) if want_score else (ori, dist)


SPECTRA_ALIGN = 32      # an overhead row's stored chunk order depends on bit 4 of its index (csrc/match_dft.hip:331)


class Spectra(object):
    """Row spectra of one side of the spectral match, f32 [B,32,128] in `data`, plus what the stored layout depends on: the
    side (`overhead`: the two sides order the 8-byte chunks of every 16-byte slot differently) and, for the overhead side, each
    row's index modulo 32 in the tensor the spectra were computed from. Slices (`rows`) and concatenations (`Spectra.cat`) are
    therefore only defined at multiples of SPECTRA_ALIGN = 32 rows; match_fwd_dft refuses the wrong side or a misaligned view
    instead of returning wrong distances."""

    def __init__(self, data, overhead, base_row=0):
        self.data, self.overhead, self.base_row = data, bool(overhead), int(base_row)

    def __len__(self):
        return self.data.shape[0]

    def rows(self, start, count):
        if self.overhead and start % SPECTRA_ALIGN:
            raise _lib.WitwError('Spectra.rows: overhead spectra can only be sliced at multiples of %d rows (start %d)'
                                 % (SPECTRA_ALIGN, start))
        return Spectra(self.data[start:start + count], self.overhead, self.base_row + start)

    @staticmethod
    def cat(parts):
        parts = list(parts)
        if len({p.overhead for p in parts}) != 1:
            raise _lib.WitwError('Spectra.cat: mixed sides')
        if parts[0].overhead and any((p.base_row % SPECTRA_ALIGN) or (len(p) % SPECTRA_ALIGN and p is not parts[-1]) for p in parts):
            raise _lib.WitwError('Spectra.cat: every overhead block but the last must hold a multiple of %d rows' % SPECTRA_ALIGN)
        return Spectra(torch.cat([p.data for p in parts]), parts[0].overhead, 0)


def spectrum_slots():
    """storage slots of 128 floats per embedding spectrum, as the loaded library lays them out (32)"""
    return int(_lib.load().witw_match_spectrum_floats(1)) // 128


def match_spectrum(embed, overhead):
    """Row spectra of embeddings [B,16,4,W] (overhead side W = 64, surface side W = We <= 64) -> Spectra over f32 [B,32,128]
    (slot t = 1..31: frequency t; slot 0: the real spectra at frequencies 0 and 32; the slot count is the library's:
    witw_match_spectrum_floats), the operand of match_fwd_dft. `overhead` (required: a fov-360 surface embedding is 64 columns wide too) names the side; the
    sides differ in the order of the 8-byte chunks inside each 16-byte slot, and an overhead row's order also depends on
    bit 4 of its index (csrc/match_dft.hip: conflict-free LDS operand reads) -- hence the 32-row alignment rule of Spectra."""
    lib = _lib.load()
    e = _dev_f32(embed, 'embed')
    if e.dim() != 4 or e.shape[1] * e.shape[2] != 64 or not (1 <= e.shape[3] <= 64):
        raise _lib.WitwError('match_spectrum: embedding must be [B,16,4,W<=64], got %s' % (tuple(e.shape),))
    if overhead and e.shape[3] != 64:
        raise _lib.WitwError('match_spectrum: an overhead embedding is 64 columns wide, got %d' % e.shape[3])
    spec = torch.empty((e.shape[0], spectrum_slots(), 128), dtype=torch.float32, device=e.device)
    _lib.check(lib.witw_match_spectrum(e.data_ptr(), spec.data_ptr(), e.shape[0], e.shape[3], int(bool(overhead)), _stream()),
               'witw_match_spectrum')
    return Spectra(spec, overhead)


def match_fwd_dft(overhead_embed, surface_embed, spec_ov=None, spec_su=None, want_score=False, want_orientation=True,
                  want_workspace=False, want_gap=False):
    """match_fwd through the row spectra (21k FLOP per pair instead of 524k): same outputs; scores agree with the direct sum to
    fp32 rounding, so an orientation can differ only between shifts whose scores tie to ~1e-6. spec_ov / spec_su: cached
    match_spectrum of the two sides (the gallery's is computed once per retrieval)."""
    lib = _lib.load()
    ov = _dev_f32(overhead_embed, 'overhead_embed')
    su = _dev_f32(surface_embed, 'surface_embed')
    if ov.dim() != 4 or su.dim() != 4 or ov.shape[1] * ov.shape[2] != 64 or ov.shape[3] != 64:
        raise _lib.WitwError('match_fwd_dft: overhead embedding must be [Bo,16,4,64], got %s' % (tuple(ov.shape),))
    if su.shape[1] != ov.shape[1] or su.shape[2] != ov.shape[2]:
        raise _lib.WitwError('match_fwd_dft: surface embedding %s does not match overhead %s' % (tuple(su.shape), tuple(ov.shape)))
    Bo, Bs, We = ov.shape[0], su.shape[0], su.shape[3]
    spec_ov = match_spectrum(ov, overhead=True) if spec_ov is None else spec_ov
    spec_su = match_spectrum(su, overhead=False) if spec_su is None else spec_su
    for name, sp, n, side in (('spec_ov', spec_ov, Bo, True), ('spec_su', spec_su, Bs, False)):
        if not isinstance(sp, Spectra):
            raise _lib.WitwError('match_fwd_dft: %s must be a Spectra (ops.match_spectrum), not a bare tensor' % name)
        if sp.overhead != side:
            raise _lib.WitwError('match_fwd_dft: %s holds %s-side spectra' % (name, 'overhead' if sp.overhead else 'surface'))
        if side and sp.base_row % SPECTRA_ALIGN:
            raise _lib.WitwError('match_fwd_dft: %s starts at row %d of its tensor; overhead spectra are laid out per index modulo %d'
                                 % (name, sp.base_row, SPECTRA_ALIGN))
        d = sp.data
        if not (d.is_cuda and d.dtype == torch.float32 and d.is_contiguous() and tuple(d.shape) == (n, spectrum_slots(), 128)):
            raise _lib.WitwError('match_fwd_dft: %s must be a contiguous float32 GPU tensor [%d,%d,128]' % (name, n, spectrum_slots()))
    spec_ov, spec_su = spec_ov.data, spec_su.data
    # want_orientation=False (retrieval: only distances are ranked) skips the int64 matrix, two thirds of the output bytes
    ori = torch.empty((Bo, Bs), dtype=torch.int64, device=ov.device) if want_orientation else None
    dist = torch.empty((Bo, Bs), dtype=torch.float32, device=ov.device)
    score = torch.empty((Bo, Bs), dtype=torch.float32, device=ov.device) if want_score else None
    ws = torch.empty(lib.witw_match_dft_workspace_floats(Bo, Bs), dtype=torch.float32, device=ov.device)
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    gap = torch.empty((Bo, Bs), dtype=torch.float32, device=ov.device) if want_gap else None
    if want_gap:
        _lib.check(lib.witw_match_fwd_dft_gap(ov.data_ptr(), su.data_ptr(), spec_ov.data_ptr(), spec_su.data_ptr(), Bo, Bs, We,
                                              _p(ori), dist.data_ptr(), _p(score), gap.data_ptr(), ws.data_ptr(), _stream()),
                   'witw_match_fwd_dft_gap')
    else:
        _lib.check(lib.witw_match_fwd_dft(ov.data_ptr(), su.data_ptr(), spec_ov.data_ptr(), spec_su.data_ptr(), Bo, Bs, We,
                                          _p(ori), dist.data_ptr(), _p(score), ws.data_ptr(), _stream()), 'witw_match_fwd_dft')
    if prof is not None:      # FLOP of this form per pair: 33 frequencies x (2 rows x K=128 x 2 + 32 shifts x K=2 x 2) = 21,120 (algorithmic; the kernel runs 32 slots)
        e1.record()
        prof.append((('match_dft', We), 33.0 * (2 * 128 * 2 + 32 * 2 * 2) * Bo * Bs, e0, e1))
    if want_gap:            # best - runner-up score per pair (how far the chosen shift is from a tie), then the workspace
        return ori, dist, gap, ws
    if want_workspace:      # window norms [Bo,64] then surface norms [Bs]: what match_pairs needs to re-score pairs of this pass
        return ori, dist, ws
    return (ori, dist, score) if want_score else (ori, dist)


# Rounding of one fp32 orientation score, relative to |ov| |su|: both match kernels stay inside it against the fp64 sum
# (tests/test_match_dft_gpu.py asserts it), so two kernels' scores of one pair differ by at most twice that and their chord
# distances 2 (1 - score / (|window| |su|)) -- at full width |window| = |ov| -- by at most 4 x (+ the few ulps of the final ops).
SCORE_ROUNDING = 2e-6
DISTANCE_EPS = 1e-5


def match_pairs(overhead_embed, surface_embed, wn, sn, pair_o, pair_s, want_orientation=True):
    """(overhead row pair_o[i], surface row pair_s[i]) -> (orientation int64 [n] or None, distance f32 [n]) with the bits
    match_fwd gives those pairs. wn / sn: the norm blocks of a match_fwd / match_fwd_dft workspace over the same tensors."""
    lib = _lib.load()
    ov = _dev_f32(overhead_embed, 'overhead_embed')
    su = _dev_f32(surface_embed, 'surface_embed')
    Bo, Bs, We = ov.shape[0], su.shape[0], su.shape[3]
    n = int(pair_o.numel())
    for name, t in (('pair_o', pair_o), ('pair_s', pair_s)):
        if not (t.is_cuda and t.dtype == torch.int32 and t.is_contiguous() and t.numel() == n):
            raise _lib.WitwError('match_pairs: %s must be a contiguous int32 GPU tensor of %d entries' % (name, n))
    if wn.numel() != Bo * 64 or sn.numel() != Bs:
        raise _lib.WitwError('match_pairs: wn / sn must hold [Bo,64] / [Bs] norms')
    ori = torch.empty((n,), dtype=torch.int64, device=ov.device) if want_orientation else None
    dist = torch.empty((n,), dtype=torch.float32, device=ov.device)
    if n:
        _lib.check(lib.witw_match_pairs(ov.data_ptr(), su.data_ptr(), _dev_f32(wn, 'wn').data_ptr(), _dev_f32(sn, 'sn').data_ptr(),
                                        pair_o.data_ptr(), pair_s.data_ptr(), n, Bo, Bs, We, _p(ori), dist.data_ptr(), None,
                                        _stream()), 'witw_match_pairs')
    return ori, dist


def rank_count_band(distance, threshold, eps):
    """distance [Bo,Bs] known to +-eps -> (counts int32 [Bs] of the rows surely below threshold - eps, pair_o, pair_s int32: the
    (row, query) inside the band, to be re-scored exactly)."""
    lib = _lib.load()
    d = _dev_f32(distance, 'distance')
    t = _dev_f32(threshold, 'threshold')
    Bo, Bs = d.shape
    if t.numel() != Bs:
        raise _lib.WitwError('rank_count_band: threshold must have one entry per query')
    counts = torch.empty((Bs,), dtype=torch.int32, device=d.device)
    n = torch.empty((1,), dtype=torch.int32, device=d.device)
    cap = max(1 << 16, (Bo * Bs) // 4096)
    while True:
        po = torch.empty((cap,), dtype=torch.int32, device=d.device)
        ps = torch.empty((cap,), dtype=torch.int32, device=d.device)
        _lib.check(lib.witw_rank_count_band(d.data_ptr(), t.data_ptr(), float(eps), counts.data_ptr(), po.data_ptr(), ps.data_ptr(),
                                            n.data_ptr(), cap, Bo, Bs, _stream()), 'witw_rank_count_band')
        got = int(n.item())
        if got <= cap:
            return counts, po[:got].contiguous(), ps[:got].contiguous()
        cap = got            # rare: more pairs in the band than the list holds -> once more with room for all


def rank_count_resolved(distance, threshold, eps, overhead_embed, surface_embed, wn, sn):
    """rank_count_band + the exact re-scoring of its band in ONE stream sequence without a host round trip: -> (counts int32 [Bs] =
    #{o : exact distance[o][q] <= threshold[q]}, n int32 [1] on the device = pairs in the band, capacity). The result is complete iff
    n <= capacity, which the caller checks once at the end of its pass (retrieve(method='dft')); beyond it the list overflowed and the
    chunk has to be redone through rank_count_band."""
    lib = _lib.load()
    d = _dev_f32(distance, 'distance')
    t = _dev_f32(threshold, 'threshold')
    ov = _dev_f32(overhead_embed, 'overhead_embed')
    su = _dev_f32(surface_embed, 'surface_embed')
    Bo, Bs = d.shape
    if t.numel() != Bs or ov.shape[0] != Bo or su.shape[0] != Bs:
        raise _lib.WitwError('rank_count_resolved: distance [Bo,Bs], threshold [Bs] and the two embedding batches must agree')
    if wn.numel() != Bo * 64 or sn.numel() != Bs:
        raise _lib.WitwError('rank_count_resolved: wn / sn must hold [Bo,64] / [Bs] norms')
    counts = torch.empty((Bs,), dtype=torch.int32, device=d.device)
    n = torch.empty((1,), dtype=torch.int32, device=d.device)
    cap = max(1 << 16, (Bo * Bs) // 4096)
    po = torch.empty((cap,), dtype=torch.int32, device=d.device)
    ps = torch.empty((cap,), dtype=torch.int32, device=d.device)
    _lib.check(lib.witw_rank_count_band(d.data_ptr(), t.data_ptr(), float(eps), counts.data_ptr(), po.data_ptr(), ps.data_ptr(),
                                        n.data_ptr(), cap, Bo, Bs, _stream()), 'witw_rank_count_band')
    _lib.check(lib.witw_match_pairs_count(ov.data_ptr(), su.data_ptr(), _dev_f32(wn, 'wn').data_ptr(), _dev_f32(sn, 'sn').data_ptr(),
                                          po.data_ptr(), ps.data_ptr(), n.data_ptr(), cap, Bo, Bs, su.shape[3], t.data_ptr(),
                                          counts.data_ptr(), _stream()), 'witw_match_pairs_count')
    return counts, n, cap


def crop_overhead(overhead_embed, orientation, surface_width):
    lib = _lib.load()
    ov = _dev_f32(overhead_embed, 'overhead_embed')
    if not (orientation.is_cuda and orientation.dtype == torch.int64 and orientation.is_contiguous()):
        raise _lib.WitwError('crop_overhead: orientation must be a contiguous int64 GPU tensor')
    Bo, Bs = orientation.shape
    c, h, w = ov.shape[1:]
    if c * h != 64 or w != 64:
        raise _lib.WitwError('crop_overhead: overhead embedding must be [Bo,16,4,64]')
    out = torch.empty((Bo, Bs, c, h, surface_width), dtype=torch.float32, device=ov.device)
    _lib.check(lib.witw_crop_overhead(ov.data_ptr(), orientation.data_ptr(), out.data_ptr(), Bo, Bs, surface_width,
                                      _stream()), 'witw_crop_overhead')
    return out


def l2_distance(overhead_cropped, surface_embed):
    lib = _lib.load()
    cr = _dev_f32(overhead_cropped, 'overhead_cropped')
    su = _dev_f32(surface_embed, 'surface_embed')
    Bo, Bs = cr.shape[:2]
    n = su[0].numel()
    if cr[0, 0].numel() != n or su.shape[0] != Bs:
        raise _lib.WitwError('l2_distance: shapes %s vs %s do not match' % (tuple(cr.shape), tuple(su.shape)))
    dist = torch.empty((Bo, Bs), dtype=torch.float32, device=cr.device)
    _lib.check(lib.witw_l2_distance(cr.data_ptr(), su.data_ptr(), dist.data_ptr(), Bo, Bs, n, _stream()), 'witw_l2_distance')
    return dist


def rank_count(distance, true_offset=0):
    """ranks[q] = #{o : D[o,q] <= D[q+true_offset, q]} as int32 [Bs]."""
    lib = _lib.load()
    d = _dev_f32(distance, 'distance')
    Bo, Bs = d.shape
    ranks = torch.empty((Bs,), dtype=torch.int32, device=d.device)
    _lib.check(lib.witw_rank_count(d.data_ptr(), ranks.data_ptr(), Bo, Bs, true_offset, _stream()), 'witw_rank_count')
    return ranks


def topk_smallest(distance, k, row_offset=0):
    """Per query column of D [Bo,Bs]: the k smallest distances and their gallery rows, (distance, index) ascending.
    -> (values f32 [Bs,k], indices int64 [Bs,k])."""
    lib = _lib.load()
    d = _dev_f32(distance, 'distance')
    Bo, Bs = d.shape
    vals = torch.empty((Bs, k), dtype=torch.float32, device=d.device)
    idx = torch.empty((Bs, k), dtype=torch.int64, device=d.device)
    nws = lib.witw_topk_workspace_bytes(Bo, Bs, int(k))
    ws = torch.empty((nws,), dtype=torch.uint8, device=d.device) if nws > 0 else None      # long galleries: rows split over workgroups
    _lib.check(lib.witw_topk_smallest_ws(d.data_ptr(), vals.data_ptr(), idx.data_ptr(), Bo, Bs, int(k), int(row_offset), _p(ws),
                                         _stream()), 'witw_topk_smallest_ws')
    return vals, idx


def rank_count_thresh(distance, threshold):
    """ranks[q] = #{o : D[o,q] <= threshold[q]} over the rows present (one gallery shard)."""
    lib = _lib.load()
    d = _dev_f32(distance, 'distance')
    t = _dev_f32(threshold, 'threshold')
    Bo, Bs = d.shape
    if t.numel() != Bs:
        raise _lib.WitwError('rank_count_thresh: threshold must have one entry per query')
    ranks = torch.empty((Bs,), dtype=torch.int32, device=d.device)
    _lib.check(lib.witw_rank_count_thresh(d.data_ptr(), t.data_ptr(), ranks.data_ptr(), Bo, Bs, _stream()),
               'witw_rank_count_thresh')
    return ranks


def dropout2d_scales(seed, encoder, step, rank, layers, batch, channels, p, device):
    """Dropout2d scales [len(layers), batch, channels] (0 or 1/(1-p)) from the counter-based generator of csrc/loss.hip."""
    import ctypes
    lib = _lib.load()
    out = torch.empty((len(layers), batch, channels), dtype=torch.float32, device=device)
    arr = (ctypes.c_int * len(layers))(*[int(v) for v in layers])
    _lib.check(lib.witw_dropout2d_scales(out.data_ptr(), int(seed) & 0xFFFFFFFFFFFFFFFF, int(encoder), int(step) & 0xFFFFFFFF, int(rank),
                                         ctypes.cast(arr, ctypes.c_void_p), len(layers), batch, channels, float(p), _stream()),
               'witw_dropout2d_scales')
    return out


def triplet_loss_fwd(distance, alpha=10.):
    """-> (loss [1] f32, workspace [4B] holding the row/column partials for the backward)."""
    lib = _lib.load()
    d = _dev_f32(distance, 'distance')
    if d.dim() != 2 or d.shape[0] != d.shape[1]:
        raise _lib.WitwError('triplet_loss: distance matrix must be square, got %s' % (tuple(d.shape),))
    B = d.shape[0]
    loss = torch.empty((1,), dtype=torch.float32, device=d.device)
    ws = torch.empty((4 * B,), dtype=torch.float32, device=d.device)
    _lib.check(lib.witw_triplet_loss_fwd(d.data_ptr(), B, float(alpha), loss.data_ptr(), ws.data_ptr(), _stream()),
               'witw_triplet_loss_fwd')
    return loss, ws


def triplet_loss_slab_fwd(distance_slab, diag, col0, alpha=10.):
    """Un-normalised loss partial over this rank's column slab [Bo,Bs] (see witw_triplet_loss_slab_fwd)."""
    lib = _lib.load()
    d = _dev_f32(distance_slab, 'distance_slab')
    g = _dev_f32(diag, 'diag')
    Bo, Bs = d.shape
    if g.numel() != Bo:
        raise _lib.WitwError('triplet_loss_slab: diag must have one entry per overhead')
    out = torch.empty((1,), dtype=torch.float32, device=d.device)
    ws = torch.empty((Bo,), dtype=torch.float32, device=d.device)
    _lib.check(lib.witw_triplet_loss_slab_fwd(d.data_ptr(), g.data_ptr(), Bo, Bs, col0, float(alpha), out.data_ptr(),
                                              ws.data_ptr(), _stream()), 'witw_triplet_loss_slab_fwd')
    return out


def triplet_loss_slab_sig(distance_slab, diag, col0, alpha=10.):
    """-> (rowsig [Bo] partial over this slab's surfaces, colsig [Bs]) for the sharded loss backward."""
    lib = _lib.load()
    d = _dev_f32(distance_slab, 'distance_slab')
    g = _dev_f32(diag, 'diag')
    Bo, Bs = d.shape
    rowsig = torch.empty((Bo,), dtype=torch.float32, device=d.device)
    colsig = torch.empty((Bs,), dtype=torch.float32, device=d.device)
    _lib.check(lib.witw_triplet_loss_slab_sig(d.data_ptr(), g.data_ptr(), Bo, Bs, col0, float(alpha), rowsig.data_ptr(),
                                              colsig.data_ptr(), _stream()), 'witw_triplet_loss_slab_sig')
    return rowsig, colsig


def triplet_loss_slab_bwd(distance_slab, diag, rowsig, colsig, grad_loss, col0, alpha=10.):
    """Gradient of the GLOBAL loss w.r.t. this rank's distance slab [Bo,Bs]; rowsig must be summed over the ranks."""
    lib = _lib.load()
    d = _dev_f32(distance_slab, 'distance_slab')
    Bo, Bs = d.shape
    gd = torch.empty_like(d)
    gl = _dev_f32(grad_loss.reshape(1), 'grad_loss')
    _lib.check(lib.witw_triplet_loss_slab_bwd(d.data_ptr(), _dev_f32(diag, 'diag').data_ptr(), _dev_f32(rowsig, 'rowsig').data_ptr(),
                                              _dev_f32(colsig, 'colsig').data_ptr(), gl.data_ptr(), gd.data_ptr(), Bo, Bs, col0,
                                              float(alpha), _stream()), 'witw_triplet_loss_slab_bwd')
    return gd


def triplet_loss_bwd(distance, ws, grad_loss, alpha=10.):
    lib = _lib.load()
    d = _dev_f32(distance, 'distance')
    g = _dev_f32(grad_loss.reshape(1).contiguous(), 'grad_loss')
    B = d.shape[0]
    gd = torch.empty_like(d)
    _lib.check(lib.witw_triplet_loss_bwd(d.data_ptr(), ws.data_ptr(), g.data_ptr(), gd.data_ptr(), B, float(alpha),
                                         _stream()), 'witw_triplet_loss_bwd')
    return gd


# ----------------------------------------------------------------------------- data path
def _host_floats(vals):
    import ctypes
    arr = (ctypes.c_float * len(vals))(*[float(v) for v in vals])
    return arr


def resize_bilinear(x, size, mean=None, std=None, n_div255=None):
    """[B,C,Hi,Wi] -> [B,C,Ho,Wo] bilinear (align_corners=False, no antialias), optionally fused
    with (x/255 - mean)/std."""
    import ctypes
    lib = _lib.load()
    x = _dev_f32(x, 'x')
    B, C, Hi, Wi = x.shape
    Ho, Wo = size
    y = torch.empty((B, C, Ho, Wo), dtype=torch.float32, device=x.device)
    m = s = None
    if mean is not None:
        m, s = _host_floats(mean), _host_floats(std)
    nd = C if n_div255 is None else n_div255
    _lib.check(lib.witw_resize_bilinear_normalize(x.data_ptr(), y.data_ptr(), B, C, Hi, Wi, Ho, Wo,
                                                  ctypes.cast(m, ctypes.c_void_p) if m is not None else None,
                                                  ctypes.cast(s, ctypes.c_void_p) if s is not None else None, nd, _stream()),
               'witw_resize_bilinear_normalize')
    return y


def resize_batched(desc, batch, channels, size, wfull=None, kind=0, mean=None, std=None, n_div255=None):
    """One launch for a batch of differently sized images. desc: int64 GPU tensor [B,5] = {device address, H, W, start column,
    channels per stored pixel} per image; kind 0 = float32 CHW sources, 1 = uint8 HWC. -> [B,channels,Ho,Wo] (see
    witw_resize_bilinear_normalize_batched for wfull / start)."""
    import ctypes
    lib = _lib.load()
    if not (desc.is_cuda and desc.dtype == torch.int64 and desc.is_contiguous() and tuple(desc.shape) == (batch, 5)):
        raise _lib.WitwError('resize_batched: desc must be a contiguous int64 GPU tensor [%d,5]' % batch)
    Ho, Wo = size
    y = torch.empty((batch, channels, Ho, Wo), dtype=torch.float32, device=desc.device)
    m = s = None
    if mean is not None:
        m, s = _host_floats(mean), _host_floats(std)
    nd = channels if n_div255 is None else n_div255
    _lib.check(lib.witw_resize_bilinear_normalize_batched(desc.data_ptr(), y.data_ptr(), batch, channels, Ho, Wo, Wo if wfull is None else wfull,
                                                          int(kind), ctypes.cast(m, ctypes.c_void_p) if m is not None else None,
                                                          ctypes.cast(s, ctypes.c_void_p) if s is not None else None, nd, _stream()),
               'witw_resize_bilinear_normalize_batched')
    return y


def normalize(x, mean, std, n_div255=None):
    import ctypes
    lib = _lib.load()
    x = _dev_f32(x, 'x')
    B, C, H, W = x.shape
    y = torch.empty_like(x)
    m, s = _host_floats(mean), _host_floats(std)
    nd = C if n_div255 is None else n_div255
    _lib.check(lib.witw_normalize(x.data_ptr(), y.data_ptr(), B, C, H, W, ctypes.cast(m, ctypes.c_void_p),
                                  ctypes.cast(s, ctypes.c_void_p), nd, _stream()), 'witw_normalize')
    return y


_POLAR_LUT = {}


def polar_lut(device, size=256, h_s=128, w_s=512):
    """Constant sampling table of PolarTransform, built on the host in fp64 exactly as
    model/cvig_fov.py:197-201 (grid) and :163-181 (clip-before-weights bilinear taps), then
    rounded to fp32 weights like torch.FloatTensor(...) does."""
    import math
    import numpy as np
    key = (str(device), size, h_s, w_s)
    if key in _POLAR_LUT:
        return _POLAR_LUT[key]
    xx, yy = np.meshgrid(range(w_s), range(h_s))
    y = (size / 2) + (size / 2) * (h_s - 1 - yy) / h_s * np.cos(2 * math.pi * xx / w_s)
    x = (size / 2) - (size / 2) * (h_s - 1 - yy) / h_s * np.sin(2 * math.pi * xx / w_s)
    x0 = np.floor(x).astype(int)
    x1 = x0 + 1
    y0 = np.floor(y).astype(int)
    y1 = y0 + 1
    x0 = np.clip(x0, 0, size - 1)
    x1 = np.clip(x1, 0, size - 1)
    y0 = np.clip(y0, 0, size - 1)
    y1 = np.clip(y1, 0, size - 1)
    wts = np.stack([(x1 - x) * (y1 - y), (x1 - x) * (y - y0), (x - x0) * (y1 - y), (x - x0) * (y - y0)], -1)
    taps = np.stack([y0 * size + x0, y1 * size + x0, y0 * size + x1, y1 * size + x1], -1)
    lut = (torch.from_numpy(taps.reshape(-1, 4).astype(np.int32)).to(device),
           torch.from_numpy(wts.reshape(-1, 4).astype(np.float32)).to(device))
    _POLAR_LUT[key] = lut
    return lut


_POLAR_TILES = {}


def polar_tiles(device, size=256, h_s=128, w_s=512):
    """Tile decomposition of the polar sampling table for witw_polar_from_raw: the h_s x w_s outputs are cut into tiles of rh
    radii x aw angles (powers of two, rh * aw <= 256: one wave's work) -- the shape whose boxes cover the least area in total -- each with the
    bounding box {x0, y0, width, height} in the size x size plane of every tap it reads. -> (int32 GPU [n_tile,8], largest box
    area, the sampling table's taps as offsets inside their tile's box: int32 GPU [h_s*w_s,4]) or None when no shape fits the kernel's limits (callers then use the separate launches)."""
    import numpy as np
    key = (str(device), size, h_s, w_s)
    if key in _POLAR_TILES:
        return _POLAR_TILES[key]
    taps = polar_lut(device, size, h_s, w_s)[0].cpu().numpy().reshape(h_s, w_s, 4)
    ty, tx = taps // size, taps % size
    best = None
    for aw in [2 ** k for k in range(9) if 2 ** k <= min(w_s, 256)]:
        for rh in [2 ** k for k in range(9) if 2 ** k <= max(1, 256 // aw)]:
            rh = min(rh, h_s)
            rows = []
            for r0 in range(0, h_s, rh):
                for c0 in range(0, w_s, aw):
                    sx, sy = tx[r0:r0 + rh, c0:c0 + aw], ty[r0:r0 + rh, c0:c0 + aw]
                    rows.append((int(sx.min()), int(sy.min()), int(sx.max() - sx.min() + 1), int(sy.max() - sy.min() + 1),
                                 r0, c0, sx.shape[0], sx.shape[1]))
            t = np.asarray(rows, dtype=np.int32)
            area = t[:, 2].astype(np.int64) * t[:, 3]
            if int(t[:, 2].max()) > 64 or int(t[:, 3].max()) > 64:
                continue
            # total box area = pixels resized per plane; a few large tiles beat many small ones at equal area (table reuse)
            score = (int(area.sum()), len(rows))
            if best is None or score < best[0]:
                best = (score, t, int(area.max()))
    out = None
    if best is not None:
        t = best[1]
        rel = np.zeros((h_s, w_s, 4), dtype=np.int32)       # every tap as an offset inside its tile's box
        for (bx0, by0, bw, _bh, r0, c0, rh, aw) in t.tolist():
            sl = (slice(r0, r0 + rh), slice(c0, c0 + aw))
            rel[sl] = (ty[sl] - by0) * bw + (tx[sl] - bx0)
        out = (torch.from_numpy(t).to(device), best[2], torch.from_numpy(rel.reshape(-1, 4)).to(device))
    _POLAR_TILES[key] = out
    return out


def polar_from_raw(x=None, desc=None, kind=0, batch=None, channels=None, mean=None, std=None, n_div255=None, size=256, h_s=128,
                   w_s=512):
    """Resize(size x size) -> ImageNormalization -> PolarTransform of the overhead image in one launch (witw_polar_from_raw);
    the same bits as resize_bilinear / resize_batched followed by polar_transform. x: fp32 [B,C,Hi,Wi] on the GPU, or desc /
    kind / batch / channels as for resize_batched. -> [B,C,h_s,w_s]."""
    import ctypes
    lib = _lib.load()
    if desc is None:
        x = _dev_f32(x, 'x')
        B, C, Hi, Wi = x.shape
        dev = x.device
    else:
        if not (desc.is_cuda and desc.dtype == torch.int64 and desc.is_contiguous() and tuple(desc.shape) == (batch, 5)):
            raise _lib.WitwError('polar_from_raw: desc must be a contiguous int64 GPU tensor [%d,5]' % batch)
        B, C, Hi, Wi, dev = batch, channels, 0, 0, desc.device
    tiles = polar_tiles(dev, size, h_s, w_s)
    if tiles is None:
        raise _lib.WitwError('polar_from_raw: a %dx%d polar table over a %d^2 image exceeds the fused kernel (use resize + polar_transform)'
                             % (h_s, w_s, size))
    tile_tab, max_box, taps = tiles
    wts = polar_lut(dev, size, h_s, w_s)[1]
    y = torch.empty((B, C, h_s, w_s), dtype=torch.float32, device=dev)
    m = s = None
    if mean is not None:
        m, s = _host_floats(mean), _host_floats(std)
    nd = C if n_div255 is None else n_div255
    _lib.check(lib.witw_polar_from_raw(x.data_ptr() if desc is None else None, None if desc is None else desc.data_ptr(), int(kind),
                                       y.data_ptr(), B, C, Hi, Wi, size, h_s, w_s, taps.data_ptr(), wts.data_ptr(), tile_tab.data_ptr(),
                                       tile_tab.shape[0], max_box, ctypes.cast(m, ctypes.c_void_p) if m is not None else None,
                                       ctypes.cast(s, ctypes.c_void_p) if s is not None else None, nd, _stream()), 'witw_polar_from_raw')
    return y


def bilinear_interpolate(im, x, y):
    """bilinear_interpolate(im, x, y) of model/cvig_fov.py:156-183 for arbitrary sample coordinates: im [C,H,W] (or
    [B,C,H,W]) on the GPU, x / y arrays of one shape (host, fp64 arithmetic as in the reference: indices clipped to the
    image BEFORE the weights are formed, weights rounded to fp32) -> [C,*x.shape] (or [B,C,*x.shape])."""
    import numpy as np
    lib = _lib.load()
    im = _dev_f32(im, 'im')
    squeeze = im.dim() == 3
    if squeeze:
        im = im.unsqueeze(0)
    B, C, H, W = im.shape
    x = np.asarray(x.cpu() if torch.is_tensor(x) else x)
    y = np.asarray(y.cpu() if torch.is_tensor(y) else y)
    if x.shape != y.shape:
        raise _lib.WitwError('bilinear_interpolate: x and y must have the same shape')
    x0 = np.floor(x).astype(int)
    x1 = x0 + 1
    y0 = np.floor(y).astype(int)
    y1 = y0 + 1
    x0 = np.clip(x0, 0, W - 1)
    x1 = np.clip(x1, 0, W - 1)
    y0 = np.clip(y0, 0, H - 1)
    y1 = np.clip(y1, 0, H - 1)
    wts = np.stack([(x1 - x) * (y1 - y), (x1 - x) * (y - y0), (x - x0) * (y1 - y), (x - x0) * (y - y0)], -1)
    taps = np.stack([y0 * W + x0, y1 * W + x0, y0 * W + x1, y1 * W + x1], -1)
    t = torch.from_numpy(np.ascontiguousarray(taps.reshape(-1, 4).astype(np.int32))).to(im.device)
    w = torch.from_numpy(np.ascontiguousarray(wts.reshape(-1, 4).astype(np.float32))).to(im.device)
    n = t.shape[0]
    out = torch.empty((B, C) + tuple(x.shape), dtype=torch.float32, device=im.device)
    _lib.check(lib.witw_bilinear_gather(im.data_ptr(), t.data_ptr(), w.data_ptr(), out.data_ptr(), B, C, H * W, n, _stream()),
               'witw_bilinear_gather')
    return out.squeeze(0) if squeeze else out


def polar_transform(x, h_s=128, w_s=512):
    """[B,C,S,S] -> [B,C,128,512] (PolarTransform, model/cvig_fov.py:186-209)."""
    lib = _lib.load()
    x = _dev_f32(x, 'x')
    B, C, S, S2 = x.shape
    if S != S2:
        raise _lib.WitwError('polar_transform: overhead image must be square')
    taps, wts = polar_lut(x.device, S, h_s, w_s)
    y = torch.empty((B, C, h_s, w_s), dtype=torch.float32, device=x.device)
    _lib.check(lib.witw_polar_transform(x.data_ptr(), taps.data_ptr(), wts.data_ptr(), y.data_ptr(), B, C, S, h_s, w_s,
                                        _stream()), 'witw_polar_transform')
    return y


def rotation_theta(angles_deg, H, W):
    """Host side of torchvision 0.9.1's F.rotate for tensors: inverse matrix of a rotation by `angle` about the
    image centre (_get_inverse_affine_matrix with centre 0, angle -> -angle), as fp32, divided by (W/2, H/2)
    (_gen_affine_grid). -> CPU fp32 [B,3,2] (row k = coefficient of x, y, 1)."""
    import math
    out = torch.empty((len(angles_deg), 3, 2), dtype=torch.float32)
    half = torch.tensor([0.5 * W, 0.5 * H], dtype=torch.float32)
    for i, a in enumerate(angles_deg):
        rot = math.radians(-float(a))
        m = torch.tensor([[math.cos(rot), math.sin(rot), 0.0], [-math.sin(rot), math.cos(rot), 0.0]], dtype=torch.float32)
        out[i] = m.t() / half
    return out


def rotate_nearest(x, angles_deg):
    """x [B,C,H,W] fp32 on the GPU rotated counter-clockwise by angles_deg[b] degrees about the centre, nearest
    neighbour, zero fill, same size (model/cvig_baseline.py:142)."""
    lib = _lib.load()
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
    x = x.contiguous()
    B, C, H, W = x.shape
    theta = rotation_theta(angles_deg, H, W).to(x.device)
    y = torch.empty_like(x)
    _lib.check(lib.witw_rotate_nearest(x.data_ptr(), theta.data_ptr(), y.data_ptr(), B, C, H, W, _stream()),
               'witw_rotate_nearest')
    return y


# ----------------------------------------------------------------------------- cvig_baseline pieces
def space_to_depth2(x, valid_hw=None, cpad=None, in_nchw=False, normalize=False, scale=None, shift=None):
    lib = _lib.load()
    x = _dev_f32(x, 'x')
    if in_nchw:
        B, C, Hp, Wp = x.shape
    else:
        B, Hp, Wp, C = x.shape
    H, W = valid_hw if valid_hw is not None else (Hp, Wp)
    cpad = cpad or (4 * C + 7) // 8 * 8
    y = torch.empty((B, (H + 1) // 2, (W + 1) // 2, cpad), dtype=torch.float32, device=x.device)
    _lib.check(lib.witw_space_to_depth2(x.data_ptr(), y.data_ptr(), B, Hp, Wp, H, W, C, cpad, int(in_nchw), int(normalize),
                                        _p(scale), _p(shift), _stream()), 'witw_space_to_depth2')
    return y


def gem_pool(x_nhwc, valid_hw, out, col0, p=3., scale=None, shift=None):
    lib = _lib.load()
    x = _dev_f32(x_nhwc, 'x')
    B, Hp, Wp, C = x.shape
    H, W = valid_hw
    _lib.check(lib.witw_gem_pool(x.data_ptr(), out.data_ptr(), B, Hp, Wp, H, W, C, out.shape[1], col0, float(p), _p(scale),
                                 _p(shift), _stream()), 'witw_gem_pool')
    return out


def embed_normalize_(f):
    lib = _lib.load()
    f = _dev_f32(f, 'f')
    _lib.check(lib.witw_embed_normalize(f.data_ptr(), f.shape[0], f.shape[1], _stream()), 'witw_embed_normalize')
    return f


def pairwise_sqdist(a, b, take_sqrt=False):
    lib = _lib.load()
    a, b = _dev_f32(a, 'a'), _dev_f32(b, 'b')
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[1]:
        raise _lib.WitwError('pairwise_sqdist: need [Na,n] and [Nb,n]')
    D = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    _lib.check(lib.witw_pairwise_sqdist(a.data_ptr(), b.data_ptr(), D.data_ptr(), a.shape[0], b.shape[0], a.shape[1],
                                        int(take_sqrt), _stream()), 'witw_pairwise_sqdist')
    return D


def exhaustive_triplet_loss(D, soft_margin=False, alpha=10., margin=1.):
    lib = _lib.load()
    D = _dev_f32(D, 'D')
    B = D.shape[0]
    loss = torch.empty((1,), dtype=torch.float32, device=D.device)
    ws = torch.empty((B,), dtype=torch.float32, device=D.device)
    _lib.check(lib.witw_exhaustive_triplet_loss(D.data_ptr(), B, int(soft_margin), float(alpha), float(margin),
                                                loss.data_ptr(), ws.data_ptr(), _stream()), 'witw_exhaustive_triplet_loss')
    return loss.reshape(())


def last_kernel_variant():
    """Name of the conv kernel instantiation the calling thread's last conv launcher picked (witw_last_kernel_variant), spelled
    as in rocprof kernel names."""
    v = _lib.load().witw_last_kernel_variant()
    return v.decode() if v else ''


# ----------------------------------------------------------------------------- bf16 inference path
def bf16_mfma16(enable=None):
    """MFMA shape of the bf16 inference forward where both kernels apply: True = 16x16x32 (default), False = 32x32x16; None only
    queries. Returns the previous setting (witw_conv3x3_bf16_mfma16)."""
    return bool(_lib.load().witw_conv3x3_bf16_mfma16(-1 if enable is None else int(bool(enable))))


def bf16_wres(enable=None):
    """Weight-resident kernel for the 64-input-channel bf16 forward (layer 5): True = on (default), False = tiled kernels; None
    only queries. Returns the previous setting (witw_conv3x3_bf16_wres)."""
    return bool(_lib.load().witw_conv3x3_bf16_wres(-1 if enable is None else int(bool(enable))))


class PackedConvBf16:
    """bf16 filter packing of one 3x3 conv for the bf16 MFMA kernel + fp32 bias padded to the channel tile."""

    def __init__(self, weight, bias, transpose_flip=False, reuse=None):
        lib = _lib.load()
        w = _dev_f32(weight.detach(), 'weight')
        # transpose_flip: the dgrad filter w_t[ci][co][kh][kw] = w[co][ci][2-kh][2-kw], built by the pack kernel itself
        self.cout, self.cin = (w.shape[1], w.shape[0]) if transpose_flip else (w.shape[0], w.shape[1])
        self.cin_pad = (self.cin + 15) // 16 * 16
        n_pk = lib.witw_conv3x3_bf16_packed_elems(self.cout, self.cin)
        if reuse is not None and not (reuse.wpk.numel() == n_pk and reuse.wpk.device == w.device and reuse.cout == self.cout):
            reuse = None
        self.wpk = reuse.wpk if reuse is not None else torch.empty(n_pk, dtype=torch.bfloat16, device=w.device)
        _lib.check(lib.witw_conv3x3_bf16_pack_weights_ex(w.data_ptr(), self.wpk.data_ptr(), self.cout, self.cin,
                                                         int(bool(transpose_flip)), _stream()), 'witw_conv3x3_bf16_pack_weights_ex')
        self.bias = reuse.bias if reuse is not None else \
            torch.zeros(lib.witw_conv3x3_bias_floats(self.cout), dtype=torch.float32, device=w.device)
        if bias is not None:
            self.bias[:self.cout].copy_(bias.detach())

    @classmethod
    def batch(cls, items):
        """items = [(weight, bias or None, transpose_flip, reuse or None)] -> [PackedConvBf16], all filter images and bias copies
        written by ONE launch per 16 entries (witw_conv3x3_bf16_pack_weights_multi): what a bf16 training step re-packs after every
        Adam update. The same bits as the constructor entry by entry."""
        import ctypes
        lib = _lib.load()
        out, tab = [], []
        for weight, bias, transpose_flip, reuse in items:
            w = _dev_f32(weight.detach(), 'weight')
            pk = cls.__new__(cls)
            pk.cout, pk.cin = (w.shape[1], w.shape[0]) if transpose_flip else (w.shape[0], w.shape[1])
            pk.cin_pad = (pk.cin + 15) // 16 * 16
            n_pk = lib.witw_conv3x3_bf16_packed_elems(pk.cout, pk.cin)
            if reuse is not None and not (reuse.wpk.numel() == n_pk and reuse.wpk.device == w.device and reuse.cout == pk.cout):
                reuse = None
            pk.wpk = reuse.wpk if reuse is not None else torch.empty(n_pk, dtype=torch.bfloat16, device=w.device)
            pk.bias = reuse.bias if reuse is not None else \
                torch.zeros(lib.witw_conv3x3_bias_floats(pk.cout), dtype=torch.float32, device=w.device)
            b = None if bias is None else _dev_f32(bias.detach(), 'bias')
            if b is not None and b.numel() != pk.cout:
                raise _lib.WitwError('PackedConvBf16.batch: bias of %d elements for %d output channels' % (b.numel(), pk.cout))
            tab.append((w, pk.wpk, b, pk.bias, pk.cout, pk.cin, int(bool(transpose_flip))))
            out.append(pk)
        k = len(tab)
        if k:
            vp = ctypes.c_void_p
            _lib.check(lib.witw_conv3x3_bf16_pack_weights_multi(
                (vp * k)(*[t[0].data_ptr() for t in tab]), (vp * k)(*[t[1].data_ptr() for t in tab]),
                (vp * k)(*[None if t[2] is None else t[2].data_ptr() for t in tab]), (vp * k)(*[t[3].data_ptr() for t in tab]),
                (ctypes.c_int * k)(*[t[4] for t in tab]), (ctypes.c_int * k)(*[t[5] for t in tab]),
                (ctypes.c_int * k)(*[t[6] for t in tab]), k, _stream()), 'witw_conv3x3_bf16_pack_weights_multi')
        return out


def nchw_to_nhwc_bf16(x, cpad=16):
    lib = _lib.load()
    x = _dev_f32(x, 'x')
    B, C, H, W = x.shape
    y = torch.empty((B, H, W, cpad), dtype=torch.bfloat16, device=x.device)
    _lib.check(lib.witw_nchw_f32_to_nhwc_bf16(x.data_ptr(), y.data_ptr(), B, C, H, W, cpad, _stream()),
               'witw_nchw_f32_to_nhwc_bf16')
    return y


def conv3x3_bf16_fwd(x_nhwc, packed, stride_h=1, circular=False, relu=True, pool=False, out_nchw_f32=False, drop_scale=None,
                     gate=None, dilate_h=False, out_h=None, want_pool_code=False):
    """x NHWC bf16 [B,H,W,Cin_pad] -> NHWC bf16 [B,Hy,Wy,Cout] (or the fp32 NCHW embedding). Training extras as in
    conv3x3_fwd: drop_scale [B,Cout] fp32, gate = bf16 tensor shaped like the output (dgrad launches), dilate_h/out_h
    = zero-interleaved input rows (dgrad of a stride-(2,1) layer)."""
    lib = _lib.load()
    if not (x_nhwc.is_cuda and x_nhwc.dtype == torch.bfloat16 and x_nhwc.is_contiguous()):
        raise _lib.WitwError('conv3x3_bf16_fwd: x must be a contiguous bfloat16 GPU tensor')
    B, H, W, C = x_nhwc.shape
    if C != packed.cin_pad:
        raise _lib.WitwError('conv3x3_bf16_fwd: input has %d channels, packed weights expect %d' % (C, packed.cin_pad))
    if dilate_h:
        if out_h is None or (out_h - 1) // 2 + 1 != H:
            raise _lib.WitwError('conv3x3_bf16_fwd: dilate_h needs out_h with (out_h-1)//2+1 == %d physical rows' % H)
        h_phys, H = H, out_h          # logical (zero-interleaved) height
    Ho = (H + 2 - 3) // stride_h + 1
    Hy, Wy = (Ho // 2, W // 2) if pool else (Ho, W)
    if out_nchw_f32:
        y = torch.empty((B, packed.cout, Hy, Wy), dtype=torch.float32, device=x_nhwc.device)
    else:
        if packed.cout % 16:
            raise _lib.WitwError('conv3x3_bf16_fwd: a bf16 NHWC output needs Cout %% 16 == 0 (next layer\'s K chunk)')
        y = torch.empty((B, Hy, Wy, packed.cout), dtype=torch.bfloat16, device=x_nhwc.device)
    if drop_scale is not None:
        drop_scale = _dev_f32(drop_scale, 'drop_scale')
        if tuple(drop_scale.shape) != (B, packed.cout):
            raise _lib.WitwError('drop_scale must be [B,Cout]')
    if gate is not None:
        if not (gate.is_cuda and gate.dtype == torch.bfloat16 and gate.is_contiguous() and tuple(gate.shape) == tuple(y.shape)):
            raise _lib.WitwError('gate must be a contiguous bfloat16 GPU tensor with the output shape %s' % (tuple(y.shape),))
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    code = torch.empty(tuple(y.shape), dtype=torch.uint8, device=y.device) if (pool and want_pool_code) else None
    _lib.check(lib.witw_conv3x3_bf16_fwd_ex(x_nhwc.data_ptr(), packed.wpk.data_ptr(), packed.bias.data_ptr(), _p(drop_scale),
                                            _p(gate), y.data_ptr(), _p(code), B, H, W, C, packed.cout, stride_h, int(circular),
                                            int(relu), int(pool), int(out_nchw_f32), int(bool(dilate_h)), _stream()),
               'witw_conv3x3_bf16_fwd_ex')
    if prof is not None:
        e1.record()
        # the 64-input-channel layer runs on its own kernel (csrc/conv3x3_bf16_wres.hip): its own launch class
        kind = 'bf16_wres' if last_kernel_variant().startswith('conv3x3_bf16_wres_kernel') else 'bf16'
        prof.append(((kind, lib.witw_conv3x3_tile_n(packed.cout), stride_h, bool(pool)) + (('dil',) if dilate_h else ()),
                     2.0 * packed.cin * packed.cout * 9 * (h_phys if dilate_h else Ho) * W * B, e0, e1))      # dilated: see conv3x3_fwd
        if PROFILE_BY_KERNEL is not None:
            PROFILE_BY_KERNEL.setdefault(last_kernel_variant(), []).append((prof[-1][1], e0, e1))
    if want_pool_code:
        return y, code
    return y


def maxpool2x2_bwd_bf16(dy, code, out_hw):
    """dy (bf16) / code (uint8) [B,Hp,Wp,C] -> dx bf16 [B,H,W,C]: gradient routed to the recorded arg-max position."""
    lib = _lib.load()
    if not (dy.is_cuda and dy.dtype == torch.bfloat16 and dy.is_contiguous() and code.dtype == torch.uint8
            and tuple(code.shape) == tuple(dy.shape)):
        raise _lib.WitwError('maxpool2x2_bwd_bf16: dy must be contiguous bfloat16 on the GPU, code uint8 of the same shape')
    B, Hp, Wp, C = dy.shape
    H, W = out_hw
    dx = torch.empty((B, H, W, C), dtype=torch.bfloat16, device=dy.device)
    _lib.check(lib.witw_maxpool2x2_bwd_bf16(dy.data_ptr(), code.data_ptr(), dx.data_ptr(), B, Hp, Wp, H, W, C, _stream()),
               'witw_maxpool2x2_bwd_bf16')
    return dx


def nhwc_bf16_to_octet(x):
    """NHWC bf16 [B,H,W,C] -> batch-octet [ceil(B/8),H,W,C,8] (the operand layout of conv3x3_wgrad_bf16)."""
    lib = _lib.load()
    if not (x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and x.dim() == 4):
        raise _lib.WitwError('nhwc_bf16_to_octet: x must be a contiguous bfloat16 NHWC GPU tensor')
    B, H, W, C = x.shape
    y = torch.empty(((B + 7) // 8, H, W, C, 8), dtype=torch.bfloat16, device=x.device)
    _lib.check(lib.witw_nhwc_bf16_to_octet(x.data_ptr(), y.data_ptr(), B, H, W, C, _stream()), 'witw_nhwc_bf16_to_octet')
    return y


def conv3x3_wgrad_bf16(x_nhwc, dz_nhwc, cin_real, stride_h=1, circular=False, want_bias=True, x_oct=None, layout='nhwc', out=None):
    """bf16 MFMA weight gradient of one conv layer: x_nhwc [B,H,W,Cin] (its input), dz_nhwc [B,Ho,W,Cout] (gradient at
    its output), both bf16 NHWC -> (dW [Cout,cin_real,3,3] fp32, db [Cout] fp32 or None).
    layout='nhwc' (round 5, default): witw_conv3x3_wgrad_bf16_nhwc reads the two tensors as they are (pixels are the MFMA's k
    index, transposed LDS reads); layout='octet': the round-1 kernel on batch-octet copies of both operands (two
    nhwc_to_octet passes per call). Same arithmetic (bf16 products, fp32 accumulation), different summation order.
    out = (dW, db): write into these contiguous fp32 tensors (the .grad views of a parallel.GradBucket) instead of fresh ones."""
    lib = _lib.load()
    for t, n in ((x_nhwc, 'x'), (dz_nhwc, 'dz')):
        if not (t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous()):
            raise _lib.WitwError('conv3x3_wgrad_bf16: %s must be a contiguous bfloat16 GPU tensor' % n)
    if layout not in ('nhwc', 'octet'):
        raise _lib.WitwError("conv3x3_wgrad_bf16: layout must be 'nhwc' or 'octet'")
    B, H, W, Cin = x_nhwc.shape
    Ho = (H + 2 - 3) // stride_h + 1
    Cout = dz_nhwc.shape[3]
    if tuple(dz_nhwc.shape[:3]) != (B, Ho, W):
        raise _lib.WitwError('conv3x3_wgrad_bf16: dz %s does not match x %s (stride %d)' % (tuple(dz_nhwc.shape),
                                                                                          tuple(x_nhwc.shape), stride_h))
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if out is not None:
        dw, db = out
        if not (dw.is_contiguous() and tuple(dw.shape) == (Cout, cin_real, 3, 3) and dw.dtype == torch.float32 and dw.is_cuda
                and db is not None and db.is_contiguous() and db.numel() == Cout and db.dtype == torch.float32):
            raise _lib.WitwError('conv3x3_wgrad_bf16: out must be contiguous float32 GPU tensors [%d,%d,3,3] and [%d]' % (Cout, cin_real, Cout))
    else:
        dw = torch.empty((Cout, cin_real, 3, 3), dtype=torch.float32, device=x_nhwc.device)
        db = torch.empty((Cout,), dtype=torch.float32, device=x_nhwc.device) if want_bias else None
    if layout == 'nhwc':
        ws = torch.empty(lib.witw_conv3x3_wgrad_bf16_nhwc_workspace_floats(B, H, W, Cin, Cout, stride_h), dtype=torch.float32,
                         device=x_nhwc.device)
        _lib.check(lib.witw_conv3x3_wgrad_bf16_nhwc(x_nhwc.data_ptr(), dz_nhwc.data_ptr(), dw.data_ptr(), _p(db), ws.data_ptr(),
                                                    B, H, W, Cin, cin_real, Cout, stride_h, int(circular), 0, _stream()),
                   'witw_conv3x3_wgrad_bf16_nhwc')
    else:
        if x_oct is None:
            x_oct = nhwc_bf16_to_octet(x_nhwc)
        dz_oct = nhwc_bf16_to_octet(dz_nhwc)
        ws = torch.empty(lib.witw_conv3x3_wgrad_bf16_workspace_floats(B, H, W, Cin, Cout, stride_h), dtype=torch.float32,
                         device=x_nhwc.device)
        _lib.check(lib.witw_conv3x3_wgrad_bf16(x_oct.data_ptr(), dz_oct.data_ptr(), dw.data_ptr(), _p(db),
                                               ws.data_ptr(), B, H, W, Cin, cin_real, Cout, stride_h, int(circular), 0, _stream()),
                   'witw_conv3x3_wgrad_bf16')
    if prof is not None:
        e1.record()
        prof.append((('wgrad_bf16', stride_h), 2.0 * cin_real * Cout * 9 * Ho * W * B, e0, e1))
    return dw, db


# ----------------------------------------------------------------------------- fp16x3: fp32-grade inference on the fp16 MFMA
class PackedConvF16x3:
    """fp16 hi / lo filter packing of one 3x3 conv for the fp16x3 kernel (19 slots per 8-channel chunk) + fp32 bias.
    transpose_flip packs the dgrad filter; reuse = a previous packing of the same layer whose buffers are overwritten."""

    def __init__(self, weight, bias, transpose_flip=False, reuse=None):
        lib = _lib.load()
        w = _dev_f32(weight.detach(), 'weight')
        self.cout, self.cin = (w.shape[1], w.shape[0]) if transpose_flip else (w.shape[0], w.shape[1])
        self.cin_pad = (self.cin + 7) // 8 * 8
        n_pk = lib.witw_conv3x3_f16x3_packed_elems(self.cout, self.cin)
        if reuse is not None and not (reuse.wpk.numel() == n_pk and reuse.wpk.device == w.device and reuse.cout == self.cout):
            reuse = None
        self.wpk = reuse.wpk if reuse is not None else torch.empty(n_pk, dtype=torch.float16, device=w.device)
        _lib.check(lib.witw_conv3x3_f16x3_pack_weights_ex(w.data_ptr(), self.wpk.data_ptr(), self.cout, self.cin,
                                                          int(bool(transpose_flip)), _stream()), 'witw_conv3x3_f16x3_pack_weights_ex')
        self.bias = reuse.bias if reuse is not None else \
            torch.zeros(lib.witw_conv3x3_bias_floats(self.cout), dtype=torch.float32, device=w.device)
        if bias is not None:
            self.bias[:self.cout].copy_(bias.detach())


def nchw_to_split_f16(x, cpad=8):
    """NCHW fp32 -> split-fp16 NHWC [B,H,W,cpad/8,2,8] (hi plane, lo plane per 8 channels)."""
    lib = _lib.load()
    x = _dev_f32(x, 'x')
    B, C, H, W = x.shape
    y = torch.empty((B, H, W, cpad // 8, 2, 8), dtype=torch.float16, device=x.device)
    _lib.check(lib.witw_nchw_f32_to_split_f16(x.data_ptr(), y.data_ptr(), B, C, H, W, cpad, _stream()), 'witw_nchw_f32_to_split_f16')
    return y


def split_f16_to_f32(x_split):
    """split-fp16 NHWC [B,H,W,C/8,2,8] -> fp32 NHWC [B,H,W,C] (hi + lo)."""
    lib = _lib.load()
    B, H, W, C8 = x_split.shape[:4]
    y = torch.empty((B, H, W, C8 * 8), dtype=torch.float32, device=x_split.device)
    _lib.check(lib.witw_split_f16_to_f32(x_split.data_ptr(), y.data_ptr(), B * H * W, C8 * 8, _stream()), 'witw_split_f16_to_f32')
    return y


_F16X3_FLAGS = {}


def _f16x3_flag(device):
    """Per-device overflow flag of the fp16x3 kernels (set when an activation leaves the fp16 range)."""
    key = (device.type, device.index)
    if key not in _F16X3_FLAGS:
        _F16X3_FLAGS[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _F16X3_FLAGS[key]


def f16x3_overflowed(device=None, reset=True):
    """True if any fp16x3 conv since the last check produced a value beyond the fp16 range (|v| > 65504) or a NaN, which
    the hi + lo representation cannot carry (synchronises: call it at the end of an evaluation, not per step)."""
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    flag = _f16x3_flag(device)
    hit = bool(flag.item())
    if hit and reset:
        flag.zero_()
    return hit


def _is_split(t):
    return t.is_cuda and t.dtype == torch.float16 and t.is_contiguous() and t.dim() == 6 and tuple(t.shape[4:]) == (2, 8)


def conv3x3_f16x3_fwd(x_split, packed, stride_h=1, circular=False, relu=True, pool=False, out_nchw_f32=False, drop_scale=None,
                      gate=None, dilate_h=False, out_h=None, want_pool_code=False):
    """x split-fp16 NHWC [B,H,W,Cin/8,2,8] -> split-fp16 NHWC [B,Hy,Wy,Cout/8,2,8] (or the fp32 NCHW embedding). Training
    extras as in conv3x3_bf16_fwd: drop_scale [B,Cout] fp32, gate = split-fp16 tensor shaped like the output, dilate_h /
    out_h = zero-interleaved input rows (dgrad of a stride-(2,1) layer)."""
    lib = _lib.load()
    if not _is_split(x_split):
        raise _lib.WitwError('conv3x3_f16x3_fwd: x must be a contiguous float16 GPU tensor shaped [B,H,W,C/8,2,8]')
    B, H, W, C8 = x_split.shape[:4]
    C = C8 * 8
    if C != packed.cin_pad:
        raise _lib.WitwError('conv3x3_f16x3_fwd: input has %d channels, packed weights expect %d' % (C, packed.cin_pad))
    if dilate_h:
        if out_h is None or (out_h - 1) // 2 + 1 != H:
            raise _lib.WitwError('conv3x3_f16x3_fwd: dilate_h needs out_h with (out_h-1)//2+1 == %d physical rows' % H)
        h_phys, H = H, out_h
    Ho = (H + 2 - 3) // stride_h + 1
    Hy, Wy = (Ho // 2, W // 2) if pool else (Ho, W)
    if out_nchw_f32:
        y = torch.empty((B, packed.cout, Hy, Wy), dtype=torch.float32, device=x_split.device)
    else:
        if packed.cout % 8:
            raise _lib.WitwError('conv3x3_f16x3_fwd: a split-fp16 output needs Cout %% 8 == 0')
        y = torch.empty((B, Hy, Wy, packed.cout // 8, 2, 8), dtype=torch.float16, device=x_split.device)
    if drop_scale is not None:
        drop_scale = _dev_f32(drop_scale, 'drop_scale')
        if tuple(drop_scale.shape) != (B, packed.cout):
            raise _lib.WitwError('drop_scale must be [B,Cout]')
    if gate is not None and not (_is_split(gate) and tuple(gate.shape) == tuple(y.shape)):
        raise _lib.WitwError('gate must be a split-fp16 GPU tensor with the output shape %s' % (tuple(y.shape),))
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    code = torch.empty((B, Hy, Wy, packed.cout), dtype=torch.uint8, device=y.device) if (pool and want_pool_code) else None
    _lib.check(lib.witw_conv3x3_f16x3_fwd_ex(x_split.data_ptr(), packed.wpk.data_ptr(), packed.bias.data_ptr(), _p(drop_scale),
                                             _p(gate), y.data_ptr(), _p(code), _f16x3_flag(x_split.device).data_ptr(), B, H, W, C,
                                             packed.cout, stride_h, int(circular), int(relu), int(pool), int(out_nchw_f32),
                                             int(bool(dilate_h)), _stream()),
               'witw_conv3x3_f16x3_fwd_ex')
    if prof is not None:
        e1.record()
        prof.append((('f16x3', lib.witw_conv3x3_tile_n(packed.cout), stride_h, bool(pool)) + (('dil',) if dilate_h else ()),
                     2.0 * packed.cin * packed.cout * 9 * (h_phys if dilate_h else Ho) * W * B, e0, e1))      # dilated: see conv3x3_fwd
    if want_pool_code:
        return y, code
    return y


def maxpool2x2_bwd_split(dy_split, code, out_hw):
    """dy split-fp16 [B,Hp,Wp,C/8,2,8] + code uint8 [B,Hp,Wp,C] -> dx split-fp16 [B,H,W,C/8,2,8]."""
    lib = _lib.load()
    if not _is_split(dy_split) or code.dtype != torch.uint8:
        raise _lib.WitwError('maxpool2x2_bwd_split: dy must be a split-fp16 tensor and code uint8')
    B, Hp, Wp, C8 = dy_split.shape[:4]
    H, W = out_hw
    dx = torch.empty((B, H, W, C8, 2, 8), dtype=torch.float16, device=dy_split.device)
    _lib.check(lib.witw_maxpool2x2_bwd_split(dy_split.data_ptr(), code.data_ptr(), dx.data_ptr(), B, Hp, Wp, H, W, C8 * 8, _stream()),
               'witw_maxpool2x2_bwd_split')
    return dx


def split_f16_to_octet(x_split):
    """split-fp16 NHWC [B,H,W,C/8,2,8] -> batch-octet split [ceil(B/8),H,W,C,2,8] (operand layout of conv3x3_wgrad_f16x3)."""
    lib = _lib.load()
    if not _is_split(x_split):
        raise _lib.WitwError('split_f16_to_octet: x must be a contiguous float16 GPU tensor shaped [B,H,W,C/8,2,8]')
    B, H, W, C8 = x_split.shape[:4]
    y = torch.empty(((B + 7) // 8, H, W, C8 * 8, 2, 8), dtype=torch.float16, device=x_split.device)
    _lib.check(lib.witw_split_f16_to_octet(x_split.data_ptr(), y.data_ptr(), B, H, W, C8 * 8, _stream()), 'witw_split_f16_to_octet')
    return y


def conv3x3_wgrad_f16x3(x_split, dz_split, cin_real, stride_h=1, circular=False, want_bias=True):
    """Weight gradient with fp32-grade products on the fp16 MFMA: x_split [B,H,W,Cin/8,2,8] (the layer's input), dz_split
    [B,Ho,W,Cout/8,2,8] (gradient at its output) -> (dW [Cout,cin_real,3,3] fp32, db [Cout] fp32 or None)."""
    lib = _lib.load()
    if not (_is_split(x_split) and _is_split(dz_split)):
        raise _lib.WitwError('conv3x3_wgrad_f16x3: operands must be contiguous float16 GPU tensors shaped [B,H,W,C/8,2,8]')
    B, H, W, Ci8 = x_split.shape[:4]
    Cin, Cout = Ci8 * 8, dz_split.shape[3] * 8
    Ho = (H + 2 - 3) // stride_h + 1
    if tuple(dz_split.shape[:3]) != (B, Ho, W):
        raise _lib.WitwError('conv3x3_wgrad_f16x3: dz %s does not match x %s (stride %d)' % (tuple(dz_split.shape),
                                                                                           tuple(x_split.shape), stride_h))
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    x_oct, dz_oct = split_f16_to_octet(x_split), split_f16_to_octet(dz_split)
    dw = torch.empty((Cout, cin_real, 3, 3), dtype=torch.float32, device=x_split.device)
    db = torch.empty((Cout,), dtype=torch.float32, device=x_split.device) if want_bias else None
    ws = torch.empty(lib.witw_conv3x3_wgrad_f16x3_workspace_floats(B, H, W, Cin, Cout, stride_h), dtype=torch.float32,
                     device=x_split.device)
    _lib.check(lib.witw_conv3x3_wgrad_f16x3(x_oct.data_ptr(), dz_oct.data_ptr(), dz_split.data_ptr(), dw.data_ptr(), _p(db),
                                            ws.data_ptr(), B, H, W, Cin, cin_real, Cout, stride_h, int(circular), 0, _stream()),
               'witw_conv3x3_wgrad_f16x3')
    if prof is not None:
        e1.record()
        prof.append((('wgrad_f16x3', stride_h), 2.0 * cin_real * Cout * 9 * Ho * W * B, e0, e1))
    return dw, db


def conv4x4_to_k3(w, cpad):
    """Conv2d(ci, co, 4, 2) weight [co,ci,4,4] -> the 3x3 filter over space-to-depth(2) channels [co,cpad,3,3] (one launch)."""
    w = _dev_f32(w.detach(), 'weight')
    co, ci = w.shape[:2]
    k3 = torch.empty((co, cpad, 3, 3), dtype=torch.float32, device=w.device)
    _lib.check(_lib.load().witw_conv4x4_to_k3(w.data_ptr(), k3.data_ptr(), co, ci, cpad, _stream()), 'witw_conv4x4_to_k3')
    return k3


def k3_to_conv4x4(dk3, ci):
    """the inverse gather for a weight gradient: [co,cpad,3,3] -> [co,ci,4,4]"""
    dk3 = _dev_f32(dk3, 'dk3')
    co, cpad = dk3.shape[:2]
    dw = torch.empty((co, ci, 4, 4), dtype=torch.float32, device=dk3.device)
    _lib.check(_lib.load().witw_k3_to_conv4x4(dk3.data_ptr(), dw.data_ptr(), co, ci, cpad, _stream()), 'witw_k3_to_conv4x4')
    return dw


def bn_train_stats(a, valid_hw, gamma, beta, running_mean=None, running_var=None, eps=1e-5, momentum=0.1):
    """Batch statistics of BatchNorm2d over the valid region of a NHWC tensor -> (mean, invstd, scale, shift)."""
    lib = _lib.load()
    a = _dev_f32(a, 'a')
    B, Hp, Wp, C = a.shape
    H, W = valid_hw
    out = [torch.empty((C,), dtype=torch.float32, device=a.device) for _ in range(4)]
    ws = torch.empty(lib.witw_bn_workspace_floats(B, H, W, C), dtype=torch.float32, device=a.device)
    _lib.check(lib.witw_bn_train_stats(a.data_ptr(), B, Hp, Wp, H, W, C, gamma.data_ptr(), beta.data_ptr(), float(eps),
                                       float(momentum), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                                       out[3].data_ptr(), _p(running_mean), _p(running_var), ws.data_ptr(), _stream()),
               'witw_bn_train_stats')
    return out


def bn_lrelu_bwd(a, dy, valid_hw, mean, invstd, gamma, slope=0.2, dy_s2d=False):
    """-> (dz, dgamma, dbeta) for y = BatchNorm_train(LeakyReLU(z)), a = LeakyReLU(z). dy_s2d: dy is the space-to-depth image
    [B, ceil(H/2), ceil(W/2), >= 4C] the next block's data-gradient conv wrote, read in place (no depth_to_space2 pass)."""
    lib = _lib.load()
    a, dy = _dev_f32(a, 'a'), _dev_f32(dy, 'dy')
    B, Hp, Wp, C = a.shape
    H, W = valid_hw
    cp = 0
    if dy_s2d:
        cp = dy.shape[3]
        if tuple(dy.shape[:3]) != (B, (H + 1) // 2, (W + 1) // 2) or cp < 4 * C:
            raise _lib.WitwError('bn_lrelu_bwd: space-to-depth gradient %s does not match activation %s valid %s' % (tuple(dy.shape), tuple(a.shape), (H, W)))
    dz = torch.empty_like(a)
    dg = torch.empty((C,), dtype=torch.float32, device=a.device)
    db = torch.empty((C,), dtype=torch.float32, device=a.device)
    ws = torch.empty(lib.witw_bn_workspace_floats(B, H, W, C), dtype=torch.float32, device=a.device)
    _lib.check(lib.witw_bn_lrelu_bwd_ex(a.data_ptr(), dy.data_ptr(), dz.data_ptr(), dg.data_ptr(), db.data_ptr(), mean.data_ptr(),
                                        invstd.data_ptr(), gamma.data_ptr(), B, Hp, Wp, H, W, C, float(slope), cp, ws.data_ptr(),
                                        _stream()), 'witw_bn_lrelu_bwd_ex')
    return dz, dg, db


def depth_to_space2(g, like, valid_hw, add=None):
    lib = _lib.load()
    g = _dev_f32(g, 'g')
    B, Hp, Wp, C = like.shape
    H, W = valid_hw
    dx = torch.empty((B, Hp, Wp, C), dtype=torch.float32, device=g.device)
    _lib.check(lib.witw_depth_to_space2(g.data_ptr(), _p(add), dx.data_ptr(), B, Hp, Wp, H, W, C, g.shape[3], _stream()),
               'witw_depth_to_space2')
    return dx


def gem_pool_bwd(a, scale, shift, f, df, valid_hw, col0, p=3., out=None):
    lib = _lib.load()
    a = _dev_f32(a, 'a')
    B, Hp, Wp, C = a.shape
    H, W = valid_hw
    acc = out is not None
    dy = out if acc else torch.empty_like(a)
    _lib.check(lib.witw_gem_pool_bwd(a.data_ptr(), scale.data_ptr(), shift.data_ptr(), f.data_ptr(), df.data_ptr(), dy.data_ptr(),
                                     B, Hp, Wp, H, W, C, f.shape[1], col0, float(p), int(acc), _stream()), 'witw_gem_pool_bwd')
    return dy


def embed_normalize_bwd(g, df):
    lib = _lib.load()
    g, df = _dev_f32(g, 'g'), _dev_f32(df, 'df')
    dg = torch.empty_like(g)
    _lib.check(lib.witw_embed_normalize_bwd(g.data_ptr(), df.data_ptr(), dg.data_ptr(), g.shape[0], g.shape[1], _stream()),
               'witw_embed_normalize_bwd')
    return dg


def exhaustive_triplet_loss_bwd(e1, e2, D, grad_loss, soft_margin=False, alpha=10., margin=1.):
    lib = _lib.load()
    e1, e2, D = _dev_f32(e1, 'e1'), _dev_f32(e2, 'e2'), _dev_f32(D, 'D')
    gl = _dev_f32(grad_loss.reshape(1).contiguous(), 'grad_loss')
    B, n = e1.shape
    de1, de2 = torch.empty_like(e1), torch.empty_like(e2)
    ws = torch.empty((B * B,), dtype=torch.float32, device=e1.device)
    _lib.check(lib.witw_exhaustive_triplet_loss_bwd(e1.data_ptr(), e2.data_ptr(), D.data_ptr(), gl.data_ptr(), de1.data_ptr(),
                                                    de2.data_ptr(), B, n, int(soft_margin), float(alpha), float(margin),
                                                    ws.data_ptr(), _stream()), 'witw_exhaustive_triplet_loss_bwd')
    return de1, de2
