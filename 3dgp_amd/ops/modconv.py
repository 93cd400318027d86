"""Modulated convolution on the fp32 matrix cores (forward).

`modulated_conv2d` keeps the reference's signature (src/training/networks_stylegan2.py:31-43) and runs the
native kernel `tdgp_modconv2d` (include/tdgp.h) -- the reference has no native op here, it issues cuDNN
grouped convolutions from Python (conv2d_resample.py / conv2d_gradfix.py:113-115).

`PackedConv` / `synthesis_layer` / `torgb_layer` are the fused forms the generator uses: modulation +
convolution (+ x2 FIR upsampling) + demodulation + noise + bias + activation in one native call, i.e. the whole
of SynthesisLayer.forward (:128-145) / ToRGBLayer.forward + skip add (:168-172, :265-269).
"""
import ctypes
import weakref

import numpy as np
import torch

from .. import _lib
from .bias_act import activation_funcs

# Per-tensor caches are keyed by the tensor OBJECT (weak reference) and validated with (data_ptr, _version): an address
# alone is not an identity -- the caching allocator hands a freed block to the next tensor of the same size.
_FIR_HOST_CACHE = {}
_PACK_CACHE = {}


def _cached(cache, tensor, make):
    """cache: {id(tensor): (weakref, stamp, value)}; entries die with their tensor (tensors cannot key a WeakKeyDictionary:
    weakref equality falls back to elementwise tensor ==)."""
    stamp = (tensor.data_ptr(), tensor._version, tuple(tensor.shape))
    key = id(tensor)
    hit = cache.get(key)
    if hit is not None and hit[0]() is tensor and hit[1] == stamp:
        return hit[2]
    val = make(tensor)
    cache[key] = (weakref.ref(tensor, lambda _r, k=key, c=cache: c.pop(k, None)), stamp, val)
    return val


def fir_host_array(resample_filter):
    """Host copy of the 4x4 resample filter as a ctypes float[16] (the C ABI takes it as a host pointer)."""
    if resample_filter is None:
        return None
    if isinstance(resample_filter, np.ndarray):
        f = np.ascontiguousarray(resample_filter, dtype=np.float32)
    else:
        f = _cached(_FIR_HOST_CACHE, resample_filter, lambda t: np.ascontiguousarray(t.detach().float().cpu().numpy()))
    if f.shape != (4, 4):
        raise NotImplementedError(f'modulated_conv2d: resample_filter of shape {tuple(f.shape)} (the generator path uses 4x4)')
    return (ctypes.c_float * 16)(*f.reshape(-1).tolist())


def fold_up2_table(fir4x4):
    """P[py, px, i, j, a, e] with  y[2A + py, 2B + px] = sum_{i,j in {-1,0,1}} x[A + i, B + j] * sum_{a,e} P[py,px,i+1,j+1,a,e] * w[a,e]
    for the x2 layer of conv2d_resample.py:108-125 with flip_weight=False (networks_stylegan2.py:138): z[2m + a, 2n + e] += x[m,n] w[a,e]
    (conv_transpose2d, stride 2, no padding), then y[Y,X] = 4 sum_{ky,kx} F[ky,kx] z[Y + ky - 1, X + kx - 1] with F = the flipped filter
    (upfirdn2d, padding 1, gain up^2).  With Y = 2A + py, m = A + i:  a = py + ky - 1 - 2i, i.e. ky = a + 1 + 2i - py (and the same along x).
    Every output parity is a 3x3 'same' correlation of x -- four stride-1 layers sharing one input, which is what the F(4x4) kernels run."""
    f = np.asarray(fir4x4, np.float64).reshape(4, 4)[::-1, ::-1]          # upfirdn2d without flip_filter convolves: taps = flipped f
    P = np.zeros((2, 2, 3, 3, 3, 3))
    for py in range(2):
        for px in range(2):
            for i in (-1, 0, 1):
                for j in (-1, 0, 1):
                    for a in range(3):
                        for e in range(3):
                            ky, kx = a + 1 + 2 * i - py, e + 1 + 2 * j - px
                            if 0 <= ky < 4 and 0 <= kx < 4:
                                P[py, px, i + 1, j + 1, a, e] = 4.0 * f[ky, kx]
    return P


class PackedConv:
    """Weights of one conv layer packed for the MFMA kernel (done once; weights are static at inference)."""

    def __init__(self, weight, wsq_from=None):
        _lib.require_cuda(weight, 'weight')
        self.weight = weight                  # the tensor this pack was made from (folded x2 form below; kept alive by the cache entry anyway)
        self._folded = {}
        cout, cin, kh, kw = weight.shape
        if kh != kw or kh not in (1, 3, 5):
            raise NotImplementedError(f'modulated_conv2d: {kh}x{kw} kernels are not on the generator path (1x1 / 3x3 / 5x5)')
        self.cout, self.cin, self.k = int(cout), int(cin), int(kh)
        nbytes = _lib.load().tdgp_modconv_pack_bytes(self.cout, self.cin, self.k)
        self.buf = torch.empty(nbytes // 4, dtype=torch.float32, device=weight.device)
        w = _lib.f32c(weight.detach())
        with torch.cuda.device(weight.device):
            _lib.call('tdgp_modconv_pack', w.data_ptr(), self.buf.data_ptr(), self.cout, self.cin, self.k, _lib.stream_of(w))
        if wsq_from is not None:
            # folded x2 weights: the demodulation sums are those of the ORIGINAL 3x3 weights (networks_stylegan2.py:62), one copy per parity
            off = int(_lib.load().tdgp_modconv_wsq_offset(self.cout, self.cin, self.k)) // 4          # bytes -> floats
            coutp = (self.cout + 3) // 4 * 4
            src_off = int(_lib.load().tdgp_modconv_wsq_offset(wsq_from.cout, wsq_from.cin, wsq_from.k)) // 4
            src_p = (wsq_from.cout + 3) // 4 * 4
            src = wsq_from.buf[src_off:src_off + self.cin * src_p].view(self.cin, src_p)[:, :wsq_from.cout]
            self.buf[off:off + self.cin * coutp].view(self.cin, coutp)[:, :self.cout] = src.repeat_interleave(4, dim=1)

    def folded_up2(self, fir4x4_host):
        """The x2 form of this 3x3 layer as four parity kernels (fold_up2_table): PackedConv of [4 Cout, Cin, 3, 3], channel 4 o + 2 py + px."""
        key = bytes(fir4x4_host)
        hit = self._folded.get(key)
        if hit is None:
            P = torch.as_tensor(fold_up2_table(np.frombuffer(key, np.float32)), device=self.weight.device)
            w64 = self.weight.detach().double()
            weff = torch.einsum('pqijab,ocab->opqcij', P, w64).reshape(4 * self.cout, self.cin, 3, 3).float().contiguous()
            hit = self._folded[key] = PackedConv(weff, wsq_from=self)
        return hit


import os as _os
FOLD_UP2 = _os.environ.get('TDGP_FOLD_UP2', '1') != '0'       # module switch for A/B runs and tests: False keeps every x2 layer on the transposed-convolution + FIR kernels


def demod_coefficients(packed, styles):
    """d[b,o] = rsqrt(sum_c s[b,c]^2 wsq[c,o] + 1e-8) (networks_stylegan2.py:62) from the pack's own sums, through the library's kernel."""
    B = styles.shape[0]
    lib = _lib.load()
    coutp = (packed.cout + 3) // 4 * 4
    meta = torch.tensor([[wsq_address(packed), 0, packed.cin, packed.cout, coutp, 0]], dtype=torch.int64, device=styles.device)
    return demod_batch(_lib.f32c(styles).reshape(-1), meta, B * packed.cout, B, packed.cout).view(B, packed.cout)


def _packed(weight):
    """PackedConv of a weight tensor, re-packed when the tensor is written in place (load_state_dict) or moved."""
    return _cached(_PACK_CACHE, weight, PackedConv)


def modconv_forward(x, packed, styles, noise=None, bias=None, up=1, demodulate=True, act='linear', alpha=None, gain=None, clamp=None,
                    fir=None, skip=None, out_layout=0, out_feat=0, dcoef=None):
    """One call of tdgp_modconv2d (x fp32) / tdgp_modconv2d_bf16 (x bf16: the reduced-precision blocks).  x [B,Cin,H,W] NCHW; returns [B,Cout,H*up,W*up] (out_layout 0) or the
    channel-last plane tensor [B,Cout/out_feat,H,W,out_feat] (out_layout 1).  `dcoef`: demodulation coefficients [B,Cout] precomputed by
    `demod_batch` (else the call computes them)."""
    _lib.require_cuda(x, 'x')
    bf16 = x.dtype == torch.bfloat16
    x = x.contiguous() if bf16 else _lib.f32c(x)
    B, cin, H, W = x.shape
    if cin != packed.cin:
        raise RuntimeError(f'modulated_conv2d: x has {cin} channels, weight expects {packed.cin}')
    spec = activation_funcs[act]
    alpha = float(spec.def_alpha if alpha is None else alpha)
    gain = float(spec.def_gain if gain is None else gain)
    clamp = float(-1 if clamp is None else clamp)
    if styles is not None:
        styles = _lib.f32c(styles)
        if tuple(styles.shape) != (B, cin):
            raise RuntimeError(f'modulated_conv2d: styles must be [{B},{cin}], got {tuple(styles.shape)}')
    nbs = 0
    if noise is not None:
        noise = _lib.f32c(noise)
        if noise.numel() == H * up * W * up:
            nbs = 0
        elif noise.numel() == B * H * up * W * up:
            nbs = H * up * W * up
        else:
            raise RuntimeError(f'modulated_conv2d: noise of shape {tuple(noise.shape)} does not broadcast to [B,1,{H * up},{W * up}]')
    if bias is not None:
        bias = _lib.f32c(bias)
    if skip is not None:
        skip = _lib.f32c(skip)
    cout = packed.cout
    lib = _lib.load()
    if (up == 2 and packed.k == 3 and not bf16 and fir is not None and skip is None and out_layout == 0 and FOLD_UP2 and cin >= 64 and H * W >= 512 and
            lib.tdgp_modconv2d_takes_folded_up2(B, cin, 4 * cout, H, W) and (noise is None or (noise.data_ptr() % 16 == 0 and nbs % 4 == 0))):
        # x2 layer: FIR folded into four parity kernels on the Winograd F(4x4) path (csrc/modconv_wino4.inc) -- the C side was ASKED first
        # (tdgp_modconv2d_takes_folded_up2: no fold, no pack, no exception for launches it does not take); TDGP_EUNSUPPORTED stays handled
        # for the conditions only the call itself can see
        pk2 = packed.folded_up2(fir)
        rep = lambda t: None if t is None else t.repeat_interleave(4, dim=-1).contiguous()      # noqa: E731   per-parity copies of [.., Cout] vectors
        if demodulate and dcoef is None:
            dcoef = demod_coefficients(packed, styles)
        ws2 = lib.tdgp_modconv2d_workspace_bytes(B, cin, 4 * cout, H, W, 3, 1)
        wsb = torch.empty(max(ws2, 4) // 4, dtype=torch.float32, device=x.device)
        y = torch.empty([B, cout, 2 * H, 2 * W], dtype=torch.float32, device=x.device)
        d4, b4 = (rep(dcoef) if demodulate else None), rep(bias)      # named: a temporary inside the argument list is freed (and its block re-used) before the call
        try:
            with torch.cuda.device(x.device):
                _lib.call('tdgp_modconv2d', x.data_ptr(), pk2.buf.data_ptr(), _lib.ptr(styles), _lib.ptr(d4), _lib.ptr(noise), nbs,
                          _lib.ptr(b4), None, None, y.data_ptr(), B, cin, 4 * cout, H, W, 3, 1, int(bool(demodulate)), spec.cuda_idx, alpha, gain, clamp,
                          2, 0, wsb.data_ptr(), ws2, _lib.stream_of(x))
            return y
        except _lib.Unsupported:
            del y, wsb
    ws_bytes = lib.tdgp_modconv2d_workspace_bytes(B, cin, cout, H, W, packed.k, up)
    ws = torch.empty(max(ws_bytes, 4) // 4, dtype=torch.float32, device=x.device)
    if bf16:
        # reduced-precision block (BASELINE configs[4]): bf16 activations in, bf16 out -- except the ToRGB form, whose output joins the
        # fp32 skip image (networks_stylegan2.py:268).  Shapes the bf16 MFMA kernels do not take are widened and run on the fp32 path.
        rgb = out_layout == 1
        # the ToRGB form (1x1, no demodulation; `skip` = the running fp32 image it joins): its result is fp32 in EITHER layout --
        # `y.to(float32); img.add_(y)`, networks_stylegan2.py:268-269 -- so the NCHW fallback must not round the accumulated image to bf16
        rgb_form = rgb or (packed.k == 1 and not demodulate) or skip is not None
        # mirror of the acceptance tests of tdgp_modconv2d_bf16 (modconv.hip): anything else is widened and runs on the fp32 kernels
        if rgb:
            takes = (packed.k == 1 and up == 1 and cout <= 96 and out_feat > 0 and out_feat % 4 == 0 and cout % out_feat == 0 and (H * W) % 4 == 0 and
                     (skip is None or (H % 2 == 0 and W % 2 == 0)))
        else:
            takes = (packed.k == 3 and not rgb_form and cin % 32 == 0 and cin <= 2048 and ((up == 2 and W % 2 == 0) or (up == 1 and W % 32 == 0)))
        def widened():
            y = modconv_forward(x.float(), packed, styles, noise=noise, bias=bias, up=up, demodulate=demodulate, act=act, alpha=alpha, gain=gain,
                                clamp=None if clamp < 0 else clamp, fir=fir, skip=skip, out_layout=out_layout, out_feat=out_feat, dcoef=dcoef)
            return y if rgb_form else y.to(torch.bfloat16)
        if not takes:
            return widened()
        y = (torch.empty([B, cout // out_feat, H, W, out_feat], dtype=torch.float32, device=x.device) if rgb else
             torch.empty([B, cout, H * up, W * up], dtype=torch.bfloat16, device=x.device))
        try:
            with torch.cuda.device(x.device):
                _lib.call('tdgp_modconv2d_bf16', x.data_ptr(), packed.buf.data_ptr(), _lib.ptr(styles), _lib.ptr(dcoef), _lib.ptr(noise), nbs, _lib.ptr(bias), fir,
                          _lib.ptr(skip), y.data_ptr(), B, cin, cout, H, W, packed.k, up, int(bool(demodulate)), spec.cuda_idx, alpha, gain, clamp,
                          out_layout, out_feat, ws.data_ptr(), ws_bytes, _lib.stream_of(x))
        except _lib.Unsupported:          # the C side's own acceptance tests (e.g. the x2 kernel's LDS budget at large Cin x B): nothing was launched
            return widened()
        return y
    if out_layout == 0:
        y = torch.empty([B, cout, H * up, W * up], dtype=torch.float32, device=x.device)
    else:
        y = torch.empty([B, cout // out_feat, H, W, out_feat], dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.call('tdgp_modconv2d', x.data_ptr(), packed.buf.data_ptr(), _lib.ptr(styles), _lib.ptr(dcoef), _lib.ptr(noise), nbs, _lib.ptr(bias), fir,
                  _lib.ptr(skip), y.data_ptr(), B, cin, cout, H, W, packed.k, up, int(bool(demodulate)), spec.cuda_idx, alpha, gain, clamp,
                  out_layout, out_feat, ws.data_ptr(), ws_bytes, _lib.stream_of(x))
    return y


def modulated_conv2d(x, weight, styles, noise=None, up=1, down=1, padding=0, resample_filter=None, demodulate=True, flip_weight=True,
                     fused_modconv=True):
    """Signature of the reference's modulated_conv2d (networks_stylegan2.py:31-43).

    Supported forms = the ones the generator forward issues (k in {1,3}; 5 for the depth adaptor) with padding == k // 2, down == 1,
    up == 1 (flip_weight=True, correlation) or up == 2 (flip_weight=False, 4x4 resample_filter).  Both values of
    `fused_modconv` give the same result (the two branches of the reference are algebraically identical).
    """
    _lib.require_cuda(x, 'x')
    cout, cin, kh, kw = weight.shape
    if x.ndim != 4 or x.shape[1] != cin:
        raise RuntimeError(f'modulated_conv2d: x must be [B,{cin},H,W], got {tuple(x.shape)}')
    if down != 1:
        raise NotImplementedError('modulated_conv2d: down > 1 is not on the generator path')
    if padding != kh // 2:
        raise NotImplementedError(f'modulated_conv2d: padding={padding} with a {kh}x{kw} kernel (generator path uses k // 2)')
    if up not in (1, 2) or (up == 2 and kh != 3):
        raise NotImplementedError(f'modulated_conv2d: up={up} with a {kh}x{kw} kernel')
    if kh > 1 and bool(flip_weight) != (up == 1):
        raise NotImplementedError('modulated_conv2d: flip_weight must be (up == 1) as in SynthesisLayer.forward (networks_stylegan2.py:138)')
    if x.dtype != torch.float32:
        raise NotImplementedError('modulated_conv2d: fp32 only (fp32_only=true in configs/model/3dgp.yaml)')
    fir = fir_host_array(resample_filter) if up == 2 else None
    return modconv_forward(x, _packed(weight), styles, noise=noise, bias=None, up=up, demodulate=demodulate, act='linear', gain=1.0, fir=fir)


class _ModulatedConv2dSame(torch.autograd.Function):
    """Autograd of the stride-1 modulated convolution (the unfused algebra of networks_stylegan2.py:60-80 under torch autograd):
        y = conv(x * s, w) * d,     d = rsqrt(sum_{c,k} (w s)^2 + 1e-8)   (demodulate)   or   d = 1
    forward        tdgp_modconv2d
    dx             tdgp_modconv2d on dy with the flipped, transposed weights, modulated by d on its input side, times s
    dw             tdgp_conv2d_weight_grad(x * s, dy * d)  +  the demodulation term  -w (dd d^3)^T s^2
    ds             sum_{yx} (dx / s-side) x                +  the demodulation term  -s ((dd d^3) W2),   W2 = sum_k w^2
    with dd[b,o] = sum_{yx} dy y / d.  The [B,C]-sized products are eager tensor ops, the convolutions are the HIP kernels."""

    @staticmethod
    def forward(ctx, x, weight, styles, demodulate):
        y = modconv_forward(x, _packed(weight), styles, demodulate=demodulate, act='linear', gain=1.0)
        ctx.demodulate = demodulate
        ctx.save_for_backward(x, weight, styles, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import conv2d_gradfix as _cg
        x, weight, styles, y = ctx.saved_tensors
        dy = dy.contiguous()
        k = weight.shape[2]
        s4 = styles[:, :, None, None]
        if ctx.demodulate:
            w2 = weight.detach().square().sum([2, 3])                                    # [O,C]
            d = (styles.square() @ w2.t() + 1e-8).rsqrt()                                # [B,O]
        else:
            d = None
        wt = weight.detach().flip([2, 3]).transpose(0, 1).contiguous()
        dxm = modconv_forward(dy, PackedConv(wt), d, demodulate=False, act='linear', gain=1.0)       # conv_T(dy * d, w)
        dx = dxm * s4 if ctx.needs_input_grad[0] else None
        ds = (dxm * x).sum([2, 3]) if ctx.needs_input_grad[2] else None
        dw = None
        dyd = dy * d[:, :, None, None] if d is not None else dy
        if ctx.needs_input_grad[1] and not _cg.weight_gradients_disabled:
            dw = _cg.conv2d_weight_grad(x * s4, dyd, weight.shape, stride=1, padding=k // 2)
        if d is not None:
            t = (dy * y).sum([2, 3]) / d * d.pow(3)                                      # dd * d^3, [B,O]
            if ds is not None:
                ds = ds - styles * (t @ w2)
            if dw is not None:
                dw = dw - weight * (t.t() @ styles.square())[:, :, None, None]
        return dx, dw, ds, None


class _ModulatedConv2dUp(torch.autograd.Function):
    """Autograd of the x2-upsampling modulated convolution of SynthesisLayer (conv2d_resample.py:109-122 under torch autograd):
        z = conv_transpose2d(x * s, w^T, stride 2)   [2H+1]      y = upfirdn2d(z, f, padding 1, gain 4) * d
    forward   tdgp_modconv2d (polyphase transposed conv + FIR, fused)
    dz        tdgp_upfirdn2d on dy * d with the flipped filter and padding 2 (upfirdn2d.py:251-265)
    dx        tdgp_conv2d(dz, w^T as [Cin,Cout,3,3], stride 2) * s        -- the adjoint of the transposed convolution
    dw        tdgp_conv2d_weight_grad with the roles swapped (`dy` := x * s, `x` := dz, stride 2), transposed back; + demodulation term
    ds        sum_{yx} (dx-side) x + demodulation term (as in the stride-1 function)."""

    @staticmethod
    def forward(ctx, x, weight, styles, fir):
        y = modconv_forward(x, _packed(weight), styles, up=2, demodulate=True, act='linear', gain=1.0, fir=fir_host_array(fir))
        ctx.save_for_backward(x, weight, styles, y, fir)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import conv2d_gradfix as _cg
        from . import upfirdn2d as _u
        x, weight, styles, y, fir = ctx.saved_tensors
        dy = dy.contiguous()
        s4 = styles[:, :, None, None]
        w2 = weight.detach().square().sum([2, 3])
        d = (styles.square() @ w2.t() + 1e-8).rsqrt()
        dz = _u.upfirdn2d(dy * d[:, :, None, None], fir, padding=2, flip_filter=True, gain=4)                # [B,Cout,2H+1,2W+1]
        wt = weight.detach().transpose(0, 1).contiguous()                                                    # [Cin,Cout,3,3]
        dxm = _cg.conv2d_strided(dz, wt, stride=2, padding=0)
        dx = dxm * s4 if ctx.needs_input_grad[0] else None
        ds = (dxm * x).sum([2, 3]) if ctx.needs_input_grad[2] else None
        dw = None
        if ctx.needs_input_grad[1] and not _cg.weight_gradients_disabled:
            dw = _cg.conv2d_weight_grad(dz, x * s4, wt.shape, stride=2, padding=0).transpose(0, 1)
        t = (dy * y).sum([2, 3]) / d * d.pow(3)
        if ds is not None:
            ds = ds - styles * (t @ w2)
        if dw is not None:
            dw = dw - weight * (t.t() @ styles.square())[:, :, None, None]
        return dx, dw, ds, None


def modulated_conv2d_up_autograd(x, weight, styles, resample_filter):
    """`modulated_conv2d(x, weight, styles, up=2, padding=1, resample_filter=f, flip_weight=False)` (3x3, demodulated, no noise) with
    gradients w.r.t. x, weight and styles on the HIP kernels."""
    return _ModulatedConv2dUp.apply(x, weight, styles, resample_filter)


def modulated_conv2d_autograd(x, weight, styles, demodulate=True):
    """`modulated_conv2d(x, weight, styles, padding=k // 2, demodulate=...)` (up = down = 1, no noise) with gradients w.r.t. x, weight
    and styles on the HIP kernels; the noise / bias / activation that follow it in a SynthesisLayer have their own
    differentiable ops (`fma`, `bias_act`)."""
    return _ModulatedConv2dSame.apply(x, weight, styles, bool(demodulate))


def wsq_address(packed):
    """Device address of a PackedConv's sum_tap W^2 table (the input of the demodulation)."""
    return packed.buf.data_ptr() + int(_lib.load().tdgp_modconv_wsq_offset(packed.cout, packed.cin, packed.k))


def demod_batch(styles_all, meta, total_floats, B, max_cout):
    """All demodulation coefficients of a forward in one launch.  meta: int64 [L,6] device tensor (see include/tdgp.h)."""
    out = torch.empty([total_floats], dtype=torch.float32, device=styles_all.device)
    with torch.cuda.device(styles_all.device):
        _lib.call('tdgp_demod_batch', styles_all.data_ptr(), meta.data_ptr(), out.data_ptr(), B, meta.shape[0], max_cout, _lib.stream_of(styles_all))
    return out
