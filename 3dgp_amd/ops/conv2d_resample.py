"""conv2d_resample: 2-D convolution with optional FIR up/down-sampling (forward).

Signature of the reference's `conv2d_resample(x, w, f, up, down, padding, groups, flip_weight, flip_filter)`
(src/torch_utils/ops/conv2d_resample.py:46).  On the generator path this algebra is executed inside
`tdgp_modconv2d`; the stand-alone function is kept for the callers either side of the path (the discriminator's
Conv2dLayer) and composes the native upfirdn2d with `conv2d_gradfix.conv2d` (HIP kernels for the forms the
generator / discriminator use; the reference's own `torch.nn.functional` fallback otherwise, conv2d_gradfix.py:36-44).
"""
import torch

from . import upfirdn2d as _u


def _conv(x, w, stride=1, padding=0, groups=1, transpose=False, correlate=True):
    # F.conv2d correlates; a true convolution flips the taps first (conv2d_resample.py:29-41)
    if not correlate and (w.shape[2] > 1 or w.shape[3] > 1):
        w = w.flip([2, 3])
    from . import conv2d_gradfix as _cg                   # conv2d_resample.py:29-41 -> conv2d_gradfix: HIP kernels for the native forms
    fn = _cg.conv_transpose2d if transpose else _cg.conv2d
    return fn(x, w, stride=stride, padding=padding, groups=groups)


def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, flip_filter=False):
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    assert isinstance(w, torch.Tensor) and w.ndim == 4 and w.dtype == x.dtype
    assert f is None or (isinstance(f, torch.Tensor) and f.ndim in [1, 2] and f.dtype == torch.float32)
    assert isinstance(up, int) and up >= 1 and isinstance(down, int) and down >= 1 and isinstance(groups, int) and groups >= 1
    cout, cin_g, kh, kw = (int(v) for v in w.shape)
    fw, fh = _u._filter_wh(f)
    px0, px1, py0, py1 = _u._quad(padding)
    if up > 1:      # the resampling filters extend the footprint: fold their half-widths into the padding
        px0, px1, py0, py1 = px0 + (fw + up - 1) // 2, px1 + (fw - up) // 2, py0 + (fh + up - 1) // 2, py1 + (fh - up) // 2
    if down > 1:
        px0, px1, py0, py1 = px0 + (fw - down + 1) // 2, px1 + (fw - down) // 2, py0 + (fh - down + 1) // 2, py1 + (fh - down) // 2
    pads = [px0, px1, py0, py1]
    pointwise = kh == 1 and kw == 1
    if pointwise and down > 1 and up == 1:            # filter+decimate first, then the cheap 1x1 conv
        x = _u.upfirdn2d(x, f, down=down, padding=pads, flip_filter=flip_filter)
        return _conv(x, w, groups=groups, correlate=flip_weight)
    if pointwise and up > 1 and down == 1:            # 1x1 conv first, then upsample
        x = _conv(x, w, groups=groups, correlate=flip_weight)
        return _u.upfirdn2d(x, f, up=up, padding=pads, gain=up ** 2, flip_filter=flip_filter)
    if down > 1 and up == 1:                          # low-pass, then strided conv
        x = _u.upfirdn2d(x, f, padding=pads, flip_filter=flip_filter)
        return _conv(x, w, stride=down, groups=groups, correlate=flip_weight)
    if up > 1:                                        # stride-`up` transposed conv, then low-pass (+ optional decimation)
        if groups == 1:
            wt = w.transpose(0, 1)
        else:
            wt = w.reshape(groups, cout // groups, cin_g, kh, kw).transpose(1, 2).reshape(groups * cin_g, cout // groups, kh, kw)
        px0, px1, py0, py1 = px0 - (kw - 1), px1 - (kw - up), py0 - (kh - 1), py1 - (kh - up)
        pxt, pyt = max(min(-px0, -px1), 0), max(min(-py0, -py1), 0)
        x = _conv(x, wt, stride=up, padding=[pyt, pxt], groups=groups, transpose=True, correlate=not flip_weight)
        x = _u.upfirdn2d(x, f, padding=[px0 + pxt, px1 + pxt, py0 + pyt, py1 + pyt], gain=up ** 2, flip_filter=flip_filter)
        return _u.upfirdn2d(x, f, down=down, flip_filter=flip_filter) if down > 1 else x
    if px0 == px1 and py0 == py1 and px0 >= 0 and py0 >= 0:     # plain convolution
        return _conv(x, w, padding=[py0, px0], groups=groups, correlate=flip_weight)
    x = _u.upfirdn2d(x, (f if up > 1 else None), up=up, padding=pads, gain=up ** 2, flip_filter=flip_filter)
    x = _conv(x, w, groups=groups, correlate=flip_weight)
    return _u.upfirdn2d(x, f, down=down, flip_filter=flip_filter) if down > 1 else x
