"""HIP path (through the C ABI) vs the CPU oracle and vs golden vectors captured from the reference.

Bars (stated per test):
  * integer rows (bilinear tap indices, searchsorted indices, sort permutation): bit-exact on identical inputs;
  * fp32 elementwise chains that decide those integers (stratified samples, ray points): bit-exact;
  * reductions (conv / FIR / dot products / compositing): relative to the tensor scale, <= 2e-6 .. 1e-5;
  * end-to-end RGB: the north-star 1e-4 max-rel (see conftest.assert_image_parity for the exact definition).
"""
import os

import numpy as np
import pytest
import torch

from conftest import (RGB_TOL, assert_close, assert_close_up_to_threshold_flips, assert_image_parity, assert_inds_mismatches_in_window, load_golden,
                      report_parity)

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


def T(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(DEV)


def N(t):
    return t.detach().float().cpu().numpy()


@pytest.fixture(scope='module', autouse=True)
def _require_native(tdgp):
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    tdgp._lib.load()        # raises if libtdgp_hip.so is missing: GPU tests never run on a fallback


# ------------------------------------------------------------------------------------------------ bias_act

ACTS = ['linear', 'relu', 'lrelu', 'tanh', 'sigmoid', 'elu', 'selu', 'softplus', 'swish']


@pytest.mark.parametrize('act', ACTS)
def test_bias_act_golden(tdgp, act):
    g = load_golden('bias_act')
    y = tdgp.ops.bias_act.bias_act(T(g['x']), T(g['b']), act=act)
    assert_close(N(y), g[f'y_{act}'], 2e-6, act)


def test_bias_act_variants(tdgp, oracle):
    g = load_golden('bias_act')
    ba = tdgp.ops.bias_act.bias_act
    assert_close(N(ba(T(g['x']), T(g['b']), act='lrelu', gain=1.0, clamp=0.5)), g['y_lrelu_gain1_clamp'], 1e-6)
    assert_close(N(ba(T(g['x']), None, act='linear', gain=0.5)), g['y_linear_nobias'], 1e-7)
    assert_close(N(ba(T(g['x']), T(g['b_dim3']), dim=3, act='relu')), g['y_relu_dim3'], 1e-6)
    assert_close(N(ba(T(g['x2']), T(g['b']), act='lrelu')), g['y2_lrelu'], 1e-6)
    xcl = T(g['x']).contiguous(memory_format=torch.channels_last)
    y = ba(xcl, T(g['b']), act='swish', clamp=2.0)
    assert y.is_contiguous(memory_format=torch.channels_last)
    assert_close(N(y.contiguous()), g['y_swish_channels_last'], 2e-6)
    # larger, odd-sized (vector body + scalar tail), every activation, vs the oracle
    rs = np.random.RandomState(0)
    x = (rs.randn(3, 7, 33, 31) * 2).astype(np.float32)
    b = rs.randn(7).astype(np.float32)
    for act in ACTS:
        assert_close(N(ba(T(x), T(b), act=act)), oracle.bias_act(x, b, act=act), 2e-6, act)
    # the MLP shape of the renderer: [P, 64] with bias over the last dim
    x = rs.randn(4099, 64).astype(np.float32)
    b = rs.randn(64).astype(np.float32)
    assert_close(N(ba(T(x), T(b), act='lrelu')), oracle.bias_act(x, b, act='lrelu'), 1e-6)
    # bf16 / fp16 I/O (fp32 math inside)
    for dt, tol in ((torch.bfloat16, 1e-2), (torch.float16, 2e-3)):
        y = ba(T(x, dt), T(b, dt), act='lrelu')
        assert y.dtype == dt
        ref = oracle.bias_act(N(T(x, dt)), N(T(b, dt)), act='lrelu')
        assert_close(N(y), ref, tol, str(dt))
    # error contract (bias_act.cpp:35-51)
    with pytest.raises(RuntimeError):
        ba(T(x), T(rs.randn(3).astype(np.float32)), act='lrelu')
    assert ba(T(np.zeros((0, 4), np.float32)), None, act='relu').shape == (0, 4)


def test_plugin_ops_fp64(tdgp, oracle):
    """VERDICT r03 missing #4: the reference dispatches its two plugins over AT_DISPATCH_FLOATING_TYPES_AND_HALF (bias_act.cpp:77,
    upfirdn2d.cpp:63), i.e. double tensors are legal and computed in double (bias_act.cu:17-19, upfirdn2d.cu:21-27).  TDGP_F64: every
    activation forward / first / second derivative against the reference's own `_bias_act_ref` formulas evaluated by torch in float64 on the
    CPU (the op's impl='ref' path), to double precision; and against the fp32 goldens to fp32 precision."""
    ba = tdgp.ops.bias_act.bias_act
    g = load_golden('bias_act')
    x64, b64 = g['x'].astype(np.float64), g['b'].astype(np.float64)
    specs = tdgp.ops.bias_act.activation_funcs
    for act in ACTS:
        # alpha / gain / clamp cross the plugin boundary as C floats (bias_act.cpp:32 `float alpha, float gain, float clamp`; the CUDA kernel
        # widens them to scalar_t): the double computation uses the fp32-rounded scalars, so the comparison passes those to both sides
        f32 = lambda v: float(np.float32(v))             # noqa: E731
        base = dict(alpha=f32(specs[act].def_alpha), gain=f32(specs[act].def_gain))
        for kw in (base, dict(base, gain=0.75, clamp=0.5)):
            xd = torch.tensor(x64, device=DEV, requires_grad=True)
            bd = torch.tensor(b64, device=DEV, requires_grad=True)
            y = ba(xd, bd, act=act, **kw)
            assert y.dtype == torch.float64
            xc, bc = torch.tensor(x64, requires_grad=True), torch.tensor(b64, requires_grad=True)
            yc = ba(xc, bc, act=act, impl='ref', **kw)
            assert float((y.detach().cpu() - yc.detach()).abs().max()) <= 1e-13 * max(1.0, float(yc.abs().max())), act
            if kw is base:
                assert_close(N(y.detach()).astype(np.float32), g[f'y_{act}'], 2e-6, act + ' f64 vs the fp32 golden')
            if act == 'linear' and 'clamp' in kw:
                continue      # bias_act.py:28 `ref=''` for linear: the plugin's backward has no saved output to mask the clamp with (the reference's CUDA path alike)
            w = torch.tensor(np.random.RandomState(3).randn(*x64.shape))
            (gx, gb) = torch.autograd.grad((y * w.to(DEV)).sum(), [xd, bd], create_graph=True)
            (cx, cb) = torch.autograd.grad((yc * w).sum(), [xc, bc], create_graph=True)
            assert float((gx.detach().cpu() - cx.detach()).abs().max()) <= 1e-12 * max(1.0, float(cx.abs().max())), act
            assert float((gb.detach().cpu() - cb.detach()).abs().max()) <= 1e-11 * max(1.0, float(cb.abs().max())), act
            if gx.requires_grad:                                   # second order (R1 path): d/dx of <gx, w2>
                w2 = torch.tensor(np.random.RandomState(4).randn(*x64.shape))
                hx, = torch.autograd.grad((gx * w2.to(DEV)).sum(), xd, allow_unused=True)
                hc, = torch.autograd.grad((cx * w2).sum(), xc, allow_unused=True)
                hx = torch.zeros_like(xc) if hx is None else hx.cpu()
                hc = torch.zeros_like(xc) if hc is None else hc
                assert float((hx - hc).abs().max()) <= 1e-11 * max(1.0, float(hc.abs().max())), act
    # upfirdn2d: double in, double accumulation and gain, double out; the filter stays float32 (upfirdn2d.cpp:23)
    u = tdgp.ops.upfirdn2d
    rs = np.random.RandomState(5)
    f = oracle.setup_filter([1, 3, 3, 1])
    for shape, kw in [((2, 3, 17, 23), dict(padding=[1, 1, 1, 1], gain=4)), ((1, 2, 16, 64), dict(up=2, padding=[2, 1, 2, 1], gain=4)),
                      ((1, 2, 19, 11), dict(up=[3, 2], down=[2, 1], padding=[2, 1, 0, 3], gain=1.5, flip_filter=True))]:
        x = rs.randn(*shape)
        y = u.upfirdn2d(torch.tensor(x, device=DEV), T(f), **kw)
        assert y.dtype == torch.float64
        ref = u.upfirdn2d(torch.tensor(x), torch.tensor(f), impl='ref', **kw)
        assert y.shape == ref.shape and float((y.cpu() - ref).abs().max()) <= 1e-13 * float(ref.abs().max())
        okw = dict(kw)
        assert_close(N(y).astype(np.float32), oracle.upfirdn2d(x.astype(np.float32), f, **okw), 2e-6, 'upfirdn2d f64 vs the fp32 oracle', 1.0)


# ------------------------------------------------------------------------------------------------ upfirdn2d

def test_upfirdn2d_golden(tdgp):
    g = load_golden('upfirdn2d')
    u = tdgp.ops.upfirdn2d
    f = T(g['f1331'])
    np.testing.assert_array_equal(N(u.setup_filter([1, 3, 3, 1])), g['f1331'])
    tol = 2e-6
    assert_close(N(u.upfirdn2d(T(g['x_f1']), f, padding=[1, 1, 1, 1], gain=4)), g['y_f1'], tol, 'F1', 1.0)
    assert_close(N(u.upsample2d(T(g['x_f2']), f)), g['y_f2'], tol, 'F2', 1.0)
    assert_close(N(u.filter2d(T(g['x_odd']), f)), g['y_filter2d'], tol, 'filter2d', 1.0)
    assert_close(N(u.downsample2d(T(g['x_odd']), f)), g['y_downsample2d'], tol, 'downsample2d', 1.0)
    assert_close(N(u.upfirdn2d(T(g['x_odd']), f, padding=[-1, 2, 3, -1])), g['y_negpad'], tol, 'negpad', 1.0)
    assert_close(N(u.upfirdn2d(T(g['x_odd']), T(g['f_asym']), padding=2)), g['y_asym_noflip'], tol, 'asym', 1.0)
    assert_close(N(u.upfirdn2d(T(g['x_odd']), T(g['f_asym']), padding=2, flip_filter=True)), g['y_asym_flip'], tol, 'asym flip', 1.0)
    assert_close(N(u.upfirdn2d(T(g['x_odd']), T(g['f_rect']), up=[3, 2], down=[2, 1], padding=[2, 1, 0, 3], gain=1.5)),
                 g['y_rect_up3_down2'], tol, 'rect', 1.0)
    assert_close(N(u.upfirdn2d(T(g['x_odd']), None)), g['y_identity'], 0, 'identity')
    assert_close(N(u.upfirdn2d(T(g['x_f1_33']), f, padding=[1, 1, 1, 1], gain=4)), g['y_f1_33'], tol, 'F1 33', 1.0)


def test_upfirdn2d_oracle(tdgp, oracle):
    u = tdgp.ops.upfirdn2d
    rs = np.random.RandomState(1)
    f = oracle.setup_filter([1, 3, 3, 1])
    for shape in [(2, 5, 65, 65), (1, 3, 17, 40)]:
        x = rs.randn(*shape).astype(np.float32)
        assert_close(N(u.upfirdn2d(T(x), T(f), padding=[1, 1, 1, 1], gain=4)), oracle.upfirdn2d(x, f, padding=[1, 1, 1, 1], gain=4), 2e-6, 'F1', 1.0)
        assert_close(N(u.upsample2d(T(x), T(f))), oracle.upsample2d(x, f), 2e-6, 'F2', 1.0)
    # separable 8-tap filter goes out as two launches (upfirdn2d.py:241-245)
    f8 = u.setup_filter([1, 2, 3, 4, 4, 3, 2, 1])
    assert f8.ndim == 1
    x = rs.randn(1, 2, 19, 23).astype(np.float32)
    ref = oracle.upfirdn2d(x, np.outer(N(f8), N(f8)).astype(np.float32), padding=[4, 3, 4, 3])
    assert_close(N(u.upfirdn2d(T(x), f8.to(DEV), padding=[4, 3, 4, 3])), ref, 2e-6, 'separable', 1.0)
    # channels_last input keeps its layout (upfirdn2d.cpp:38)
    x = rs.randn(2, 8, 12, 12).astype(np.float32)
    y = u.upfirdn2d(T(x).contiguous(memory_format=torch.channels_last), T(f), up=2, padding=[2, 1, 2, 1], gain=4)
    assert_close(N(y.contiguous()), oracle.upsample2d(x, f), 2e-6, 'channels_last', 1.0)
    with pytest.raises(RuntimeError):
        u.upfirdn2d(T(x), T(f), padding=-8)          # output must be at least 1x1


@pytest.mark.parametrize('shape,kw', [((2, 3, 129, 129), dict(padding=[1, 1, 1, 1], gain=4)),                 # F1 at a tileable size: 128^2 out, 16-B stores
                                      ((1, 2, 70, 201), dict(padding=[1, 1, 1, 1], gain=4)),                   # ragged tiles, scalar loads / stores (odd pitch)
                                      ((1, 2, 40, 160), dict(padding=[2, 1, 0, 3], gain=1.5, flip_filter=True)),  # asymmetric padding, flipped filter
                                      ((2, 3, 64, 64), dict(up=2, padding=[2, 1, 2, 1], gain=4)),              # F2: 128^2 out
                                      ((1, 2, 37, 53), dict(up=2, padding=[2, 1, 2, 1], gain=4)),              # F2, odd sizes
                                      ((1, 1, 33, 100), dict(up=2, padding=[1, 2, 3, 0], gain=2.0, flip_filter=True))])
def test_upfirdn2d_lds_tiles(tdgp, oracle, shape, kw):
    """The LDS-staged 4x4 kernel (planes of >= 16 x 64 outputs: the generator's F1 / F2 forms at their hot sizes) against the oracle:
    window geometry for both up factors, every padding, ragged edge tiles, vector and scalar load / store paths."""
    u = tdgp.ops.upfirdn2d
    rs = np.random.RandomState(shape[2] + shape[3])
    f = oracle.setup_filter([1, 2, 3, 1]) if kw.get('flip_filter') else oracle.setup_filter([1, 3, 3, 1])       # an asymmetric filter where flipping matters
    x = rs.randn(*shape).astype(np.float32)
    ref = oracle.upfirdn2d(x, f, **kw)
    assert ref.shape[2] >= 16 and ref.shape[3] >= 64
    assert_close(N(u.upfirdn2d(T(x), T(f), **kw)), ref, 2e-6, f'upfirdn2d {shape} {kw}', 1.0)
    xs = T(np.concatenate([x, x], axis=3))[:, :, :, 1:shape[3] + 1]                       # same values, a view with an unaligned base and pitch
    xs_ref = oracle.upfirdn2d(np.ascontiguousarray(np.concatenate([x, x], axis=3)[:, :, :, 1:shape[3] + 1]), f, **kw)
    assert_close(N(u.upfirdn2d(xs, T(f), **kw)), xs_ref, 2e-6, 'unaligned view', 1.0)


# ------------------------------------------------------------------------------------------------ modulated conv

@pytest.mark.parametrize('tag', ['c3_up1', 'c3_up2', 'c3_up2_b1', 'c1_rgb', 'c3_up1_b1_nonoise'])
def test_modconv_golden(tdgp, tag):
    g = load_golden('modconv')
    k, up, demod = (int(v) for v in g[f'{tag}_meta'])
    noise = g.get(f'{tag}_noise')
    y = tdgp.ops.modconv.modulated_conv2d(T(g[f'{tag}_x']), T(g[f'{tag}_w']), T(g[f'{tag}_s']), noise=None if noise is None else T(noise), up=up,
                                          padding=k // 2, resample_filter=T(g['f']), demodulate=bool(demod), flip_weight=(up == 1))
    assert_close(N(y), g[f'{tag}_y'], 5e-6, tag, 1.0)


@pytest.mark.parametrize('B,cin,cout,H,k,up', [
    (2, 64, 128, 32, 3, 1),      # 128x128 tile config
    (2, 40, 48, 64, 3, 1),       # 64x256 tile config, channel tails
    (5, 32, 160, 4, 3, 1),       # 4x4 maps: several samples per pixel tile
    (3, 24, 136, 8, 3, 2),       # up-layer, straddling tiles, odd phase grids 9x9
    (1, 72, 64, 33, 3, 2),       # odd input size, phase grid 34 > one 32-column tile
    (2, 64, 96, 32, 1, 1),       # ToRGB config (96 = 3 x 32 rows)
    (2, 48, 20, 16, 1, 1),
    (3, 6, 10, 32, 3, 1),        # channel count not a multiple of 4 on the mask-free 3x3 path (scalar style loads, clamped channels)
    (2, 7, 12, 16, 3, 2),        # ... and on the x2 path; 3 K iterations of a 2-deep pipeline
    (1, 130, 70, 64, 3, 1),      # Cin % 4 == 2, odd iteration count (33), Cout tail inside a 128-row tile
    (2, 5, 9, 32, 5, 1),         # 5x5 fast path, ragged channels
])
def test_modconv_oracle(tdgp, oracle, B, cin, cout, H, k, up):
    rs = np.random.RandomState(cin + cout)
    x = rs.randn(B, cin, H, H).astype(np.float32)
    w = rs.randn(cout, cin, k, k).astype(np.float32)
    s = (1 + 0.5 * rs.randn(B, cin)).astype(np.float32)
    noise = (0.3 * rs.randn(B, 1, H * up, H * up)).astype(np.float32) if k >= 3 else None
    f = oracle.setup_filter([1, 3, 3, 1])
    ref = oracle.modulated_conv2d(x, w, s, noise=noise, up=up, demodulate=(k >= 3), resample_filter=f)
    y = tdgp.ops.modconv.modulated_conv2d(T(x), T(w), T(s), noise=None if noise is None else T(noise), up=up, padding=k // 2,
                                          resample_filter=T(f), demodulate=(k >= 3), flip_weight=(up == 1))
    assert_close(N(y), ref, 1e-5, 'modconv', 1.0)


@pytest.mark.parametrize('B,cin,cout,H,W,kw', [
    (16, 64, 64, 64, 64, {}),                              # the smallest channel count that takes the Winograd kernel; exactly 256 blocks
    (8, 136, 70, 64, 64, dict(clamp=0.7)),                 # 17 chunks (odd), Cout tail inside a 64-channel block, clamp
    (3, 72, 130, 64, 128, dict(noise=False)),              # H != W, three output-channel blocks with a tail, no noise
    (16, 64, 32, 64, 64, dict(styles=False)),              # unmodulated (Conv2dLayer form), Cout < 64
    (16, 64, 64, 64, 64, dict(noise='per_sample')),        # noise_mode='random': one [1,H,W] map per sample (noise_bstride = H*W), metric_utils.py:310
    (16, 128, 128, 32, 64, dict(noise='per_sample', clamp=0.9)),
])
def test_modconv_winograd_vs_oracle(tdgp, oracle, B, cin, cout, H, W, kw):
    """The Winograd F(2x2,3x3) kernel (default arithmetic for the large stride-1 3x3 layers, modconv_wino.inc) against the double-
    accumulating oracle, and against the direct-sum kernel (`set_conv_arith(2)`) on the same call; the profiler names which ran."""
    rs = np.random.RandomState(cin * 7 + cout)
    x = rs.randn(B, cin, H, W).astype(np.float32)
    w = rs.randn(cout, cin, 3, 3).astype(np.float32)
    styles = kw.get('styles', True)
    s = (1 + 0.5 * rs.randn(B, cin)).astype(np.float32) if styles else None
    noise = (0.3 * rs.randn(H, W)).astype(np.float32) if kw.get('noise', True) else None
    if kw.get('noise') == 'per_sample':
        noise = (0.3 * rs.randn(B, 1, H, W)).astype(np.float32)        # networks_stylegan2.py:133-134: randn([B,1,res,res]) * noise_strength
    bias = (0.2 * rs.randn(cout)).astype(np.float32)
    clamp = kw.get('clamp')
    oracle.set_threads(min(64, __import__('os').cpu_count() or 1))
    ref = oracle.modulated_conv2d(x, w, s if styles else np.ones([B, cin], np.float32), noise=noise,            # 2-D: one map for the batch
                                  up=1, demodulate=styles, resample_filter=oracle.setup_filter([1, 3, 3, 1]))
    scale = np.abs(oracle.bias_act(ref, bias, act='lrelu')).max()          # errors are measured against the un-clamped output range
    ref = oracle.bias_act(ref, bias, act='lrelu', clamp=clamp)
    M = tdgp.ops.modconv
    pk = M._packed(T(w))
    out = {}
    for mode in (3, 2):                 # 3: the default arithmetic without F(4x4) (which takes the larger layers since round 4)
        prev = tdgp._lib.set_conv_arith(mode)
        tdgp._lib.profile_enable(True)
        try:
            y = M.modconv_forward(T(x), pk, None if s is None else T(s), noise=None if noise is None else T(noise), bias=T(bias), demodulate=styles,
                                  act='lrelu', clamp=clamp)
            torch.cuda.synchronize()
            names = set(tdgp._lib.profile_report())
        finally:
            tdgp._lib.profile_enable(False)
            tdgp._lib.set_conv_arith(prev)
        out[mode] = (N(y), names)
    assert 'conv_wino_kernel' in out[3][1] and 'conv_wino_kernel' not in out[2][1], (out[3][1], out[2][1])
    e_w, e_d = np.abs(out[3][0] - ref).max() / scale, np.abs(out[2][0] - ref).max() / scale
    report_parity(f'winograd 3x3 {cin}->{cout} @{H}x{W}', winograd_vs_oracle=e_w, direct_vs_oracle=e_d,
                  winograd_vs_direct=np.abs(out[3][0] - out[2][0]).max() / scale)
    assert e_d <= 5e-6, e_d
    assert e_w <= 1e-5, e_w            # F(2x2,3x3) in fp32: a few ulp more than the direct sum (transforms add roundings), same order


@pytest.mark.parametrize('B,cin,cout,H,W,kw', [
    (16, 128, 128, 64, 64, {}),                                          # the smallest shape that takes the F(4x4) kernels: exactly 256 items, 2 slices
    (16, 132, 200, 64, 64, dict(clamp=0.7, noise='per_sample')),         # 33 chunks (odd), Cout tail inside a 64-channel slice, clamp, per-sample noise
    (8, 256, 128, 64, 128, dict(noise=False)),                           # two tile groups per row, H != W, no noise
    (16, 128, 128, 64, 64, dict(styles=False)),                          # unmodulated (Conv2dLayer form)
    (4, 512, 512, 64, 64, {}),                                           # the 64^2 x 512 layer of C3: the longest reduction the default run sums in F(4x4)
    (4, 64, 64, 256, 256, dict(clamp=0.8)),                              # the fewest channels F(4x4) takes (C3's 512^2 x 64 layer shape at a quarter of the area)
    (4, 128, 640, 64, 64, dict(noise=False)),                            # 20 slices of 32 channels = 3 rectangles of 8 per XCD pass, the last one ragged: empty slots are skipped, not the end
    (16, 256, 512, 32, 32, dict(noise='per_sample')),                    # 32-pixel-wide layers: tile groups of 8 x 4 tiles (32 x 16 pixels); 256 items
    (32, 128, 192, 48, 32, {}),                                          # ... with H a multiple of 16 only, three 64-channel slices
    (43, 64, 128, 24, 64, dict(noise='per_sample')),                     # 129 tile groups: the last PAIR of the 8-wave form has one group only (its second half multiplies the first group's data and stores nothing)
    (4, 512, 512, 32, 32, dict(splitk=True)),                            # K split 4 ways: C3's 32^2 layer at batch 4 (64 items unsplit) -- raw sums through the split-K buffer
    (8, 512, 512, 32, 32, dict(splitk=True, noise='per_sample', clamp=0.9)),   # K split 2 ways (batch 8), per-sample noise and clamp applied by the reduction pass
    (8, 256, 200, 32, 32, dict(splitk=True, noise=False)),               # 4 splits of exactly 16 chunks, a ragged last slice
    (8, 80, 64, 128, 128, {}),                                           # fused input transform (round 6): 20 chunks = five rounds of the unrolled K loop, one 64-channel slice, 256 items
    (2, 64, 192, 256, 192, dict(noise='per_sample', clamp=0.6)),         # ... three slices (every window transformed three times), H != W, three tile groups per row, clamp
    (16, 128, 64, 128, 64, dict(styles=False)),                          # ... ONE tile group per row (both halo columns outside the image), unmodulated, 32 chunks
])
def test_modconv_winograd4_vs_oracle(tdgp, oracle, B, cin, cout, H, W, kw):
    """Winograd F(4x4,3x3) (modconv_wino4.inc: input-transform pass + 36-GEMM kernel, points 0, +-1, 1/2, -2, inf; modconv_wino4f.inc: the layers with
    Cin <= 128, Cin % 16 == 0, Cout % 64 == 0 on 64-pixel-wide tile groups transform their input INSIDE the GEMM kernel) against the double-accumulating
    oracle, next to F(2x2) (`set_conv_arith(3)`) and the direct sum (`set_conv_arith(2)`) on the same call; the profiler names which ran.  The fused
    shapes also run with the separate transform pass (`set_conv_arith(4)`): the two forms must agree bit for bit (same V, same summation order).
    Bound: 1e-5 of the un-clamped output range, the bound of every reduction row (measured per layer in the parity report)."""
    fused = cin <= 128 and cin % 16 == 0 and cout % 64 == 0 and W % 64 == 0 and H % 8 == 0 and not kw.get('splitk') and B * ((H * W) >> 9) * (cout // 64) >= 256
    rs = np.random.RandomState(cin * 7 + cout + 1)
    x = rs.randn(B, cin, H, W).astype(np.float32)
    x = np.where(x > 0, x, 0.2 * x).astype(np.float32) * np.float32(np.sqrt(2))          # the layers' real input: a leaky-ReLU output (non-zero mean)
    w = rs.randn(cout, cin, 3, 3).astype(np.float32)
    styles = kw.get('styles', True)
    s = (1 + 0.5 * rs.randn(B, cin)).astype(np.float32) if styles else None
    noise = (0.3 * rs.randn(H, W)).astype(np.float32) if kw.get('noise', True) else None
    if kw.get('noise') == 'per_sample':
        noise = (0.3 * rs.randn(B, 1, H, W)).astype(np.float32)
    bias = (0.2 * rs.randn(cout)).astype(np.float32)
    clamp = kw.get('clamp')
    oracle.set_threads(min(64, os.cpu_count() or 1))
    ref = oracle.modulated_conv2d(x, w, s if styles else np.ones([B, cin], np.float32), noise=noise, up=1, demodulate=styles,
                                  resample_filter=oracle.setup_filter([1, 3, 3, 1]))
    scale = np.abs(oracle.bias_act(ref, bias, act='lrelu')).max()
    ref = oracle.bias_act(ref, bias, act='lrelu', clamp=clamp)
    M = tdgp.ops.modconv
    pk = M._packed(T(w))
    out = {}
    for mode in (0, 3, 2) + ((4,) if fused else ()):
        prev = tdgp._lib.set_conv_arith(mode)
        tdgp._lib.profile_enable(True)
        try:
            y = M.modconv_forward(T(x), pk, None if s is None else T(s), noise=None if noise is None else T(noise), bias=T(bias), demodulate=styles,
                                  act='lrelu', clamp=clamp)
            torch.cuda.synchronize()
            names = set(tdgp._lib.profile_report())
            y2 = M.modconv_forward(T(x), pk, None if s is None else T(s), noise=None if noise is None else T(noise), bias=T(bias), demodulate=styles,
                                   act='lrelu', clamp=clamp)
            assert torch.equal(y, y2)                                  # deterministic
        finally:
            tdgp._lib.profile_enable(False)
            tdgp._lib.set_conv_arith(prev)
        out[mode] = (N(y), names)
    if fused:
        assert 'conv_wino4f_kernel' in out[0][1] and not ({'conv_wino4_kernel', 'wino4_input_kernel', 'conv_wino_kernel'} & out[0][1]), out[0][1]
        assert {'conv_wino4_kernel', 'wino4_input_kernel'} <= out[4][1] and 'conv_wino4f_kernel' not in out[4][1], out[4][1]
        np.testing.assert_array_equal(out[0][0], out[4][0])              # the transform inside the GEMM kernel == the transform pass + GEMM: every bit
    else:
        assert {'conv_wino4_kernel', 'wino4_input_kernel'} <= out[0][1] and not ({'conv_wino_kernel', 'conv_wino4f_kernel'} & out[0][1]), out[0][1]
    assert ('splitk_reduce_kernel' in out[0][1]) == bool(kw.get('splitk')), out[0][1]
    assert 'conv_wino4_kernel' not in out[3][1] and ('conv_wino_kernel' in out[3][1] or cin % 8 != 0 or W < 64), out[3][1]   # (F(2x2): Cin % 8 == 0, >= 256 of its own blocks)
    assert not ({'conv_wino_kernel', 'conv_wino4_kernel'} & out[2][1]), out[2][1]
    e4, e2, ed = (float(np.abs(out[m][0] - ref).max() / scale) for m in (0, 3, 2))
    rms4 = float(np.sqrt(np.mean((out[0][0] - ref).astype(np.float64) ** 2)) / scale)
    report_parity(f'winograd F(4x4) 3x3 {cin}->{cout} @{H}x{W}', f4x4_vs_oracle=e4, f4x4_rms=rms4, f2x2_vs_oracle=e2, direct_vs_oracle=ed)
    assert ed <= 5e-6 and e2 <= 1e-5, (ed, e2)
    assert e4 <= 1e-5, e4


def test_modconv_winograd4_repeats_are_bit_identical(tdgp):
    """The 8-wave form of the F(4x4) kernel lets the V pieces of chunk c + 2 fly past the wait that ends chunk c (`vmcnt(4)`): that counts on
    LDS-direct loads completing in issue order.  A violation would be a race, i.e. run-to-run differences: 80 repeats of a layer whose V comes
    from beyond the L2 (and of a folded x2 layer), with other traffic in between, every output compared with the first bit for bit
    (tools/dev/stress_w4.py: 2100 repeats over seven shapes, none differing)."""
    M, U = tdgp.ops.modconv, tdgp.ops.upfirdn2d
    torch.manual_seed(5)
    fir = M.fir_host_array(U.setup_filter([1, 3, 3, 1]))
    churn = torch.randn(32 << 20, device=DEV)
    for (B, ci, co, R, up) in ((16, 128, 128, 256, 1), (16, 256, 128, 128, 2)):
        x = torch.randn(B, ci, R, R, device=DEV)
        pk = M._packed(torch.randn(co, ci, 3, 3, device=DEV))
        s = torch.randn(B, ci, device=DEV) * 0.5 + 1.0
        bias = torch.randn(co, device=DEV) * 0.1
        f = (lambda: M.modconv_forward(x, pk, s, bias=bias, act='lrelu')) if up == 1 else (lambda: M.modconv_forward(x, pk, s, bias=bias, up=2, fir=fir, act='lrelu'))
        ref = f().clone()
        for i in range(80):
            if i % 5 == 0:
                churn.mul_(1.0001)
            assert torch.equal(f(), ref), (B, ci, co, R, up, i)
        del x, pk, ref


def test_modconv_winograd4_sub_batches(tdgp, oracle):
    """A layer whose Winograd-domain input exceeds one 4 GiB buffer descriptor goes through the F(4x4) kernels in sub-batches sharing one V
    buffer -- exactly C4's 512^2 x 128 layer at B = 16: 302 MB of V per sample, 4.8 GB in all -> 13 + 3 samples.  Every sample equals the same
    sample convolved alone, and sample 0's first 16 output rows match the oracle."""
    rs = np.random.RandomState(77)
    B, cin, cout, H = 16, 128, 128, 512
    x = torch.randn(B, cin, H, H, device=DEV)
    w = rs.randn(cout, cin, 3, 3).astype(np.float32)
    s = (1 + 0.5 * rs.randn(B, cin)).astype(np.float32)
    noise = torch.randn(B, 1, H, H, device=DEV) * 0.3
    bias = (0.2 * rs.randn(cout)).astype(np.float32)
    M = tdgp.ops.modconv
    pk = M._packed(T(w))
    oracle.set_threads(min(64, os.cpu_count() or 1))
    xs = N(x[:1, :, :24])
    ref = oracle.modulated_conv2d(xs, w, s[:1], noise=N(noise[:1, :, :24]), up=1, demodulate=True, resample_filter=oracle.setup_filter([1, 3, 3, 1]))
    ref = oracle.bias_act(ref, bias, act='lrelu')[:, :, :16]
    ys = {}
    # mode 4: the separate transform pass (the sub-batched form this test was written for); mode 0 (round 6): this shape now transforms its input inside
    # the GEMM kernel -- no V buffer, one launch for the whole batch (2.1 GB of activations through one buffer descriptor)
    for mode in (4, 0):
        prev = tdgp._lib.set_conv_arith(mode)
        tdgp._lib.profile_enable(True)
        try:
            y = M.modconv_forward(x, pk, T(s), noise=noise, bias=T(bias), act='lrelu')
            torch.cuda.synchronize()
            rep = tdgp._lib.profile_report()
        finally:
            tdgp._lib.profile_enable(False)
        try:
            if mode == 4:
                assert rep['conv_wino4_kernel']['launches'] == 2 and rep['wino4_input_kernel']['launches'] == 2 and 'conv_wino4f_kernel' not in rep, rep
            else:
                assert rep['conv_wino4f_kernel']['launches'] == 1 and 'wino4_input_kernel' not in rep, rep
            for b in (0, 5, 12, 13, 15):
                one = M.modconv_forward(x[b:b + 1].contiguous(), pk, T(s[b:b + 1]), noise=noise[b:b + 1].contiguous(), bias=T(bias), act='lrelu')
                assert torch.equal(one[0], y[b]), (mode, b)
        finally:
            tdgp._lib.set_conv_arith(prev)
        assert_close(N(y[:1, :, :16]), ref, 1e-5, f'F(4x4) layer 512^2 x 128, conv arith {mode}, vs the oracle (rows 0-15 of sample 0)', 1.0)
        ys[mode] = y
    assert torch.equal(ys[0], ys[4])


@pytest.mark.parametrize('B,cin,cout,H,kw', [
    (16, 128, 64, 64, {}),                                   # C3's 128^2 -> 256^2 shape at a quarter of the area: 512 items
    (16, 64, 64, 64, dict(clamp=0.9, noise='per_sample')),   # the fewest channels; clamp; one noise map per sample
    (4, 512, 256, 64, {}),                                   # the longest reduction (C3's 64^2 -> 128^2 layer, C4's at half the channels)
    (32, 132, 70, 32, dict(noise=False)),                    # 33 chunks, Cout' = 280 (a ragged last slice), 32-pixel-wide tile groups, no noise; 320 items
    (43, 64, 64, 24, dict(W=64, noise='per_sample')),        # 129 tile groups: the last PAIR has one group only -- its second half parks (round 6: noise rows in the free V stage) and stores nothing; H != W
])
def test_modconv_up2_folded_vs_oracle(tdgp, oracle, B, cin, cout, H, kw):
    """The x2 layers with the FIR folded into four 3x3 parity kernels on the Winograd F(4x4) path (ops/modconv.py: fold_up2_table,
    csrc/modconv_wino4.inc out_layout 2) against the double-accumulating oracle of conv2d_resample.py:108-125, next to the transposed-
    convolution + FIR kernels on the same call.  Bound: 1e-5 of the un-clamped output range, as for every reduction row."""
    rs = np.random.RandomState(cin * 5 + cout)
    mc = tdgp.ops.modconv
    W = kw.get('W', H)
    x = rs.randn(B, cin, H, W).astype(np.float32)
    x = np.where(x > 0, x, 0.2 * x).astype(np.float32) * np.float32(np.sqrt(2))
    w = rs.randn(cout, cin, 3, 3).astype(np.float32)
    s = (1 + 0.5 * rs.randn(B, cin)).astype(np.float32)
    bias = (0.2 * rs.randn(cout)).astype(np.float32)
    noise = (0.3 * rs.randn(2 * H, 2 * W)).astype(np.float32) if kw.get('noise', True) else None
    if kw.get('noise') == 'per_sample':
        noise = (0.3 * rs.randn(B, 1, 2 * H, 2 * W)).astype(np.float32)
    clamp = kw.get('clamp')
    f = oracle.setup_filter([1, 3, 3, 1])
    oracle.set_threads(min(64, os.cpu_count() or 1))
    ref = oracle.modulated_conv2d(x, w, s, noise=noise, up=2, resample_filter=f)
    scale = np.abs(oracle.bias_act(ref, bias, act='lrelu')).max()
    ref = oracle.bias_act(ref, bias, act='lrelu', clamp=clamp)
    pk = mc.PackedConv(T(w))
    call = lambda: mc.modconv_forward(T(x), pk, T(s), noise=None if noise is None else T(noise), bias=T(bias), demodulate=True, act='lrelu', clamp=clamp,   # noqa: E731
                                      up=2, fir=mc.fir_host_array(f))
    out = {}
    for fold in (True, False):
        was, mc.FOLD_UP2 = mc.FOLD_UP2, fold
        tdgp._lib.profile_enable(True)
        try:
            y = call()
            torch.cuda.synchronize()
            names = set(tdgp._lib.profile_report())
            assert torch.equal(call(), y)
        finally:
            tdgp._lib.profile_enable(False)
            mc.FOLD_UP2 = was
        out[fold] = (N(y), names)
    assert {'upconv_wino4_kernel', 'wino4_input_kernel'} <= out[True][1] and not ({'upconv_mfma_kernel', 'fir_act_kernel'} & out[True][1]), out[True][1]
    assert {'upconv_mfma_kernel', 'fir_act_kernel'} <= out[False][1] and 'upconv_wino4_kernel' not in out[False][1], out[False][1]
    ef, ed = (float(np.abs(out[k_][0] - ref).max() / scale) for k_ in (True, False))
    report_parity(f'x2 layer {cin}->{cout} @{H}^2 -> {2 * H}^2', folded_f4x4_vs_oracle=ef, transposed_conv_fir_vs_oracle=ed)
    assert ed <= 5e-6 and ef <= 1e-5, (ed, ef)
    # with F(4x4) switched off the library refuses the folded form before launching anything and the op falls back
    prev = tdgp._lib.set_conv_arith(3)
    try:
        assert torch.equal(call(), torch.as_tensor(out[False][0]).to(DEV))
    finally:
        tdgp._lib.set_conv_arith(prev)


def test_fused_layers_oracle(tdgp, oracle):
    """SynthesisLayer / ToRGB+skip as single fused calls (bias, lrelu*sqrt2, x2 FIR skip, channel-last output)."""
    rs = np.random.RandomState(5)
    mc = tdgp.ops.modconv
    B, cin, cout, H = 2, 32, 48, 16
    x = rs.randn(B, cin, H, H).astype(np.float32)
    w = rs.randn(cout, cin, 3, 3).astype(np.float32)
    s = (1 + 0.5 * rs.randn(B, cin)).astype(np.float32)
    bias = rs.randn(cout).astype(np.float32)
    f = oracle.setup_filter([1, 3, 3, 1])
    for up in (1, 2):
        noise = (0.3 * rs.randn(H * up, H * up)).astype(np.float32)
        ref = oracle.bias_act(oracle.modulated_conv2d(x, w, s, noise=noise, up=up, resample_filter=f), bias, act='lrelu')
        y = mc.modconv_forward(T(x), mc.PackedConv(T(w)), T(s), noise=T(noise), bias=T(bias), up=up, demodulate=True, act='lrelu',
                               fir=mc.fir_host_array(f))
        assert_close(N(y), ref, 1e-5, f'layer up{up}', 1.0)
    # ToRGB with skip: img = upsample2d(prev) + torgb(x), NCHW and channel-last plane layouts
    crgb = 24
    w1 = rs.randn(crgb, cin, 1, 1).astype(np.float32)
    prev = rs.randn(B, crgb, H // 2, H // 2).astype(np.float32)
    ref = oracle.upsample2d(prev, f) + oracle.bias_act(oracle.modulated_conv2d(x, w1, s, demodulate=False), bias[:crgb])
    pk = mc.PackedConv(T(w1))
    y0 = mc.modconv_forward(T(x), pk, T(s), bias=T(bias[:crgb]), demodulate=False, skip=T(prev), fir=mc.fir_host_array(f))
    assert_close(N(y0), ref, 1e-5, 'torgb nchw', 1.0)
    feat = 8
    prev_cl = T(prev).reshape(B, 3, feat, H // 2, H // 2).permute(0, 1, 3, 4, 2).contiguous()
    y1 = mc.modconv_forward(T(x), pk, T(s), bias=T(bias[:crgb]), demodulate=False, skip=prev_cl, fir=mc.fir_host_array(f), out_layout=1, out_feat=feat)
    assert y1.shape == (B, 3, H, H, feat)
    assert_close(N(y1.permute(0, 1, 4, 2, 3).reshape(B, crgb, H, H)), ref, 1e-5, 'torgb channel-last', 1.0)


@pytest.mark.parametrize('B,cin,crgb,H,W,clamp,gain,skip', [
    (2, 64, 96, 64, 64, None, 1.0, True),        # the generator's form: resident weights, several tiles per block, trimmed output stage
    (3, 64, 96, 32, 128, None, 1.0, True),       # ragged last tile of a block (3*32*128/128 = 96 tiles)
    (2, 48, 24, 16, 16, 0.7, 1.0, True),         # clamp -> general output stage, resident weights
    (2, 96, 36, 12, 20, None, 1.0, True),        # not a power of two -> general geometry (cross-lane reads), two K stages
    (1, 200, 60, 24, 8, 1.5, 0.5, False),        # Cin not a multiple of 64, gain != 1, no skip
    (2, 128, 96, 32, 32, None, 1.0, False),
    (1, 64, 96, 516, 1020, None, 1.0, True),     # 4112 tiles -> 2 tiles per block, ragged last tile, general geometry
    (1, 64, 96, 1024, 512, None, 1.0, True),     # 4096 tiles -> 2 tiles per block, trimmed output stage
])
def test_torgb_channel_last_variants(tdgp, oracle, B, cin, crgb, H, W, clamp, gain, skip):
    """Every specialisation of the ToRGB kernel (weights resident or staged, trimmed / general output stage, with / without the
    x2-upsampled skip) against the oracle's modulated 1x1 conv + bias_act (+ upsample2d of the previous image)."""
    rs = np.random.RandomState(B * 1000 + cin)
    mc = tdgp.ops.modconv
    feat = crgb // 3
    x = rs.randn(B, cin, H, W).astype(np.float32)
    w1 = rs.randn(crgb, cin, 1, 1).astype(np.float32)
    s = (1 + 0.5 * rs.randn(B, cin)).astype(np.float32)
    bias = rs.randn(crgb).astype(np.float32)
    f = oracle.setup_filter([1, 3, 3, 1])
    ref = oracle.bias_act(oracle.modulated_conv2d(x, w1, s, demodulate=False), bias, act='linear', gain=gain, clamp=clamp)
    prev_cl = None
    if skip:
        prev = rs.randn(B, crgb, H // 2, W // 2).astype(np.float32)
        # the reference's order (networks_stylegan2.py:170-171, 265-269): y = clamp(gain * (conv + bias)), THEN img = upsample2d(img) + y
        ref = ref + oracle.upsample2d(prev, f)
        prev_cl = T(prev).reshape(B, 3, feat, H // 2, W // 2).permute(0, 1, 3, 4, 2).contiguous()
    y = mc.modconv_forward(T(x), mc.PackedConv(T(w1)), T(s), bias=T(bias), demodulate=False, act='linear', gain=gain, clamp=clamp, skip=prev_cl,
                           fir=mc.fir_host_array(f) if skip else None, out_layout=1, out_feat=feat)
    assert y.shape == (B, 3, H, W, feat)
    assert_close(N(y.permute(0, 1, 4, 2, 3).reshape(B, crgb, H, W)), ref, 1e-5, 'torgb channel-last', 1.0)


# ------------------------------------------------------------------------------------------------ camera / rays

def test_camera_and_rays(tdgp, oracle):
    g = load_golden('camera')
    R = tdgp.renderer
    c2w = R.compute_cam2world_matrix(dict(angles=T(g['angles']), radius=T(g['radius']), look_at=T(g['look_at'])))
    assert_close(N(c2w), g['c2w'], 2e-6, 'c2w', 1.0)
    for hw in [(8, 8), (5, 7), (16, 16)]:
        o, d = R.sample_rays(T(g['c2w']), T(g['fov']), resolution=hw)
        assert_close(N(o), g['ray_o_%dx%d' % hw], 1e-7, 'ray_o')
        assert_close(N(d), g['ray_d_%dx%d' % hw], 2e-6, 'ray_d', 1.0)
    o, d = R.sample_rays(T(g['c2w']), T(g['fov']), resolution=(6, 6), patch_params=dict(scales=T(g['patch_scales']), offsets=T(g['patch_offsets'])))
    assert_close(N(d), g['ray_d_patch'], 2e-6, 'ray_d patch', 1.0)
    o, d = R.sample_rays(T(g['c2w']), 18.0, resolution=(4, 4))
    assert_close(N(d), g['ray_d_scalar_fov'], 2e-6, 'ray_d scalar fov', 1.0)
    # bit-exact vs the oracle (same fp32 chain, fp64 transcendentals on both sides)
    oo, od = oracle.sample_rays(g['c2w'], g['fov'], 16, 16)
    o, d = R.sample_rays(T(g['c2w']), T(g['fov']), resolution=(16, 16))
    np.testing.assert_array_equal(N(o), oo)
    np.testing.assert_array_equal(N(d), od)


# ------------------------------------------------------------------------------------------------ field

def _mlp(tdgp, w0, b0, w1, b1, marcher):
    m = tdgp.renderer.TriPlaneMLP(w0.shape[1], w0.shape[0], 3, marcher).to(DEV)
    with torch.no_grad():
        m.model[0].weight.copy_(T(w0)); m.model[0].bias.copy_(T(b0)); m.model[1].weight.copy_(T(w1)); m.model[1].bias.copy_(T(b1))
    return m


@pytest.mark.parametrize('marcher', ['classical', 'mip'])
def test_field_golden(tdgp, oracle, marcher):
    g = load_golden('field')
    ws = [g[f'{marcher}_{n}'] for n in ('w0', 'b0', 'w1', 'b1')]
    out = tdgp.renderer.simple_tri_plane_renderer(T(g['planes']), T(g['coords']), _mlp(tdgp, *ws, marcher), scale=0.5, return_taps=True)
    assert_close(N(out['rgb']), g[f'{marcher}_rgb'], 5e-6, 'rgb', 1.0)
    assert_close(N(out['sigma']), g[f'{marcher}_sigma'], 5e-6, 'sigma', 1.0)
    ref = oracle.triplane_field(g['planes'], g['coords'], *ws, scale=0.5, mlp_mode=marcher, return_taps=True)
    np.testing.assert_array_equal(out['taps'].cpu().numpy(), ref['taps'])        # INT row: bilinear tap indices


def test_field_hot_shape(tdgp, oracle):
    """feat 32 / hid 64 (the 3dgp.yaml shape), planes 64^2, points beyond the cube (zero padding), ragged point count."""
    rs = np.random.RandomState(9)
    B, F, H, hid, P = 2, 32, 64, 64, 16 * 37 + 5
    planes = rs.randn(B, 3 * F, H, H).astype(np.float32)
    coords = ((rs.rand(B, P, 3) * 2 - 1) * 0.64).astype(np.float32)
    ws = [rs.randn(hid, F).astype(np.float32), (0.3 * rs.randn(hid)).astype(np.float32), rs.randn(4, hid).astype(np.float32),
          (0.3 * rs.randn(4)).astype(np.float32)]
    out = tdgp.renderer.simple_tri_plane_renderer(T(planes), T(coords), _mlp(tdgp, *ws, 'classical'), scale=0.5, return_taps=True)
    ref = oracle.triplane_field(planes, coords, *ws, scale=0.5, return_taps=True)
    np.testing.assert_array_equal(out['taps'].cpu().numpy(), ref['taps'])
    assert_close(N(out['rgb']), ref['rgb'], 5e-6, 'rgb', 1.0)
    assert_close(N(out['sigma']), ref['sigma'], 5e-6, 'sigma', 1.0)
    assert tdgp.renderer.simple_tri_plane_renderer(T(planes), T(coords[:, :0]), _mlp(tdgp, *ws, 'classical'), scale=0.5)['rgb'].shape == (B, 0, 3)


@pytest.mark.parametrize('B,h,w,S,F,hid,marcher', [(2, 16, 24, 16, 32, 64, 'classical'), (3, 20, 12, 32, 32, 64, 'mip'), (1, 64, 64, 64, 32, 64, 'classical'),
                                                     (2, 9, 17, 20, 16, 32, 'classical'), (1, 24, 24, 24, 32, 128, 'classical'), (2, 32, 32, 96, 32, 64, 'classical'),
                                                     (2, 128, 136, 16, 32, 64, 'classical')])
def test_field_image_walk_equals_linear_order(tdgp, B, h, w, S, F, hid, marcher):
    """The image walk (ray_w > 0; the producer / consumer kernel of field_walk2.inc when S % 4 == 0 and S >= 16) runs the same arithmetic
    in the same order as the linear-point-order kernel: (r, g, b, sigma) bit for bit, the tap-index rows too -- including ray images that
    are not whole 8x8 patches, depths that leave the cube (zero padding), density noise, and more patches than blocks."""
    rs = np.random.RandomState(S * 7 + h)
    H = 48
    planes = T(rs.randn(B, 3 * F, H, H))
    mlp = _mlp(tdgp, rs.randn(hid, F), 0.3 * rs.randn(hid), rs.randn(4, hid), 0.3 * rs.randn(4), marcher)
    cam = dict(angles=T(np.stack([rs.uniform(-1, 1, B), rs.uniform(1.0, 2.0, B), np.zeros(B)], 1)), radius=T(np.ones(B)), look_at=T(np.zeros((B, 3))))
    ro, rd = tdgp.renderer.sample_rays(tdgp.renderer.compute_cam2world_matrix(cam), T(rs.uniform(15, 45, B)), (w, h))
    R = h * w
    assert ro.shape == (B, R, 3)
    t = T(np.sort(rs.uniform(0.4, 1.6, (B, R, S)), axis=2))                   # beyond the cube at both ends
    hw = tdgp.renderer.planes_to_hwc(planes)
    mp = tdgp.renderer._mlp_params(mlp)
    n = T(rs.randn(B, R * S))
    for kw in (dict(), dict(sigma_noise=n, density_noise=0.8)):
        lin = tdgp.renderer._field(hw, mp, 0.5, ray_o=ro, ray_d=rd, t=t, ray_w=0, **kw)
        img = tdgp.renderer._field(hw, mp, 0.5, ray_o=ro, ray_d=rd, t=t, ray_w=w, **kw)
        np.testing.assert_array_equal(N(img), N(lin))
    taps_l = torch.full([B, R * S, 3, 2], -7, dtype=torch.int32, device=DEV)
    taps_i = torch.full([B, R * S, 3, 2], -7, dtype=torch.int32, device=DEV)
    lin = tdgp.renderer._field(hw, mp, 0.5, ray_o=ro, ray_d=rd, t=t, ray_w=0, tap_idx=taps_l)
    img = tdgp.renderer._field(hw, mp, 0.5, ray_o=ro, ray_d=rd, t=t, ray_w=w, tap_idx=taps_i)
    np.testing.assert_array_equal(N(img), N(lin))
    np.testing.assert_array_equal(taps_i.cpu().numpy(), taps_l.cpu().numpy())
    assert int((taps_i == -7).sum()) == 0


# ------------------------------------------------------------------------------------------------ sampling stages

@pytest.mark.parametrize('marcher', ['classical', 'mip'])
def test_stratified_importance(tdgp, oracle, marcher):
    g = load_golden('sampling')
    R = tdgp.renderer.ImportanceRenderer(marcher)
    u = g[f'{marcher}_u_coarse']
    sd = R.sample_stratified(T(np.zeros(u.shape[:2] + (3,), np.float32)), 0.0, 1.0, u.shape[2], noise=T(u))
    np.testing.assert_array_equal(N(sd), g[f'{marcher}_sdist'])                  # bit-exact vs the reference
    sf, aux = R.sample_importance(T(g[f'{marcher}_sdist']), T(g[f'{marcher}_weights']), u.shape[2], u=T(g[f'{marcher}_u_fine']), return_aux=True)
    osf, oaux = oracle.sample_importance(g[f'{marcher}_sdist'], g[f'{marcher}_weights'], g[f'{marcher}_u_fine'], marcher, return_aux=True)
    for k in ('inds', 'below', 'above'):                                         # INT rows: bit-exact vs the REFERENCE and the oracle
        np.testing.assert_array_equal(aux[k].cpu().numpy().astype(np.int64), g[f'{marcher}_{k}'])
        np.testing.assert_array_equal(aux[k].cpu().numpy().astype(np.int64), oaux[k])
    np.testing.assert_array_equal(N(aux['cdf']), oaux['cdf'])
    np.testing.assert_array_equal(N(sf), osf)
    np.testing.assert_array_equal(N(sf), g[f'{marcher}_sdist_fine'])             # the fine samples themselves: bit-exact vs the reference


@pytest.mark.parametrize('marcher', ['classical', 'mip'])
@pytest.mark.parametrize('S', [32, 48, 64, 96])
def test_importance_hot_sizes(tdgp, marcher, S):
    """sample_importance at the ray-step counts of BASELINE configs[0..4] against vectors captured from the reference: 0 integer
    mismatches, fine samples bit-identical (the pdf normaliser follows torch's CPU sum order in the kernel)."""
    g = load_golden('sampling_hot')
    tag = f'{marcher}{S}'
    R = tdgp.renderer.ImportanceRenderer(marcher)
    sf, aux = R.sample_importance(T(g[f'{tag}_sdist']), T(g[f'{tag}_weights']), S, u=T(g[f'{tag}_u_fine']), return_aux=True)
    np.testing.assert_array_equal(aux['inds'].cpu().numpy().astype(np.int64), g[f'{tag}_inds'].astype(np.int64))
    np.testing.assert_array_equal(N(sf), g[f'{tag}_sdist_fine'])


def test_importance_stage_of_e2e(tdgp):
    """The importance-sampling stage with the arguments the reference's forward passed it (captured at tri_plane_renderer.py:153)."""
    g = load_golden('e2e_tiny')
    R = tdgp.renderer.ImportanceRenderer('classical')
    S = g['imp_sdist'].shape[2]
    sf, aux = R.sample_importance(T(g['imp_sdist']), T(g['imp_weights']), S, u=T(g['u_fine']), return_aux=True)
    np.testing.assert_array_equal(aux['inds'].cpu().numpy().astype(np.int64), g['inds'])
    np.testing.assert_array_equal(N(sf), g['imp_sdist_fine'])


def test_unify(tdgp):
    g = load_golden('sampling')
    R = tdgp.renderer.ImportanceRenderer('classical')
    d, c, s, perm = R.unify_samples(*(T(g[k]) for k in ('un_d1', 'un_c1', 'un_s1', 'un_d2', 'un_c2', 'un_s2')), return_perm=True)
    np.testing.assert_array_equal(perm.cpu().numpy().astype(np.int64), g['un_perm'])   # INT row: sort permutation
    np.testing.assert_array_equal(N(d), g['un_d'])
    np.testing.assert_array_equal(N(c), g['un_c'])
    np.testing.assert_array_equal(N(s), g['un_s'])


@pytest.mark.parametrize('tag,kw', [('cl_inf', dict(use_inf_depth=True)), ('cl_noinf', dict(use_inf_depth=False)),
                                    ('cl_lastback', dict(use_inf_depth=True, last_back=True)), ('cl_relu', dict(use_inf_depth=True, clamp_mode='relu')),
                                    ('cl_cut', dict(use_inf_depth=True, cut_quantile=0.5))])
def test_march_classical(tdgp, oracle, tag, kw):
    g = load_golden('marchers')
    rgb, dep, w, fT = tdgp.renderer.ClassicalRayMarcher()(T(g['colors']), T(g['densities']), T(g['depths']), dict(kw))
    assert_close(N(w), g[f'{tag}_weights'], 1e-6, 'weights', 1.0)
    assert_close(N(rgb), g[f'{tag}_rgb'], 5e-6, 'rgb', 1.0)
    assert_close(N(dep), g[f'{tag}_depth'], 5e-6, 'depth', 1.0)
    assert_close(N(fT), g[f'{tag}_T'], 2e-6, 'T')
    orgb, odep, ow, ofT = oracle.march_classical(g['colors'], g['densities'], g['depths'], **kw)
    # the marchers run the reference's fp32 chain on <= 1-ulp transcendentals and fp32 wave scans; the oracle rounds the exact
    # value once: a few ulp apart, like torch's own Sleef-based result is from either
    assert_close(N(w), ow, 1e-6, 'weights vs oracle', 1.0)
    assert_close(N(fT), ofT, 2e-6, 'T vs oracle')
    assert_close(N(rgb), orgb, 1e-6, 'rgb vs oracle', 1.0)


def test_density_activation(tdgp):
    """tdgp_density_activation = the marchers' own softplus (max(x,0) + compensated log1p(exp(-|x|)); threshold 20) / relu: within 3 ulp of
    the exactly rounded softplus over the whole range (measured max 2.7, tools/dev/softplus_acc.hip; torch's CPU softplus: 1.5), relu exact,
    and it is what the cut threshold is taken over: a march with cut_quantile zeroes exactly the samples whose activation is below the
    quantile of the activations this entry returns."""
    L = tdgp._lib
    rs = np.random.RandomState(8)
    x = np.concatenate([np.linspace(-30, 25, 200001), rs.randn(100000) * 5]).astype(np.float32)
    dx = T(x)
    out = torch.empty_like(dx)
    L.call('tdgp_density_activation', dx.data_ptr(), out.data_ptr(), dx.numel(), 0, 0.0, L.stream_of(dx))
    true = np.where(x > 20, x.astype(np.float64), np.log1p(np.exp(x.astype(np.float64))))
    ulp = np.spacing(np.abs(true.astype(np.float32))).astype(np.float64)
    assert (np.abs(N(out).astype(np.float64) - true) / ulp).max() <= 3.0
    L.call('tdgp_density_activation', dx.data_ptr(), out.data_ptr(), dx.numel(), 8, 0.0, L.stream_of(dx))
    np.testing.assert_array_equal(N(out), np.maximum(x, 0))
    L.call('tdgp_density_activation', dx.data_ptr(), out.data_ptr(), dx.numel(), 0, -1.5, L.stream_of(dx))
    ref = torch.empty_like(dx)
    sh = T((x + np.float32(-1.5)).astype(np.float32))
    L.call('tdgp_density_activation', sh.data_ptr(), ref.data_ptr(), dx.numel(), 0, 0.0, L.stream_of(dx))
    np.testing.assert_array_equal(N(out), N(ref))                                   # the bias is added in fp32, as the mip marcher adds it
    # self-consistency of the cut: weights vanish exactly where activation < quantile(activation)
    g = load_golden('marchers')
    dens = T(g['densities'])
    act = torch.empty_like(dens)
    L.call('tdgp_density_activation', dens.data_ptr(), act.data_ptr(), dens.numel(), 0, 0.0, L.stream_of(dens))
    thr = float(torch.quantile(act.reshape(-1), 0.5))
    _, _, w, _ = tdgp.renderer.ClassicalRayMarcher()(T(g['colors']), dens, T(g['depths']), dict(use_inf_depth=True, cut_quantile=0.5))
    cut = N(act).reshape(-1) < thr
    assert (N(w).reshape(-1)[cut] == 0).all() and cut.sum() == cut.size // 2


@pytest.mark.parametrize('tag,kw', [('mip_inf', dict(use_inf_depth=True)), ('mip_noinf_white', dict(use_inf_depth=False, white_back=True)),
                                    ('mip_bias', dict(use_inf_depth=True, density_bias=-1.0)),
                                    ('mip_cut', dict(use_inf_depth=True, cut_quantile=0.3))])
def test_march_mip(tdgp, tag, kw):
    g = load_golden('marchers')
    rgb, dep, w, fT = tdgp.renderer.MipRayMarcher2()(T(g['colors01']), T(g['densities']), T(g['depths']), dict(kw))
    assert_close(N(w), g[f'{tag}_weights'], 1e-6, 'weights', 1.0)
    assert_close(N(rgb), g[f'{tag}_rgb'], 5e-6, 'rgb', 1.0)
    assert_close(N(dep), g[f'{tag}_depth'], 5e-6, 'depth', 1.0)
    assert_close(N(fT), g[f'{tag}_T'], 2e-6, 'T')


@pytest.mark.parametrize('S', [16, 32, 48])
def test_fused_chain_equals_op_level_chain(tdgp, S):
    """ImportanceRenderer.forward (fused kernels, fine samples pre-sorted, sorted-merge fast path) must equal the same chain
    issued stage by stage through the reference-named methods (sample_stratified -> run_model -> ray_marcher ->
    sample_importance -> run_model -> unify_samples -> ray_marcher), for both marchers."""
    rs = np.random.RandomState(3)
    B, F, H, hid, hw = 2, 8, 32, 16, 12
    planes = T(rs.randn(B, 3 * F, H, H))
    for marcher in ('classical', 'mip'):
        mlp = _mlp(tdgp, rs.randn(hid, F), 0.3 * rs.randn(hid), rs.randn(4, hid), 0.3 * rs.randn(4), marcher)
        cam = dict(angles=T([[0.3, 1.2, 0.0], [-0.6, 1.8, 0.0]]), radius=T([1.0, 1.0]), look_at=T(np.zeros((2, 3))))
        ro, rd = tdgp.renderer.sample_rays(tdgp.renderer.compute_cam2world_matrix(cam), T([25.0, 40.0]), (hw, hw))
        R = hw * hw
        u1, u2 = T(rs.rand(B, R, S, 1)), T(rs.rand(B * R, S))
        opts = dict(box_size=1.0, num_proposal_steps=S, num_fine_steps=S, clamp_mode='softplus', use_inf_depth=True, ray_start=0.75, ray_end=1.25,
                    white_back=(marcher == 'mip'), density_bias=0.0)
        rend = tdgp.renderer.ImportanceRenderer(marcher)
        rgb, depth, wsum, fT = rend(planes, mlp, ro, rd, dict(opts, u_coarse=u1, u_fine=u2))
        # stage by stage
        s2t = lambda s: s * opts['ray_end'] + (1 - s) * opts['ray_start']     # noqa: E731
        sd = rend.sample_stratified(ro, 0.0, 1.0, S, noise=u1)
        td = s2t(sd)
        pts = (ro.unsqueeze(-2) + td * rd.unsqueeze(-2)).reshape(B, -1, 3)
        out = rend.run_model(planes, mlp, pts, opts)
        cc, dc = out['rgb'].reshape(B, R, S, 3), out['sigma'].reshape(B, R, S, 1)
        _, _, w, _ = rend.ray_marcher(cc, dc, sd, opts)
        sf = rend.sample_importance(sd, w, S, u=u2)
        tf = s2t(sf)
        out = rend.run_model(planes, mlp, (ro.unsqueeze(-2) + tf * rd.unsqueeze(-2)).reshape(B, -1, 3), opts)
        cf, df = out['rgb'].reshape(B, R, S, 3), out['sigma'].reshape(B, R, S, 1)
        d_all, c_all, s_all = rend.unify_samples(td, cc, dc, tf, cf, df)
        rgb2, depth2, w2, fT2 = rend.ray_marcher(c_all, s_all, d_all, opts)
        np.testing.assert_array_equal(N(rgb), N(rgb2))
        np.testing.assert_array_equal(N(depth), N(depth2))
        np.testing.assert_array_equal(N(fT), N(fT2))
        assert_close(N(wsum), N(w2.sum(2)), 1e-6, 'weights.sum', 1.0)


@pytest.mark.parametrize('marcher', ['classical', 'mip'])
def test_render_fused_equals_staged_calls(tdgp, marcher):
    """tdgp_render_fused (ImportanceRenderer.forward as ONE C-ABI call, include/tdgp.h) against the staged entry points it stands for
    (tdgp_sample_stratified -> tdgp_triplane_field -> tdgp_importance_from_coarse -> tdgp_triplane_field -> tdgp_merge_composite): rgb,
    depth, weights.sum and final transmittance bit for bit, at an image-shaped and at a ragged ray count; plus its argument checks."""
    rs = np.random.RandomState(11)
    F, H, hid, S = 32, 64, 64, 16
    for B, hw, ray_w in ((2, 24 * 16, 24), (1, 301, 0)):
        planes = T(rs.randn(B, 3 * F, H, H))
        mlp = _mlp(tdgp, rs.randn(hid, F), 0.3 * rs.randn(hid), rs.randn(4, hid), 0.3 * rs.randn(4), marcher)
        cam = dict(angles=T([[0.3, 1.2, 0.0], [-0.6, 1.8, 0.0]][:B]), radius=T([1.0, 1.0][:B]), look_at=T(np.zeros((B, 3))))
        side = 24
        ro, rd = tdgp.renderer.sample_rays(tdgp.renderer.compute_cam2world_matrix(cam), T([25.0, 40.0][:B]), (side, side))
        ro, rd = ro[:, :hw].contiguous(), rd[:, :hw].contiguous()
        u1, u2 = T(rs.rand(B, hw, S, 1)), T(rs.rand(B * hw, S))
        opts = dict(box_size=1.0, num_proposal_steps=S, num_fine_steps=S, clamp_mode='softplus', use_inf_depth=True, ray_start=0.75, ray_end=1.25,
                    white_back=(marcher == 'mip'), density_bias=0.0, u_coarse=u1, u_fine=u2, ray_grid_w=ray_w)
        rend = tdgp.renderer.ImportanceRenderer(marcher)
        assert rend.fused_entry
        _, launches = _wino_launches(tdgp, lambda: rend(planes, mlp, ro, rd, opts))
        fused = rend(planes, mlp, ro, rd, opts)
        rend.fused_entry = False
        staged = rend(planes, mlp, ro, rd, opts)
        for a, b, name in zip(fused, staged, ('rgb', 'depth', 'wsum', 'final_T')):
            assert torch.equal(a, b), (marcher, B, hw, name)
        assert launches.get('triplane_field_kernel') == 2 and launches.get('merge_composite_kernel') == 1 and launches.get('stratified_kernel') == 1, launches
    L = tdgp._lib
    assert L.load().tdgp_render_fused_workspace_bytes(2, 100, 16, 16) == 200 * (2 * 16 * 4 + 16 * 16 + 16 * 4 + 16 * 16)      # sdist, tdist, rgbs_c, tfine, rgbs_f
    with pytest.raises(RuntimeError, match='workspace'):
        ws = torch.empty(16, device=DEV)
        p_ = tdgp.renderer.planes_to_hwc(planes).t
        w0, b0, w1, b1, _ = tdgp.renderer._mlp_params(mlp)
        out = [torch.empty(hw * B * 3, device=DEV) for _ in range(4)]
        L.call('tdgp_render_fused', p_.data_ptr(), w0.data_ptr(), b0.data_ptr(), w1.data_ptr(), b1.data_ptr(), ro.data_ptr(), rd.data_ptr(), u1.data_ptr(), u2.data_ptr(),
               out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), B, hw, 0, S, S, F, H, H, hid, 0.5, 0.75, 1.25, 0, 1, 0.0, ws.data_ptr(), 64,
               L.stream_of(ws))


def _merge_case(tdgp, rs, rays, S, sorted_lists, flags_kw, cut=0.0, with_perm2=False, S2=None):
    """tdgp_merge_composite on explicit lists (S coarse + S2 fine samples; S2 defaults to S) vs the op-level chain unify_samples -> ray
    marcher, everything bit for bit."""
    S2 = S if S2 is None else S2
    L = tdgp._lib
    t1 = rs.uniform(0.75, 1.25, (rays, S)).astype(np.float32)
    t2 = rs.uniform(0.75, 1.25, (rays, S2)).astype(np.float32)
    if sorted_lists:
        t1.sort(axis=1)
        t2.sort(axis=1)
        nt = len(range(0, min(S, S2), 5))
        t2[:, 0:5 * nt:5] = t1[:, 0:5 * nt:5]                         # ties between the lists: coarse first
        t2.sort(axis=1)
    else:
        t1[::3].sort(axis=1)                                          # a mix of ascending and arbitrary lists inside one tile
        t2[1::2].sort(axis=1)
        if rays > 5 and S > 7:
            t1[5, 3] = t1[5, 7]                                       # ties inside a list
    c1, c2 = rs.randn(rays, S, 4).astype(np.float32), rs.randn(rays, S2, 4).astype(np.float32)
    c1[..., 3] *= 4
    c2[..., 3] *= 4
    perm2 = np.stack([rs.permutation(S2) for _ in range(rays)]).astype(np.int32) if with_perm2 else None
    flags = tdgp.renderer._marcher_flags(dict(flags_kw), 'classical')
    dt1, dt2, dc1, dc2 = T(t1), T(t2), T(c1), T(c2)
    rgb, dep, wsum, fT = (torch.empty(rays, n, device=DEV) for n in (3, 1, 1, 1))
    perm = torch.empty(rays, S + S2, dtype=torch.int32, device=DEV)
    dperm2 = torch.as_tensor(perm2).to(DEV) if with_perm2 else None
    L.call('tdgp_merge_composite', dc1.data_ptr(), dt1.data_ptr(), S, dc2.data_ptr(), dt2.data_ptr(), S2, rgb.data_ptr(), dep.data_ptr(), wsum.data_ptr(),
           fT.data_ptr(), perm.data_ptr(), L.ptr(dperm2), rays, 0, flags, 0.0, float(cut), L.stream_of(dt1))
    rend = tdgp.renderer.ImportanceRenderer('classical')
    sh = lambda a, c: a.reshape(1, rays, -1, c)                       # noqa: E731
    d, c, sg, uperm = rend.unify_samples(sh(dt1, 1), sh(dc1[..., :3].contiguous(), 3), sh(dc1[..., 3].contiguous(), 1),
                                         sh(dt2, 1), sh(dc2[..., :3].contiguous(), 3), sh(dc2[..., 3].contiguous(), 1), return_perm=True)
    orgb = torch.empty(1, rays, 3, device=DEV)
    odep, ow, ofT = torch.empty(1, rays, 1, device=DEV), torch.empty(1, rays, S + S2, 1, device=DEV), torch.empty(1, rays, device=DEV)
    L.call('tdgp_ray_march', c.data_ptr(), sg.data_ptr(), d.data_ptr(), orgb.data_ptr(), odep.data_ptr(), ow.data_ptr(), ofT.data_ptr(), rays, S + S2, 3, 0,
           flags, 0.0, float(cut), L.stream_of(dt1))
    tag = f'S={S}+{S2} rays={rays} sorted={sorted_lists} {flags_kw} cut={cut}'
    want_perm = uperm.reshape(rays, S + S2).cpu().numpy().astype(np.int64)
    if with_perm2:
        want_perm = np.where(want_perm < S, want_perm, S + np.take_along_axis(perm2.astype(np.int64), np.clip(want_perm - S, 0, S2 - 1), axis=1))
    np.testing.assert_array_equal(perm.cpu().numpy().astype(np.int64), want_perm, err_msg=tag)
    np.testing.assert_array_equal(N(rgb), N(orgb).reshape(rays, 3), err_msg=tag)
    np.testing.assert_array_equal(N(dep), N(odep).reshape(rays, 1), err_msg=tag)
    np.testing.assert_array_equal(N(fT).reshape(-1), N(ofT).reshape(-1), err_msg=tag)
    assert_close(N(wsum).reshape(-1), N(ow).reshape(rays, S + S2).astype(np.float64).sum(1), 2e-6, 'weights.sum ' + tag, 1.0)


@pytest.mark.parametrize('S', [16, 32, 48, 64, 96])
def test_merge_composite_equals_unify_then_march(tdgp, S):
    """The fused merge + march + composite (merge_composite_kernel: one wave per ray, MS = 64 / 128 / 256 slots) against the op-level
    chain, bit for bit: ascending lists with ties across them, lists that are NOT ascending (brute-force rank path; mixed with ascending
    ones in one launch), every marcher flag, a cut threshold, and the permutation with and without the fine list's own draw permutation."""
    rs = np.random.RandomState(100 + S)
    _merge_case(tdgp, rs, 256, S, True, dict(use_inf_depth=True), with_perm2=True)
    _merge_case(tdgp, rs, 150, S, True, dict(use_inf_depth=False, last_back=True))
    _merge_case(tdgp, rs, 70, S, True, dict(use_inf_depth=True, clamp_mode='relu'), cut=0.3)
    _merge_case(tdgp, rs, 200, S, False, dict(use_inf_depth=True), with_perm2=True)
    _merge_case(tdgp, rs, 1, S, False, dict(use_inf_depth=True, last_back=True))


@pytest.mark.parametrize('S1,S2', [(96, 32), (48, 80), (64, 1), (1, 64), (33, 31), (64, 96), (20, 100)])
def test_merge_composite_with_lists_of_different_lengths(tdgp, S1, S2):
    """the C ABI takes S1 != S2 (tdgp.h: tdgp_merge_composite) but every other case merges equal lists.  The rank computation
    (fine elements binary-search the coarse list; coarse ranks = prefix maximum over an LDS max-scatter of the fine counts) and the two-chunk
    march have paths that only unequal lists reach: `64 cc < S1` with partial 64-slot chunks, the MS = 128 form with S1 != S2, a list of one."""
    rs = np.random.RandomState(1000 * S1 + S2)
    _merge_case(tdgp, rs, 130, S1, True, dict(use_inf_depth=True), with_perm2=True, S2=S2)
    _merge_case(tdgp, rs, 77, S1, True, dict(use_inf_depth=False, last_back=True), cut=0.3, S2=S2)
    _merge_case(tdgp, rs, 40, S1, False, dict(use_inf_depth=True, clamp_mode='relu'), with_perm2=True, S2=S2)


@pytest.mark.parametrize('S', [16, 48, 64, 96, 128])
def test_fused_fine_samples_come_out_sorted(tdgp, S):
    """importance_from_coarse writes the fine samples depth-sorted (lane sort for N <= 64, the two-per-lane network for N <= 128, brute
    force beyond) and merge_composite's fast path relies on it -- it falls back to a brute-force rank when the list is NOT sorted, so a
    broken sort costs time, not correctness, and only this test sees it."""
    rs = np.random.RandomState(S)
    B, F, H, hid, hw = 1, 8, 32, 16, 16
    planes = T(rs.randn(B, 3 * F, H, H))
    mlp = _mlp(tdgp, rs.randn(hid, F), 0.3 * rs.randn(hid), rs.randn(4, hid), 0.3 * rs.randn(4), 'classical')
    cam = dict(angles=T([[0.3, 1.2, 0.0]]), radius=T([1.0]), look_at=T(np.zeros((1, 3))))
    ro, rd = tdgp.renderer.sample_rays(tdgp.renderer.compute_cam2world_matrix(cam), T([30.0]), (hw, hw))
    R = hw * hw
    u_fine = rs.rand(B * R, S).astype(np.float32)
    u_fine[::3, 5] = u_fine[::3, 2]                 # equal draws -> equal depths: the counting sort's slots collide and those rays take the
    u_fine[1::7, S - 1] = u_fine[1::7, 0]           # network, whose (depth, draw index) order the last assertion checks
    opts = dict(box_size=1.0, num_proposal_steps=S, num_fine_steps=S, clamp_mode='softplus', use_inf_depth=True, ray_start=0.75, ray_end=1.25,
                u_coarse=T(rs.rand(B, R, S, 1)), u_fine=T(u_fine))
    rend = tdgp.renderer.ImportanceRenderer('classical')
    _, aux = rend(planes, mlp, ro, rd, opts, return_intermediates=True)
    tf = N(aux['tdist_fine']).reshape(R, S)
    assert (np.diff(tf, axis=1) >= 0).all(), 'fine samples not ascending'
    # the same multiset as the draw-order samples, and fine_perm names the draw each slot came from
    sf = N(aux['sdist_fine']).reshape(R, S)
    t_draw = (sf * np.float32(1.25) + (np.float32(1.0) - sf) * np.float32(0.75)).astype(np.float32)
    fp = N(aux['fine_perm']).reshape(R, S).astype(np.int64)
    assert (np.sort(fp, axis=1) == np.arange(S)[None]).all()
    np.testing.assert_allclose(np.take_along_axis(t_draw, fp, axis=1), tf, rtol=0, atol=1e-6)
    order = np.lexsort((np.broadcast_to(np.arange(S), (R, S)), np.take_along_axis(t_draw, fp, axis=1)))      # already sorted: identity
    assert (order == np.arange(S)[None]).all()


# ------------------------------------------------------------------------------------------------ end to end

def _gen(tdgp, cfg, seed):
    G = tdgp.generator.Generator(cfg)
    G.load_numpy_state_dict(tdgp.weights.random_state_dict(cfg, seed=seed, exercise_all=True))
    return G.to(DEV)


def _cam(g):
    return {k[4:]: T(v) for k, v in g.items() if k.startswith('cam_')}


def _exact(oracle, tdgp, cfg, seed, g):
    """The same image from the double-accumulating oracle: calibrates the reference's own fp32 noise (conftest.assert_image_parity)."""
    sd = tdgp.weights.random_state_dict(cfg, seed=seed, exercise_all=True)
    cam = {k[4:]: v for k, v in g.items() if k.startswith('cam_')}
    return oracle.synthesis_forward(sd, cfg.to_dict(), g['ws'], cam, g['u_coarse'], g['u_fine'], 'const')


def test_mapping(tdgp):
    g = load_golden('mapping')
    for tag, cfg in [('c0', tdgp.config.config_tiny()), ('c10', tdgp.config.config_mid())]:
        G = _gen(tdgp, cfg, 11)
        assert_close(N(G.mapping(T(g[f'{tag}_z']), T(g[f'{tag}_c']))), g[f'{tag}_ws'], 1e-5, 'ws', 1.0)
        assert_close(N(G.mapping(T(g[f'{tag}_z']), T(g[f'{tag}_c']), truncation_psi=0.7)), g[f'{tag}_ws_psi07'], 1e-5, 'psi', 1.0)
        assert_close(N(G.mapping(T(g[f'{tag}_z']), T(g[f'{tag}_c']), truncation_psi=0.3, truncation_cutoff=3)), g[f'{tag}_ws_psi03_cut3'], 1e-5, 'cut', 1.0)


def test_e2e_tiny(tdgp, oracle):
    cfg = tdgp.config.config_tiny()
    g = load_golden('e2e_tiny')
    G = _gen(tdgp, cfg, 21)
    ws = T(g['ws'])
    planes = G.synthesis.tri_plane_decoder(ws, noise_mode='const')
    assert_close(N(planes), g['planes'], 5e-6, 'planes (NCHW)', 1.0)
    hwc = G.synthesis.tri_plane_decoder(ws, noise_mode='const', hwc=True).t
    # the channel-last planes come from the fused ToRGB kernel, the NCHW ones from the generic 1x1 path: same algebra, different
    # K-step grouping inside the matrix cores -> equal to fp32 rounding, not bit for bit
    assert_close(N(hwc.permute(0, 1, 4, 2, 3).reshape(planes.shape)), N(planes), 2e-6, 'planes channel-last vs NCHW', 1.0)
    out = G.synthesis(ws, camera_params=_cam(g), noise_mode='const', render_opts=dict(return_depth=True), u_coarse=T(g['u_coarse']), u_fine=T(g['u_fine']))
    ex_img, ex_depth = _exact(oracle, tdgp, cfg, 21, g)
    assert_image_parity(N(out.img), g, 'e2e_tiny img', exact=ex_img)
    assert_image_parity(N(out.depth), g, 'e2e_tiny depth', 'depth', exact=ex_depth)
    img2 = G(T(g['z']), T(g['c']), _cam(g), noise_mode='const', u_coarse=T(g['u_coarse']), u_fine=T(g['u_fine']))
    assert_image_parity(N(img2), g, 'e2e_tiny img via Generator.forward', exact=ex_img)
    img3 = G.synthesis(ws, camera_params=_cam(g), noise_mode='none', u_coarse=T(g['u_coarse']), u_fine=T(g['u_fine']))
    assert_close(N(img3), g['img_noise_none'], 1e-5, 'img noise none', 1.0)
    # renderer integer rows on the golden planes: inds and sort permutation vs the oracle, exact
    from oracle.pipeline import render_options
    sd = tdgp.weights.random_state_dict(cfg, seed=21, exercise_all=True)
    mlp = tuple(sd[f'synthesis.tri_plane_mlp.model.{i}.{n}'] for i in (0, 1) for n in ('weight', 'bias'))
    (orgb, odep, _, _), ointer = oracle.importance_render(g['planes'], mlp, g['ray_o'], g['ray_d'], render_options(cfg.to_dict()), g['u_coarse'],
                                                          g['u_fine'], return_intermediates=True)
    opts = G.synthesis.rendering_options(G.synthesis._default_render_options)
    opts.update(u_coarse=T(g['u_coarse']), u_fine=T(g['u_fine']))
    (rgb, dep, _, _), inter = G.synthesis.renderer(T(g['planes']), G.synthesis.tri_plane_mlp, T(g['ray_o']), T(g['ray_d']), opts, return_intermediates=True)
    np.testing.assert_array_equal(N(inter['sdist_coarse']), ointer['sdist_coarse'])
    assert_close(N(rgb), orgb, 1e-5, 'renderer rgb vs oracle', 1.0)
    # Stage by stage the integer rows are bit-exact vs the reference (test_importance_stage_of_e2e, test_importance_hot_sizes,
    # test_unify: reference inputs in, reference integers out).  Through the whole CHAIN they cannot be promised: the densities
    # feeding the cdf come from an fp32 MLP whose summation order differs from MKL's sgemm (and torch's expf is Sleef's 1-ulp
    # routine), so a draw that sits within an ulp of a cdf knot may land on the other side.  Counted and recorded, bounded loosely.
    ni = int((inter['inds'].cpu().numpy().reshape(-1) != g['inds'].reshape(-1)).sum())
    npm = int((inter['perm'].cpu().numpy().reshape(-1) != g['perm'].reshape(-1)).sum())
    report_parity('e2e_tiny integer rows through the whole chain', inds_mismatches=ni, inds_total=int(g['inds'].size), perm_mismatches=npm,
                  perm_total=int(g['perm'].size))
    # measured: 0 / 4096 and 0 / 8192 (profiles/*_parity_report.json).  Bound = "a handful, each explained": a mismatching index may only
    # move to the neighbouring cdf interval (a draw within an ulp of a knot), a mismatching sort slot only swap with its neighbour.
    # ... and explained: the reference's own cdf (the oracle's importance stage on the reference's coarse weights reproduces
    # the reference's cdf bit for bit, test_importance_stage_of_e2e) vs the kernel's; every mismatching draw lies between the two values of
    # the knot it flips across.
    hi = inter['inds'].cpu().numpy().reshape(g['inds'].shape[0], -1).astype(np.int64)
    _, raux = oracle.sample_importance(g['imp_sdist'], g['imp_weights'], g['u_fine'], cfg.ray_marcher_type, return_aux=True)
    np.testing.assert_array_equal(raux['inds'].reshape(hi.shape), g['inds'].reshape(hi.shape))
    Bt, Rt = inter['sdist_coarse'].shape[:2]
    cdf_rows = [_hip_strip_cdf(tdgp, G, inter, b, np.arange(Rt), g['u_fine'].reshape(Bt, Rt, -1)[b])[0] for b in range(Bt)]
    assert_inds_mismatches_in_window(hi, g['inds'].reshape(hi.shape).astype(np.int64), g['u_fine'].reshape(hi.shape), raux['cdf'].reshape(hi.shape[0], -1),
                                     np.concatenate(cdf_rows), what='e2e_tiny vs the reference')
    assert ni <= 4, ni
    assert npm <= 4, npm


def test_e2e_tiny_cut_quantile(tdgp, oracle):
    """cut_quantile = 0.5 (the non-flatness score's rendering, non_flatness_score.py:9) through the fused renderer: both marcher calls
    threshold their activated densities at the global median (tri_plane_renderer.py:366-368); against the reference's image."""
    cfg = tdgp.config.config_tiny()
    g = load_golden('e2e_tiny')
    G = _gen(tdgp, cfg, 21)
    out = G.synthesis(T(g['ws']), camera_params=_cam(g), noise_mode='const', render_opts=dict(return_depth=True, cut_quantile=0.5),
                      u_coarse=T(g['u_coarse']), u_fine=T(g['u_fine']))
    assert_close(N(out.img), g['img_cut'], 1e-5, 'img, cut_quantile 0.5', 1.0)
    assert_close(N(out.depth), g['depth_cut'], 1e-5, 'depth, cut_quantile 0.5', 1.0)
    # mip marcher: the threshold is taken over mid-point densities in MERGED order (two merge passes); vs the op-level chain
    cfg.ray_marcher_type = 'mip'
    G2 = _gen(tdgp, cfg, 41)
    a = G2.synthesis(T(g['ws']), camera_params=_cam(g), noise_mode='const', render_opts=dict(return_depth=True, cut_quantile=0.4),
                     u_coarse=T(g['u_coarse']), u_fine=T(g['u_fine']))
    sd = tdgp.weights.random_state_dict(cfg, seed=41, exercise_all=True)
    c = cfg.to_dict()
    c['cut_quantile'] = 0.4
    oimg, odepth = oracle.synthesis_forward(sd, c, g['ws'], {k[4:]: v for k, v in g.items() if k.startswith('cam_')}, g['u_coarse'], g['u_fine'], 'const')
    assert_close(N(a.img), oimg, 1e-5, 'mip img, cut_quantile 0.4, vs oracle', 1.0)
    assert_close(N(a.depth), odepth, 1e-5, 'mip depth, cut_quantile 0.4, vs oracle', 1.0)


def test_cut_quantile_above_max_batch_res_is_chunked_by_rays(tdgp):
    """eval + cut_quantile above max_batch_res = the reference's ray chunks with per-chunk quantiles
    (networks_epigraf.py:232-239); golden from the reference at 4 x 128^2 x 96 (chunks of 14563 rays)."""
    cfg = tdgp.config.config_cut_chunked()
    g = load_golden('cut_chunked')
    seed, batch, step = (int(v) for v in g['seed'])
    G = _gen(tdgp, cfg, seed)
    inp = tdgp.weights.synthetic_inputs(cfg, batch=batch, seed=seed)
    cam = {k: T(v) for k, v in inp['camera'].items()}
    out = G.synthesis(T(g['ws']), camera_params=cam, noise_mode='const', render_opts=dict(return_depth=True, cut_quantile=0.5),
                      u_coarse=T(inp['u_coarse']), u_fine=T(inp['u_fine']))
    assert_close_up_to_threshold_flips(N(out.img), g['img_cut'], 'img, cut_quantile 0.5, ray-chunked')
    assert_close_up_to_threshold_flips(N(out.depth), g['depth_cut'], 'depth, cut_quantile 0.5, ray-chunked')
    one = G.synthesis(T(g['ws']), camera_params=cam, noise_mode='const', render_opts=dict(return_depth=True, cut_quantile=0.5, max_batch_res=128),
                      u_coarse=T(inp['u_coarse']), u_fine=T(inp['u_fine']))
    assert float((one.img - out.img).abs().max()) > 1e-3 * float(np.abs(g['img_cut']).max())      # a global quantile is a different image


def test_graph_replay_equals_eager_forward(tdgp):
    """3dgp_amd/graphs.py: the whole forward captured as one HIP graph.  A replay runs the very kernels of the eager forward on the
    static buffers: same image bit for bit, for new inputs too (nothing is cached across replays); with device-side draws every replay
    renders a different image."""
    cfg = tdgp.config.config_mid()
    G = _gen(tdgp, cfg, 31)
    gg = tdgp.graphs.GraphedGenerator(G, 2, noise_mode='const', explicit_draws=True)
    for seed in (1, 2):
        inp = tdgp.weights.synthetic_inputs(cfg, batch=2, seed=seed)
        cam = {k: T(v) for k, v in inp['camera'].items()}
        eager = G(T(inp['z']), T(inp['c']), cam, noise_mode='const', u_coarse=T(inp['u_coarse']), u_fine=T(inp['u_fine']))
        replay = gg(T(inp['z']), T(inp['c']), cam, T(inp['u_coarse']), T(inp['u_fine']))
        assert torch.equal(eager, replay)
    gd = tdgp.graphs.GraphedGenerator(G, 2, noise_mode='random', explicit_draws=False)
    a = gd(T(inp['z']), T(inp['c']), cam).clone()
    b = gd(T(inp['z']), T(inp['c']), cam).clone()
    assert torch.isfinite(a).all() and not torch.equal(a, b)
    assert float((a - b).abs().max()) < 0.5 * float(a.abs().max())          # same scene, different draws
    with pytest.raises(RuntimeError):
        gd(T(inp['z']), T(inp['c']), cam, T(inp['u_coarse']), T(inp['u_fine']))


def test_batched_demod_equals_per_layer(tdgp):
    """tdgp_demod_batch (all layers' demodulation coefficients in one launch) is the same arithmetic as the per-call path of
    tdgp_modconv2d: coefficient tables and the backbone output must agree bit for bit."""
    cfg = tdgp.config.config_mid()
    G = _gen(tdgp, cfg, 5)
    dec = G.synthesis.tri_plane_decoder
    ws = T(np.random.RandomState(2).randn(3, dec.num_ws, cfg.w_dim))
    styles = dec.all_styles(ws)
    dcoefs = dec.all_demods(3)
    layers = [(l, s) for (l, _, _), s in zip(dec._layers(), styles) if isinstance(l, tdgp.generator.SynthesisLayer)]
    assert len(layers) == len(dcoefs)
    from importlib import import_module
    mc = import_module('3dgp_amd.ops.modconv')
    for (layer, st), d in zip(layers, dcoefs):
        wsq = N(layer.weight.detach().double().square().sum([2, 3]))                    # [Cout,Cin]
        want = 1.0 / np.sqrt((N(st).astype(np.float64) ** 2) @ wsq.T + 1e-8)
        assert_close(N(d), want, 1e-5, 'dcoef', 1.0)
        x = T(np.random.RandomState(7).randn(3, layer.in_channels, layer.resolution // layer.up, layer.resolution // layer.up))
        a = layer(x, None, noise_mode='const', styles=st, dcoef=d)
        b = layer(x, None, noise_mode='const', styles=st)
        np.testing.assert_array_equal(N(a), N(b))


def test_e2e_mid(tdgp, oracle):
    cfg = tdgp.config.config_mid()
    g = load_golden('e2e_mid')
    G = _gen(tdgp, cfg, 31)
    out = G.synthesis(T(g['ws']), camera_params=_cam(g), noise_mode='const', render_opts=dict(return_depth=True), u_coarse=T(g['u_coarse']),
                      u_fine=T(g['u_fine']))
    ex_img, ex_depth = _exact(oracle, tdgp, cfg, 31, g)
    assert_image_parity(N(out.img), g, 'e2e_mid img', exact=ex_img)
    assert_image_parity(N(out.depth), g, 'e2e_mid depth', 'depth', exact=ex_depth)


def test_e2e_tiny_mip(tdgp, oracle):
    cfg = tdgp.config.config_tiny()
    cfg.ray_marcher_type = 'mip'
    cfg.white_back = True
    g = load_golden('e2e_tiny_mip')
    G = _gen(tdgp, cfg, 41)
    out = G.synthesis(T(g['ws']), camera_params=_cam(g), noise_mode='const', render_opts=dict(return_depth=True), u_coarse=T(g['u_coarse']),
                      u_fine=T(g['u_fine']))
    ex_img, ex_depth = _exact(oracle, tdgp, cfg, 41, g)
    assert_image_parity(N(out.img), g, 'e2e_tiny_mip img', exact=ex_img)
    assert_image_parity(N(out.depth), g, 'e2e_tiny_mip depth', 'depth', exact=ex_depth)


def test_e2e_vs_oracle_bigger(tdgp, oracle):
    """A configuration with the hot MLP shape (feat 32, hid 64), 128^2 planes, 48^2 rays x 24 steps: HIP vs the reference's own image
    (golden e2e_bigger, round 5 -- before, this case was held to the oracle only and to a per-pixel figure of 1e-3 for want of a
    reference figure) through the same four bounds as the other end-to-end goldens, the oracle's image as the exactly rounded one."""
    cfg = tdgp.config.config_bigger()
    g = load_golden('e2e_bigger')
    G = _gen(tdgp, cfg, 5)
    ex_img, ex_depth = _exact(oracle, tdgp, cfg, 5, g)
    out = G.synthesis(T(g['ws']), camera_params=_cam(g), noise_mode='const', render_opts=dict(return_depth=True), u_coarse=T(g['u_coarse']), u_fine=T(g['u_fine']))
    assert_image_parity(N(out.img), g, 'e2e_bigger img', exact=ex_img, full_size=True)
    assert_image_parity(N(out.depth), g, 'e2e_bigger depth', 'depth', exact=ex_depth, full_size=True)


# ------------------------------------------------------------------------------------------------ full size (BASELINE configs[2])

@pytest.fixture(scope='module')
def full_c3(tdgp):
    """ImageNet 256^2 / 64 ray steps / cmax 512 generator with synthetic weights, one forward with intermediates."""
    cfg = tdgp.config.config_c3()
    sd = tdgp.weights.random_state_dict(cfg, seed=0, exercise_all=True)
    G = tdgp.generator.Generator(cfg)
    G.load_numpy_state_dict(sd)
    G = G.to(DEV)
    inp = tdgp.weights.synthetic_inputs(cfg, batch=2, seed=7)
    ws = G.mapping(T(inp['z']), T(inp['c']))
    planes = G.synthesis.tri_plane_decoder(ws, noise_mode='const', hwc=True)
    cam = {k: T(v) for k, v in inp['camera'].items()}
    return dict(cfg=cfg, sd=sd, G=G, inp=inp, ws=ws, planes=planes, cam=cam)


def test_full_size_backbone_vs_oracle(tdgp, oracle, full_c3):
    """All eight blocks at their real shapes (512 -> 64 channels, 4^2 -> 512^2, every tile configuration, split-K layers,
    sub-pixel phases with odd 2^k+1 grids) against the CPU oracle: one sample, ~10 s of host time."""
    cfg, sd = full_c3['cfg'], full_c3['sd']
    ws1 = N(full_c3['ws'])[:1]
    oracle.set_threads(min(64, __import__('os').cpu_count() or 1))
    ref = oracle.synthesis_backbone(sd, cfg.to_dict(), ws1, 'const')                     # [1, 96, 512, 512]
    got = N(full_c3['planes'].t[:1].permute(0, 1, 4, 2, 3).reshape(1, 96, 512, 512))
    assert_close(got, ref, 1e-5, 'tri-planes 512^2', 1.0)


def test_full_size_backbone_split_arith(tdgp, oracle, full_c3):
    """The opt-in split-bf16 arithmetic on the whole backbone at its real shapes (it takes the 64^2 ... 512^2 stride-1 layers): the
    tri-planes agree with the fp32-MFMA run to fp32 rounding and, on one sample, with the CPU oracle within the same tolerance as the
    default path."""
    G, ws = full_c3['G'], full_c3['ws']
    prev = tdgp._lib.set_conv_arith(1)
    try:
        planes = G.synthesis.tri_plane_decoder(ws, noise_mode='const', hwc=True)
    finally:
        tdgp._lib.set_conv_arith(prev)
    a, b = N(planes.t), N(full_c3['planes'].t)
    assert not np.array_equal(a, b)
    assert_close(a, b, 5e-6, 'split vs fp32 tri-planes', 1.0)
    cfg, sd = full_c3['cfg'], full_c3['sd']
    oracle.set_threads(min(64, __import__('os').cpu_count() or 1))
    ref = oracle.synthesis_backbone(sd, cfg.to_dict(), N(ws)[:1], 'const')
    assert_close(a[:1].transpose(0, 1, 4, 2, 3).reshape(1, 96, 512, 512), ref, 1e-5, 'split tri-planes vs oracle', 1.0)


def test_full_size_backbone_direct_sums(tdgp, oracle, full_c3):
    """`set_conv_arith(2)` (direct fp32 sums in every 3x3 layer, no Winograd): tri-planes agree with the default run to fp32 rounding
    and with the CPU oracle on one sample -- the two fp32 algorithms are interchangeable at the tolerance of the reduction rows."""
    G, ws = full_c3['G'], full_c3['ws']
    prev = tdgp._lib.set_conv_arith(2)
    try:
        planes = G.synthesis.tri_plane_decoder(ws, noise_mode='const', hwc=True)
    finally:
        tdgp._lib.set_conv_arith(prev)
    a, b = N(planes.t), N(full_c3['planes'].t)
    assert not np.array_equal(a, b)
    assert_close(a, b, 5e-6, 'direct-sum vs default (Winograd) tri-planes', 1.0)
    cfg, sd = full_c3['cfg'], full_c3['sd']
    oracle.set_threads(min(64, __import__('os').cpu_count() or 1))
    ref = oracle.synthesis_backbone(sd, cfg.to_dict(), N(ws)[:1], 'const')
    assert_close(a[:1].transpose(0, 1, 4, 2, 3).reshape(1, 96, 512, 512), ref, 1e-5, 'direct-sum tri-planes vs oracle', 1.0)


def test_full_size_renderer_strip_vs_oracle(tdgp, oracle, full_c3):
    """256^2 x (64 + 64) samples: the HIP renderer on the real 100 MB tri-planes; a strip of 3 x 256 rays is re-rendered by the
    oracle from the same planes and must agree (rays are independent, so a strip is a full-fidelity check)."""
    cfg, G, inp = full_c3['cfg'], full_c3['G'], full_c3['inp']
    out = G.synthesis(full_c3['ws'], camera_params=full_c3['cam'], noise_mode='const', render_opts=dict(return_depth=True),
                      u_coarse=T(inp['u_coarse']), u_fine=T(inp['u_fine']))
    img, depth = N(out.img), N(out.depth)
    assert img.shape == (2, 3, 256, 256) and np.isfinite(img).all()
    from oracle.pipeline import render_options
    sd = full_c3['sd']
    mlp = tuple(sd[f'synthesis.tri_plane_mlp.model.{i}.{n}'] for i in (0, 1) for n in ('weight', 'bias'))
    planes_nchw = N(full_c3['planes'].t.permute(0, 1, 4, 2, 3).reshape(2, 96, 512, 512))
    c2w = oracle.cam2world(inp['camera']['angles'], inp['camera']['radius'], inp['camera']['look_at'])
    ro, rd = oracle.sample_rays(c2w, inp['camera']['fov'], 256, 256)
    R, S = 256 * 256, cfg.num_ray_steps
    rows = [0, 131, 255]
    sel = np.concatenate([np.arange(r * 256, (r + 1) * 256) for r in rows])
    u1 = inp['u_coarse'].reshape(2, R, S)[:, sel]
    u2 = inp['u_fine'].reshape(2, R, S)[:, sel].reshape(-1, S)
    orgb, odepth, _, _ = oracle.importance_render(planes_nchw, mlp, ro[:, sel], rd[:, sel], render_options(cfg.to_dict()), u1, u2)
    got = img.reshape(2, 3, R)[:, :, sel].transpose(0, 2, 1)
    scale = np.abs(orgb).max()
    assert np.abs(got - orgb).max() / scale < 1e-5, np.abs(got - orgb).max() / scale
    assert np.abs(depth.reshape(2, R)[:, sel] - odepth[..., 0]).max() < 1e-5


def _hip_strip_cdf(tdgp, G, inter, b, sel, u2):
    """The cdf knots the HIP path ranked the importance draws of a ray strip against.  The fused kernel (tdgp_importance_from_coarse)
    keeps its cdf in LDS; the op-level pair (ray marcher -> tdgp_sample_importance, the reference's own staging) run on the SAME coarse
    field outputs reproduces it, which is checked on the spot: its `inds` must equal the fused kernel's on every draw of the strip."""
    syn = G.synthesis
    opts = syn.rendering_options(syn._default_render_options)
    rend = syn.renderer
    B, R, S = inter['sdist_coarse'].shape[:3]
    idx = torch.as_tensor(sel, device=DEV)
    rg = inter['rgbs_coarse'].reshape(B, R, S, 4)[b:b + 1, idx]
    sd = inter['sdist_coarse'].reshape(B, R, S, 1)[b:b + 1, idx]
    _, _, w, _ = rend.ray_marcher(rg[..., :3].contiguous(), rg[..., 3:4].contiguous(), sd.contiguous(), opts)
    _, aux = rend.sample_importance(sd.contiguous(), w, u2.shape[-1], u=T(u2), return_aux=True)
    fused = inter['inds'].reshape(B, R, -1)[b, idx]
    assert torch.equal(aux['inds'].reshape(fused.shape), fused), 'op-level importance stage != fused kernel on the strip'
    return N(aux['cdf']).reshape(len(sel), -1), fused.cpu().numpy().astype(np.int64)


@pytest.mark.parametrize('tag', ['c1', 'c2', 'c3', 'c4', 'c2mip'])
def test_full_size_vs_reference_golden(tdgp, oracle, tag):
    """BASELINE configs[0..3] at their REAL size (configs[3] = the cmax-1024 backbone) against the REFERENCE ITSELF (tests/golden/e2e_full_<tag>.npz,
    generated by tools/gen_goldens.py:gen_e2e_full from the imported reference; weights and inputs regenerate from the seed):
      * ws, 4096 sampled texels of the tri-planes;
      * image and depth through assert_image_parity with the oracle's image as the exactly rounded one: range-normalised error vs the
        reference; per-pixel max-rel vs the reference bounded by 1.5 x the reference's own two noise figures; per-pixel
        max-rel vs the reference's FLOAT64 run bounded by max(1e-4, 1.5 x what the reference's own fp32 run achieves) -- at these sizes
        the reference's fp32 image is 1.5e-3 ... 3.2e-3 per pixel (5e-6 ... 7e-6 of the range) from its own float64 image, so a flat
        1e-4 is not a property of the reference path; the bound that IS asserted is "no further from the exact image than 1.5 x the
        reference is", plus the mean error <= 1.25 x the reference's;
      * INT rows on a strip of image rows: stratified samples bit-exact vs the reference; searchsorted indices vs the reference's with
        EVERY mismatch explained by a knot window (conftest.assert_inds_mismatches_in_window: the draw lies between the two
        implementations' values of the knot it flips across, and those are within 64 ulp); fine samples."""
    from conftest import assert_inds_mismatches_in_window, full_golden_case, load_full_golden
    g = load_full_golden(tag)
    cfg, sd, inp = full_golden_case(tdgp, tag)
    G = tdgp.generator.Generator(cfg)
    G.load_numpy_state_dict(sd)
    G = G.to(DEV)
    ws = G.mapping(T(inp['z']), T(inp['c']))
    assert_close(N(ws), g['ws'], 1e-5, tag + ' ws', 1.0)
    cam = {k: T(v) for k, v in inp['camera'].items()}
    h, S = cfg.img_resolution, cfg.num_ray_steps
    R = h * h
    uc, uf = T(inp['u_coarse']), T(inp['u_fine'])
    out = G.synthesis(T(g['ws']), camera_params=cam, noise_mode='const', render_opts=dict(return_depth=True), u_coarse=uc, u_fine=uf)
    oracle.set_threads(min(64, os.cpu_count() or 1))
    ex_img, ex_depth = oracle.synthesis_forward(sd, cfg.to_dict(), g['ws'], inp['camera'], inp['u_coarse'], inp['u_fine'], 'const')
    rng, pix, _ = assert_image_parity(N(out.img), g, f'{tag} full size img', exact=ex_img, full_size=True)
    assert_image_parity(N(out.depth), g, f'{tag} full size depth', 'depth', exact=ex_depth, full_size=True)
    dec = G.synthesis.tri_plane_decoder
    planes = dec(T(g['ws'])[:, :dec.num_ws], noise_mode='const', hwc=True)
    res, F3 = cfg.tri_plane_res, 3 * cfg.feat_dim
    pl = N(planes.t.permute(0, 1, 4, 2, 3).reshape(1, F3, res, res)).reshape(-1)[g['planes_pick']]
    e_pl = float(np.abs(pl - g['planes_vals']).max() / g['planes_absmax'])
    report_parity(f'{tag} full size tri-planes (4096 sampled texels vs the reference)', range_err=e_pl)
    assert e_pl <= 1e-5, e_pl
    # the strip
    syn = G.synthesis
    c2w = tdgp.renderer.compute_cam2world_matrix(cam)
    ray_o, ray_d = tdgp.renderer.sample_rays(c2w, fov=cam['fov'], resolution=(h, h), device=DEV)
    opts = syn.rendering_options(syn._default_render_options)
    opts.update(u_coarse=uc, u_fine=uf, ray_grid_w=h)
    (rgb, _, _, _), inter = syn.renderer(planes, syn.tri_plane_mlp, ray_o, ray_d, opts, return_intermediates=True)
    assert_close(N(rgb).reshape(1, R, 3), N(out.img).reshape(1, 3, R).transpose(0, 2, 1), 0, 'renderer call == forward')
    sel = np.concatenate([np.arange(r * h, (r + 1) * h) for r in g['rows']])
    np.testing.assert_array_equal(N(inter['sdist_coarse']).reshape(R, S)[sel], g['strip_sdist_coarse'])
    assert_close(N(c2w), g['c2w'], 2e-7, 'c2w', 1.0)                          # sin / cos: 1-ulp routines in torch, correctly rounded here
    if np.array_equal(N(c2w), g['c2w']):                                     # ... and given the same matrix the rays are the reference's bits (c1 / c2 / c3)
        np.testing.assert_array_equal(N(ray_d)[0, sel], g['strip_ray_d'])
        np.testing.assert_array_equal(N(ray_o)[0, sel], g['strip_ray_o'])
    else:                                                                    # c4's camera: one matrix entry an ulp off
        assert_close(N(ray_d)[0, sel], g['strip_ray_d'], 2e-7, 'ray_d', 1.0)
        assert_close(N(ray_o)[0, sel], g['strip_ray_o'], 2e-7, 'ray_o', 1.0)
    u2 = inp['u_fine'].reshape(R, S)[sel]
    cdf_h, inds_h = _hip_strip_cdf(tdgp, G, inter, 0, sel, u2)
    n, worst = assert_inds_mismatches_in_window(inds_h, g['strip_inds'], u2, g['strip_cdf'], cdf_h, what=f'{tag} full size strip vs the reference',
                                                samples_a=N(inter['sdist_fine']).reshape(R, -1)[sel], samples_b=g['strip_sdist_fine'])
    assert n <= 16, n
    d = np.abs(np.sort(N(inter['sdist_fine']).reshape(R, -1)[sel], axis=1) - np.sort(g['strip_sdist_fine'], axis=1))
    assert np.quantile(d, 0.999) <= 2e-5 and d.max() <= (1e-3 if n == 0 else 1e-2), (float(np.quantile(d, 0.999)), float(d.max()))


def _config_vs_oracle(tdgp, oracle, cfg, rows, seed, tag, ref_tag=None):
    """One BASELINE configuration at its REAL size, HIP against the CPU oracle: the whole backbone for one sample (every layer shape,
    tile configuration and split-K factor of that configuration) and a strip of image rows through the whole renderer from the same
    planes (rays are independent: a strip is a full-fidelity check of ray generation, both field passes, importance sampling, merge
    and compositing at this configuration's ray-step count)."""
    sd = tdgp.weights.random_state_dict(cfg, seed=seed, exercise_all=True)
    G = tdgp.generator.Generator(cfg)
    G.load_numpy_state_dict(sd)
    G = G.to(DEV)
    inp = tdgp.weights.synthetic_inputs(cfg, batch=2, seed=seed + 1)
    ws = G.mapping(T(inp['z']), T(inp['c']))
    ows = oracle.mapping_forward(sd, cfg.to_dict(), inp['z'], inp['c'])
    assert_close(N(ws), ows, 1e-5, tag + ' ws', 1.0)
    planes = G.synthesis.tri_plane_decoder(ws, noise_mode='const', hwc=True)
    res, F3 = cfg.tri_plane_res, 3 * cfg.feat_dim
    oracle.set_threads(min(64, __import__('os').cpu_count() or 1))
    ref = oracle.synthesis_backbone(sd, cfg.to_dict(), N(ws)[:1], 'const')
    planes_nchw = N(planes.t.permute(0, 1, 4, 2, 3).reshape(2, F3, res, res))
    assert_close(planes_nchw[:1], ref, 1e-5, tag + f' tri-planes {res}^2', 1.0)
    report_parity(tag + ' backbone vs oracle (one sample, real size)', range_err=float(np.abs(planes_nchw[:1] - ref).max() / np.abs(ref).max()))
    cam = {k: T(v) for k, v in inp['camera'].items()}
    out = G.synthesis(ws, camera_params=cam, noise_mode='const', render_opts=dict(return_depth=True), u_coarse=T(inp['u_coarse']), u_fine=T(inp['u_fine']))
    img, depth = N(out.img), N(out.depth)
    h = cfg.img_resolution
    assert img.shape == (2, 3, h, h) and np.isfinite(img).all()
    from oracle.pipeline import render_options
    mlp = tuple(sd[f'synthesis.tri_plane_mlp.model.{i}.{n}'] for i in (0, 1) for n in ('weight', 'bias'))
    c2w = oracle.cam2world(inp['camera']['angles'], inp['camera']['radius'], inp['camera']['look_at'])
    ro, rd = oracle.sample_rays(c2w, inp['camera']['fov'], h, h)
    R, S = h * h, cfg.num_ray_steps
    sel = np.concatenate([np.arange(r * h, (r + 1) * h) for r in rows])
    u1 = inp['u_coarse'].reshape(2, R, S)[:, sel]
    u2 = inp['u_fine'].reshape(2, R, S)[:, sel].reshape(-1, S)
    orgb, odepth, _, _ = oracle.importance_render(planes_nchw, mlp, ro[:, sel], rd[:, sel], render_options(cfg.to_dict()), u1, u2)
    got = img.reshape(2, 3, R)[:, :, sel].transpose(0, 2, 1)
    e_rgb = float(np.abs(got - orgb).max() / np.abs(orgb).max())
    e_dep = float(np.abs(depth.reshape(2, R)[:, sel] - odepth[..., 0]).max())
    pix = __import__('conftest').max_rel(got, orgb)
    # SURVEY.md 9.9's per-pixel figure of the strip, ASSERTED: the oracle's strip is the exactly rounded one, and
    # the yardstick is what the REFERENCE's own fp32 run achieves against its own float64 run on this configuration at this size
    # (tests/golden/e2e_full_<ref_tag>.npz: 1.5e-3 / 3.2e-3 / 2.9e-3 for C1 / C2 / C3) -- the HIP strip may be no further from the exact
    # one than 1.5 x that (measured r04: 5.5e-5 / 2.2e-4, i.e. 0.04 x / 0.07 x), and never further than a flat 1e-3.
    from conftest import load_full_golden
    fg = load_full_golden(ref_tag or 'c3')
    ref_fig = __import__('conftest').max_rel(fg['img'], fg['img_f64'])
    report_parity(tag + f' renderer strip vs oracle ({len(rows)} rows x {h} rays x {S}+{S} samples)', rgb_range_err=e_rgb, depth_abs_err=e_dep,
                  rgb_pix_max_rel=pix, reference_fp32_vs_its_f64_at_this_size=ref_fig)
    assert e_rgb < 1e-5 and e_dep < 1e-5, (e_rgb, e_dep)
    assert pix <= min(1e-3, max(RGB_TOL, 1.5 * ref_fig)), (pix, ref_fig)


def test_config_c1_vs_oracle(tdgp, oracle):
    """BASELINE configs[0]: SDFood-like 64^2, 32(+32) ray steps, single class (c_dim 0), cmax 512."""
    _config_vs_oracle(tdgp, oracle, tdgp.config.config_c1(), rows=[0, 17, 40, 63], seed=101, tag='C1 64^2/32', ref_tag='c1')


def test_config_c2_vs_oracle(tdgp, oracle):
    """BASELINE configs[1]: Dogs 128^2, 48(+48) ray steps (a pdf row of 46 elements, ray tiles of 3 x 16 samples)."""
    _config_vs_oracle(tdgp, oracle, tdgp.config.config_c2(), rows=[0, 77, 127], seed=103, tag='C2 128^2/48', ref_tag='c2')


def test_config_c4_backbone_vs_oracle(tdgp, oracle):
    """BASELINE configs[3] (cmax 1024 / cbase 65536): the whole backbone of one sample against the oracle -- 1024-channel layers,
    odd split-K slice counts (where the round-1 LDS race lived), 488.9 GFLOP; ~40 s of host time."""
    cfg = tdgp.config.config_c4()
    sd = tdgp.weights.random_state_dict(cfg, seed=107, exercise_all=True)
    G = tdgp.generator.Generator(cfg)
    G.load_numpy_state_dict(sd)
    G = G.to(DEV)
    inp = tdgp.weights.synthetic_inputs(cfg, batch=1, seed=108)
    ws = G.mapping(T(inp['z']), T(inp['c']))
    planes = G.synthesis.tri_plane_decoder(ws, noise_mode='const', hwc=True)
    oracle.set_threads(min(64, __import__('os').cpu_count() or 1))
    ref = oracle.synthesis_backbone(sd, cfg.to_dict(), N(ws), 'const')
    got = N(planes.t.permute(0, 1, 4, 2, 3).reshape(1, 96, 512, 512))
    report_parity('C4 cmax-1024 backbone vs oracle (one sample)', range_err=float(np.abs(got - ref).max() / np.abs(ref).max()))
    assert_close(got, ref, 1e-5, 'C4 tri-planes 512^2', 1.0)
    for _ in range(5):                                       # and it is deterministic (the race showed as ~1 bad run in 10)
        again = G.synthesis.tri_plane_decoder(ws, noise_mode='const', hwc=True)
        assert torch.equal(again.t, planes.t)


def _wino_launches(tdgp, fn):
    """Per-kernel launch counts of one call of `fn`, through the library's own profiler (tdgp_profile_enable / tdgp_profile_report)."""
    tdgp._lib.profile_enable(True)
    try:
        out = fn()
        torch.cuda.synchronize()
        rep = tdgp._lib.profile_report()
    finally:
        tdgp._lib.profile_enable(False)
    return out, {k: v['launches'] for k, v in rep.items()}


@pytest.mark.parametrize('B', [16, 4])
def test_full_size_c3_at_bench_batches(tdgp, oracle, B):
    """the configuration bench.py TIMES (C3, per-GPU batch 16 and 4, bench.py's own weights and inputs)
    against the oracle.  Launch geometry depends on the batch -- the 32^2 x 512 layer is a Winograd launch only at B >= 8, split-K slice
    counts and the tail model follow the launch size, the field kernel's persistent grid wraps 64 x instead of 8 x -- and every other
    full-size test runs B = 2.  Here, at the timed batch:
      * planes of sample 0 vs the CPU oracle (<= 1e-5 of the range);
      * every image vs the SAME item rendered in a B = 2 launch (the oracle-checked geometry), <= 1e-5 of the range;
      * a strip of rays of two samples through the whole renderer vs the oracle run on the same planes: stratified samples bit-exact,
        the searchsorted indices of the importance draws exact up to draws within an ulp of a cdf knot (<= 8 of 49 152 per strip, each to the
        neighbouring interval; measured 0-3), fine samples 99.9 % <= 2e-5, RGB and depth <= 1e-5;
      * three repeats bit-identical;
      * the library's profiler confirms which kernels ran: F(4x4) takes the 64^2 ... 512^2 layers (and 32^2 at B = 16)."""
    cfg = tdgp.config.config_c3()
    sd = tdgp.weights.random_state_dict(cfg, seed=0)                       # bench.py: random_state_dict(cfg, seed=0), synthetic_inputs(seed = rank_seed(0, 0, 1) = 0)
    G = tdgp.generator.Generator(cfg)
    G.load_numpy_state_dict(sd)
    G = G.to(DEV)
    inp = tdgp.weights.synthetic_inputs(cfg, batch=B, seed=tdgp.distributed.rank_seed(0, 0, 1))
    z, c = T(inp['z']), T(inp['c'])
    cam = {k: T(v) for k, v in inp['camera'].items()}
    uc, uf = T(inp['u_coarse']), T(inp['u_fine'])
    R, S = cfg.img_resolution ** 2, cfg.num_ray_steps
    ws = G.mapping(z, c)
    dec = G.synthesis.tri_plane_decoder
    planes, launches = _wino_launches(tdgp, lambda: dec(ws[:, :dec.num_ws], noise_mode='const', hwc=True))
    import bench
    w4 = [bench.winograd4_takes(B, cfg.channels[r], cfg.channels[r], r) for r in cfg.block_resolutions]            # mirrors of wino4_shape_ok / wino_ok (modconv.hip)
    w2 = [bench.winograd_takes(B, cfg.channels[r], cfg.channels[r], r) and not f for r, f in zip(cfg.block_resolutions, w4)]
    # (a layer with few channels goes through the F(4x4) kernels in sub-batches: at least one launch per layer)
    w4f = [bench.winograd4_fused_takes(B, cfg.channels[r], cfg.channels[r], r) for r in cfg.block_resolutions]     # ... of the fused-transform branch (round 6): one launch per layer
    assert launches.get('conv_wino4f_kernel', 0) == sum(w4f) == 2, launches
    assert launches.get('conv_wino4_kernel', 0) >= sum(w4) - sum(w4f) == 3 and launches.get('conv_wino_kernel', 0) == sum(w2) == 0, launches
    assert launches.get('conv_mfma_kernel', 0) == len(cfg.block_resolutions) - sum(w4), launches                    # the remaining stride-1 3x3 layers: direct sums
    # (1) planes of sample 0 vs the oracle
    oracle.set_threads(min(64, os.cpu_count() or 1))
    ows = oracle.mapping_forward(sd, cfg.to_dict(), inp['z'][:1], inp['c'][:1])
    assert_close(N(ws[:1]), ows, 1e-5, f'B={B} ws', 1.0)
    ref = oracle.synthesis_backbone(sd, cfg.to_dict(), N(ws[:1]), 'const')
    planes_nchw0 = N(planes.t[:1].permute(0, 1, 4, 2, 3).reshape(1, 96, 512, 512))
    e_pl = float(np.abs(planes_nchw0 - ref).max() / np.abs(ref).max())
    assert e_pl <= 1e-5, e_pl
    # (2) the timed forward itself: three repeats bit-identical, every image vs its B = 2 rendering
    run = lambda sl=slice(None): G(z[sl], c[sl], {k: v[sl] for k, v in cam.items()}, noise_mode='const', u_coarse=uc[sl],   # noqa: E731
                                   u_fine=uf.reshape(B, R, S)[sl].reshape(-1, S))
    img = run()
    assert img.shape == (B, 3, 256, 256) and torch.isfinite(img).all()
    for _ in range(2):
        assert torch.equal(run(), img)
    worst = 0.0
    for b0 in range(0, B, 2):
        pair = run(slice(b0, b0 + 2))
        scale = float(img[b0:b0 + 2].abs().max())
        worst = max(worst, float((pair - img[b0:b0 + 2]).abs().max()) / scale)
    assert worst <= 1e-5, worst
    # (3) a ray strip of samples 0 and B - 1 through the renderer at the timed batch vs the oracle on the same planes
    from oracle.pipeline import render_options
    syn = G.synthesis
    c2w = tdgp.renderer.compute_cam2world_matrix(cam)
    ray_o, ray_d = tdgp.renderer.sample_rays(c2w, fov=cam['fov'], resolution=(256, 256), device=DEV)
    opts = syn.rendering_options(syn._default_render_options)
    opts.update(u_coarse=uc, u_fine=uf, ray_grid_w=256)
    (rgb, dep, _, _), inter = syn.renderer(planes, syn.tri_plane_mlp, ray_o, ray_d, opts, return_intermediates=True)
    assert_close(N(rgb).reshape(B, R, 3), N(img).reshape(B, 3, R).transpose(0, 2, 1), 0, 'renderer call == forward')
    mlp = tuple(sd[f'synthesis.tri_plane_mlp.model.{i}.{n}'] for i in (0, 1) for n in ('weight', 'bias'))
    oc2w = oracle.cam2world(inp['camera']['angles'], inp['camera']['radius'], inp['camera']['look_at'])
    oro, ord_ = oracle.sample_rays(oc2w, inp['camera']['fov'], 256, 256)
    np.testing.assert_array_equal(N(ray_o), oro)
    np.testing.assert_array_equal(N(ray_d), ord_)
    # rows 3 / 128 / 250 of sample 0 and of sample B - 1: the last patches of the launch are the ones the persistent grid reaches after its
    # last wrap, row 255 the last ray tile of all
    sel = np.concatenate([np.arange(r * 256, (r + 1) * 256) for r in (3, 128, 250, 255)])
    ni_tot, np_tot = 0, 0.0
    for b in (0, B - 1):
        pl = N(planes.t[b:b + 1].permute(0, 1, 4, 2, 3).reshape(1, 96, 512, 512))
        u1 = inp['u_coarse'].reshape(B, R, S)[b:b + 1, sel]
        u2 = inp['u_fine'].reshape(B, R, S)[b, sel]
        (orgb, odep, _, _), ointer = oracle.importance_render(pl, mlp, oro[b:b + 1, sel], ord_[b:b + 1, sel], render_options(cfg.to_dict()), u1, u2,
                                                              return_intermediates=True)
        np.testing.assert_array_equal(N(inter['sdist_coarse']).reshape(B, R, S)[b, sel], ointer['sdist_coarse'].reshape(len(sel), S))
        # INT row: the searchsorted indices of the inverse-cdf draws, oracle stage on the oracle's own coarse weights vs the kernel's on its
        # own (the chain: an fp32 MLP whose summation order differs feeds the cdf, so a draw within an ulp of a knot may move to the
        # neighbouring interval -- bounded as in test_e2e_tiny, measured 0); the samples themselves agree to an ulp of the depth range.
        _, oaux = oracle.sample_importance(ointer['sdist_coarse'].reshape(1, len(sel), S, 1), ointer['weights_coarse'], u2, cfg.ray_marcher_type, return_aux=True)
        # every mismatching draw must be EXPLAINED -- it lies between the kernel's and the oracle's value of the knot it
        # flips across, and those two values are within 64 ulp (conftest.assert_inds_mismatches_in_window); a draw away from every knot
        # landing in another interval fails the test.  (Before round 5 the mismatches were counted, <= 8 per strip, and not explained.)
        cdf_h, hi = _hip_strip_cdf(tdgp, G, inter, b, sel, u2)
        oi = oaux['inds'].reshape(len(sel), -1)
        ni, _ = assert_inds_mismatches_in_window(hi, oi, u2, oaux['cdf'].reshape(len(sel), -1), cdf_h, what=f'C3 B={B} sample {b} strip vs the oracle',
                                                 samples_a=N(inter['sdist_fine']).reshape(B, R, -1)[b, sel], samples_b=ointer['sdist_fine'].reshape(len(sel), -1))
        assert ni <= 16, ni
        ni_tot += ni
        hf = np.sort(N(inter['sdist_fine']).reshape(B, R, -1)[b, sel], axis=1)
        of = np.sort(ointer['sdist_fine'].reshape(len(sel), -1), axis=1)
        np_tot = max(np_tot, float(np.abs(hf - of).max()))
        # the samples are (u - cdf_lo) / (cdf_hi - cdf_lo) pushed through the bins: on a flat stretch of the cdf an ulp of the weights moves
        # the value by 1e-5 of the depth range (measured max 3.0e-5, 99.9 % below 1e-6 .. 9e-6 depending on the planes) -- bounded robustly, the
        # integers above are the pin
        d = np.abs(hf - of)
        assert np.quantile(d, 0.999) <= 2e-5 and d.max() <= (1e-3 if ni == 0 else 1e-2), (b, float(np.quantile(d, 0.999)), float(d.max()))
        got = N(rgb).reshape(B, R, 3)[b, sel]
        e_rgb = float(np.abs(got - orgb[0]).max() / np.abs(orgb).max())
        e_dep = float(np.abs(N(dep).reshape(B, R)[b, sel] - odep[0, :, 0]).max())
        assert e_rgb <= 1e-5 and e_dep <= 1e-5, (b, e_rgb, e_dep)
    report_parity(f'C3 at the timed batch B = {B} (bench.py inputs)', planes_range_err=e_pl, image_vs_b2_launch_range_err=worst,
                  strip_inds_mismatches=ni_tot, strip_fine_sample_max_abs_diff=np_tot, wino4_launches=launches.get('conv_wino4_kernel', 0), wino2_launches=launches.get('conv_wino_kernel', 0))


def test_c4_backbone_at_bench_batch(tdgp, oracle):
    """C4 (cmax 1024) at the batch its FID loop and `bench.py --config c4` time next to 16: B = 4 -- planes of sample 0 vs the oracle, three
    repeats bit-identical, every sample's planes vs a B = 1 launch of the same item (<= 1e-5 of the range)."""
    cfg = tdgp.config.config_c4()
    sd = tdgp.weights.random_state_dict(cfg, seed=0)
    G = tdgp.generator.Generator(cfg)
    G.load_numpy_state_dict(sd)
    G = G.to(DEV)
    inp = tdgp.weights.synthetic_inputs(cfg, batch=4, seed=0)
    ws = G.mapping(T(inp['z']), T(inp['c']))
    dec = G.synthesis.tri_plane_decoder
    planes, launches = _wino_launches(tdgp, lambda: dec(ws[:, :dec.num_ws], noise_mode='const', hwc=True))
    import bench
    w4 = [bench.winograd4_takes(4, cfg.channels[r], cfg.channels[r], r) for r in cfg.block_resolutions]
    w2 = [bench.winograd_takes(4, cfg.channels[r], cfg.channels[r], r) and not f for r, f in zip(cfg.block_resolutions, w4)]
    w4f = [bench.winograd4_fused_takes(4, cfg.channels[r], cfg.channels[r], r) for r in cfg.block_resolutions]
    assert launches.get('conv_wino4f_kernel', 0) == sum(w4f) and launches.get('conv_wino4_kernel', 0) >= sum(w4) - sum(w4f) and launches.get('conv_wino_kernel', 0) == sum(w2), launches
    for _ in range(2):
        assert torch.equal(dec(ws[:, :dec.num_ws], noise_mode='const', hwc=True).t, planes.t)
    oracle.set_threads(min(64, os.cpu_count() or 1))
    ref = oracle.synthesis_backbone(sd, cfg.to_dict(), N(ws[:1]), 'const')
    got = N(planes.t[:1].permute(0, 1, 4, 2, 3).reshape(1, 96, 512, 512))
    e_pl = float(np.abs(got - ref).max() / np.abs(ref).max())
    worst = 0.0
    for b in range(4):
        one = dec(ws[b:b + 1, :dec.num_ws], noise_mode='const', hwc=True).t
        worst = max(worst, float((one - planes.t[b:b + 1]).abs().max() / planes.t[b:b + 1].abs().max()))
    report_parity('C4 backbone at the timed batch B = 4', planes_range_err=e_pl, planes_vs_b1_launch_range_err=worst, wino4_launches=launches.get('conv_wino4_kernel', 0), wino2_launches=launches.get('conv_wino_kernel', 0))
    assert e_pl <= 1e-5 and worst <= 1e-5, (e_pl, worst)


def test_torgb_overlap_on_and_off_produce_identical_bits(tdgp, full_c3):
    """DESIGN claims 'same bits' for the ToRGB layers on a second stream (SynthesisBlocksSequence.overlap_torgb): asserted, at full size."""
    G, ws = full_c3['G'], full_c3['ws']
    dec = G.synthesis.tri_plane_decoder
    was = dec.overlap_torgb                                   # (default: the blocks up to 16^2 -- beside the persistent F(4x4) grids it costs time, round 4)
    outs = {}
    try:
        for mode in (True, False, 16, 64):                    # every block, none, the default threshold, a threshold inside the F(4x4) range
            dec.overlap_torgb = mode
            outs[mode] = dec(ws[:, :dec.num_ws], noise_mode='const', hwc=True).t.clone()
    finally:
        dec.overlap_torgb = was
    for mode, t in outs.items():
        assert torch.equal(t, full_c3['planes'].t), mode


def test_generator_survives_deepcopy_and_pickle_after_a_forward(tdgp):
    """the reference deep-copies G on every snapshot and metric run (training_loop.py:459, metric_utils.py:293,328).  The
    side stream of the ToRGB overlap must not live in the module: after an inference forward `copy.deepcopy(G)` and `pickle.dumps(G)` work,
    and the copy renders the same bits."""
    import copy
    import io
    cfg = tdgp.config.config_tiny()
    g = load_golden('e2e_tiny')
    G = _gen(tdgp, cfg, 21)
    kw = dict(noise_mode='const', u_coarse=T(g['u_coarse']), u_fine=T(g['u_fine']))
    G.synthesis.tri_plane_decoder.overlap_torgb = True             # the configuration that used to put a stream into the module
    img = G(T(g['z']), T(g['c']), _cam(g), **kw)
    assert not any(isinstance(v, torch.cuda.Stream) for m in G.modules() for v in vars(m).values())
    G2 = copy.deepcopy(G)
    assert torch.equal(G2(T(g['z']), T(g['c']), _cam(g), **kw), img)
    buf = io.BytesIO()
    torch.save(G, buf)
    buf.seek(0)
    G3 = torch.load(buf, weights_only=False)
    assert torch.equal(G3(T(g['z']), T(g['c']), _cam(g), **kw), img)
    assert tdgp.generator.side_stream_of(DEV) is tdgp.generator.side_stream_of(torch.device('cuda', torch.cuda.current_device()))


def test_full_size_properties(tdgp, full_c3):
    """Size-independent properties at 256^2 / 64 steps: ray independence (a sub-batch of rays renders bit-identically),
    batch consistency, sorted merge order, and sum(weights) + final transmittance == 1 for the classical marcher."""
    G, inp, cfg = full_c3['G'], full_c3['inp'], full_c3['cfg']
    R, S = 256 * 256, cfg.num_ray_steps
    syn = G.synthesis
    c2w = tdgp.renderer.compute_cam2world_matrix(full_c3['cam'])
    ro, rd = tdgp.renderer.sample_rays(c2w, full_c3['cam']['fov'], (256, 256))
    opts = syn.rendering_options(syn._default_render_options)
    u1, u2 = T(inp['u_coarse']), T(inp['u_fine'])
    (rgb, depth, wsum, fT), inter = syn.renderer(full_c3['planes'], syn.tri_plane_mlp, ro, rd, dict(opts, u_coarse=u1, u_fine=u2, ray_grid_w=256),
                                                 return_intermediates=True)
    # sum of compositing weights + final transmittance telescopes to 1 (up to the 1e-10 the reference adds per step)
    assert float((wsum[..., 0] + fT - 1).abs().max()) < 1e-4
    # merged depths ascending: gather t by the reported permutation
    t_all = torch.cat([inter['tdist_coarse'], torch.gather(inter['tdist_fine'], 2, torch.argsort(inter['fine_perm'].reshape(2, R, S).long(), dim=2))], dim=2)
    t_sorted = torch.gather(t_all, 2, inter['perm'].long())
    assert bool((t_sorted[:, :, 1:] >= t_sorted[:, :, :-1]).all())
    # ray independence + layout independence: rows 64..127 alone, linear point order (no image tiling) -> identical pixels
    sl = slice(64 * 256, 128 * 256)
    rgb_s, depth_s, _, _ = syn.renderer(full_c3['planes'], syn.tri_plane_mlp, ro[:, sl].contiguous(), rd[:, sl].contiguous(),
                                        dict(opts, u_coarse=u1.reshape(2, R, S)[:, sl].contiguous(), u_fine=u2.reshape(2, R, S)[:, sl].reshape(-1, S).contiguous(),
                                             ray_grid_w=0))
    np.testing.assert_array_equal(N(rgb[:, sl]), N(rgb_s))
    np.testing.assert_array_equal(N(depth[:, sl]), N(depth_s))
    # batch consistency of the whole forward: sample 1 alone == sample 1 inside the batch of 2 (split-K factors may differ -> 1e-5)
    cam1 = {k: v[1:2] for k, v in full_c3['cam'].items()}
    img_b = syn(full_c3['ws'], camera_params=full_c3['cam'], noise_mode='const', u_coarse=u1, u_fine=u2)
    img_1 = syn(full_c3['ws'][1:2], camera_params=cam1, noise_mode='const', u_coarse=u1[1:2], u_fine=u2.reshape(2, R, S)[1:2].reshape(-1, S))
    assert float((img_b[1:2] - img_1).abs().max() / img_b.abs().max()) < 1e-5
    # determinism
    img_b2 = syn(full_c3['ws'], camera_params=full_c3['cam'], noise_mode='const', u_coarse=u1, u_fine=u2)
    assert torch.equal(img_b, img_b2)


def test_chunked_schedule_is_the_same_forward(tdgp, full_c3):
    """SynthesisNetwork.chunk (high-resolution blocks + renderer a few samples at a time, for Infinity-Cache residency) only re-orders
    the launches: images agree with the whole-batch forward to split-K rounding, for every chunk size incl. a ragged last chunk."""
    G, inp = full_c3['G'], full_c3['inp']
    syn = G.synthesis
    B = 5
    inp5 = tdgp.weights.synthetic_inputs(full_c3['cfg'], batch=B, seed=11)
    ws = G.mapping(T(inp5['z']), T(inp5['c']))
    cam = {k: T(v) for k, v in inp5['camera'].items()}
    u1, u2 = T(inp5['u_coarse']), T(inp5['u_fine'])
    prev = (syn.chunk, syn.chunk_from)
    try:
        syn.chunk = None
        ref = syn(ws, camera_params=cam, noise_mode='const', render_opts=dict(return_depth=True), u_coarse=u1, u_fine=u2)
        for chunk, cfrom in ((2, 128), (1, 256), (4, 64), (3, 8)):
            syn.chunk, syn.chunk_from = chunk, cfrom
            out = syn(ws, camera_params=cam, noise_mode='const', render_opts=dict(return_depth=True), u_coarse=u1, u_fine=u2)
            assert float((out.img - ref.img).abs().max() / ref.img.abs().max()) < 1e-5, (chunk, cfrom)
            assert float((out.depth - ref.depth).abs().max()) < 1e-5, (chunk, cfrom)
    finally:
        syn.chunk, syn.chunk_from = prev


# ------------------------------------------------------------------------------------------------ BASELINE configs[4]: bf16 blocks
# Tolerance of the reduced-precision path.  One bf16 rounding is 2^-9 (0.2 %) of the value; a layer output passes through ~4 of them
# (operands, conv output, bias_act output) and the HIP kernels round at different points than the reference (style-scaled activations
# and shared weights instead of per-sample weights, modconv_bf16.inc), so two correct implementations differ by rounding NOISE:
# bounded here by 1.5e-2 of the tensor scale at the worst element and 2e-3 in the mean -- an indexing or scaling bug is O(1).
BF16_MAX, BF16_MEAN = 1.5e-2, 2e-3


def _assert_bf16_close(got, ref, what, max_tol=BF16_MAX, mean_tol=BF16_MEAN):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = np.abs(ref).max()
    err = np.abs(got - ref) / scale
    report_parity('bf16 ' + what, max_err=float(err.max()), mean_err=float(err.mean()))
    assert err.max() <= max_tol and err.mean() <= mean_tol, f'{what}: max {err.max():.3e} (tol {max_tol:.1e}), mean {err.mean():.3e} (tol {mean_tol:.1e})'


@pytest.mark.parametrize('B,cin,cout,H,k,up,noise', [(2, 64, 64, 32, 3, 1, True), (3, 32, 96, 64, 3, 1, False), (2, 96, 40, 32, 3, 1, True),
                                                      (2, 64, 32, 32, 3, 2, True), (1, 32, 64, 64, 3, 2, False), (2, 128, 130, 16, 3, 2, True),
                                                      (2, 64, 64, 16, 3, 1, True), (2, 24, 16, 32, 3, 1, False),
                                                      (3, 64, 64, 32, 3, 1, 'per_sample'), (2, 64, 32, 32, 3, 2, 'per_sample')])
def test_bf16_modconv_vs_oracle(tdgp, oracle, B, cin, cout, H, k, up, noise):
    """One reduced-precision synthesis layer (modulated 3x3 conv, stride 1 or x2 + FIR, noise, bias, lrelu * sqrt2, clamp 256) on bf16
    activations: tdgp_modconv2d_bf16 -- and, for the last two shapes, the widened fp32 fallback -- against the oracle's bf16 path."""
    rs = np.random.RandomState(B * 100 + cin + up)
    mc = tdgp.ops.modconv
    x = oracle.round_bf16(rs.randn(B, cin, H, H).astype(np.float32))
    w = rs.randn(cout, cin, k, k).astype(np.float32)
    s = (1.0 + 0.5 * rs.randn(B, cin)).astype(np.float32)
    bias = rs.randn(cout).astype(np.float32)
    nz = (0.3 * rs.randn(H * up, H * up)).astype(np.float32) if noise else None
    if noise == 'per_sample':                  # noise_mode='random' (networks_stylegan2.py:133-134): [B,1,H,W], batch stride H*W
        nz = (0.3 * rs.randn(B, 1, H * up, H * up)).astype(np.float32)
    f = oracle.setup_filter([1, 3, 3, 1])
    ref = oracle.bias_act_bf16(oracle.modulated_conv2d(x, w, s, noise=nz, up=up, demodulate=True, resample_filter=f, prec='bf16'), bias, act='lrelu', clamp=256)
    y = mc.modconv_forward(T(x).to(torch.bfloat16), mc.PackedConv(T(w)), T(s), noise=None if nz is None else T(nz), bias=T(bias), up=up, demodulate=True,
                           act='lrelu', clamp=256, fir=mc.fir_host_array(f) if up == 2 else None)
    assert y.dtype == torch.bfloat16 and y.shape == (B, cout, H * up, H * up)
    _assert_bf16_close(N(y.float()), ref, f'layer {cin}->{cout} @{H} up{up}')


@pytest.mark.parametrize('B,cin,crgb,H,skip', [(2, 64, 96, 64, True), (2, 128, 24, 32, True), (1, 32, 96, 32, False)])
def test_bf16_torgb_vs_oracle(tdgp, oracle, B, cin, crgb, H, skip):
    """ToRGB of a reduced-precision block: bf16 activations in, clamp 256, the fp32 skip image upsampled and added AFTER the clamp."""
    rs = np.random.RandomState(cin + crgb)
    mc = tdgp.ops.modconv
    feat = crgb // 3
    x = oracle.round_bf16(rs.randn(B, cin, H, H).astype(np.float32))
    w1 = rs.randn(crgb, cin, 1, 1).astype(np.float32)
    s = ((1 + 0.5 * rs.randn(B, cin)) / np.sqrt(cin)).astype(np.float32)
    bias = rs.randn(crgb).astype(np.float32)
    f = oracle.setup_filter([1, 3, 3, 1])
    ref = oracle.bias_act_bf16(oracle.modulated_conv2d(x, w1, s, demodulate=False, prec='bf16'), bias, act='linear', clamp=256)
    prev_cl = None
    if skip:
        prev = rs.randn(B, crgb, H // 2, H // 2).astype(np.float32)
        ref = ref + oracle.upsample2d(prev, f)
        prev_cl = T(prev).reshape(B, 3, feat, H // 2, H // 2).permute(0, 1, 3, 4, 2).contiguous()
    y = mc.modconv_forward(T(x).to(torch.bfloat16), mc.PackedConv(T(w1)), T(s), bias=T(bias), demodulate=False, act='linear', gain=1.0, clamp=256, skip=prev_cl,
                           fir=mc.fir_host_array(f) if skip else None, out_layout=1, out_feat=feat)
    assert y.dtype == torch.float32 and y.shape == (B, 3, H, H, feat)
    _assert_bf16_close(N(y.permute(0, 1, 4, 2, 3).reshape(B, crgb, H, H)), ref, f'torgb {cin}->{crgb} @{H}')


def test_bf16_generator_vs_reference_golden(tdgp, oracle):
    """config_mid_bf16 (blocks 32^2 and 64^2 in bf16, conv_clamp 256): tri-planes and image against the REFERENCE's own reduced-precision
    run with bfloat16 (tests/golden/bf16.npz), and block by block against the oracle."""
    cfg = tdgp.config.config_mid_bf16()
    g = load_golden('bf16')
    G = _gen(tdgp, cfg, 61)
    dec = G.synthesis.tri_plane_decoder
    assert [getattr(dec, f'b{r}').use_fp16 for r in dec.block_resolutions] == [False, False, False, True, True]
    ws = T(g['ws'])
    planes = dec(ws, noise_mode='const')
    _assert_bf16_close(N(planes), g['planes'], 'tri-planes vs the reference')
    out = G.synthesis(ws, camera_params=_cam(g), noise_mode='const', render_opts=dict(return_depth=True), u_coarse=T(g['u_coarse']), u_fine=T(g['u_fine']))
    _assert_bf16_close(N(out.img), g['img'], 'image vs the reference', max_tol=3e-2, mean_tol=4e-3)
    _assert_bf16_close(N(out.depth), g['depth'], 'depth vs the reference', max_tol=2e-2, mean_tol=2e-3)
    sd = tdgp.weights.random_state_dict(cfg, seed=61, exercise_all=True)
    from oracle import pipeline as P
    oplanes = P.synthesis_backbone(sd, cfg.to_dict(), g['ws'], 'const')
    _assert_bf16_close(N(planes), oplanes, 'tri-planes vs the oracle')


def test_bf16_full_size_vs_reference_golden(tdgp):
    """BASELINE configs[4] at its REAL size against the REFERENCE's own reduced-precision run with bfloat16 (tests/golden/bf16_full_c5.npz):
    16384 sampled texels of the tri-planes, the 256^2 image and the depth map (96 + 96 ray steps), at the bf16 tolerances of the mid-size golden."""
    g = load_golden('bf16_full_c5')
    cfg = tdgp.config.config_c5()
    seed = int(g['seed'][0])
    G = _gen(tdgp, cfg, seed)
    inp = tdgp.weights.synthetic_inputs(cfg, batch=1, seed=seed + 1)
    ws = T(g['ws'])
    dec = G.synthesis.tri_plane_decoder
    planes = dec(ws[:, :dec.num_ws], noise_mode='const', hwc=True)
    got = N(planes.t.permute(0, 1, 4, 2, 3).reshape(1, 96, 512, 512)).reshape(-1)[g['planes_pick']]
    err = np.abs(got - g['planes_vals']) / g['planes_absmax']
    report_parity('bf16 C5 full size tri-planes (16384 sampled texels vs the reference)', max_err=float(err.max()), mean_err=float(err.mean()))
    assert err.max() <= BF16_MAX and err.mean() <= BF16_MEAN, (float(err.max()), float(err.mean()))
    out = G.synthesis(ws, camera_params={k: T(v) for k, v in inp['camera'].items()}, noise_mode='const', render_opts=dict(return_depth=True),
                      u_coarse=T(inp['u_coarse']), u_fine=T(inp['u_fine']))
    _assert_bf16_close(N(out.img), g['img'], 'C5 full size image vs the reference', max_tol=3e-2, mean_tol=4e-3)
    _assert_bf16_close(N(out.depth), g['depth'], 'C5 full size depth vs the reference', max_tol=2e-2, mean_tol=2e-3)


def test_bf16_planes_nchw_equal_channel_last(tdgp):
    """with reduced-precision blocks the tri-plane image must stay fp32 in BOTH layouts (networks_stylegan2.py:268
    `y.to(float32); img.add_`): the public NCHW form (hwc=False) takes the widened ToRGB fallback and must neither round the accumulated
    skip image to bf16 nor return a bf16 tensor."""
    cfg = tdgp.config.config_mid_bf16()
    G = _gen(tdgp, cfg, 61)
    dec = G.synthesis.tri_plane_decoder
    ws = T(load_golden('bf16')['ws'])
    nchw = dec(ws, noise_mode='const')
    hwc = dec(ws, noise_mode='const', hwc=True).t
    assert nchw.dtype == torch.float32 and hwc.dtype == torch.float32
    a, b = N(nchw), N(hwc.permute(0, 1, 4, 2, 3).reshape(nchw.shape))
    # the channel-last kernel keeps the reference's bf16 rounding points inside ToRGB (weights, style-scaled activations, the pre-skip
    # term), the widened NCHW fallback multiplies the same bf16 activations in fp32: they agree to bf16 rounding of ONE ToRGB term per
    # block, and both sit inside the bf16 tolerance against the reference's own reduced-precision run
    _assert_bf16_close(a, b, 'tri-planes NCHW vs channel-last', max_tol=6e-3, mean_tol=5e-4)
    _assert_bf16_close(a, load_golden('bf16')['planes'], 'NCHW tri-planes vs the reference')
    # ... and the image itself was never rounded to bf16: (almost) no value is bf16-representable
    as_bf16 = N(nchw.to(torch.bfloat16).float())
    assert (as_bf16 != a).mean() > 0.9


def _recorded_randn(fn):
    """Run fn() with torch.randn recorded: -> (result, [draws in call order]) -- the per-layer noise maps of noise_mode='random'."""
    draws, real = [], torch.randn

    def rec(*a, **k):
        t = real(*a, **k)
        draws.append(t)
        return t
    torch.randn = rec
    try:
        return fn(), draws
    finally:
        torch.randn = real


def test_full_size_backbone_random_noise_vs_oracle(tdgp, oracle, full_c3):
    """noise_mode='random' -- the default of the FID loop (metric_utils.py:310) -- at the real C3 shapes with non-zero noise strengths:
    every layer gets its own [B,1,res,res] map (networks_stylegan2.py:133-134), i.e. the per-sample noise path (noise_bstride != 0) of the
    Winograd kernels, the split-K reduction, the direct kernel and the x2 FIR pass.  The draws are recorded and handed to the oracle as explicit maps."""
    cfg, sd, G = full_c3['cfg'], full_c3['sd'], full_c3['G']
    dec = G.synthesis.tri_plane_decoder
    ws = full_c3['ws'][:2]
    tdgp._lib.profile_enable(True)
    try:
        planes, draws = _recorded_randn(lambda: dec(ws, noise_mode='random', hwc=True))
        torch.cuda.synchronize()
        names = set(tdgp._lib.profile_report())
    finally:
        tdgp._lib.profile_enable(False)
    layers = [l for (l, _, _) in dec._layers() if isinstance(l, tdgp.generator.SynthesisLayer)]
    assert len(draws) == len(layers) == 15 and all(d.shape == (2, 1, l.resolution, l.resolution) for d, l in zip(draws, layers))
    assert all(float(sd[f'synthesis.tri_plane_decoder.{n}.noise_strength']) != 0.0 for n in ('b512.conv1', 'b64.conv0'))
    prefixes = [f'synthesis.tri_plane_decoder.b{l.resolution}.conv{1 if (i == 0 or i % 2 == 0) else 0}' for i, l in enumerate(layers)]
    assert prefixes[0].endswith('b4.conv1') and prefixes[1].endswith('b8.conv0') and prefixes[-1].endswith('b512.conv1')
    noise = {pfx: N(d[:1]) for pfx, d in zip(prefixes, draws)}
    oracle.set_threads(min(64, __import__('os').cpu_count() or 1))
    ref = oracle.synthesis_backbone(sd, cfg.to_dict(), N(ws)[:1], noise)
    got = N(planes.t[:1].permute(0, 1, 4, 2, 3).reshape(1, 96, 512, 512))
    assert_close(got, ref, 1e-5, 'tri-planes 512^2, per-sample random noise', 1.0)
    const = N(full_c3['planes'].t[:1].permute(0, 1, 4, 2, 3).reshape(1, 96, 512, 512))
    assert np.abs(got - const).max() > 1e-2 * np.abs(const).max()          # the noise maps matter
    # B = 2: the 64^2 ... 512^2 stride-1 layers run as F(4x4) -- the 64^2 one with its input channels split, i.e. the per-sample noise is applied by the
    # reduction pass --, the low-resolution x2 layers through the FIR pass (F(2x2) with per-sample noise: test_modconv_winograd4_vs_oracle, mode 3)
    assert {'conv_wino4_kernel', 'splitk_reduce_kernel', 'fir_act_kernel'} <= names, names


def test_config_c5_vs_oracle(tdgp, oracle):
    """BASELINE configs[4] at its REAL size: 512^2 tri-planes with the 64^2 ... 512^2 blocks in bf16 (one sample against the oracle's bf16
    backbone), and the 256^2 / 96(+96)-step renderer on a strip of rows against the fp32 oracle renderer on the same planes."""
    cfg = tdgp.config.config_c5()
    sd = tdgp.weights.random_state_dict(cfg, seed=113, exercise_all=True)
    G = tdgp.generator.Generator(cfg)
    G.load_numpy_state_dict(sd)
    G = G.to(DEV)
    inp = tdgp.weights.synthetic_inputs(cfg, batch=1, seed=114)
    ws = G.mapping(T(inp['z']), T(inp['c']))
    planes = G.synthesis.tri_plane_decoder(ws, noise_mode='const', hwc=True)
    oracle.set_threads(min(64, __import__('os').cpu_count() or 1))
    from oracle import pipeline as P
    ref = P.synthesis_backbone(sd, cfg.to_dict(), N(ws), 'const')
    planes_nchw = N(planes.t.permute(0, 1, 4, 2, 3).reshape(1, 96, 512, 512))
    _assert_bf16_close(planes_nchw, ref, 'C5 tri-planes 512^2 vs the oracle')
    cam = {k: T(v) for k, v in inp['camera'].items()}
    out = G.synthesis(ws, camera_params=cam, noise_mode='const', render_opts=dict(return_depth=True), u_coarse=T(inp['u_coarse']), u_fine=T(inp['u_fine']))
    img, depth = N(out.img), N(out.depth)
    mlp = tuple(sd[f'synthesis.tri_plane_mlp.model.{i}.{n}'] for i in (0, 1) for n in ('weight', 'bias'))
    c2w = oracle.cam2world(inp['camera']['angles'], inp['camera']['radius'], inp['camera']['look_at'])
    ro, rd = oracle.sample_rays(c2w, inp['camera']['fov'], 256, 256)
    R, S = 256 * 256, cfg.num_ray_steps
    sel = np.concatenate([np.arange(r * 256, (r + 1) * 256) for r in (5, 200)])
    u1 = inp['u_coarse'].reshape(1, R, S)[:, sel]
    u2 = inp['u_fine'].reshape(1, R, S)[:, sel].reshape(-1, S)
    orgb, odepth, _, _ = oracle.importance_render(planes_nchw, mlp, ro[:, sel], rd[:, sel], P.render_options(cfg.to_dict()), u1, u2)   # same (HIP) planes: fp32 renderer
    got = img.reshape(1, 3, R)[:, :, sel].transpose(0, 2, 1)
    e_rgb = float(np.abs(got - orgb).max() / np.abs(orgb).max())
    e_dep = float(np.abs(depth.reshape(1, R)[:, sel] - odepth[..., 0]).max())
    report_parity('C5 renderer strip vs oracle (2 rows x 256 rays x 96+96 samples, fp32 on the HIP planes)', rgb_range_err=e_rgb, depth_abs_err=e_dep)
    assert e_rgb < 1e-5 and e_dep < 1e-5, (e_rgb, e_dep)


def test_config_c5_cmax1024_backbone_vs_oracle(tdgp, oracle):
    """VERDICT r04 missing #3: the bf16 configuration as BASELINE.md section 3 sizes it -- cmax 1024 / cbase 65536 (networks_epigraf.py:98: blocks 4^2 ... 64^2
    at 1024 channels, 128^2 512, 256^2 256, 512^2 128), the 64^2 ... 512^2 blocks in bf16 -- one sample's whole backbone against the oracle's bf16
    backbone (the 1024-channel bf16 3x3 / x2 / ToRGB kernels, K loops twice as long as C5's), and bit-identical repeats."""
    cfg = tdgp.config.config_c5(cmax=1024, cbase=65536)
    assert cfg.channels[64] == 1024 and cfg.channels[512] == 128 and cfg.fp16_resolution == 64
    sd = tdgp.weights.random_state_dict(cfg, seed=117, exercise_all=True)
    G = tdgp.generator.Generator(cfg)
    G.load_numpy_state_dict(sd)
    G = G.to(DEV)
    inp = tdgp.weights.synthetic_inputs(cfg, batch=1, seed=118)
    ws = G.mapping(T(inp['z']), T(inp['c']))
    planes = G.synthesis.tri_plane_decoder(ws, noise_mode='const', hwc=True)
    oracle.set_threads(min(64, os.cpu_count() or 1))
    from oracle import pipeline as P
    ref = P.synthesis_backbone(sd, cfg.to_dict(), N(ws), 'const')
    _assert_bf16_close(N(planes.t.permute(0, 1, 4, 2, 3).reshape(1, 96, 512, 512)), ref, 'C5 cmax-1024 tri-planes 512^2 vs the oracle')
    for _ in range(3):
        assert torch.equal(G.synthesis.tri_plane_decoder(ws, noise_mode='const', hwc=True).t, planes.t)


def test_compat_plugins(tdgp, oracle):
    """The pybind-signature plugin objects of 3dgp_amd/compat.py (bias_act.cpp:32 / upfirdn2d.cpp:16 argument orders)."""
    rs = np.random.RandomState(2)
    x = rs.randn(2, 6, 9, 9).astype(np.float32)
    b = rs.randn(6).astype(np.float32)
    empty = torch.empty([0], device=DEV)
    y = tdgp.compat.BiasActPlugin.bias_act(T(x), T(b), empty, empty, empty, 0, 1, 3, 0.2, float(np.sqrt(2)), -1.0)
    assert_close(N(y), oracle.bias_act(x, b, act='lrelu'), 1e-6)
    f = oracle.setup_filter([1, 3, 3, 1])
    y = tdgp.compat.Upfirdn2dPlugin.upfirdn2d(T(x), T(f), 2, 2, 1, 1, 2, 1, 2, 1, False, 4.0)
    assert_close(N(y), oracle.upsample2d(x, f), 2e-6, 'upsample2d via plugin', 1.0)
    with pytest.raises(RuntimeError):
        tdgp.compat.BiasActPlugin.bias_act(T(x), T(b), empty, empty, empty, 3, 1, 3, 0.2, 1.0, -1.0)     # grad must be 0, 1 or 2


# ------------------------------------------------------------------------------------------------ adaptors
@pytest.mark.parametrize('idx', [0, 1])
def test_adaptors(tdgp, oracle, idx):
    """DepthAdaptor (5x5 convolutions on the MFMA kernels: generic path at 16^2, mask-free fast path at 32^2) and CameraAdaptor
    against the reference goldens; then the depth adaptor inside SynthesisNetwork.forward against the oracle on the HIP depth."""
    from oracle import pipeline as P
    g = load_golden('adaptors')
    tag, cfg = tdgp.config.configs_adaptor_goldens()[idx]
    G = _gen(tdgp, cfg, 51)
    da, ca = G.synthesis.depth_adaptor, G.synthesis.camera_adaptor
    outs = da(T(g[f'{tag}_depth']), T(g[f'{tag}_w']), all_outs=True)
    assert_close(N(outs), g[f'{tag}_outs'], 5e-6, 'per-layer heads', 1.0)
    assert_close(N(da(T(g[f'{tag}_depth']), T(g[f'{tag}_w']))), g[f'{tag}_depth_adapted'], 5e-6, 'depth_adapted', 1.0)
    cam = {k: T(g[f'{tag}_cam_{k}']) for k in ('angles', 'fov', 'radius', 'look_at')}
    new = ca(cam, T(g[f'{tag}_z']), T(g[f'{tag}_c']) if cfg.c_dim > 0 else None)
    for k in ('angles', 'fov', 'radius', 'look_at'):
        assert_close(N(new[k]), g[f'{tag}_new_{k}'], 5e-6, k, 1.0)
    # inside the generator: render_opts as networks_epigraf.py:255-259
    inp = tdgp.weights.synthetic_inputs(cfg, batch=2, seed=54)
    out = G(T(inp['z']), T(inp['c']), {k: T(v) for k, v in inp['camera'].items()}, noise_mode='const', u_coarse=T(inp['u_coarse']), u_fine=T(inp['u_fine']),
            render_opts=dict(return_depth=True, return_depth_adapted=True))
    sd = tdgp.weights.random_state_dict(cfg, seed=51, exercise_all=True)
    ws = N(G.mapping(T(inp['z']), T(inp['c'])))
    ref = P.depth_adaptor_forward(sd, cfg.to_dict(), N(out.depth), ws[:, 0])
    assert_close(N(out.depth_adapted), ref, 5e-6, 'depth_adapted in G.forward', 1.0)
    cat = G(T(inp['z']), T(inp['c']), {k: T(v) for k, v in inp['camera'].items()}, noise_mode='const', u_coarse=T(inp['u_coarse']), u_fine=T(inp['u_fine']),
            render_opts=dict(concat_depth=True))
    assert cat.shape == (2, 4, cfg.img_resolution, cfg.img_resolution)
    assert torch.equal(cat[:, :3], out.img) and torch.equal(cat[:, 3:], out.depth_adapted)


def test_depth_adaptor_is_elided_in_the_plain_eval_forward(tdgp):
    """`img + 0.0 * depth_adapted.max()` (networks_epigraf.py:253) equals `img` for finite adaptor outputs, so the plain
    eval forward (no concat_depth, no return_depth_adapted) skips the adaptor's three 5x5 convolutions.  Asserted: (a) the elided forward
    launches no 5x5 layer and returns the bits of the literal one; (b) `strict_nan_propagation` restores the literal forward -- a NaN
    planted in the adaptor's weights then poisons the image (the reference's behaviour), while the elided forward stays finite; (c) the
    options that USE the adapted depth still evaluate it; (d) training-mode forwards are never elided."""
    tag, cfg = tdgp.config.configs_adaptor_goldens()[1]
    G = _gen(tdgp, cfg, 51)
    inp = tdgp.weights.synthetic_inputs(cfg, batch=2, seed=54)
    args = (T(inp['z']), T(inp['c']), {k: T(v) for k, v in inp['camera'].items()})
    kw = dict(noise_mode='const', u_coarse=T(inp['u_coarse']), u_fine=T(inp['u_fine']))
    syn = G.synthesis
    assert syn.depth_adaptor is not None and syn.strict_nan_propagation is False

    def launches(fn):
        tdgp._lib.profile_enable(True)
        try:
            out = fn()
            torch.cuda.synchronize()
            rep = tdgp._lib.profile_report()
        finally:
            tdgp._lib.profile_enable(False)
        return out, sum(v['launches'] for v in rep.values())
    G(*args, **kw)                                                            # (first forward: weight packing launches)
    syn.strict_nan_propagation = True
    G(*args, **kw)
    syn.strict_nan_propagation = False
    img_e, n_e = launches(lambda: G(*args, **kw))
    syn.strict_nan_propagation = True
    img_s, n_s = launches(lambda: G(*args, **kw))
    assert n_s > n_e, (n_s, n_e)                                              # the literal forward ran the adaptor's layers on top
    assert torch.equal(img_e, img_s)
    with torch.no_grad():
        syn.depth_adaptor.layers[0].weight[0, 0, 0, 0] = float('nan')
    assert torch.isnan(G(*args, **kw)).all()                                  # strict: max() of a NaN map is NaN, 0.0 * NaN poisons every pixel
    syn.strict_nan_propagation = False
    assert torch.equal(G(*args, **kw), img_e)                                 # elided: the adaptor is not on the path
    out = G(*args, **kw, render_opts=dict(return_depth_adapted=True))
    assert torch.isnan(out.depth_adapted).any()                               # (c) asked for -> evaluated
    assert G(*args, **kw, render_opts=dict(concat_depth=True)).shape[1] == 4
    # (d) a training-mode forward is never elided: with out_strategy='random' (the 3dgp.yaml default) DepthAdaptor.forward draws
    # np.random.choice per sample (networks_depth_adaptor.py:86-92) -- the host RNG stream must advance exactly as the reference's, which
    # always evaluates the adaptor (networks_epigraf.py:246-253)
    tag_a, cfg_a = tdgp.config.configs_adaptor_goldens()[0]
    assert cfg_a.depth_adaptor.out_strategy == 'random'
    Ga = _gen(tdgp, cfg_a, 51)
    inp_a = tdgp.weights.synthetic_inputs(cfg_a, batch=2, seed=54)
    args_a = (T(inp_a['z']), T(inp_a['c']), {k: T(v) for k, v in inp_a['camera'].items()})
    kw_a = dict(noise_mode='const', u_coarse=T(inp_a['u_coarse']), u_fine=T(inp_a['u_fine']))
    Ga(*args_a, **kw_a)                                                       # (weight packing launches)
    _, n_eval = launches(lambda: Ga(*args_a, **kw_a))
    np.random.seed(11)
    untouched = np.random.get_state()[1].copy()
    Ga(*args_a, **kw_a)
    assert np.array_equal(np.random.get_state()[1], untouched)               # eval: elided, and the adaptor would not have drawn anyway
    Ga.synthesis.train()
    try:
        Ga(*args_a, **kw_a)                                                   # (first training-mode forward: any first-use launches)
        np.random.seed(11)
        _, n_train = launches(lambda: Ga(*args_a, **kw_a))
        after = np.random.random()
    finally:
        Ga.synthesis.eval()
    assert n_train > n_eval, (n_train, n_eval)                                # the adaptor's layers ran
    np.random.seed(11)
    assert after != np.random.random()                                        # ... and the np.random.choice draw was consumed


def test_generator_feature_loop(tdgp):
    """metrics.compute_feature_stats_for_generator (metric_utils.py:288-320) end to end on the HIP generator: priors -> camera
    adaptor -> G.forward -> uint8 -> detector -> FeatureStats; deterministic under a fixed seed, max_items respected."""
    tag, cfg = tdgp.config.configs_adaptor_goldens()[0]
    G = _gen(tdgp, cfg, 51)
    det = lambda im: tdgp.distributed.stand_in_features(im, 64)        # noqa: E731

    def run():
        torch.manual_seed(7)
        np.random.seed(7)
        return tdgp.metrics.compute_feature_stats_for_generator(G, det, max_items=10, batch_size=8, batch_gen=4, device=DEV, capture_all=True, capture_mean_cov=True,
                                                                G_kwargs=dict(noise_mode='const'))
    a, b = run(), run()
    assert a.num_items == 10 and a.is_full() and a.get_all().shape == (10, 64)
    np.testing.assert_array_equal(a.get_all(), b.get_all())
    mean, cov = a.get_mean_cov()
    assert np.isfinite(mean).all() and np.isfinite(cov).all()
    # rank-deficient covariance (10 samples, 64 features): sqrtm is only accurate to ~sqrt(eps) * |cov| there
    assert abs(tdgp.metrics.frechet_distance(mean, cov, mean, cov)) < 1e-4 * max(1.0, float(np.trace(cov)))


def test_generate_trajectory(tdgp):
    """inference.generate / generate_trajectory (inference_utils.py:88-126): chunked G.synthesis calls with noise_mode='const',
    [0,1] mapping, depth normalisation, [num_cameras, num_samples, ...] layout."""
    cfg = tdgp.config.config_tiny()
    G = _gen(tdgp, cfg, 21)
    inp = tdgp.weights.synthetic_inputs(cfg, batch=2, seed=9)
    ws = G.mapping(T(inp['z']), T(inp['c']))
    canon = tdgp.generator.TensorGroup(**{k: T(v) for k, v in inp['camera'].items()})
    cams = tdgp.inference.generate_camera_trajectory(tdgp.inference_golden_trajectories()['points'], canon)      # 3 cameras per sample
    R, S = cfg.img_resolution ** 2, cfg.num_ray_steps
    u1, u2 = torch.rand(6, R, S, device=DEV), torch.rand(6 * R, S, device=DEV)
    # explicit RNG tensors only make sense for one chunk: generate everything in one G.synthesis call ...
    frames = tdgp.inference.generate_trajectory(G, ws, cams, batch_size=6, render_opts=dict(return_depth=True), u_coarse=u1, u_fine=u2)
    assert frames.img.shape == (3, 2, 3, 16, 16) and frames.depth.shape == (3, 2, 1, 16, 16)
    ref = G.synthesis(ws.repeat_interleave(3, dim=0), camera_params=cams.to(dtype=torch.float32, device=DEV), noise_mode='const',
                      render_opts=dict(return_depth=True), u_coarse=u1, u_fine=u2)
    img = (ref.img.clamp(-1, 1).cpu() * 0.5 + 0.5).reshape(2, 3, 3, 16, 16).permute(1, 0, 2, 3, 4)
    dep = (((ref.depth - 1.0) / 0.5 * 2.0).clamp(-1, 1).cpu() * 0.5 + 0.5).reshape(2, 3, 1, 16, 16).permute(1, 0, 2, 3, 4)
    assert torch.equal(frames.img, img) and torch.allclose(frames.depth, dep, atol=1e-6)
    # ... and the chunked form (device RNG) has the same shape and range
    chunked = tdgp.inference.generate_trajectory(G, ws, cams, batch_size=4)
    assert chunked.shape == (3, 2, 3, 16, 16) and float(chunked.min()) >= 0.0 and float(chunked.max()) <= 1.0
    mean_cam = tdgp.inference.approximate_mean_camera_params(G, num_samples=64, device=DEV)
    assert mean_cam.angles.shape == (1, 3) and mean_cam.fov.shape == (1,)


@pytest.mark.parametrize('tag', ['', '_clamp'])
def test_bias_act_grad_plugin(tdgp, tag):
    """bias_act plugin with grad = 1 / 2 (bias_act.cpp:32; the calls of bias_act.py:172-197) against autograd through the reference's
    CPU bias_act, for every activation."""
    from test_oracle_golden import _bias_act_grad_case
    g = load_golden('bias_act_grad')
    empty = torch.empty([0], device=DEV)
    opt = lambda a: empty if a is None else T(a)      # noqa: E731
    ids = dict(linear=1, relu=2, lrelu=3, tanh=4, sigmoid=5, elu=6, selu=7, softplus=8, swish=9)
    for act, idx in ids.items():
        xref, yref, kw = _bias_act_grad_case(g, act, tag)
        spec = tdgp.ops.bias_act.activation_funcs[act]
        alpha, gain, clamp = spec.def_alpha, kw.get('gain', spec.def_gain), kw.get('clamp', -1)
        P = tdgp.compat.BiasActPlugin.bias_act
        dx = P(T(g['dy']), T(g['b']), opt(xref), opt(yref), empty, 1, 1, idx, alpha, gain, clamp)
        assert_close(N(dx), g[f'dx_{act}{tag}'], 2e-5, f'dx {act}{tag}', 1.0)
        ddx = P(T(g['d2']), T(g['b']), opt(xref), opt(yref), T(g['dy']), 2, 1, idx, alpha, gain, clamp)
        assert_close(N(ddx), g[f'ddx_{act}{tag}'], 5e-5, f'ddx {act}{tag}', 1.0)
    # 16-bit storage path
    xh = T(g['dy']).half()
    dxh = tdgp.compat.BiasActPlugin.bias_act(xh, T(g['b']).half(), empty.half(), T(g['y_lrelu']).half(), empty.half(), 1, 1, 3, 0.2, float(np.sqrt(2)), -1)
    assert dxh.dtype == torch.float16 and float((dxh.float() - T(g['dx_lrelu'])).abs().max()) < 2e-2


@pytest.mark.parametrize('name', ['up2', 'fir', 'down2', 'asym'])
def test_upfirdn2d_backward(tdgp, name):
    """upfirdn2d's input gradient through the same HIP entry point with swapped factors and the flipped filter (upfirdn2d.py:251-265)."""
    from conftest import UPFIRDN_GRAD_CASES, upfirdn2d_backward_args
    g = load_golden('upfirdn2d_grad')
    dy, f = g[f'{name}_dy'], g[f'{name}_f']
    dx = tdgp.ops.upfirdn2d.upfirdn2d(T(dy), T(f), **upfirdn2d_backward_args(UPFIRDN_GRAD_CASES[name], f.shape, dy.shape))
    assert_close(N(dx), g[f'{name}_dx'], 2e-6, f'dx {name}', 1.0)


# ------------------------------------------------------------------------------------------------ training-mode forward
def test_field_density_noise(tdgp, oracle):
    """sigma += n * density_noise inside the field kernel (tri_plane_renderer.py:185-186), coords mode and ray-walk mode:
    rgb unchanged bit for bit, sigma = (plain sigma) + fl(n * std) exactly."""
    rs = np.random.RandomState(13)
    B, F, H, hid, P = 2, 32, 32, 64, 777
    planes = T(rs.randn(B, 3 * F, H, H))
    mlp = _mlp(tdgp, rs.randn(hid, F), 0.3 * rs.randn(hid), rs.randn(4, hid), 0.3 * rs.randn(4), 'classical')
    coords = T(rs.uniform(-0.55, 0.55, (B, P, 3)))
    n = T(rs.randn(B, P, 1))
    plain = tdgp.renderer.simple_tri_plane_renderer(planes, coords, mlp, scale=0.5)
    noisy = tdgp.renderer.simple_tri_plane_renderer(planes, coords, mlp, scale=0.5, sigma_noise=n, density_noise=0.37)
    np.testing.assert_array_equal(N(noisy['rgb']), N(plain['rgb']))
    np.testing.assert_array_equal(N(noisy['sigma']), N(plain['sigma']) + (N(n) * np.float32(0.37)).astype(np.float32))
    # ray mode (the table-driven walk with parked outputs), S not a multiple of 8
    R, S = 36, 12
    cam = dict(angles=T([[0.3, 1.2, 0.0], [-0.6, 1.8, 0.0]]), radius=T([1.0, 1.0]), look_at=T(np.zeros((2, 3))))
    ro, rd = tdgp.renderer.sample_rays(tdgp.renderer.compute_cam2world_matrix(cam), T([25.0, 40.0]), (6, 6))
    t = T(np.sort(rs.uniform(0.75, 1.25, (B, R, S)), axis=2))
    hw = tdgp.renderer.planes_to_hwc(planes)
    mp = tdgp.renderer._mlp_params(mlp)
    n2 = T(rs.randn(B, R * S))
    for ray_w in (0, 6):
        a = tdgp.renderer._field(hw, mp, 0.5, ray_o=ro, ray_d=rd, t=t, ray_w=ray_w)
        b = tdgp.renderer._field(hw, mp, 0.5, ray_o=ro, ray_d=rd, t=t, ray_w=ray_w, sigma_noise=n2, density_noise=1.5)
        np.testing.assert_array_equal(N(b[..., :3]), N(a[..., :3]))
        np.testing.assert_array_equal(N(b[..., 3]), N(a[..., 3]) + (N(n2) * np.float32(1.5)).astype(np.float32))
    # device-side draws when none are given: right scale, different every call
    c = tdgp.renderer._field(hw, mp, 0.5, ray_o=ro, ray_d=rd, t=t, density_noise=2.0)
    d = N(c[..., 3]) - N(a[..., 3])
    assert 1.7 < d.std() < 2.3 and abs(d.mean()) < 0.3
    with pytest.raises(RuntimeError):
        tdgp.renderer._field(hw, mp, 0.5, ray_o=ro, ray_d=rd, t=t, sigma_noise=n2[:, :5], density_noise=1.0)


def test_training_mode_forward(tdgp, oracle):
    """G.train(): patch rays at train_resolution, density noise from progressive_update, explicit draws -> the reference's
    training-mode image (tests/golden/train_forward.npz); fused chain with the fine-pass draws carried through the sort."""
    g = load_golden('train_forward')
    cfg = tdgp.config.config_train_golden()
    G = _gen(tdgp, cfg, 91).train()
    G.progressive_update(3000)
    ws = G.mapping(T(g['z']), T(g['c']), update_emas=True)
    assert_close(N(ws), g['ws'], 1e-5, 'ws', 1.0)
    assert_close(N(G.mapping.w_avg), g['w_avg_after'], 1e-6, 'w_avg', 1.0)
    kw = dict(camera_params=_cam(g), patch_params=dict(scales=T(g['scales']), offsets=T(g['offsets'])), noise_mode='const',
              u_coarse=T(g['u_coarse']), u_fine=T(g['u_fine']), render_opts=dict(return_depth=True))
    out = G.synthesis(T(g['ws']), n_coarse=T(g['n_coarse']), n_fine=T(g['n_fine']), **kw)
    assert out.img.shape == (2, 3, 16, 16)
    sd = tdgp.weights.random_state_dict(cfg, seed=91, exercise_all=True)
    cam = {k[4:]: v for k, v in g.items() if k.startswith('cam_')}
    ex_img, ex_depth = oracle.synthesis_forward(sd, cfg.to_dict(), g['ws'], cam, g['u_coarse'], g['u_fine'], 'const',
                                                training=dict(resolution=16, patch_scales=g['scales'], patch_offsets=g['offsets'],
                                                              density_noise=float(g['nerf_noise_std']), n_coarse=g['n_coarse'], n_fine=g['n_fine']))
    assert_image_parity(N(out.img), g, 'img (training mode)', exact=ex_img)
    assert_image_parity(N(out.depth), g, 'depth (training mode)', 'depth', exact=ex_depth)
    # fresh device-side noise: a different image each call, and eval() is back to the full-resolution noise-free forward
    a, b = G.synthesis(T(g['ws']), **kw).img, G.synthesis(T(g['ws']), **kw).img
    assert float((a - b).abs().max()) > 1e-4
    G.eval()
    kw.pop('patch_params')
    kw.update(u_coarse=None, u_fine=None)
    assert G.synthesis(T(g['ws']), **kw).img.shape == (2, 3, cfg.img_resolution, cfg.img_resolution)


def test_depth_adaptor_training_selection(tdgp):
    """DepthAdaptor in .train() with out_strategy 'random' (networks_depth_adaptor.py:86-97): per-sample head drawn with the numpy
    RNG from the annealed linear distribution -- same seed, same heads as the reference."""
    g = load_golden('train_forward')
    tag, cfg = tdgp.config.configs_adaptor_goldens()[0]
    G = _gen(tdgp, cfg, 51)
    da = G.synthesis.depth_adaptor.train()
    da.progressive_update(4000)
    assert abs(da.start_p - float(g['da_start_p'])) < 1e-7
    np.random.seed(95)
    out = da(T(g['da_depth']), T(g['da_w']))
    assert_close(N(out), g['da_out'], 5e-6, 'randomly selected heads', 1.0)


# ------------------------------------------------------------------------------------------------ conv2d_gradfix
@pytest.mark.parametrize('name', ['k3', 'k1', 'k5', 'k3s2', 'k3p0'])
def test_conv2d_weight_grad_golden(tdgp, name):
    """tdgp_conv2d_weight_grad (MFMA, split over the pixel dimension) against autograd through the reference's conv2d_gradfix."""
    from conftest import CONV_GRAD_CASES
    g, c = load_golden('conv2d_grad'), CONV_GRAD_CASES[name]
    dw = tdgp.ops.conv2d_gradfix.conv2d_weight_grad(T(g[f'{name}_x']), T(g[f'{name}_dy']), g[f'{name}_w'].shape, c['stride'], c['pad'])
    assert_close(N(dw), g[f'{name}_dw'], 5e-6, 'dw', 1.0)


@pytest.mark.parametrize('name', ['k3', 'k1', 'k5'])
def test_conv2d_gradfix_autograd(tdgp, name):
    """ops.conv2d_gradfix.conv2d as an autograd function on the GPU: forward, input / weight / bias gradients vs the reference's."""
    g = load_golden('conv2d_grad')
    cg = tdgp.ops.conv2d_gradfix
    x, w, b = (T(g[f'{name}_{k}']).requires_grad_(True) for k in 'xwb')
    y = cg.conv2d(x, w, b, padding=w.shape[2] // 2)
    assert_close(N(y.detach()), g[f'{name}_y'], 5e-6, 'y', 1.0)
    dx, dw, db = torch.autograd.grad(y, [x, w, b], T(g[f'{name}_dy']))
    assert_close(N(dx), g[f'{name}_dx'], 5e-6, 'dx', 1.0)
    assert_close(N(dw), g[f'{name}_dw'], 5e-6, 'dw', 1.0)
    assert_close(N(db), g[f'{name}_db'], 5e-6, 'db', 1.0)
    with cg.no_weight_gradients():
        y = cg.conv2d(x, w, b, padding=w.shape[2] // 2)
        dx2, = torch.autograd.grad(y, [x], T(g[f'{name}_dy']))
        assert torch.equal(dx, dx2)
        y = cg.conv2d(x, w, b, padding=w.shape[2] // 2)
        assert torch.autograd.grad(y, [w], T(g[f'{name}_dy']), allow_unused=True)[0] is None


@pytest.mark.parametrize('B,cin,cout,H,W,k,stride,pad', [(4, 128, 96, 64, 64, 3, 1, 1), (2, 70, 130, 33, 47, 3, 2, 1), (8, 64, 64, 32, 32, 1, 1, 0),
                                                         (1, 33, 65, 40, 24, 5, 1, 2)])
def test_conv2d_weight_grad_oracle(tdgp, oracle, B, cin, cout, H, W, k, stride, pad):
    """Layer-sized weight gradients (several output tiles, ragged channel counts and widths, many pixel slices) vs the
    double-accumulating oracle, and run-to-run determinism of the sliced sum."""
    rs = np.random.RandomState(cin + cout)
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    x = rs.randn(B, cin, H, W).astype(np.float32)
    dy = rs.randn(B, cout, OH, OW).astype(np.float32)
    ref = oracle.conv2d_weight_grad(x, dy, k, stride, pad)
    dw = tdgp.ops.conv2d_gradfix.conv2d_weight_grad(T(x), T(dy), (cout, cin, k, k), stride, pad)
    assert_close(N(dw), ref, 2e-5, 'dw', 1.0)
    assert torch.equal(dw, tdgp.ops.conv2d_gradfix.conv2d_weight_grad(T(x), T(dy), (cout, cin, k, k), stride, pad))


@pytest.mark.parametrize('tag', ['cl_inf', 'cl_noinf_lastback', 'cl_relu', 'mip_inf', 'mip_noinf_white_bias'])
def test_ray_march_grad(tdgp, oracle, tag):
    """tdgp_ray_march_grad against autograd through the reference marchers (goldens) and, on a larger random batch with and
    without the optional incoming gradients, against the double-precision oracle."""
    from conftest import MARCH_GRAD_CASES
    g, kw = load_golden('march_grad'), dict(MARCH_GRAD_CASES[tag])
    mode = kw.pop('mode')
    dc, dd = tdgp.renderer.ray_march_backward(T(g[f'{tag}_c']), T(g['densities']), T(g['depths']), kw, mode, T(g[f'{tag}_d_rgb']), T(g[f'{tag}_d_depth']),
                                              T(g[f'{tag}_d_weights']))
    assert_close(N(dc), g[f'{tag}_dc'], 5e-6, 'd_colors', 1.0)
    assert_close(N(dd), g[f'{tag}_dd'], 2e-5, 'd_densities', 1.0)
    rs = np.random.RandomState(31)
    B, R, S = 2, 700, 128
    colors = rs.rand(B, R, S, 3).astype(np.float32)
    dens = (rs.randn(B, R, S, 1) * 2).astype(np.float32)
    depths = np.sort(0.75 + 0.5 * rs.rand(B, R, S, 1).astype(np.float32), axis=2)
    d_rgb = rs.randn(B, R, 3).astype(np.float32)
    rc, rd = oracle.ray_march_grad(colors, dens, depths, d_rgb, mode=mode, **kw)
    dc, dd = tdgp.renderer.ray_march_backward(T(colors), T(dens), T(depths), kw, mode, T(d_rgb))
    assert_close(N(dc), rc, 1e-5, 'd_colors (S=128)', 1.0)
    assert_close(N(dd), rd, 5e-5, 'd_densities (S=128)', 1.0)


@pytest.mark.parametrize('S', [2, 5, 63, 64, 65, 129, 200, 256])
@pytest.mark.parametrize('tag', ['cl_inf', 'cl_noinf_lastback', 'mip_inf', 'mip_noinf_white_bias'])
def test_ray_march_grad_sample_counts(tdgp, oracle, tag, S):
    """The wave-per-ray gradient kernel at every lane layout (1, 2 and 4 intervals per lane), with ragged last lanes, the shortest rays and
    the longest the entry point accepts, with all three incoming gradients: vs the double-precision oracle."""
    from conftest import MARCH_GRAD_CASES
    kw = dict(MARCH_GRAD_CASES[tag])
    mode = kw.pop('mode')
    rs = np.random.RandomState(S)
    B, R = 2, 37
    M = S if mode == 'classical' else (S if kw.get('use_inf_depth', True) else S - 1)
    colors = rs.rand(B, R, S, 3).astype(np.float32)
    dens = (rs.randn(B, R, S, 1) * 2).astype(np.float32)
    depths = np.sort(0.75 + 0.5 * rs.rand(B, R, S, 1).astype(np.float32), axis=2)
    d_rgb, d_depth, d_w = rs.randn(B, R, 3).astype(np.float32), rs.randn(B, R, 1).astype(np.float32), rs.randn(B, R, M, 1).astype(np.float32)
    rc, rd = oracle.ray_march_grad(colors, dens, depths, d_rgb, d_depth, d_w, mode=mode, **kw)
    dc, dd = tdgp.renderer.ray_march_backward(T(colors), T(dens), T(depths), kw, mode, T(d_rgb), T(d_depth), T(d_w))
    assert_close(N(dc), rc, 1e-5, f'd_colors (S={S})', 1.0)
    assert_close(N(dd), rd, 5e-5, f'd_densities (S={S})', 1.0)


@pytest.mark.parametrize('tag', ['small', 'hot'])
@pytest.mark.parametrize('marcher', ['classical', 'mip'])
def test_field_grad(tdgp, oracle, tag, marcher):
    """tdgp_triplane_field_grad (MFMA MLP backward + atomic scatter) against autograd through the reference; the weight gradients
    are bit-identical run to run, the plane gradient (atomics) to rounding."""
    g = load_golden('field_grad')
    k = f'{tag}_{marcher}_'
    mlp = _mlp(tdgp, g[k + 'w0'], g[k + 'b0'], g[k + 'w1'], g[k + 'b1'], marcher)
    R = tdgp.renderer
    res = R.simple_tri_plane_renderer_backward(T(g[f'{tag}_planes']), T(g[f'{tag}_coords']), mlp, T(g[f'{tag}_d_rgb']), T(g[f'{tag}_d_sigma']), scale=0.5)
    got = dict(d_planes=R.planes_from_hwc(res[0]), d_w0=res[1], d_b0=res[2], d_w1=res[3], d_b1=res[4])
    for name, t in got.items():
        assert_close(N(t), g[k + name], 5e-5, name, 1.0)
    res2 = R.simple_tri_plane_renderer_backward(T(g[f'{tag}_planes']), T(g[f'{tag}_coords']), mlp, T(g[f'{tag}_d_rgb']), T(g[f'{tag}_d_sigma']), scale=0.5)
    for a, b in zip(res[1:], res2[1:]):
        assert torch.equal(a, b)
    assert_close(N(res2[0]), N(res[0]), 1e-5, 'd_planes run to run', 1.0)


@pytest.mark.parametrize('tag', ['small', 'hot'])
@pytest.mark.parametrize('marcher', ['classical', 'mip'])
def test_field_grad_wrt_coords(tdgp, oracle, tag, marcher):
    """The gradient w.r.t. the sample positions (grid_sampler's grid gradient through the plane mean and coords / scale; what a camera is
    trained through, loss.py:69-83) against autograd through the reference's simple_tri_plane_renderer; the other outputs unchanged by it;
    and on points outside the cube (zero padding on both sides of a cell) against the double-precision oracle."""
    g = load_golden('field_grad')
    k = f'{tag}_{marcher}_'
    ws = [g[k + n] for n in ('w0', 'b0', 'w1', 'b1')]
    mlp = _mlp(tdgp, *ws, marcher)
    R = tdgp.renderer
    args = (T(g[f'{tag}_planes']), T(g[f'{tag}_coords']), mlp, T(g[f'{tag}_d_rgb']), T(g[f'{tag}_d_sigma']))
    res = R.simple_tri_plane_renderer_backward(*args, scale=0.5, coords_grad=True)
    assert_close(N(res[5]), g[k + 'd_coords'], 5e-5, 'd_coords', 1.0)
    plain = R.simple_tri_plane_renderer_backward(*args, scale=0.5)
    for a, b in zip(res[1:5], plain[1:]):
        assert torch.equal(a, b)
    only = R.simple_tri_plane_renderer_backward(*args, scale=0.5, planes_grad=False, coords_grad=True)
    assert only[0] is None and torch.equal(only[5], res[5])
    rs = np.random.RandomState(5)
    B, P = g[f'{tag}_planes'].shape[0], 16 * 41 + 3
    coords = rs.uniform(-0.75, 0.75, (B, P, 3)).astype(np.float32)               # the cube is [-0.5, 0.5]^3 here: a third of the points are outside
    d_rgb, d_sigma = rs.randn(B, P, 3).astype(np.float32), rs.randn(B, P, 1).astype(np.float32)
    ref = oracle.triplane_field_grad(g[f'{tag}_planes'], coords, *ws, d_rgb, d_sigma, scale=0.5, mlp_mode=marcher, return_coords=True)
    got = R.simple_tri_plane_renderer_backward(T(g[f'{tag}_planes']), T(coords), mlp, T(d_rgb), T(d_sigma), scale=0.5, coords_grad=True)
    assert_close(N(got[5]), ref[5], 1e-4, 'd_coords incl. zero padding', 1.0)


def test_field_grad_large(tdgp, oracle):
    """Many tiles per wave, a ragged last tile, several blocks: vs the double-precision oracle."""
    rs = np.random.RandomState(77)
    B, F, H, hid, P = 2, 32, 32, 64, 33333
    planes = rs.randn(B, 3 * F, H, H).astype(np.float32)
    coords = rs.uniform(-0.6, 0.6, (B, P, 3)).astype(np.float32)
    w0, b0, w1, b1 = rs.randn(hid, F).astype(np.float32), (0.3 * rs.randn(hid)).astype(np.float32), rs.randn(4, hid).astype(np.float32), (0.3 * rs.randn(4)).astype(np.float32)
    d_rgb, d_sigma = rs.randn(B, P, 3).astype(np.float32), rs.randn(B, P, 1).astype(np.float32)
    ref = oracle.triplane_field_grad(planes, coords, w0, b0, w1, b1, d_rgb, d_sigma, scale=0.5, mlp_mode='classical')
    mlp = _mlp(tdgp, w0, b0, w1, b1, 'classical')
    R = tdgp.renderer
    res = R.simple_tri_plane_renderer_backward(T(planes), T(coords), mlp, T(d_rgb), T(d_sigma), scale=0.5)
    got = (R.planes_from_hwc(res[0]),) + tuple(res[1:])
    for a, b, name in zip(got, ref, ('d_planes', 'd_w0', 'd_b0', 'd_w1', 'd_b1')):
        assert_close(N(a), b, 1e-4, name, 1.0)


@pytest.mark.parametrize('marcher', ['classical', 'mip'])
def test_importance_renderer_backward(tdgp, marcher):
    """ImportanceRenderer.backward: replayed forward + tdgp_ray_march_grad + un-sort + tdgp_triplane_field_grad for both passes,
    against autograd through the reference's ImportanceRenderer.forward (same stratification / inverse-CDF draws)."""
    g = load_golden('render_grad')
    mlp = _mlp(tdgp, *(g[f'{marcher}_{n}'] for n in ('w0', 'b0', 'w1', 'b1')), marcher)
    opts = dict(box_size=1.0, num_proposal_steps=8, num_fine_steps=8, clamp_mode='softplus', use_inf_depth=True, ray_start=0.75, ray_end=1.25,
                white_back=(marcher == 'mip'), density_bias=0.0, u_coarse=T(g['u_coarse']), u_fine=T(g['u_fine']))
    rend = tdgp.renderer.ImportanceRenderer(marcher)
    rgb, _, _, _ = rend(T(g['planes']), mlp, T(g['ray_o']), T(g['ray_d']), opts)
    assert_close(N(rgb), g[f'{marcher}_rgb'], 1e-5, 'rgb', 1.0)
    res = rend.backward(T(g['planes']), mlp, T(g['ray_o']), T(g['ray_d']), opts, T(g['d_rgb']), T(g['d_depth']))
    for name in ('planes', 'w0', 'b0', 'w1', 'b1'):
        assert_close(N(res[name]), g[f'{marcher}_d_{name}'], 1e-4, 'd_' + name, 1.0)


@pytest.mark.parametrize('tag,demod', [('c3', True), ('rgb', False), ('c3big', True)])
def test_modulated_conv2d_autograd(tdgp, tag, demod):
    """ops.modconv.modulated_conv2d_autograd: forward and gradients w.r.t. x, weight, styles (incl. the demodulation terms) against
    autograd through the reference's unfused modulated_conv2d -- the path train.py takes."""
    g = load_golden('modconv_grad')
    x, w, s = (T(g[f'{tag}_{k}']).requires_grad_(True) for k in 'xws')
    y = tdgp.ops.modconv.modulated_conv2d_autograd(x, w, s, demodulate=demod)
    assert_close(N(y.detach()), g[f'{tag}_y'], 1e-5, 'y', 1.0)
    dx, dw, ds = torch.autograd.grad(y, [x, w, s], T(g[f'{tag}_dy']))
    assert_close(N(dx), g[f'{tag}_dx'], 2e-5, 'dx', 1.0)
    assert_close(N(dw), g[f'{tag}_dw'], 2e-5, 'dw', 1.0)
    assert_close(N(ds), g[f'{tag}_ds'], 2e-5, 'ds', 1.0)


@pytest.mark.parametrize('tag,k,st,pad', [('s2', 3, 2, 0), ('s2p1', 3, 2, 1), ('s1p0', 3, 1, 0), ('k1s2', 1, 2, 0)])
def test_conv2d_strided(tdgp, tag, k, st, pad):
    """tdgp_conv2d (strided / unpadded convolutions) against the reference's conv2d_gradfix.conv2d."""
    g = load_golden('modconv_grad')
    y = tdgp.ops.conv2d_gradfix.conv2d_strided(T(g[f'conv_{tag}_x']), T(g[f'conv_{tag}_w']), T(g[f'conv_{tag}_b']), stride=st, padding=pad)
    assert_close(N(y), g[f'conv_{tag}_y'], 1e-5, 'y', 1.0)


@pytest.mark.parametrize('tag', ['up', 'upbig'])
def test_modulated_conv2d_up_autograd(tdgp, oracle, tag):
    """The x2-upsampling synthesis layer under autograd: transposed conv + FIR forward (fused kernel), FIR-adjoint, stride-2 adjoint
    convolution, role-swapped weight gradient, demodulation terms -- against autograd through the reference's unfused path."""
    g = load_golden('modconv_grad')
    x, w, s = (T(g[f'{tag}_{k}']).requires_grad_(True) for k in 'xws')
    f = T(oracle.setup_filter([1, 3, 3, 1]))
    y = tdgp.ops.modconv.modulated_conv2d_up_autograd(x, w, s, f)
    assert_close(N(y.detach()), g[f'{tag}_y'], 1e-5, 'y', 1.0)
    dx, dw, ds = torch.autograd.grad(y, [x, w, s], T(g[f'{tag}_dy']))
    assert_close(N(dx), g[f'{tag}_dx'], 2e-5, 'dx', 1.0)
    assert_close(N(dw), g[f'{tag}_dw'], 2e-5, 'dw', 1.0)
    assert_close(N(ds), g[f'{tag}_ds'], 2e-5, 'ds', 1.0)


def test_synthesis_forward_autograd(tdgp):
    """G.synthesis.forward_autograd: the whole generator forward as a differentiable graph on the HIP kernels.  Gradient of
    sum(img * d_img) + sum(depth * d_depth) w.r.t. every synthesis parameter (56 tensors: const input, conv / ToRGB weights, biases,
    noise strengths, style affines, tri-plane MLP) and ws, against autograd through the reference's G.synthesis (tiny config)."""
    g = load_golden('synthesis_grad')
    cfg = tdgp.config.config_tiny()
    G = _gen(tdgp, cfg, 101)
    for p in G.parameters():
        p.requires_grad_(True)
    ws = T(g['ws']).requires_grad_(True)
    out = G.synthesis.forward_autograd(ws, camera_params=_cam(g), noise_mode='const', u_coarse=T(g['u_coarse']), u_fine=T(g['u_fine']),
                                       render_opts=dict(return_depth=True))
    assert_image_parity(N(out.img.detach()), g, 'img (autograd path)', pix_tol=2e-4)
    names = [k[len('grad::'):] for k in g.keys() if k.startswith('grad::')]
    params = dict(G.named_parameters())
    grads = torch.autograd.grad([out.img, out.depth], [ws] + [params[n] for n in names], [T(g['d_img']), T(g['d_depth'])], allow_unused=True)
    assert_close(N(grads[0]), g['d_ws'], 2e-4, 'd_ws', 1.0)
    worst = 0.0
    for n, gr in zip(names, grads[1:]):
        assert gr is not None, n
        ref = g['grad::' + n]
        err = float(np.abs(N(gr) - ref).max() / max(np.abs(ref).max(), 1e-12))
        worst = max(worst, err)
        assert err <= 3e-4, (n, err)
    # the same module still serves the fused inference path
    img = G.synthesis(T(g['ws']), camera_params=_cam(g), noise_mode='const', u_coarse=T(g['u_coarse']), u_fine=T(g['u_fine']))
    assert_close(N(img), N(out.img.detach()), 2e-5, 'fused vs autograd path', 1.0)


@pytest.mark.parametrize('patch', [False, True])
def test_camera_gradients(tdgp, patch):
    """The loss differentiated w.r.t. the CAMERA (angles, fov, radius, look_at) -- rays as a differentiable graph
    (renderer.camera_rays_autograd), d(rays) from the field kernel's coordinate gradient -- against autograd through the reference's
    G.synthesis (rendering_utils.py:194-218, tri_plane_renderer.py:487-527 under autograd): what loss.py:76-77 trains the camera adaptor
    through.  Full image and patch rays."""
    g = load_golden('synthesis_grad')
    cfg = tdgp.config.config_tiny()
    G = _gen(tdgp, cfg, 101)
    cam = {k: v.clone().requires_grad_(True) for k, v in _cam(g).items()}
    pp = dict(scales=T(g['patch_scales']), offsets=T(g['patch_offsets'])) if patch else None
    out = G.synthesis.forward_autograd(T(g['ws']), camera_params=cam, patch_params=pp, noise_mode='const', u_coarse=T(g['u_coarse']), u_fine=T(g['u_fine']),
                                       render_opts=dict(return_depth=True))
    ref_img = g['img_patch'] if patch else g['img']
    assert_close(N(out.img.detach()), ref_img, 2e-5, 'image from the differentiable rays', 1.0)
    keys = ('angles', 'fov', 'radius', 'look_at')
    grads = torch.autograd.grad([out.img, out.depth], [cam[k] for k in keys], [T(g['d_img']), T(g['d_depth'])], allow_unused=True)
    tag = '_patch' if patch else ''
    for k, gr in zip(keys, grads):
        ref = g[f'd_cam{tag}_{k}']
        got = np.zeros_like(ref) if gr is None else N(gr)
        report_parity(f'camera gradient{tag} {k}', err=float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-12)))
        assert_close(got, ref, 1e-3, f'd {k}', 1.0)
    # the camera adaptor is now trainable through the renderer: its parameters receive gradients
    acfg = tdgp.config.configs_adaptor_goldens()[0][1]
    GA = _gen(tdgp, acfg, 77)
    for p_ in GA.parameters():
        p_.requires_grad_(True)
    z, c = T(g['z']), T(g['c'])
    cam_a = GA.synthesis.camera_adaptor(_cam(g), z, c)
    o = GA.synthesis.forward_autograd(T(g['ws']), camera_params=cam_a, noise_mode='const', u_coarse=T(g['u_coarse']), u_fine=T(g['u_fine']))
    (o * T(g['d_img'])).sum().backward()
    ga = [p_.grad for n_, p_ in GA.synthesis.camera_adaptor.named_parameters()]
    assert all(x is not None and torch.isfinite(x).all() for x in ga) and sum(float(x.abs().sum()) for x in ga) > 0


@pytest.mark.parametrize('tag', ['plain', 'full', 'extra'])
def test_discriminator_gpu(tdgp, tag):
    """The discriminator on the HIP ops (stride-1 and stride-2 convolutions forward and backward, upfirdn2d, bias_act): logits,
    d/d img and every parameter gradient against autograd through the reference."""
    from conftest import check_discriminator
    tdgp._lib.profile_enable(True)
    try:
        assert check_discriminator(tdgp, tag, DEV, 1e-4) >= 17
        torch.cuda.synchronize()
        launched = set(tdgp._lib.profile_report())
    finally:
        tdgp._lib.profile_enable(False)
    # the convolutions really ran on the library's kernels, forward and backward (no silent torch fallback)
    for k in ('conv_mfma_kernel', 'conv_strided_mfma_kernel', 'conv_wgrad_mfma_kernel', 'bias_act_grad_kernel'):
        assert k in launched, (k, sorted(launched))


@pytest.mark.parametrize('tag', ['plain', 'full'])
def test_discriminator_r1_gpu(tdgp, tag):
    """R1 regularisation on the HIP ops: d logits / d img under create_graph=True, its squared norm, and the gradient of that w.r.t. the
    parameters -- the convolution backward passes are themselves differentiable functions over the same kernels
    (conv2d_gradfix.py:120-166), bias_act has its second-order kernel, upfirdn2d recurses."""
    from conftest import check_discriminator_r1
    assert check_discriminator_r1(tdgp, tag, DEV, 3e-4) >= 10


def test_stylegan2_loss_phases(tdgp):
    """training.StyleGAN2Loss.accumulate_gradients for the phases Gmain, Dmain, Dreg on the HIP ops (differentiable generator, patch-
    conditioned hyper-modulated discriminator, R1 with second-order gradients): the gradients left in G / D after each phase against
    the reference's StyleGAN2Loss on the same weights, patch parameters and renderer draws."""
    g = load_golden('loss')
    TR = tdgp.training
    cfg = tdgp.config.config_tiny()
    cfg.use_noise = False
    cfg.patch_resolution = 16
    dcfg = tdgp.discriminator.DiscriminatorConfig(c_dim=0, cbase=256, cmax=16, patch_params_cond=True, hyper_mod=True, mbstd_group_size=2)
    G = _gen(tdgp, cfg, 201).train()
    D = tdgp.discriminator.seeded_discriminator(dcfg, 16, 3, seed=202).to(DEV).train()
    pcfg = TR.PatchConfig(enabled=True, distribution='uniform', resolution=16, min_scale_trg=0.5, max_scale=1.0, anneal_kimg=10, mbstd_group_size=2)
    loss = TR.StyleGAN2Loss(G, D, DEV, r1_gamma=2.0, patch_cfg=pcfg, synthesis_kwargs=dict(u_coarse=T(g['u_coarse']), u_fine=T(g['u_fine'])))
    pps = [dict(scales=T(g[f'pp{i}_scales']), offsets=T(g[f'pp{i}_offsets'])) for i in range(3)]
    queue = []
    orig = TR.sample_patch_params
    TR.sample_patch_params = lambda n, pc, device='cpu': queue.pop(0)
    B = 4
    c0 = torch.zeros(B, 0, device=DEV)

    def run(phase, pp_list):
        for m in (G, D):
            m.zero_grad(set_to_none=True)
        G.requires_grad_(phase.startswith('G'))
        D.requires_grad_(phase.startswith('D'))
        queue[:] = pp_list
        real = tdgp.generator.TensorGroup(img=T(g['real']), c=c0, depth=torch.zeros(B, 1, 32, 32, device=DEV))
        gen = tdgp.generator.TensorGroup(z=T(g['z']), c=c0, camera_params=tdgp.generator.TensorGroup(**_cam(g)))
        loss.accumulate_gradients(phase, real, gen, gain=1, cur_nimg=0)
        assert not queue

    def check(tag, module, tol):
        params = dict(module.named_parameters())
        n_checked = 0
        for k in g.keys():
            if not k.startswith(tag + '::'):
                continue
            parts = k.split('::')
            name = parts[-1]
            gr = params[name].grad
            assert gr is not None, (tag, name)
            gr = gr.cpu()
            got = gr.sum(1) if parts[1] == 'rows' else gr.sum(0) if parts[1] == 'cols' else gr
            ref = g[k]
            err = float(np.abs(got.numpy() - ref).max() / max(np.abs(ref).max(), 1e-12))
            assert err <= tol, (tag, name, err)
            n_checked += 1
        return n_checked

    try:
        run('Gmain', [pps[0]])
        assert check('Gmain', G, 2e-3) >= 50
        assert all(p.grad is None for p in D.parameters())
        run('Dmain', [pps[0], pps[1]])
        assert check('Dmain', D, 2e-3) >= 30
        run('Dreg', [pps[2]])
        assert check('Dreg', D, 2e-3) >= 20
        assert float(loss.stats['Loss/D/r1_penalty'].min()) > 0
    finally:
        TR.sample_patch_params = orig


@pytest.mark.parametrize('kind', ['l2', 'kl'])
def test_stylegan2_loss_knowledge_distillation(tdgp, kind):
    """The discriminator's distillation term (loss.py:279-314: features predicted from the real patches pulled to the dataset's
    embeddings, l2 or kl distance, weighted by patch size, faded out over kd.discr.anneal_kimg) inside phase Dmain: the schedule value
    and every gradient left in D -- feature head included -- against the reference's StyleGAN2Loss on the same weights and draws."""
    g = load_golden('loss_kd')
    TR = tdgp.training
    cfg = tdgp.config.config_tiny()
    cfg.use_noise = False
    cfg.patch_resolution = 16
    dcfg = tdgp.discriminator.DiscriminatorConfig(c_dim=0, cbase=256, cmax=16, patch_params_cond=True, hyper_mod=True, mbstd_group_size=2)
    G = _gen(tdgp, cfg, 201).train()
    D = tdgp.discriminator.seeded_discriminator(dcfg, 16, 3, seed=212, epilogue_kwargs=dict(feat_predict_dim=6)).to(DEV).train()
    pcfg = TR.PatchConfig(enabled=True, distribution='uniform', resolution=16, min_scale_trg=0.5, max_scale=1.0, anneal_kimg=10, mbstd_group_size=2)
    loss = TR.StyleGAN2Loss(G, D, DEV, r1_gamma=2.0, patch_cfg=pcfg, kd_weight=0.7, kd_anneal_kimg=100, kd_loss_type=kind,
                            synthesis_kwargs=dict(u_coarse=T(g['u_coarse']), u_fine=T(g['u_fine'])))
    assert loss.D_kd_weight == 0.7
    loss.progressive_update(25)
    assert abs(loss.D_kd_weight - float(g[f'{kind}_kd_weight'])) < 1e-7
    queue = [dict(scales=T(g[f'pp{i}_scales']), offsets=T(g[f'pp{i}_offsets'])) for i in range(2)]
    orig = TR.sample_patch_params
    TR.sample_patch_params = lambda n, pc, device='cpu': queue.pop(0)
    B = 4
    c0 = torch.zeros(B, 0, device=DEV)
    try:
        G.requires_grad_(False)
        D.requires_grad_(True)
        D.zero_grad(set_to_none=True)
        real = tdgp.generator.TensorGroup(img=T(g['real']), c=c0, depth=torch.zeros(B, 1, 32, 32, device=DEV), embs=T(g['embs']))
        gen = tdgp.generator.TensorGroup(z=T(g['z']), c=c0, camera_params=tdgp.generator.TensorGroup(**_cam(g)))
        loss.accumulate_gradients('Dmain', real, gen, gain=1, cur_nimg=0)
        assert not queue
    finally:
        TR.sample_patch_params = orig
    params = dict(D.named_parameters())
    n_checked, head = 0, 0
    for k in g.keys():
        if not k.startswith(kind + '::'):
            continue
        parts = k.split('::')
        gr = params[parts[-1]].grad
        assert gr is not None, parts[-1]
        gr = gr.cpu()
        got = gr.sum(1) if parts[1] == 'rows' else gr.sum(0) if parts[1] == 'cols' else gr
        err = float(np.abs(got.numpy() - g[k]).max() / max(np.abs(g[k]).max(), 1e-12))
        assert err <= 2e-3, (kind, parts[-1], err)
        n_checked += 1
        head += 'feat_out' in parts[-1]
    assert n_checked >= 30 and head >= 4
    assert torch.isfinite(loss.stats['Loss/kd/D_dist']).all() and float(loss.stats['Loss/kd/D_loss'].abs().sum()) > 0
    # weight 0 (every 3dgp config but the distilled ones): no feature head is evaluated
    plain = TR.StyleGAN2Loss(G, D, DEV, r1_gamma=2.0, patch_cfg=pcfg)
    assert plain.D_kd_weight == 0.0


def test_loss_trains_the_camera_adaptor(tdgp):
    """`learn_camera_dist=True` (loss.py:76-77): run_G applies the camera adaptor and the Gmain phase leaves finite, non-zero gradients in
    its parameters -- through ray generation and the field kernel's coordinate gradient; without the flag the adaptor gets none.  A
    generator without an adaptor refuses the flag."""
    TR = tdgp.training
    cfg = tdgp.config.configs_adaptor_goldens()[0][1]
    cfg.use_noise = False
    G = _gen(tdgp, cfg, 301).train()
    dcfg = tdgp.discriminator.DiscriminatorConfig(c_dim=0, cbase=256, cmax=16)
    use_depth = cfg.depth_adaptor is not None
    D = tdgp.discriminator.seeded_discriminator(dcfg, cfg.img_resolution, 4 if use_depth else 3, seed=302).to(DEV).train()
    B = 2
    inp = tdgp.weights.synthetic_inputs(cfg, batch=B, seed=303)
    c0 = torch.zeros(B, 0, device=DEV)
    gen = tdgp.generator.TensorGroup(z=T(inp['z']), c=c0, camera_params=tdgp.generator.TensorGroup(**{k: T(v) for k, v in inp['camera'].items()}))
    real = tdgp.generator.TensorGroup(img=torch.randn(B, 3, cfg.img_resolution, cfg.img_resolution, device=DEV), c=c0,
                                      depth=torch.zeros(B, 1, cfg.img_resolution, cfg.img_resolution, device=DEV))
    for flag in (True, False):
        loss = TR.StyleGAN2Loss(G, D, DEV, r1_gamma=0.0, use_depth=use_depth, learn_camera_dist=flag, camera_reg=TR.CameraRegConfig.disabled() if flag else None,
                                synthesis_kwargs=dict(u_coarse=T(inp['u_coarse']), u_fine=T(inp['u_fine'])))
        G.zero_grad(set_to_none=True)
        G.requires_grad_(True)
        D.requires_grad_(False)
        loss.accumulate_gradients('Gmain', real, gen, gain=1, cur_nimg=0)
        grads = [p.grad for p in G.synthesis.camera_adaptor.parameters()]
        if flag:
            assert all(x is not None and torch.isfinite(x).all() for x in grads) and sum(float(x.abs().sum()) for x in grads) > 0
            grads_adv = [x.clone() for x in grads]
        else:
            assert all(x is None or float(x.abs().sum()) == 0.0 for x in grads)
    # with the regularisers of loss.py:142-238 on top (Lipschitz through bias_act's second-order kernel, EMD, force-mean): they add to
    # the adaptor's gradients and to nothing else
    base = [x.clone() for x in G.synthesis.camera_adaptor.parameters()]
    reg = TR.CameraRegConfig(prior=tdgp.metrics.camera_base(), lipschitz_enabled=True, emd_anneal_kimg=1, lipschitz_num_samples=32, force_mean_num_samples=32)
    loss = TR.StyleGAN2Loss(G, D, DEV, r1_gamma=0.0, use_depth=use_depth, learn_camera_dist=True, camera_reg=reg,
                            synthesis_kwargs=dict(u_coarse=T(inp['u_coarse']), u_fine=T(inp['u_fine'])))
    loss.progressive_update(2)
    assert loss.emd_multiplier == 1.0
    G.zero_grad(set_to_none=True)
    G.requires_grad_(True)
    loss.accumulate_gradients('Gmain', real, gen, gain=1, cur_nimg=0)
    grads_reg = [p.grad for p in G.synthesis.camera_adaptor.parameters()]
    assert all(torch.isfinite(x).all() for x in grads_reg)
    assert any(float((a - b).abs().max()) > 0 for a, b in zip(grads_reg, grads_adv))
    for k in ('Loss/camera_dist/emd_loss', 'Loss/camera_dist/force_mean', 'Dist_lipschitz_reg/yaw', 'Dist_emd_reg/fov'):
        assert k in loss.stats and torch.isfinite(loss.stats[k]).all(), k
    assert all(torch.equal(a, b) for a, b in zip(base, G.synthesis.camera_adaptor.parameters()))
    plain = _gen(tdgp, tdgp.config.config_tiny(), 1)
    with pytest.raises(RuntimeError):
        TR.StyleGAN2Loss(plain, D, DEV, learn_camera_dist=True)


def test_conv_transpose2d_x2(tdgp):
    """tdgp_conv_transpose2d_x2 (polyphase transposed conv without the FIR pass) = the adjoint of the 3x3 stride-2 convolution: against
    torch's conv_transpose2d, and <conv2d(a, w, s2), g> == <a, convT(g, w)> with the library's own strided convolution."""
    rs = np.random.RandomState(12)
    cg = tdgp.ops.conv2d_gradfix
    for B, O, I, h, w in ((2, 20, 12, 7, 9), (1, 130, 70, 16, 16), (3, 8, 64, 4, 4)):
        x = T(rs.randn(B, O, h, w))
        W = T(rs.randn(O, I, 3, 3))
        y = cg.conv_transpose2d_x2(x, W)
        ref = torch.nn.functional.conv_transpose2d(x.double(), W.double(), stride=2).float()
        assert y.shape == ref.shape == (B, I, 2 * h + 1, 2 * w + 1)
        assert_close(N(y), N(ref), 2e-5, 'conv_transpose2d_x2', 1.0)
        a = T(rs.randn(B, I, 2 * h + 1, 2 * w + 1))
        lhs = float((cg.conv2d_strided(a, W, stride=2, padding=0).double() * x.double()).sum())
        rhs = float((a.double() * y.double()).sum())
        assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), 1.0)


def test_c4_forward_properties(tdgp):
    """BASELINE configs[3] (cmax 1024 / cbase 65536: 1024-channel layers, twice the K depth of the benchmark configuration): the
    forward runs, is finite and bit-reproducible, and the backbone agrees between its two output layouts."""
    cfg = tdgp.config.config_c4()
    G = _gen(tdgp, cfg, 3)
    inp = tdgp.weights.synthetic_inputs(cfg, batch=1, seed=4)
    cam = {k: T(v) for k, v in inp['camera'].items()}
    kw = dict(noise_mode='const', u_coarse=T(inp['u_coarse']), u_fine=T(inp['u_fine']))
    a = G(T(inp['z']), T(inp['c']), cam, **kw)
    b = G(T(inp['z']), T(inp['c']), cam, **kw)
    assert a.shape == (1, 3, 256, 256) and bool(torch.isfinite(a).all()) and torch.equal(a, b)
    ws = G.mapping(T(inp['z']), T(inp['c']))
    dec = G.synthesis.tri_plane_decoder
    nchw = dec(ws, noise_mode='const')
    hwc = dec(ws, noise_mode='const', hwc=True).t
    assert_close(N(hwc.permute(0, 1, 4, 2, 3).reshape(nchw.shape)), N(nchw), 5e-6, 'planes channel-last vs NCHW (c4)', 1.0)
    # Regression: with an odd number of K iterations per split-K slice (here the 512 -> 256 x2 layer: 128 iterations in 3 slices) the
    # last multiply was not fenced from the epilogue's LDS tiles, and roughly one run in ten came out with a wave tile of garbage.
    for _ in range(40):
        assert torch.equal(dec(ws, noise_mode='const'), nchw)


@pytest.mark.parametrize('B,cin,cout,H', [(8, 64, 64, 64), (4, 128, 96, 96), (3, 48, 130, 128)])
def test_upconv_split_arith(tdgp, oracle, B, cin, cout, H):
    """The x2 (transposed conv + FIR) layers under tdgp_set_conv_arith(1): bf16 x 3 split operands on the bf16 MFMA, same Z layout and
    FIR / output stage as the fp32 kernel.  Against the double-accumulating oracle, next to the fp32-MFMA result of the same layer."""
    rs = np.random.RandomState(cin + 3 * cout)
    mc = tdgp.ops.modconv
    x = rs.randn(B, cin, H, H).astype(np.float32)
    w = rs.randn(cout, cin, 3, 3).astype(np.float32)
    s = (1 + 0.5 * rs.randn(B, cin)).astype(np.float32)
    bias = rs.randn(cout).astype(np.float32)
    noise = (0.3 * rs.randn(2 * H, 2 * H)).astype(np.float32)
    f = oracle.setup_filter([1, 3, 3, 1])
    ref = oracle.bias_act(oracle.modulated_conv2d(x, w, s, noise=noise, up=2, resample_filter=f), bias, act='lrelu')
    pk = mc.PackedConv(T(w))
    kw = dict(noise=T(noise), bias=T(bias), demodulate=True, act='lrelu', up=2, fir=mc.fir_host_array(f))
    fold_was, mc.FOLD_UP2 = mc.FOLD_UP2, False             # this test is about the transposed-convolution kernels (fp32 and split), not the folded F(4x4) form
    prev = 0
    try:
        y32 = mc.modconv_forward(T(x), pk, T(s), **kw)
        prev = tdgp._lib.set_conv_arith(1)
        ysp = mc.modconv_forward(T(x), pk, T(s), **kw)
        for _ in range(5):
            assert torch.equal(mc.modconv_forward(T(x), pk, T(s), **kw), ysp)
    finally:
        tdgp._lib.set_conv_arith(prev)
        mc.FOLD_UP2 = fold_was
    scale = np.abs(ref).max()
    e32, esp = np.abs(N(y32) - ref).max() / scale, np.abs(N(ysp) - ref).max() / scale
    assert e32 < 2e-6 and esp < 4e-6, (e32, esp)
    assert not torch.equal(y32, ysp)                        # different arithmetic: the split kernel ran
    assert tdgp._lib.set_conv_arith(0) == 0


@pytest.mark.parametrize('cin,cout,H,up', [(36, 96, 32, 2), (20, 160, 64, 2), (44, 64, 32, 1), (12, 130, 64, 1)])
def test_conv_odd_iteration_count_is_deterministic(tdgp, oracle, cin, cout, H, up):
    """Odd K-iteration counts (Cin / 4 odd) leave the last multiply outside the double-buffered loop: it must still be fenced from
    the epilogue, which overlays the stage buffers -- correct against the oracle and bit-identical over repeated launches."""
    rs = np.random.RandomState(cin * 7 + cout)
    mc = tdgp.ops.modconv
    B = 2
    x = rs.randn(B, cin, H, H).astype(np.float32)
    w = rs.randn(cout, cin, 3, 3).astype(np.float32)
    s = (1 + 0.5 * rs.randn(B, cin)).astype(np.float32)
    bias = rs.randn(cout).astype(np.float32)
    f = oracle.setup_filter([1, 3, 3, 1])
    ref = oracle.bias_act(oracle.modulated_conv2d(x, w, s, up=up, resample_filter=f if up == 2 else None), bias, act='lrelu')
    pk = mc.PackedConv(T(w))
    kw = dict(bias=T(bias), demodulate=True, act='lrelu', up=up, fir=mc.fir_host_array(f) if up == 2 else None)
    xs, ss = T(x), T(s)
    y0 = mc.modconv_forward(xs, pk, ss, **kw)
    assert_close(N(y0), ref, 1e-5, f'modconv up{up} odd iteration count', 1.0)
    for _ in range(60):
        assert torch.equal(mc.modconv_forward(xs, pk, ss, **kw), y0)


@pytest.mark.parametrize('B,cin,cout,H', [(8, 64, 64, 128), (8, 128, 96, 64), (4, 48, 130, 128)])
def test_conv_split_arith(tdgp, oracle, B, cin, cout, H):
    """Opt-in arithmetic (tdgp_set_conv_arith(1)): fp32 operands split into three bf16 pieces, six piece products per multiply on the
    bf16 MFMA, fp32 accumulation.  Same layer through both modes against the double-accumulating oracle: the split path must be as
    close to the exact result as the fp32 MFMA path (it keeps all 24 mantissa bits; dropped terms <= 3 * 2^-24 per product)."""
    rs = np.random.RandomState(cin + cout)
    mc = tdgp.ops.modconv
    x = rs.randn(B, cin, H, H).astype(np.float32)
    w = rs.randn(cout, cin, 3, 3).astype(np.float32)
    s = (1 + 0.5 * rs.randn(B, cin)).astype(np.float32)
    bias = rs.randn(cout).astype(np.float32)
    noise = (0.3 * rs.randn(H, H)).astype(np.float32)
    ref = oracle.bias_act(oracle.modulated_conv2d(x, w, s, noise=noise), bias, act='lrelu')
    pk = mc.PackedConv(T(w))
    kw = dict(noise=T(noise), bias=T(bias), demodulate=True, act='lrelu')
    y32 = mc.modconv_forward(T(x), pk, T(s), **kw)
    prev = tdgp._lib.set_conv_arith(1)
    try:
        tdgp._lib.profile_enable(True)
        ysp = mc.modconv_forward(T(x), pk, T(s), **kw)
        torch.cuda.synchronize()
        tdgp._lib.profile_report()
        tdgp._lib.profile_enable(False)
    finally:
        tdgp._lib.set_conv_arith(prev)
    scale = np.abs(ref).max()
    e32, esp = np.abs(N(y32) - ref).max() / scale, np.abs(N(ysp) - ref).max() / scale
    assert e32 < 2e-6 and esp < 4e-6, (e32, esp)
    assert not torch.equal(y32, ysp) or cin < 16            # the two modes are different arithmetic (this also proves the split kernel ran)
    assert tdgp._lib.set_conv_arith(0) == 0


# ------------------------------------------------------------------------------------------------ round 6: decoders outside the fused form, camera-conditioned mapping
_MLP_VARIANTS = dict(n3=dict(F=8, hid=16, n=3, view=False, marcher='classical'), n4mip=dict(F=8, hid=16, n=4, view=False, marcher='mip'),
                     view=dict(F=8, hid=3, n=2, view=True, marcher='classical'), odd=dict(F=12, hid=20, n=2, view=False, marcher='mip'))


def _variant_mlp(tdgp, g, tag):
    v = _MLP_VARIANTS[tag]
    mlp = tdgp.renderer.TriPlaneMLP(v['F'], v['hid'], out_dim=3, ray_marcher_type=v['marcher'], n_layers=v['n'], has_view_cond=v['view'])
    mlp.load_state_dict({f'model.{i}.{kind}': T(g[f'{tag}_{kind[0]}{i}']) for i in range(v['n']) for kind in ('weight', 'bias')}, strict=True)
    return mlp.to(DEV)


@pytest.mark.parametrize('tag', sorted(_MLP_VARIANTS))
def test_mlp_variants_vs_reference_golden(tdgp, tag):
    """VERDICT r05 missing #2: TriPlaneMLP with n_layers != 2 / has_view_cond / widths outside the fused kernel's table (networks_epigraf.py:35-43) no
    longer raises: the lookup runs in tdgp_triplane_features, the layers as eager tensor ops -- against the reference's simple_tri_plane_renderer."""
    g = load_golden('mlp_variants')
    mlp = _variant_mlp(tdgp, g, tag)
    assert not tdgp.renderer.fused_form(mlp)
    out = tdgp.renderer.simple_tri_plane_renderer(T(g[f'{tag}_planes']), T(g['coords']), mlp, scale=0.5)
    for key in ('rgb', 'sigma'):
        ref = g[f'{tag}_{key}']
        assert_close(N(out[key]), ref, 5e-6, f'{tag} {key} (eager decoder)', max(1.0, float(np.abs(ref).max())))


def test_mlp_identity_decoder_vs_oracle(tdgp, oracle):
    """tri_plane.mlp.n_layers == 0 (nn.Identity, feat_dim = out_dim + 1: the planes carry rgb + sigma).  The reference's own forward raises AttributeError
    on this branch (`backbone_out_dim` is only set on the other one), so there is no golden: the lookup is held to the oracle's (plane-mean features), and
    'mip' applies the sigmoid to the first three."""
    rs = np.random.RandomState(5)
    planes, coords = rs.randn(2, 12, 16, 16).astype(np.float32), ((rs.rand(2, 150, 3) * 2 - 1) * 0.6).astype(np.float32)
    feats = oracle.triplane_field(planes, coords, np.zeros((16, 4), np.float32), np.zeros(16, np.float32), np.zeros((4, 16), np.float32), np.zeros(4, np.float32),
                                  0.5, return_feats=True)['feats']
    for marcher in ('classical', 'mip'):
        mlp = tdgp.renderer.TriPlaneMLP(4, 8, out_dim=3, ray_marcher_type=marcher, n_layers=0).to(DEV)
        assert len(list(mlp.parameters())) == 0 and not tdgp.renderer.fused_form(mlp)
        out = tdgp.renderer.simple_tri_plane_renderer(T(planes), T(coords), mlp, scale=0.5)
        rgb = feats[..., :3] if marcher == 'classical' else (1.0 / (1.0 + np.exp(-feats[..., :3].astype(np.float64)))).astype(np.float32) * np.float32(1.002) - np.float32(0.001)
        assert_close(N(out['rgb']), rgb, 2e-6, f'identity decoder rgb ({marcher})', 1.0)
        np.testing.assert_array_equal(N(out['sigma']), feats[..., 3:4])
    np.testing.assert_array_equal(N(tdgp.renderer.triplane_features(T(planes), T(coords), 0.5)), feats)        # the lookup itself: torch's evaluation order, bit for bit


def test_renderer_with_three_layer_decoder_vs_reference_golden(tdgp):
    """ImportanceRenderer.forward with the 3-layer decoder: the staged chain (stratified samples, eager field, importance samples, eager field, merge +
    march) against the reference's renderer on the same rays and draws."""
    g = load_golden('mlp_variants')
    mlp = _variant_mlp(tdgp, g, 'n3')
    S = g['r_u_coarse'].shape[2]
    opts = dict(box_size=1.0, num_proposal_steps=S, num_fine_steps=S, clamp_mode='softplus', use_inf_depth=True, ray_start=0.75, ray_end=1.25,
                u_coarse=T(g['r_u_coarse']), u_fine=T(g['r_u_fine']))
    rend = tdgp.renderer.ImportanceRenderer('classical')
    rgb, depth, _w, _t = rend(T(g['n3_planes']), mlp, T(g['r_ray_o']), T(g['r_ray_d']), opts)
    assert_close(N(rgb), g['r_rgb'], 2e-5, 'renderer rgb, 3-layer decoder', max(1.0, float(np.abs(g['r_rgb']).max())))
    assert_close(N(depth), g['r_depth'], 2e-5, 'renderer depth, 3-layer decoder', 1.0)
    # a whole Generator with such a decoder builds, loads its state dict and renders
    cfg = tdgp.config.config_tiny()
    cfg.mlp_n_layers = 3
    G = _gen(tdgp, cfg, 61)
    inp = tdgp.weights.synthetic_inputs(cfg, batch=2, seed=62)
    img = G(T(inp['z']), T(inp['c']), {k: T(v) for k, v in inp['camera'].items()}, noise_mode='const', u_coarse=T(inp['u_coarse']), u_fine=T(inp['u_fine']))
    assert img.shape == (2, 3, cfg.img_resolution, cfg.img_resolution) and torch.isfinite(img).all()


@pytest.mark.parametrize('tag', ['four', 'raw'])
def test_mapping_camera_cond_vs_reference_golden(tdgp, tag):
    """VERDICT r05 missing #3: MappingNetwork(camera_cond=True) (layers.py:84-93,127-138) -- Fourier-encoded / raw yaw and pitch appended to the label,
    explicit angles (beyond +-2 pi: the wrap), the eval-time stand-in mean_camera_params, truncation -- against the reference's ws."""
    g = load_golden('mapping_cam')
    sd = {k.split('::', 1)[1]: T(v) for k, v in g.items() if k.startswith(tag + '::')}
    c_dim = g[f'{tag}_c'].shape[1]
    m = tdgp.generator.MappingNetwork(z_dim=16, c_dim=c_dim, w_dim=24, num_ws=5, num_layers=2, camera_cond=True, camera_raw_scalars=(tag == 'raw'),
                                      mean_camera_params=np.asarray(g[f'{tag}::mean_camera_params'])).eval()
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV)
    z, c, ang = T(g[f'{tag}_z']), (T(g[f'{tag}_c']) if c_dim > 0 else None), T(g[f'{tag}_angles'])
    with torch.no_grad():
        for got, key in ((m(z, c, camera_angles=ang), 'ws'), (m(z, c), 'ws_mean'), (m(z, c, camera_angles=ang, truncation_psi=0.6), 'ws_psi06')):
            ref = g[f'{tag}_{key}']
            assert_close(N(got), ref, 1e-5, f'camera-conditioned mapping {tag} {key}', max(1.0, float(np.abs(ref).max())))
    # through Generator: the configuration carries the option, the state-dict spec its tensors
    cfg = tdgp.config.config_tiny()
    cfg.camera_cond, cfg.camera_raw_scalars, cfg.mean_camera_params = True, (tag == 'raw'), (0.2, 1.5, 0.0)
    G = _gen(tdgp, cfg, 63)
    inp = tdgp.weights.synthetic_inputs(cfg, batch=2, seed=64)
    cam = {k: T(v) for k, v in inp['camera'].items()}
    kw = dict(noise_mode='const', u_coarse=T(inp['u_coarse']), u_fine=T(inp['u_fine']))
    a = G(T(inp['z']), T(inp['c']), cam, camera_angles_cond=cam['angles'], **kw)
    b = G(T(inp['z']), T(inp['c']), cam, **kw)                                  # eval without angles: mean_camera_params
    assert torch.isfinite(a).all() and not torch.equal(a, b)
