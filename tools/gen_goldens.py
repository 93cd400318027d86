#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the REFERENCE (build container only).

The reference ships no tests or golden vectors for the generator-forward path,
so parity is pinned by (inputs -> outputs) pairs captured here from the reference's own CPU/PyTorch
path: `/root/reference/src` is imported with two stub modules (`omegaconf`, `torchvision`, which the
G-forward import chain needs only nominally) and its functions are called on seeded inputs.
Only the resulting arrays are committed -- data, never reference source.

Random tensors the reference draws internally (torch.rand_like / torch.rand, tri_plane_renderer.py:225,279)
are replaced by explicit seeded arrays through a monkey-patch so that they can be fed as *inputs* to the
oracle and to the HIP kernels.  Integer intermediates the reference never returns (searchsorted indices,
sort permutation) are captured by wrapping torch.searchsorted / torch.sort.

Run:  python tools/gen_goldens.py        (needs /root/reference; writes tests/golden/)
"""
import importlib
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('TDGP_REFERENCE', '/root/reference')
OUT = os.path.join(REPO, 'tests', 'golden')


def _import_reference():
    om = types.ModuleType('omegaconf')

    class DictConfig(dict):
        pass
    om.DictConfig = DictConfig
    om.OmegaConf = object
    sys.modules.setdefault('omegaconf', om)
    tv = types.ModuleType('torchvision')
    tv.__path__ = []                                 # a package, so `import torchvision.transforms.functional` resolves to the stubs below
    sys.modules.setdefault('torchvision', tv)
    for sub in ('torchvision.transforms', 'torchvision.transforms.functional', 'torchvision.utils', 'torchvision.io'):
        m = types.ModuleType(sub)
        m.__path__ = []
        sys.modules.setdefault(sub, m)
    try:
        import PIL  # noqa: F401
    except ImportError:                              # inference_utils.py imports PIL for its (unused here) grid-drawing helpers
        pil = types.ModuleType('PIL')
        pil.__path__ = []
        for name in ('Image', 'ImageDraw', 'ImageFont'):
            setattr(pil, name, types.ModuleType('PIL.' + name))
            sys.modules.setdefault('PIL.' + name, getattr(pil, name))
        sys.modules.setdefault('PIL', pil)
    sys.path.insert(0, REF)


_import_reference()
import torch  # noqa: E402
from src.dnnlib import EasyDict, TensorGroup  # noqa: E402
from src.torch_utils.ops import bias_act as ref_bias_act  # noqa: E402
from src.torch_utils.ops import upfirdn2d as ref_upfirdn2d  # noqa: E402
from src.training import networks_stylegan2 as ref_sg2  # noqa: E402
from src.training import tri_plane_renderer as ref_tpr  # noqa: E402
from src.training import rendering_utils as ref_ru  # noqa: E402
from src.training import layers as ref_layers  # noqa: E402
from src.training.networks_epigraf import Generator, TriPlaneMLP  # noqa: E402

sys.path.insert(0, REPO)
tdgp = importlib.import_module('3dgp_amd')

T = torch.from_numpy


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrays)
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays')


class PatchedRNG:
    """Feed explicit tensors to the reference's torch.rand_like / torch.rand draws (in call order)."""

    def __init__(self, rand_like=(), rand=(), randn_like=()):
        self.q_like, self.q, self.q_nlike = list(rand_like), list(rand), list(randn_like)

    def __enter__(self):
        self.o_like, self.o, self.o_nlike = torch.rand_like, torch.rand, torch.randn_like

        def randn_like(x, *a, **k):
            v = self.q_nlike.pop(0)
            assert tuple(v.shape) == tuple(x.shape), (v.shape, x.shape)
            return v.clone()
        torch.randn_like = randn_like

        def rand_like(x, *a, **k):
            v = self.q_like.pop(0)
            assert tuple(v.shape) == tuple(x.shape), (v.shape, x.shape)
            return v.clone()

        def rand(*size, **k):
            v = self.q.pop(0)
            size = tuple(size[0]) if len(size) == 1 and not isinstance(size[0], int) else tuple(size)
            assert tuple(v.shape) == size, (v.shape, size)
            return v.clone()
        torch.rand_like, torch.rand = rand_like, rand
        return self

    def __exit__(self, *exc):
        torch.rand_like, torch.rand, torch.randn_like = self.o_like, self.o, self.o_nlike
        assert not self.q_like and not self.q and not self.q_nlike, 'unused RNG tensors'


class Capture:
    """Record the outputs of torch.searchsorted / torch.sort while the reference runs."""

    def __enter__(self):
        self.inds, self.perm, self.cdf = [], [], []
        self.o_ss, self.o_sort = torch.searchsorted, torch.sort

        def ss(*a, **k):
            r = self.o_ss(*a, **k)
            self.inds.append(r.clone())
            self.cdf.append(a[0].clone())                # the knots the draws were ranked against (tri_plane_renderer.py:282)
            return r

        def sort(*a, **k):
            r = self.o_sort(*a, **k)
            self.perm.append(r[1].clone())
            return r
        torch.searchsorted, torch.sort = ss, sort
        return self

    def __exit__(self, *exc):
        torch.searchsorted, torch.sort = self.o_ss, self.o_sort


# ------------------------------------------------------------------------------------------ op level

def gen_bias_act():
    g = np.random.RandomState(1)
    arrays = {}
    x = (g.randn(2, 5, 4, 3) * 3).astype(np.float32)
    x.flat[:6] = [0.0, -0.0, 25.0, -25.0, 90.0, -90.0]
    b = g.randn(5).astype(np.float32)
    arrays['x'], arrays['b'] = x, b
    for act in ref_bias_act.activation_funcs:
        arrays[f'y_{act}'] = npy(ref_bias_act.bias_act(T(x), T(b), act=act, impl='ref'))
    arrays['y_lrelu_gain1_clamp'] = npy(ref_bias_act.bias_act(T(x), T(b), act='lrelu', gain=1.0, clamp=0.5, impl='ref'))
    arrays['y_lrelu_alpha'] = npy(ref_bias_act.bias_act(T(x), T(b), act='lrelu', alpha=0.01, gain=2.0, impl='ref'))
    arrays['y_linear_nobias'] = npy(ref_bias_act.bias_act(T(x), None, act='linear', gain=0.5, impl='ref'))
    b3 = g.randn(3).astype(np.float32)
    arrays['b_dim3'] = b3
    arrays['y_relu_dim3'] = npy(ref_bias_act.bias_act(T(x), T(b3), dim=3, act='relu', impl='ref'))
    x2 = g.randn(7, 5).astype(np.float32)
    arrays['x2'] = x2
    arrays['y2_lrelu'] = npy(ref_bias_act.bias_act(T(x2), T(b), act='lrelu', impl='ref'))
    xcl = T(x).contiguous(memory_format=torch.channels_last)
    arrays['y_swish_channels_last'] = npy(ref_bias_act.bias_act(xcl, T(b), act='swish', clamp=2.0, impl='ref').contiguous())
    save('bias_act', **arrays)


def gen_bias_act_grad():
    """First / second derivative of the reference's CPU bias_act (autograd through `_bias_act_ref`, bias_act.py:91-120): the values
    its CUDA plugin returns for grad = 1 / 2 (bias_act.cu:60-147) -- dx = dy * d act / dx, d_x = d2 * dy * d2 act / dx2."""
    g = np.random.RandomState(81)
    x = (g.randn(2, 5, 4, 3) * 1.5).astype(np.float32)
    b = g.randn(5).astype(np.float32)
    dy = g.randn(*x.shape).astype(np.float32)
    d2 = g.randn(*x.shape).astype(np.float32)
    arrays = dict(x=x, b=b, dy=dy, d2=d2)
    for act in ref_bias_act.activation_funcs:
        for tag, kw in (('', {}), ('_clamp', dict(clamp=0.8, gain=1.3))):
            xt = T(x).clone().requires_grad_(True)
            dyt = T(dy).clone().requires_grad_(True)
            y = ref_bias_act.bias_act(xt, T(b), act=act, impl='ref', **kw)
            dx, = torch.autograd.grad(y, xt, dyt, create_graph=True)
            ddx = torch.autograd.grad(dx, xt, T(d2), allow_unused=True, retain_graph=True)[0] if dx.requires_grad else None
            ddy = torch.autograd.grad(dx, dyt, T(d2), allow_unused=True)[0] if dx.requires_grad else None
            arrays[f'y_{act}{tag}'] = npy(y)
            arrays[f'dx_{act}{tag}'] = npy(dx)
            arrays[f'ddx_{act}{tag}'] = np.zeros_like(x) if ddx is None else npy(ddx)
            arrays[f'ddy_{act}{tag}'] = np.zeros_like(x) if ddy is None else npy(ddy)
    save('bias_act_grad', **arrays)


def gen_upfirdn2d():
    g = np.random.RandomState(2)
    arrays = {}
    f = ref_upfirdn2d.setup_filter([1, 3, 3, 1])
    arrays['f1331'] = npy(f)
    arrays['f1331_flip_gain'] = npy(ref_upfirdn2d.setup_filter([1, 2, 3, 4], flip_filter=True, gain=2.0))
    x = g.randn(2, 3, 9, 9).astype(np.float32)
    arrays['x_f1'] = x                                  # F1: post-conv_transpose FIR
    arrays['y_f1'] = npy(ref_upfirdn2d.upfirdn2d(T(x), f, padding=[1, 1, 1, 1], gain=4, impl='ref'))
    x = g.randn(2, 3, 4, 4).astype(np.float32)
    arrays['x_f2'] = x                                  # F2: img upsample
    arrays['y_f2'] = npy(ref_upfirdn2d.upsample2d(T(x), f, impl='ref'))
    x = g.randn(1, 2, 7, 5).astype(np.float32)
    arrays['x_odd'] = x
    arrays['y_filter2d'] = npy(ref_upfirdn2d.filter2d(T(x), f, impl='ref'))
    arrays['y_downsample2d'] = npy(ref_upfirdn2d.downsample2d(T(x), f, impl='ref'))
    arrays['y_negpad'] = npy(ref_upfirdn2d.upfirdn2d(T(x), f, padding=[-1, 2, 3, -1], impl='ref'))
    fa = ref_upfirdn2d.setup_filter([1, 2, 3, 4], normalize=True)
    arrays['f_asym'] = npy(fa)
    arrays['y_asym_noflip'] = npy(ref_upfirdn2d.upfirdn2d(T(x), fa, padding=2, impl='ref'))
    arrays['y_asym_flip'] = npy(ref_upfirdn2d.upfirdn2d(T(x), fa, padding=2, flip_filter=True, impl='ref'))
    fr = torch.from_numpy(g.randn(3, 5).astype(np.float32))    # non-square filter, up 3 / down 2
    arrays['f_rect'] = npy(fr)
    arrays['y_rect_up3_down2'] = npy(ref_upfirdn2d.upfirdn2d(T(x), fr, up=[3, 2], down=[2, 1], padding=[2, 1, 0, 3], gain=1.5, impl='ref'))
    arrays['y_identity'] = npy(ref_upfirdn2d.upfirdn2d(T(x), None, impl='ref'))
    x = g.randn(1, 2, 33, 33).astype(np.float32)
    arrays['x_f1_33'] = x
    arrays['y_f1_33'] = npy(ref_upfirdn2d.upfirdn2d(T(x), f, padding=[1, 1, 1, 1], gain=4, impl='ref'))
    save('upfirdn2d', **arrays)


def gen_upfirdn2d_grad():
    """Input gradient of upfirdn2d (autograd through `_upfirdn2d_ref`): what Upfirdn2dCuda.backward computes as another upfirdn2d
    with up <-> down swapped, the flipped filter and the padding of upfirdn2d.py:251-265."""
    g = np.random.RandomState(91)
    arrays = {}
    f = ref_upfirdn2d.setup_filter([1, 3, 3, 1])
    fa = ref_upfirdn2d.setup_filter([1, 2, 3, 4])
    cases = dict(up2=dict(f=f, up=2, down=1, padding=[2, 1, 2, 1], gain=4.0), fir=dict(f=f, up=1, down=1, padding=[1, 1, 1, 1], gain=4.0),
                 down2=dict(f=f, up=1, down=2, padding=[1, 1, 1, 1], gain=1.0), asym=dict(f=fa, up=[3, 2], down=[2, 1], padding=[2, 1, 0, 3], gain=1.5, flip_filter=True))
    for name, kw in cases.items():
        x = T(g.randn(2, 3, 8, 6).astype(np.float32)).requires_grad_(True)
        y = ref_upfirdn2d.upfirdn2d(x, impl='ref', **kw)
        dy = T(g.randn(*y.shape).astype(np.float32))
        dx, = torch.autograd.grad(y, x, dy)
        arrays.update({f'{name}_dy': npy(dy), f'{name}_dx': npy(dx), f'{name}_f': npy(kw['f'])})
    save('upfirdn2d_grad', **arrays)


CONV_GRAD_CASES = dict(k3=dict(B=2, cin=10, cout=12, H=9, W=14, k=3, stride=1, pad=1), k1=dict(B=3, cin=7, cout=5, H=8, W=8, k=1, stride=1, pad=0),
                       k5=dict(B=1, cin=4, cout=6, H=12, W=10, k=5, stride=1, pad=2), k3s2=dict(B=2, cin=6, cout=8, H=12, W=16, k=3, stride=2, pad=1),
                       k3p0=dict(B=2, cin=5, cout=4, H=10, W=9, k=3, stride=1, pad=0))


def gen_conv2d_grad():
    """conv2d_gradfix.conv2d (conv2d_gradfix.py:34-39; on CPU tensors the reference's own fallback to torch conv2d) under autograd:
    input / weight / bias gradients for a given dy."""
    from src.torch_utils.ops import conv2d_gradfix as ref_cg
    from src.torch_utils.ops import fma as ref_fma
    arrays = {}
    for name, c in CONV_GRAD_CASES.items():
        g = np.random.RandomState(sum(map(ord, name)) + 7)
        x = T(g.randn(c['B'], c['cin'], c['H'], c['W']).astype(np.float32)).requires_grad_(True)
        w = T(g.randn(c['cout'], c['cin'], c['k'], c['k']).astype(np.float32)).requires_grad_(True)
        b = T(g.randn(c['cout']).astype(np.float32)).requires_grad_(True)
        y = ref_cg.conv2d(x, w, b, stride=c['stride'], padding=c['pad'])
        dy = T(g.randn(*y.shape).astype(np.float32))
        dx, dw, db = torch.autograd.grad(y, [x, w, b], dy)
        arrays.update({f'{name}_x': npy(x), f'{name}_w': npy(w), f'{name}_b': npy(b), f'{name}_y': npy(y), f'{name}_dy': npy(dy), f'{name}_dx': npy(dx),
                       f'{name}_dw': npy(dw), f'{name}_db': npy(db)})
    # fma (fma.py): broadcast shapes of the unfused modulated conv, x * dcoefs + noise
    g = np.random.RandomState(19)
    a = T(g.randn(2, 6, 5, 5).astype(np.float32)).requires_grad_(True)
    bb = T(g.randn(2, 6, 1, 1).astype(np.float32)).requires_grad_(True)
    cc = T(g.randn(1, 1, 5, 5).astype(np.float32)).requires_grad_(True)
    out = ref_fma.fma(a, bb, cc)
    dout = T(g.randn(*out.shape).astype(np.float32))
    da, dbb, dcc = torch.autograd.grad(out, [a, bb, cc], dout)
    arrays.update(fma_a=npy(a), fma_b=npy(bb), fma_c=npy(cc), fma_out=npy(out), fma_dout=npy(dout), fma_da=npy(da), fma_db=npy(dbb), fma_dc=npy(dcc))
    save('conv2d_grad', **arrays)


def gen_modconv_grad():
    """Autograd through the reference's unfused modulated_conv2d (networks_stylegan2.py:60-80, the training path): gradients w.r.t.
    x, weight and styles for the stride-1 forms of the synthesis layers (3x3 demodulated, 1x1 ToRGB)."""
    arrays = {}
    for tag, (B, cin, cout, H, W, k, demod) in dict(c3=(2, 12, 10, 9, 12, 3, True), rgb=(3, 16, 6, 8, 8, 1, False), c3big=(2, 40, 36, 8, 8, 3, True)).items():
        g = np.random.RandomState(70 + cin)
        x = T(g.randn(B, cin, H, W).astype(np.float32)).requires_grad_(True)
        w = T(g.randn(cout, cin, k, k).astype(np.float32)).requires_grad_(True)
        s = T((1 + 0.5 * g.randn(B, cin)).astype(np.float32)).requires_grad_(True)
        y = ref_sg2.modulated_conv2d(x=x, weight=w, styles=s, padding=k // 2, demodulate=demod, fused_modconv=False)
        dy = T(g.randn(*y.shape).astype(np.float32))
        dx, dw, ds = torch.autograd.grad(y, [x, w, s], dy)
        arrays.update({f'{tag}_x': npy(x), f'{tag}_w': npy(w), f'{tag}_s': npy(s), f'{tag}_y': npy(y), f'{tag}_dy': npy(dy), f'{tag}_dx': npy(dx),
                       f'{tag}_dw': npy(dw), f'{tag}_ds': npy(ds)})
    # x2-upsampling layer (transposed conv + FIR), SynthesisLayer.forward arguments (networks_stylegan2.py:137-139)
    f = ref_upfirdn2d.setup_filter([1, 3, 3, 1])
    for tag, (B, cin, cout, H, W) in dict(up=(2, 12, 10, 6, 9), upbig=(1, 40, 36, 8, 8)).items():
        g = np.random.RandomState(90 + cin)
        x = T(g.randn(B, cin, H, W).astype(np.float32)).requires_grad_(True)
        w = T(g.randn(cout, cin, 3, 3).astype(np.float32)).requires_grad_(True)
        s = T((1 + 0.5 * g.randn(B, cin)).astype(np.float32)).requires_grad_(True)
        y = ref_sg2.modulated_conv2d(x=x, weight=w, styles=s, up=2, padding=1, resample_filter=f, flip_weight=False, fused_modconv=False)
        dy = T(g.randn(*y.shape).astype(np.float32))
        dx, dw, ds = torch.autograd.grad(y, [x, w, s], dy)
        arrays.update({f'{tag}_x': npy(x), f'{tag}_w': npy(w), f'{tag}_s': npy(s), f'{tag}_y': npy(y), f'{tag}_dy': npy(dy), f'{tag}_dx': npy(dx),
                       f'{tag}_dw': npy(dw), f'{tag}_ds': npy(ds)})
    # plain strided convolutions (forward): the forms of tdgp_conv2d
    for tag, (B, cin, cout, H, W, k, st, pad) in dict(s2=(2, 10, 7, 13, 17, 3, 2, 0), s2p1=(2, 6, 9, 12, 16, 3, 2, 1), s1p0=(1, 5, 4, 9, 10, 3, 1, 0),
                                                      k1s2=(2, 8, 6, 8, 8, 1, 2, 0)).items():
        g = np.random.RandomState(95 + cin)
        x, w, b = g.randn(B, cin, H, W).astype(np.float32), g.randn(cout, cin, k, k).astype(np.float32), g.randn(cout).astype(np.float32)
        from src.torch_utils.ops import conv2d_gradfix as ref_cg
        arrays.update({f'conv_{tag}_x': x, f'conv_{tag}_w': w, f'conv_{tag}_b': b, f'conv_{tag}_y': npy(ref_cg.conv2d(T(x), T(w), T(b), stride=st, padding=pad))})
    save('modconv_grad', **arrays)


def gen_modconv():
    g = np.random.RandomState(3)
    arrays = {}
    f = ref_upfirdn2d.setup_filter([1, 3, 3, 1])
    arrays['f'] = npy(f)
    for tag, B, cin, cout, H, k, up, demod, noise in [
        ('c3_up1', 3, 6, 5, 8, 3, 1, True, 'const'),
        ('c3_up2', 3, 6, 5, 5, 3, 2, True, 'batch'),
        ('c3_up2_b1', 1, 4, 7, 4, 3, 2, True, None),
        ('c1_rgb', 3, 6, 9, 8, 1, 1, False, None),
        ('c3_up1_b1_nonoise', 1, 5, 3, 6, 3, 1, True, None),
    ]:
        x = g.randn(B, cin, H, H).astype(np.float32)
        w = g.randn(cout, cin, k, k).astype(np.float32)
        s = (1.0 + 0.5 * g.randn(B, cin)).astype(np.float32)
        nz = None
        if noise == 'const':
            nz = (0.3 * g.randn(H * up, H * up)).astype(np.float32)
        elif noise == 'batch':
            nz = (0.3 * g.randn(B, 1, H * up, H * up)).astype(np.float32)
        y = ref_sg2.modulated_conv2d(x=T(x), weight=T(w), styles=T(s), noise=None if nz is None else T(nz), up=up,
                                     padding=k // 2, resample_filter=f, demodulate=demod, flip_weight=(up == 1), fused_modconv=True)
        arrays[f'{tag}_x'], arrays[f'{tag}_w'], arrays[f'{tag}_s'], arrays[f'{tag}_y'] = x, w, s, npy(y)
        if nz is not None:
            arrays[f'{tag}_noise'] = nz
        arrays[f'{tag}_meta'] = np.array([k, up, int(demod)], dtype=np.int64)
    save('modconv', **arrays)


def _mlp_cfg(F, hid, marcher):
    return EasyDict(tri_plane=EasyDict(feat_dim=F, mlp=EasyDict(n_layers=2, hid_dim=hid)), has_view_cond=False,
                    ray_marcher_type=marcher)


def gen_field():
    g = np.random.RandomState(4)
    arrays = {}
    B, F, R, hid, P = 2, 8, 16, 16, 300
    planes = g.randn(B, 3 * F, R, R).astype(np.float32)
    coords = (g.rand(B, P, 3).astype(np.float32) * 2 - 1) * 0.62          # |coord| up to 0.62 > cube half-size 0.5
    coords[0, :4] = [[0.5, 0.5, 0.5], [-0.5, -0.5, -0.5], [0.0, 0.0, 0.0], [0.5, -0.5, 0.25]]
    arrays['planes'], arrays['coords'] = planes, coords
    for marcher in ('classical', 'mip'):
        torch.manual_seed(5)
        mlp = TriPlaneMLP(_mlp_cfg(F, hid, marcher), out_dim=3).eval()
        with torch.no_grad():
            mlp.model[0].bias.copy_(T(g.randn(hid).astype(np.float32) * 0.3))
            mlp.model[1].bias.copy_(T(g.randn(4).astype(np.float32) * 0.3))
            out = ref_tpr.simple_tri_plane_renderer(T(planes), T(coords), mlp, scale=0.5)
        for i in (0, 1):
            arrays[f'{marcher}_w{i}'] = npy(mlp.model[i].weight)
            arrays[f'{marcher}_b{i}'] = npy(mlp.model[i].bias)
        arrays[f'{marcher}_rgb'], arrays[f'{marcher}_sigma'] = npy(out['rgb']), npy(out['sigma'])
    # the raw bilinear+mean features (grid_sample output before the MLP)
    x = T(planes).reshape(B * 3, F, R, R)
    c = T(coords) / 0.5
    c2d = torch.stack([c[..., [0, 1]], c[..., [0, 2]], c[..., [1, 2]]], dim=1).view(B * 3, 1, P, 2)
    feats = torch.nn.functional.grid_sample(x, c2d, mode='bilinear', align_corners=True).view(B, 3, F, P)
    arrays['feats_mean'] = npy(feats.permute(0, 1, 3, 2).mean(dim=1))
    save('field', **arrays)


def gen_mlp_variants():
    """TriPlaneMLP outside the fused kernel's form (networks_epigraf.py:35-43; VERDICT r05 missing #2): n_layers = 0 (nn.Identity, the planes carry
    rgb + sigma), 3 and 4 layers, has_view_cond (which the reference's forward only survives with hid_dim == out_dim), widths outside the kernel's table
    -- simple_tri_plane_renderer on seeded planes / coordinates, and one whole ImportanceRenderer.forward with the 3-layer decoder."""
    g = np.random.RandomState(44)
    arrays = {}
    B, R, P = 2, 16, 200
    # (n_layers = 0 cannot be captured: the reference's own forward raises AttributeError there -- `backbone_out_dim` is only set on the other branch)
    variants = dict(n3=dict(F=8, hid=16, n=3, view=False, marcher='classical'),
                    n4mip=dict(F=8, hid=16, n=4, view=False, marcher='mip'), view=dict(F=8, hid=3, n=2, view=True, marcher='classical'),
                    odd=dict(F=12, hid=20, n=2, view=False, marcher='mip'))
    coords = (g.rand(B, P, 3).astype(np.float32) * 2 - 1) * 0.62
    arrays['coords'] = coords
    mlps = {}
    for tag, v in variants.items():
        planes = g.randn(B, 3 * v['F'], R, R).astype(np.float32)
        cfg = EasyDict(tri_plane=EasyDict(feat_dim=v['F'], mlp=EasyDict(n_layers=v['n'], hid_dim=v['hid'])), has_view_cond=v['view'], ray_marcher_type=v['marcher'])
        torch.manual_seed(45)
        mlp = TriPlaneMLP(cfg, out_dim=3).eval()
        with torch.no_grad():
            if v['n'] > 0:
                for i, fc in enumerate(mlp.model):
                    fc.bias.copy_(T(g.randn(*fc.bias.shape).astype(np.float32) * 0.3))
                    arrays[f'{tag}_w{i}'], arrays[f'{tag}_b{i}'] = npy(fc.weight), npy(fc.bias)
            out = ref_tpr.simple_tri_plane_renderer(T(planes), T(coords), mlp, scale=0.5)
        arrays[f'{tag}_planes'], arrays[f'{tag}_rgb'], arrays[f'{tag}_sigma'] = planes, npy(out['rgb']), npy(out['sigma'])
        mlps[tag] = (mlp, planes)
    # the whole renderer with the 3-layer decoder (rays of an 8 x 8 image, 8 + 8 samples): same draws as inputs
    mlp, planes = mlps['n3']
    S, hw = 8, 8
    cam = dict(angles=np.array([[0.3, 1.2, 0.0], [-0.4, 1.7, 0.0]], np.float32), radius=np.ones(2, np.float32), look_at=np.zeros((2, 3), np.float32))
    c2w = ref_ru.compute_cam2world_matrix(TensorGroup(angles=T(cam['angles']), radius=T(cam['radius']), look_at=T(cam['look_at'])))
    ro, rd = ref_tpr.sample_rays(c2w, fov=T(np.array([25.0, 35.0], np.float32)), resolution=(hw, hw))
    u1, u2 = g.rand(B, hw * hw, S, 1).astype(np.float32), g.rand(B * hw * hw, S).astype(np.float32)
    opts = EasyDict(box_size=1.0, num_proposal_steps=S, num_fine_steps=S, clamp_mode='softplus', use_inf_depth=True, ray_start=0.75, ray_end=1.25, max_batch_res=128,
                    white_back=False, last_back=False, density_noise=0.0, cut_quantile=0.0, density_bias=0.0)
    rend = ref_tpr.ImportanceRenderer('classical')
    with torch.no_grad(), PatchedRNG(rand_like=[T(u1)], rand=[T(u2)]):
        res = rend(T(planes).view(B, 3, 8, R, R), mlp, ro, rd, opts)
    arrays.update(r_angles=cam['angles'], r_fov=np.array([25.0, 35.0], np.float32), r_ray_o=npy(ro), r_ray_d=npy(rd), r_u_coarse=u1, r_u_fine=u2,
                  r_rgb=npy(res[0]), r_depth=npy(res[1]))
    save('mlp_variants', **arrays)


def gen_mapping_cam():
    """MappingNetwork with camera conditioning (layers.py:84-93,127-138; VERDICT r05 missing #3): Fourier-encoded and raw yaw / pitch appended to the label,
    explicit angles and the eval-time stand-in `mean_camera_params`."""
    from src.training.layers import MappingNetwork as RefMapping
    g = np.random.RandomState(46)
    arrays = {}
    for tag, c_dim, raw in (('four', 5, False), ('raw', 0, True)):
        torch.manual_seed(47)
        mean_cam = np.array([0.35, 1.45, 0.0, 1.0, 18.0], np.float32)
        m = RefMapping(z_dim=16, c_dim=c_dim, w_dim=24, num_ws=5, num_layers=2, camera_cond=True, camera_cond_drop_p=0.0, camera_raw_scalars=raw,
                       mean_camera_params=T(mean_cam)).eval()
        with torch.no_grad():
            for n_, p_ in m.named_parameters():
                if n_.endswith('bias'):
                    p_.copy_(T((g.randn(*p_.shape) * (10.0 if n_.startswith('fc') else 0.1)).astype(np.float32)))
            m.w_avg.copy_(T(g.randn(24).astype(np.float32) * 0.1))
            z = g.randn(4, 16).astype(np.float32)
            c = np.eye(c_dim, dtype=np.float32)[g.randint(0, c_dim, 4)] if c_dim > 0 else np.zeros((4, 0), np.float32)
            ang = np.stack([g.uniform(-7.0, 7.0, 4), g.uniform(0.3, 2.8, 4), np.zeros(4)], 1).astype(np.float32)     # yaw beyond +-2 pi: the wrap matters
            ws = m(T(z), T(c) if c_dim > 0 else None, camera_angles=T(ang))
            ws_mean = m(T(z), T(c) if c_dim > 0 else None)
            ws_psi = m(T(z), T(c) if c_dim > 0 else None, camera_angles=T(ang), truncation_psi=0.6)
        for n_, v_ in m.state_dict().items():
            arrays[f'{tag}::{n_}'] = npy(v_)
        arrays.update({f'{tag}_z': z, f'{tag}_c': c, f'{tag}_angles': ang, f'{tag}_ws': npy(ws), f'{tag}_ws_mean': npy(ws_mean), f'{tag}_ws_psi06': npy(ws_psi)})
    save('mapping_cam', **arrays)


def gen_field_grad():
    """Autograd through simple_tri_plane_renderer + TriPlaneMLP (grid_sample backward + MLP backward): gradients w.r.t. the planes
    and the four MLP tensors, both MLP output modes, two (feat, hid) sizes."""
    arrays = {}
    for tag, (B, F, R, hid, P) in dict(small=(2, 8, 16, 16, 300), hot=(1, 32, 16, 64, 400)).items():
        g = np.random.RandomState(44 + F)
        planes = g.randn(B, 3 * F, R, R).astype(np.float32)
        coords = (g.rand(B, P, 3).astype(np.float32) * 2 - 1) * 0.62
        coords[0, :4] = [[0.5, 0.5, 0.5], [-0.5, -0.5, -0.5], [0.0, 0.0, 0.0], [0.5, -0.5, 0.25]]
        d_rgb, d_sigma = g.randn(B, P, 3).astype(np.float32), g.randn(B, P, 1).astype(np.float32)
        arrays.update({f'{tag}_planes': planes, f'{tag}_coords': coords, f'{tag}_d_rgb': d_rgb, f'{tag}_d_sigma': d_sigma})
        for marcher in ('classical', 'mip'):
            torch.manual_seed(6)
            mlp = TriPlaneMLP(_mlp_cfg(F, hid, marcher), out_dim=3)
            with torch.no_grad():
                mlp.model[0].bias.copy_(T(g.randn(hid).astype(np.float32) * 0.3))
                mlp.model[1].bias.copy_(T(g.randn(4).astype(np.float32) * 0.3))
            x = T(planes).requires_grad_(True)
            cc = T(coords).requires_grad_(True)          # round 3: the gradient w.r.t. the sample positions too (camera training, loss.py:69-83)
            out = ref_tpr.simple_tri_plane_renderer(x, cc, mlp, scale=0.5)
            params = [mlp.model[0].weight, mlp.model[0].bias, mlp.model[1].weight, mlp.model[1].bias]
            grads = torch.autograd.grad([out['rgb'], out['sigma']], [x] + params + [cc], [T(d_rgb), T(d_sigma)])
            for name, t in zip(('w0', 'b0', 'w1', 'b1'), params):
                arrays[f'{tag}_{marcher}_{name}'] = npy(t)
            for name, t in zip(('d_planes', 'd_w0', 'd_b0', 'd_w1', 'd_b1', 'd_coords'), grads):
                arrays[f'{tag}_{marcher}_{name}'] = npy(t)
    save('field_grad', **arrays)


def gen_render_grad():
    """Autograd through ImportanceRenderer.forward (tri_plane_renderer.py:126-170): gradient of sum(rgb * d_rgb) + sum(depth * d_depth)
    w.r.t. the planes and the decoder tensors, both marchers, with the stratification / inverse-CDF draws fixed."""
    arrays = {}
    g = np.random.RandomState(61)
    B, F, H, hid, hw, S = 2, 8, 16, 16, 6, 8
    R = hw * hw
    planes = g.randn(B, 3 * F, H, H).astype(np.float32)
    cam = TensorGroup(angles=T(np.array([[0.3, 1.2, 0.0], [-0.6, 1.8, 0.0]], np.float32)), radius=T(np.ones(2, np.float32)),
                      look_at=T(np.zeros((2, 3), np.float32)))
    ro, rd = ref_tpr.sample_rays(ref_ru.compute_cam2world_matrix(cam), T(np.array([25.0, 40.0], np.float32)), (hw, hw))
    u1, u2 = g.rand(B, R, S, 1).astype(np.float32), g.rand(B * R, S).astype(np.float32)
    d_rgb, d_depth = g.randn(B, R, 3).astype(np.float32), g.randn(B, R, 1).astype(np.float32)
    arrays.update(planes=planes, ray_o=npy(ro), ray_d=npy(rd), u_coarse=u1, u_fine=u2, d_rgb=d_rgb, d_depth=d_depth)
    for marcher in ('classical', 'mip'):
        torch.manual_seed(7)
        mlp = TriPlaneMLP(_mlp_cfg(F, hid, marcher), out_dim=3)
        with torch.no_grad():
            mlp.model[0].bias.copy_(T(g.randn(hid).astype(np.float32) * 0.3))
            mlp.model[1].bias.copy_(T(g.randn(4).astype(np.float32) * 0.3))
        opts = EasyDict(box_size=1.0, num_proposal_steps=S, num_fine_steps=S, clamp_mode='softplus', use_inf_depth=True, ray_start=0.75, ray_end=1.25,
                        white_back=(marcher == 'mip'), last_back=False, density_bias=0.0, cut_quantile=0.0, max_batch_res=64, density_noise=0.0)
        x = T(planes).requires_grad_(True)
        rend = ref_tpr.ImportanceRenderer(ray_marcher_type=marcher)
        with PatchedRNG(rand_like=[T(u1)], rand=[T(u2)]):
            rgb, depth, wsum, fT = rend(x.view(B, 3, F, H, H), mlp, ro, rd, opts)
        params = [mlp.model[0].weight, mlp.model[0].bias, mlp.model[1].weight, mlp.model[1].bias]
        grads = torch.autograd.grad([rgb, depth], [x] + params, [T(d_rgb), T(d_depth)])
        for name, t in zip(('w0', 'b0', 'w1', 'b1'), params):
            arrays[f'{marcher}_{name}'] = npy(t)
        for name, t in zip(('d_planes', 'd_w0', 'd_b0', 'd_w1', 'd_b1'), grads):
            arrays[f'{marcher}_{name}'] = npy(t)
        arrays[f'{marcher}_rgb'] = npy(rgb)
    save('render_grad', **arrays)


def gen_sampling():
    g = np.random.RandomState(6)
    arrays = {}
    B, R, S = 2, 37, 12
    ray_o = torch.zeros(B, R, 3)
    for marcher in ('classical', 'mip'):
        rend = ref_tpr.ImportanceRenderer(marcher)
        u = g.rand(B, R, S, 1).astype(np.float32)
        with PatchedRNG(rand_like=[T(u)]):
            sd = rend.sample_stratified(ray_o, 0.0, 1.0, S)
        arrays[f'{marcher}_u_coarse'], arrays[f'{marcher}_sdist'] = u, npy(sd)
        # importance sampling from given weights
        Wn = S
        wts = np.abs(g.randn(B, R, Wn, 1)).astype(np.float32) * (g.rand(B, R, Wn, 1) > 0.3)
        wts[0, 0] = 0.0                               # all-zero weights ray
        wts[0, 1, :, 0] = 0.0; wts[0, 1, 5, 0] = 1.0  # single spike
        u2 = g.rand(B * R, S).astype(np.float32)
        u2[0, :3] = [0.0, 0.99999994, 0.5]
        with PatchedRNG(rand=[T(u2)]), Capture() as cap:
            sf = rend.sample_importance(sd, T(wts.astype(np.float32)), S)
        inds = cap.inds[0]
        arrays[f'{marcher}_weights'] = wts.astype(np.float32)
        arrays[f'{marcher}_u_fine'] = u2
        arrays[f'{marcher}_sdist_fine'] = npy(sf)
        arrays[f'{marcher}_inds'] = npy(inds).astype(np.int64)
        arrays[f'{marcher}_below'] = np.maximum(npy(inds) - 1, 0).astype(np.int64)
        arrays[f'{marcher}_above'] = np.minimum(npy(inds), S - 2).astype(np.int64)
    # unify_samples incl. permutation
    rend = ref_tpr.ImportanceRenderer('classical')
    S2 = 9
    d1 = np.sort(g.rand(B, R, S, 1).astype(np.float32), axis=2)
    d2 = g.rand(B, R, S2, 1).astype(np.float32)
    c1, c2 = g.randn(B, R, S, 3).astype(np.float32), g.randn(B, R, S2, 3).astype(np.float32)
    s1, s2 = g.randn(B, R, S, 1).astype(np.float32), g.randn(B, R, S2, 1).astype(np.float32)
    with Capture() as cap:
        d, c, s = rend.unify_samples(T(d1), T(c1), T(s1), T(d2), T(c2), T(s2))
    arrays.update(un_d1=d1, un_d2=d2, un_c1=c1, un_c2=c2, un_s1=s1, un_s2=s2, un_d=npy(d), un_c=npy(c), un_s=npy(s),
                  un_perm=npy(cap.perm[0])[..., 0].astype(np.int64))
    save('sampling', **arrays)


def gen_sampling_hot():
    """sample_importance at the ray-step counts of BASELINE configs[0..4] (S = 32, 48, 64, 96): the pdf rows have S-2 = 30, 46,
    62, 94 elements, i.e. 3 / 5 / 7 / 11 whole 8-lane vectors plus a 6-element tail in torch's CPU sum kernel -- every branch of
    its accumulation order (4-way interleave, left-over vectors, scalar tail).  Weights are realistic compositing weights
    (alpha * transmittance of random densities, many near zero) so cdf knots cluster the way they do in a render."""
    g = np.random.RandomState(66)
    arrays = {}
    for marcher in ('classical', 'mip'):
        rend = ref_tpr.ImportanceRenderer(marcher)
        for S in (32, 48, 64, 96):
            B, R = 1, 64
            u = g.rand(B, R, S, 1).astype(np.float32)
            with PatchedRNG(rand_like=[T(u)]):
                sd = rend.sample_stratified(torch.zeros(B, R, 3), 0.0, 1.0, S)
            sig = np.maximum(g.randn(B, R, S, 1) * 4.0 - 1.0, 0).astype(np.float32) * (g.rand(B, R, S, 1) > 0.5)
            alpha = 1.0 - np.exp(-sig * (1.0 / S) * 20.0)
            Tr = np.cumprod(np.concatenate([np.ones_like(alpha[:, :, :1]), 1.0 - alpha[:, :, :-1] + 1e-10], 2), 2)
            wts = (alpha * Tr).astype(np.float32)
            Wn = S if marcher == 'classical' else S - 1            # MipRayMarcher2 without inf depth returns S-1 weights
            wts = wts[:, :, :Wn]
            u2 = g.rand(B * R, S).astype(np.float32)
            with PatchedRNG(rand=[T(u2)]), Capture() as cap:
                sf = rend.sample_importance(sd, T(wts), S)
            tag = f'{marcher}{S}'
            arrays[f'{tag}_sdist'], arrays[f'{tag}_weights'], arrays[f'{tag}_u_fine'] = npy(sd), wts, u2
            arrays[f'{tag}_sdist_fine'] = npy(sf)
            arrays[f'{tag}_inds'] = npy(cap.inds[0]).astype(np.int16)
    save('sampling_hot', **arrays)


def gen_marchers():
    g = np.random.RandomState(7)
    arrays = {}
    B, R, S = 2, 29, 14
    colors = g.randn(B, R, S, 3).astype(np.float32)
    dens = (g.randn(B, R, S, 1) * 4).astype(np.float32)
    dens[0, 0, :, 0] = 30.0            # softplus threshold branch
    dens[0, 1, :, 0] = -30.0           # nearly empty ray
    depths = np.sort(0.75 + 0.5 * g.rand(B, R, S, 1).astype(np.float32), axis=2)
    arrays.update(colors=colors, densities=dens, depths=depths)
    cm = ref_tpr.ClassicalRayMarcher()
    for tag, opts in [('cl_inf', dict(use_inf_depth=True)), ('cl_noinf', dict(use_inf_depth=False)),
                      ('cl_lastback', dict(use_inf_depth=True, last_back=True)), ('cl_relu', dict(use_inf_depth=True, clamp_mode='relu'))]:
        ro = EasyDict(clamp_mode='softplus', cut_quantile=0.0, density_bias=0.0, last_back=False, white_back=False)
        ro.update(opts)
        rgb, dep, w, fT = cm(T(colors), T(dens), T(depths), ro)
        arrays.update({f'{tag}_rgb': npy(rgb), f'{tag}_depth': npy(dep), f'{tag}_weights': npy(w), f'{tag}_T': npy(fT)})
    # cut_quantile = 0.5 (the non-flatness score's setting, non_flatness_score.py:9): densities below the global median are zeroed
    ro = EasyDict(clamp_mode='softplus', cut_quantile=0.5, density_bias=0.0, last_back=False, white_back=False, use_inf_depth=True)
    rgb, dep, w, fT = cm(T(colors), T(dens), T(depths), ro)
    arrays.update({'cl_cut_rgb': npy(rgb), 'cl_cut_depth': npy(dep), 'cl_cut_weights': npy(w), 'cl_cut_T': npy(fT)})
    mm = ref_tpr.MipRayMarcher2()
    colors01 = (1 / (1 + np.exp(-colors))).astype(np.float32)
    ro = EasyDict(clamp_mode='softplus', cut_quantile=0.3, density_bias=0.0, white_back=False, use_inf_depth=True)
    rgb, dep, w, fT = mm(T(colors01), T(dens), T(depths), ro)
    arrays.update({'mip_cut_rgb': npy(rgb), 'mip_cut_depth': npy(dep), 'mip_cut_weights': npy(w), 'mip_cut_T': npy(fT)})
    arrays['colors01'] = colors01
    for tag, opts in [('mip_inf', dict(use_inf_depth=True, white_back=False)), ('mip_noinf_white', dict(use_inf_depth=False, white_back=True)),
                      ('mip_bias', dict(use_inf_depth=True, white_back=False, density_bias=-1.0))]:
        ro = EasyDict(clamp_mode='softplus', cut_quantile=0.0, density_bias=0.0)
        ro.update(opts)
        rgb, dep, w, fT = mm(T(colors01), T(dens), T(depths), ro)
        arrays.update({f'{tag}_rgb': npy(rgb), f'{tag}_depth': npy(dep), f'{tag}_weights': npy(w), f'{tag}_T': npy(fT)})
    save('marchers', **arrays)


MARCH_GRAD_CASES = dict(cl_inf=dict(mode='classical', use_inf_depth=True), cl_noinf_lastback=dict(mode='classical', use_inf_depth=False, last_back=True),
                        cl_relu=dict(mode='classical', use_inf_depth=True, clamp_mode='relu'), mip_inf=dict(mode='mip', use_inf_depth=True),
                        mip_noinf_white_bias=dict(mode='mip', use_inf_depth=False, white_back=True, density_bias=-1.0))


def gen_march_grad():
    """Autograd through the reference ray marchers (tri_plane_renderer.py:299-398): gradients w.r.t. colours and raw densities for
    given d_rgb, d_depth, d_weights."""
    g = np.random.RandomState(23)
    B, R, S = 2, 9, 12
    colors = g.randn(B, R, S, 3).astype(np.float32)
    dens = (g.randn(B, R, S, 1) * 3).astype(np.float32)
    dens[0, 0, :, 0] = 25.0            # beyond the softplus threshold
    dens[0, 1, :, 0] = -30.0           # nearly empty ray
    dens[0, 2, 3, 0] = 40.0            # an opaque sample in the middle of the ray: q = 1e-10 behind it
    depths = np.sort(0.75 + 0.5 * g.rand(B, R, S, 1).astype(np.float32), axis=2)
    arrays = dict(colors=colors, densities=dens, depths=depths)
    for tag, kw in MARCH_GRAD_CASES.items():
        mip = kw['mode'] == 'mip'
        ro = EasyDict(clamp_mode=kw.get('clamp_mode', 'softplus'), cut_quantile=0.0, density_bias=kw.get('density_bias', 0.0), last_back=kw.get('last_back', False),
                      white_back=kw.get('white_back', False), use_inf_depth=kw['use_inf_depth'])
        c = T(colors if not mip else (1 / (1 + np.exp(-colors))).astype(np.float32)).requires_grad_(True)
        d = T(dens).requires_grad_(True)
        rgb, dep, w, fT = (ref_tpr.MipRayMarcher2() if mip else ref_tpr.ClassicalRayMarcher())(c, d, T(depths), ro)
        d_rgb, d_dep, d_w = T(g.randn(*rgb.shape).astype(np.float32)), T(g.randn(*dep.shape).astype(np.float32)), T(g.randn(*w.shape).astype(np.float32))
        dc, dd = torch.autograd.grad([rgb, dep, w], [c, d], [d_rgb, d_dep, d_w])
        arrays.update({f'{tag}_c': npy(c), f'{tag}_d_rgb': npy(d_rgb), f'{tag}_d_depth': npy(d_dep), f'{tag}_d_weights': npy(d_w), f'{tag}_dc': npy(dc),
                       f'{tag}_dd': npy(dd)})
    save('march_grad', **arrays)


def gen_camera():
    g = np.random.RandomState(8)
    arrays = {}
    B = 5
    angles = np.stack([g.uniform(-3, 3, B), g.uniform(0.2, 2.9, B), np.zeros(B)], 1).astype(np.float32)
    radius = g.uniform(0.8, 1.3, B).astype(np.float32)
    look_at = np.stack([g.uniform(0, 6.28, B), g.uniform(0.1, 3.0, B), g.uniform(0, 0.2, B)], 1).astype(np.float32)
    fov = g.uniform(10, 45, B).astype(np.float32)
    cam = TensorGroup(angles=T(angles), radius=T(radius), fov=T(fov), look_at=T(look_at))
    c2w = ref_ru.compute_cam2world_matrix(cam)
    arrays.update(angles=angles, radius=radius, look_at=look_at, fov=fov, c2w=npy(c2w))
    for (h, w) in [(8, 8), (5, 7), (16, 16)]:
        o, d = ref_tpr.sample_rays(c2w, fov=T(fov), resolution=(h, w))
        arrays[f'ray_o_{h}x{w}'], arrays[f'ray_d_{h}x{w}'] = npy(o), npy(d)
    ps = g.uniform(0.3, 0.8, (B, 2)).astype(np.float32)
    po = g.uniform(0.0, 0.2, (B, 2)).astype(np.float32)
    o, d = ref_tpr.sample_rays(c2w, fov=T(fov), resolution=(6, 6), patch_params=dict(scales=T(ps), offsets=T(po)))
    arrays.update(patch_scales=ps, patch_offsets=po, ray_o_patch=npy(o), ray_d_patch=npy(d))
    o, d = ref_tpr.sample_rays(c2w, fov=18.0, resolution=(4, 4))            # python-float fov: one shared ray fan
    arrays.update(ray_o_scalar_fov=npy(o), ray_d_scalar_fov=npy(d))
    # KATs from scripts/testing/validate_ray_bounds.py
    arrays['kat_frustum'] = np.array([-0.6344010829925537, 0.6351816654205322, -0.1962990164756775, 0.19598299264907837])
    save('camera', **arrays)


def ref_cfg(cfg):
    """Our GeneratorConfig -> the EasyDict the reference Generator reads."""
    return EasyDict(
        z_dim=cfg.z_dim, w_dim=cfg.w_dim, c_dim=cfg.c_dim, map_depth=cfg.map_depth, cbase=cfg.cbase, cmax=cfg.cmax,
        fmaps=cfg.fmaps, use_noise=cfg.use_noise,
        tri_plane=EasyDict(res=cfg.tri_plane_res, feat_dim=cfg.feat_dim, mlp=EasyDict(n_layers=2, hid_dim=cfg.mlp_hid)),
        has_view_cond=False, ray_marcher_type=cfg.ray_marcher_type, num_ray_steps=cfg.num_ray_steps,
        max_batch_res=cfg.max_batch_res, density_bias=cfg.density_bias, use_inf_depth=cfg.use_inf_depth, use_full_box=False,
        camera=EasyDict(ray=EasyDict(start=cfg.ray_start, end=cfg.ray_end), cube_scale=cfg.cube_scale),
        dataset=EasyDict(white_back=cfg.white_back, last_back=cfg.last_back),
        patch=EasyDict(enabled=False, resolution=cfg.img_resolution),
        nerf_noise_std_init=0.0, nerf_noise_kimg_growth=0,
        depth_adaptor=EasyDict(enabled=False), camera_adaptor=EasyDict(enabled=False),
    )


def build_ref_generator(cfg, sd):
    G = Generator(ref_cfg(cfg), img_resolution=cfg.img_resolution, img_channels=3, mapping_kwargs={}, num_fp16_res=0,
                  conv_clamp=None, fused_modconv_default='inference_only').eval()
    G.load_state_dict({k: T(v) for k, v in sd.items()}, strict=True)     # strict: our key/shape spec == reference's
    return G


def gen_mapping():
    arrays = {}
    for tag, cfg in [('c0', tdgp.config.config_tiny()), ('c10', tdgp.config.config_mid())]:
        sd = tdgp.weights.random_state_dict(cfg, seed=11, exercise_all=True)
        G = build_ref_generator(cfg, sd)
        inp = tdgp.weights.synthetic_inputs(cfg, batch=3, seed=12)
        with torch.no_grad():
            ws = G.mapping(T(inp['z']), T(inp['c']))
            ws_t = G.mapping(T(inp['z']), T(inp['c']), truncation_psi=0.7)
            ws_tc = G.mapping(T(inp['z']), T(inp['c']), truncation_psi=0.3, truncation_cutoff=3)
        arrays.update({f'{tag}_z': inp['z'], f'{tag}_c': inp['c'], f'{tag}_ws': npy(ws), f'{tag}_ws_psi07': npy(ws_t),
                       f'{tag}_ws_psi03_cut3': npy(ws_tc)})
    save('mapping', **arrays)


def gen_e2e(tag, cfg, batch, seed, keep_intermediates):
    sd = tdgp.weights.random_state_dict(cfg, seed=seed, exercise_all=True)
    G = build_ref_generator(cfg, sd)
    inp = tdgp.weights.synthetic_inputs(cfg, batch=batch, seed=seed + 1)
    cam = TensorGroup(**{k: T(v) for k, v in inp['camera'].items()})
    R, S = cfg.img_resolution ** 2, cfg.num_ray_steps
    arrays = dict(z=inp['z'], c=inp['c'], u_coarse=inp['u_coarse'], u_fine=inp['u_fine'],
                  **{'cam_' + k: v for k, v in inp['camera'].items()})
    arrays['seed'] = np.array([seed, batch], dtype=np.int64)
    with torch.no_grad():
        ws = G.mapping(T(inp['z']), T(inp['c']))
        stage = {}
        rend = G.synthesis.renderer
        o_si = rend.sample_importance

        def si(z_vals, weights, n):                      # the importance-sampling stage as the reference calls it (:153)
            r = o_si(z_vals, weights, n)
            stage.update(imp_sdist=npy(z_vals), imp_weights=npy(weights), imp_sdist_fine=npy(r))
            return r
        rend.sample_importance = si
        with PatchedRNG(rand_like=[T(inp['u_coarse']).reshape(batch, R, S, 1)], rand=[T(inp['u_fine'])]), Capture() as cap:
            out = G.synthesis(ws, camera_params=cam, noise_mode='const', render_opts=dict(return_depth=True))
        rend.sample_importance = o_si
        arrays.update(ws=npy(ws), img=npy(out.img), depth=npy(out.depth))
        if keep_intermediates:
            arrays.update(stage)
        # The reference's own fp32 reproducibility: same inputs, native (non-oneDNN) convolutions, 1 thread.
        # |img - img_alt| is the noise floor any fp32 re-implementation is compared against (parity).
        torch.backends.mkldnn.enabled = False
        torch.set_num_threads(1)
        with PatchedRNG(rand_like=[T(inp['u_coarse']).reshape(batch, R, S, 1)], rand=[T(inp['u_fine'])]):
            alt = G.synthesis(ws, camera_params=cam, noise_mode='const', render_opts=dict(return_depth=True))
        torch.backends.mkldnn.enabled = True
        torch.set_num_threads(8)
        arrays.update(img_alt=npy(alt.img), depth_alt=npy(alt.depth))
        # The reference ITSELF in float64 on the same ws / cameras / draws: the exactly-rounded image the fp32 runs (the reference's
        # and ours) are both measured against -- the pin of the RGB bound that does not lean on this repo's oracle (VERDICT r02 #4a).
        G64 = build_ref_generator(cfg, sd).double()
        # cameras -> c2w -> rays are computed in fp32 as always (rendering_utils.py builds fp32 constants; the rays are bit-identical between
        # the implementations anyway) and handed to the float64 run: "exact" = exactly rounded given those rays, ws and draws
        import src.training.networks_epigraf as ref_epi
        c2w32 = ref_ru.compute_cam2world_matrix(cam)
        ro32, rd32 = ref_tpr.sample_rays(c2w32, fov=cam.fov, resolution=(cfg.img_resolution,) * 2)
        o_c2w, o_rays = ref_epi.compute_cam2world_matrix, ref_epi.sample_rays
        ref_epi.compute_cam2world_matrix = lambda camera_params: c2w32
        ref_epi.sample_rays = lambda *a, **k: (ro32.double(), rd32.double())
        torch.set_default_dtype(torch.float64)          # linspace / arange / constants created inside the renderer
        f32 = torch.float32
        torch.float32 = torch.float64                   # SynthesisBlock casts to `torch.float32` by name (networks_stylegan2.py:237,250,268) and
        try:                                            # conv2d_resample.py:73 asserts the FIR's dtype against it: the name means "working precision"
            with PatchedRNG(rand_like=[T(inp['u_coarse']).reshape(batch, R, S, 1).double()], rand=[T(inp['u_fine']).double()]):
                o64 = G64.synthesis(ws.double(), camera_params=cam, noise_mode='const', render_opts=dict(return_depth=True))
        finally:
            torch.float32 = f32
            torch.set_default_dtype(torch.float32)
            ref_epi.compute_cam2world_matrix, ref_epi.sample_rays = o_c2w, o_rays
        assert o64.img.dtype == torch.float64
        arrays.update(img_f64=npy(o64.img), depth_f64=npy(o64.depth))
        if keep_intermediates:
            planes = G.synthesis.tri_plane_decoder(ws, noise_mode='const')
            c2w = ref_ru.compute_cam2world_matrix(cam)
            ro, rd = ref_tpr.sample_rays(c2w, fov=cam.fov, resolution=(cfg.img_resolution,) * 2)
            arrays.update(planes=npy(planes), c2w=npy(c2w), ray_o=npy(ro), ray_d=npy(rd),
                          inds=npy(cap.inds[0]).astype(np.int64), perm=npy(cap.perm[0])[..., 0].astype(np.int64))
            # per-block activations for the backbone
            x = img = None
            dec = G.synthesis.tri_plane_decoder
            w_idx = 0
            for r in dec.block_resolutions:
                blk = getattr(dec, f'b{r}')
                x, img = blk(x, img, ws.narrow(1, w_idx, blk.num_conv + blk.num_torgb), noise_mode='const')
                w_idx += blk.num_conv
                arrays[f'x{r}'] = npy(x)
            # the non-flatness score's rendering: cut_quantile = 0.5 through the whole renderer (both marcher calls)
            with PatchedRNG(rand_like=[T(inp['u_coarse']).reshape(batch, R, S, 1)], rand=[T(inp['u_fine'])]):
                cut = G.synthesis(ws, camera_params=cam, noise_mode='const', render_opts=dict(return_depth=True, cut_quantile=0.5))
            arrays.update(img_cut=npy(cut.img), depth_cut=npy(cut.depth))
            # also the 'none' noise mode image (no noise inputs at all)
            with PatchedRNG(rand_like=[T(inp['u_coarse']).reshape(batch, R, S, 1)], rand=[T(inp['u_fine'])]):
                arrays['img_noise_none'] = npy(G.synthesis(ws, camera_params=cam, noise_mode='none'))
    save(tag, **arrays)



# BASELINE.json configs[0..2] at their REAL size: one image each from the reference itself.
FULL_CONFIGS = dict(c1=('config_c1', 101, [0, 17, 40, 63]), c2=('config_c2', 103, [0, 77, 127]), c3=('config_c3', 105, [3, 128, 250]),
                    c4=('config_c4', 109, [7, 130, 251]),          # configs[3]: cmax 1024 / cbase 65536
                    c2mip=('config_c2', 111, [5, 64, 120]))        # configs[1]'s shape with the 'mip' ray marcher (MipRayMarcher2, white background)


def _pack_diff(x, base):
    """(x - base) as float16 in units of its own maximum: a float64 / re-run image stored next to the fp32 one it differs from by
    ~1e-7 of the range costs 2 bytes per pixel and is recovered to ~1e-10 of the range (conftest.unpack_full_golden)."""
    d = np.asarray(x, np.float64) - np.asarray(base, np.float64)
    scale = float(np.abs(d).max()) or 1.0
    return (d / scale).astype(np.float16), np.float64(scale)


def gen_e2e_full(tag):
    """`e2e_full_<tag>.npz`: the reference Generator at a BASELINE configuration's real size (512^2 tri-planes, 512-channel backbone,
    the configuration's image size and ray-step count), batch 1, weights = random_state_dict(cfg, seed, exercise_all=True) and inputs =
    synthetic_inputs(cfg, 1, seed + 1) -- both regenerate from the seed, so only outputs are stored: img, depth (fp32), the reference's
    own float64 run and its native-convolution re-run as packed differences, and for a strip of image rows the integer rows of the
    importance stage (searchsorted indices, the cdf knots they were ranked against, sort permutation) plus the fine samples."""
    cfg_name, seed, rows = FULL_CONFIGS[tag]
    cfg = getattr(tdgp.config, cfg_name)()
    if tag.endswith('mip'):
        cfg.ray_marcher_type, cfg.white_back = 'mip', True
    sd = tdgp.weights.random_state_dict(cfg, seed=seed, exercise_all=True)
    G = build_ref_generator(cfg, sd)
    inp = tdgp.weights.synthetic_inputs(cfg, batch=1, seed=seed + 1)
    cam = TensorGroup(**{k: T(v) for k, v in inp['camera'].items()})
    h = cfg.img_resolution
    R, S = h * h, cfg.num_ray_steps
    sel = np.concatenate([np.arange(r * h, (r + 1) * h) for r in rows])
    rl = lambda: [T(inp['u_coarse']).reshape(1, R, S, 1)]       # noqa: E731
    import time
    with torch.no_grad():
        ws = G.mapping(T(inp['z']), T(inp['c']))
        stage = {}
        rend = G.synthesis.renderer
        o_si = rend.sample_importance

        def si(z_vals, weights, n):
            r = o_si(z_vals, weights, n)
            stage.update(sdist=npy(z_vals)[0, sel, :, 0], weights=npy(weights)[0, sel, :, 0], sdist_fine=npy(r)[0, sel, :, 0])
            return r
        rend.sample_importance = si
        t0 = time.time()
        with PatchedRNG(rand_like=rl(), rand=[T(inp['u_fine'])]), Capture() as cap:
            out = G.synthesis(ws, camera_params=cam, noise_mode='const', render_opts=dict(return_depth=True))
        print(f'{tag}: reference fp32 forward {time.time() - t0:.1f} s')
        rend.sample_importance = o_si
        img, depth = npy(out.img), npy(out.depth)
        arrays = dict(seed=np.array([seed, 1], np.int64), rows=np.array(rows, np.int64), ws=npy(ws), img=img, depth=depth,
                      strip_sdist_coarse=stage['sdist'], strip_weights_coarse=stage['weights'], strip_sdist_fine=stage['sdist_fine'],
                      strip_inds=npy(cap.inds[0]).reshape(R, S)[sel].astype(np.uint8), strip_cdf=npy(cap.cdf[0]).reshape(R, -1)[sel],
                      strip_perm=npy(cap.perm[0]).reshape(R, 2 * S)[sel].astype(np.uint8))
        c2w = ref_ru.compute_cam2world_matrix(cam)
        ro, rd = ref_tpr.sample_rays(c2w, fov=cam.fov, resolution=(h, h))
        arrays.update(c2w=npy(c2w), strip_ray_o=npy(ro)[0, sel], strip_ray_d=npy(rd)[0, sel])
        # tri-planes: too large to store (100 MB); a fixed sample of 4096 texels pins the backbone against the reference directly
        planes = npy(G.synthesis.tri_plane_decoder(ws[:, :G.synthesis.tri_plane_decoder.num_ws], noise_mode='const'))
        pick = np.random.RandomState(seed).randint(0, planes.size, 4096)
        arrays.update(planes_pick=pick.astype(np.int64), planes_vals=planes.reshape(-1)[pick], planes_absmax=np.float32(np.abs(planes).max()))
        # the reference's own fp32 reproducibility (native convolutions, 1 thread) and its float64 run: see gen_e2e
        torch.backends.mkldnn.enabled = False
        torch.set_num_threads(1)
        t0 = time.time()
        with PatchedRNG(rand_like=rl(), rand=[T(inp['u_fine'])]):
            alt = G.synthesis(ws, camera_params=cam, noise_mode='const', render_opts=dict(return_depth=True))
        print(f'{tag}: reference native-conv 1-thread forward {time.time() - t0:.1f} s')
        torch.backends.mkldnn.enabled = True
        torch.set_num_threads(8)
        G64 = build_ref_generator(cfg, sd).double()
        import src.training.networks_epigraf as ref_epi
        c2w32 = ref_ru.compute_cam2world_matrix(cam)
        ro32, rd32 = ref_tpr.sample_rays(c2w32, fov=cam.fov, resolution=(h, h))
        o_c2w, o_rays = ref_epi.compute_cam2world_matrix, ref_epi.sample_rays
        ref_epi.compute_cam2world_matrix = lambda camera_params: c2w32
        ref_epi.sample_rays = lambda *a, **k: (ro32.double(), rd32.double())
        torch.set_default_dtype(torch.float64)
        f32 = torch.float32
        torch.float32 = torch.float64
        t0 = time.time()
        try:
            with PatchedRNG(rand_like=[T(inp['u_coarse']).reshape(1, R, S, 1).double()], rand=[T(inp['u_fine']).double()]):
                o64 = G64.synthesis(ws.double(), camera_params=cam, noise_mode='const', render_opts=dict(return_depth=True))
        finally:
            torch.float32 = f32
            torch.set_default_dtype(torch.float32)
            ref_epi.compute_cam2world_matrix, ref_epi.sample_rays = o_c2w, o_rays
        print(f'{tag}: reference float64 forward {time.time() - t0:.1f} s')
        assert o64.img.dtype == torch.float64
        for key, base, a, b in (('img', img, alt.img, o64.img), ('depth', depth, alt.depth, o64.depth)):
            arrays[key + '_alt_d16'], arrays[key + '_alt_scale'] = _pack_diff(npy(a), base)
            arrays[key + '_f64_d16'], arrays[key + '_f64_scale'] = _pack_diff(npy(b), base)
    save('e2e_full_' + tag, **arrays)


def gen_e2e_full_all():
    for tag in FULL_CONFIGS:
        gen_e2e_full(tag)


def gen_cut_chunked():
    """cut_quantile above max_batch_res in eval: the reference renders through run_batchwise over ray chunks of
    2**24 // (B * num_ray_steps * 3) rays (networks_epigraf.py:232-239), so torch.quantile -- and the random draws -- are taken PER
    CHUNK.  Smallest case that really chunks: 4 x 128^2 rays x 96 steps -> chunks of 14563 rays.  Inputs are regenerated from the seed
    (weights.synthetic_inputs: numpy RandomState), only the outputs are stored."""
    cfg = tdgp.config.config_cut_chunked()
    batch, seed = 4, 51
    sd = tdgp.weights.random_state_dict(cfg, seed=seed, exercise_all=True)
    G = build_ref_generator(cfg, sd)
    inp = tdgp.weights.synthetic_inputs(cfg, batch=batch, seed=seed)
    cam = TensorGroup(**{k: T(v) for k, v in inp['camera'].items()})
    R, S = cfg.img_resolution ** 2, cfg.num_ray_steps
    step = 2 ** 24 // (batch * S * 3)
    assert step < R and cfg.img_resolution > cfg.max_batch_res, (step, R)
    uc, uf = T(inp['u_coarse']).reshape(batch, R, S), T(inp['u_fine']).reshape(batch, R, S)
    likes = [uc[:, a:a + step].reshape(batch, -1, S, 1) for a in range(0, R, step)]
    rands = [uf[:, a:a + step].reshape(-1, S) for a in range(0, R, step)]
    with torch.no_grad():
        ws = G.mapping(T(inp['z']), T(inp['c']))
        with PatchedRNG(rand_like=likes, rand=rands):
            cut = G.synthesis(ws, camera_params=cam, noise_mode='const', render_opts=dict(return_depth=True, cut_quantile=0.5))
    save('cut_chunked', seed=np.array([seed, batch, step], dtype=np.int64), ws=npy(ws), img_cut=npy(cut.img), depth_cut=npy(cut.depth))


class _CudaFlag(torch.Tensor):
    """A tensor that reports device type 'cuda': SynthesisBlock.forward forces fp32 off-GPU (networks_stylegan2.py:235-236) and this
    is the one thing it looks at.  unbind() hands plain tensors on, so nothing downstream sees the flag."""
    @property
    def device(self):
        return types.SimpleNamespace(type='cuda')

    def unbind(self, dim=0):
        return tuple(t.as_subclass(torch.Tensor) for t in super().unbind(dim))


class _Bf16ForFp16:
    """Run the reference's reduced-precision path with bfloat16: its code spells the dtype `torch.float16` (networks_stylegan2.py:51,237)
    and looks it up at call time."""
    def __enter__(self):
        self.f16 = torch.float16
        torch.float16 = torch.bfloat16

    def __exit__(self, *exc):
        torch.float16 = self.f16


def gen_bf16():
    """BASELINE configs[4]'s arithmetic: the reference's own reduced-precision blocks (`use_fp16`, conv_clamp 256) run here on the CPU
    with bfloat16 -- op level (modulated_conv2d on bf16 activations: pre-normalisation, bf16 per-sample weights, bf16 outputs) and the
    whole generator with its two highest-resolution blocks in bf16."""
    g = np.random.RandomState(55)
    arrays = {}
    f = ref_upfirdn2d.setup_filter([1, 3, 3, 1])
    arrays['f'] = npy(f)
    with _Bf16ForFp16():
        for tag, (B, cin, cout, H, k, up, demod, noise) in dict(c3=(2, 16, 24, 12, 3, 1, True, True), up=(2, 16, 8, 8, 3, 2, True, True),
                                                                rgb=(3, 24, 12, 10, 1, 1, False, False)).items():
            x = T(g.randn(B, cin, H, H).astype(np.float32)).bfloat16()
            w = T(g.randn(cout, cin, k, k).astype(np.float32))
            s = T((1.0 + 0.5 * g.randn(B, cin)).astype(np.float32))
            nz = T((0.3 * g.randn(H * up, H * up)).astype(np.float32)) if noise else None
            y = ref_sg2.modulated_conv2d(x=x, weight=w, styles=s, noise=nz, up=up, padding=k // 2, resample_filter=f, demodulate=demod,
                                         flip_weight=(up == 1), fused_modconv=True)
            assert y.dtype == torch.bfloat16
            arrays.update({f'{tag}_x': npy(x.float()), f'{tag}_w': npy(w), f'{tag}_s': npy(s), f'{tag}_y': npy(y.float()),
                           f'{tag}_meta': np.array([k, up, int(demod)])})
            if noise:
                arrays[f'{tag}_noise'] = npy(nz)
            b = T(g.randn(cout).astype(np.float32))
            act = 'lrelu' if demod else 'linear'
            arrays[f'{tag}_b'] = npy(b)
            arrays[f'{tag}_act'] = npy(ref_bias_act.bias_act(y, b.to(y.dtype), act=act, clamp=256).float())
    cfg = tdgp.config.config_mid_bf16()
    sd = tdgp.weights.random_state_dict(cfg, seed=61, exercise_all=True)
    G = Generator(ref_cfg(cfg), img_resolution=cfg.img_resolution, img_channels=3, mapping_kwargs={}, num_fp16_res=cfg.num_fp16_res,
                  conv_clamp=cfg.conv_clamp, fused_modconv_default='inference_only').eval()
    G.load_state_dict({k: T(v) for k, v in sd.items()}, strict=True)
    dec = G.synthesis.tri_plane_decoder
    assert [getattr(dec, f'b{r}').use_fp16 for r in dec.block_resolutions] == [False, False, False, True, True]
    inp = tdgp.weights.synthetic_inputs(cfg, batch=1, seed=62)
    cam = TensorGroup(**{k: T(v) for k, v in inp['camera'].items()})
    R, S = cfg.img_resolution ** 2, cfg.num_ray_steps
    o_fwd = dec.forward
    dec.forward = lambda ws, **kw: o_fwd(ws.as_subclass(_CudaFlag), **kw)
    with torch.no_grad(), _Bf16ForFp16():
        ws = G.mapping(T(inp['z']), T(inp['c']))
        planes = dec(ws, noise_mode='const')
        x = img = None
        w_idx = 0
        wsf = ws.as_subclass(_CudaFlag)
        for r in dec.block_resolutions:                 # per-block activations (bf16 for the last two blocks)
            blk = getattr(dec, f'b{r}')
            x, img = blk(x, img, wsf.narrow(1, w_idx, blk.num_conv + blk.num_torgb), noise_mode='const')
            w_idx += blk.num_conv
            if r >= 16:
                # bf16 tensors are stored as their 16 bits (the upper half of the fp32 pattern)
                arrays[f'x{r}'] = npy(x.float()) if x.dtype == torch.float32 else (npy(x.float()).view(np.uint32) >> 16).astype(np.uint16)
            arrays[f'x{r}_is_bf16'] = np.array(x.dtype == torch.bfloat16)
        with PatchedRNG(rand_like=[T(inp['u_coarse']).reshape(1, R, S, 1)], rand=[T(inp['u_fine'])]):
            out = G.synthesis(ws, camera_params=cam, noise_mode='const', render_opts=dict(return_depth=True))
    dec.forward = o_fwd
    assert planes.dtype == torch.float32 and arrays['x64_is_bf16'] and not arrays['x16_is_bf16']
    arrays.update(z=inp['z'], c=inp['c'], u_coarse=inp['u_coarse'], u_fine=inp['u_fine'], ws=npy(ws), planes=npy(planes), img=npy(out.img), depth=npy(out.depth),
                  **{'cam_' + k: v for k, v in inp['camera'].items()})
    save('bf16', **arrays)


def gen_bf16_full():
    """`bf16_full_c5.npz`: BASELINE configs[4] at its REAL size from the reference's own reduced-precision path run with bfloat16 (as gen_bf16):
    512-channel backbone with the four highest-resolution blocks (64^2 ... 512^2) in bf16, 256^2 rays x 96 + 96 steps, batch 1.  Weights and inputs
    regenerate from the seed; stored: ws, 16384 sampled texels of the fp32 tri-planes (+ their index and the tensor's max), img, depth."""
    import time
    cfg = tdgp.config.config_c5()
    seed = 121
    sd = tdgp.weights.random_state_dict(cfg, seed=seed, exercise_all=True)
    G = Generator(ref_cfg(cfg), img_resolution=cfg.img_resolution, img_channels=3, mapping_kwargs={}, num_fp16_res=cfg.num_fp16_res,
                  conv_clamp=cfg.conv_clamp, fused_modconv_default='inference_only').eval()
    G.load_state_dict({k: T(v) for k, v in sd.items()}, strict=True)
    dec = G.synthesis.tri_plane_decoder
    assert [getattr(dec, f'b{r}').use_fp16 for r in dec.block_resolutions] == [False] * 4 + [True] * 4
    inp = tdgp.weights.synthetic_inputs(cfg, batch=1, seed=seed + 1)
    cam = TensorGroup(**{k: T(v) for k, v in inp['camera'].items()})
    R, S = cfg.img_resolution ** 2, cfg.num_ray_steps
    o_fwd = dec.forward
    dec.forward = lambda ws, **kw: o_fwd(ws.as_subclass(_CudaFlag), **kw)
    t0 = time.time()
    with torch.no_grad(), _Bf16ForFp16():
        ws = G.mapping(T(inp['z']), T(inp['c']))
        planes = dec(ws, noise_mode='const')
        print(f'c5: reference bf16 backbone {time.time() - t0:.1f} s')
        with PatchedRNG(rand_like=[T(inp['u_coarse']).reshape(1, R, S, 1)], rand=[T(inp['u_fine'])]):
            out = G.synthesis(ws, camera_params=cam, noise_mode='const', render_opts=dict(return_depth=True))
    dec.forward = o_fwd
    print(f'c5: reference bf16 forward {time.time() - t0:.1f} s')
    assert planes.dtype == torch.float32
    pl = npy(planes)
    pick = np.random.RandomState(seed).randint(0, pl.size, 16384)
    save('bf16_full_c5', seed=np.array([seed, 1], np.int64), ws=npy(ws), planes_pick=pick.astype(np.int64), planes_vals=pl.reshape(-1)[pick],
         planes_absmax=np.float32(np.abs(pl).max()), planes_absmean=np.float32(np.abs(pl).mean()), img=npy(out.img), depth=npy(out.depth))



def ref_camera_cfg(r):
    mm = lambda t: EasyDict(min=t[0], max=t[1])   # noqa: E731
    return EasyDict(origin=EasyDict(angles=EasyDict(yaw=mm(r.yaw), pitch=mm(r.pitch))), fov=mm(r.fov),
                    look_at=EasyDict(angles=EasyDict(yaw=mm(r.look_at_yaw), pitch=mm(r.look_at_pitch)), radius=mm(r.look_at_radius)))


def gen_adaptors():
    """DepthAdaptor / CameraAdaptor forward (networks_depth_adaptor.py, networks_camera_adaptor.py), eval mode."""
    from src.training.networks_depth_adaptor import DepthAdaptor
    from src.training.networks_camera_adaptor import CameraAdaptor
    arrays = {}
    for tag, cfg in tdgp.config.configs_adaptor_goldens():
        sd = tdgp.weights.random_state_dict(cfg, seed=51, exercise_all=True)
        da, ca = cfg.depth_adaptor, cfg.camera_adaptor
        rda = DepthAdaptor(EasyDict(kernel_size=da.kernel_size, hid_dim=da.hid_dim, num_hid_layers=da.num_hid_layers, out_strategy=da.out_strategy,
                                    near_plane_offset_max_fraction=da.near_plane_offset_max_fraction, near_plane_offset_bias=da.near_plane_offset_bias,
                                    selection_start_p=0.1, anneal_kimg=10000), min_depth=cfg.ray_start, max_depth=cfg.ray_end).eval()
        pfx = 'synthesis.depth_adaptor.'
        rda.load_state_dict({k[len(pfx):]: T(v) for k, v in sd.items() if k.startswith(pfx)}, strict=True)
        g = np.random.RandomState(52)
        B, h = 2, cfg.img_resolution
        depth = g.uniform(cfg.ray_start, cfg.ray_end, (B, 1, h, h)).astype(np.float32)
        w = g.randn(B, cfg.w_dim).astype(np.float32)
        with torch.no_grad():
            x = rda.normalize(T(depth), T(w))
            heads = [x]
            for layer in rda.layers:
                x = layer(x)
                heads.append(rda.head(x))
            out = rda(T(depth), T(w))
        arrays.update({f'{tag}_depth': depth, f'{tag}_w': w, f'{tag}_outs': npy(torch.stack(heads).transpose(0, 1)), f'{tag}_depth_adapted': npy(out)})

        rca = CameraAdaptor(EasyDict(camera=ref_camera_cfg(ca.camera), residual=ca.residual, lr_multiplier=ca.lr_multiplier, z_dim=cfg.z_dim, c_dim=cfg.c_dim,
                                     hid_dim=ca.hid_dim, embed_dim=ca.embed_dim,
                                     adjust=EasyDict(angles=ca.adjust_angles, radius=ca.adjust_radius, fov=ca.adjust_fov, look_at=ca.adjust_look_at))).eval()
        pfx = 'synthesis.camera_adaptor.'
        rca.load_state_dict({k[len(pfx):]: T(v) for k, v in sd.items() if k.startswith(pfx)}, strict=True)
        inp = tdgp.weights.synthetic_inputs(cfg, batch=5, seed=53)
        cam = TensorGroup(**{k: T(v) for k, v in inp['camera'].items()})
        with torch.no_grad():
            new = rca(cam, T(inp['z']), T(inp['c']) if cfg.c_dim > 0 else None)
        arrays.update({f'{tag}_z': inp['z'], f'{tag}_c': inp['c'], **{f'{tag}_cam_{k}': v for k, v in inp['camera'].items()},
                       **{f'{tag}_new_{k}': npy(new[k]) for k in ('angles', 'fov', 'radius', 'look_at')}})
    save('adaptors', **arrays)


def gen_camera_regs():
    """Camera-adaptor regularisers of `learn_camera_dist` (loss.py:142-178 Lipschitz, :224-235 force-mean) on the reference's CameraAdaptor:
    the per-component terms and the gradients they leave in the adaptor's parameters.  The loss lines sit inline in
    StyleGAN2Loss.accumulate_gradients, so the few lines of arithmetic around the adaptor are restated here; the adaptor, its forward,
    the Jacobian through it and roll / unroll are the reference's own.  (The EMD term needs POT, which is not vendored: it is pinned in
    tests/test_training.py against the assignment-problem solution instead.)"""
    from src.training.networks_camera_adaptor import CameraAdaptor
    arrays = {}
    weights = dict(angles=0.7, radius=0.3, fov=1.3, look_at=0.2)
    for tag, cfg in tdgp.config.configs_adaptor_goldens():
        sd = tdgp.weights.random_state_dict(cfg, seed=51, exercise_all=True)
        ca = cfg.camera_adaptor
        rca = CameraAdaptor(EasyDict(camera=ref_camera_cfg(ca.camera), residual=ca.residual, lr_multiplier=ca.lr_multiplier, z_dim=cfg.z_dim, c_dim=cfg.c_dim,
                                     hid_dim=ca.hid_dim, embed_dim=ca.embed_dim,
                                     adjust=EasyDict(angles=ca.adjust_angles, radius=ca.adjust_radius, fov=ca.adjust_fov, look_at=ca.adjust_look_at))).train()
        pfx = 'synthesis.camera_adaptor.'
        rca.load_state_dict({k[len(pfx):]: T(v) for k, v in sd.items() if k.startswith(pfx)}, strict=True)
        inp = tdgp.weights.synthetic_inputs(cfg, batch=24, seed=57)
        z, c = T(inp['z']), (T(inp['c']) if cfg.c_dim > 0 else torch.zeros(24, 0))
        prior = TensorGroup(**{k: T(v) for k, v in inp['camera'].items()})
        prior_raw = rca.unroll_camera_params(prior).requires_grad_(True)
        post_raw = rca.unroll_camera_params(rca(rca.roll_camera_params(prior_raw), z, c))
        grad_i = lambda i: torch.autograd.grad(outputs=[post_raw[:, i].sum()], inputs=[prior_raw], create_graph=True, only_inputs=True)[0][:, i]   # noqa: E731
        norms = torch.stack([grad_i(i) for i in range(post_raw.shape[1])], dim=1).abs()
        regs = (norms + 1.0 / (norms + 1e-4)).mean(dim=0, keepdim=True)
        r = rca.roll_camera_params(regs + regs.max() * 0.0)
        loss_lip = (r.angles * weights['angles'])[:, :2].sum() + (r.radius * weights['radius']).sum() + (r.fov * weights['fov']).sum() + (r.look_at * weights['look_at']).sum()
        rca.zero_grad(set_to_none=True)
        loss_lip.backward()
        arrays.update({f'{tag}_z': inp['z'], f'{tag}_c': inp['c'], **{f'{tag}_cam_{k}': v for k, v in inp['camera'].items()},
                       f'{tag}_jac_diag': npy(norms), f'{tag}_lipschitz_regs': npy(regs), f'{tag}_lipschitz_loss': npy(loss_lip)})
        arrays.update({f'{tag}_lip::{n}': npy(p.grad) for n, p in rca.named_parameters() if p.grad is not None})
        mean_angles = torch.tensor([0.1, 1.5, 0.0])
        post = rca(prior, z, c)
        raw = (post.angles.mean(dim=0) - mean_angles + 1e-8).square().sum().sqrt()
        rca.zero_grad(set_to_none=True)
        (10.0 * raw + 0.0 * post.max()).backward()
        arrays[f'{tag}_force_mean'] = npy(10.0 * raw)
        arrays.update({f'{tag}_fm::{n}': npy(p.grad) for n, p in rca.named_parameters() if p.grad is not None})
    save('camera_regs', **arrays)


def gen_metrics():
    """FeatureStats accumulation (metric_utils.py:104-169) and the camera prior sampler (rendering_utils.py:146-152)."""
    from src.metrics.metric_utils import FeatureStats
    arrays = {}
    g = np.random.RandomState(61)
    feats = [g.randn(70, 16).astype(np.float32) * 3 + 1 for _ in range(3)]
    st = FeatureStats(capture_all=True, capture_mean_cov=True, max_items=200)
    for f in feats:
        st.append(f)
    mean, cov = st.get_mean_cov()
    arrays.update(fs_feats=np.stack(feats), fs_mean=mean, fs_cov=cov, fs_num_items=np.array(st.num_items), fs_all=st.get_all())
    cam = tdgp.metrics.camera_base()
    to_easy = lambda d: EasyDict({k: to_easy(v) if isinstance(v, dict) else v for k, v in d.items()})      # noqa: E731
    for tag, mod in (('base', {}), ('uniform', dict(origin=dict(radius=cam['origin']['radius'], angles=dict(dist='uniform', yaw=dict(min=-1.57, max=1.57),
                                                                                                  pitch=dict(min=0.785398163, max=2.35619449))),
                                                    look_at=dict(radius=dict(dist='uniform', min=0.0, max=0.2), angles=cam['look_at']['angles'])))):
        cfg = {**cam, **mod}
        torch.manual_seed(62)
        np.random.seed(62)
        cp = ref_ru.sample_camera_params(to_easy(cfg), 6, 'cpu')
        arrays.update({f'cam_{tag}_{k}': npy(cp[k]) for k in ('angles', 'fov', 'radius', 'look_at')})
    save('metrics', **arrays)


def gen_trajectories():
    """generate_camera_trajectory (inference_utils.py:140-186) for every trajectory type (deterministic tensor arithmetic)."""
    from src.training import inference_utils as iu
    g = np.random.RandomState(71)
    canon = TensorGroup(angles=T(g.uniform(-1, 1, (3, 3)).astype(np.float32)), fov=T(g.uniform(10, 40, 3).astype(np.float32)),
                        radius=T(np.ones(3, np.float32)), look_at=T(g.uniform(0, 1, (3, 3)).astype(np.float32)))
    arrays = {f'canon_{k}': npy(v) for k, v in canon.items()}
    for name, tr in tdgp.inference_golden_trajectories().items():
        cp = iu.generate_camera_trajectory(EasyDict(tr), canon)
        arrays.update({f'{name}_{k}': np.asarray(npy(v) if isinstance(v, torch.Tensor) else v) for k, v in cp.items()})
    save('trajectories', **arrays)


def gen_train_forward():
    """Training-mode generator forward (networks_epigraf.py:191-194,220-233; no gradients): patch-wise rays at train_resolution,
    density noise from the progressive schedule, W moving average, the depth adaptor's random head selection."""
    arrays = {}
    cfg = tdgp.config.config_train_golden()
    sd = tdgp.weights.random_state_dict(cfg, seed=91, exercise_all=True)
    rc = ref_cfg(cfg)
    rc.patch = EasyDict(enabled=True, resolution=cfg.patch_resolution)
    rc.nerf_noise_std_init, rc.nerf_noise_kimg_growth = cfg.nerf_noise_std_init, cfg.nerf_noise_kimg_growth
    G = Generator(rc, img_resolution=cfg.img_resolution, img_channels=3, mapping_kwargs={}, num_fp16_res=0, conv_clamp=None,
                  fused_modconv_default='inference_only')
    G.load_state_dict({k: T(v) for k, v in sd.items()}, strict=True)
    G.train()
    G.progressive_update(3000)
    arrays['nerf_noise_std'] = np.array(G.synthesis.nerf_noise_std)
    B, h, S = 2, cfg.patch_resolution, cfg.num_ray_steps
    R = h * h
    inp = tdgp.weights.synthetic_inputs(cfg, batch=B, seed=92)
    g = np.random.RandomState(93)
    u_coarse, u_fine = g.rand(B, R, S, 1).astype(np.float32), g.rand(B * R, S).astype(np.float32)
    n_coarse, n_fine = g.randn(B, R * S, 1).astype(np.float32), g.randn(B, R * S, 1).astype(np.float32)
    scales = g.uniform(0.3, 0.8, (B, 2)).astype(np.float32)
    offsets = (g.uniform(0, 1, (B, 2)) * (1 - scales)).astype(np.float32)
    cam = TensorGroup(**{k: T(v) for k, v in inp['camera'].items()})
    with torch.no_grad():
        ws = G.mapping(T(inp['z']), T(inp['c']), update_emas=True)
        arrays['w_avg_after'] = npy(G.mapping.w_avg)
        with PatchedRNG(rand_like=[T(u_coarse)], rand=[T(u_fine)], randn_like=[T(n_coarse), T(n_fine)]):
            out = G.synthesis(ws, camera_params=cam, patch_params=dict(scales=T(scales), offsets=T(offsets)), noise_mode='const',
                              render_opts=dict(return_depth=True))
        torch.backends.mkldnn.enabled = False          # the reference's own fp32 noise floor (see gen_e2e)
        torch.set_num_threads(1)
        with PatchedRNG(rand_like=[T(u_coarse)], rand=[T(u_fine)], randn_like=[T(n_coarse), T(n_fine)]):
            alt = G.synthesis(ws, camera_params=cam, patch_params=dict(scales=T(scales), offsets=T(offsets)), noise_mode='const',
                              render_opts=dict(return_depth=True))
        torch.backends.mkldnn.enabled = True
        torch.set_num_threads(8)
    arrays.update(z=inp['z'], c=inp['c'], ws=npy(ws), u_coarse=u_coarse, u_fine=u_fine, n_coarse=n_coarse, n_fine=n_fine, scales=scales, offsets=offsets,
                  img=npy(out.img), depth=npy(out.depth), img_alt=npy(alt.img), depth_alt=npy(alt.depth),
                  **{'cam_' + k: v for k, v in inp['camera'].items()})
    # depth adaptor, training mode: per-sample random head (numpy RNG) after 4000 kimg of annealing
    from src.training.networks_depth_adaptor import DepthAdaptor
    tag, acfg = tdgp.config.configs_adaptor_goldens()[0]
    asd = tdgp.weights.random_state_dict(acfg, seed=51, exercise_all=True)
    da = acfg.depth_adaptor
    rda = DepthAdaptor(EasyDict(kernel_size=da.kernel_size, hid_dim=da.hid_dim, num_hid_layers=da.num_hid_layers, out_strategy=da.out_strategy,
                                near_plane_offset_max_fraction=da.near_plane_offset_max_fraction, near_plane_offset_bias=da.near_plane_offset_bias,
                                selection_start_p=da.selection_start_p, anneal_kimg=da.anneal_kimg), min_depth=acfg.ray_start, max_depth=acfg.ray_end)
    pfx = 'synthesis.depth_adaptor.'
    rda.load_state_dict({k[len(pfx):]: T(v) for k, v in asd.items() if k.startswith(pfx)}, strict=True)
    rda.train()
    rda.progressive_update(4000)
    g = np.random.RandomState(94)
    depth = g.uniform(acfg.ray_start, acfg.ray_end, (6, 1, acfg.img_resolution, acfg.img_resolution)).astype(np.float32)
    w = g.randn(6, acfg.w_dim).astype(np.float32)
    np.random.seed(95)
    with torch.no_grad():
        out = rda(T(depth), T(w))
    arrays.update(da_depth=depth, da_w=w, da_out=npy(out), da_start_p=np.array(rda.start_p))
    save('train_forward', **arrays)


def gen_synthesis_grad():
    """Autograd through the reference's G.synthesis (eval mode, noise_mode='const', unfused modulated convolutions as in training):
    gradient of sum(img * d_img) + sum(depth * d_depth) w.r.t. every synthesis parameter and ws, tiny configuration."""
    cfg = tdgp.config.config_tiny()
    sd = tdgp.weights.random_state_dict(cfg, seed=101, exercise_all=True)
    G = Generator(ref_cfg(cfg), img_resolution=cfg.img_resolution, img_channels=3, mapping_kwargs={}, num_fp16_res=0, conv_clamp=None,
                  fused_modconv_default=False).eval()
    G.load_state_dict({k: T(v) for k, v in sd.items()}, strict=True)
    B = 2
    inp = tdgp.weights.synthetic_inputs(cfg, batch=B, seed=102)
    cam = TensorGroup(**{k: T(v) for k, v in inp['camera'].items()})
    R, S = cfg.img_resolution ** 2, cfg.num_ray_steps
    g = np.random.RandomState(103)
    d_img, d_depth = g.randn(B, 3, cfg.img_resolution, cfg.img_resolution).astype(np.float32), g.randn(B, 1, cfg.img_resolution, cfg.img_resolution).astype(np.float32)
    with torch.no_grad():
        ws0 = G.mapping(T(inp['z']), T(inp['c']))
    ws = ws0.clone().requires_grad_(True)
    with PatchedRNG(rand_like=[T(inp['u_coarse']).reshape(B, R, S, 1)], rand=[T(inp['u_fine'])]):
        out = G.synthesis(ws, camera_params=cam, noise_mode='const', render_opts=dict(return_depth=True))
    names = [n for n, p in G.synthesis.named_parameters()]
    params = [p for n, p in G.synthesis.named_parameters()]
    grads = torch.autograd.grad([out.img, out.depth], [ws] + params, [T(d_img), T(d_depth)], allow_unused=True)
    arrays = dict(z=inp['z'], c=inp['c'], ws=npy(ws0), u_coarse=inp['u_coarse'], u_fine=inp['u_fine'], d_img=d_img, d_depth=d_depth, img=npy(out.img),
                  depth=npy(out.depth), d_ws=npy(grads[0]), **{'cam_' + k: v for k, v in inp['camera'].items()})
    for n, gr in zip(names, grads[1:]):
        if gr is not None:
            arrays['grad::synthesis.' + n] = npy(gr)
    # round 3: the same loss differentiated w.r.t. the CAMERA (rendering_utils.py:194-218 and tri_plane_renderer.py:487-527 are
    # differentiable in the reference: this is the path loss.py:76-77 trains the camera adaptor through), full image and a patch
    for tag, pp in (('', None), ('_patch', dict(scales=T(np.array([[0.6, 0.7], [0.9, 0.5]], np.float32)), offsets=T(np.array([[0.2, 0.1], [0.05, 0.3]], np.float32))))):
        camg = TensorGroup(**{k: T(v).clone().requires_grad_(True) for k, v in inp['camera'].items()})
        with PatchedRNG(rand_like=[T(inp['u_coarse']).reshape(B, R, S, 1)], rand=[T(inp['u_fine'])]):
            o2 = G.synthesis(ws0, camera_params=camg, patch_params=pp, noise_mode='const', render_opts=dict(return_depth=True))
        keys = ('angles', 'fov', 'radius', 'look_at')
        cg = torch.autograd.grad([o2.img, o2.depth], [camg[k] for k in keys], [T(d_img), T(d_depth)], allow_unused=True)
        for k, gr in zip(keys, cg):
            arrays[f'd_cam{tag}_{k}'] = npy(gr) if gr is not None else np.zeros_like(inp['camera'][k])
        if pp is not None:
            arrays.update(patch_scales=npy(pp['scales']), patch_offsets=npy(pp['offsets']), img_patch=npy(o2.img))
    save('synthesis_grad', **arrays)


D_CASES = dict(plain=dict(cfg=dict(c_dim=0, cbase=256, cmax=16), res=32, img_channels=3, B=4),
               full=dict(cfg=dict(c_dim=5, cbase=256, cmax=16, patch_params_cond=True, hyper_mod=True), res=32, img_channels=4, B=4),
               extra=dict(cfg=dict(c_dim=3, cbase=256, cmax=16, num_additional_start_blocks=1), res=16, img_channels=3, B=3))


def gen_discriminator():
    """Discriminator forward (networks_discriminator.py:259-287, fp32) and the gradient of sum(logits * d) w.r.t. the image and every
    parameter, for: the plain StyleGAN2 form, the 3dgp form (class + patch conditioning, hyper-modulation, RGB-D input) and a form
    with an additional non-down-sampling start block.  Weights: our module's own seeded initialisation, loaded strictly into the
    reference module (key / shape parity)."""
    from src.training.networks_discriminator import Discriminator as RefD
    arrays = {}
    for tag, case in D_CASES.items():
        cfg = tdgp.discriminator.DiscriminatorConfig(**case['cfg'])
        mine = tdgp.discriminator.seeded_discriminator(cfg, case['res'], case['img_channels'], seed=300 + len(tag))
        rcfg = EasyDict(c_dim=cfg.c_dim, cbase=cfg.cbase, cmax=cfg.cmax, fmaps=cfg.fmaps, num_additional_start_blocks=cfg.num_additional_start_blocks,
                        patch=EasyDict(patch_params_cond=cfg.patch_params_cond), hyper_mod=cfg.hyper_mod, camera_cond=False, camera_cond_drop_p=0.0,
                        mbstd_group_size=cfg.mbstd_group_size)
        ref = RefD(rcfg, input_resolution=case['res'], img_channels=case['img_channels'], num_fp16_res=0, conv_clamp=None,
                   epilogue_kwargs=dict(mbstd_group_size=cfg.mbstd_group_size))
        ref.load_state_dict(mine.state_dict(), strict=True)
        g = np.random.RandomState(310 + len(tag))
        B = case['B']
        img = T(g.randn(B, case['img_channels'], case['res'], case['res']).astype(np.float32)).requires_grad_(True)
        c = np.zeros((B, cfg.c_dim), np.float32)
        if cfg.c_dim:
            c[np.arange(B), g.randint(0, cfg.c_dim, B)] = 1.0
        pp = dict(scales=g.uniform(0.3, 1.0, (B, 2)).astype(np.float32), offsets=g.uniform(0.0, 0.5, (B, 2)).astype(np.float32))
        logits, _ = ref(img, T(c), patch_params={k: T(v) for k, v in pp.items()})
        d = T(g.randn(B).astype(np.float32))
        names = [n for n, p in ref.named_parameters()]
        grads = torch.autograd.grad(logits, [img] + [p for n, p in ref.named_parameters()], d, allow_unused=True)
        arrays.update({f'{tag}_img': npy(img), f'{tag}_c': c, f'{tag}_scales': pp['scales'], f'{tag}_offsets': pp['offsets'], f'{tag}_logits': npy(logits),
                       f'{tag}_d': npy(d), f'{tag}_d_img': npy(grads[0])})
        # R1 regularisation (loss.py: r1_grads = grad(real_logits.sum(), real_img, create_graph=True); penalty = r1_grads.square().sum([1,2,3])):
        # the gradient of sum(penalty * e) w.r.t. every parameter is second order in the network
        img2 = T(npy(img)).requires_grad_(True)
        logits2, _ = ref(img2, T(c), patch_params={k: T(v) for k, v in pp.items()})
        r1_grads, = torch.autograd.grad([logits2.sum()], [img2], create_graph=True)
        penalty = r1_grads.square().sum([1, 2, 3])
        e = T(g.rand(B).astype(np.float32))
        r1 = torch.autograd.grad((penalty * e).sum(), [p for n, p in ref.named_parameters()], allow_unused=True)
        arrays.update({f'{tag}_r1_penalty': npy(penalty), f'{tag}_r1_e': npy(e)})
        for n, gr in zip(names, r1):
            if gr is None:
                continue
            if gr.numel() > 20000:
                arrays[f'{tag}::r1rows::{n}'] = npy(gr.sum(dim=1))
                arrays[f'{tag}::r1cols::{n}'] = npy(gr.sum(dim=0))
            else:
                arrays[f'{tag}::r1::{n}'] = npy(gr)
        # weights are NOT stored: the test rebuilds the module from the same seeds.  Gradients of large matrices (the 512-wide
        # hyper-modulation mapping, the 1001 x 256 embedding) are stored as their row and column sums.
        for n, gr in zip(names, grads[1:]):
            if gr is None:
                continue
            if gr.numel() > 20000:
                arrays[f'{tag}::gradrows::{n}'] = npy(gr.sum(dim=1))
                arrays[f'{tag}::gradcols::{n}'] = npy(gr.sum(dim=0))
            else:
                arrays[f'{tag}::grad::{n}'] = npy(gr)
    save('discriminator', **arrays)


def loss_golden_setup():
    """Shared description of the loss golden: tiny generator rendered patch-wise at 16^2, patch-conditioned hyper-modulated
    discriminator, dataset resolution 32."""
    cfg = tdgp.config.config_tiny()
    cfg.use_noise = False
    cfg.patch_resolution = 16
    dcfg = tdgp.discriminator.DiscriminatorConfig(c_dim=0, cbase=256, cmax=16, patch_params_cond=True, hyper_mod=True, mbstd_group_size=2)
    return cfg, dcfg


def gen_loss():
    """StyleGAN2Loss.accumulate_gradients (loss.py:117-330) for the phases Gmain, Dmain, Dreg on tiny networks: the gradients left in
    G / D after each phase, with every random draw (patch parameters, renderer draws) fixed."""
    ot = types.ModuleType('ot')
    sys.modules.setdefault('ot', ot)
    from src.training import loss as ref_loss
    from src.training.networks_discriminator import Discriminator as RefD
    cfg, dcfg = loss_golden_setup()
    B, S, res = 4, cfg.num_ray_steps, cfg.patch_resolution
    R = res * res
    sd = tdgp.weights.random_state_dict(cfg, seed=201, exercise_all=True)
    rc = ref_cfg(cfg)
    rc.patch = EasyDict(enabled=True, resolution=res)
    rc.nerf_noise_std_init, rc.nerf_noise_kimg_growth = 0.0, 1
    G = Generator(rc, img_resolution=cfg.img_resolution, img_channels=3, mapping_kwargs={}, num_fp16_res=0, conv_clamp=None, fused_modconv_default='inference_only')
    G.load_state_dict({k: T(v) for k, v in sd.items()}, strict=True)
    G.train()
    mineD = tdgp.discriminator.seeded_discriminator(dcfg, res, 3, seed=202)
    rd = EasyDict(c_dim=0, cbase=dcfg.cbase, cmax=dcfg.cmax, fmaps=1.0, num_additional_start_blocks=0, patch=EasyDict(patch_params_cond=True), hyper_mod=True,
                  camera_cond=False, camera_cond_drop_p=0.0, mbstd_group_size=2, logits_clamp_val=1e7)
    D = RefD(rd, input_resolution=res, img_channels=3, num_fp16_res=0, conv_clamp=None, epilogue_kwargs=dict(mbstd_group_size=2))
    D.load_state_dict(mineD.state_dict(), strict=True)
    D.train()
    full = EasyDict(model=EasyDict(loss_kwargs=EasyDict(blur_init_sigma=0, blur_fade_kimg=0, adv_loss_type='non_saturating', pl_weight=0.0, pl_start_kimg=0,
                                                       kd=EasyDict(discr=EasyDict(weight=0.0, anneal_kimg=1, loss_type='l2'))),
                               generator=EasyDict(camera_cond_spoof_p=0.5), discriminator=rd),
                    training=EasyDict(patch=EasyDict(enabled=True, distribution='uniform', min_scale_trg=0.5, max_scale=1.0, anneal_kimg=10, resolution=res,
                                                     mbstd_group_size=2, patch_params_cond=True),
                                      learn_camera_dist=False, use_depth=False, blur_real_depth_sigma=0.0))
    loss = ref_loss.StyleGAN2Loss(full, 'cpu', G, D, augment_pipe=None, r1_gamma=2.0)
    g = np.random.RandomState(203)
    inp = tdgp.weights.synthetic_inputs(cfg, batch=B, seed=204)
    u1, u2 = g.rand(B, R, S, 1).astype(np.float32), g.rand(B * R, S).astype(np.float32)
    real = g.randn(B, 3, 32, 32).astype(np.float32)
    sx = g.uniform(0.5, 1.0, (3, B // 2)).astype(np.float32)
    pps = []
    for i in range(3):
        sc = np.repeat(np.stack([sx[i], sx[i]], 1), 2, axis=0)
        pps.append(dict(scales=sc, offsets=(np.repeat(g.rand(B // 2, 2).astype(np.float32), 2, axis=0) * (1 - sc)).astype(np.float32)))
    arrays = dict(z=inp['z'], u_coarse=u1, u_fine=u2, real=real, **{'cam_' + k: v for k, v in inp['camera'].items()})
    for i, pp in enumerate(pps):
        arrays[f'pp{i}_scales'], arrays[f'pp{i}_offsets'] = pp['scales'], pp['offsets']
    queue = []
    ref_loss.sample_patch_params = lambda n, pcfg, device='cpu': {k: T(v) for k, v in queue.pop(0).items()}
    cam = TensorGroup(**{k: T(v) for k, v in inp['camera'].items()})
    c0 = torch.zeros(B, 0)

    def run(phase, pp_list, n_render):
        for m in (G, D):
            m.zero_grad(set_to_none=True)
        G.requires_grad_(phase.startswith('G'))
        D.requires_grad_(phase.startswith('D'))
        queue[:] = pp_list
        real_data = TensorGroup(img=T(real), c=c0, depth=torch.zeros(B, 1, 32, 32), camera_angles=cam.angles)
        gen_data = TensorGroup(z=T(inp['z']), c=c0, camera_params=cam, camera_angles_cond=cam.angles)
        with PatchedRNG(rand_like=[T(u1)] * n_render, rand=[T(u2)] * n_render):
            loss.accumulate_gradients(phase=phase, real_data=real_data, gen_data=gen_data, gain=1, cur_nimg=0)
        assert not queue

    run('Gmain', [pps[0]], 1)
    for n, p in G.named_parameters():
        if p.grad is not None:
            arrays[f'Gmain::{n}'] = npy(p.grad)
    run('Dmain', [pps[0], pps[1]], 1)
    run2 = {n: npy(p.grad) for n, p in D.named_parameters() if p.grad is not None}
    run('Dreg', [pps[2]], 0)
    run3 = {n: npy(p.grad) for n, p in D.named_parameters() if p.grad is not None}
    for tag, grads in (('Dmain', run2), ('Dreg', run3)):
        for n, v in grads.items():
            if v.size > 20000:
                arrays[f'{tag}::rows::{n}'], arrays[f'{tag}::cols::{n}'] = v.sum(1), v.sum(0)
            else:
                arrays[f'{tag}::{n}'] = v
    # patch sampling / extraction / blur (training_utils.py:22-143, loss.py:332-338) with seeded RNGs
    from src.training import training_utils as ref_tu
    for dist, extra in (('uniform', {}), ('beta', dict(alpha=1.0, beta=0.4))):
        pc = EasyDict(distribution=dist, min_scale=0.3, max_scale=0.9, mbstd_group_size=2, **extra)
        np.random.seed(7)
        torch.manual_seed(7)
        pp = ref_tu.sample_patch_params(8, pc, device='cpu')
        arrays[f'sp_{dist}_scales'], arrays[f'sp_{dist}_offsets'] = npy(pp['scales']), npy(pp['offsets'])
    arrays['patches'] = npy(ref_tu.extract_patches(T(real), {k: T(v) for k, v in pps[0].items()}, resolution=16))
    arrays['blurred'] = npy(ref_loss.maybe_blur(T(real), 1.3))
    save('loss', **arrays)


def gen_loss_kd():
    """The discriminator's knowledge-distillation term (loss.py:279-314) through the reference's own StyleGAN2Loss: phase Dmain with
    kd.discr.weight 0.7 at kimg 0, l2 and kl distances, patch-size sample weights -- the gradients left in D (feature head included)."""
    ot = types.ModuleType('ot')
    sys.modules.setdefault('ot', ot)
    from src.training import loss as ref_loss
    from src.training.networks_discriminator import Discriminator as RefD
    cfg, dcfg = loss_golden_setup()
    B, S, res = 4, cfg.num_ray_steps, cfg.patch_resolution
    R = res * res
    FD = 6
    sd = tdgp.weights.random_state_dict(cfg, seed=201, exercise_all=True)
    rc = ref_cfg(cfg)
    rc.patch = EasyDict(enabled=True, resolution=res)
    rc.nerf_noise_std_init, rc.nerf_noise_kimg_growth = 0.0, 1
    G = Generator(rc, img_resolution=cfg.img_resolution, img_channels=3, mapping_kwargs={}, num_fp16_res=0, conv_clamp=None, fused_modconv_default='inference_only')
    G.load_state_dict({k: T(v) for k, v in sd.items()}, strict=True)
    G.train()
    mineD = tdgp.discriminator.seeded_discriminator(dcfg, res, 3, seed=212, epilogue_kwargs=dict(feat_predict_dim=FD))
    rd = EasyDict(c_dim=0, cbase=dcfg.cbase, cmax=dcfg.cmax, fmaps=1.0, num_additional_start_blocks=0, patch=EasyDict(patch_params_cond=True), hyper_mod=True,
                  camera_cond=False, camera_cond_drop_p=0.0, mbstd_group_size=2, logits_clamp_val=1e7)
    D = RefD(rd, input_resolution=res, img_channels=3, num_fp16_res=0, conv_clamp=None, epilogue_kwargs=dict(mbstd_group_size=2, feat_predict_dim=FD))
    D.load_state_dict(mineD.state_dict(), strict=True)
    D.train()
    g = np.random.RandomState(213)
    inp = tdgp.weights.synthetic_inputs(cfg, batch=B, seed=204)
    u1, u2 = g.rand(B, R, S, 1).astype(np.float32), g.rand(B * R, S).astype(np.float32)
    real = g.randn(B, 3, 32, 32).astype(np.float32)
    embs = g.randn(B, FD).astype(np.float32)
    sx = g.uniform(0.5, 1.0, (2, B // 2)).astype(np.float32)
    pps = []
    for i in range(2):
        sc = np.repeat(np.stack([sx[i], sx[i]], 1), 2, axis=0)
        pps.append(dict(scales=sc, offsets=(np.repeat(g.rand(B // 2, 2).astype(np.float32), 2, axis=0) * (1 - sc)).astype(np.float32)))
    arrays = dict(z=inp['z'], u_coarse=u1, u_fine=u2, real=real, embs=embs, **{'cam_' + k: v for k, v in inp['camera'].items()})
    for i, pp in enumerate(pps):
        arrays[f'pp{i}_scales'], arrays[f'pp{i}_offsets'] = pp['scales'], pp['offsets']
    queue = []
    ref_loss.sample_patch_params = lambda n, pcfg, device='cpu': {k: T(v) for k, v in queue.pop(0).items()}
    cam = TensorGroup(**{k: T(v) for k, v in inp['camera'].items()})
    c0 = torch.zeros(B, 0)
    for kind in ('l2', 'kl'):
        full = EasyDict(model=EasyDict(loss_kwargs=EasyDict(blur_init_sigma=0, blur_fade_kimg=0, adv_loss_type='non_saturating', pl_weight=0.0, pl_start_kimg=0,
                                                           kd=EasyDict(discr=EasyDict(weight=0.7, anneal_kimg=100, loss_type=kind))),
                                   generator=EasyDict(camera_cond_spoof_p=0.5), discriminator=rd),
                        training=EasyDict(patch=EasyDict(enabled=True, distribution='uniform', min_scale_trg=0.5, max_scale=1.0, anneal_kimg=10, resolution=res,
                                                         mbstd_group_size=2, patch_params_cond=True),
                                          learn_camera_dist=False, use_depth=False, blur_real_depth_sigma=0.0))
        loss = ref_loss.StyleGAN2Loss(full, 'cpu', G, D, augment_pipe=None, r1_gamma=2.0)
        loss.progressive_update(25)                                  # a quarter into the fade-out: weight 0.525
        for m in (G, D):
            m.zero_grad(set_to_none=True)
        G.requires_grad_(False)
        D.requires_grad_(True)
        queue[:] = [pps[0], pps[1]]
        real_data = TensorGroup(img=T(real), c=c0, depth=torch.zeros(B, 1, 32, 32), camera_angles=cam.angles, embs=T(embs))
        gen_data = TensorGroup(z=T(inp['z']), c=c0, camera_params=cam, camera_angles_cond=cam.angles)
        with PatchedRNG(rand_like=[T(u1)], rand=[T(u2)]):
            loss.accumulate_gradients(phase='Dmain', real_data=real_data, gen_data=gen_data, gain=1, cur_nimg=0)
        assert not queue
        arrays[f'{kind}_kd_weight'] = np.float32(loss.D_kd_weight)
        for n, p in D.named_parameters():
            if p.grad is not None:
                v = npy(p.grad)
                if v.size > 20000:
                    arrays[f'{kind}::rows::{n}'], arrays[f'{kind}::cols::{n}'] = v.sum(1), v.sum(0)
                else:
                    arrays[f'{kind}::{n}'] = v
    save('loss_kd', **arrays)


class _GoldenDataset:
    """Stand-in for the reference's ImageFolder dataset in iterate_random_conditioning: labels and camera angles are pure
    functions of the item index."""

    def __init__(self, c_dim=5, size=11):
        self.c_dim, self.size = c_dim, size

    def __len__(self):
        return self.size

    def get_label(self, i):
        v = np.zeros(self.c_dim, np.float32)
        v[i % self.c_dim] = 1.0
        return v

    def get_camera_angles(self, i):
        return np.array([0.1 * i - 0.5, 1.2 + 0.03 * i, 0.0], np.float32)


def gen_harness():
    """seeds -> ws (scripts/inference.py:87-150) and the conditioning iterator of the metric loops (metric_utils.py:60-101)."""
    hy = types.ModuleType('hydra')
    hy.main = lambda **kw: (lambda f: f)
    sys.modules.setdefault('hydra', hy)
    sys.modules['torchvision.utils'].make_grid = None
    ds = types.ModuleType('torchvision.datasets')
    ds.__path__ = []
    ds.VisionDataset = object
    fo = types.ModuleType('torchvision.datasets.folder')
    fo.pil_loader = None
    sys.modules.setdefault('torchvision.datasets', ds)
    sys.modules.setdefault('torchvision.datasets.folder', fo)
    sys.modules['torchvision'].datasets = ds
    from scripts import inference as ref_inf
    from src.metrics import metric_utils as mu
    arrays = {}
    cfg = tdgp.config.config_mid()
    sd = tdgp.weights.random_state_dict(cfg, seed=11, exercise_all=True)
    G = build_ref_generator(cfg, sd)
    seeds = [3, 17, 17, 2024, 5, 8]
    arrays['seeds'] = np.array(seeds)
    arrays['classes'] = np.array([1, 4])
    arrays['z'] = npy(ref_inf.sample_z_from_seeds(seeds, G.z_dim))
    arrays['c'] = npy(ref_inf.sample_c_from_seeds(seeds, G.c_dim))
    with torch.no_grad():
        for tag, kw in (('psi1', dict(psi=1.0)), ('psi06', dict(psi=0.6)), ('psi06_cls', dict(psi=0.6, classes=[1, 4])),
                        ('psi1_cls', dict(psi=1.0, classes=[1, 4])), ('interp', dict(psi=0.8, num_interp_steps=5))):
            torch.manual_seed(81)
            ws, z, c = ref_inf.sample_ws_from_seeds(G, seeds, EasyDict(truncation_psi=kw['psi']), 'cpu', num_interp_steps=kw.get('num_interp_steps', 0),
                                                    classes=kw.get('classes'))
            arrays[f'ws_{tag}'] = npy(ws)
            if not isinstance(z, tuple):
                arrays[f'z_{tag}'], arrays[f'c_{tag}'] = npy(z), npy(c)
    # the conditioning iterator: conditional generator (labels from the dataset), custom angles, frontal camera, unconditional prior
    cam = tdgp.metrics.camera_base()
    to_easy = lambda d: EasyDict({k: to_easy(v) if isinstance(v, dict) else v for k, v in d.items()})      # noqa: E731
    custom = {**cam, 'origin': dict(radius=cam['origin']['radius'], angles=dict(dist='custom'))}
    torch.Tensor.pin_memory = lambda self: self      # no GPU in the build container: page-locking is a no-op for the values
    for tag, c_dim, camcfg, frontal in (('cond', 5, cam, False), ('custom', 5, custom, False), ('frontal', 5, cam, True), ('uncond', 0, cam, False),
                                        ('uncond_custom', 0, custom, False)):
        opts = EasyDict(G=EasyDict(c_dim=c_dim, cfg=EasyDict(camera=to_easy(camcfg))), device='cpu',
                        dataset_kwargs=EasyDict(class_name='__main__._GoldenDataset', c_dim=max(c_dim, 1), size=11))
        torch.manual_seed(82)
        np.random.seed(82)
        it = mu.iterate_random_conditioning(opts, 4, frontal_camera=frontal)
        for step in range(2):
            c, cp = next(it)
            arrays[f'it_{tag}_{step}_c'] = npy(c)
            arrays.update({f'it_{tag}_{step}_{k}': npy(cp[k]) for k in ('angles', 'fov', 'radius', 'look_at')})
    save('harness', **arrays)


def main():
    torch.set_num_threads(8)
    if len(sys.argv) > 1:                      # regenerate selected files only: python tools/gen_goldens.py adaptors e2e
        for name in sys.argv[1:]:
            if name == 'e2e':
                gen_e2e_all()
            elif name == 'e2e_full':
                gen_e2e_full_all()
            elif name.startswith('e2e_full_'):
                gen_e2e_full(name[len('e2e_full_'):])
            else:
                globals()['gen_' + name]()
        return
    gen_adaptors()
    gen_camera_regs()
    gen_metrics()
    gen_trajectories()
    gen_harness()
    gen_synthesis_grad()
    gen_discriminator()
    gen_loss()
    gen_loss_kd()
    gen_train_forward()
    gen_bias_act()
    gen_bias_act_grad()
    gen_upfirdn2d()
    gen_upfirdn2d_grad()
    gen_conv2d_grad()
    gen_march_grad()
    gen_field_grad()
    gen_render_grad()
    gen_modconv_grad()
    gen_modconv()
    gen_field()
    gen_sampling()
    gen_sampling_hot()
    gen_bf16()
    gen_bf16_full()
    gen_marchers()
    gen_camera()
    gen_mapping()
    gen_cut_chunked()
    gen_e2e_all()
    gen_e2e_full_all()


def gen_e2e_all():
    gen_e2e('e2e_tiny', tdgp.config.config_tiny(), batch=2, seed=21, keep_intermediates=True)
    gen_e2e('e2e_mid', tdgp.config.config_mid(), batch=2, seed=31, keep_intermediates=False)
    cfg = tdgp.config.config_tiny()
    cfg.ray_marcher_type = 'mip'
    cfg.white_back = True
    gen_e2e('e2e_tiny_mip', cfg, batch=1, seed=41, keep_intermediates=False)
    gen_e2e('e2e_bigger', tdgp.config.config_bigger(), batch=2, seed=5, keep_intermediates=False)


if __name__ == '__main__':
    main()
