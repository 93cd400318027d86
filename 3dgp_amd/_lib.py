"""ctypes binding of libtdgp_hip.so (the C ABI declared in include/tdgp.h).

This is the "plugin loader" of the build: the counterpart of the reference's
`custom_ops.get_plugin()` (src/torch_utils/custom_ops.py:59-155), except that nothing is compiled
at run time -- the prebuilt gfx950 library is dlopen'ed.  If it is missing or fails to load the
product path raises; there is no CPU or PyTorch fallback behind these entry points.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libtdgp_hip.so')     # the one library of the product; tests / tools/dev rebind this attribute before load()

P = c_void_p
_PROTOTYPES = {
    'tdgp_version': (c_int, []),
    'tdgp_last_error': (c_char_p, []),
    'tdgp_profile_enable': (c_int, [c_int]),
    'tdgp_profile_report': (c_int64, [c_char_p, c_int64]),
    'tdgp_device_fault': (c_int, [c_int]),
    'tdgp_bias_act': (c_int, [P, P, P, c_int64, c_int, c_int64, c_int, c_float, c_float, c_float, c_int, P]),
    'tdgp_bias_act_grad': (c_int, [P, P, P, P, P, P, c_int64, c_int, c_int64, c_int, c_int, c_float, c_float, c_float, c_int, P]),
    'tdgp_upfirdn2d': (c_int, [P, P, P, c_int, c_int, c_int, c_int, POINTER(c_int64), c_int, c_int, POINTER(c_int64),
                               c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, P]),
    'tdgp_modconv_pack_bytes': (c_int64, [c_int, c_int, c_int]),
    'tdgp_modconv_pack': (c_int, [P, P, c_int, c_int, c_int, P]),
    'tdgp_modconv2d_workspace_bytes': (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    'tdgp_modconv2d_takes_folded_up2': (c_int, [c_int, c_int, c_int, c_int, c_int]),
    'tdgp_conv2d': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'tdgp_conv2d_weight_grad_workspace_bytes': (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    'tdgp_conv2d_weight_grad': (c_int, [P, P, P, P, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'tdgp_conv_transpose2d_x2': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, c_int64, P]),
    'tdgp_set_conv_arith': (c_int, [c_int]),
    'tdgp_modconv_wsq_offset': (c_int64, [c_int, c_int, c_int]),
    'tdgp_demod_batch': (c_int, [P, P, P, c_int, c_int, c_int, P]),
    'tdgp_modconv2d': (c_int, [P, P, P, P, P, c_int64, P, POINTER(c_float), P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                               c_int, c_int, c_float, c_float, c_float, c_int, c_int, P, c_int64, P]),
    'tdgp_modconv2d_bf16': (c_int, [P, P, P, P, P, c_int64, P, POINTER(c_float), P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                    c_int, c_int, c_float, c_float, c_float, c_int, c_int, P, c_int64, P]),
    'tdgp_cast_f32_bf16': (c_int, [P, P, c_int64, P]),
    'tdgp_style_affine': (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    'tdgp_cam2world': (c_int, [P, P, P, P, c_int, P]),
    'tdgp_sample_rays': (c_int, [P, P, c_int, P, P, P, P, c_int, c_int, c_int, P]),
    'tdgp_sample_stratified': (c_int, [P, P, P, c_int64, c_int, c_int, c_float, c_float, P]),
    'tdgp_density_activation': (c_int, [P, P, c_int64, c_int, c_float, P]),
    'tdgp_planes_to_hwc': (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    'tdgp_triplane_features': (c_int, [P, P, P, c_int, c_int64, c_int, c_int, c_int, c_float, P]),
    'tdgp_triplane_field': (c_int, [P, P, P, P, P, P, P, P, P, P, c_float, P, P, c_int, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                                    c_int, P]),
    'tdgp_ray_march': (c_int, [P, P, P, P, P, P, P, c_int64, c_int, c_int, c_int, c_int, c_float, c_float, P]),
    'tdgp_triplane_field_grad_workspace_bytes': (c_int64, [c_int, c_int64, c_int, c_int]),
    'tdgp_triplane_field_grad': (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, c_int64, c_int, c_int64, c_int, c_int, c_int, c_int, c_float, c_int, P]),
    'tdgp_ray_march_grad': (c_int, [P, P, P, P, P, P, P, P, c_int64, c_int, c_int, c_int, c_int, c_float, P]),
    'tdgp_sample_importance': (c_int, [P, P, P, P, P, P, P, P, c_int64, c_int, c_int, c_int, c_int, P]),
    'tdgp_unify_samples': (c_int, [P, P, P, c_int, P, P, P, c_int, P, P, P, P, c_int64, c_int, P]),
    'tdgp_importance_from_coarse': (c_int, [P, P, P, P, P, P, P, c_int64, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, P]),
    'tdgp_merge_composite': (c_int, [P, P, c_int, P, P, c_int, P, P, P, P, P, P, c_int64, c_int, c_int, c_float, c_float, P]),
    'tdgp_render_fused_workspace_bytes': (c_int64, [c_int, c_int64, c_int, c_int]),
    'tdgp_render_fused': (c_int, [P] * 13 + [c_int, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_int, c_int, c_float, P, c_int64, P]),
    'tdgp_rays_to_image': (c_int, [P, P, c_int, c_int, P]),
}
EXPORTS = tuple(_PROTOTYPES)

_lib = None
_load_error = None


def load():
    """dlopen the prebuilt library (once).  Raises RuntimeError if it is missing or broken."""
    global _lib, _load_error
    if _lib is not None:
        return _lib
    if _load_error is not None:
        raise RuntimeError(_load_error)
    if not os.path.exists(LIB_PATH):
        _load_error = (f'{LIB_PATH} not found: the gfx950 HIP library is not built. Run `python -c "import __graft_entry__ as g; '
                       f'g.build()"` (hipcc) first. There is no CPU/PyTorch fallback for the native ops.')
        raise RuntimeError(_load_error)
    try:
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
    except (OSError, AttributeError) as e:
        _load_error = f'failed to load {LIB_PATH}: {e}'
        raise RuntimeError(_load_error) from e
    _lib = lib
    return _lib


def available():
    try:
        load()
        return True
    except RuntimeError:
        return False


def ptr(t):
    """Device pointer of a tensor (or None)."""
    return None if t is None else t.data_ptr()


def stream_of(t):
    return torch.cuda.current_stream(t.device).cuda_stream


class Unsupported(RuntimeError):
    """TDGP_EUNSUPPORTED: a valid request this build has no kernel for (include/tdgp.h).  Returned before anything is launched."""


def call(name, *args):
    """Call an entry point; non-zero return -> RuntimeError carrying tdgp_last_error()."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.tdgp_last_error()
        raise (Unsupported if rc == -2 else RuntimeError)(f'{name} failed ({rc}): {msg.decode() if msg else "?"}')


def require_cuda(t, what):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError(f'{what} must reside on the GPU (got {getattr(t, "device", type(t))}); the HIP ops have no CPU path')


def f32c(t):
    """Contiguous fp32 view/copy."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def profile_enable(on=True):
    """Start (and clear) / stop per-kernel HIP-event timing inside the library."""
    load().tdgp_profile_enable(int(bool(on)))


def profile_report():
    """-> {kernel: dict(launches, total_ms, avg_ms, min_ms, max_ms)}; blocks until the recorded launches finished."""
    lib = load()
    need = lib.tdgp_profile_report(None, 0)
    buf = ctypes.create_string_buffer(int(need) + 16)
    lib.tdgp_profile_report(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, n, tot, mn, mx = line.split()
        out[name] = dict(launches=int(n), total_ms=float(tot), avg_ms=float(tot) / max(int(n), 1), min_ms=float(mn), max_ms=float(mx))
    return out


def device_fault(clear=False):
    """The library's device-fault word (include/tdgp.h: a bounded in-kernel wait that ran out); read it after synchronising."""
    return int(load().tdgp_device_fault(int(bool(clear))))


def raise_on_device_fault(where=''):
    """Call AFTER a synchronisation point (an image fetched to the host, the end of a training step, a graph replay whose outputs were read):
    a launch whose bounded in-kernel wait ran out has set the fault word -- its results are invalid.  Reads, clears and raises; the C side
    only notices at the entry of the NEXT field / per-ray call, which the last forward of a run never reaches."""
    code = device_fault(clear=True)
    if code:
        raise RuntimeError(f'libtdgp_hip: device fault {code} reported{" in " + where if where else ""}: a bounded in-kernel wait ran out, '
                           'the results of that launch are invalid (the fault word has been cleared)')


def set_conv_arith(mode):
    """0 (default): fp32 MFMA, Winograd F(4x4,3x3) / F(2x2,3x3) on the layers where they pay; 2: direct sums only; 3: as 0 without F(4x4); 1: split-bf16 (3 x bf16 pieces per operand, 6 piece products, fp32 accumulation) for the 3x3 stride-1
    convolutions with W % 32 == 0 and >= 256 output tiles.  Process-wide; returns the previous mode."""
    lib = load()
    prev = int(lib.tdgp_set_conv_arith(int(mode)))
    if prev < 0:
        msg = lib.tdgp_last_error()
        raise RuntimeError(f'tdgp_set_conv_arith failed ({prev}): {msg.decode() if msg else "?"}')
    return prev
