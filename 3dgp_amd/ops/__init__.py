"""Op API of the hot path -- same module names, function names, argument order and defaults as the
reference's `src/torch_utils/ops` package (bias_act, upfirdn2d, conv2d_resample) plus the ops the
reference leaves in eager PyTorch (modulated_conv2d).  `impl='cuda'` dispatches to the gfx950 HIP
kernels through the C ABI; `impl='ref'` is a plain-PyTorch restatement kept for API parity."""
from . import bias_act, upfirdn2d, modconv, conv2d_resample, conv2d_gradfix, fma  # noqa: F401
