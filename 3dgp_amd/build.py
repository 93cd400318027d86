"""Build libtdgp_hip.so (gfx950) in-tree with hipcc.  No JIT at import time, no hipify, no Triton.

The reference JIT-compiles its CUDA plugins on first use through torch.utils.cpp_extension
(src/torch_utils/custom_ops.py:59-155).  Here the library is built ahead of time by
`__graft_entry__.build()` (hipcc cross-compiles without a GPU) and travels with the tree.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(CSRC, 'libtdgp_hip.so')
SOURCES = ['core.hip', 'bias_act.hip', 'upfirdn2d.hip', 'modconv.hip', 'conv_grad.hip', 'camera_rays.hip', 'field.hip', 'sampling.hip', 'render_grad.hip']
HEADERS = ['common.h', 'modconv_bf16.inc', 'modconv_wino.inc', 'modconv_wino4.inc', 'field_walk2.inc', os.path.join('..', '..', 'include', 'tdgp.h')]
# -ffp-contract=off: fp32 chains that decide integer rows must round like the reference's eager ops;
# fused multiply-adds are written explicitly (fmaf_) where wanted.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-fvisibility=hidden',
         '-Wno-unused-result']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def verify_field_isa(asm_path):
    """Run isa_check over the assembly of field.hip; raises (and removes the listing, so the next build checks again) on a violation."""
    sys.path.insert(0, HERE)
    try:
        import isa_check
    finally:
        sys.path.pop(0)
    res = isa_check.check_walk2_asm(open(asm_path).read())
    bad = [(k, why, ins) for k, (_, v) in res.items() for why, ins in v]
    if not res or bad:
        os.remove(asm_path)
        raise RuntimeError('ISA check of triplane_walk2_kernel failed (3dgp_amd/isa_check.py): ' +
                           ('no walk2 instantiation found in the listing' if not res else '; '.join(f'{k}: {why} | {ins}' for k, why, ins in bad[:8])))
    return res


def build_native(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link libtdgp_hip.so.  Returns the library path."""
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, 'build')
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace('.hip', '.o'))
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + FLAGS + ['-c', s, '-o', o]
        if verbose:
            print(' '.join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed for {s}:\n{r.stderr}')
        return o

    # field.hip carries hand-issued loads whose safety is a property of the generated code (isa_check.py): whenever it is recompiled its
    # assembly is produced next to the object and checked; a violation fails the build.
    field_src, field_asm = os.path.join(CSRC, 'field.hip'), os.path.join(objdir, 'field.s')
    check_field = any(s == field_src for s, _ in jobs) or _stale(field_asm, [field_src] + hdrs)

    def asm_one(_):
        r = subprocess.run([hipcc] + FLAGS + ['-S', '--cuda-device-only', '-o', field_asm, field_src], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc -S failed for {field_src}:\n{r.stderr}')

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs) + 1))) as ex:
        fut = ex.submit(asm_one, None) if check_field else None
        list(ex.map(compile_one, jobs))
        if fut is not None:
            fut.result()
    if check_field:
        verify_field_isa(field_asm)
    objs = [os.path.join(objdir, s.replace('.hip', '.o')) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stderr}')
    return LIB


if __name__ == '__main__':
    print(build_native(force='--force' in sys.argv, verbose=True))
