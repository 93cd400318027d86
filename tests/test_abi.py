"""The C-ABI library: builds for gfx950, loads without a GPU, exports every symbol include/tdgp.h declares,
and the product never reaches into oracle/ (no compute calls here)."""
import os
import re
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(REPO, 'include', 'tdgp.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(tdgp_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_entry_points():
    syms = header_symbols()
    assert 'tdgp_bias_act' in syms and 'tdgp_upfirdn2d' in syms and 'tdgp_modconv2d' in syms and 'tdgp_triplane_field' in syms
    assert len(syms) >= 20


def test_library_builds_loads_and_exports(tdgp):
    lib_path = tdgp.build.build_native()
    assert os.path.exists(lib_path)
    lib = tdgp._lib.load()
    assert lib.tdgp_version() >= 100
    for s in header_symbols():
        assert hasattr(lib, s), f'{s} declared in include/tdgp.h but not exported'
    assert sorted(tdgp._lib.EXPORTS) == header_symbols()
    out = subprocess.run(['nm', '-D', '--defined-only', lib_path], capture_output=True, text=True).stdout
    exported = sorted(set(re.findall(r' T (tdgp_\w+)', out)))
    assert exported == header_symbols()


def test_library_is_gfx950_only(tdgp):
    lib_path = tdgp.build.build_native()
    data = open(lib_path, 'rb').read()
    assert b'gfx950' in data
    for other in (b'gfx942', b'gfx90a', b'sm_80', b'sm_90'):
        assert other not in data


def test_ops_refuse_cpu_tensors_on_native_only_paths(tdgp):
    import torch
    x = torch.zeros(1, 3, 4, 4)
    with pytest.raises(RuntimeError, match='GPU'):
        tdgp.ops.modconv.modulated_conv2d(x, torch.zeros(2, 3, 3, 3), torch.ones(1, 3), padding=1)
    with pytest.raises(RuntimeError, match='GPU'):
        tdgp.renderer.simple_tri_plane_renderer(torch.zeros(1, 24, 4, 4), torch.zeros(1, 5, 3), tdgp.renderer.TriPlaneMLP(8, 16))


def test_product_never_imports_the_oracle():
    bad = []
    for root, _, files in os.walk(os.path.join(REPO, '3dgp_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(root, f), errors='ignore').read()
                if re.search(r'^\s*(import|from)\s+oracle\b', src, flags=re.M) or 'tdgp_oracle' in src or 'orc_' in src:
                    bad.append(os.path.join(root, f))
    assert not bad, bad


def test_reference_api_surface(tdgp):
    """Same names / defaults as src.torch_utils.ops.{bias_act,upfirdn2d,conv2d_resample} and the renderer callables."""
    import inspect
    sig = lambda f: list(inspect.signature(f).parameters)   # noqa: E731
    assert sig(tdgp.ops.bias_act.bias_act) == ['x', 'b', 'dim', 'act', 'alpha', 'gain', 'clamp', 'impl']
    assert sig(tdgp.ops.upfirdn2d.upfirdn2d) == ['x', 'f', 'up', 'down', 'padding', 'flip_filter', 'gain', 'impl']
    assert sig(tdgp.ops.upfirdn2d.upsample2d) == ['x', 'f', 'up', 'padding', 'flip_filter', 'gain', 'impl']
    assert sig(tdgp.ops.upfirdn2d.setup_filter) == ['f', 'device', 'normalize', 'flip_filter', 'gain', 'separable']
    assert sig(tdgp.ops.conv2d_resample.conv2d_resample) == ['x', 'w', 'f', 'up', 'down', 'padding', 'groups', 'flip_weight', 'flip_filter']
    assert sig(tdgp.ops.modconv.modulated_conv2d) == ['x', 'weight', 'styles', 'noise', 'up', 'down', 'padding', 'resample_filter', 'demodulate',
                                                      'flip_weight', 'fused_modconv']
    assert sig(tdgp.renderer.sample_rays) == ['c2w', 'fov', 'resolution', 'patch_params', 'device']
    assert sig(tdgp.renderer.simple_tri_plane_renderer)[:4] == ['x', 'coords', 'mlp', 'scale']
    assert sig(tdgp.renderer.ImportanceRenderer.forward)[:6] == ['self', 'planes', 'decoder', 'ray_origins', 'ray_directions', 'rendering_options']
    assert sorted(tdgp.ops.bias_act.activation_funcs) == sorted(['linear', 'relu', 'lrelu', 'tanh', 'sigmoid', 'elu', 'selu', 'softplus', 'swish'])
    assert [tdgp.ops.bias_act.activation_funcs[k].cuda_idx for k in ['linear', 'relu', 'lrelu', 'tanh', 'sigmoid', 'elu', 'selu', 'softplus', 'swish']] == list(range(1, 10))


def test_ref_impl_paths_cpu(tdgp, oracle):
    """`impl='ref'` (API parity with the reference) against the oracle -- CPU tensors, plain PyTorch ops."""
    import numpy as np
    import torch
    rs = np.random.RandomState(0)
    x = rs.randn(2, 5, 9, 7).astype(np.float32)
    b = rs.randn(5).astype(np.float32)
    for act in tdgp.ops.bias_act.activation_funcs:
        y = tdgp.ops.bias_act.bias_act(torch.as_tensor(x), torch.as_tensor(b), act=act, impl='ref').numpy()
        assert np.abs(y - oracle.bias_act(x, b, act=act)).max() < 1e-5
    f = tdgp.ops.upfirdn2d.setup_filter([1, 3, 3, 1])
    y = tdgp.ops.upfirdn2d.upsample2d(torch.as_tensor(x), f, impl='ref').numpy()
    assert np.abs(y - oracle.upsample2d(x, f.numpy())).max() < 1e-5
    y = tdgp.ops.upfirdn2d.upfirdn2d(torch.as_tensor(x), f, up=[3, 2], down=[2, 1], padding=[2, 1, 0, 3], gain=1.5, impl='ref').numpy()
    assert np.abs(y - oracle.upfirdn2d(x, f.numpy(), up=[3, 2], down=[2, 1], padding=[2, 1, 0, 3], gain=1.5)).max() < 1e-5


def test_state_dict_layout_matches_reference_names(tdgp):
    cfg = tdgp.config.config_mid()
    G = tdgp.generator.Generator(cfg)
    res = G.load_numpy_state_dict(tdgp.weights.random_state_dict(cfg, seed=1), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    keys = set(G.state_dict())
    assert 'synthesis.tri_plane_decoder.b4.const' in keys and 'synthesis.tri_plane_decoder.b8.conv0.affine.weight' in keys
    assert 'synthesis.tri_plane_mlp.model.1.bias' in keys and 'mapping.w_avg' in keys and 'mapping.embed.weight' in keys


def test_walk2_isa_check_runs_in_the_build(tdgp):
    """the hand-issued tap loads of triplane_walk2_kernel rely on a property of the GENERATED code (3dgp_amd/isa_check.py).  The
    build runs the check whenever field.hip is recompiled and keeps the listing it checked; here the same check is re-run on that listing
    (produced now if a pre-built tree was copied without it), every instantiation must be present and clean."""
    import importlib
    build = importlib.import_module('3dgp_amd.build')
    asm = os.path.join(build.CSRC, 'build', 'field.s')
    if not os.path.exists(asm):
        subprocess.check_call([build._hipcc()] + build.FLAGS + ['-S', '--cuda-device-only', '-o', asm, os.path.join(build.CSRC, 'field.hip')], stderr=subprocess.DEVNULL)
    res = build.verify_field_isa(asm)
    assert len(res) >= 2 and all(not bad for _, bad in res.values())
    main = [k for k in res if 'ILi8ELi4ELb0E' in k]            # the C3 instantiation: feat 32, hid 64, no tap output
    assert main and res[main[0]][0]['tap_loads'] == 48 and res[main[0]][0]['compiler_vmcnt_waits'] == []
    # and the checker does catch a violation: an instruction reading a tap register right after its load
    isa = importlib.import_module('3dgp_amd.isa_check')
    lines = open(asm).read().splitlines()
    i0 = next(i for i, ln in enumerate(lines) if re.match(r'^_ZN\S*triplane_walk2_kernelILi8ELi4ELb0E\S*:', ln))
    i1 = next(j for j in range(i0, len(lines)) if lines[j].strip().startswith('.amdhsa_kernel'))
    body = lines[i0:i1]
    heads = [i for i, ln in enumerate(body) if 'Loop Header: Depth=1' in ln]
    k = next(i for i in range(heads[-1], len(body)) if body[i].strip().startswith('buffer_load_dwordx4'))
    dst = re.search(r'v\[(\d+):\d+\]', body[k]).group(1)
    body.insert(k + 1, f'\tv_add_f32_e32 v0, v{dst}, v{dst}')
    _, bad = isa.check_kernel(body, 8)
    assert bad and 'touches the destination' in bad[0][0]


def test_wino4_isa_check_runs_in_the_build(tdgp):
    """conv3_wino4_kernel's LDS-direct loads are hand-issued (M0 written inside the asm statement, which hipcc will not let a clobber
    list name) and its pair-form K loop ends in a hand-counted `vmcnt(4)`.  isa_check.check_wino4_asm verifies on the generated code that M0 is
    written only by those statements, that the K loop holds nothing but the 8 expected LDS-direct issue sites (U before V) and that vmcnt(4) is
    its only vector-memory wait; the build runs it whenever modconv.hip is recompiled.  Here: re-run on the kept listing, then three injected
    violations must each be caught."""
    import importlib
    build = importlib.import_module('3dgp_amd.build')
    isa = importlib.import_module('3dgp_amd.isa_check')
    asm = os.path.join(build.CSRC, 'build', 'modconv.s')
    if not os.path.exists(asm):
        subprocess.check_call([build._hipcc()] + build.FLAGS + ['-S', '--cuda-device-only', '-o', asm, os.path.join(build.CSRC, 'modconv.hip')], stderr=subprocess.DEVNULL)
    res_all = build.verify_modconv_isa(asm)
    res = {k: v for k, v in res_all.items() if 'wino4f' not in k and 'torgb' not in k}
    # torgb_mfma_kernel<.., FAST> (round 6): hand-issued skip taps in batches of 4 under hand-counted waits
    rest = {k: v for k, v in res_all.items() if 'torgb' in k}
    assert len(rest) == 9 and all(not bad for _, bad in rest.values()), {k[-40:]: bad[:1] for k, (_, bad) in rest.items() if bad}       # 6 one-role instances + 3 of torgb_ws_kernel
    assert all(s_['tap_loads'] == 4 * s_['batches'] == 4 * s_['hand_waits'] for s_, _ in rest.values())
    assert sorted(s_['stages'] for s_, _ in rest.values()) == [1] * 6 + [2] * 3
    assert len(res) == 2 and all(not bad for _, bad in res.values())
    assert all(s['lds_direct_sites'] == 8 and s['mfma'] == 36 and s['m0_writes'] > 0 for s, _ in res.values())
    # conv3_wino4f_kernel (round 6): every vector-memory read of its item loop is hand-issued; no spill, nothing of the compiler's to wait for behind the K loop
    resf = {k: v for k, v in res_all.items() if 'wino4f' in k}
    assert len(resf) == 1 and all(not bad for _, bad in resf.values())
    sf = next(iter(resf.values()))[0]
    assert sf['mfma'] == 144 and sf['lds_direct_sites'] >= 40 and sf['atomics'] >= 2 and sf['stores_behind_loop'] >= 16, sf
    lines = open(asm).read().splitlines()
    i0 = next(i for i, ln in enumerate(lines) if re.match(r'^_ZN\S*conv3_wino4_kernelILb1ELb1E\S*:', ln))
    i1 = next(j for j in range(i0, len(lines)) if lines[j].strip().startswith('.amdhsa_kernel'))
    body = lines[i0:i1]
    w = next(i for i, ln in enumerate(body) if 's_waitcnt vmcnt(4)' in ln)
    for inject, needle in (('\tglobal_load_dword v0, v[2:3], off', 'other than an LDS-direct load'), ('\ts_movrels_b32 s0, s1 ; uses m0', None),
                           ('\tv_readlane_b32 s0, v1, m0', 'm0 is touched')):
        if needle is None:
            continue
        b2 = list(body)
        b2.insert(w + 8, inject)                                   # inside the K loop (its body follows the latch block in the listing)
        _, bad = isa.check_wino4_kernel(b2)
        assert bad and any(needle in why for why, _ in bad), (inject, bad[:2])
    b3 = [ln for i, ln in enumerate(body) if not (i > w and ' lds' in ln and i < w + 120)]      # drop the first LDS-direct sites of the loop
    _, bad = isa.check_wino4_kernel(b3)
    assert bad
    # the fused kernel: a spill reload, a compiler-side load behind the K loop, a touched atomic destination are each caught
    f0 = next(i for i, ln in enumerate(lines) if re.match(r'^_Z\S*conv3_wino4f_kernel\S*:', ln))
    f1 = next(j for j in range(f0, len(lines)) if lines[j].strip().startswith('.amdhsa_kernel'))
    fbody = lines[f0:f1]
    last_mfma = max(i for i, ln in enumerate(fbody) if 'v_mfma' in ln)
    first_store = next(i for i in range(last_mfma, len(fbody)) if 'global_store_dwordx4' in fbody[i])
    at = next(i for i, ln in enumerate(fbody) if re.search(r'global_atomic_add v\d+, v\d+, v\d+, s\[', ln) and 's_waitcnt vmcnt(0)' not in ''.join(fbody[i:i + 6]))
    dst = re.search(r'global_atomic_add (v\d+),', fbody[at]).group(1)
    for pos, inject, needle in ((first_store, '\tscratch_load_dword v1, off, off offset:4', 'scratch access'),
                                (first_store, '\tglobal_load_dword v1, v[2:3], off', 'behind the K loop'),
                                (first_store, '\ts_waitcnt vmcnt(2)', 'wait behind the K loop'),
                                (at + 1, f'\tv_mov_b32_e32 v1, {dst}', 'destination of a returning atomic')):
        b2 = list(fbody)
        b2.insert(pos, inject)
        _, bad = isa.check_wino4f_kernel(b2)
        assert bad and any(needle in why for why, _ in bad), (inject, bad[:2])
    # ToRGB: a compiler-side load between two hand-counted waits, and a copy of a tap register still in flight, are each caught
    t0 = next(i for i, ln in enumerate(lines) if re.match(r'^_ZN\S*torgb_mfma_kernelILi3ELb1ELb1ELb0E\S*:', ln))
    t1 = next(j for j in range(t0, len(lines)) if lines[j].strip().startswith('.amdhsa_kernel'))
    tbody = lines[t0:t1]
    w8 = [i for i, ln in enumerate(tbody) if ln.strip() == 's_waitcnt vmcnt(8)' and tbody[i - 1].strip().startswith(';;#ASMSTART')]
    hl = [i for i, ln in enumerate(tbody) if ln.strip().startswith('global_load_dwordx4') and tbody[i - 1].strip().startswith(';;#ASMSTART')]
    dst = re.search(r'global_load_dwordx4 v\[(\d+):', tbody[hl[8]]).group(1)
    for pos, inject, needle in ((w8[3] + 2, '\tglobal_load_dword v1, v[2:3], off', 'between the hand-counted waits'),
                                (hl[8] + 2, f'\tv_mov_b32_e32 v1, v{dst}', 'in flight')):
        b2 = list(tbody)
        b2.insert(pos, inject)
        _, bad = isa.check_torgb_kernel(b2, 3)
        assert bad and any(needle in why for why, _ in bad), (inject, bad[:2])
    # a listing without loop annotations is reported as unreadable, not as an IndexError
    with pytest.raises(isa.IsaListingError):
        isa.check_kernel(['_Zfoo:', '\ts_endpgm'], 8)


def test_device_fault_word_is_exported_and_quiet(tdgp):
    """include/tdgp.h tdgp_device_fault: readable without a GPU (no fault word can be allocated -> 0), never raises."""
    assert tdgp._lib.device_fault() == 0 and tdgp._lib.device_fault(clear=True) == 0


def test_fold_up2_table_is_the_x2_layer(tdgp):
    """ops/modconv.fold_up2_table: the stride-2 transposed 3x3 convolution + 4x4 FIR of conv2d_resample.py:108-125 (flip_weight=False) as four
    3x3 'same' correlations of x, one per output parity -- against the op's own reference path (torch CPU ops, float64), image borders
    included, for the generator's symmetric filter and for an asymmetric one (the flip convention)."""
    import numpy as np
    import torch
    M, CR = tdgp.ops.modconv, tdgp.ops.conv2d_resample
    rs = np.random.RandomState(0)
    x = torch.tensor(rs.randn(2, 3, 7, 9))
    w = torch.tensor(rs.randn(4, 3, 3, 3))
    for f in (tdgp.ops.upfirdn2d.setup_filter([1, 3, 3, 1]), torch.tensor(rs.rand(4, 4), dtype=torch.float32)):
        ref = CR.conv2d_resample(x, w, f=f, up=2, padding=1, flip_weight=False)
        P = torch.tensor(M.fold_up2_table(f.numpy()))
        weff = torch.einsum('pqijab,ocab->opqcij', P, w).reshape(16, 3, 3, 3)
        ph = torch.nn.functional.conv2d(x, weff, padding=1)                                   # channel 4 o + 2 py + px
        got = ph.reshape(2, 4, 2, 2, 7, 9).permute(0, 1, 4, 2, 5, 3).reshape(2, 4, 14, 18)
        assert float((got - ref).abs().max()) <= 1e-12 * float(ref.abs().max())
