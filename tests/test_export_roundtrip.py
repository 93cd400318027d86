"""Checkpoint interchange: reference Generator -> pickle -> tools/export_reference_checkpoint.py ->
3dgp_amd.weights.load_exported.  The round trip needs the reference's classes, so it runs only where the reference tree exists
(the build container); the option checks of the loader run everywhere."""
import importlib
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('TDGP_REFERENCE', '/root/reference')


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'src')), reason='reference tree only exists in the build container')
def test_export_roundtrip_through_the_reference_pickle():
    """tools/check_export_roundtrip.py: a reference Generator with both adaptors, pickled the way the reference saves snapshots,
    exported and loaded back -- configuration and all 129 tensors survive bit for bit."""
    out = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'check_export_roundtrip.py')], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and 'export round trip OK' in out.stdout, out.stderr[-2000:]


def test_loader_rejects_unsupported_generator_options():
    """A checkpoint trained with an option that changes the forward and is not implemented here (fp16 blocks with conv_clamp, use_full_box,
    ray_start='auto') must not load silently.  Round 6: view conditioning, camera_cond and a tri-plane MLP with n_layers != 2 ARE implemented
    (configuration fields mlp_n_layers / has_view_cond / camera_cond ...) and load."""
    tdgp = importlib.import_module('3dgp_amd')
    base = tdgp.config.config_tiny().to_dict()
    ok = dict(base, checked_options=dict(use_full_box=False, mlp_n_layers=2, has_view_cond=False, camera_cond=False, fp32_only=True, num_fp16_res=0,
                                         ray_start_is_auto=False))
    assert tdgp.weights.config_from_json(ok).to_dict() == base
    for fine in (dict(mlp_n_layers=3), dict(has_view_cond=True), dict(camera_cond=True)):
        d = dict(base, checked_options=dict(ok['checked_options'], **fine))
        assert tdgp.weights.config_from_json(d).to_dict() == base
    d = dict(base, mlp_n_layers=3, camera_cond=True, mean_camera_params=[0.0, 1.57, 0.0])
    assert tdgp.weights.config_from_json(d).mlp_n_layers == 3 and tdgp.weights.config_from_json(d).camera_cond
    for bad in (dict(use_full_box=True), dict(fp32_only=False, num_fp16_res=4), dict(ray_start_is_auto=True)):
        d = dict(base, checked_options=dict(ok['checked_options'], **bad))
        with pytest.raises(NotImplementedError):
            tdgp.weights.config_from_json(d)


def test_exporter_flags_unsupported_options():
    sys.path.insert(0, os.path.join(REPO, 'tools'))
    try:
        ex = importlib.import_module('export_reference_checkpoint')
    finally:
        sys.path.pop(0)
    cfg = dict(use_full_box=False, fp32_only=False, num_fp16_res=4, tri_plane=dict(mlp=dict(n_layers=2)), camera=dict(ray=dict(start=0.75)))
    checked, bad = ex.unsupported_options(cfg)
    assert checked['num_fp16_res'] == 4 and len(bad) == 1 and 'fp16' in bad[0]
    cfg.update(fp32_only=True, use_full_box=True)
    assert len(ex.unsupported_options(cfg)[1]) == 1
    cfg.update(use_full_box=False)
    assert ex.unsupported_options(cfg)[1] == []
