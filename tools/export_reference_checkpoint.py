#!/usr/bin/env python3
"""Export a reference 3DGP checkpoint to a neutral container this package loads without the reference on the path.

    python tools/export_reference_checkpoint.py network-snapshot.pkl out_dir/            # needs the reference importable (TDGP_REFERENCE)

The reference stores `pickle.dump(dict(G=..., D=..., G_ema=..., ...))` of `persistence`-decorated modules (`src/torch_utils/
persistence.py`, `scripts/utils.py:150-204`): unpickling needs the reference's own classes.  This tool does that once, on a
machine that has the reference, and writes
    out_dir/generator.npz    every tensor of `G_ema.state_dict()` the generator forward reads (same key names)
    out_dir/generator.json   the configuration fields of `G_ema.cfg` that `3dgp_amd.GeneratorConfig` consumes
`3dgp_amd.weights.load_exported(out_dir)` rebuilds (config, state dict) on the GPU box.  Only data travels.
"""
import json
import os
import pickle
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('TDGP_REFERENCE', '/root/reference')


def unsupported_options(c, fp16_blocks=(), conv_clamp=None):
    """Generator options that change the forward and that 3dgp_amd does not implement.  A checkpoint trained with any of them would
    export and load cleanly and then render DIFFERENT images, so the exporter refuses (and records the checked values in
    generator.json so that `weights.config_from_json` can refuse a hand-edited file too)."""
    get = lambda path, d=None: _get(c, path, d)         # noqa: E731
    checked = dict(use_full_box=bool(get('use_full_box', False)), mlp_n_layers=int(get('tri_plane.mlp.n_layers', 2)),
                   has_view_cond=bool(get('tri_plane.view_cond', get('has_view_cond', False)) or False), camera_cond=bool(get('camera_cond', False)),
                   fp32_only=bool(get('fp32_only', True)) and not fp16_blocks, num_fp16_res=max(int(get('num_fp16_res', 0) or 0), len(fp16_blocks)),
                   conv_clamp=None if conv_clamp is None else float(conv_clamp), ray_start_is_auto=isinstance(get('camera.ray.start'), str))
    bad = []
    if checked['use_full_box'] or checked['ray_start_is_auto']:
        bad.append("use_full_box / ray_start='auto' (never resolved by the reference renderer either)")
    # (round 6: tri_plane.mlp.n_layers != 2, has_view_cond and camera_cond are implemented -- the decoder's eager path / MappingNetwork(camera_cond) --
    #  and exported as configuration fields below)
    if not checked['fp32_only'] and checked['num_fp16_res'] > 0:
        bad.append(f"fp32_only=false with num_fp16_res={checked['num_fp16_res']} (fp16 blocks + conv_clamp=256: the exported fp32 path has no clamp)")
    return checked, bad


def cfg_to_json(G):
    c = G.cfg
    get = lambda o, path, d=None: _get(o, path, d)      # noqa: E731
    # fp16 is decided per block at construction (networks_epigraf.py:99-108: `num_fp16_res` is a constructor argument that train.py:271-273
    # zeroes when fp32_only), so the modules themselves are asked
    dec = G.synthesis.tri_plane_decoder
    blocks = [getattr(dec, f'b{r}') for r in dec.block_resolutions]
    fp16_blocks = [int(b.resolution) for b in blocks if getattr(b, 'use_fp16', False)]
    checked, bad = unsupported_options(c, fp16_blocks, getattr(blocks[-1].conv1, 'conv_clamp', None))
    if bad:
        raise NotImplementedError('checkpoint uses generator options 3dgp_amd does not implement: ' + '; '.join(bad))
    out = dict(z_dim=int(G.z_dim), w_dim=int(G.w_dim), c_dim=int(G.c_dim), map_depth=int(get(c, 'map_depth', 2)), cbase=int(c.cbase), cmax=int(c.cmax),
               fmaps=float(get(c, 'fmaps', 1.0)), use_noise=bool(get(c, 'use_noise', True)), tri_plane_res=int(c.tri_plane.res), feat_dim=int(c.tri_plane.feat_dim),
               mlp_hid=int(c.tri_plane.mlp.hid_dim), mlp_n_layers=int(get(c, 'tri_plane.mlp.n_layers', 2)), has_view_cond=bool(get(c, 'has_view_cond', False) or False),
               camera_cond=getattr(G.mapping, 'camera_scalar_enc', None) is not None,
               camera_raw_scalars=bool(getattr(getattr(G.mapping, 'camera_scalar_enc', None), 'use_raw', False)),
               camera_cond_drop_p=float(getattr(G.mapping, 'camera_cond_drop_p', 0.0)),
               mean_camera_params=None if getattr(G.mapping, 'mean_camera_params', None) is None else [float(v) for v in G.mapping.mean_camera_params.tolist()],
               ray_marcher_type=str(c.ray_marcher_type), num_ray_steps=int(c.num_ray_steps),
               ray_start=float(c.camera.ray.start), ray_end=float(c.camera.ray.end), cube_scale=float(c.camera.cube_scale), use_inf_depth=bool(c.use_inf_depth),
               last_back=bool(get(c, 'dataset.last_back', False)), white_back=bool(get(c, 'dataset.white_back', False)), density_bias=float(get(c, 'density_bias', 0.0)),
               img_resolution=int(G.img_resolution), max_batch_res=int(get(c, 'max_batch_res', 128)), depth_adaptor=None, camera_adaptor=None)
    if get(c, 'depth_adaptor.enabled', False):
        d = c.depth_adaptor
        out['depth_adaptor'] = dict(kernel_size=int(d.kernel_size), hid_dim=int(d.hid_dim), num_hid_layers=int(d.num_hid_layers), out_strategy=str(d.out_strategy),
                                    near_plane_offset_max_fraction=float(d.near_plane_offset_max_fraction), near_plane_offset_bias=float(d.near_plane_offset_bias))
    if get(c, 'camera_adaptor.enabled', False):
        a = c.camera_adaptor
        cam = a.camera
        rng = lambda o: [float(o.min), float(o.max)]         # noqa: E731
        out['camera_adaptor'] = dict(hid_dim=int(a.hid_dim), embed_dim=int(a.embed_dim), lr_multiplier=float(a.lr_multiplier), residual=bool(get(a, 'residual', False)),
                                     adjust_angles=bool(a.adjust.angles), adjust_radius=bool(a.adjust.radius), adjust_fov=bool(a.adjust.fov),
                                     adjust_look_at=bool(a.adjust.look_at),
                                     camera=dict(yaw=rng(cam.origin.angles.yaw), pitch=rng(cam.origin.angles.pitch), fov=rng(cam.fov),
                                                 look_at_yaw=rng(cam.look_at.angles.yaw), look_at_pitch=rng(cam.look_at.angles.pitch),
                                                 look_at_radius=rng(cam.look_at.radius)))
    out['checked_options'] = checked
    return out


def _get(o, path, default=None):
    for k in path.split('.'):
        if isinstance(o, dict):
            if k not in o:
                return default
            o = o[k]
        elif hasattr(o, k):
            o = getattr(o, k)
        else:
            return default
    return o


def export(G, out_dir):
    os.makedirs(out_dir, exist_ok=True)
    cfg = cfg_to_json(G)
    sd = {k: v.detach().cpu().numpy() for k, v in G.state_dict().items()}
    np.savez(os.path.join(out_dir, 'generator.npz'), **sd)
    json.dump(cfg, open(os.path.join(out_dir, 'generator.json'), 'w'), indent=1)
    return cfg, sd


def main():
    sys.path.insert(0, REF)
    pkl, out_dir = sys.argv[1], sys.argv[2]
    with open(pkl, 'rb') as f:
        data = pickle.load(f)
    G = data['G_ema'] if isinstance(data, dict) else data
    cfg, sd = export(G.eval(), out_dir)
    print(f'{len(sd)} tensors, {sum(v.size for v in sd.values()) / 1e6:.1f} M parameters -> {out_dir}')


if __name__ == '__main__':
    main()
