#!/usr/bin/env python3
"""Build-container check (needs /root/reference): reference Generator with both adaptors -> pickle -> tools/export_reference_checkpoint.py -> 3dgp_amd.weights.load_exported; configuration and every tensor must survive."""
# round trip in the build container: reference Generator (random weights, adaptors on) -> pickle -> export tool -> this package
import sys, os, pickle, importlib, subprocess, tempfile, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_goldens as gg            # imports the reference with its stubs
import torch
t = gg.tdgp
tag, cfg = t.config.configs_adaptor_goldens()[1]
sd = t.weights.random_state_dict(cfg, seed=77, exercise_all=True)
rc = gg.ref_cfg(cfg)
da, ca = cfg.depth_adaptor, cfg.camera_adaptor
rc.depth_adaptor = gg.EasyDict(enabled=True, kernel_size=da.kernel_size, hid_dim=da.hid_dim, num_hid_layers=da.num_hid_layers, out_strategy=da.out_strategy,
                               near_plane_offset_max_fraction=da.near_plane_offset_max_fraction, near_plane_offset_bias=da.near_plane_offset_bias, selection_start_p=0.1, anneal_kimg=10000)
rc.camera_adaptor = gg.EasyDict(enabled=True, camera=gg.ref_camera_cfg(ca.camera), residual=ca.residual, lr_multiplier=ca.lr_multiplier, z_dim=cfg.z_dim, c_dim=cfg.c_dim,
                                hid_dim=ca.hid_dim, embed_dim=ca.embed_dim, adjust=gg.EasyDict(angles=ca.adjust_angles, radius=ca.adjust_radius, fov=ca.adjust_fov, look_at=ca.adjust_look_at))
G = gg.Generator(rc, img_resolution=cfg.img_resolution, img_channels=3, mapping_kwargs={}, num_fp16_res=0, conv_clamp=None, fused_modconv_default='inference_only').eval()
print(G.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True))
import export_reference_checkpoint as ex
d = tempfile.mkdtemp()
with open(os.path.join(d, 'snap.pkl'), 'wb') as f: pickle.dump(dict(G_ema=G), f)
with open(os.path.join(d, 'snap.pkl'), 'rb') as f: G2 = pickle.load(f)['G_ema']
ex.export(G2, os.path.join(d, 'out'))
cfg2, sd2 = t.weights.load_exported(os.path.join(d, 'out'))
assert cfg2.to_dict() == cfg.to_dict(), (cfg2.to_dict(), cfg.to_dict())
assert list(sd2) == list(sd) and all(np.array_equal(np.ravel(sd2[k]), np.ravel(sd[k])) for k in sd)
print('export round trip OK:', len(sd2), 'tensors')
