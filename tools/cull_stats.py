#!/usr/bin/env python3
"""How much of the leaf stream the matrix-core kernels still scan on the benchmark frame (block-cone culling)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytrace_amd import api, scenes, _lib
_lib.check(_lib.lib().rt_init(0))
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
world, lights = scenes.synthetic()
sc = api.Scene(world, lights)
rgba = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda")
hit = torch.empty((h, w), dtype=torch.int32, device="cuda")
for _ in range(3):
    sc.render_device(w, h, rgba.data_ptr(), hit.data_ptr())
    st = sc.stats()
blocks = -(-sc.num_leaves // 32)
pw = -(-w // 32) * -(-h // 8)
sw = st["shadow_rays"] / 256
print(f"{w}x{h}, {sc.num_leaves} leaves = {blocks} blocks; primary waves {pw}, shadow waves ~{sw:.0f}")
print(f"k_primary_mfma: {st['ms_primary']:.3f} ms, blocks scanned {st['mfma_blocks_primary']} = {st['mfma_blocks_primary']/pw:.1f} per wave ({100*st['mfma_blocks_primary']/(pw*blocks):.2f} % of brute force)")
print(f"k_shadow_mfma:  {st['ms_shadow']:.3f} ms, blocks scanned {st['mfma_blocks_shadow']} = {st['mfma_blocks_shadow']/sw:.1f} per wave ({100*st['mfma_blocks_shadow']/(sw*blocks):.2f} % of brute force)")
print(f"k_setup {st['ms_setup']:.3f} ms, k_final {st['ms_final']:.3f} ms, total {st['ms_total']:.3f} ms; exact evals {st['exact_evals']}, shadow rays {st['shadow_rays']}")
sc.render_device(w, h, None, hit.data_ptr())
st1 = sc.stats()
print(f"primary only: exact evals {st1['exact_evals']} = {st1['exact_evals']/(w*h):.2f} per ray; shadow: {st['exact_evals']-st1['exact_evals']} = {(st['exact_evals']-st1['exact_evals'])/max(1,st['shadow_rays']):.2f} per shadow ray")
