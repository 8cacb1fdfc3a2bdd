#!/usr/bin/env python3
"""Regenerates the fixtures in this directory from the CPU oracle (oracle/libbdpt_oracle.so).

  python tests/golden/make_golden.py

The reference renderer cannot be built in this environment (its GLM and assimp dependencies are
absent), so these vectors are NOT reference output: they freeze the oracle's answers on seeded
inputs, so that (a) a change of the oracle shows up as a diff and (b) the HIP backend can be checked
on a machine where only this repository exists. Inputs are regenerated from the seeds stored in each
file (tests/util.py), outputs are stored in full.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from util import GOLDEN_SCENES, bsdf_queries, capi, ray_batch, scenes, segment_batch  # noqa: E402

import oracle_py  # noqa: E402

SEED = 0x5EED
N_RAYS = 4096
N_SEGS = 4096
SCENES = GOLDEN_SCENES


def main():
    for name, make in SCENES.items():
        sc = make()
        sc.desc.camera.ratio = 1.0
        params = scenes.make_params(32, 32)
        orc = oracle_py.Oracle(sc, params)
        rays = ray_batch(N_RAYS, seed=42)
        segs = segment_batch(N_SEGS, seed=43)
        hits = orc.intersect_batch(rays)
        vis = orc.visibility_batch(segs)
        img, cn = orc.render_passes(0, 4, SEED, num_threads=4, counters=True)
        out = {
            "n_rays": N_RAYS, "ray_seed": 42, "n_segs": N_SEGS, "seg_seed": 43,
            "hit": hits["hit"].astype(np.int8), "object": hits["object"], "face": hits["face"].astype(np.int32),
            "d": hits["d"], "Ns": hits["Ns"], "tex_u": hits["tex_u"], "tex_v": hits["tex_v"],
            "visible": np.packbits(vis),
            "image_w": 32, "image_h": 32, "image_passes": 4, "image_seed": SEED, "image_sum": img,
            "counters": np.array([cn[k] for k in ("closest_hit_rays", "shadow_rays", "eye_vertices", "light_vertices",
                                                  "connections", "splats")], np.int64),
        }
        np.savez_compressed(os.path.join(HERE, f"scene_{name}.npz"), **out)
        print(name, "hits", int(hits["hit"].sum()), "visible", int(vis.sum()), "mean", img.mean() / 4)
    q = bsdf_queries(2048, seed=7)
    np.savez_compressed(os.path.join(HERE, "bsdf.npz"), n=2048, seed=7,
                        eval_f=oracle_py.bsdf_eval(q)["f"], pdf=oracle_py.bsdf_pdf(q)["pdf"],
                        scatter_dir=oracle_py.bsdf_scatter(q)["dir"], scatter_pdf=oracle_py.bsdf_scatter(q)["pdf"],
                        scatter_f=oracle_py.bsdf_scatter(q)["f"], scatter_ok=oracle_py.bsdf_scatter(q)["ok"].astype(np.int8),
                        scatter_specular=oracle_py.bsdf_scatter(q)["specular"].astype(np.int8))
    # full vertex dumps of a few pixel-samples of C2 (glass + mirror: specular chains, caustic splats)
    sc, params, _ = scenes.make_config("C2")
    params.img_res_x = params.img_res_y = 32
    orc = oracle_py.Oracle(sc, params)
    dumps = {}
    for k, (px, py, pas) in enumerate([(5, 20, 0), (16, 22, 1), (22, 24, 2), (12, 26, 3), (16, 3, 0), (0, 31, 7)]):
        eye, light, I, splats = orc.trace_pixel(pas, SEED, px, py)
        dumps[f"p{k}_where"] = np.array([px, py, pas])
        dumps[f"p{k}_eye"] = eye
        dumps[f"p{k}_light"] = light
        dumps[f"p{k}_I"] = I
        dumps[f"p{k}_splats"] = splats
    np.savez_compressed(os.path.join(HERE, "paths_C2.npz"), **dumps)


if __name__ == "__main__":
    main()
