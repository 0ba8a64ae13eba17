#!/usr/bin/env python3
"""Render a measured-BRDF sphere with the wavefront caller (mitsuba_customization_amd/wavefront.py).

    python examples/wavefront_sphere.py --width 768 --height 512 --spp 16 --out gpurun_out/sphere.png

Sphere and ground use two synthetic MERL-layout tables (no real MERL file ships with the reference; pass
--sphere-merl / --ground-merl to use real .binary files).  Prints one JSON line with the time split between the
BSDF queue calls and the rest of the loop."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=768)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--sphere-merl", default=None)
    ap.add_argument("--ground-merl", default=None)
    ap.add_argument("--sphere-rgl", default=None, help="an RGL *_rgb.bsdf file (or \"synthetic\") for the sphere: the queue call evaluates "
                                                        "the adaptive-parameterisation material next to the ground's table")
    ap.add_argument("--sampling", choices=["cosine", "table"], default="cosine")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    import torch
    from mitsuba_customization_amd import host, synth, wavefront

    with host.MerlHip(0) as gpu:
        gpu.set_option(host.OPT_SAMPLING, 1 if args.sampling == "table" else 0)
        if args.sphere_rgl:
            sphere = gpu.upload_rgl(synth.make_rgl_fields(seed=3, n_phi=1, n_theta=8, res=32, res_ndf=64, res_sigma=32)) if args.sphere_rgl == "synthetic" \
                else gpu.load_rgl(args.sphere_rgl)
        else:
            sphere = gpu.load_merl(args.sphere_merl) if args.sphere_merl else gpu.upload_merl(synth.make_table("ggx_tab", seed=11))
        ground = gpu.load_merl(args.ground_merl) if args.ground_merl else gpu.upload_merl(synth.make_table("ggx_tab", seed=5))
        assert (sphere, ground) == (0, 1)
        shade = wavefront.GpuShade(gpu)
        wavefront.render(shade, 64, 48, spp=1, max_depth=2)                       # warm-up
        image, st = wavefront.render(shade, args.width, args.height, spp=args.spp, max_depth=args.depth)
        rays = args.width * args.height * args.spp
        line = {
            "image": [args.width, args.height], "spp": args.spp, "max_depth": args.depth, "sampling": args.sampling,
            "camera_paths": rays, "bsdf_queue_calls": st.bounces, "bsdf_units": st.queued_units,
            "seconds_total": round(st.total_seconds, 4), "seconds_in_bsdf_calls": round(st.shade_seconds, 4),
            "bsdf_share_of_time": round(st.shade_seconds / st.total_seconds, 4),
            "bsdf_Munits_per_s": round(st.queued_units / st.shade_seconds / 1e6, 1),
            "Mpaths_per_s": round(rays / st.total_seconds / 1e6, 2),
            "mean_radiance": [round(float(v), 5) for v in image.mean(dim=(0, 1))],
        }
        if args.out:
            os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
            wavefront.write_png(args.out, image)
            line["out"] = args.out
        print(json.dumps(line))


if __name__ == "__main__":
    main()
