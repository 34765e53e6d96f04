#!/usr/bin/env python3
"""Write a synthetic capture file with the reference's schema (captured_data.py:94-108, 136-149) by tracing a
ground-truth mesh on the GPU -- the reference's own captures (data/<name>.h5) are not distributed.

    python tools/make_capture.py --name horse --mesh data/horse_scan.ply --camera redmi --views 72 --out /tmp/horse.h5

The file is then read by drt_amd.captured_data.get_data / Data_Pointgray / Data_Redmi exactly like a real capture:
    python -m drt_amd.reconstruct --name horse --capture /tmp/horse.h5
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from drt_amd import captured_data as cd, diffrender as Render, mesh_io, views  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--name", default="hand")
    ap.add_argument("--mesh", default=None, help="ground-truth mesh (default data/<name>_scan.ply, else data/<name>_vh.ply)")
    ap.add_argument("--camera", choices=sorted(cd.CAMERAS), default=None, help="default: the camera the reference uses for --name")
    ap.add_argument("--views", type=int, default=cd.N_CAPTURE_VIEWS)
    ap.add_argument("--ior", type=float, default=1.4723)
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mesh_path = args.mesh or next(p for p in (os.path.join(root, "data", f"{args.name}_scan.ply"), os.path.join(root, "data", f"{args.name}_vh.ply")) if os.path.exists(p))
    camera = args.camera or ("pointgray" if args.name in cd.POINTGRAY_CAM else "redmi")
    Render.intIOR = args.ior
    mesh = mesh_io.read_ply(mesh_path)
    center, extent = views.mesh_frame(mesh.vertices)
    scene = Render.Scene(mesh, 0)
    arrays = cd.synthetic_capture_arrays(scene, center, extent, camera, args.views)
    cd.write_capture(args.out, arrays)
    print(f"{args.out}: {args.views} views of {mesh_path} ({camera}), " + ", ".join(f"{k} {v.shape} {v.dtype}" for k, v in arrays.items()))


if __name__ == "__main__":
    main()
