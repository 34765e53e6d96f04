"""Replays tests/golden/hand_trajectory.npz in several modes and prints loss / parameter drift (tests/test_gpu_trajectory.py asserts it)."""
import sys; sys.path.insert(0, 'tests')
import numpy as np, torch
import trajectory_case as tc
from conftest import data_path
from drt_amd import mesh_io, diffrender as Render, optim as O
g = tc.load()
hand = mesh_io.read_ply(data_path("hand_vh.ply"))
Render.intIOR = float(g["ior"]); Render.resx = Render.resy = int(g["res"])
Vs = g["vertices"].astype(np.float64)
hp = dict(O.HyperParams, IOR=float(g["ior"]), momentum=float(g["momentum"]), ray_w=float(g["ray_w"]), sm_w=float(g["sm_w"]), vh_w=float(g["vh_w"]))
def run(mode, concurrent=True, n=30):
    scene = Render.Scene(mesh_io.TriMesh(Vs, hand.faces), 0); scene.mean_len = float(g["mean_len"])
    data = tc.RecordedCapture(g, hand.vertices, "cuda")
    out = []
    if mode == "iter":
        st = O.FusedIteration(scene, data, hp, float(g["lr"]), concurrent=concurrent)
        for it in range(n):
            total, parts = st.step()
            out.append((float(total), st.parameter.clone().cpu().numpy(), [float(x) for x in parts]))
        return out
    lc = O.Loss_calculator(scene, data, hp, fused=(mode == "fused"))
    lc.CONCURRENT_TERMS = concurrent
    init, par, opt = O.setup_opt(scene, float(g["lr"]), hp)
    for it in range(n):
        opt.zero_grad(); v = init + par; scene.update_verticex(v)
        loss, parts = lc.all_loss(); loss.backward(); opt.step()
        out.append((float(loss.detach()), par.detach().clone().cpu().numpy(), [float(x) for x in parts]))
    return out
base = run("dropin")
for mode, conc in (("fused", True), ("fused", False), ("iter", True), ("iter", False), ("dropin", False)):
    o = run(mode, conc)
    d = [np.abs(o[it][1] - base[it][1]).max() for it in range(len(o))]
    first = next((it for it, x in enumerate(d) if x > 1e-13), None)
    print(mode, "concurrent" if conc else "serial", "first it with |param - dropin| > 1e-13:", first, " drift per it:", " ".join(f"{x:.1e}" for x in d[:12]))
    if first is not None:
        print("    parts there:", o[first][2], "dropin:", base[first][2])
