import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import golden, fixture_view, data_path, IOR
from drt_amd import diffrender as Render, mesh_io
Render.intIOR = IOR
Render.DENSE_FACE_IDS = True
g = golden("hand_r64_v5"); o, d, sp, valid = fixture_view(g)
hand = mesh_io.read_ply(data_path("hand_vh.ply"))
Render.resx = Render.resy = 64
scene = Render.Scene(data_path("hand_vh.ply"), 0)
V = torch.tensor(hand.vertices, dtype=torch.float64, device="cuda")
scene.update_verticex(V)
for it in range(3):
    with torch.no_grad():
        oo, od, mk = scene.render_transparent(o.cuda(), d.cuda())
    vi = torch.nonzero(mk[:, 0]).squeeze(1).cpu().numpy()
    print("call", it, "mask equal", np.array_equal(vi, g["valid_ind"]))
    if np.array_equal(vi, g["valid_ind"]):
        eo = np.abs(oo[vi].cpu().numpy() - g["out_ori"]).max(1); ed = np.abs(od[vi].cpu().numpy() - g["out_dir"]).max(1)
        bad = np.nonzero((eo > 1e-9) | (ed > 1e-11))[0]
        print("  bad rows", len(bad), "max err ori %.3g dir %.3g" % (eo.max(), ed.max()), "rays", vi[bad][:12])
        f1 = scene.last_face1.cpu().numpy(); f2 = scene.last_face2.cpu().numpy()
        b2 = np.full(len(o), -1, np.int64); b2[g["b2_ind"]] = g["b2_face"] if "b2_face" in g.files else -1
        if len(bad): print("  f2 of bad rays", f2[vi[bad]][:12], "golden keys", [k for k in g.files][:40])
