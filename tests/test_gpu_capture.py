"""The capture path end to end (reference captured_data.py:85-165): a capture file with the reference's
schema is written from the synthetic generator, loaded through Data_Pointgray / Data_Redmi into the device-resident
view cache, and driven through render_transparent + ray_loss + one optimisation iteration on the GPU; every tensor of
every view, and the results, must equal the in-memory SyntheticData path."""
import numpy as np
import pytest
import torch

from conftest import IOR, data_path
from drt_amd import mesh_io, views

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("camera,name,ext", [("pointgray", "hand", ".h5"), ("redmi", "horse", ".h5"), ("pointgray", "mouse", ".npz")])
def test_capture_file_to_gpu_equals_in_memory_views(tmp_path, camera, name, ext):
    from drt_amd import captured_data as cd, diffrender as Render, optim as O
    Render.intIOR = IOR
    cls = cd.CAMERAS[camera]
    resy, resx = cls.resy, cls.resx
    Render.resy, Render.resx = resy, resx
    gt = mesh_io.read_ply(data_path("hand_vh.ply"))
    center, extent = views.mesh_frame(gt.vertices)
    scene_gt = Render.Scene(gt, 0)
    n = 3
    arrays = cd.synthetic_capture_arrays(scene_gt, center, extent, camera, n)
    P = resy * resx
    assert arrays["cam_proj"].shape == (n, 4, 4) and arrays["cam_k"].shape == (3, 3) and arrays["mask"].shape == (n, resy, resx)
    assert arrays["screen_position"].shape == ((n, P, 3) if cls.rays_from_file else (n, resy, resx, 3))
    assert ("ray_dir" in arrays) == cls.rays_from_file and set(np.unique(arrays["mask"])) == {0, 255}
    path = str(tmp_path / (name + ext))
    cd.write_capture(path, arrays)

    hp = dict(O.HyperParams, name=name, num_view=n)
    assert type(cd.get_data(hp, path=path)) is cls                      # the reference's name -> camera table (optim.py:132-143)
    data = cls(hp, path=path)
    assert (data.resy, data.resx, data.n_total, data.num_view) == (resy, resx, n, n)
    ref = cd.SyntheticData(scene_gt, center, extent, resx, resy, num_view=n, n_total=n)

    # a coarser, displaced hull is what gets optimised against the capture
    from conftest import golden
    hull = views.displaced_ground_truth(mesh_io.TriMesh(golden("hand_smooth_sm")["vertices"].astype(np.float64), gt.faces), sigma=0.05, seed=2)
    scene = Render.Scene(hull, 0)
    for k in range(n):
        a, b = data.get_view(k), ref.get_view(k)
        assert data.get_view(k) is a                                   # resident: the second call uploads nothing
        for x, y, what in zip(a[:5], b[:5], ("screen_pixel", "valid", "soft mask", "origin", "ray_dir")):
            assert x.is_cuda and x.dtype == y.dtype and x.shape == y.shape, what
            if what in ("origin", "ray_dir") and not cls.rays_from_file:
                # the phone camera's rays are generated from K, R at load time (captured_data.py:147) on the host, the
                # in-memory views generate them on the device: same formula, last-bit differences of the matrix products
                assert torch.allclose(x, y, rtol=0, atol=1e-14), what
            else:
                assert torch.equal(x, y), f"view {k}: {what} differs between the capture file and the in-memory view"
        for x, y in zip(a[5], b[5]):
            assert x.is_cuda and torch.allclose(x, y, rtol=1e-13, atol=1e-13)
        assert 0.01 < a[1].float().mean().item() < 0.5
        V = scene.vertices.detach().clone().requires_grad_(True)
        scene.update_verticex(V)
        oo, od, mk = scene.render_transparent(a[3], a[4])
        loss = Render.ray_loss(oo, od, mk, a[0], a[1])
        g, = torch.autograd.grad(loss, V)
        V2 = scene.vertices.detach().clone().requires_grad_(True)
        scene.update_verticex(V2)
        oo2, od2, mk2 = scene.render_transparent(a[3], a[4])
        loss2 = Render.ray_loss(oo2, od2, mk2, b[0], b[1])
        g2, = torch.autograd.grad(loss2, V2)
        assert torch.equal(oo, oo2) and torch.equal(od, od2) and torch.equal(mk, mk2) and loss.item() > 0
        assert loss.item() == pytest.approx(loss2.item(), rel=1e-12)
        assert torch.allclose(g, g2, rtol=1e-12, atol=1e-14 * g2.abs().max().item())

    # one iteration of the reference's loop on the loaded capture (ray + silhouette + smoothness terms, drop-in shapes)
    np.random.seed(3)
    lc = O.Loss_calculator(scene, data, hp)
    init_vertices, parameter, opt = O.setup_opt(scene, 0.05, hp)
    opt.zero_grad()
    scene.update_verticex(init_vertices + parameter)
    total, parts = lc.all_loss()
    total.backward()
    assert torch.isfinite(total) and torch.isfinite(parameter.grad).all() and parameter.grad.abs().max().item() > 0
    opt.step()
