"""Data side of the path against vectors produced by the reference's own
captured_data.py (tests/golden/make_golden_data.py)."""
import os

import numpy as np
import pytest
import torch

from drt_amd import captured_data as cd, views

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _data(name, num_view, rng=None):
    d = cd.Data()
    d.name, d.num_view, d.rng = name, num_view, rng
    return d


def _draw(d, iters=120):
    ray, silh = d.ray_view_generator(), d.silh_view_generator()
    r, s = [], []
    for _ in range(iters):                      # one iteration of optim.py: a refraction view, then 8 silhouette views
        r.append(next(ray))
        s.append([next(silh) for _ in range(8)])
    return np.array(r), np.array(s)


def test_view_schedule_replays_the_reference():
    g = np.load(os.path.join(GOLD, "view_schedule.npz"))
    for k, (name, num_view, seed) in enumerate(zip(g["names"], g["num_view"], g["seed"])):
        np.random.seed(int(seed))                # global legacy state, like the reference
        r, s = _draw(_data(str(name), int(num_view)))
        np.testing.assert_array_equal(r, g[f"ray_{k}"])
        np.testing.assert_array_equal(s, g[f"silh_{k}"])
        # a private RandomState with the same seed gives the same schedule and leaves the global state alone
        state = np.random.get_state()[1].copy()
        r2, s2 = _draw(_data(str(name), int(num_view), rng=np.random.RandomState(int(seed))))
        np.testing.assert_array_equal(r2, r)
        np.testing.assert_array_equal(s2, s)
        np.testing.assert_array_equal(np.random.get_state()[1], state)


def test_mouse_subset_and_strides():
    np.random.seed(0)
    seen = {next(g) for g in [_data("mouse", 72).ray_view_generator()] for _ in range(33 * 4)}
    assert seen == {i % 72 for i in list(range(-5, 10)) + list(range(22, 40))}
    np.random.seed(0)
    gen = _data("hand", 9).ray_view_generator()
    assert {next(gen) for _ in range(90)} == set(range(0, 72, 8))


def test_generate_ray_matches_reference():
    g = np.load(os.path.join(GOLD, "generate_ray.npz"))
    o, d = views.generate_ray(13, 17, g["K_inv"], g["R_inv"])
    np.testing.assert_allclose(o.numpy(), g["origin"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(d.numpy(), g["dir"], rtol=0, atol=1e-14)


class _SmallPointgray(cd._CaptureFile):
    resy, resx, rays_from_file = 13, 17, True


class _SmallRedmi(cd._CaptureFile):
    resy, resx, rays_from_file = 13, 17, False


def test_capture_loader_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    n, resy, resx = 4, 13, 17
    cams = views.turntable_cameras(np.zeros(3), 100.0, n, resx, resy)
    sp = rng.standard_normal((n, resy * resx, 3))
    sp[:, ::5, 0] = 0                                     # invalid pixels (captured_data.py:106)
    mask = np.zeros((n, resy, resx), dtype=np.uint8)
    mask[:, 3:9, 4:12] = 255
    rays = [views.generate_ray(resy, resx, c[3], c[2]) for c in cams]
    path = str(tmp_path / "hand.npz")
    np.savez(path, cam_proj=np.stack([c[0] for c in cams]), cam_k=cams[0][1], screen_position=sp, mask=mask,
             ray_origin=np.stack([r[0].numpy() for r in rays]), ray_dir=np.stack([r[1].numpy() for r in rays]))
    hp = {"name": "hand", "num_view": 4}
    for cls in (_SmallPointgray, _SmallRedmi):
        data = cls(hp, path=path, device="cpu")
        assert data.n_total == n and (data.resy, data.resx) == (resy, resx)
        target, valid, soft, origin, ray_dir, camera_M = data.get_view(2)
        assert data.get_view(2)[0] is target              # resident: no second upload
        np.testing.assert_array_equal(target.numpy(), sp[2])
        np.testing.assert_array_equal(valid.numpy(), sp[2][:, 0] != 0)
        np.testing.assert_allclose(origin.numpy(), rays[2][0].numpy(), atol=1e-12)
        np.testing.assert_allclose(ray_dir.numpy(), rays[2][1].numpy(), atol=1e-14)
        np.testing.assert_allclose(camera_M[2].numpy() @ camera_M[0].numpy(), np.eye(4), atol=1e-12)
        img = soft.view(resy, resx).numpy()
        assert img.min() >= 0 and img.max() <= 1 and img[5, 7] == 1.0 and img[0, 0] == 0.0
        assert (img[-1] == 0.5).all()                     # captured_data.py:19
        assert soft.dtype == torch.float64 and valid.dtype == torch.bool
    with pytest.raises(ValueError):
        cd.Data_Pointgray(hp, path=path, device="cpu")    # 13x17 masks are not a 960x1280 capture
    with pytest.raises(ValueError):
        cd.get_data({"name": "teapot", "num_view": 72})
    with pytest.raises(FileNotFoundError):
        cd.get_data({"name": "dog", "num_view": 72}, data_path=str(tmp_path))


def test_view_order_of_the_recorded_trajectory():
    """The stochastic view order of tests/golden/hand_trajectory.npz -- produced by the reference's own generators (captured_data.py:61-82)
    inside its own Loss_calculator (one refraction view, then eight silhouette views per iteration) under np.random.seed -- comes out of
    drt_amd.captured_data.Data's generators under the same seed, consumed in the same order."""
    from conftest import golden
    from drt_amd import captured_data as cd
    g = golden("hand_trajectory")
    d = cd.Data()
    d.name, d.num_view, d.n_total = "hand", 72, 72
    np.random.seed(int(g["seed"]))
    ray, silh = d.ray_view_generator(), d.silh_view_generator()
    for it in range(len(g["ray_schedule"])):
        assert next(ray) == int(g["ray_schedule"][it]), it
        assert [next(silh) for _ in range(8)] == [int(k) for k in g["silh_schedule"][it]], it
