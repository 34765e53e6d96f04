"""The capture side of the path: view tuples, their device-resident cache and the view schedule.

Host-side mirror of the reference's captured_data.py (same class and method names):

    Data.get_view / ray_view_generator / silh_view_generator      captured_data.py:43-82
    Data_Pointgray (960x1280, rays stored in the capture)          captured_data.py:85-124
    Data_Redmi (1080x1920, rays generated from K, R)               captured_data.py:126-165
    get_data                                                       optim.py:132-143

What is different, and why:
  * ``get_view`` does not copy nine pinned tensors host->device per call (captured_data.py:44-59, the cost
    SURVEY.md a16 names): a view is uploaded once and stays resident -- 72 views of 960x1280 are 7.3 GB of
    the 288 GB of HBM;
  * the view schedule draws from numpy's GLOBAL legacy generator exactly like the reference
    (``np.random.shuffle``, lazily, ray and silhouette generators interleaved), so that after the same
    ``np.random.seed`` the reference's one-view-per-step SGD trajectory visits the same views
    (tests/golden/view_schedule.npz holds the reference's own sequences); ``rng=`` substitutes a private
    ``np.random.RandomState``;
  * the HDF5 captures are not distributed with the reference and h5py is not part of this image: the loader reads
    the capture's datasets (``cam_proj`` [72,4,4], ``cam_k`` [3,3], ``screen_position`` [72,P,3], ``mask``
    [72,resy,resx], optional ``ray_origin`` / ``ray_dir`` [72,P,3]) from the ``.h5`` with h5py where it is
    importable and otherwise with drt_amd.hdf5_lite (pure Python; plain numeric datasets, contiguous or chunked,
    optionally deflate-compressed), or from an ``.npz`` with the same keys (tools/h5_to_npz.py converts).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import views

Float = torch.float64
N_CAPTURE_VIEWS = 72          # the turntable of every capture (captured_data.py:64, 79, 98)


class Data:
    """Base: a list (or dict) ``Views`` of host tuples + the reference's accessors."""

    name = ""
    num_view = N_CAPTURE_VIEWS
    n_total = N_CAPTURE_VIEWS
    device = "cuda"
    rng = None                  # None: numpy's global legacy state, like the reference

    def get_view(self, V_index):
        """(screen_pixel f64 [P,3], valid bool [P], mask f64 [P], origin f64 [P,3], ray_dir f64 [P,3],
        (R, K, R^-1, K^-1)) on the device; uploaded on first use, resident afterwards."""
        cache = self.__dict__.setdefault("_resident", {})
        V_index = int(V_index)
        hit = cache.get(V_index)
        if hit is None:
            screen_pixel, valid, mask, origin, ray_dir, camera_M = self.Views[V_index]
            dev = self.device
            hit = (screen_pixel.to(dev), valid.to(dev), mask.to(dev), origin.to(dev), ray_dir.to(dev),
                   tuple(m.to(dev) for m in camera_M))
            cache[V_index] = hit
        return hit

    def get_binding(self, scene, V_index):
        """The rays (and targets) of view ``V_index`` as an explicit handle of ``scene`` (drt_amd.diffrender.RayBinding): the resident views
        of a capture ARE constants, so the caller's loop need not rely on tensor-identity heuristics to get the trusted-grid path and
        recycled outputs.  One handle per (scene, view), all of a scene's handles sharing one output set (a loop renders one view per
        iteration, reference optim.py:95-97: the outputs of a call are valid until the next call on any view of this capture)."""
        book = self.__dict__.setdefault("_bindings", {})
        per_scene = book.get(id(scene))
        if per_scene is None or per_scene[0]() is not scene:
            import weakref
            from . import diffrender
            per_scene = book[id(scene)] = (weakref.ref(scene), {}, diffrender.RayBinding.Shared())
        V_index = int(V_index)
        b = per_scene[1].get(V_index)
        if b is None:
            screen_pixel, valid, _, origin, ray_dir, _ = self.get_view(V_index)
            b = per_scene[1][V_index] = scene.bind_rays(origin, ray_dir, screen_pixel, valid, shared=per_scene[2])
        return b

    def make_resident(self, ids=None):
        for k in (range(len(self.Views)) if ids is None else ids):
            self.get_view(k)
        return self

    def _shuffle(self, index):
        (np.random if self.rng is None else self.rng).shuffle(index)

    def ray_view_generator(self):
        n = self.n_total
        index = list(np.arange(0, n, n // self.num_view))
        if self.name == "mouse":                                  # the reference's hand-picked subset (captured_data.py:66-69)
            index = list(np.arange(-5, 10)) + list(np.arange(22, 40))
        while True:
            self._shuffle(index)
            for i in index:
                yield int(i % n)

    def silh_view_generator(self):
        n = self.n_total
        index = list(np.arange(n))
        while True:
            self._shuffle(index)
            for i in index:
                yield int(i % n)


def _open_capture(path):
    """Mapping of the capture's datasets: ``.npz``, or ``.h5`` / ``.hdf5`` through h5py or drt_amd.hdf5_lite."""
    if not os.path.exists(path):
        raise FileNotFoundError(f"capture {path!r} not found")
    if path.endswith(".npz"):
        return np.load(path)
    try:
        import h5py
    except ImportError:
        from . import hdf5_lite                 # pure-Python reader for plain numeric datasets (what the captures hold)
        return hdf5_lite.File(path)
    return h5py.File(path, "r")


class _CaptureFile(Data):
    resx = resy = 0
    rays_from_file = False

    def __init__(self, HyperParams, path=None, data_path="./data/", device="cuda", rng=None, pin=True):
        self.num_view = HyperParams["num_view"]
        self.name = HyperParams["name"]
        self.device, self.rng = device, rng
        if path is None:
            stem = os.path.join(data_path, self.name)
            path = next((stem + ext for ext in (".npz", ".h5") if os.path.exists(stem + ext)), stem + ".h5")
        cap = _open_capture(path)
        pin = pin and torch.cuda.is_available()

        def host(a, dtype):
            t = torch.tensor(np.asarray(a), dtype=dtype)
            return t.pin_memory() if pin else t

        K = np.asarray(cap["cam_k"][:], dtype=np.float64)
        K_inverse = np.linalg.inv(K)
        self.n_total = len(cap["cam_proj"])
        self.Views = []
        for i in range(self.n_total):
            R = np.asarray(cap["cam_proj"][i], dtype=np.float64)
            R_inverse = np.linalg.inv(R)
            screen_pixel = np.asarray(cap["screen_position"][i]).reshape([-1, 3])
            valid = screen_pixel[:, 0] != 0
            mask = np.asarray(cap["mask"][i])
            if mask.shape != (self.resy, self.resx):
                raise ValueError(f"mask of view {i} is {mask.shape}, this camera is {(self.resy, self.resx)}")
            if self.rays_from_file:
                ray_origin, ray_dir = cap["ray_origin"][i], cap["ray_dir"][i]
            else:
                ray_origin, ray_dir = views.generate_ray(self.resy, self.resx, K_inverse, R_inverse)
            soft = views.process_mask(mask)
            camera_M = (host(R, Float), host(K, Float), host(R_inverse, Float), host(K_inverse, Float))
            self.Views.append((host(screen_pixel, Float), host(valid, torch.bool), host(soft, Float).reshape(-1),
                               host(ray_origin, Float), host(ray_dir, Float), camera_M))
        if hasattr(cap, "close"):
            cap.close()


class Data_Pointgray(_CaptureFile):
    """Captures of the PointGrey camera: 960x1280, per-pixel rays stored in the file (captured_data.py:85-124)."""
    resy, resx, rays_from_file = 960, 1280, True


class Data_Redmi(_CaptureFile):
    """Captures of the phone camera: 1080x1920, pinhole rays from K and R (captured_data.py:126-165)."""
    resy, resx, rays_from_file = 1080, 1920, False


REDMI_CAM = ("tiger", "pig", "horse", "rabbit")
POINTGRAY_CAM = ("hand", "mouse", "dog", "monkey")


def get_data(HyperParams, **kw):
    """reference optim.py:132-143."""
    name = HyperParams["name"]
    if name in POINTGRAY_CAM:
        return Data_Pointgray(HyperParams, **kw)
    if name in REDMI_CAM:
        return Data_Redmi(HyperParams, **kw)
    raise ValueError(f"unknown capture {name!r}: expected one of {POINTGRAY_CAM + REDMI_CAM}")


class SyntheticData(Data):
    """Turntable views of a ground-truth mesh traced through the same path, with the tuple layout of
    Data.get_view; built on the device and resident there (stands in for the undistributed captures)."""

    def __init__(self, scene_gt, center, extent, resx, resy, num_view=72, device="cuda", n_total=72, view_ids=None, seed=0, name="synthetic"):
        self.resx, self.resy, self.num_view, self.n_total = resx, resy, num_view, n_total
        self.name, self.device = name, device
        self.rng = np.random.RandomState(seed)

        def render_gt(o, d):
            with torch.no_grad():
                return scene_gt.render_transparent(o, d)

        def hit_gt(o, d):
            return scene_gt.render_mask(o, d) > 0

        ids = list(range(n_total)) if view_ids is None else list(view_ids)
        vs = views.make_views(render_gt, hit_gt, center, extent, n_total, resx, resy, device=device, view_ids=ids)
        self.Views = dict(zip(ids, vs))
        self._resident = dict(self.Views)


# ---------------------------------------------------------------------------------------------------------------
# writing a capture file (the reference only reads them; its captures are not distributed).  The arrays follow the
# schema its loaders index (captured_data.py:94-108 PointGrey, 136-149 Redmi): per-view world->camera matrices
# ``cam_proj`` [n,4,4], intrinsics ``cam_k`` [3,3], background-pattern positions ``screen_position`` ([n,P,3] for the
# PointGrey camera, [n,resy,resx,3] for the phone -- the Redmi loader reshapes), object masks ``mask`` [n,resy,resx]
# uint8 in {0,255}, and for the PointGrey camera the calibrated per-pixel rays ``ray_origin`` / ``ray_dir`` [n,P,3].
# ---------------------------------------------------------------------------------------------------------------
CAMERAS = {"pointgray": Data_Pointgray, "redmi": Data_Redmi}


def synthetic_capture_arrays(scene_gt, center, extent, camera="pointgray", n_views=N_CAPTURE_VIEWS, view_ids=None, device="cuda"):
    """Trace a ground-truth scene (``drt_amd.diffrender.Scene``) from a turntable of ``n_views`` cameras of the given
    kind and return the capture's datasets as numpy arrays (``view_ids``: store only these views, in this order)."""
    cls = CAMERAS[camera]
    resy, resx = cls.resy, cls.resx
    cams = views.turntable_cameras(center, extent, n_views, resx, resy)
    ids = list(range(n_views)) if view_ids is None else list(view_ids)
    out = {"cam_proj": [], "screen_position": [], "mask": []}
    if cls.rays_from_file:
        out["ray_origin"], out["ray_dir"] = [], []
    for k in ids:
        R, K, Rinv, Kinv = cams[k]
        origin, ray_dir = views.generate_ray(resy, resx, Kinv, Rinv, device=device)
        with torch.no_grad():
            out_ori, out_dir, m = scene_gt.render_transparent(origin, ray_dir)
            hit = scene_gt.render_mask(origin, ray_dir) > 0
        sp = views.screen_targets(out_ori, out_dir, m, cams[k], center, extent).cpu().numpy()
        out["cam_proj"].append(np.asarray(R, dtype=np.float64))
        out["screen_position"].append(sp if cls.rays_from_file else sp.reshape(resy, resx, 3))
        out["mask"].append(hit.view(resy, resx).cpu().numpy().astype(np.uint8) * 255)
        if cls.rays_from_file:
            out["ray_origin"].append(origin.cpu().numpy())
            out["ray_dir"].append(ray_dir.cpu().numpy())
    arrays = {k: np.stack(v) for k, v in out.items()}
    arrays["cam_k"] = np.asarray(cams[0][1], dtype=np.float64)
    return arrays


def write_capture(path, arrays):
    """``<name>.h5`` (h5py where importable, else drt_amd.hdf5_lite.write_simple: same datasets, plain contiguous layout)
    or ``<name>.npz``."""
    if path.endswith(".npz"):
        np.savez(path, **arrays)
        return path
    try:
        import h5py
    except ImportError:
        from . import hdf5_lite
        return hdf5_lite.write_simple(path, arrays)
    with h5py.File(path, "w") as f:
        for k, v in arrays.items():
            f.create_dataset(k, data=v)
    return path
