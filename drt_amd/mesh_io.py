"""Triangle-mesh I/O and topology tables for the refraction-tracing path.

Replaces what the reference gets from ``trimesh`` inside ``Scene.update_mesh`` /
``Scene.init_edge`` (reference DiffRender.py:303-317, 338-355): PLY load/export,
the watertight assertion, the unique-edge -> two-face table ``E2F``, ``Edges``
and ``mean_len``.  Pure numpy, host side; runs once per topology change, never
per iteration.

PLY dialect handled: binary little-endian / ascii, vertex ``x y z`` float32 (+ any
extra scalar vertex properties, which are skipped), faces ``uchar n`` + n * int32.
"""
from __future__ import annotations

import os
import numpy as np

_PLY_TYPES = {
    "char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4",
    "float": "f4", "double": "f8", "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2",
    "int32": "i4", "uint32": "u4", "float32": "f4", "float64": "f8",
}


class TriMesh:
    """Minimal mesh record: ``vertices`` f64 [V,3], ``faces`` i64 [F,3].

    Mirrors the attributes of the trimesh object the reference keeps in
    ``Scene.mesh`` that its callers touch: ``vertices`` (assigned every
    iteration, DiffRender.py:381), ``faces``, ``export`` (optim.py:50,226).
    """

    def __init__(self, vertices, faces):
        self.vertices = np.ascontiguousarray(vertices, dtype=np.float64)
        self.faces = np.ascontiguousarray(faces, dtype=np.int64)
        if self.vertices.ndim != 2 or self.vertices.shape[1] != 3:
            raise ValueError("vertices must be [V,3]")
        if self.faces.ndim != 2 or self.faces.shape[1] != 3:
            raise ValueError("faces must be [F,3]")
        if len(self.faces) and (self.faces.min() < 0 or self.faces.max() >= len(self.vertices)):
            raise ValueError("face index out of range")

    # -- trimesh-compatible derived tables (reference DiffRender.py:343-351) --
    @property
    def edges(self):
        """Directed edges, 3 per face in face order: (v0,v1),(v1,v2),(v2,v0)."""
        return self.faces[:, [0, 1, 1, 2, 2, 0]].reshape(-1, 2)

    @property
    def edges_sorted(self):
        return np.sort(self.edges, axis=1)

    @property
    def edges_face(self):
        return np.repeat(np.arange(len(self.faces), dtype=np.int64), 3)

    @property
    def is_watertight(self):
        """Every undirected edge is shared by exactly two faces."""
        if len(self.faces) == 0:
            return False
        _, counts = np.unique(_edge_keys(self.edges_sorted, len(self.vertices)), return_counts=True)
        return bool(np.all(counts == 2))

    def export(self, path):
        write_ply(path, self.vertices, self.faces)
        return path


def _edge_keys(edges_sorted, n_vertices):
    return edges_sorted[:, 0].astype(np.int64) * np.int64(n_vertices) + edges_sorted[:, 1].astype(np.int64)


def group_rows_pairs(edges_sorted, n_vertices):
    """Indices [E,2] of directed-edge rows that form each undirected edge.

    Same contract as ``trimesh.grouping.group_rows(rows, require_count=2)`` used at
    DiffRender.py:348, with the (implementation-defined there) ordering pinned:
    groups ascend by (min vertex, max vertex); inside a group the two rows ascend
    by row index.  Edges with a count other than two are dropped.
    """
    keys = _edge_keys(edges_sorted, n_vertices)
    order = np.argsort(keys, kind="stable")
    ks = keys[order]
    start = np.flatnonzero(np.concatenate(([True], ks[1:] != ks[:-1])))
    count = np.diff(np.concatenate((start, [len(ks)])))
    sel = start[count == 2]
    return np.stack((order[sel], order[sel + 1]), axis=1)


def edge_tables(mesh: TriMesh):
    """(Edges i64 [E,2], E2F i64 [E,2,3], mean_len float) as Scene.init_edge builds them.

    reference DiffRender.py:338-355: ``mean_len`` averages all 3F directed edges.
    """
    es = mesh.edges_sorted
    groups = group_rows_pairs(es, len(mesh.vertices))
    edges = es[groups[:, 0]]
    e2f_index = mesh.edges_face[groups]              # [E,2]
    e2f = mesh.faces[e2f_index]                      # [E,2,3]
    d = mesh.vertices[mesh.edges[:, 0]] - mesh.vertices[mesh.edges[:, 1]]
    mean_len = float(np.linalg.norm(d, axis=1).mean())
    return edges, e2f, mean_len


# ----------------------------------------------------------------------------- PLY
def read_ply(path) -> TriMesh:
    with open(path, "rb") as f:
        raw = f.read()
    end = raw.find(b"end_header")
    if end < 0 or not raw.startswith(b"ply"):
        raise ValueError(f"{path}: not a PLY file")
    nl = raw.find(b"\n", end)
    header = raw[:end].decode("ascii", "replace").splitlines()
    body = raw[nl + 1:]
    fmt = None
    elements = []  # (name, count, [(kind, ...)])
    for line in header:
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] == "element":
            elements.append((tok[1], int(tok[2]), []))
        elif tok[0] == "property":
            if tok[1] == "list":
                elements[-1][2].append(("list", _PLY_TYPES[tok[2]], _PLY_TYPES[tok[3]], tok[4]))
            else:
                elements[-1][2].append(("scalar", _PLY_TYPES[tok[1]], tok[2]))
    if fmt not in ("binary_little_endian", "ascii"):
        raise ValueError(f"{path}: unsupported PLY format {fmt}")
    verts = faces = None
    if fmt == "ascii":
        toks = body.split()
        pos = 0
        for name, count, props in elements:
            if name == "vertex":
                k = len(props)
                arr = np.array(toks[pos:pos + count * k], dtype=np.float64).reshape(count, k)
                names = [p[2] for p in props]
                verts = arr[:, [names.index("x"), names.index("y"), names.index("z")]]
                pos += count * k
            elif name == "face":
                out = np.empty((count, 3), dtype=np.int64)
                for i in range(count):
                    n = int(toks[pos])
                    if n != 3:
                        raise ValueError("only triangle faces are supported")
                    out[i] = [int(t) for t in toks[pos + 1:pos + 4]]
                    pos += 1 + n
                faces = out
            else:
                raise ValueError(f"unsupported ascii element {name}")
        return TriMesh(verts, faces)
    off = 0
    for name, count, props in elements:
        if all(p[0] == "scalar" for p in props):
            dt = np.dtype([(p[2], "<" + p[1]) for p in props])
            arr = np.frombuffer(body, dtype=dt, count=count, offset=off)
            off += dt.itemsize * count
            if name == "vertex":
                verts = np.stack([arr["x"], arr["y"], arr["z"]], axis=1).astype(np.float64)
        elif name == "face" and len(props) == 1 and props[0][0] == "list":
            _, ct, it, _ = props[0]
            dt = np.dtype([("n", "<" + ct), ("idx", "<" + it, (3,))])
            arr = np.frombuffer(body, dtype=dt, count=count, offset=off)
            off += dt.itemsize * count
            if count and not np.all(arr["n"] == 3):
                raise ValueError("only triangle faces are supported")
            faces = arr["idx"].astype(np.int64)
        else:
            raise ValueError(f"unsupported element layout for {name}")
    if verts is None or faces is None:
        raise ValueError(f"{path}: missing vertex or face element")
    return TriMesh(verts, faces)


def write_ply(path, vertices, faces):
    """Binary little-endian PLY, float32 xyz + uchar/int32 faces (the dialect of data/*.ply)."""
    v = np.asarray(vertices, dtype="<f4")
    f = np.asarray(faces)
    hdr = ("ply\nformat binary_little_endian 1.0\ncomment drt_amd generated\n"
           f"element vertex {len(v)}\nproperty float x\nproperty float y\nproperty float z\n"
           f"element face {len(f)}\nproperty list uchar int vertex_indices\nend_header\n")
    rec = np.empty(len(f), dtype=np.dtype([("n", "u1"), ("idx", "<i4", (3,))]))
    rec["n"] = 3
    rec["idx"] = f
    d = os.path.dirname(os.path.abspath(path))
    os.makedirs(d, exist_ok=True)
    with open(path, "wb") as fh:
        fh.write(hdr.encode("ascii"))
        fh.write(v.tobytes())
        fh.write(rec.tobytes())


def load(path, process=False) -> TriMesh:
    """``trimesh.load(path, process=False)`` look-alike (DiffRender.py:304)."""
    return read_ply(path)


# ----------------------------------------------------------------------------- generators
def subdivide_midpoint(mesh: TriMesh, float32_positions=True) -> TriMesh:
    """One 1->4 midpoint subdivision (hand_vh 4 390 -> 17 560, horse_vh 12 562 -> 50 248 tris).

    Used to reach the "~50k tris" workloads of BASELINE.json configs[2..3].  New
    vertices are edge midpoints, rounded through float32 like a PLY round trip.
    """
    V = len(mesh.vertices)
    es = mesh.edges_sorted
    keys = _edge_keys(es, V)
    uniq, inv = np.unique(keys, return_inverse=True)
    a = uniq // V
    b = uniq % V
    mid = 0.5 * (mesh.vertices[a] + mesh.vertices[b])
    if float32_positions:
        mid = mid.astype(np.float32).astype(np.float64)
    m = (V + inv).reshape(-1, 3)        # midpoint id on edges (v0v1),(v1v2),(v2v0) of each face
    f = mesh.faces
    new_faces = np.concatenate([
        np.stack([f[:, 0], m[:, 0], m[:, 2]], 1),
        np.stack([m[:, 0], f[:, 1], m[:, 1]], 1),
        np.stack([m[:, 2], m[:, 1], f[:, 2]], 1),
        np.stack([m[:, 0], m[:, 1], m[:, 2]], 1),
    ]).reshape(4, -1, 3).transpose(1, 0, 2).reshape(-1, 3)
    return TriMesh(np.concatenate([mesh.vertices, mid]), new_faces)


def icosphere(subdivisions=3, radius=50.0, noise=0.0, seed=0) -> TriMesh:
    """Watertight procedural mesh (licence-free stand-in when data/*.ply is absent)."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t],
                  [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4],
                  [11, 10, 2], [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8],
                  [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    m = TriMesh(v / np.linalg.norm(v[0]), f)
    for _ in range(subdivisions):
        m = subdivide_midpoint(m, float32_positions=False)
        m.vertices = m.vertices / np.linalg.norm(m.vertices, axis=1, keepdims=True)
    rng = np.random.default_rng(seed)
    r = radius * (1.0 + noise * rng.standard_normal(len(m.vertices)))
    verts = (m.vertices * r[:, None]).astype(np.float32).astype(np.float64)
    return TriMesh(verts, m.faces)


def vertex_normals(mesh: TriMesh):
    """Area-weighted unit vertex normals (for synthetic ground-truth displacement)."""
    tri = mesh.vertices[mesh.faces]
    fn = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    vn = np.zeros_like(mesh.vertices)
    for k in range(3):
        np.add.at(vn, mesh.faces[:, k], fn)
    ln = np.linalg.norm(vn, axis=1, keepdims=True)
    ln[ln == 0] = 1.0
    return vn / ln
