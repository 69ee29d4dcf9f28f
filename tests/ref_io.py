"""Readers/writers for the binary formats shared by oracle/ref_driver.cpp, the
oracle library and the golden fixtures under tests/golden/ (test infrastructure).
"""
import struct

import numpy as np

SCENE_MAGIC = 0x4E43534F  # 'OSCN'

MATERIAL_DTYPE = np.dtype([
    ("diffuse", "<f4", 3), ("specular", "<f4", 4), ("transmission", "<f4", 3), ("ior", "<f4"),
    ("emit", "<f4", 3), ("is_light", "<i4")])
SPHERE_DTYPE = np.dtype([("center", "<f4", 3), ("r", "<f4"), ("mat", "<u4")])
BOX_DTYPE = np.dtype([("min", "<f4", 3), ("max", "<f4", 3), ("mat", "<u4")])
CYLINDER_DTYPE = np.dtype([("base", "<f4", 3), ("axis", "<f4", 3), ("r", "<f4"), ("mat", "<u4")])
LIGHT_DTYPE = np.dtype([("type", "<u4"), ("index", "<u4")])

UNIT_REC_DTYPE = np.dtype([("op", "<u4"), ("a", "<f4", 24)])


class SceneDump:
    """Flattened scene exactly as the reference holds it after main()'s assembly."""

    def __init__(self):
        self.width = self.height = 0
        self.ambient = None
        self.camera = None  # 4x3: p, x_axis, y_axis, z_axis
        self.root_aabb = None  # 2x3
        self.materials = self.spheres = self.boxes = self.cylinders = self.lights = None
        self.meshes = []  # dicts: vertices (n,3) f32, indices (m,) u32, mat, aabb_min, aabb_max
        self.octree = None  # (nodes, leaves, record_bytes, max_leaf_bytes)


def read_scene_dump(path_or_bytes):
    data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    off = 0

    def take(dtype, count):
        nonlocal off
        arr = np.frombuffer(data, dtype=dtype, count=count, offset=off).copy()
        off += arr.nbytes
        return arr

    hdr = take("<u4", 9)
    assert hdr[0] == SCENE_MAGIC, "bad scene dump magic"
    nmat, nsph, nbox, ncyl, nmesh, nlight, w, h = (int(v) for v in hdr[1:])
    s = SceneDump()
    s.width, s.height = w, h
    s.ambient = take("<f4", 3)
    s.camera = take("<f4", 12).reshape(4, 3)
    s.root_aabb = take("<f4", 6).reshape(2, 3)
    s.materials = take(MATERIAL_DTYPE, nmat)
    s.spheres = take(SPHERE_DTYPE, nsph)
    s.boxes = take(BOX_DTYPE, nbox)
    s.cylinders = take(CYLINDER_DTYPE, ncyl)
    s.lights = take(LIGHT_DTYPE, nlight)
    for _ in range(nmesh):
        nv, ni, mat = (int(v) for v in take("<u4", 3))
        lo = take("<f4", 3)
        hi = take("<f4", 3)
        verts = take("<f4", 3 * nv).reshape(nv, 3)
        idx = take("<u4", ni)
        s.meshes.append(dict(vertices=verts, indices=idx, mat=mat, aabb_min=lo, aabb_max=hi))
    s.octree = tuple(int(v) for v in take("<u4", 4))
    assert off == len(data), "trailing bytes in scene dump"
    return s


def scene_digest(s):
    """Small, exact summary of a SceneDump: used as a committed fixture instead of the
    multi-megabyte dump (sha256 over every array's bytes + a few literal values)."""
    import hashlib
    hsh = hashlib.sha256()
    for arr in (s.ambient, s.camera, s.materials, s.spheres, s.boxes, s.cylinders, s.lights):
        hsh.update(np.ascontiguousarray(arr).tobytes())
    mesh_info = []
    for m in s.meshes:
        hsh.update(m["vertices"].tobytes())
        hsh.update(m["indices"].tobytes())
        hsh.update(np.asarray(m["aabb_min"], "<f4").tobytes())
        hsh.update(np.asarray(m["aabb_max"], "<f4").tobytes())
        mesh_info.append(dict(
            vertex_count=int(len(m["vertices"])), index_count=int(len(m["indices"])), mat=int(m["mat"]),
            first_vertex_bits=[int(v) for v in m["vertices"][0].view("<u4")] if len(m["vertices"]) else [],
            last_vertex_bits=[int(v) for v in m["vertices"][-1].view("<u4")] if len(m["vertices"]) else [],
            first_indices=[int(v) for v in m["indices"][:6]], last_indices=[int(v) for v in m["indices"][-6:]],
            vertices_sha256=hashlib.sha256(m["vertices"].tobytes()).hexdigest(),
            indices_sha256=hashlib.sha256(m["indices"].tobytes()).hexdigest(),
            aabb_min_bits=[int(v) for v in np.asarray(m["aabb_min"], "<f4").view("<u4")],
            aabb_max_bits=[int(v) for v in np.asarray(m["aabb_max"], "<f4").view("<u4")]))
    return dict(
        counts=dict(materials=len(s.materials), spheres=len(s.spheres), boxes=len(s.boxes),
                    cylinders=len(s.cylinders), meshes=len(s.meshes), lights=len(s.lights)),
        camera_bits=[int(v) for v in s.camera.reshape(-1).view("<u4")],
        ambient_bits=[int(v) for v in s.ambient.view("<u4")],
        materials_sha256=hashlib.sha256(s.materials.tobytes()).hexdigest(),
        spheres_sha256=hashlib.sha256(s.spheres.tobytes()).hexdigest(),
        boxes_sha256=hashlib.sha256(s.boxes.tobytes()).hexdigest(),
        cylinders_sha256=hashlib.sha256(s.cylinders.tobytes()).hexdigest(),
        lights=[[int(t), int(i)] for t, i in zip(s.lights["type"], s.lights["index"])],
        meshes=mesh_info,
        sha256=hsh.hexdigest())


def make_unit_records(op, rows):
    """rows: (n, k<=24) float32 -> packed unit records."""
    rows = np.asarray(rows, dtype="<f4")
    rec = np.zeros(len(rows), dtype=UNIT_REC_DTYPE)
    rec["op"] = op
    rec["a"][:, : rows.shape[1]] = rows
    return rec


def read_unit_output(path_or_bytes):
    data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    return np.frombuffer(data, dtype="<f4").reshape(-1, 8).copy()


def bits(a):
    return np.ascontiguousarray(a, dtype="<f4").view("<u4")


def pack_u32_as_f32(u):
    return struct.unpack("<f", struct.pack("<I", int(u) & 0xFFFFFFFF))[0]
