"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE: the CPU checker).

Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(HERE), "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")


class V3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Mesh(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("vertex_count", C.c_uint32), ("indices", C.c_void_p),
                ("index_count", C.c_uint32), ("mat", C.c_uint32), ("aabb_min", V3), ("aabb_max", V3)]


class Camera(C.Structure):
    _fields_ = [("p", V3), ("x_axis", V3), ("y_axis", V3), ("z_axis", V3)]


class SceneDesc(C.Structure):
    _fields_ = [("materials", C.c_void_p), ("material_count", C.c_uint32),
                ("spheres", C.c_void_p), ("sphere_count", C.c_uint32),
                ("boxes", C.c_void_p), ("box_count", C.c_uint32),
                ("cylinders", C.c_void_p), ("cylinder_count", C.c_uint32),
                ("meshes", C.c_void_p), ("mesh_count", C.c_uint32),
                ("lights", C.c_void_p), ("light_count", C.c_uint32),
                ("with_reference_csg", C.c_int32), ("octree_depth", C.c_uint32)]


class TreeStats(C.Structure):
    _fields_ = [("nodes", C.c_uint32), ("nonempty_leaves", C.c_uint32), ("max_leaf_records", C.c_uint32),
                ("record_bytes", C.c_uint64)]


class RenderStats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays", C.c_uint64), ("node_pops", C.c_uint64),
                ("child_tests", C.c_uint64), ("tri_tests", C.c_uint64), ("analytic_tests", C.c_uint64),
                ("shapes_tested", C.c_uint64), ("final_rng", C.c_uint32), ("seconds", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


POLICY = {"tile32": 0, "whole": 1, "pixel": 2, "chunk": 3, "sample": 3}

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.oracle_scene_create.restype = C.c_void_p
        L.oracle_scene_create.argtypes = [C.POINTER(SceneDesc)]
        L.oracle_scene_destroy.argtypes = [C.c_void_p]
        L.oracle_tree_stats.argtypes = [C.c_void_p, C.POINTER(TreeStats)]
        L.oracle_tiled_raytrace.restype = C.c_uint64
        L.oracle_tiled_raytrace.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_void_p, C.c_int32, C.c_int32,
                                            C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint32),
                                            C.c_uint32, C.c_float, C.POINTER(RenderStats)]
        L.oracle_render_image.restype = C.c_int
        L.oracle_render_image.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32,
                                          C.c_uint32, C.c_float, C.c_int32, C.POINTER(RenderStats)]
        L.oracle_job_seed.restype = C.c_uint32
        L.oracle_job_seed.argtypes = [C.c_uint32, C.c_uint32]
        L.oracle_raycast.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_void_p,
                                     C.POINTER(C.c_uint32)]
        L.oracle_unit_batch.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.oracle_rng_table.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
        L.oracle_rgbe.restype = C.c_uint32
        L.oracle_rgbe.argtypes = [C.c_float, C.c_float, C.c_float]
        L.oracle_write_hdr.restype = C.c_int
        L.oracle_write_hdr.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32]
        L.oracle_camera.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_int32, C.POINTER(Camera)]
        _lib = L
    return _lib


def _v3(a):
    return V3(float(a[0]), float(a[1]), float(a[2]))


class OracleScene:
    """Scene handle built from flattened arrays (ref_io.SceneDump-like object)."""

    def __init__(self, scene, with_reference_csg=True, octree_depth=10):
        L = lib()
        self._keep = []
        mats = np.ascontiguousarray(scene.materials)
        sph = np.ascontiguousarray(scene.spheres)
        box = np.ascontiguousarray(scene.boxes)
        cyl = np.ascontiguousarray(scene.cylinders)
        lights = np.ascontiguousarray(scene.lights)
        meshes = (Mesh * max(1, len(scene.meshes)))()
        for i, m in enumerate(scene.meshes):
            v = np.ascontiguousarray(m["vertices"], dtype="<f4")
            ix = np.ascontiguousarray(m["indices"], dtype="<u4")
            self._keep += [v, ix]
            meshes[i] = Mesh(v.ctypes.data, len(v), ix.ctypes.data, len(ix), int(m["mat"]),
                             _v3(m["aabb_min"]), _v3(m["aabb_max"]))
        self._keep += [mats, sph, box, cyl, lights, meshes]
        d = SceneDesc(mats.ctypes.data, len(mats), sph.ctypes.data, len(sph), box.ctypes.data, len(box),
                      cyl.ctypes.data, len(cyl), C.addressof(meshes), len(scene.meshes),
                      lights.ctypes.data, len(lights), 1 if with_reference_csg else 0, octree_depth)
        self.handle = L.oracle_scene_create(C.byref(d))
        cam = np.asarray(scene.camera, dtype="<f4").reshape(4, 3)
        self.camera = Camera(_v3(cam[0]), _v3(cam[1]), _v3(cam[2]), _v3(cam[3]))

    def close(self):
        if self.handle:
            lib().oracle_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_camera(self, cam4x3):
        cam = np.asarray(cam4x3, dtype="<f4").reshape(4, 3)
        self.camera = Camera(_v3(cam[0]), _v3(cam[1]), _v3(cam[2]), _v3(cam[3]))

    def tree_stats(self):
        st = TreeStats()
        lib().oracle_tree_stats(self.handle, C.byref(st))
        return dict(nodes=st.nodes, nonempty_leaves=st.nonempty_leaves, max_leaf_records=st.max_leaf_records,
                    record_bytes=st.record_bytes)

    def render(self, width, height, spp, seed, policy, chunk=1, rect=None, rr=0.8, threads=1):
        out = np.zeros((height, width, 3), dtype="<f4")
        x0, y0, x1, y1 = rect if rect else (0, 0, width, height)
        if policy == "sample":
            chunk = 1
        st = RenderStats()
        rc = lib().oracle_render_image(self.handle, C.byref(self.camera), out.ctypes.data, width, height, x0, y0, x1,
                                       y1, POLICY[policy], seed, spp, chunk, rr, threads, C.byref(st))
        if rc != 0:
            raise ValueError("oracle_render_image rejected the arguments")
        return out, st.as_dict()

    def tiled_raytrace(self, out, x0, y0, x1, y1, rng_state, spp, rr=0.8):
        height, width = out.shape[:2]
        state = C.c_uint32(rng_state)
        n = lib().oracle_tiled_raytrace(self.handle, C.byref(self.camera), out.ctypes.data, width, height, x0, y0, x1,
                                        y1, C.byref(state), spp, rr, None)
        return n, state.value

    def raycast(self, origins, dirs):
        origins = np.ascontiguousarray(origins, dtype="<f4")
        dirs = np.ascontiguousarray(dirs, dtype="<f4")
        n = len(origins)
        t = np.zeros(n, "<f4")
        nrm = np.zeros((n, 3), "<f4")
        mat = np.zeros(n, "<u4")
        L = lib()
        tt = C.c_float()
        mm = C.c_uint32()
        for i in range(n):
            L.oracle_raycast(self.handle, origins[i].ctypes.data, dirs[i].ctypes.data, C.byref(tt),
                             nrm[i].ctypes.data, C.byref(mm))
            t[i] = tt.value
            mat[i] = mm.value
        return t, nrm, mat


def unit_batch(records):
    records = np.ascontiguousarray(records)
    out = np.zeros((len(records), 8), "<f4")
    lib().oracle_unit_batch(records.ctypes.data, len(records), out.ctypes.data)
    return out


def rng_table(seed, n):
    nbytes = n * 8 + n * 4 + n * 4 + n * 12 + 4
    buf = np.zeros(nbytes, np.uint8)
    lib().oracle_rng_table(seed, n, buf.ctypes.data)
    return buf.tobytes()


def job_seed(master, job):
    return lib().oracle_job_seed(master & 0xFFFFFFFF, job & 0xFFFFFFFF)


def camera(p, quat_xyzw, ratio, width, height):
    p = np.ascontiguousarray(p, "<f4")
    q = np.ascontiguousarray(quat_xyzw, "<f4")
    cam = Camera()
    lib().oracle_camera(p.ctypes.data, q.ctypes.data, ratio, width, height, C.byref(cam))
    return np.array([[v.x, v.y, v.z] for v in (cam.p, cam.x_axis, cam.y_axis, cam.z_axis)], dtype="<f4")
