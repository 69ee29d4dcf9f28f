#!/usr/bin/env python3
"""Generates the golden fixtures in this directory from the REFERENCE ITSELF.

Runs only in the dev container: it drives oracle/_ref/ref_det and oracle/_ref/ref_glibc,
i.e. the reference's own ray.cpp / parser.cpp / random.h compiled where they lie under
/root/reference by oracle/Makefile (the driver around them is oracle/ref_driver.cpp).
What is committed is data only: inputs and the reference's outputs.

  ref_det   = reference + deterministic libm (oracle/det_math.h)  -> bit-exact anchor
  ref_glibc = reference + glibc libm, "as shipped" on this box     -> distance report

usage: python tests/golden/make_golden.py
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_io  # noqa: E402

REF_DET = os.path.join(ROOT, "oracle", "_ref", "ref_det")
REF_GLIBC = os.path.join(ROOT, "oracle", "_ref", "ref_glibc")
DATA = os.path.join(ROOT, "data") + "/"
TMP = tempfile.mkdtemp(prefix="golden_")

SCENES = ["testscene", "c2_analytic", "c3_bunny_room", "c4_dwarf_room", "letters", "glass_room", "rand_a", "rand_b"]
# (policy, W, H, spp, chunk, seed)
RENDERS = [("tile32", 64, 64, 2, 1, 12345), ("whole", 24, 16, 2, 1, 999), ("pixel", 40, 30, 3, 1, 7),
           ("chunk", 40, 30, 4, 2, 7), ("sample", 33, 17, 3, 1, 31337)]


def run(binary, *args):
    out = subprocess.check_output([binary] + [str(a) for a in args])
    text = out.decode().strip()
    return json.loads(text.splitlines()[-1]) if text.startswith("{") or "\n{" in text else text


def unit_vectors(rng, n):
    v = rng.normal(size=(n, 3)).astype("<f4")
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype("<f4")


def make_unit_inputs():
    rng = np.random.default_rng(20241004)
    recs = []
    N = 192
    # 1 triangle: v0 v1 v2 o d
    v = rng.uniform(-2, 2, size=(N, 9)).astype("<f4")
    o = rng.uniform(-4, 4, size=(N, 3)).astype("<f4")
    centroid = (v[:, 0:3] + v[:, 3:6] + v[:, 6:9]) / 3
    d = centroid + rng.normal(scale=0.6, size=(N, 3)).astype("<f4") - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype("<f4")
    rows = np.concatenate([v, o, d], axis=1)
    rows[0, 12:15] = v[0, 3:6] - v[0, 0:3]          # ray parallel to the triangle (in-plane direction)
    rows[1, 9:12] = v[1, 0:3]; rows[1, 12:15] = (0, 0, 1)  # origin on a vertex
    rows[2, 3:6] = rows[2, 0:3]                     # degenerate triangle
    rows[3, 12:15] = -rows[3, 12:15]                # pointing away
    bary = np.array([0.5, 0.5, 0.0], "<f4")         # through an edge midpoint
    rows[4, 12:15] = (bary[0] * v[4, 0:3] + bary[1] * v[4, 3:6]) - o[4]
    rows[5, 12:15] = v[5, 0:3] - o[5]               # through a vertex (unnormalised direction)
    recs.append(ref_io.make_unit_records(1, rows))
    # 2 sphere: c r o d
    c = rng.uniform(-2, 2, size=(N, 3)).astype("<f4")
    r = rng.uniform(0.05, 1.5, size=(N, 1)).astype("<f4")
    o = rng.uniform(-4, 4, size=(N, 3)).astype("<f4")
    d = c + rng.normal(scale=0.8, size=(N, 3)).astype("<f4") * r - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype("<f4")
    rows = np.concatenate([c, r, o, d], axis=1)
    rows[0, 4:7] = rows[0, 0:3]                                  # origin at the centre (inside)
    rows[1, 4:7] = rows[1, 0:3] + (rows[1, 3] * 0.5, 0, 0)       # inside, off-centre
    rows[2, 0:3] = (0, 0, 0); rows[2, 3] = 1; rows[2, 4:7] = (-3, 1, 0); rows[2, 7:10] = (1, 0, 0)  # tangent
    rows[3, 7:10] = (0, 0, 0)                                    # zero direction
    rows[4, 3] = 0                                               # zero radius
    rows[5, 0:3] = (0, 0, 0); rows[5, 3] = 10; rows[5, 4:7] = (0, 0, 0); rows[5, 7:10] = (2, 2, 2)  # macos_main.mm:600
    recs.append(ref_io.make_unit_records(2, rows))
    # 3 aab: min max o d
    lo = rng.uniform(-3, 0, size=(N, 3)).astype("<f4")
    hi = lo + rng.uniform(0.05, 3, size=(N, 3)).astype("<f4")
    o = rng.uniform(-5, 5, size=(N, 3)).astype("<f4")
    d = (lo + hi) / 2 + rng.normal(scale=1.0, size=(N, 3)).astype("<f4") - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype("<f4")
    rows = np.concatenate([lo, hi, o, d], axis=1)
    rows[0, 9:12] = (1, 0, 0)                                    # axis-parallel: 1/0 = inf
    rows[1, 9:12] = (0, -1, 0)
    rows[2, 6:9] = (lo[2] + hi[2]) / 2                           # origin inside
    rows[3, 6] = hi[3, 0]                                        # origin exactly on the max face
    rows[4, 6] = lo[4, 0]; rows[4, 9:12] = (0, 1, 0)             # on the min face, parallel: 0 * inf = NaN
    rows[5, 0:6] = (-1.5, -1.5, -1.5, 1.5, 1.5, 1.5); rows[5, 6:9] = (0, 0, 0); rows[5, 9:12] = (1, 1, 1)  # macos_main.mm:416
    rows[6, 9:12] = -rows[6, 9:12]                               # box behind the ray
    recs.append(ref_io.make_unit_records(3, rows))
    # 4 cylinder: base axis r o d
    base = rng.uniform(-2, 2, size=(N, 3)).astype("<f4")
    axis = rng.normal(size=(N, 3)).astype("<f4") * rng.uniform(0.2, 3, size=(N, 1)).astype("<f4")
    r = rng.uniform(0.05, 0.8, size=(N, 1)).astype("<f4")
    o = rng.uniform(-4, 4, size=(N, 3)).astype("<f4")
    d = base + axis * rng.uniform(0, 1, size=(N, 1)).astype("<f4") + rng.normal(scale=0.5, size=(N, 3)).astype("<f4") - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype("<f4")
    rows = np.concatenate([base, axis, r, o, d], axis=1)
    rows[0, 3:6] = (0, 0, 0.4)                                   # axis = +z (identity frame)
    rows[1, 3:6] = (0, 0, -0.4)                                  # axis = -z (identity frame too, sic)
    rows[2, 3:6] = (-2.4, 0, 0)
    rows[3, 3:6] = (0, 2.4, 0)
    rows[4, 3:6] = (0, 0, 5.0); rows[4, 10:13] = (1, 0, 0)       # ray perpendicular to the axis: d.z = 0
    rows[5, 3:6] = (0, 0, 5.0); rows[5, 10:13] = (0, 0, 1)       # ray along the axis: a = 0
    rows[6, 7:10] = rows[6, 0:3] + 0.5 * rows[6, 3:6]            # origin inside
    recs.append(ref_io.make_unit_records(4, rows))
    # materials used by the BSDF tables: (Kd, Ks, Kt, ior)
    mats = np.array([
        [0.6, 0.6, 0.6, 0, 0, 0, 0, 0, 0, 1.0],        # diffuse
        [0.2, 0.2, 0.2, 1, 1, 1, 0, 0, 0, 1.0],        # diffuse + specular
        [0, 0, 0, 1, 1, 1, 0, 0, 0, 1.0],              # mirror
        [0, 0, 0, 0, 0, 0, 1, 1, 1, 1.4],              # glass
        [0, 0, 0, 0.2, 0, 0, 1, 0, 0, 1.2],            # red glass: logf(0) on two channels
        [0.4, 0.5, 0.2, 1, 1, 1, 0, 0, 0, 1.0],
        [0, 0, 0, 0, 0, 0, 0, 0, 0, 1.0],              # all zero: 0/0 lobe weights
        [0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.4, 0.4, 0.4, 1.33]], "<f4")
    m = mats[rng.integers(0, len(mats), size=N)]
    m[: len(mats)] = mats
    Nn = unit_vectors(rng, N)
    Nn[0] = (0, 0, 1); Nn[1] = (0, 0, -1); Nn[2] = (0, 0, 0.99995); Nn[3] = (1, 0, 0)
    wo = unit_vectors(rng, N)
    wi = unit_vectors(rng, N)
    flip = (np.sum(wo * Nn, axis=1) < 0) & (rng.uniform(size=N) < 0.7)
    wo[flip] = -wo[flip]
    rough = np.full((N, 1), 0.01, "<f4")
    seeds = rng.integers(1, 2 ** 32, size=N, dtype=np.uint64).astype("<u4").view("<f4").reshape(N, 1)
    # 5 sample_brdf: seed N wo rough Kd Ks Kt ior
    recs.append(ref_io.make_unit_records(5, np.concatenate([seeds, Nn, wo, rough, m], axis=1)))
    # 6 pdf_brdf: N wi wo rough Kd Ks Kt ior
    recs.append(ref_io.make_unit_records(6, np.concatenate([Nn, wi, wo, rough, m], axis=1)))
    # near-specular configurations (wi close to the mirror direction) exercise the GGX lobe
    refl = (2 * np.sum(wo * Nn, axis=1, keepdims=True) * Nn - wo).astype("<f4")
    wi2 = refl + rng.normal(scale=0.004, size=(N, 3)).astype("<f4")
    wi2 = (wi2 / np.linalg.norm(wi2, axis=1, keepdims=True)).astype("<f4")
    recs.append(ref_io.make_unit_records(6, np.concatenate([Nn, wi2, wo, rough, m], axis=1)))
    # 7 eval_scattering: N wi wo Kd Ks Kt ior rough dist
    dist = rng.uniform(0.01, 8, size=(N, 1)).astype("<f4")
    recs.append(ref_io.make_unit_records(7, np.concatenate([Nn, wi, wo, m, rough, dist], axis=1)))
    recs.append(ref_io.make_unit_records(7, np.concatenate([Nn, wi2, wo, m, rough, dist], axis=1)))
    # 8 sample_lobe: N c phi
    cphi = np.concatenate([rng.uniform(0, 1, size=(N, 1)), rng.uniform(0, 2 * np.pi, size=(N, 1))], axis=1).astype("<f4")
    recs.append(ref_io.make_unit_records(8, np.concatenate([Nn * rng.uniform(0.5, 2, size=(N, 1)).astype("<f4"), cphi], axis=1)))
    # 9 libm: x y
    xy = np.concatenate([rng.uniform(-7, 7, size=(N, 1)), rng.uniform(-6, 6, size=(N, 1))], axis=1).astype("<f4")
    xy[N // 2:, 0] = rng.uniform(0, 1, size=N - N // 2)           # pow bases in [0,1]
    xy[N // 2:N // 2 + 40, 1] = 5.0
    xy[N // 2 + 40:N // 2 + 80, 1] = 4.0
    special = [0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 0.5, 2.0, -2.0, 3.0, 1e-40, 2.71828175, 6.2831855]
    k = 0
    for a in special:
        for b in special[:7]:
            if k < N // 2:
                xy[k] = (a, b)
                k += 1
    recs.append(ref_io.make_unit_records(9, xy))
    # 10 normalize
    vv = rng.normal(size=(N, 3)).astype("<f4") * (10.0 ** rng.uniform(-8, 4, size=(N, 1))).astype("<f4")
    vv[0] = 0; vv[1] = (1e-7, 0, 0); vv[2] = (9.9e-7, 0, 0); vv[3] = (1.0e-6, 0, 0); vv[4] = (-1e-6, 1e-6, 0)
    recs.append(ref_io.make_unit_records(10, vv))
    # 11 fresnel / ggx / geometry: Ks l_dot_h N H rough w
    H = unit_vectors(rng, N)
    H[: N // 2] = (Nn[: N // 2] + rng.normal(scale=0.01, size=(N // 2, 3))).astype("<f4")
    H = (H / np.linalg.norm(H, axis=1, keepdims=True)).astype("<f4")
    ldh = rng.uniform(-1.05, 1.05, size=(N, 1)).astype("<f4")
    recs.append(ref_io.make_unit_records(11, np.concatenate([m[:, 3:6], ldh, Nn, H, rough, wo], axis=1)))
    return np.concatenate(recs)


def golden_c5(manifest):
    """C5 (BASELINE.json configs[4]) on its 99 458-triangle decimation, the size the reference build can
    hold (at n = 708 its fixed arenas overflow): generated mesh -> reference loader, octree, renders."""
    import hashlib
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_heightfield
    n = 224
    scn, nv, nf = make_heightfield.write_scene(n, TMP)
    base = TMP + "/"
    name = "c5_heightfield_%d" % n
    dump = os.path.join(TMP, name + ".dump")
    info = run(REF_DET, "scene-dump", scn, base, 64, 48, dump)
    dg = ref_io.scene_digest(ref_io.read_scene_dump(dump))
    dg["octree"] = info
    dg["ply_sha256"] = hashlib.sha256(open(os.path.join(TMP, name + ".ply"), "rb").read()).hexdigest()
    dg["vertices"], dg["triangles"] = nv, nf
    manifest["scenes"][name] = dg
    manifest["renders"] = [e for e in manifest["renders"] if e["scene"] != name]
    arrays = {}
    for policy, W, H, spp, chunk, seed in [("pixel", 64, 48, 4, 1, 99), ("chunk", 40, 30, 4, 2, 7), ("tile32", 64, 64, 1, 1, 12345)]:
        out = os.path.join(TMP, "r.f32")
        js = run(REF_DET, "render", scn, base, W, H, spp, seed, policy, out, chunk)
        key = "%s_%dx%d_%dspp_c%d_s%d" % (policy, W, H, spp, chunk, seed)
        arrays[key] = np.fromfile(out, "<f4").reshape(H, W, 3)
        manifest["renders"].append(dict(scene=name, key=key, policy=policy, width=W, height=H, spp=spp, chunk=chunk, seed=seed,
                                        shapes_tested=js["shapes_tested"], final_rng=js["final_rng"]))
    np.savez_compressed(os.path.join(HERE, "renders_%s.npz" % name), **arrays)


def golden_showcase():
    """a crop of an image the reference itself wrote (showcase/1.hdr): header, file size, 32x64 RGBE pixels"""
    raw = open("/root/reference/showcase/1.hdr", "rb").read()
    hdr = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n+Y 720 +X 1280\n"
    assert raw.startswith(hdr)
    px = np.frombuffer(raw[len(hdr):], dtype=np.uint8).reshape(720, 1280, 4)
    np.savez_compressed(os.path.join(HERE, "showcase1_crop.npz"), header=np.frombuffer(hdr, dtype=np.uint8), file_size=np.int64(len(raw)),
                        rows=np.array([300, 332]), cols=np.array([560, 624]), rgbe=np.ascontiguousarray(px[300:332, 560:624]))


def main():
    if "--only-showcase" in sys.argv:
        golden_showcase()
        return
    if "--only-c5" in sys.argv:  # add / refresh the C5 fixtures without touching the others
        manifest = json.load(open(os.path.join(HERE, "manifest.json")))
        golden_c5(manifest)
        json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1, sort_keys=True)
        return
    for b in (REF_DET, REF_GLIBC):
        if not os.path.exists(b):
            sys.exit("missing %s: run `make -C oracle ref` in the dev container first" % b)
    manifest = {"generator": "tests/golden/make_golden.py", "renders": [], "scenes": {}, "raycasts": {}}

    # extra scene texts authored for coverage (reference grammar)
    room = open(DATA + "c3_bunny_room.scn").read().split("brdf 0.700000 0.700000 0.700000")[0]
    open(DATA + "letters.scn", "w").write(
        room + "brdf 0.800000 0.300000 0.300000 0.000000 0.000000 0.000000 10 0.000000 0.000000 0.000000 1.0\n"
        "mesh letterX.ply  0.500000 -0.500000 1.200000 0.4  q 0.707107 0.707107 0 0\n"
        "brdf 0.300000 0.300000 0.800000 1.000000 1.000000 1.000000 10 0.000000 0.000000 0.000000 1.0\n"
        "mesh letterY.ply  -0.800000 0.600000 1.400000 0.4  z 30 q 0.707107 0.707107 0 0\n"
        "light 2 2 1\nsphere 0.000000 0.000000 2.800000 0.9\n")
    open(DATA + "glass_room.scn", "w").write(
        room + "brdf 0.000000 0.000000 0.000000 0.000000 0.000000 0.000000 10 1.000000 1.000000 1.000000 1.4\n"
        "sphere 0.000000 0.000000 0.900000 0.5\n"
        "brdf 0.000000 0.000000 0.000000 0.200000 0.000000 0.000000 10 1.000000 0.000000 0.000000 1.2\n"
        "sphere 1.000000 1.000000 0.800000 0.3\n"
        "brdf 0.000000 0.000000 0.000000 1.000000 1.000000 1.000000 20 0.000000 0.000000 0.000000 1.0\n"
        "sphere -1.000000 0.800000 0.700000 0.3\n"
        "cylinder 1.200000 -1.200000 0.400000 -0.000000 2.400000 0.000000 0.05\n"
        "light 3 3 2\nsphere 0.000000 0.000000 2.800000 0.9\n")

    # two machine-written scenes: the closed room, then random shapes / materials / rotated meshes, numbers in
    # several spellings (plain, many digits, exponent on a float mantissa)
    for tag, sd in (("rand_a", 101), ("rand_b", 202)):
        r = np.random.default_rng(sd)

        def lit(lo, hi):
            v = r.uniform(lo, hi)
            k = int(r.integers(0, 4))
            return ("%.6f" % v) if k == 0 else ("%.3f" % v) if k == 1 else ("%.11f" % v) if k == 2 else ("%.5fe1" % (v / 10.0))
        L = [room.rstrip("\n")]
        for _ in range(14):
            kind = int(r.integers(0, 6))
            if kind == 0:
                L.append("brdf %s %s %s 0.000000 0.000000 0.000000 %d" % (lit(0.1, 0.9), lit(0.1, 0.9), lit(0.1, 0.9), r.integers(1, 200)))
            elif kind == 1:
                L.append("brdf %s %s %s %s %s %s %d %s %s %s %s" % (lit(0, 0.5), lit(0, 0.5), lit(0, 0.5), lit(0, 1), lit(0, 1), lit(0, 1), r.integers(1, 200),
                                                                 lit(0, 1), lit(0, 1), lit(0, 1), lit(1.05, 1.6)))
            elif kind == 2:
                L.append("sphere %s %s %s %s" % (lit(-2.5, 2.5), lit(-2.5, 2.5), lit(0.3, 2.5), lit(0.05, 0.6)))
            elif kind == 3:
                L.append("box %s %s %s %s %s %s" % (lit(-2.5, 2), lit(-2.5, 2), lit(0, 2), lit(0.1, 1), lit(0.1, 1), lit(0.1, 1)))
            elif kind == 4:
                L.append("cylinder %s %s %s %s %s %s %s" % (lit(-2.5, 2.5), lit(-2.5, 2.5), lit(0, 2), lit(-1.5, 1.5), lit(-1.5, 1.5), lit(-1.5, 1.5), lit(0.03, 0.3)))
            else:
                mesh, scale = [("dwarf.obj", (0.008, 0.02)), ("letterX.ply", (0.2, 0.5)), ("letterY.ply", (0.2, 0.5))][int(r.integers(0, 3))]
                rot = ("z %d " % r.integers(-180, 180)) if r.integers(0, 2) else "z %s " % lit(-180, 180)
                L.append("mesh %s  %s %s %s %s  %sq %s %s %s %s" % (mesh, lit(-1.5, 1.5), lit(-1.5, 1.5), lit(0.5, 2), lit(*scale), rot,
                                                                   lit(-1, 1), lit(-1, 1), lit(-1, 1), lit(-1, 1)))
        L.append("light 3 3 2\nsphere 0.000000 0.000000 2.800000 0.9")
        open(DATA + tag + ".scn", "w").write("\n".join(L) + "\n")

    # RNG streams (random.h)
    for seed in (12345, 1, 4294967295, 2463534242):
        path = os.path.join(HERE, "rng_%d.bin" % seed)
        run(REF_DET, "rng", seed, 64, path)

    # per-function tables
    recs = make_unit_inputs()
    recs.tofile(os.path.join(TMP, "unit_in.bin"))
    run(REF_DET, "unit", os.path.join(TMP, "unit_in.bin"), os.path.join(TMP, "unit_det.bin"))
    run(REF_GLIBC, "unit", os.path.join(TMP, "unit_in.bin"), os.path.join(TMP, "unit_glibc.bin"))
    np.savez_compressed(os.path.join(HERE, "unit_tables.npz"), records=recs,
                        ref_det=ref_io.read_unit_output(os.path.join(TMP, "unit_det.bin")),
                        ref_glibc=ref_io.read_unit_output(os.path.join(TMP, "unit_glibc.bin")))

    for name in SCENES:
        scn = DATA + name + ".scn"
        dump = os.path.join(TMP, name + ".dump")
        info = run(REF_DET, "scene-dump", scn, DATA, 64, 48, dump)
        sd = ref_io.read_scene_dump(dump)
        dg = ref_io.scene_digest(sd)
        dg["octree"] = info
        manifest["scenes"][name] = dg
        renders = RENDERS if name != "testscene" else RENDERS + [("whole", 64, 64, 4, 1, 12345)]
        arrays = {}
        for policy, W, H, spp, chunk, seed in renders:
            out = os.path.join(TMP, "r.f32")
            js = run(REF_DET, "render", scn, DATA, W, H, spp, seed, policy, out, chunk)
            key = "%s_%dx%d_%dspp_c%d_s%d" % (policy, W, H, spp, chunk, seed)
            arrays[key] = np.fromfile(out, "<f4").reshape(H, W, 3)
            entry = dict(scene=name, key=key, policy=policy, width=W, height=H, spp=spp, chunk=chunk, seed=seed,
                         shapes_tested=js["shapes_tested"], final_rng=js["final_rng"])
            if policy in ("chunk", "pixel"):
                js2 = run(REF_GLIBC, "render", scn, DATA, W, H, spp, seed, policy, out, chunk)
                g = np.fromfile(out, "<f4").reshape(H, W, 3)
                arrays[key + "__glibc"] = g
                err = np.sqrt(((g - arrays[key]) ** 2).sum(axis=2))
                entry["glibc_distance"] = dict(bit_equal_fraction=float((g.view("<u4") == arrays[key].view("<u4")).all(axis=2).mean()),
                                               mean_l2=float(err.mean()), max_l2=float(err.max()),
                                               fraction_below_1e4=float((err < 1e-4).mean()), final_rng=js2["final_rng"])
            manifest["renders"].append(entry)
        np.savez_compressed(os.path.join(HERE, "renders_%s.npz" % name), **arrays)

        # closest-hit table (raycast_top_most_node)
        import zlib
        rng = np.random.default_rng(zlib.crc32(name.encode()))
        n = 400
        o = np.stack([rng.uniform(-2.5, 14.5, n), rng.uniform(-2.5, 14.5, n), rng.uniform(0.05, 8.8, n)], axis=1).astype("<f4")
        o[: n // 2] = np.stack([rng.uniform(-1.5, 1.5, n // 2), rng.uniform(-1.8, 1.5, n // 2), rng.uniform(0.05, 2.5, n // 2)], axis=1)
        d = unit_vectors(rng, n)
        rays = np.concatenate([o, d], axis=1).astype("<f4")
        rays.tofile(os.path.join(TMP, "rays.bin"))
        run(REF_DET, "raycast", scn, DATA, os.path.join(TMP, "rays.bin"), os.path.join(TMP, "hits.bin"))
        hits = np.fromfile(os.path.join(TMP, "hits.bin"), dtype=np.dtype([("t", "<f4"), ("n", "<f4", 3), ("mat", "<u4")]))
        np.savez_compressed(os.path.join(HERE, "raycast_%s.npz" % name), rays=rays, t=hits["t"], n=hits["n"], mat=hits["mat"])

    golden_c5(manifest)
    golden_showcase()

    # cross-check values recorded by the survey, re-measured here on ref_glibc
    js = run(REF_GLIBC, "render", DATA + "testscene.scn", DATA, 64, 64, 4, 12345, "whole", os.path.join(TMP, "x.f32"))
    import hashlib
    manifest["survey_crosscheck"] = dict(shapes_tested=js["shapes_tested"], final_rng=js["final_rng"],
                                         sha256=hashlib.sha256(open(os.path.join(TMP, "x.f32"), "rb").read()).hexdigest())
    json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1, sort_keys=True)
    print("wrote fixtures to", HERE)


if __name__ == "__main__":
    main()
