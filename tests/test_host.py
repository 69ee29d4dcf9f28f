"""Host side of the product on CPU: loaders (.scn / PLY / OBJ, bit-exact with the reference's
number lexer and placement), trees, HDR writer, the C ABI surface and its error behaviour.
No compute calls: the render path needs a GPU and must say so."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import ref_io
from conftest import DATA, GOLDEN, ROOT, assert_bits_equal

SCENES = ["testscene", "c2_analytic", "c3_bunny_room", "c4_dwarf_room", "letters", "glass_room", "rand_a", "rand_b"]
# + the 99 458-triangle decimation of BASELINE.json's synthetic-mesh config (generated, see conftest.load_scene)
SCENES_C5 = SCENES + ["c5_heightfield_224"]


# ---- C ABI surface ----------------------------------------------------------------------
def test_library_exports_every_declared_symbol(api):
    L = api.lib()
    for name in api.EXPORTS:
        assert hasattr(L, name), name
    hdr = open(os.path.join(ROOT, "include", "ort.h")).read()
    import re
    declared = set(re.findall(r"\b(ort_[a-z0-9_]+)\s*\(", hdr)) - {"ort_unit_eval_device)"}
    assert declared == set(api.EXPORTS)
    assert L.ort_abi_version() == 3


def test_dynamic_symbols_are_c_linkage(api):
    out = subprocess.check_output(["nm", "-D", "--defined-only", api.LIB_PATH]).decode()
    names = {line.split()[-1] for line in out.splitlines() if line.strip()}
    for name in api.EXPORTS:
        assert name in names, name


def test_render_without_device_fails_loudly(api, load_scene):
    """no CPU fallback: rendering a scene that is not resident on a GPU is an error."""
    s = api.Scene.load_scn(os.path.join(DATA, "c2_analytic.scn")).commit()
    with pytest.raises(api.OrtError) as e:
        s.render(8, 8, 1, 1)
    assert e.value.code == api.ERR_NO_DEVICE
    out = np.zeros((8, 8, 3), "<f4")
    with pytest.raises(api.OrtError) as e:
        s.tiled_raytrace(out, 0, 0, 8, 8, 1, 1)
    assert e.value.code == api.ERR_NO_DEVICE


def test_call_order_and_argument_errors(api):
    s = api.Scene.load_scn(os.path.join(DATA, "c2_analytic.scn"))
    with pytest.raises(api.OrtError) as e:
        s.tree_info()
    assert e.value.code == api.ERR_STATE
    with pytest.raises(api.OrtError) as e:
        s.render(8, 8, 1, 1)
    assert e.value.code == api.ERR_STATE
    s.commit()
    for kwargs in [dict(width=0, height=8, spp=1, seed=1), dict(width=8, height=8, spp=0, seed=1),
                   dict(width=8, height=8, spp=6, seed=1, policy="chunk", chunk=4),
                   dict(width=8, height=8, spp=1, seed=1, rect=(4, 4, 4, 8)),
                   dict(width=8, height=8, spp=1, seed=1, rect=(0, 0, 9, 8))]:
        with pytest.raises(api.OrtError) as e:
            s.render(**kwargs)
        assert e.value.code == api.ERR_INVALID, kwargs
    with pytest.raises(api.OrtError) as e:
        api.Scene.load_scn(os.path.join(DATA, "does_not_exist.scn"))
    assert e.value.code == api.ERR_IO


# ---- scene ingestion ------------------------------------------------------------------------
@pytest.mark.parametrize("name", SCENES_C5)
def test_loader_matches_reference_dump(api, manifest, load_scene, name):
    """materials, shapes, light list, placed vertices, indices, mesh AABBs and camera basis are
    byte-identical to what the reference holds after its own parse + placement
    (parser.cpp:1184-1446,384-570,687-982; macos_main.mm:342-414,550-556)."""
    flat = load_scene(name).flatten(64, 48)
    flat.root_aabb = None
    got = ref_io.scene_digest(flat)
    want = manifest["scenes"][name]
    for key in ("counts", "camera_bits", "ambient_bits", "materials_sha256", "spheres_sha256", "boxes_sha256",
                "cylinders_sha256", "lights", "meshes", "sha256"):
        assert got[key] == want[key], key


def test_bunny_counts(load_scene):
    info = load_scene("c3_bunny_room").info()
    assert info.triangle_count == 69451
    assert load_scene("c4_dwarf_room").info().triangle_count == 1896
    assert load_scene("testscene").info().light_count == 12


def _parse_one_number(api, literal):
    """pushes a literal through the .scn lexer via a sphere radius"""
    s = api.Scene.parse_scn("brdf 0.5 0.5 0.5 0.0 0.0 0.0 10\nsphere 0.0 0.0 0.0 %s\n" % literal)
    return s.flatten(8, 8).spheres["r"][0]


def test_number_lexer_quirk(api):
    """parser.cpp:158-250: digits accumulate in f64 and are scaled by (double)0.1f per character,
    so values are typically 1-3 ulp off strtof ("5.553228" -> 5.55322838)."""
    v = _parse_one_number(api, "5.553228")
    assert np.float32(v) == np.float32(5.55322838)
    assert np.float32(v) != np.float32(5.553228)
    cam = api.Scene.load_scn(os.path.join(DATA, "testscene.scn")).flatten(8, 8).camera[0]
    assert abs(cam[0] - 5.55323) < 1e-5
    assert np.float32(_parse_one_number(api, "-0.5")) == np.float32(-0.5)
    assert _parse_one_number(api, "2.5e+1") == pytest.approx(25.0, rel=1e-6)
    assert _parse_one_number(api, "2.5e-2") == pytest.approx(0.025, rel=1e-6)


def test_scn_grammar_errors_where_the_reference_asserts(api):
    # sphere fields must be float literals (eat_and_check_scn_token, parser.cpp:1135-1142)
    with pytest.raises(api.OrtError) as e:
        api.Scene.parse_scn("brdf 0.5 0.5 0.5 0.0 0.0 0.0 10\nsphere 0 0 0 1\n")
    assert e.value.code == api.ERR_PARSE
    # light colours must be integer literals (parser.cpp:1236-1248)
    with pytest.raises(api.OrtError) as e:
        api.Scene.parse_scn("light 1.0 1.0 1.0\n")
    assert e.value.code == api.ERR_PARSE
    # tabs are not whitespace (parser.cpp:143-156): "sphere\t1.0" is ONE token, so a field goes missing
    with pytest.raises(api.OrtError) as e:
        api.Scene.parse_scn("brdf 0.5 0.5 0.5 0.0 0.0 0.0 10\nsphere\t1.0 1.0 1.0 1.0\n")
    assert e.value.code == api.ERR_PARSE
    # keywords match by prefix (parser.cpp:13-32): "boxes" is "box"
    s = api.Scene.parse_scn("brdf 0.5 0.5 0.5 0.0 0.0 0.0 10\nboxes 0.0 0.0 0.0 1.0 1.0 1.0\n")
    assert s.info().box_count == 1
    # unknown words are skipped silently
    s = api.Scene.parse_scn("# comment\nbrdf 0.5 0.5 0.5 0.0 0.0 0.0 10\nsphere 0.0 0.0 0.0 1.0\n")
    assert s.info().sphere_count == 1
    # missing mesh file -> IO error (the reference would crash)
    with pytest.raises(api.OrtError) as e:
        api.Scene.parse_scn("brdf 0.5 0.5 0.5 0.0 0.0 0.0 10\nmesh nothere.ply 0.0 0.0 0.0 1.0 q 1 0 0 0\n", DATA + "/")
    assert e.value.code == api.ERR_IO


def test_light_list_rules(api):
    """spheres go on the light list iff their material is a light; EVERY cylinder does
    (parser.cpp:1262-1266,1345-1348)."""
    s = api.Scene.parse_scn(
        "brdf 0.5 0.5 0.5 0.0 0.0 0.0 10\nsphere 0.0 0.0 0.0 1.0\ncylinder 0.0 0.0 0.0 0.0 0.0 1.0 0.1\n"
        "light 1 2 3\nsphere 1.0 0.0 0.0 1.0\ncylinder 0.0 0.0 0.0 0.0 1.0 0.0 0.1\n")
    f = s.flatten(8, 8)
    assert [(int(t), int(i)) for t, i in zip(f.lights["type"], f.lights["index"])] == [(2, 0), (1, 1), (2, 1)]
    assert f.materials["is_light"].tolist() == [0, 0, 1]
    assert f.materials["emit"][2].tolist() == [1.0, 2.0, 3.0]


def test_ply_fan_triangulation(load_scene):
    """quads become (i0, previous, next) fans (parser.cpp:544-566): letterX has 2 quads, letterY 3."""
    f = load_scene("letters").flatten(8, 8)
    assert [len(m["indices"]) for m in f.meshes] == [12, 18]
    assert f.meshes[0]["indices"].tolist() == [3, 2, 1, 3, 1, 0, 7, 6, 5, 7, 5, 4]


def test_mesh_aabb_flt_min_quirk(load_scene):
    """mesh AABB max starts at FLT_MIN, the smallest POSITIVE float (macos_main.mm:383)."""
    f = load_scene("testscene").flatten(8, 8)
    mx = np.array([m["aabb_max"] for m in f.meshes])
    assert (mx >= np.float32(1.17549435e-38)).all()


# ---- trees ---------------------------------------------------------------------------------
@pytest.mark.parametrize("name", SCENES_C5)
def test_reference_compatible_octree_counts(manifest, load_scene, name):
    """ort_reftree.cpp rebuilds the reference's loose octree: same node and leaf counts."""
    ti = load_scene(name).tree_info()
    want = manifest["scenes"][name]["octree"]
    assert ti["ref_node_count"] == want["octree_nodes"]
    assert ti["ref_nonempty_leaves"] == want["octree_leaves"]


def test_fast_tree_shape(load_scene):
    ti = load_scene("c3_bunny_room").tree_info()
    assert ti["prologue_prims"] == 8  # six room slabs, the table, the light: tested outright, not in the tree
    assert load_scene("testscene").tree_info()["prologue_prims"] == 12  # its nine boxes and three largest spheres; the rest stays in the tree
    assert ti["leaf_count"] == ti["node_count"] + 1  # binary tree with leaves encoded in child words
    assert ti["max_leaf_prims"] <= 16
    assert ti["max_depth"] <= 60  # kTreeDepthBudget (ort_scene.h): the kernels' smallest stack, 20 LDS + 40 scratch entries
    assert ti["node_bytes"] == ti["node_count"] * 64


def test_empty_and_tiny_scenes_commit(api):
    mats = np.zeros(2, api.MATERIAL_DTYPE)
    mats["diffuse"][1] = 0.5
    s = api.Scene.from_arrays(mats).commit()
    assert s.tree_info()["node_count"] == 1
    sph = np.zeros(1, api.SPHERE_DTYPE)
    sph["r"] = 1
    sph["mat"] = 1
    s = api.Scene.from_arrays(mats, spheres=sph).commit()
    ti = s.tree_info()  # a lone analytic shape is tested outright (prologue), the tree stays empty
    assert ti["prologue_prims"] == 1 and ti["leaf_count"] == 0 and ti["node_count"] == 1
    with pytest.raises(api.OrtError):
        bad = sph.copy()
        bad["mat"] = 7
        api.Scene.from_arrays(mats, spheres=bad)


# ---- output ---------------------------------------------------------------------------------
def test_rgbe_matches_oracle(api, oracle):
    """v3_to_rgbe (macos_main.mm:242-261): scale 255 with roundf, zero below 1e-32."""
    rng = np.random.default_rng(5)
    vals = (10.0 ** rng.uniform(-6, 3, size=(500, 3))).astype("<f4")
    vals[0] = 0
    vals[1] = (1e-33, 0, 0)
    vals[2] = (1, 1, 4)
    vals[3] = (0.5, 0.25, 0.125)
    L = oracle.lib()
    for r, g, b in vals:
        assert api.rgbe(r, g, b) == L.oracle_rgbe(r, g, b)
    assert api.rgbe(0, 0, 0) == 0
    # frexp(4) = 0.5 * 2^3; denom = 0.5 * 255 / 4 = 31.875; roundf(127.5) = 128 (half away from zero)
    assert api.rgbe(1, 1, 4) == (32 | (32 << 8) | (128 << 16) | ((3 + 128) << 24))


def test_hdr_file_layout(api, oracle, tmp_path):
    """header text, uncompressed RGBE, buffer row H-1 written first (macos_main.mm:263-287,683-707)."""
    rng = np.random.default_rng(6)
    img = rng.uniform(0, 3, size=(5, 7, 3)).astype("<f4")
    p1, p2 = str(tmp_path / "a.hdr"), str(tmp_path / "b.hdr")
    api.write_hdr(p1, img)
    assert oracle.lib().oracle_write_hdr(p2.encode(), img.ctypes.data, 7, 5) == 0
    a = open(p1, "rb").read()
    assert a == open(p2, "rb").read()
    header = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n+Y 5 +X 7\n"
    assert a.startswith(header)
    assert len(a) == len(header) + 5 * 7 * 4  # the showcase files obey the same size rule: 50 + 1280*720*4
    first = np.frombuffer(a[len(header):len(header) + 4], "<u4")[0]
    assert first == api.rgbe(*img[4, 0])


# ---- the command-line driver (replacement for main(), macos_main.mm:289-710) -------------------------
CLI = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "offline_raytracer_amd", "bin", "ort_render")


def test_cli_usage_and_failures(api, tmp_path):
    """argument errors exit 2; a missing scene exits 1 with the library's message; on a machine without a
    HIP device the render step fails loudly instead of falling back to anything"""
    import subprocess
    assert os.path.exists(CLI), "bin/ort_render is built by the csrc Makefile"
    assert subprocess.run([CLI], capture_output=True).returncode == 2
    assert subprocess.run([CLI, "--scene"], capture_output=True).returncode == 2
    r = subprocess.run([CLI, "--scene", str(tmp_path / "nope.scn")], capture_output=True, text=True)
    assert r.returncode == 1 and "load failed" in r.stderr
    if api.device_count() == 0:
        r = subprocess.run([CLI, "--scene", os.path.join(DATA, "c2_analytic.scn"), "--width", "16", "--height", "8", "--spp", "1",
                            "--out", str(tmp_path / "x.hdr")], capture_output=True, text=True)
        assert r.returncode != 0 and not os.path.exists(tmp_path / "x.hdr")
        assert "device" in r.stderr.lower()


def test_mesh_paths_with_dotted_directories(api, tmp_path):
    """the reference's get_extension stops at the first '.' of the whole path (parser.cpp:91-108: "does not
    work if there was a directory with ."); the product uses the file name's last '.'"""
    import shutil
    d = tmp_path / "scene.v1.dir"
    d.mkdir()
    for f in ("c4_dwarf_room.scn", "dwarf.obj"):
        shutil.copy(os.path.join(DATA, f), d)
    assert api.Scene.load_scn(str(d / "c4_dwarf_room.scn")).info().triangle_count == 1896
    assert api.Scene.load_scn(os.path.join(DATA, "..", "data", "c3_bunny_room.scn")).info().triangle_count == 69451
    (d / "bad.scn").write_text("brdf 0.5 0.5 0.5 0.0 0.0 0.0 10\nmesh dwarf.xyz 0.0 0.0 0.0 1.0\n")
    with pytest.raises(api.OrtError):
        api.Scene.load_scn(str(d / "bad.scn"))


def test_rgbe_consistent_with_reference_output_file(api):
    """tests/golden/showcase1_crop.npz holds the header and a 32x64-pixel crop of showcase/1.hdr, an image
    the reference itself wrote (v3_to_rgbe lives in the macOS-only translation unit and cannot be compiled
    here, so this is the only reference OUTPUT the writer can be held against).  Header text and file size
    follow the same rule as ours; every pixel is 0 or has its largest mantissa byte in [128, 255] (frexp
    normalisation); and decoding a pixel (byte * 2^(e-128) / 255) and encoding it again with the product's
    encoder reproduces the reference's four bytes (except where the top byte is 255: m = 1.0 renormalises)."""
    z = np.load(os.path.join(GOLDEN, "showcase1_crop.npz"))
    header = bytes(z["header"])
    assert header == b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n+Y 720 +X 1280\n"
    assert int(z["file_size"]) == len(header) + 1280 * 720 * 4
    px = z["rgbe"].astype(np.int64).reshape(-1, 4)
    top = px[:, :3].max(axis=1)
    nonzero = px[:, 3] > 0
    assert ((top[nonzero] >= 128) & (top[nonzero] <= 255)).all()
    checked = 0
    for r, g, b, e in px[nonzero & (top < 255)]:
        s = float(np.ldexp(1.0, int(e) - 128)) / 255.0
        want = int(r) | int(g) << 8 | int(b) << 16 | int(e) << 24
        assert api.rgbe(np.float32(r * s), np.float32(g * s), np.float32(b * s)) == want
        checked += 1
    assert checked > 1500
