"""Developer script (dev container only: it runs the reference builds oracle/_ref/ref_det and ref_glibc, compiled from
/root/reference by oracle/Makefile): the distance between the parity anchor (reference + deterministic libm) and the
as-shipped reference (reference + glibc libm) AT BASELINE SCALE -- a 128x72 window of the 1920x1080 frame at
1024 spp in 64-sample jobs (CHUNK policy), same seeds, for the C2 / C3 / C4 scenes.  A libm result that differs in its
last bit can flip a comparison somewhere along a path; everything after it in that 64-sample job then decorrelates, which
moves the pixel by O(1/16) of one job's mean.  Writes per-pixel L2 statistics into tests/golden/manifest.json
("glibc_distance_baseline"); tests/test_oracle_golden.py asserts the stated bounds.  Both builds' windows are kept as
fixtures (tests/golden/baseline_windows.npz: <scene>__det, <scene>__glibc) so that the GPU tests can hold the HIP image
against the reference's own pixels at BASELINE parameters: bit-equal to __det, within the stated bound of __glibc.
usage: python3 tools/glibc_distance_baseline.py [workers=8]"""
import json, os, subprocess, sys, tempfile
from multiprocessing import Pool
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP, CHUNK, SEED = 1920, 1080, 1024, 64, 12345
WIN = {"c2_analytic": (896, 504, 1024, 576), "c3_bunny_room": (896, 504, 1024, 576), "c4_dwarf_room": (896, 504, 1024, 576)}
BANDS = 8


def run(job):
    binary, scene, band = job
    x0, y0, x1, y1 = WIN[scene]
    ya = y0 + (y1 - y0) * band // BANDS
    yb = y0 + (y1 - y0) * (band + 1) // BANDS
    out = os.path.join(tempfile.gettempdir(), "gd_%s_%s_%d.f32" % (binary, scene, band))
    cmd = [os.path.join(ROOT, "oracle", "_ref", binary), "render", os.path.join(ROOT, "data", scene + ".scn"), os.path.join(ROOT, "data") + "/",
           str(W), str(H), str(SPP), str(SEED), "chunk", out, str(CHUNK), str(x0), str(ya), str(x1), str(yb)]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
    img = np.fromfile(out, "<f4").reshape(H, W, 3)[ya:yb, x0:x1].copy()
    os.remove(out)
    return (binary, scene, band, img)


if __name__ == "__main__":
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    jobs = [(b, s, k) for s in WIN for b in ("ref_det", "ref_glibc") for k in range(BANDS)]
    with Pool(workers) as pool:
        res = pool.map(run, jobs, chunksize=1)
    report = {}
    windows = {}
    for scene in WIN:
        imgs = {}
        for b in ("ref_det", "ref_glibc"):
            imgs[b] = np.concatenate([r[3] for r in sorted((r for r in res if r[0] == b and r[1] == scene), key=lambda r: r[2])], axis=0)
        a, g = imgs["ref_det"], imgs["ref_glibc"]
        windows[scene + "__det"], windows[scene + "__glibc"] = a, g
        l2 = np.sqrt(((a.astype(np.float64) - g.astype(np.float64)) ** 2).sum(axis=2)).ravel()
        hist_edges = [0.0, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1e9]
        hist = np.histogram(l2, bins=hist_edges)[0].tolist()
        report[scene] = {
            "window": list(WIN[scene]), "width": W, "height": H, "spp": SPP, "chunk": CHUNK, "seed": SEED, "pixels": int(l2.size),
            "bit_equal_fraction": float((a.view("<u4") == g.view("<u4")).all(axis=2).mean()),
            "fraction_below_1e4": float((l2 < 1e-4).mean()), "fraction_below_1e3": float((l2 < 1e-3).mean()),
            "max_l2": float(l2.max()), "mean_l2": float(l2.mean()), "median_l2": float(np.median(l2)),
            "l2_histogram_edges": hist_edges[:-1] + ["inf"], "l2_histogram": hist,
            "mean_value": float(a.mean()),
        }
        print(scene, json.dumps(report[scene]))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "baseline_windows.npz"), **windows)
    mpath = os.path.join(ROOT, "tests", "golden", "manifest.json")
    m = json.load(open(mpath))
    m["glibc_distance_baseline"] = report
    json.dump(m, open(mpath, "w"), indent=1, sort_keys=True)
