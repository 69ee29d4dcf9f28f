"""Developer script (no GPU): the kernel's lane code compiled for the host (tools/host_sim) against the oracle on
many seeds, scenes and roulette probabilities, in parallel processes.  Prints differences; exit 1 if any.
usage: python3 tools/cpu_hunt.py SECONDS [WORKERS] [SEED0]"""
import os, subprocess, sys, time, tempfile
from multiprocessing import Pool
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

SCENES = ["testscene", "c2_analytic", "glass_room", "rand_a", "rand_b", "c4_dwarf_room", "letters", "c3_bunny_room"]


def one(k):
    import oracle_lib
    from offline_raytracer_amd import api
    name = SCENES[k % len(SCENES)]
    seed = SEED0 + k
    rr = (0.8, 0.95, 0.8, 0.5, 0.95)[k % 5]
    policy, chunk = (("chunk", 8), ("pixel", 0))[(k // 8) % 2]
    W, H, spp = 256, 144, 16
    scn = os.path.join(ROOT, "data", name + ".scn")
    sc = api.Scene.load_scn(scn).commit()
    ref, _ = oracle_lib.OracleScene(sc.flatten(W, H)).render(W, H, spp, seed, policy, chunk=max(chunk, 1), rr=rr, threads=1)
    out = os.path.join(tempfile.gettempdir(), "hunt_%d.f32" % os.getpid())
    subprocess.run([os.path.join(ROOT, "tools", "host_sim"), scn, os.path.join(ROOT, "data") + "/", str(W), str(H), str(spp), str(seed), policy,
                    str(chunk), out], capture_output=True, env=dict(os.environ, SIM_RR=repr(rr), **({"SIM_WIDE": "1"} if (k // 3) % 2 else {})))  # every other triple of cases walks the 4-wide tree
    img = np.fromfile(out, dtype="<f4").reshape(H, W, 3)
    d = int((img.view("<u4") != ref.view("<u4")).any(axis=2).sum())
    return (name, seed, rr, policy, d)


if __name__ == "__main__":
    budget = float(sys.argv[1]); workers = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    SEED0 = int(sys.argv[3]) if len(sys.argv) > 3 else 500000
    t0 = time.time(); n = 0; bad = 0; k = 0
    with Pool(workers, initializer=lambda s=SEED0: globals().__setitem__("SEED0", s)) as pool:
        while time.time() - t0 < budget:
            for name, seed, rr, policy, d in pool.map(one, range(k, k + workers * 4)):
                n += 1
                if d:
                    bad += 1
                    print("DIFF %s seed %d rr %.2f %s: %d pixels" % (name, seed, rr, policy, d), flush=True)
            k += workers * 4
    print("%d renders of 256x144x16 (%.2e paths), %d with differences" % (n, n * 256 * 144 * 16, bad))
    sys.exit(1 if bad else 0)
