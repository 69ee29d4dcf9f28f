"""Developer script: sweep kernel build variants (ORT_LIB) and the refill threshold on the GPU box."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1].split(",") if len(sys.argv) > 1 else ["w2", "w3", "w4", "w5"]
thr = sys.argv[2].split(",") if len(sys.argv) > 2 else ["16", "24", "32", "40", "48"]
scene = sys.argv[3] if len(sys.argv) > 3 else "c3_bunny_room"
for l in libs:
    for t in thr:
        env = dict(os.environ, ORT_LIB=os.path.join(ROOT, "offline_raytracer_amd", "lib", "libort_%s.so" % l), ORT_REFILL_BELOW=t)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_run.py"), scene, "1920", "1080", "64", "64", "3"], env=env, capture_output=True, text=True)
        best = max([float(x.split("->")[1].split()[0]) for x in out.stdout.splitlines() if "->" in x] or [0])
        print("lib %s refill_below %s : %.1f Mpaths/s %s" % (l, t, best, out.stderr.strip()[-200:] if best == 0 else ""), flush=True)
