"""Developer script (dev container): copy the measurements of tools/r3_stamp_call.sh (gpurun_out/r3final2: PMC summaries, scaling
proxy, phase shares) and tools/r3_benchlines.sh (gpurun_out/r3lines: bench lines, kernel stats) into profiles/ under their round-3
names, make the PMC stamps, and print the figures DESIGN.md quotes.  usage: python3 tools/r3_collect.py"""
import json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r3final2")
P = os.path.join(ROOT, "profiles")
LINES = os.path.join(ROOT, "gpurun_out", "r3lines")
for d, a, b in ((LINES, "bench.json", "r03_bench.json"), (LINES, "bench_c2.json", "r03_bench_c2.json"), (LINES, "bench_c4.json", "r03_bench_c4.json"),
                (LINES, "bench_c5.json", "r03_bench_c5.json"), (LINES, "bench_tile32.json", "r03_bench_tile32.json"),
                (LINES, "rehearsal.json", "r03_bench_rehearsal_2ranks_one_gpu.json"), (LINES, "bench_kernel_stats.csv", "r03_bench_kernel_stats.csv"),
                (SRC, "scaling_proxy.json", "r03_scaling_proxy.json")):
    if os.path.exists(os.path.join(d, a)):
        shutil.copy(os.path.join(d, a), os.path.join(P, b))
for tag, key in (("c3", "c3_bunny_room 1920x1080 1024spp"), ("c2", "c2_analytic 1920x1080 1024spp"), ("c4", "c4_dwarf_room 3840x2160 512spp"),
                 ("c5", "c5_heightfield_708 3840x2160 256spp")):
    shutil.copy(os.path.join(SRC, "stamps", tag + "_summary.txt"), os.path.join(P, "r03_pmc_summary_%s.txt" % tag))
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_pmc_stamp.py"), os.path.join(P, "r03_pmc_summary_%s.txt" % tag),
                           key + " chunk64 x1", "profiles/r03_pmc_summary_%s.txt" % tag, os.path.join(P, "r03_pmc_stamp_%s.json" % tag)], stdout=subprocess.DEVNULL)
for name, out in (("util_c3_exchange.log", "r03_util_phases_exchange.txt"), ("util_c3_plain.log", "r03_util_phases_plain.txt")):
    lines = [l for l in open(os.path.join(SRC, name)) if l.startswith(("util ", "phase "))]
    open(os.path.join(P, out), "w").writelines(lines)
VALU_PEAK = 78.65
for tag in ("c3", "c2", "c4", "c5"):
    st = json.load(open(os.path.join(P, "r03_pmc_stamp_%s.json" % tag)))
    b = json.load(open(os.path.join(P, "r03_bench%s.json" % ("" if tag == "c3" else "_" + tag))))
    k_s = b["roofline"]["kernel_ms"] * 1e-3
    useful = st["useful_lane_ops_per_launch"] / k_s / 1e12
    print("%s: %.1f Mpaths/s kernel %.2f ms | valu frac %.3f (%.1f T lane-op/s) lanes %.3f issue %.3f wait %.2f insts %.3e | hbm div frac %.3f | traffic %.1f GB = %.2f TB/s l2 %.2f | contract %.3f"
          % (tag, b["value"], b["roofline"]["kernel_ms"], useful / VALU_PEAK, useful, st["lanes_active"], st["counters"]["SQ_INSTS_VALU"] / k_s / (VALU_PEAK * 1e12 / 64),
             st["wave_cycles_waiting_on_memory"], st["counters"]["SQ_INSTS_VALU"], b["roofline"]["hbm"]["frac"], st["hbm_bytes_per_launch"] / 1e9,
             st["hbm_bytes_per_launch"] / k_s / 1e12, st["l2_hit_rate"], b["roofline"]["hbm"]["contract_frac"]))
pr = json.load(open(os.path.join(P, "r03_scaling_proxy.json")))
for n, w in pr["worlds"].items():
    print("N=%s max %.2f ms imbalance %.3f speedup %s" % (n, w["max_ms"], w["imbalance"], w["predicted_speedup"]))
b = json.load(open(os.path.join(P, "r03_bench.json")))
print("cpu baseline", b.get("cpu_baseline"))
print(open(os.path.join(P, "r03_bench_kernel_stats.csv")).read()[:600])
