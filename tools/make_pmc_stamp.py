"""Developer script (runs here, after a gpurun of tools/prof_pmc.sh on the bench workload): turns the PMC summary
into profiles/r02_pmc_stamp.json, stamped with the hash of the kernel sources it was measured on.  bench.py quotes
lanes_active / valu_issue_frac / traffic from it only while that hash matches the tree.
usage: python3 tools/make_pmc_stamp.py gpurun_out/pmc/summary.txt "<workload key>" [profiles/<copy of the summary>] [profiles/<stamp>.json]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

vals = {}
for line in open(sys.argv[1]):
    p = line.split()
    if len(p) >= 3 and p[1] == "mean_per_dispatch":
        vals[p[0]] = float(p[2])
lanes = vals["SQ_THREAD_CYCLES_VALU"] / (64.0 * vals["SQ_ACTIVE_INST_VALU"])
cycles = vals["GRBM_GUI_ACTIVE"] / 8.0  # rocprofv3 sums the 8 XCDs (MI355X_MICROARCH.md, DVFS)
valu_issue = vals["SQ_INSTS_VALU"] / (1024.0 * cycles / 2.0)  # 1024 SIMDs, one wave64 VALU instruction per 2 cycles each
# gfx950: FETCH_SIZE counts 64 B per 128 B request of wide reads (guide, HBM section): doubled; WRITE_SIZE is exact; both in KB
hbm = 2.0 * vals["FETCH_SIZE"] * 1024.0 + vals["WRITE_SIZE"] * 1024.0
commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
stamp = {
    "kernel_hash": bench.kernel_source_hash(), "git_commit_at_stamp": commit, "workload_key": sys.argv[2],
    "lanes_active": lanes, "valu_issue_frac": valu_issue, "hbm_bytes_per_launch": hbm,
    "l2_hit_rate": vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]),
    "wave_cycles_waiting_on_memory": vals["SQ_WAIT_ANY"] / vals["SQ_WAVE_CYCLES"],
    "wave_cycles_issue_stalled": vals["SQ_WAIT_INST_ANY"] / vals["SQ_WAVE_CYCLES"],
    "source": sys.argv[3] if len(sys.argv) > 3 else sys.argv[1],
    "counters": vals,
}
stamp["git_commit"] = commit
out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "profiles", "r03_pmc_stamp.json")
# useful vector lane-operations per launch: what bench.py's "valu" roofline divides by the live kernel time
stamp["useful_lane_ops_per_launch"] = vals["SQ_INSTS_VALU"] * 64.0 * lanes
json.dump(stamp, open(out, "w"), indent=1)
print(json.dumps({k: stamp[k] for k in ("kernel_hash", "lanes_active", "valu_issue_frac", "hbm_bytes_per_launch", "l2_hit_rate")}))
