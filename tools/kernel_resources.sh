#!/bin/bash
# Developer script: registers / spills / scratch of every path-trace kernel variant (hipcc -Rpass-analysis=kernel-resource-usage).
# usage: tools/kernel_resources.sh [extra hipcc flags]
cd "$(dirname "$0")/../offline_raytracer_amd/csrc"
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -ffp-contract=off -fno-math-errno --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt \
  -Rpass-analysis=kernel-resource-usage "$@" -c ort_kernels.hip -o /tmp/ort_k.o 2>&1 | python3 -c '
import re, sys
cur = None
rows = {}
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark: +(\w[^:]*): (\S+)", line)
    if m and cur: rows[cur][m.group(1).strip()] = m.group(2)
for k, v in rows.items():
    if "pt_persistent" not in k and "wf_" not in k: continue
    name = k.replace("_ZN3ort", "").replace("EvNS_9SceneViewENS_9RenderHotE", "")
    print(name, " ".join("%s=%s" % (a, b) for a, b in v.items()))
'
