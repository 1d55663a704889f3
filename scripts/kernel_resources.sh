#!/bin/bash
# Register / spill table of one HIP source (compiler remarks; runs without a GPU):  scripts/kernel_resources.sh gemm_pipe.hip [extra flags]
cd "$(dirname "$0")/../paddlemix_amd/csrc"
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast "$@" -Rpass-analysis=kernel-resource-usage -c $f -o /dev/null 2>&1 |
  python3 -c '
import re, subprocess, sys
rows, cur = [], {}
for ln in sys.stdin:
    m = re.search(r"remark:\s+(Function Name|VGPRs|SGPRs Spill|VGPRs Spill|SGPRs|ScratchSize \[bytes/lane\]): (\S+)", ln)
    if not m: continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    else: cur[k] = v
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
for r, n in sorted(zip(rows, names), key=lambda x: x[1]):
    n = n.replace("void ", "").replace("sd::", "").replace("(GemmArgs)", "").replace("GemmCfg", "")
    print("%4s vgpr %4s vspill %4s sspill %5s scratch  %s" % (r.get("VGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"), r.get("ScratchSize [bytes/lane]"), n))
'
