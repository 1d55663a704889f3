#!/bin/bash
# Round 6, session 28: does MI355X_SD_NO_W4 reach the debug library from a Python process? kernel names under rocprofv3, both arms;
# and the W8A8 FF2 launch timed in both arms
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp
cat > /tmp/t.py <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from paddlemix_amd import ops, _lib
ops.init(0)
M, N, K = 32768, 1536, 6144
g = torch.Generator(device="cuda").manual_seed(1)
qa = torch.randint(0, 120, (M, K), device="cuda", dtype=torch.uint8, generator=g)
qw = torch.randint(0, 120, (N, K), device="cuda", dtype=torch.uint8, generator=g)
sa = torch.rand(M, device="cuda") + 0.5
sw = torch.rand(N, device="cuda") + 0.5
for _ in range(20): out = ops.linear_f8(qa, sa, qw, sw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(300): out = ops.linear_f8(qa, sa, qw, sw)
e1.record(); torch.cuda.synchronize()
a = torch.randn(8192, 1280, device="cuda").to(torch.bfloat16); w = torch.randn(3840, 1280, device="cuda").to(torch.bfloat16)
for _ in range(5): ops.linear(a, w, None)
torch.cuda.synchronize()
print(os.path.basename(_lib.LIB_PATH), "NO_W4=", os.environ.get("MI355X_SD_NO_W4"), "w8a8 32768x1536x6144: %.1f us" % (e0.elapsed_time(e1) * 1000 / 300))
PY
for arm in w4 now4; do
  if [ $arm = now4 ]; then export MI355X_SD_NO_W4=1; else unset MI355X_SD_NO_W4; fi
  rm -rf /tmp/p28
  MI355X_SD_LIB=dbg rocprofv3 --kernel-trace --stats -d /tmp/p28 -o r --output-format csv -- python /tmp/t.py > /tmp/p28.log 2>&1
  grep "w8a8" /tmp/p28.log
  F=$(find /tmp/p28 -name "*kernel_stats.csv" | head -1)
  echo "== $arm kernels:"; cut -d, -f1,2,4 $F | grep -i "gemm" | head -6
done > $O/r06_s28_switch_check.txt 2>&1
cat $O/r06_s28_switch_check.txt | cut -c1-200
