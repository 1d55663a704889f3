#!/bin/bash
# Round 4, session 11: rate, clock and board power of the two bf16 MFMA shapes with nothing else running (scripts/probes/mfma_power_probe.hip).
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
T=r04_s11
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe scripts/probes/mfma_power_probe.hip 2>/dev/null
for cfg in "16 1" "32 1" "16 0" "32 0" "16 1" "32 1"; do
  set -- $cfg
  ( for i in 1 2 3 4 5 6 7 8; do sleep 0.6; rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power \(W\)|sclk' | sed 's/.*: //' | tr '\n' ' '; echo; done > /tmp/smi_$1_$2.txt ) &
  /tmp/mfma_probe $1 5 $2
  wait
  echo "   rocm-smi during the run (sclk, W): $(sed -n '3,7p' /tmp/smi_$1_$2.txt | tr '\n' ';')"
done > $O/${T}_mfma_power_probe.txt 2>&1
cat $O/${T}_mfma_power_probe.txt
