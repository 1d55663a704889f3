#!/bin/bash
# round-3 GPU session 8: every GEMM / conv shape of the SDXL step (+ SD3 block GEMMs) under every tile family (forced), to refit
# the tile cost model of pick_tile (gemm.hip)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
: > $O/r03_s8_tile_sweep.txt
v() { local label=$1; shift; env GEMM_VARIANTS_ALL=1 "$@" timeout 200 python scripts/gemm_variants.py --label "$label" --rounds 2 --reps 6 2>&1 | grep -v "amdgpu.ids" >> $O/r03_s8_tile_sweep.txt; }
v model_p1.00 MI355X_SD_GEMM_P257=1.0
v tile128 MI355X_SD_GEMM_TILE=128
v tile129 MI355X_SD_GEMM_TILE=129
v tile160 MI355X_SD_GEMM_TILE=160
v tile257 MI355X_SD_GEMM_TILE=257
v tile320 MI355X_SD_GEMM_TILE=320
grep -v VARIANT_TIMES $O/r03_s8_tile_sweep.txt
