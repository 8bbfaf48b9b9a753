#!/bin/bash
# SQ / TCC counter passes for one command (rocprofv3 --pmc, program directly after "--"), csv per pass under
# gpurun_out/<tag>/passN; summarise with tools/summarise_counters.py. Separate passes: 8 SQ slots per pass, and
# FETCH_SIZE / WRITE_SIZE do not fit one TCC pass (MI355X_MICROARCH.md, rocprofv3 PMC slots).
#   usage: tools/collect_counters.sh <tag> <program> [args...]
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p "$out"
cd "$GRAFT_REPO_ROOT"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT"
P3="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_WAVES"
P4="FETCH_SIZE"
P5="WRITE_SIZE"
P6="GRBM_GUI_ACTIVE"
i=0
for p in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6"; do
  i=$((i+1))
  rocprofv3 --pmc $p --output-format csv -d "$out/pass$i" -- "$@" > "$out/pass$i.log" 2>&1 || echo "pass $i failed (see $out/pass$i.log)"
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- "$@" > "$out/trace.log" 2>&1 || echo "trace failed"
echo "counters collected under gpurun_out/$tag"
