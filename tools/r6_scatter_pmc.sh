#!/bin/bash
# round 6: request counters of the two bucket scatters that are in the tree (wave2 = default, KC_SCATTER_V1 = round 4's) on the cfg[2] build alone
# (tools/r5_build_ab.py), and the build's wall time back to back / after an idle second.  Through gpurun, repo root.
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for var in "" "KC_SCATTER_V1=1"; do
  echo "== variant: ${var:-default (wave2)}"
  bash tools/r5_build_pmc.sh 2500000 "$var" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_STALL_sum" "GRBM_GUI_ACTIVE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" 2>&1 | grep -v "^$"
done
cd $GRAFT_REPO_ROOT
echo "== wall, back to back"; AB_REPS=4 python3 tools/r5_build_ab.py 2500000 "" "KC_SCATTER_V1=1" 2>&1 | tail -2
echo "== wall, after an idle second before every build"; AB_REPS=4 AB_SLEEP=1 python3 tools/r5_build_ab.py 2500000 "" "KC_SCATTER_V1=1" 2>&1 | tail -2
