#!/bin/bash
# fvad_engine_run's lane-group schedules (context option run_groups: sixteenths per group) on tools/pcie_run.py's five cases:
#     bash tools/pcie_sweep.sh [schedule ...]      (on the GPU box, from the repository root)
S=${*:-"4,4,4,4 2,2,4,8 1,3,4,8 4,4,8 4,12 2,6,8 4,8,4 2,4,4,4,2 1,3,4,4,3,1 4,4,4,2,2 2,2,4,4,4 8,8 8,4,4 4,4,4,3,1"}
for g in $S; do
  echo "== $g"
  python tools/pcie_run.py run_groups=$g 2>/dev/null | grep -v "last NSNet2"
done
