#!/bin/bash
# diag/alloc_after_exit.sh — the series for diag/alloc_after_exit.hip: each allocation right after a process that held 40 GB has exited
R=${GRAFT_REPO_ROOT:-$PWD}
hipcc --offload-arch=gfx950 -O2 -o /tmp/alloc_after_exit $R/diag/alloc_after_exit.hip || exit 1
A=/tmp/alloc_after_exit
echo "--- on a quiet device"; sleep 3; $A take 1 16; sleep 3; $A take 4 4; sleep 3; $A vmm 4 4
for rep in 1 2 3 4 5 6; do
	for spec in "take 1 16" "vmm 4 4" "take 1 11" "vmm 3 3.67" "take 2 8"; do
		$A hold 40 > /dev/null; $A $spec
	done
done
