#!/bin/bash
# round 4, session o: planner caps of the mixed chunks after the fragment ring (tile phase 41 k -> 21 k cycles per wave): tiles per chunk,
# hex slots per chunk, tile nodes per chunk - 10 M-dof octree mesh, symmetry classes, one process per knob.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r04o"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python tools/iter_ab.py oct10ms ebe 100 "PCG_EBE_TILE_CAP=24|12|16|32|48" > "$OUT/tilecap.json" 2> "$OUT/tilecap.log"; grep us_per "$OUT/tilecap.log" | grep "'rep': 1" | cut -c40-260
timeout 900 python tools/iter_ab.py oct10ms ebe 100 "PCG_EBE_HEX_CAP=512|448|384|320" > "$OUT/hexcap.json" 2> "$OUT/hexcap.log"; grep us_per "$OUT/hexcap.log" | grep "'rep': 1" | cut -c40-260
timeout 900 python tools/iter_ab.py oct10ms ebe 100 "PCG_EBE_NODE_CAP=768|704|640" > "$OUT/nodecap.json" 2> "$OUT/nodecap.log"; grep us_per "$OUT/nodecap.log" | grep "'rep': 1" | cut -c40-260
timeout 900 python tools/iter_ab.py oct10ms ebe 100 "PCG_EBE_XCD=0|1|4|8|16" > "$OUT/xcd.json" 2> "$OUT/xcd.log"; grep us_per "$OUT/xcd.log" | grep "'rep': 1" | cut -c40-260
