#!/bin/bash
# A longer builder-run soak (GPU box, ~20 min): 3 000 random scenes under a fresh seed, ALL 8 192 segments of the headline
# batch against the oracle, the per-candidate trace of 1 024 configs[2] segments and 600 scenes.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
O=${1:-gpurun_out/soak_big}; mkdir -p $O
( time WSPR_SCENES=3000 WSPR_SCENE_SEED=${SEED:-777} python -m pytest tests/test_gpu_parity.py -q -x -k "randomised_scenes" ) > $O/scenes_3000.txt 2>&1
( time WSPR_CONFIG3_ORACLE_SEGMENTS=8192 python -m pytest tests/test_gpu_configs.py -q -x -k "config3_8192" ) > $O/config3_all_8192_oracle_segments.txt 2>&1
( time WSPR_TRACE_SCENES=600 WSPR_TRACE_CONFIG3=1024 python tests/trace_parity.py scenes config3 ) > $O/trace_600_scenes_1024_segments.txt 2>&1
tail -n 5 $O/*.txt
