#!/bin/bash
# Builder-run soak beyond the defaults of `pytest -m gpu` (GPU box): more random scenes, more configs[2] segments against
# the oracle, the per-candidate trace on more scenes / segments / crowded bands, full-size raw segments.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
O=${1:-gpurun_out/soak}; mkdir -p $O
( time WSPR_SCENES=1200 WSPR_SCENE_SEED=40404 python -m pytest tests/test_gpu_parity.py -q -x -k "randomised_scenes" ) > $O/scenes_1200.txt 2>&1
( time WSPR_CONFIG3_ORACLE_SEGMENTS=2048 python -m pytest tests/test_gpu_configs.py -q -x -k "config3_8192" ) > $O/config3_2048_oracle_segments.txt 2>&1
( time WSPR_TRACE_SCENES=300 WSPR_TRACE_CONFIG3=256 python tests/trace_parity.py scenes config3 ) > $O/trace_300_scenes_256_segments.txt 2>&1
( time python tools/crowded_soak.py ) > $O/crowded_trace_48x4.txt 2>&1
( time python tools/raw_soak.py 6 4100 ) > $O/raw_soak_6.txt 2>&1
tail -n 6 $O/*.txt
( time WSPR_HASH_SEGMENTS=1536 WSPR_HASH_SEED=5 python -m pytest tests/test_gpu_hashtable.py -q -x -s -k "0.05" ) > $O/hashtable_1536_segments.txt 2>&1
tail -n 4 $O/hashtable_1536_segments.txt
