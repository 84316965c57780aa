#!/bin/bash
# round 4, batch 70: the seeded step sweep at 2 000 configurations, once with the library's choice of mapping (small for whole tiles, one evader, k <= 4) and once
# with the tile mapping forced; the predictor sweep at 400
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export HNS_FUZZ_SEEDS=2000 HNS_FUZZ_TP_SEEDS=400
( time timeout 1500 python -m pytest tests/test_hip_fuzz.py -q -x -k "bit_exact" -p no:cacheprovider 2>&1 | tail -3 ) 2>&1 | tail -6
( time HNS_STEP_MAPPING=tile timeout 1500 python -m pytest tests/test_hip_fuzz.py -q -x -k "bit_exact" -p no:cacheprovider 2>&1 | tail -3 ) 2>&1 | tail -6
( time timeout 1500 python -m pytest tests/test_hip_tp.py -q -x -k random_configuration -p no:cacheprovider 2>&1 | tail -3 ) 2>&1 | tail -6
