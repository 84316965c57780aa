#!/bin/bash
# round 3, batch 77: the whole GPU suite on the product build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
