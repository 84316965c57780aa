#!/bin/bash
# round 4, batch 43: two-level priorities of the pursuer waves (high through integration, low behind it): variants, other pursuer counts and batch sizes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b43; mkdir -p $O
for rep in 1 2; do for lib in "" $(ls build/variants/*.so); do
  echo "== ${lib:-product}"; HNS_LIBRARY=$lib timeout 200 python tools/small_batch.py 65536 --cylinders=8 2>&1 | grep "E="
done; done 2>&1 | tee $O/ab.txt
for lib in "" build/variants/libhns_p1100.so; do echo "== ${lib:-product}: other shapes (one evader)";
  for a in 1 2 4 6; do HNS_LIBRARY=$lib timeout 200 python tools/small_batch.py 65536 --cylinders=8 --agents=$a --steps=1000 2>&1 | grep "E=" | sed "s/^/A=$a /"; done
  HNS_LIBRARY=$lib HNS_STEP_MAPPING=tile timeout 200 python tools/small_batch.py 32768 49152 131072 262144 --cylinders=8 --steps=500 2>&1 | grep "E="
  HNS_LIBRARY=$lib timeout 200 python tools/small_batch.py 65536 --cylinders=5 --steps=1000 2>&1 | grep "E=" | sed "s/^/C=5 /"
  HNS_LIBRARY=$lib timeout 200 python tools/small_batch.py 65536 --cylinders=16 --steps=1000 2>&1 | grep "E=" | sed "s/^/C=16 /"
done 2>&1 | tee $O/shapes.txt
