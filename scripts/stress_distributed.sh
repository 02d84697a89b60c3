#!/bin/bash
# On the GPU box: the distributed-vs-single-rank worker of the test-suite over more world sizes and system sizes
# (all ranks share the one GPU over gloo).  Prints one DIST_RESULT line per run.
port=29700
for world in 2 3 4 5; do
  for bodies in 4000 30000; do
    for mixed in 0 1; do
      [ "$mixed" = "1" ] && [ "$bodies" = "30000" ] && continue   # ellipsoid narrow phase on the host oracle side is slow
      port=$((port + 1))
      # (the larger rod runs with the cold tier forced on: DIST_TIER=3, see tests/dist_worker.py)
      tier=""; [ "$bodies" = "30000" ] && tier=3
      DIST_TIER=$tier DIST_BODIES=$bodies DIST_MIXED=$mixed MASTER_ADDR=127.0.0.1 timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 \
        --nproc-per-node $world --master-addr 127.0.0.1 --master-port $port tests/dist_worker.py 2>&1 | grep "DIST_RESULT\|FAIL " \
        || echo "DIST_RESULT MISSING world $world bodies $bodies mixed $mixed"
      sleep 3   # let every rank of the previous run exit: the box allows 6 processes on the GPU
    done
  done
done
