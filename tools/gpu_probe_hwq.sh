#!/bin/bash
# do the pass's compute streams and the upload queues get in each other's way in the runtime's hardware queues?  (GPU_MAX_HW_QUEUES, default 4)
exec </dev/null
cd $GRAFT_REPO_ROOT
for q in 4 8 12; do
  echo "GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q timeout 300 python tools/gpu_block_pass_leg.py 2>/dev/null | python -c "
import json, sys
d = json.load(sys.stdin)
for k in ('flags_only', 'with_memo_seeding', 'two_in_flight_arrival_pipeline', 'two_in_flight_arrival_pipeline_with_memo_seeding', 'three_callers_flags_only', 'distinct_creators', 'idemix_every_5th_creator'):
    v = d.get(k, {})
    print('   %-52s %.3f' % (k, v.get('ms_per_block_aggregate', v.get('median_ms_per_block', -1))))"
done
