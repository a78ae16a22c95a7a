#!/bin/bash
# Round 6: key-split attention merged by the last-arriving workgroup (ZH_ATTN_FUSED_MERGE=1: zh_attention_f16_splitk_fused) against the merge
# launch, ABAB on one box: config 3 through the drop-in module (one 480x640 image per call: 12 encoder + 6 cross-attention merges), and the
# pseudo-label path at 1 and 4 images per call (12 long-sequence merges + the decoder's).
for rep in 1 2 3; do
  for f in 0 1; do
    echo "fused=$f c3: $(ZH_ATTN_FUSED_MERGE=$f python3 tools/c3_bench.py 2>/dev/null | grep "exact.*480x640")"
    echo "fused=$f pseudo: $(ZH_ATTN_FUSED_MERGE=$f python3 tools/selfmask_prof_run.py 1 exact 2>/dev/null | tail -1) | $(ZH_ATTN_FUSED_MERGE=$f python3 tools/selfmask_prof_run.py 4 exact 2>/dev/null | tail -1)"
  done
done
