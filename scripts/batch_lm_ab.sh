#!/bin/bash
# A/B of the solve kernel's footprint with several sequences side by side, on one box (two rounds, interleaved)
for round in 1 2; do
  for v in "" "LSA_LM_CACHE=0" "LSA_LM_RECORDS=1024" "LSA_LM_RECORDS=2048" "LSA_LM_CACHE=0 LSA_LM_RECORDS=1024" "LSA_LM_BLOCKS=16 LSA_LM_RECORDS=2048"; do
    for S in 4 8; do
      env $v timeout -k 10 200 python scripts/batch_lm_ab.py $S 2>/dev/null | tail -1
    done
  done
done
