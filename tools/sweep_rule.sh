#!/bin/bash
# On the GPU box: team grid (workgroups: variant libraries under librectify_amd/build/variants) x hand-over threshold of
# regional frames (tiles), verdict "16 long walks": long bars, natural frame (single and batch), synthetic frames 3, 4.
cp librectify_amd/librectify_amd.so /tmp/lib_keep.so
export LIBRECTIFY_FLOOD_T1_REGIONAL_RULE=1
for g in ${1:-512 1024}; do cp librectify_amd/build/variants/lib_g$g.so librectify_amd/librectify_amd.so; for t in ${2:-64 48 32 24}; do
  export LIBRECTIFY_FLOOD_T1_REGIONAL=$t
  echo "== team grid $g, regional hand-over at $t tiles"
  python3 tools/run_long.py 2>&1 | tail -1 | sed 's/.*second_tier_seeds/long  second_tier_seeds/; s/.slabs.*\[/ [/'
  python3 tools/run_doc4k.py 2>&1 | tail -1 | sed 's/.*second_tier_seeds/doc4k second_tier_seeds/; s/.slabs.*\[/ [/'
  python3 tools/run_doc4k_batch.py 2>&1 | tail -1
done; done
cp /tmp/lib_keep.so librectify_amd/librectify_amd.so
