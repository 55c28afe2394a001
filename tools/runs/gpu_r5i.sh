#!/bin/bash
# r05 call I: BASELINE config 5 with the reference's own composition list in stage 1 at the precision that list needs
# ("high"), beside the r04 line's configuration (text + image, mixed) on the same box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05i; mkdir -p $O
timeout 500 python bench.py --config tft2v_sr600 --stage1 vcomposer --stage1-precision high --steps 50 > $O/two_stage_vcomposer_high.json 2> $O/two_stage_vcomposer_high.err
tail -c 900 $O/two_stage_vcomposer_high.json; tail -3 $O/two_stage_vcomposer_high.err
timeout 400 python bench.py --config tft2v_sr600 --steps 50 > $O/two_stage_text_image_mixed.json 2> $O/two_stage_text_image_mixed.err
tail -c 700 $O/two_stage_text_image_mixed.json
