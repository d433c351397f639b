#!/bin/bash
# one-sweep attention backward: timing-only ablation builds (S4F_FB_ABL bit mask, see attn_bwd.hip) against the shipped build,
# same box, same call.  Build here first:  for a in 1 2 4 ...; do S4F_FB_ABL=$a S4F_LIB_OUT=$PWD/tools/exp/_fbv/libs4f_abl$a.so python -m s4former_amd.build; done
cd $GRAFT_REPO_ROOT
python3 tools/exp/fb_time.py
for f in tools/exp/_fbv/libs4f_abl*.so; do S4F_LIB=$GRAFT_REPO_ROOT/$f python3 tools/exp/fb_time.py; done
python3 tools/exp/fb_time.py
