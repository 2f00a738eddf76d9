#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/tune_lm.py --sweep "split_down:0,6,10,12,16,0;split_qkv:0,5,9,12,0;split_o:0,6,10,12,0"
