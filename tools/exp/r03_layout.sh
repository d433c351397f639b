#!/bin/bash
# which streams share a hardware queue at N = 1, and does another first-use order step faster?
cd $GRAFT_REPO_ROOT
run() {
  S4F_LAYOUT_REPORT=1 S4F_PRETOUCH="$1" timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile 2>&1 >/dev/null | grep "^\[layout\]"
}
run ""
run "decode,aux,side,opt"
run "decode,aux,opt,side"
run "decode,aux,burn,side,opt"
run "decode,aux,burn,burn,side,opt"
run "decode,side,aux,opt"
run "side,decode,aux,opt"
run "opt,decode,aux,side"
run "decode,aux,side,burn,opt"
run "decode,aux,side,burn,burn,opt"
run "decode,aux,side,burn,burn,burn,opt"
run ""
