#!/bin/bash
# usage on the GPU box: bash tools/exp/exit_cost.sh   (builds nothing: tools/exp/exit_cost is built beforehand)
B=$GRAFT_REPO_ROOT/tools/exp/exit_cost
run() {
  T0=$(date +%s.%N); $B "$@" 2> /tmp/ec.txt; T1=$(date +%s.%N)
  python3 -c "
import sys,re
s=open('/tmp/ec.txt').read().strip(); b=float(re.search(r'leaves at ([0-9.]+)',s).group(1))
print('%-28s wall %.3f after-main %.3f | %s' % (' '.join(sys.argv[3:]), float(sys.argv[2])-float(sys.argv[1]), float(sys.argv[2])-b, s.split(' leaves')[0]))" $T0 $T1 "$@"
}
run 0 1 0 0 1
run 0 1 0 2048 1 0 0
run 0 1 0 2048 1 0 16
run 0 1 0 2048 1 0 4
run 0 1 0 4096 1 0 0
run 0 1 0 4096 1 0 16
