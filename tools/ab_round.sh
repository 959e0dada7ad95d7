#!/bin/bash
# Same-box comparison of this tree with another checkout of the repository (e.g. last round's final commit as a git worktree
# under _r3/, library built in place): bench.py of both, alternating.
#   git worktree add -f _r3 <commit> && (cd _r3 && python -m musicgeneration_amd._build)      (here)
#   bash tools/ab_round.sh _r3 [rounds=2]                                                      (GPU box, repo root)
OTHER=${1:?path of the other checkout}; R=${2:-2}
one() { (cd $1 && timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-decode --no-kernel-timing $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), 'ms', round(d['value']/1e6,3), 'M events/s')"); }
for r in $(seq $R); do one $OTHER ""; one . "--no-cfg4"; done
