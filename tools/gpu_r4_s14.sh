cd $GRAFT_REPO_ROOT
for e in 0 4 1 5 2 6 7 16 20 23 15; do TS_PW_EXP=$e timeout 120 python tools/diag/pw_tile_bench.py 2>&1 | grep TS_PW; echo; done
