cd $GRAFT_REPO_ROOT
for e in 0 1 2 4 3 7 16 48 55 63; do echo "TS_PWX=$e: $(TS_PWX=$e timeout 100 python tools/diag/pw_tile_bench.py 2>&1 | grep 'Cin= 512 Cout= 512' | cut -c20-70)"; done
