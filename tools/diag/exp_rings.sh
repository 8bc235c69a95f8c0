# Compile-time pipeline depths of the split kernel: weight-fragment ring (TS_SPLIT_RING) and window prefetch distance (TS_WIN_DIST)
for f in "" "-DTS_SPLIT_RING=4" "-DTS_WIN_DIST=2" "-DTS_SPLIT_RING=3"; do
  TS_CXXFLAGS="$f" python -c "from thunder_speech_amd import build; build.build(force=True, verbose=False)" > /dev/null 2>&1
  echo "flags: $f"; python tools/bench_tcs.py 2>&1 | tail -1
done
