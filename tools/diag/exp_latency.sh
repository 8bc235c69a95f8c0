for e in 0 64 128 192; do
  TS_CXXFLAGS=-DTS_EXP=$e python -c "from thunder_speech_amd import build; build.build(force=True, verbose=False)" > /dev/null 2>&1
  echo "TS_EXP=$e"; python tools/bench_one.py 512 512 63 0 20 2>&1 | tail -1; python tools/bench_one.py 512 512 51 0 20 2>&1 | tail -1
done
