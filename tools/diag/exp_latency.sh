# Timing experiments on the split kernel (wrong results, timing only): which per-stage input is the stage time waiting for?
#   TS_EXP bit 64: taps fetched once   128: input rows fetched once   256: consumer A fragments read once   512: weight fragments loaded once
for e in ${EXPS:-0 64 128 192 256 512 768 960}; do
  TS_CXXFLAGS=-DTS_EXP=$e python -c "from thunder_speech_amd import build; build.build(force=True, verbose=False)" > /dev/null 2>&1
  echo "TS_EXP=$e"; python tools/bench_one.py 512 512 63 0 20 2>&1 | tail -1
done
