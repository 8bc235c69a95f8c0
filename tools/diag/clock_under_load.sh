# What does the shader clock do while the split kernel runs?  (rocm-smi samples next to a looping layer benchmark)
python tools/bench_one.py 512 512 63 0 60000 > /tmp/loop.log 2>&1 &
PID=$!
sleep 4
for i in 1 2 3 4 5; do rocm-smi --showclocks 2>/dev/null | grep -iE "sclk|mclk" | head -2; rocm-smi --showpower 2>/dev/null | grep -iE "power" | head -1; sleep 0.5; done
wait $PID
tail -1 /tmp/loop.log
echo idle:; rocm-smi --showclocks 2>/dev/null | grep -iE "sclk" | head -1
