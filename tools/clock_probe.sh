# Samples clocks and power while the headline GEMM runs back to back (is the chip power-limited?).
# usage: tools/clock_probe.sh [env...]   (run on the GPU box through gpurun)
R=$GRAFT_REPO_ROOT
env "$@" $R/build/tools/gemm_bench 4096 4096 4096 20000 0 > /tmp/gb.log 2>&1 &
PID=$!
sleep 2
for i in 1 2 3 4; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power" | tr -s ' ' | head -8
  echo --
  sleep 1
done
wait $PID
tail -1 /tmp/gb.log
