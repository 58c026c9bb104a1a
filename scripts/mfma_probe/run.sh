#!/bin/bash
# v_mfma_f64_16x16x4_f64 issue rate with the clocks recorded beside it (is the gap between the
# measured 47.7 TFLOP/s and the data sheet's 78.6 a clock or an issue-rate effect?  Round 5: neither -- an artifact of the
# probe's loop, see rate.hip; overlap.hip: fp64 MFMA and fp64 vector FMAs share one pipe)
#   bash scripts/mfma_probe/run.sh > gpurun_out/mfma_rate.txt
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -o rate rate.hip 2>/dev/null || exit 1
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -o overlap overlap.hip 2>/dev/null || exit 1
echo "== rocm-smi before"; rocm-smi --showclocks 2>&1 | grep -iE "sclk|mclk|fclk|socclk" | head -8
( ./rate & PID=$!; sleep 0.4; echo "== rocm-smi while the probe runs"; rocm-smi --showclocks 2>&1 | grep -iE "sclk" | head -4; rocm-smi --showpower 2>&1 | grep -iE "power" | head -3; wait $PID )
echo "== rocm-smi after"; rocm-smi --showclocks 2>&1 | grep -iE "sclk" | head -4
echo "== do fp64 MFMA and fp64 vector FMAs overlap on a SIMD?"; ./overlap
