set -e
timeout -k 10 600 python -u -m pytest tests -m gpu -q -x --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
for n in 16384 4096 1024; do
timeout -k 10 200 python tools/plan_sweep.py $n 40 ",,," >> gpurun_out/ab.txt 2>&1
done
