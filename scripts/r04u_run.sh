set -e
mkdir -p gpurun_out/r04u
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r04u/pytest.log 2>&1
ROUNDS=3 timeout -k 10 300 bash scripts/ab_quick.sh variants/liblmx_dqlean_LEAN.so variants/liblmx_pipe.so > gpurun_out/r04u/ab.txt 2>&1
for i in 1 2; do
  echo "## pipelined depth labels" >> gpurun_out/r04u/trace.log
  timeout -k 10 120 python scripts/single_frame_trace2.py 400 >> gpurun_out/r04u/trace.log 2>&1
  echo "## previous build (LMX_SO_PATH=variants/liblmx_dqlean_LEAN.so: no two-ended stores either)" >> gpurun_out/r04u/trace.log
  LMX_SO_PATH=variants/liblmx_dqlean_LEAN.so timeout -k 10 120 python scripts/single_frame_trace2.py 400 >> gpurun_out/r04u/trace.log 2>&1
done
echo ALL DONE
