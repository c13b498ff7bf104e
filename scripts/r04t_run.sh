set -e
mkdir -p gpurun_out/r04t
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "streamed or stream or one_frame or timeout or batch_equals or hipgraph" > gpurun_out/r04t/pytest.log 2>&1
for i in 1 2; do
  for rows in 64 96 48 32; do
    echo "## two-ended, band $rows" >> gpurun_out/r04t/trace.log
    LMX_STREAM_BAND_ROWS=$rows timeout -k 10 120 python scripts/single_frame_trace2.py 400 >> gpurun_out/r04t/trace.log 2>&1
  done
  echo "## LMX_ONE_STORE_THREAD=1 band 96" >> gpurun_out/r04t/trace.log
  LMX_STREAM_BAND_ROWS=96 LMX_ONE_STORE_THREAD=1 timeout -k 10 120 python scripts/single_frame_trace2.py 400 >> gpurun_out/r04t/trace.log 2>&1
done
echo ALL DONE
