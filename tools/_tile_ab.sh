export PYTHONUNBUFFERED=1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "costs_argmin or layouts or ragged or baseline_configs" 2>&1 | tail -2
timeout -k 10 200 python tools/sweep.py "S,0,256,4096,50" "S,0,1024,4096,50" "S,0,256,4096,30" "S,0,1,262144,50" "S,0,64,16384,65" 2>&1 | grep -v "amdgpu.ids"
