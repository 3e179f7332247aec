mkdir -p gpurun_out/c2
timeout 900 python -m pytest tests/test_gpu_bf16_mixed.py tests/test_gpu_hdemucs.py tests/test_gpu_conv.py -x -q > gpurun_out/c2/pytest.log 2>&1; tail -5 gpurun_out/c2/pytest.log
for i in 1 2; do
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('z16time on ', d['ms_per_step'])"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-enc-z16-time 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('z16time off', d['ms_per_step'])"
done
