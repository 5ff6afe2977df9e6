timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_pl.log 2>&1; tail -3 gpurun_out/gpu_tests_pl.log
SW_VARIANTS='[{}, {"KG_PART_CHUNKS": 5}, {"KG_PART_CHUNKS": 1}]' timeout -k 10 280 python tools/sweep_scan.py > gpurun_out/sw_pl.jsonl 2> gpurun_out/sw_pl.err
