set -e
B="timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline"
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), "Mreads/s kernel_ms", round(d["roofline"]["kernel_ms"],3))'
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
$B | python -c "$J" m1
$B --p-n 0 | python -c "$J" m1_noN
$B --miss 0 --p-n 0 | python -c "$J" m0_noN
$B --miss 0 --p-n 0 --phred 1 | python -c "$J" m0_noN_noPhred
$B --miss 0 --p-n 0 --phred 1 --reads 2000000 | python -c "$J" m0_noN_noPhred_2M
$B --p-n 0 --reads 2000000 | python -c "$J" m1_noN_2M
