set -e
B="timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --p-n 0"
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), "Mreads/s kernel_ms", round(d["roofline"]["kernel_ms"],3))'
$B --miss 0 --phred 1 | python -c "$J" R150_m0_noPhred
$B --miss 0 --phred 1 --read-len 32 | python -c "$J" R32_m0_noPhred
$B --miss 0 --read-len 32 | python -c "$J" R32_m0_phred
$B --miss 1 --read-len 32 | python -c "$J" R32_m1_phred
$B --miss 1 --read-len 20 | python -c "$J" R20_m1_phred
$B --miss 0 --phred 1 --reads 2000000 | python -c "$J" R150_m0_noPhred_2Mreads
$B --miss 1 --reads 2000000 | python -c "$J" R150_m1_2Mreads
