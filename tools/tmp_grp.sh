cd /root/repo
for K in 1024 272; do for N in 1024 512; do for M in 16384 32768 49152; do python tools/gemm_bf16_one.py $M $N $K 1 1 2; done; done; done
for M in 1024 2048; do python tools/gemm_bf16_one.py $M 1024 16384 0 0 0 8; python tools/gemm_bf16_one.py $M 272 16384 0 0 0 22; done
python tools/gemm_bf16_one.py 512 1024 16384 0 0 0 16; python tools/gemm_bf16_one.py 1024 1024 16384 0 0 0 16;
for M in 16384 32768; do python tools/gemm_one.py 0 $M 1024 1024 1 1 2; python tools/gemm_one.py 0 $M 1024 272 1 1 2; python tools/gemm_one.py 0 $M 512 1024 1 1 2; done
