set -u
T=$1
bash tools/profile_gpu.sh $T > gpurun_out/${T}_profile_gpu.log 2>&1
python3 tools/summarize_pmc.py $T > gpurun_out/${T}_summarize_pmc.log 2>&1
bash tools/profile_rows.sh > gpurun_out/${T}_profile_rows.log 2>&1
for M in semiglobal banded-affine one-vs-many packed; do cp gpurun_out/prof_rows/${M}_kernel_stats.csv gpurun_out/${T}_${M}_kernel_stats.csv 2>/dev/null; done
bash tools/profile_sg_pmc.sh > gpurun_out/${T}_profile_sg_pmc.log 2>&1
python3 tools/summarize_sg_pmc.py $T > gpurun_out/${T}_summarize_sg_pmc.log 2>&1
bash tools/profile_host_batch.sh $T packed > /dev/null 2>&1
bash tools/profile_host_batch.sh $T ovm > /dev/null 2>&1
bash tools/profile_host_batch.sh $T pairs > /dev/null 2>&1
smith-waterman-simd_amd/bin/swmi_speedtest 1048576 1048576 65536 > gpurun_out/${T}_swmi_speedtest.txt 2>&1
python3 tools/sg_sweep_matrix.py 1 1024 4096 16384 32768 49152 65536 81920 98304 131072 262144 2>&1 | grep "sweep -1" > gpurun_out/${T}_sg_matrix.txt
SWMI_SG_EXACT=1 python3 tools/sg_sweep_matrix.py 1 16384 32768 65536 131072 262144 2>&1 | grep "sweep -1" > gpurun_out/${T}_sg_matrix_exact.txt
python3 bench.py --mode banded-affine --steps 20 --warmup 3 > gpurun_out/${T}_bench_banded_affine.json 2>/dev/null
python3 bench.py > gpurun_out/${T}_bench_default.json 2>/dev/null
echo profile-all-done
