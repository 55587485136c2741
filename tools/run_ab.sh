set -e
for cfg in "10 10 3900 1" "10 10 2048 1" "10 10 1536 1" "14 10 2048 1" "14 10 1536 1" "14 10 1024 1" "8 10 2048 1" "8 10 3900 1" "5 5 3900 1" "5 5 8192 1" "14 10 3072 1"; do
  set -- $cfg
  echo "cfg kj=$1 kd=$2 waves=$3 minchunk=$4" >> gpurun_out/slab_sweep.txt
  FLUID_TB_KJ=$1 FLUID_TB_KD=$2 FLUID_TB_WAVES=$3 FLUID_TB_MINCHUNK=$4 timeout -k 10 120 python tools/slab_time.py 8192 200 8 4 4 2>&1 | grep -v amdgpu >> gpurun_out/slab_sweep.txt
done
