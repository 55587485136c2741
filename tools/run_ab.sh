set -e
timeout -k 10 300 python tools/plan_sweep.py 4096 40 "10,5,4096,4;10,5,3900,1;10,5,3800,1;10,5,3600,1;10,5,7800,1;10,10,3900,1;8,5,3900,1;8,5,4800,1" >> gpurun_out/ab.txt 2>&1
for cfg in "10 10 8192 1" "10 10 3952 1" "10 10 3800 1" "10 10 3648 1" "10 10 7600 1" "10 5 3800 1" "14 10 2898 1" "8 10 4700 1"; do
  set -- $cfg
  echo "cfg kj=$1 kd=$2 waves=$3 minchunk=$4" >> gpurun_out/slab_sweep.txt
  FLUID_TB_KJ=$1 FLUID_TB_KD=$2 FLUID_TB_WAVES=$3 FLUID_TB_MINCHUNK=$4 timeout -k 10 120 python tools/slab_time.py 16384 40 8 2>&1 | grep -v amdgpu >> gpurun_out/slab_sweep.txt
done
