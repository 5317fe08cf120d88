# Phase stamps of the headline rollout under load (tools/stamp_rollout.py) for lib/new.so and lib/base.so on the same box.
set -e
O=gpurun_out/ab1; mkdir -p $O
L=tianshou_marl_amd/lib
cp $L/new.so $L/libtsmarl_hip.so
timeout -k 10 200 python tools/stamp_rollout.py 1024 > $O/stamp_new.txt 2>&1
cp $L/base.so $L/libtsmarl_hip.so
timeout -k 10 200 python tools/stamp_rollout.py 1024 > $O/stamp_base.txt 2>&1
cp $L/new.so $L/libtsmarl_hip.so
