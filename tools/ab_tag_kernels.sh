set -e
O=gpurun_out/abtag; mkdir -p $O
L=tianshou_marl_amd/lib
cp $L/base.so $L/libtsmarl_hip.so
timeout -k 10 200 python tools/stamp_rollout_tag.py > $O/stamp_base.txt 2>&1
bash tools/job_step_kernels.sh tag > $O/kern_base.txt 2>&1
cp $L/new.so $L/libtsmarl_hip.so
timeout -k 10 200 python tools/stamp_rollout_tag.py > $O/stamp_new2.txt 2>&1
bash tools/job_step_kernels.sh tag > $O/kern_new.txt 2>&1
head -3 $O/kern_base.txt $O/kern_new.txt
