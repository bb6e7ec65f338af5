#!/bin/bash
# round 4: generic attention kernel (T = 1024, cfg 5) old vs new softmax, per-op times of one px128 forward, same box
D=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc
O=gpurun_out/${TAG:-r4_attn1024}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention" > $O/test.txt 2>&1; echo "pytest rc=$?" >> $O/test.txt; tail -3 $O/test.txt
grep -q "rc=0" $O/test.txt || exit 1
{
for rep in 1 2; do
for v in old new; do
  lib=$D/libmi355_sampler.so; [ $v = old ] && lib=$D/libmi355_sampler_oldattn.so
  MI355_SAMPLER_LIB=$lib timeout -k 10 300 python bench.py --workload px128_inpaint_ddim100_b128 --steps 1 --warmup 1 --nfe 2 --no-cpu-baseline --profile-out $O/p_$v.json > /dev/null 2>&1
  echo "== $v"; python tools/show_profile.py $O/p_$v.json | grep -E "^forward|attention"
done
done
} 2>&1 | tee $O/times.txt
