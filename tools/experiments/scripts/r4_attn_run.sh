D=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc
mkdir -p gpurun_out/r4_attn_stamps
TAG=r4_attn4 TESTS="tests/test_gpu_unet.py" KEXPR="persistent_kernel" ENVS="old:MI355_ATTN_FUSE=3;new:MI355_ATTN_FUSE=1;prio:MI355_SAMPLER_LIB=$D/libmi355_sampler_attn512.so" bash tools/r4_e2e_ab.sh || exit 1
for v in 256 768; do MI355_SAMPLER_LIB=$D/libmi355_sampler_attn$v.so timeout -k 10 300 python bench.py --steps 1 --warmup 0 --nfe 2 --no-cpu-baseline > gpurun_out/r4_attn_stamps/out$v.txt 2>&1; echo "== $v"; grep "attn stamps" gpurun_out/r4_attn_stamps/out$v.txt | tail -4; done
ABLS="0 512 0 512" bash tools/experiments/scripts/r4_attn_ablate.sh
