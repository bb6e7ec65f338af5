D=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc
TAG=r4_gnfin TESTS="tests/test_gpu_unet.py tests/test_gpu_ops.py" KEXPR="group or gn or norm or diagnostic or epilogue" ENVS="old:MI355_SAMPLER_LIB=$D/libmi355_sampler_oldgn.so;new:MI355_ATTN_FUSE=1" PROFILE=1 bash tools/r4_e2e_ab.sh
