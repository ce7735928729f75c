mkdir -p gpurun_out/tk3
H=scripts/matrix_xcd_hist
run() { tag=$1; shift; timeout 200 $H "$@" > gpurun_out/tk3/$tag.txt 2>&1; echo "$tag: $(grep '^# by' gpurun_out/tk3/$tag.txt | tr '\n' ' ')"; }
run icount icount 1024 6
grep -A18 "^launch 3" gpurun_out/tk3/icount.txt | cut -c1-220
run a_1024_4_default 1024 4 12 2
run b_1024_8_2x256_1x64 1024 8 12 2 2 256 1 64
run c_1024_8_2x128_1x64 1024 8 12 2 2 128 1 64
run d_1024_16_4x256_1x64 1024 16 12 2 4 256 1 64
run e_1024_16_4x512_1x128 1024 16 12 2 4 512 1 128
run f_1024_4_1x128 1024 4 12 2 1 128 1 0
run g_1024_8_4x256_1x128 1024 8 12 2 4 256 1 128
run h_1024_16_8x256_2x128 1024 16 12 2 8 256 2 128
run i_8192_16_default 8192 16 8 2
run j_8192_16_4x256_1x64 8192 16 8 2 4 256 1 64
