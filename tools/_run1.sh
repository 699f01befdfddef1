mkdir -p gpurun_out/r06
a="--f 200 --solver lu"
bash tools/ab_env.sh "$a" "CUMF_ALS_LIB=/root/repo/variants/libALS_head.so" "CUMF_ALS_LIB=-" 1 5 2 2>&1 | tee gpurun_out/r06/ab_wg_lu_blocked3.txt
bash tools/ab_env.sh "$a" "CUMF_ALS_LIB=/root/repo/variants/libALS_nslp.so" "CUMF_ALS_LIB=-" 2 5 2 2>&1 | tee -a gpurun_out/r06/ab_wg_lu_blocked3.txt
python tools/check_large_lu.py 160 200 206 2>&1 | grep -v amdgpu.ids | tail -6
