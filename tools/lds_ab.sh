L=gpurun_build
timeout 600 python tools/ab_bench.py --libs $L/libdfx_s66.so,$L/libdfx_s68.so,$L/libdfx_s72.so,$L/libdfx_s67.so,$L/libdfx_s65.so,$L/libdfx_s80.so --pairs 128 --distinct --steps 20 --rounds 2 2>&1 | grep -v amdgpu.ids | grep round
export TMPDIR=/tmp
for v in s66 s68 s65; do
DFX_LIB=$PWD/$L/libdfx_$v.so timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-include-regex "k_sfm_step" --output-format csv -d /tmp/lds_$v -o pmc -- python tools/ab_bench.py --worker --pairs 128 --distinct --steps 3 --preroll 5 > /dev/null 2>&1
echo $v; python tools/pmc_summary.py /tmp/lds_$v
done
