cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02Q
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_INSTS_SMEM -d gpurun_out/r02Q/pmc -o q -- python tools/bench_cnn.py --steps 2 --warmup 1 > gpurun_out/r02Q/pmc.log 2>&1
python tools/pmc_conv.py gpurun_out/r02Q/pmc/q_results.db
tail -3 gpurun_out/r02Q/pmc.log
