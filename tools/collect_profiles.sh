set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2p
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2p/fwd -- python3 bench.py --mode fwd --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2p/fwd.json 2> gpurun_out/r2p/fwd.err && \
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2p/default -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r2p/default.json 2> gpurun_out/r2p/default.err && \
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r2p/fetch -- python3 bench.py --mode fwd --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2p/fetch.json 2> gpurun_out/r2p/fetch.err && cp gpurun_out/bench_layers.json gpurun_out/r2p/layers_fetch.json && \
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r2p/write -- python3 bench.py --mode fwd --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2p/write.json 2> gpurun_out/r2p/write.err && \
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d gpurun_out/r2p/pmc_a -- python3 bench.py --steps 1 --warmup 1 --scenes 1 --no-cpu-baseline > gpurun_out/r2p/pmc_a.json 2> gpurun_out/r2p/pmc_a.err && \
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/r2p/pmc_b -- python3 bench.py --steps 1 --warmup 1 --scenes 1 --no-cpu-baseline > gpurun_out/r2p/pmc_b.json 2> gpurun_out/r2p/pmc_b.err
echo done rc=$?
