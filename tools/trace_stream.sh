# timeline of the pipelined path: kernel + memory-copy trace of 40 batches at depth 3 (rocprofv3), then the overlap summary
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/trace_stream
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/trace_stream -o t -- python $R/tools/bench_stream.py --batches ${BATCHES:-40} ${DEPTHS:---only-depth ${DEPTH:-4}} "$@" > $R/gpurun_out/trace_stream/run.log 2>&1
cd $R
python tools/trace_overlap.py gpurun_out/trace_stream
