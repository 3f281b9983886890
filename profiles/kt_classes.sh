cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/kt && \
LSX_SERIAL=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt/c3 -o kt -- python3 profiles/steptime.py c3 1000 10 > gpurun_out/kt/c3.log 2>&1 && \
LSX_SERIAL=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt/c4 -o kt -- python3 profiles/steptime.py c4 1250 10 > gpurun_out/kt/c4.log 2>&1 && \
python3 profiles/kt_sum.py $(find gpurun_out/kt/c3 -name '*kernel_trace.csv') 13 > gpurun_out/kt/c3_sum.txt && python3 profiles/kt_sum.py $(find gpurun_out/kt/c4 -name '*kernel_trace.csv') 13 > gpurun_out/kt/c4_sum.txt; tail -2 gpurun_out/kt/c3.log gpurun_out/kt/c4.log; find gpurun_out/kt -name '*kernel_trace.csv' -delete
