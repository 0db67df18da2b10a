#!/bin/bash
# tools/profile_default.sh <outdir>: the default `python bench.py` run under
# rocprofv3 --kernel-trace --stats (on the GPU box, through gpurun); summarise with
#   python tools/summarize_profile.py <outdir> r02
set -e
out=${1:-gpurun_out/prof}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py > $out/bench_default.log 2> $out/bench_default.err
