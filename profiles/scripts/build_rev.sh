#!/bin/bash
# build_rev.sh <git-rev> <out.so> [DEFINES...]: the kernel library of an earlier commit, for same-box A/B against the working tree
# (profiles/kernel_bench.py --lib <out.so>).
set -e
rev=$1; out=$(realpath -m $2); shift 2
tmp=$(mktemp -d /tmp/drtk_rev.XXXXXX)
git archive $rev drtk_amd include | tar -x -C $tmp
python $tmp/drtk_amd/build.py --variant $out "$@"
rm -rf $tmp
