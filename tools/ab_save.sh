#!/bin/bash
# Build the CURRENT kernel sources as arm A of tools/ab.sh: .ab/libA.so (a binary, git-ignored, travels to the GPU box; delete .ab/
# when the experiment is over).  Run here, before editing the sources for arm B.
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/.ab && python -m mocca_envs_amd.build --out $R/.ab/libA.so "$@" && rm -rf $R/.ab/libA.so.objs
