#!/bin/bash
# Save the current kernel sources as arm A of tools/ab.sh (.ab_src/ travels to the GPU box, is not tracked by git)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/.ab_src && cp $R/mocca_envs_amd/csrc/*.h $R/mocca_envs_amd/csrc/*.hip $R/.ab_src/
