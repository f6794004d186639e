#!/bin/bash
# A/B against an earlier commit on the SAME GPU box: builds `git archive <commit>` into ab_old/ (git-ignored, travels with
# the gpurun snapshot); the probes under tools/ take PLK_AB_ROOT=ab_old to import that tree instead of the working tree.
# usage (in the container): tools/ab_build.sh <commit>
set -e
cd "$(dirname "$0")/.."
rm -rf ab_old; mkdir ab_old
git archive "$1" plonkit_amd include | tar -x -C ab_old
( cd ab_old && python -m plonkit_amd.build > /dev/null )
echo "$1" > ab_old/COMMIT; ls -la ab_old/plonkit_amd/lib/
