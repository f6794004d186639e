#!/bin/bash
# A/B of COMPILER FLAGS on the same GPU box: builds the working tree once more into ab_<tag>/ with extra hipcc flags
# (PLK_HIPCC_EXTRA, plonkit_amd/build.py); probes take PLK_AB_ROOT=ab_<tag>.  usage: tools/ab_flags.sh <tag> <flags...>
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
rm -rf "ab_$tag"; mkdir "ab_$tag"
cp -r plonkit_amd include "ab_$tag/"
rm -rf "ab_$tag/plonkit_amd/build" "ab_$tag/plonkit_amd/lib" "ab_$tag/plonkit_amd/__pycache__"
( cd "ab_$tag" && PLK_HIPCC_EXTRA="$*" python -m plonkit_amd.build > /dev/null )
echo "$*" > "ab_$tag/FLAGS"; ls -la "ab_$tag/plonkit_amd/lib/" | tail -2
