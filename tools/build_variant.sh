#!/bin/bash
# A/B builds: tools/build_variant.sh <tag> <file-to-rebuild.hip> "<extra flags>"  ->  rtlsdr-wsprd_amd/libwspr_mi355x.so.<tag>
# (the default library is rebuilt afterwards; on the GPU box: cp the variant over libwspr_mi355x.so before a run)
set -e
cd "$(dirname "$0")/.."
touch rtlsdr-wsprd_amd/csrc/$2
WSPR_EXTRA_FLAGS="$3" bash rtlsdr-wsprd_amd/csrc/build.sh > /dev/null
cp rtlsdr-wsprd_amd/libwspr_mi355x.so rtlsdr-wsprd_amd/libwspr_mi355x.so.$1
touch rtlsdr-wsprd_amd/csrc/$2
bash rtlsdr-wsprd_amd/csrc/build.sh > /dev/null
echo "built variant $1"
