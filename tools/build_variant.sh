#!/bin/bash
# development aid (build container): builds the library of another git revision (or of the working tree with extra -D flags)
# into tools/_libs/lib_<tag>.so for tools/ab.sh.   usage: tools/build_variant.sh <tag> <git rev | WORK> [extra hipcc flags]
set -e
tag=$1; rev=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
if [ "$rev" = WORK ]; then
  mkdir -p $tmp/rnb-neus-fork_amd $tmp/include
  cp -r $root/rnb-neus-fork_amd/csrc $tmp/rnb-neus-fork_amd/ && cp $root/include/rnbneus.h $tmp/include/
else
  (cd $root && git archive $rev rnb-neus-fork_amd/csrc include) | tar -x -C $tmp
fi
mkdir -p $tmp/rnb-neus-fork_amd/build $root/tools/_libs
echo '#define RNB_BUILD_ID "variant-'$tag'"' > $tmp/rnb-neus-fork_amd/build/build_id.h
cd $tmp/rnb-neus-fork_amd/csrc
pids=()
for f in *.hip; do
  extra=""
  case $f in sampling.hip|raygen.hip) extra="-ffp-contract=off";; sweep_mv.hip) extra="-fno-slp-vectorize";; esac
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fvisibility=hidden -Wno-unused-function $extra "$@" -c $f -o ../build/${f%.hip}.o &
  pids+=($!)
  if [ ${#pids[@]} -ge 6 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/_libs/lib_$tag.so ../build/*.o
rm -rf $tmp
echo built tools/_libs/lib_$tag.so
