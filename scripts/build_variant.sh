#!/bin/bash
# an A/B build of the library:  scripts/build_variant.sh NAME "FLAGS" [PATCH]  ->  scripts/ab/libdynogfx_NAME.so
#   scripts/build_variant.sh base ""                                                        the tree as it is
#   scripts/build_variant.sh dma2 "-DCT_LDSDMA=1 -DCT_RING=2" scripts/ab_src/chol_ldsdma.patch   LDS-DMA operand staging (round 6: neutral, not kept)
# (only dynogfx.hip includes chol_tiles.h / kernels.h; the other objects are the default build's.  A patch is applied to a copy of csrc/.)
set -e
name=$1; flags=$2; patch=${3:-}
root="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p $root/scripts/ab
(cd $root/dynosam_amd/csrc && python build.py > /dev/null)
src=$root/dynosam_amd/csrc
if [ -n "$patch" ]; then
  tmp=$(mktemp -d); mkdir -p $tmp/dynosam_amd $tmp/include
  cp -r $root/dynosam_amd/csrc $tmp/dynosam_amd/; cp $root/include/*.h $tmp/include/
  (cd $tmp && patch -p1 -s < $root/$patch)
  src=$tmp/dynosam_amd/csrc
fi
(cd $src && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c dynogfx.hip -o $root/scripts/ab/dynogfx_$name.o)
objs=$(for s in dynoflow dynowindow dynosmoother dynoparallel dynotracker dynoformulation; do echo $root/dynosam_amd/csrc/build/$s.o; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $root/scripts/ab/dynogfx_$name.o $objs -o $root/scripts/ab/libdynogfx_$name.so
[ -n "$patch" ] && rm -rf $tmp
echo scripts/ab/libdynogfx_$name.so
