#!/bin/bash
# usage: tools/ab_build.sh <name> <source.hip> [extra hipcc flags ...]
# A/B builds of ONE translation unit: compiles <source.hip> (a path; e.g. an older revision of csrc/snarf.hip checked out to /tmp, or the
# tree's own with -D switches) with the flags build.py uses for the unit of that name and links it with the tree's other objects into
# intrinsicavatar_amd/_ab/libia_amd_<name>.so; select it with IA_AMD_LIB (tools/search_ab.py runs a list of them on one box).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; SRC=$2; shift 2
TU=$(basename "$SRC" .hip)
mkdir -p $ROOT/intrinsicavatar_amd/_ab
python -c "import sys; sys.path.insert(0, '$ROOT'); from intrinsicavatar_amd import build; build.build()"
O=$ROOT/intrinsicavatar_amd/_ab/${TU}_$NAME.o
FLAGS=$(python -c "import sys; sys.path.insert(0, '$ROOT'); from intrinsicavatar_amd import build; print(' '.join(build.COMMON + build.SOURCES['$TU.hip']))")
/opt/rocm/bin/hipcc $FLAGS -I$ROOT/intrinsicavatar_amd/csrc "$@" -c $SRC -o $O
OBJS=$(ls $ROOT/intrinsicavatar_amd/_obj/*.o | grep -v "/${TU}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/intrinsicavatar_amd/_ab/libia_amd_$NAME.so $OBJS $O
echo built intrinsicavatar_amd/_ab/libia_amd_$NAME.so
