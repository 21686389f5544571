# A build of the library beside the product one, for A-B runs on one box: tools/experiments/libspangpu_<name>.so (git-ignored; travels
# to the GPU box with the snapshot; SPANGPU_LIB=<path> makes spandsp_amd/engine.py load it).
# Usage: [ONLY=modem_v17q] bash tools/build_variant.sh <name> [worktree | <commit>] [extra make arguments, e.g. EXTRA=-DSOMETHING]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; from=${2:-worktree}; shift; shift || true
B=/tmp/spangpu_variant/$name
rm -rf $B; mkdir -p $B/spandsp_amd/csrc $B/include $ROOT/tools/experiments
if [ "$from" != worktree ]; then
  git -C $ROOT archive $from include spandsp_amd/csrc | tar -x -C $B
else
  cp $ROOT/include/*.h $B/include/
  cp $ROOT/spandsp_amd/csrc/*.hip $ROOT/spandsp_amd/csrc/*.hpp $ROOT/spandsp_amd/csrc/*.inc $ROOT/spandsp_amd/csrc/*.c $ROOT/spandsp_amd/csrc/*.h $ROOT/spandsp_amd/csrc/Makefile $B/spandsp_amd/csrc/
fi
if [ -n "$ONLY" ]; then
  # ONLY=<translation unit>: every other object is taken from the product build (which must be up to date)
  cp -p $ROOT/spandsp_amd/csrc/*.o $ROOT/spandsp_amd/csrc/*.co $B/spandsp_amd/csrc/
  rm -f $B/spandsp_amd/csrc/$ONLY.o
  touch -d "2000-01-01" $B/spandsp_amd/csrc/*.hpp $B/spandsp_amd/csrc/*.inc $B/spandsp_amd/csrc/*.hip $B/spandsp_amd/csrc/*.c $B/spandsp_amd/csrc/*.h $B/include/*.h
fi
make -C $B/spandsp_amd/csrc -j${JOBS:-8} "$@" > $B/build.log 2>&1 || { tail -20 $B/build.log; exit 1; }
cp $B/spandsp_amd/libspangpu.so $ROOT/tools/experiments/libspangpu_$name.so
echo "built $name"
