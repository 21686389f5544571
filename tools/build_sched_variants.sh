# Builds of the library with ONE LLVM instruction scheduler for every translation unit (default / max-ilp / iterative-ilp) into
# tools/experiments/libspangpu_sched_<name>.so (git-ignored; they travel to the GPU box with the snapshot).  The product library
# chooses a scheduler per unit by measurement (csrc/Makefile); tests/test_sched_variants_gpu.py runs the receivers' and the echo
# canceller's parity tests against these builds too: the lanes of a channel hand data over through LDS inside one wavefront, and
# an ordering that only holds under one scheduler's instruction order shows as a mismatch under another.
# Usage: bash tools/build_sched_variants.sh [default max-ilp iterative-ilp]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/tools/experiments
for v in ${@:-default max-ilp iterative-ilp}; do
  B=/tmp/spangpu_sched/$v
  rm -rf $B; mkdir -p $B/spandsp_amd/csrc $B/include
  cp $ROOT/include/*.h $B/include/
  cp $ROOT/spandsp_amd/csrc/*.hip $ROOT/spandsp_amd/csrc/*.hpp $ROOT/spandsp_amd/csrc/*.inc $ROOT/spandsp_amd/csrc/*.c $ROOT/spandsp_amd/csrc/*.h $ROOT/spandsp_amd/csrc/Makefile $B/spandsp_amd/csrc/
  make -C $B/spandsp_amd/csrc -j${JOBS:-8} FORCE_SCHED=$v > $B/build.log 2>&1 || { tail -20 $B/build.log; exit 1; }
  cp $B/spandsp_amd/libspangpu.so $ROOT/tools/experiments/libspangpu_sched_$(echo $v | tr - _).so
  echo "built $v"
done
