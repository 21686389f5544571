# Builder's helper: the product library, and next to it an instrumented copy (phase stamps in the quad kernels) for tools/quad_prof.py.
set -e
cd "$(dirname "$0")/../spandsp_amd/csrc"
touch modem_api.hip
make -j8 EXTRA=-DSPG_QUAD_PROF > /dev/null
cp ../libspangpu.so ../libspangpu_prof.so
touch modem_api.hip
make -j8 > /dev/null
echo built
