// modem_v17q.hip -- the four-lanes-per-channel V.17 receiver kernel in a translation unit of its own, compiled with the
// iterative ILP scheduler (Makefile; the measurements are in modem_v27q.hip's header).  The kernel itself is v17_quad.hpp
// (reference: src/v17rx.c:396-1295).
#include <hip/hip_runtime.h>

#include "v17_quad.hpp"

namespace spg {

void launch_v17_quad(const V17Launch &L, hipStream_t stream)
{
    hipLaunchKernelGGL((v17_quad_kernel<16, 4>), dim3((L.n_ch + 63)/64), dim3(256), 0, stream, L);
}

}   // namespace spg
