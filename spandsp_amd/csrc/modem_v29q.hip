// modem_v29q.hip -- the four-lanes-per-channel V.29 receiver kernel in a translation unit of its own, compiled with the
// iterative ILP scheduler (Makefile; the measurements are in modem_v27q.hip's header).  The kernel itself is v29_quad.hpp
// (reference: src/v29rx.c:400-965).
#include <hip/hip_runtime.h>

#include "v29_quad.hpp"

namespace spg {

void launch_v29_quad(const V29Launch &L, hipStream_t stream)
{
    hipLaunchKernelGGL((v29_quad_kernel<16, 4>), dim3((L.n_ch + 63)/64), dim3(256), 0, stream, L);
}

}   // namespace spg
