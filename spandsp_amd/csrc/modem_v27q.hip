// modem_v27q.hip -- the four-lanes-per-channel V.27ter receiver kernel in a translation unit of its own, so that it can be
// compiled with the instruction scheduler that suits it (Makefile).  A lone wave per SIMD is bound by the latency of its
// dependent chains (profiles/r4_probe_issue.log: 8.4 cycles from an instruction to the one that reads its result, 5.2 between
// independent ones); LLVM's ILP schedulers spread the chains further apart than the default (occupancy-minded) one does.
// Measured at 16 384 channels x 160 samples, microseconds per launch by tools/bench_paths.py's clock
// (profiles/r4_sched_max_ilp_ab.log, r4_sched_iterative_ilp_ab.log, r4_sched_quad_units.log):
//                      default     -amdgpu-sched-strategy=max-ilp     =iterative-ilp
//   V.27ter quad        113.8                104.5                        106.4
//   V.29 quad           155.6                155.1                        148.0
//   V.17 quad           214.6                228.4                        197.8
// so V.27ter takes max-ilp (here), V.29 and V.17 iterative-ilp (modem_v29q.hip, modem_v17q.hip), and the one-lane-per-channel
// kernels of the big banks stay in modem_api.hip with the default.  Scheduling does not touch results: the same parity tests.
// (Reference: src/v27ter_rx.c:863-1028; the kernel itself is v27ter_quad.hpp.)
#include <hip/hip_runtime.h>

#include "v27ter_quad.hpp"

namespace spg {

void launch_v27ter_quad(const V27Launch &L, hipStream_t stream)
{
    hipLaunchKernelGGL((v27ter_quad_kernel<16, 4>), dim3((L.n_ch + 63)/64), dim3(256), 0, stream, L);
}

}   // namespace spg
