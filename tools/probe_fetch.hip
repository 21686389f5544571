// tools/probe_fetch.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of this
// repository's kernels (MI355X_MICROARCH.md, HBM section: the counters are calibrated for wide coalesced reads only --
// "calibrate on a known byte count in your own access pattern").  Every kernel moves exactly BYTES bytes, each byte once:
//   rd_dword_coalesced     lane == element, one dword per lane and instruction        (tone banks' state rows, modem state words)
//   rd_x4_coalesced        16 bytes per lane, lanes adjacent                           (the calibrated case; the tone frames' LDS-DMA)
//   rd_x4_lane_rows        every lane walks its own 128 contiguous bytes, 8 x dwordx4  (echo: fir_taps32 slices)
//   rd_short_lane_rows     every lane walks its own 64 contiguous bytes, 32 x sshort   (echo: fir_taps16 / history slices)
//   wr_dword_coalesced, wr_x4_lane_rows, wr_short_lane_rows: the stores of the same shapes
// Run under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); tools/gpu_round5.sh calib.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_fetch.hip -o tools/probe_fetch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr size_t BYTES = 1ull << 30;

__global__ void rd_dword_coalesced(const int *p, int *sink, size_t n)
{
    size_t i = (size_t) blockIdx.x*blockDim.x + threadIdx.x;
    int acc = 0;
    for (;  i < n;  i += (size_t) gridDim.x*blockDim.x)
        acc += p[i];
    if (acc == 0x12345678)
        *sink = acc;
}

__global__ void rd_x4_coalesced(const int4 *p, int *sink, size_t n)
{
    size_t i = (size_t) blockIdx.x*blockDim.x + threadIdx.x;
    int acc = 0;
    for (;  i < n;  i += (size_t) gridDim.x*blockDim.x)
    {
        const int4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 0x12345678)
        *sink = acc;
}

__global__ void rd_x4_lane_rows(const int4 *p, int *sink, size_t rows)
{
    size_t r = (size_t) blockIdx.x*blockDim.x + threadIdx.x;
    int acc = 0;
    for (;  r < rows;  r += (size_t) gridDim.x*blockDim.x)
    {
#pragma unroll
        for (int k = 0;  k < 8;  k++)
        {
            const int4 v = p[r*8 + k];
            acc += v.x + v.y + v.z + v.w;
        }
    }
    if (acc == 0x12345678)
        *sink = acc;
}

__global__ void rd_short_lane_rows(const short *p, int *sink, size_t rows)
{
    size_t r = (size_t) blockIdx.x*blockDim.x + threadIdx.x;
    int acc = 0;
    for (;  r < rows;  r += (size_t) gridDim.x*blockDim.x)
    {
#pragma unroll
        for (int k = 0;  k < 32;  k++)
            acc += *(const volatile short *) (p + r*32 + k);               // (volatile: no merging into wider loads)
    }
    if (acc == 0x12345678)
        *sink = acc;
}

__global__ void wr_dword_coalesced(int *p, size_t n)
{
    size_t i = (size_t) blockIdx.x*blockDim.x + threadIdx.x;
    for (;  i < n;  i += (size_t) gridDim.x*blockDim.x)
        p[i] = (int) i;
}

__global__ void wr_x4_lane_rows(int4 *p, size_t rows)
{
    size_t r = (size_t) blockIdx.x*blockDim.x + threadIdx.x;
    for (;  r < rows;  r += (size_t) gridDim.x*blockDim.x)
    {
#pragma unroll
        for (int k = 0;  k < 8;  k++)
            p[r*8 + k] = make_int4((int) r, k, 2, 3);
    }
}

__global__ void wr_short_lane_rows(short *p, size_t rows)
{
    size_t r = (size_t) blockIdx.x*blockDim.x + threadIdx.x;
    for (;  r < rows;  r += (size_t) gridDim.x*blockDim.x)
    {
#pragma unroll
        for (int k = 0;  k < 32;  k++)
            *(volatile short *) (p + r*32 + k) = (short) (r + k);
    }
}

int main()
{
    void *buf;
    int *sink;
    CK(hipMalloc(&buf, BYTES));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, BYTES));
    const int grid = 256*8, block = 256;
    for (int rep = 0;  rep < 3;  rep++)
    {
        hipLaunchKernelGGL(rd_dword_coalesced, dim3(grid), dim3(block), 0, 0, (const int *) buf, sink, BYTES/4);
        hipLaunchKernelGGL(rd_x4_coalesced, dim3(grid), dim3(block), 0, 0, (const int4 *) buf, sink, BYTES/16);
        hipLaunchKernelGGL(rd_x4_lane_rows, dim3(grid), dim3(block), 0, 0, (const int4 *) buf, sink, BYTES/128);
        hipLaunchKernelGGL(rd_short_lane_rows, dim3(grid), dim3(block), 0, 0, (const short *) buf, sink, BYTES/64);
        hipLaunchKernelGGL(wr_dword_coalesced, dim3(grid), dim3(block), 0, 0, (int *) buf, BYTES/4);
        hipLaunchKernelGGL(wr_x4_lane_rows, dim3(grid), dim3(block), 0, 0, (int4 *) buf, BYTES/128);
        hipLaunchKernelGGL(wr_short_lane_rows, dim3(grid), dim3(block), 0, 0, (short *) buf, BYTES/64);
        CK(hipDeviceSynchronize());
    }
    printf("every kernel moved %zu bytes\n", BYTES);
    return 0;
}
