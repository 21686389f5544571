#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../spandsp_amd/csrc/echo_pair.hpp"
using namespace spg;
__global__ void k(const int *a, const int *b, const int *c, int *o)
{
    const int i = threadIdx.x + blockIdx.x*64;
    int fifteen = 15; asm volatile("" : "+v"(fifteen));
    o[6*i + 0] = dot2_i16(a[i], b[i], c[i]);
    o[6*i + 1] = mad_i16<0>(a[i], b[i], c[i]);
    o[6*i + 2] = mad_i16<1>(a[i], b[i], c[i]);
    int p = b[i]; put_tap16<0>(p, c[i], fifteen); o[6*i + 3] = p;
    p = b[i]; put_tap16<1>(p, c[i], fifteen); o[6*i + 4] = p;
    o[6*i + 5] = __builtin_amdgcn_alignbit(a[i], b[i], 16);
}
int main()
{
    const int n = 64*64;
    int *h = (int *) malloc(n*4*9), *d;
    srand(5);
    for (int i = 0; i < 3*n; i++) h[i] = (rand() << 16) ^ rand() ^ (rand() << 31);
    for (int i = 0; i < 64; i++) { h[i] = (i & 1) ? 0x80008000 : 0x7fff7fff; h[n + i] = (i & 2) ? 0x80008000 : 0x7fff8000; }
    hipMalloc(&d, n*4*9); hipMemcpy(d, h, n*4*3, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, 0, d, d + n, d + 2*n, d + 3*n);
    hipMemcpy(h + 3*n, d + 3*n, n*4*6, hipMemcpyDeviceToHost);
    int bad[6] = {0};
    for (int i = 0; i < n; i++)
    {
        const int a = h[i], b = h[n + i], c = h[2*n + i]; const int *o = h + 3*n + 6*i;
        const short al = a & 0xffff, ah = a >> 16, bl = b & 0xffff, bh = b >> 16;
        int w[6];
        w[0] = (int) ((unsigned) (al*bl) + (unsigned) (ah*bh) + (unsigned) c);
        w[1] = (int) ((unsigned) (al*bl) + (unsigned) c);
        w[2] = (int) ((unsigned) (ah*bl) + (unsigned) c);
        w[3] = (b & 0xffff0000) | (((unsigned) c >> 15) & 0xffff);
        w[4] = (b & 0x0000ffff) | ((((unsigned) c >> 15) & 0xffff) << 16);
        w[5] = (int) (((unsigned) b >> 16) | ((unsigned) a << 16));
        for (int q = 0; q < 6; q++) if (w[q] != o[q]) { if (bad[q]++ < 2) printf("op %d: a %08x b %08x c %08x got %08x want %08x\n", q, a, b, c, o[q], w[q]); }
    }
    printf("bad: %d %d %d %d %d %d\n", bad[0], bad[1], bad[2], bad[3], bad[4], bad[5]);
    return 0;
}
