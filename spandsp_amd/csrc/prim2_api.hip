// prim2_api.hip -- the rest of SURVEY 8(a)'s primitives as batched entry points of their own (round 6):
//   periodogram(), periodogram_prepare(), periodogram_apply(), periodogram_freq_error()     src/tone_detect.c:208-250, :299-312
//   fixed_sqrt32()                                          src/math_fixed.c:158-169 (table: make_math_fixed_tables.c)
//   dds_lookup_complexf(), dds_complexf() (= lookup + dds_advancef())                       src/dds_float.c:2135-2187
//   arctan2()                                               src/spandsp/arctan2.h:47-80
// The last three run inside the receiver kernels (quad_round_front.inc, v29_dev.hpp ...: the AGC's root of the signal power,
// the carrier's phasor, the decision's angle) and are proven there through the receivers' state words; here each is one
// launch over N independent items, one lane per item, on the same tables (modem_tables.c) and, for arctan2, the same device
// function (v29_common.hpp) -- the direct evidence, over their whole domains (tests/test_prim2_gpu.py).  The periodograms
// are not used by any receiver of the path; SURVEY section 2 lists them with the Goertzel core (tone_detect.h:202-249).
// Built -ffp-contract=off like the rest: every product and sum rounded by itself, in the reference's order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>

#include "../../include/spangpu.h"
#include "modem_tables.h"
#include "v29_common.hpp"

extern "C" int spangpu_set_error(int code, const char *msg);

#define P2_TRY(x) do { if ((x) != hipSuccess) return spangpu_set_error(SPANGPU_ERR_HIP, #x " failed"); } while (0)

namespace {

// periodogram(), tone_detect.c:208-225: x += coeffs[i] "times" the folded pair, real and imaginary sums each in index order
__global__ void periodogram_kernel(const float2 *coeffs, long long cs, const float2 *amp, long long as, float2 *out, int items, int len)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if (i >= items)
        return;
    const float2 *c = coeffs + (size_t) i*cs;
    const float2 *a = amp + (size_t) i*as;
    float xre = 0.0f;
    float xim = 0.0f;
    for (int k = 0;  k < len/2;  k++)
    {
        const float2 p = a[k];
        const float2 q = a[len - 1 - k];
        const float sre = p.x + q.x, sim = p.y + q.y;
        const float dre = p.x - q.x, dim = p.y - q.y;
        xre += (c[k].x*sre - c[k].y*dim);
        xim += (c[k].x*sim + c[k].y*dre);
    }
    out[i] = make_float2(xre, xim);
}

// periodogram_prepare(), tone_detect.c:228-239
__global__ void periodogram_prepare_kernel(const float2 *amp, long long as, float2 *sum, float2 *diff, int items, int len)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if (i >= items)
        return;
    const float2 *a = amp + (size_t) i*as;
    float2 *s = sum + (size_t) i*(len/2);
    float2 *d = diff + (size_t) i*(len/2);
    for (int k = 0;  k < len/2;  k++)
    {
        const float2 p = a[k];
        const float2 q = a[len - 1 - k];
        s[k] = make_float2(p.x + q.x, p.y + q.y);
        d[k] = make_float2(p.x - q.x, p.y - q.y);
    }
}

// periodogram_apply(), tone_detect.c:242-255
__global__ void periodogram_apply_kernel(const float2 *coeffs, long long cs, const float2 *sum, const float2 *diff, float2 *out, int items, int len)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if (i >= items)
        return;
    const float2 *c = coeffs + (size_t) i*cs;
    const float2 *s = sum + (size_t) i*(len/2);
    const float2 *d = diff + (size_t) i*(len/2);
    float xre = 0.0f;
    float xim = 0.0f;
    for (int k = 0;  k < len/2;  k++)
    {
        xre += (c[k].x*s[k].x - c[k].y*d[k].y);
        xim += (c[k].x*s[k].y + c[k].y*d[k].x);
    }
    out[i] = make_float2(xre, xim);
}

// periodogram_freq_error(), tone_detect.c:299-310 (complex_mulf: re = a.re*b.re - a.im*b.im, im = a.re*b.im + a.im*b.re)
__global__ void periodogram_freq_error_kernel(float2 offset, float scale, const float2 *last, const float2 *now, float *out, int items)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if (i >= items)
        return;
    const float2 l = last[i];
    const float2 r = now[i];
    const float pre = l.x*offset.x - l.y*offset.y;
    const float pim = l.x*offset.y + l.y*offset.x;
    out[i] = scale*(r.y*pre - r.x*pim)/(r.x*r.x + r.y*r.y);
}

// fixed_sqrt32(), math_fixed.c:158-169, on the receivers' table (V29Tables::sqrt_tab, from spg_make_sqrt_table())
__global__ void fixed_sqrt32_kernel(const uint16_t *tab, const uint32_t *x, uint16_t *out, int items)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if (i >= items)
        return;
    uint32_t xx = x[i];
    uint16_t r = 0;
    if (xx != 0)
    {
        const int top = 31 - __builtin_clz(xx);
        const int shift = 30 - (top & ~1);
        xx <<= shift;
        r = (uint16_t) (tab[((xx >> 24) & 0xFF) - 64] >> (shift >> 1));
    }
    out[i] = r;
}

// dds_complexf(), dds_float.c:2179-2187, n times: the phasor of every step and the accumulator after the last one
// (n = 1, rate 0 is dds_lookup_complexf(); the accumulator moves as dds_advancef() moves it)
__global__ void dds_complexf_kernel(const float *sine, uint32_t *acc, const int32_t *rate, float2 *out, int items, int n)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if (i >= items)
        return;
    uint32_t a = acc[i];
    const uint32_t r = (uint32_t) rate[i];
    float2 *o = out + (size_t) i*n;
    for (int k = 0;  k < n;  k++)
    {
        o[k] = make_float2(sine[(uint32_t) (a + (1u << 30)) >> 21], sine[a >> 21]);
        a += r;
    }
    acc[i] = a;
}

__global__ void arctan2_kernel(const float *y, const float *x, int32_t *out, int items)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if (i >= items)
        return;
    out[i] = spg::v29_arctan2(y[i], x[i]);
}

template <typename T>
int to_device(const T *src, size_t count, int mem, T **dev, bool *owned)
{
    *owned = false;
    if (mem == SPANGPU_MEM_DEVICE)
    {
        *dev = (T *) src;
        return SPANGPU_OK;
    }
    P2_TRY(hipMalloc((void **) dev, count*sizeof(T) + 16));
    if (hipMemcpy(*dev, src, count*sizeof(T), hipMemcpyHostToDevice) != hipSuccess)
    {
        (void) hipFree(*dev);
        *dev = nullptr;
        return spangpu_set_error(SPANGPU_ERR_HIP, "hipMemcpy (host to device) failed");
    }
    *owned = true;
    return SPANGPU_OK;
}

struct Held
{
    void *p[8];
    int n = 0;
    ~Held()
    {
        for (int i = 0;  i < n;  i++)
            (void) hipFree(p[i]);
    }
    void keep(void *q, bool owned)
    {
        if (owned)
            p[n++] = q;
    }
};

int ready(int device, int items, int n, const void *a, const void *b, const void *c)
{
    if (items <= 0  ||  n <= 0  ||  a == nullptr  ||  b == nullptr  ||  c == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (spangpu_device_count() <= 0)
        return spangpu_set_error(SPANGPU_ERR_NO_DEVICE, "no HIP device: libspangpu has no CPU fallback");
    P2_TRY(hipSetDevice(device));
    return SPANGPU_OK;
}

int finish(int mem, void *host, const void *dev, size_t bytes)
{
    P2_TRY(hipGetLastError());
    if (mem != SPANGPU_MEM_DEVICE)
        P2_TRY(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
    else
        P2_TRY(hipDeviceSynchronize());
    return SPANGPU_OK;
}

// the receivers' sine and square root tables on a device, made once per device (host code: modem_tables.c)
constexpr int kMaxDevices = 16;
float *g_sine[kMaxDevices];
uint16_t *g_sqrt[kMaxDevices];
std::mutex g_tables_lock;

int tables(int device, const float **sine, const uint16_t **sq)
{
    if (device < 0  ||  device >= kMaxDevices)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "device out of range");
    std::lock_guard<std::mutex> guard(g_tables_lock);
    if (g_sine[device] == nullptr)
    {
        float s[SPG_SINE_LEN];
        uint16_t q[194];
        spg_make_sine_table(s);
        spg_make_sqrt_table(q);
        q[193] = 0;
        float *ds = nullptr;
        uint16_t *dq = nullptr;
        P2_TRY(hipMalloc((void **) &ds, sizeof(s)));
        P2_TRY(hipMalloc((void **) &dq, sizeof(q)));
        P2_TRY(hipMemcpy(ds, s, sizeof(s), hipMemcpyHostToDevice));
        P2_TRY(hipMemcpy(dq, q, sizeof(q), hipMemcpyHostToDevice));
        g_sine[device] = ds;
        g_sqrt[device] = dq;
    }
    *sine = g_sine[device];
    *sq = g_sqrt[device];
    return SPANGPU_OK;
}

}   // namespace

extern "C" {

// complex values as {re, im} float pairs (complexf_t); strides in complex elements, 0 = the same row for every item
int spangpu_periodogram_batch(int device, const float *coeffs, long long c_stride, const float *amp, long long a_stride, float *out,
                              int items, int len, int mem)
{
    int rc = ready(device, items, len, coeffs, amp, out);
    if (rc != SPANGPU_OK)
        return rc;
    if (c_stride < 0  ||  a_stride < 0  ||  (c_stride != 0  &&  c_stride < len/2)  ||  (a_stride != 0  &&  a_stride < len))
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "a stride is 0 (one row for all items) or at least a row long");
    Held h;
    float *dc, *da, *dout;
    bool o;
    if ((rc = to_device(coeffs, 2*(c_stride  ?  (size_t) c_stride*items  :  (size_t) (len/2) + 1), mem, &dc, &o)) < 0) return rc;
    h.keep(dc, o);
    if ((rc = to_device(amp, 2*(a_stride  ?  (size_t) a_stride*items  :  (size_t) len), mem, &da, &o)) < 0) return rc;
    h.keep(da, o);
    if ((rc = to_device((const float *) out, 2*(size_t) items, mem, &dout, &o)) < 0) return rc;
    h.keep(dout, o);
    hipLaunchKernelGGL(periodogram_kernel, dim3((items + 63)/64), dim3(64), 0, 0, (const float2 *) dc, c_stride, (const float2 *) da, a_stride,
                       (float2 *) dout, items, len);
    return finish(mem, out, dout, 2*(size_t) items*sizeof(float));
}

// sum and diff: [items][len/2] complex each
int spangpu_periodogram_prepare_batch(int device, const float *amp, long long a_stride, float *sum, float *diff, int items, int len, int mem)
{
    int rc = ready(device, items, len, amp, sum, diff);
    if (rc != SPANGPU_OK)
        return rc;
    if (len < 2  ||  a_stride < 0  ||  (a_stride != 0  &&  a_stride < len))
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad length or stride");
    Held h;
    float *da, *ds, *dd;
    bool o;
    const size_t half = (size_t) (len/2);
    if ((rc = to_device(amp, 2*(a_stride  ?  (size_t) a_stride*items  :  (size_t) len), mem, &da, &o)) < 0) return rc;
    h.keep(da, o);
    if ((rc = to_device((const float *) sum, 2*half*items, mem, &ds, &o)) < 0) return rc;
    h.keep(ds, o);
    if ((rc = to_device((const float *) diff, 2*half*items, mem, &dd, &o)) < 0) return rc;
    h.keep(dd, o);
    hipLaunchKernelGGL(periodogram_prepare_kernel, dim3((items + 63)/64), dim3(64), 0, 0, (const float2 *) da, a_stride, (float2 *) ds, (float2 *) dd, items, len);
    P2_TRY(hipGetLastError());
    if (mem != SPANGPU_MEM_DEVICE)
    {
        P2_TRY(hipMemcpy(sum, ds, 2*half*items*sizeof(float), hipMemcpyDeviceToHost));
        P2_TRY(hipMemcpy(diff, dd, 2*half*items*sizeof(float), hipMemcpyDeviceToHost));
    }
    else
    {
        P2_TRY(hipDeviceSynchronize());
    }
    return (int) half;
}

int spangpu_periodogram_apply_batch(int device, const float *coeffs, long long c_stride, const float *sum, const float *diff, float *out,
                                    int items, int len, int mem)
{
    int rc = ready(device, items, len, coeffs, sum, diff);
    if (rc != SPANGPU_OK)
        return rc;
    if (out == nullptr  ||  len < 2  ||  c_stride < 0  ||  (c_stride != 0  &&  c_stride < len/2))
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    Held h;
    float *dc, *ds, *dd, *dout;
    bool o;
    const size_t half = (size_t) (len/2);
    if ((rc = to_device(coeffs, 2*(c_stride  ?  (size_t) c_stride*items  :  half), mem, &dc, &o)) < 0) return rc;
    h.keep(dc, o);
    if ((rc = to_device(sum, 2*half*items, mem, &ds, &o)) < 0) return rc;
    h.keep(ds, o);
    if ((rc = to_device(diff, 2*half*items, mem, &dd, &o)) < 0) return rc;
    h.keep(dd, o);
    if ((rc = to_device((const float *) out, 2*(size_t) items, mem, &dout, &o)) < 0) return rc;
    h.keep(dout, o);
    hipLaunchKernelGGL(periodogram_apply_kernel, dim3((items + 63)/64), dim3(64), 0, 0, (const float2 *) dc, c_stride, (const float2 *) ds,
                       (const float2 *) dd, (float2 *) dout, items, len);
    return finish(mem, out, dout, 2*(size_t) items*sizeof(float));
}

// phase_offset: the {re, im} periodogram_generate_phase_offset() made (host memory); last / result: [items] complex
int spangpu_periodogram_freq_error_batch(int device, const float *phase_offset, float scale, const float *last_result, const float *result,
                                         float *out, int items, int mem)
{
    int rc = ready(device, items, 1, phase_offset, last_result, result);
    if (rc != SPANGPU_OK)
        return rc;
    if (out == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "null output");
    Held h;
    float *dl, *dr, *dout;
    bool o;
    if ((rc = to_device(last_result, 2*(size_t) items, mem, &dl, &o)) < 0) return rc;
    h.keep(dl, o);
    if ((rc = to_device(result, 2*(size_t) items, mem, &dr, &o)) < 0) return rc;
    h.keep(dr, o);
    if ((rc = to_device((const float *) out, (size_t) items, mem, &dout, &o)) < 0) return rc;
    h.keep(dout, o);
    hipLaunchKernelGGL(periodogram_freq_error_kernel, dim3((items + 63)/64), dim3(64), 0, 0, make_float2(phase_offset[0], phase_offset[1]), scale,
                       (const float2 *) dl, (const float2 *) dr, dout, items);
    return finish(mem, out, dout, (size_t) items*sizeof(float));
}

int spangpu_fixed_sqrt32_batch(int device, const uint32_t *x, uint16_t *out, int items, int mem)
{
    int rc = ready(device, items, 1, x, out, out);
    if (rc != SPANGPU_OK)
        return rc;
    const float *sine;
    const uint16_t *sq;
    if ((rc = tables(device, &sine, &sq)) < 0)
        return rc;
    Held h;
    uint32_t *dx;
    uint16_t *dout;
    bool o;
    if ((rc = to_device(x, (size_t) items, mem, &dx, &o)) < 0) return rc;
    h.keep(dx, o);
    if ((rc = to_device((const uint16_t *) out, (size_t) items, mem, &dout, &o)) < 0) return rc;
    h.keep(dout, o);
    hipLaunchKernelGGL(fixed_sqrt32_kernel, dim3((items + 255)/256), dim3(256), 0, 0, sq, (const uint32_t *) dx, dout, items);
    return finish(mem, out, dout, (size_t) items*sizeof(uint16_t));
}

// out: [items][n] complex; phase_acc[items] is advanced n times by phase_rate[items]
int spangpu_dds_complexf_batch(int device, uint32_t *phase_acc, const int32_t *phase_rate, float *out, int items, int n, int mem)
{
    int rc = ready(device, items, n, phase_acc, phase_rate, out);
    if (rc != SPANGPU_OK)
        return rc;
    const float *sine;
    const uint16_t *sq;
    if ((rc = tables(device, &sine, &sq)) < 0)
        return rc;
    Held h;
    uint32_t *da;
    int32_t *dr;
    float *dout;
    bool o;
    if ((rc = to_device((const uint32_t *) phase_acc, (size_t) items, mem, &da, &o)) < 0) return rc;
    h.keep(da, o);
    if ((rc = to_device(phase_rate, (size_t) items, mem, &dr, &o)) < 0) return rc;
    h.keep(dr, o);
    if ((rc = to_device((const float *) out, 2*(size_t) items*n, mem, &dout, &o)) < 0) return rc;
    h.keep(dout, o);
    hipLaunchKernelGGL(dds_complexf_kernel, dim3((items + 63)/64), dim3(64), 0, 0, sine, da, (const int32_t *) dr, (float2 *) dout, items, n);
    P2_TRY(hipGetLastError());
    if (mem != SPANGPU_MEM_DEVICE)
        P2_TRY(hipMemcpy(phase_acc, da, (size_t) items*sizeof(uint32_t), hipMemcpyDeviceToHost));
    return finish(mem, out, dout, 2*(size_t) items*n*sizeof(float));
}

int spangpu_arctan2_batch(int device, const float *y, const float *x, int32_t *out, int items, int mem)
{
    int rc = ready(device, items, 1, y, x, out);
    if (rc != SPANGPU_OK)
        return rc;
    Held h;
    float *dy, *dx;
    int32_t *dout;
    bool o;
    if ((rc = to_device(y, (size_t) items, mem, &dy, &o)) < 0) return rc;
    h.keep(dy, o);
    if ((rc = to_device(x, (size_t) items, mem, &dx, &o)) < 0) return rc;
    h.keep(dx, o);
    if ((rc = to_device((const int32_t *) out, (size_t) items, mem, &dout, &o)) < 0) return rc;
    h.keep(dout, o);
    hipLaunchKernelGGL(arctan2_kernel, dim3((items + 255)/256), dim3(256), 0, 0, (const float *) dy, (const float *) dx, dout, items);
    return finish(mem, out, dout, (size_t) items*sizeof(int32_t));
}

}   // extern "C"
