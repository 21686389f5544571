// prim_api.hip -- the inner primitives of the modem receivers as batched entry points of their own (include/spangpu.h:
// spangpu_vec_* / spangpu_cvec_* / spangpu_power_meter_*), SURVEY 8(a) rows a11, a12, a19.
//
// Inside the receiver kernels these run fused (quad_round_front.inc, v29_quad.hpp ...) and are proven through the receivers'
// bit-exact float state.  Here each is one launch over N independent items, one lane per item, with the reference's scalar
// order of operations (this file, like the rest of the library, is built -ffp-contract=off: every product and sum rounded by
// itself): the direct evidence that the arithmetic is the reference's, on inputs a receiver never produces -- denormals,
// cancellation, infinities.
//   vec_circular_dot_prodf   src/vector_float.c:890-900, 932-939      z = dot(x[pos:], y[:n-pos]) + dot(x[:pos], y[n-pos:])
//   vec_circular_lmsf        src/vector_float.c:982-1000 (leak 0.9999f, :942)
//   cvec_circular_dot_prodf  src/complex_vector_float.c:137-150, 187-196 (non-conjugating; the two parts added at the end)
//   cvec_circular_lmsf       src/complex_vector_float.c:201-219
//   power_meter_update       src/power_meter.c:65-70 (over a row of samples; the reading after the last one)
// Rows are item-major ([item][n]); x and y of an item may be the same for all items (stride 0).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/spangpu.h"

extern "C" int spangpu_set_error(int code, const char *msg);

#define PR_TRY(x) do { if ((x) != hipSuccess) return spangpu_set_error(SPANGPU_ERR_HIP, #x " failed"); } while (0)

namespace {

__global__ void vec_circ_dot_kernel(const float *x, long long xs, const float *y, long long ys, const int32_t *pos, float *z, int items, int n)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if (i >= items)
        return;
    const float *xi = x + (size_t) i*xs;
    const float *yi = y + (size_t) i*ys;
    const int p = pos[i];
    if (p < 0  ||  p > n)
    {
        z[i] = __uint_as_float(0x7FC00000u);        // (a position outside the row: nothing is read; the result says so)
        return;
    }
    float a = 0.0f;
    for (int k = 0;  k < n - p;  k++)
        a += xi[p + k]*yi[k];
    float b = 0.0f;
    for (int k = 0;  k < p;  k++)
        b += xi[k]*yi[n - p + k];
    z[i] = a + b;
}

__global__ void vec_circ_lms_kernel(const float *x, long long xs, float *y, long long ys, const int32_t *pos, const float *err, int items, int n)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if (i >= items)
        return;
    const float *xi = x + (size_t) i*xs;
    float *yi = y + (size_t) i*ys;
    const int p = pos[i];
    if (p < 0  ||  p > n)
        return;                 // (a position outside the row: nothing is read or written for this item)
    const float e = err[i];
    for (int k = 0;  k < n - p;  k++)
        yi[k] = yi[k]*0.9999f + xi[p + k]*e;
    for (int k = 0;  k < p;  k++)
        yi[n - p + k] = yi[n - p + k]*0.9999f + xi[k]*e;
}

__global__ void cvec_circ_dot_kernel(const float2 *x, long long xs, const float2 *y, long long ys, const int32_t *pos, float2 *z, int items, int n)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if (i >= items)
        return;
    const float2 *xi = x + (size_t) i*xs;
    const float2 *yi = y + (size_t) i*ys;
    const int p = pos[i];
    if (p < 0  ||  p > n)
    {
        z[i] = make_float2(__uint_as_float(0x7FC00000u), __uint_as_float(0x7FC00000u));     // (nothing is read; the result says so)
        return;
    }
    float are = 0.0f;
    float aim = 0.0f;
    for (int k = 0;  k < n - p;  k++)
    {
        are += (xi[p + k].x*yi[k].x - xi[p + k].y*yi[k].y);
        aim += (xi[p + k].x*yi[k].y + xi[p + k].y*yi[k].x);
    }
    float bre = 0.0f;
    float bim = 0.0f;
    for (int k = 0;  k < p;  k++)
    {
        bre += (xi[k].x*yi[n - p + k].x - xi[k].y*yi[n - p + k].y);
        bim += (xi[k].x*yi[n - p + k].y + xi[k].y*yi[n - p + k].x);
    }
    z[i] = make_float2(are + bre, aim + bim);
}

__global__ void cvec_circ_lms_kernel(const float2 *x, long long xs, float2 *y, long long ys, const int32_t *pos, const float2 *err, int items, int n)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if (i >= items)
        return;
    const float2 *xi = x + (size_t) i*xs;
    float2 *yi = y + (size_t) i*ys;
    const int p = pos[i];
    if (p < 0  ||  p > n)
        return;                 // (a position outside the row: nothing is read or written for this item)
    const float2 e = err[i];
    for (int k = 0;  k < n;  k++)
    {
        const float2 xv = (k < n - p)  ?  xi[p + k]  :  xi[k - (n - p)];
        float2 yv = yi[k];
        yv.x = yv.x*0.9999f + (xv.y*e.y + xv.x*e.x);
        yv.y = yv.y*0.9999f + (xv.x*e.y - xv.y*e.x);
        yi[k] = yv;
    }
}

__global__ void power_meter_kernel(const int16_t *amp, long long stride, int32_t *reading, const int32_t *shift, int items, int n)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if (i >= items)
        return;
    const int16_t *a = amp + (size_t) i*stride;
    int32_t r = reading[i];
    const int sh = shift[i] & 31;            // (what a 32-bit arithmetic shift does with its count on the reference's x86-64, too)
    for (int k = 0;  k < n;  k++)
        r += ((a[k]*a[k] - r) >> sh);
    reading[i] = r;
}

// godard_ted_rx(), godard.c:144-162: the two band edge filters over a row of samples (state and descriptor words: spangpu.h)
__global__ void godard_rx_kernel(uint32_t *state, const uint32_t *desc, long long ds, const float *samples, long long stride, int items, int n)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if (i >= items)
        return;
    uint32_t *st = state + (size_t) i*8;
    const uint32_t *d = desc + (size_t) i*ds;
    const float lc0 = __uint_as_float(d[0]), lc1 = __uint_as_float(d[1]);
    const float hc0 = __uint_as_float(d[3]), hc1 = __uint_as_float(d[4]);
    float l0 = __uint_as_float(st[0]), l1 = __uint_as_float(st[1]);
    float h0 = __uint_as_float(st[2]), h1 = __uint_as_float(st[3]);
    const float *x = samples + (size_t) i*stride;
    for (int k = 0;  k < n;  k++)
    {
        const float sample = x[k];
        float v = l0*lc0 + l1*lc1 + sample;
        l1 = l0;
        l0 = v;
        v = h0*hc0 + h1*hc1 + sample;
        h1 = h0;
        h0 = v;
    }
    st[0] = __float_as_uint(l0);
    st[1] = __float_as_uint(l1);
    st[2] = __float_as_uint(h0);
    st[3] = __float_as_uint(h1);
}

// godard_ted_per_baud(), godard.c:165-220
__global__ void godard_baud_kernel(uint32_t *state, const uint32_t *desc, long long ds, int32_t *correction, int items)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if (i >= items)
        return;
    uint32_t *st = state + (size_t) i*8;
    const uint32_t *d = desc + (size_t) i*ds;
    const float l0 = __uint_as_float(st[0]), l1 = __uint_as_float(st[1]);
    const float h0 = __uint_as_float(st[2]), h1 = __uint_as_float(st[3]);
    float v = l1*h0*__uint_as_float(d[2]) - l0*h1*__uint_as_float(d[5]) + l1*h1*__uint_as_float(d[6]);
    const float p = v - __uint_as_float(st[5]);
    st[5] = st[4];
    st[4] = __float_as_uint(v);
    const float phase = __uint_as_float(st[6]) - p;
    st[6] = __float_as_uint(phase);
    v = fabsf(phase);
    int corr = 0;
    if (v > __uint_as_float(d[8]))                          // fine_trigger
    {
        int step = (v > __uint_as_float(d[7]))  ?  (int32_t) d[9]  :  (int32_t) d[10];
        if (phase < 0.0f)
            step = -step;
        corr = step;
        st[7] = (uint32_t) ((int32_t) st[7] + step);
    }
    correction[i] = corr;
}

// bring a host array to the device (or pass a device pointer through); *owned says whether to free it
template <typename T>
int stage_in(const T *src, size_t count, int mem, T **dev, bool *owned)
{
    *owned = false;
    if (mem == SPANGPU_MEM_DEVICE)
    {
        *dev = (T *) src;
        return SPANGPU_OK;
    }
    PR_TRY(hipMalloc((void **) dev, count*sizeof(T) + 16));
    if (hipMemcpy(*dev, src, count*sizeof(T), hipMemcpyHostToDevice) != hipSuccess)
    {
        (void) hipFree(*dev);           // (the caller's Staged has not been told of it yet)
        *dev = nullptr;
        return spangpu_set_error(SPANGPU_ERR_HIP, "hipMemcpy (host to device) failed");
    }
    *owned = true;
    return SPANGPU_OK;
}

struct Staged
{
    void *p[6];
    int n = 0;
    ~Staged()
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

int check(int device, int items, int n, const void *a, const void *b, const void *c, const void *d)
{
    if (items <= 0  ||  n <= 0  ||  a == nullptr  ||  b == nullptr  ||  c == nullptr  ||  d == nullptr)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "bad arguments");
    if (spangpu_device_count() <= 0)
        return spangpu_set_error(SPANGPU_ERR_NO_DEVICE, "no HIP device: libspangpu has no CPU fallback");
    PR_TRY(hipSetDevice(device));
    return SPANGPU_OK;
}

// positions in host memory are looked at before anything is staged
int check_pos(const int32_t *pos, int items, int n, int mem)
{
    if (mem != SPANGPU_MEM_HOST)
        return SPANGPU_OK;
    for (int i = 0;  i < items;  i++)
    {
        if (pos[i] < 0  ||  pos[i] > n)
            return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "a position outside its row (0 <= pos <= n)");
    }
    return SPANGPU_OK;
}

}   // namespace

extern "C" {

int spangpu_vec_circular_dot_prodf_batch(int device, const float *x, long long x_stride, const float *y, long long y_stride, const int32_t *pos,
                                         float *z, int items, int n, int mem)
{
    int rc = check(device, items, n, x, y, pos, z);
    if (rc != SPANGPU_OK)
        return rc;
    if ((rc = check_pos(pos, items, n, mem)) != SPANGPU_OK)
        return rc;
    Staged st;
    float *dx, *dy, *dz;
    int32_t *dp;
    bool o;
    if ((rc = stage_in(x, (size_t) (x_stride  ?  (size_t) x_stride*items  :  (size_t) n), mem, &dx, &o)) < 0) return rc;
    st.keep(dx, o);
    if ((rc = stage_in(y, (size_t) (y_stride  ?  (size_t) y_stride*items  :  (size_t) n), mem, &dy, &o)) < 0) return rc;
    st.keep(dy, o);
    if ((rc = stage_in(pos, (size_t) items, mem, &dp, &o)) < 0) return rc;
    st.keep(dp, o);
    if ((rc = stage_in(z, (size_t) items, mem, &dz, &o)) < 0) return rc;
    st.keep(dz, o);
    hipLaunchKernelGGL(vec_circ_dot_kernel, dim3((items + 63)/64), dim3(64), 0, 0, dx, x_stride, dy, y_stride, dp, dz, items, n);
    PR_TRY(hipGetLastError());
    if (mem != SPANGPU_MEM_DEVICE)
        PR_TRY(hipMemcpy(z, dz, (size_t) items*sizeof(float), hipMemcpyDeviceToHost));
    else
        PR_TRY(hipDeviceSynchronize());
    return SPANGPU_OK;
}

int spangpu_vec_circular_lmsf_batch(int device, const float *x, long long x_stride, float *y, long long y_stride, const int32_t *pos,
                                    const float *error, int items, int n, int mem)
{
    int rc = check(device, items, n, x, y, pos, error);
    if (rc != SPANGPU_OK)
        return rc;
    if ((rc = check_pos(pos, items, n, mem)) != SPANGPU_OK)
        return rc;
    if (y_stride < n)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "the rows of y are written: they cannot overlap");
    Staged st;
    float *dx, *dy, *de;
    int32_t *dp;
    bool o;
    if ((rc = stage_in(x, (size_t) (x_stride  ?  (size_t) x_stride*items  :  (size_t) n), mem, &dx, &o)) < 0) return rc;
    st.keep(dx, o);
    if ((rc = stage_in((const float *) y, (size_t) y_stride*items, mem, &dy, &o)) < 0) return rc;
    st.keep(dy, o);
    if ((rc = stage_in(pos, (size_t) items, mem, &dp, &o)) < 0) return rc;
    st.keep(dp, o);
    if ((rc = stage_in(error, (size_t) items, mem, &de, &o)) < 0) return rc;
    st.keep(de, o);
    hipLaunchKernelGGL(vec_circ_lms_kernel, dim3((items + 63)/64), dim3(64), 0, 0, dx, x_stride, dy, y_stride, dp, de, items, n);
    PR_TRY(hipGetLastError());
    if (mem != SPANGPU_MEM_DEVICE)
        PR_TRY(hipMemcpy(y, dy, (size_t) y_stride*items*sizeof(float), hipMemcpyDeviceToHost));
    else
        PR_TRY(hipDeviceSynchronize());
    return SPANGPU_OK;
}

// complex values as {re, im} float pairs (complexf_t); strides in complex elements
int spangpu_cvec_circular_dot_prodf_batch(int device, const float *x, long long x_stride, const float *y, long long y_stride, const int32_t *pos,
                                          float *z, int items, int n, int mem)
{
    int rc = check(device, items, n, x, y, pos, z);
    if (rc != SPANGPU_OK)
        return rc;
    if ((rc = check_pos(pos, items, n, mem)) != SPANGPU_OK)
        return rc;
    Staged st;
    float *dx, *dy, *dz;
    int32_t *dp;
    bool o;
    if ((rc = stage_in(x, 2*(size_t) (x_stride  ?  (size_t) x_stride*items  :  (size_t) n), mem, &dx, &o)) < 0) return rc;
    st.keep(dx, o);
    if ((rc = stage_in(y, 2*(size_t) (y_stride  ?  (size_t) y_stride*items  :  (size_t) n), mem, &dy, &o)) < 0) return rc;
    st.keep(dy, o);
    if ((rc = stage_in(pos, (size_t) items, mem, &dp, &o)) < 0) return rc;
    st.keep(dp, o);
    if ((rc = stage_in(z, 2*(size_t) items, mem, &dz, &o)) < 0) return rc;
    st.keep(dz, o);
    hipLaunchKernelGGL(cvec_circ_dot_kernel, dim3((items + 63)/64), dim3(64), 0, 0, (const float2 *) dx, x_stride, (const float2 *) dy, y_stride, dp,
                       (float2 *) dz, items, n);
    PR_TRY(hipGetLastError());
    if (mem != SPANGPU_MEM_DEVICE)
        PR_TRY(hipMemcpy(z, dz, 2*(size_t) items*sizeof(float), hipMemcpyDeviceToHost));
    else
        PR_TRY(hipDeviceSynchronize());
    return SPANGPU_OK;
}

int spangpu_cvec_circular_lmsf_batch(int device, const float *x, long long x_stride, float *y, long long y_stride, const int32_t *pos,
                                     const float *error, int items, int n, int mem)
{
    int rc = check(device, items, n, x, y, pos, error);
    if (rc != SPANGPU_OK)
        return rc;
    if ((rc = check_pos(pos, items, n, mem)) != SPANGPU_OK)
        return rc;
    if (y_stride < n)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "the rows of y are written: they cannot overlap");
    Staged st;
    float *dx, *dy, *de;
    int32_t *dp;
    bool o;
    if ((rc = stage_in(x, 2*(size_t) (x_stride  ?  (size_t) x_stride*items  :  (size_t) n), mem, &dx, &o)) < 0) return rc;
    st.keep(dx, o);
    if ((rc = stage_in((const float *) y, 2*(size_t) y_stride*items, mem, &dy, &o)) < 0) return rc;
    st.keep(dy, o);
    if ((rc = stage_in(pos, (size_t) items, mem, &dp, &o)) < 0) return rc;
    st.keep(dp, o);
    if ((rc = stage_in(error, 2*(size_t) items, mem, &de, &o)) < 0) return rc;
    st.keep(de, o);
    hipLaunchKernelGGL(cvec_circ_lms_kernel, dim3((items + 63)/64), dim3(64), 0, 0, (const float2 *) dx, x_stride, (float2 *) dy, y_stride, dp,
                       (const float2 *) de, items, n);
    PR_TRY(hipGetLastError());
    if (mem != SPANGPU_MEM_DEVICE)
        PR_TRY(hipMemcpy(y, dy, 2*(size_t) y_stride*items*sizeof(float), hipMemcpyDeviceToHost));
    else
        PR_TRY(hipDeviceSynchronize());
    return SPANGPU_OK;
}

// reading[i] after power_meter_update() over amp[i*stride .. + n) with shift[i], starting from reading[i]
int spangpu_power_meter_update_batch(int device, const int16_t *amp, long long stride, int32_t *reading, const int32_t *shift, int items, int n, int mem)
{
    int rc = check(device, items, n, amp, reading, shift, shift);
    if (rc != SPANGPU_OK)
        return rc;
    Staged st;
    int16_t *da;
    int32_t *dr, *ds;
    bool o;
    if ((rc = stage_in(amp, (size_t) stride*items, mem, &da, &o)) < 0) return rc;
    st.keep(da, o);
    if ((rc = stage_in((const int32_t *) reading, (size_t) items, mem, &dr, &o)) < 0) return rc;
    st.keep(dr, o);
    if ((rc = stage_in(shift, (size_t) items, mem, &ds, &o)) < 0) return rc;
    st.keep(ds, o);
    hipLaunchKernelGGL(power_meter_kernel, dim3((items + 63)/64), dim3(64), 0, 0, (const int16_t *) da, stride, dr, (const int32_t *) ds, items, n);
    PR_TRY(hipGetLastError());
    if (mem != SPANGPU_MEM_DEVICE)
        PR_TRY(hipMemcpy(reading, dr, (size_t) items*sizeof(int32_t), hipMemcpyDeviceToHost));
    else
        PR_TRY(hipDeviceSynchronize());
    return SPANGPU_OK;
}

// items Godard timing error detectors, each over its row of n samples (godard_ted_rx() n times)
int spangpu_godard_ted_rx_batch(int device, uint32_t *state, const uint32_t *desc, long long desc_stride, const float *samples, long long stride,
                                int items, int n, int mem)
{
    int rc = check(device, items, n, state, desc, samples, samples);
    if (rc != SPANGPU_OK)
        return rc;
    if ((desc_stride != 0  &&  desc_stride < 12)  ||  stride < 0  ||  (items > 1  &&  stride != 0  &&  stride < n))
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "a descriptor stride is 0 or at least 12 words, a sample stride 0 or at least a row");
    Staged st;
    uint32_t *dst, *dd;
    float *dx;
    bool o;
    if ((rc = stage_in((const uint32_t *) state, (size_t) items*8, mem, &dst, &o)) < 0) return rc;
    st.keep(dst, o);
    if ((rc = stage_in(desc, (desc_stride  ?  (size_t) desc_stride*items  :  (size_t) 12), mem, &dd, &o)) < 0) return rc;
    st.keep(dd, o);
    if ((rc = stage_in(samples, (size_t) stride*(items - 1) + n, mem, &dx, &o)) < 0) return rc;
    st.keep(dx, o);
    hipLaunchKernelGGL(godard_rx_kernel, dim3((items + 63)/64), dim3(64), 0, 0, dst, (const uint32_t *) dd, desc_stride, (const float *) dx, stride, items, n);
    PR_TRY(hipGetLastError());
    if (mem != SPANGPU_MEM_DEVICE)
        PR_TRY(hipMemcpy(state, dst, (size_t) items*8*sizeof(uint32_t), hipMemcpyDeviceToHost));
    else
        PR_TRY(hipDeviceSynchronize());
    return SPANGPU_OK;
}

// godard_ted_per_baud() of every item: correction[i] = what the call returns (the step to add to eq_put_step)
int spangpu_godard_ted_per_baud_batch(int device, uint32_t *state, const uint32_t *desc, long long desc_stride, int32_t *correction, int items, int mem)
{
    int rc = check(device, items, 1, state, desc, correction, correction);
    if (rc != SPANGPU_OK)
        return rc;
    if (desc_stride != 0  &&  desc_stride < 12)
        return spangpu_set_error(SPANGPU_ERR_BAD_ARG, "a descriptor stride is 0 (one for all items) or at least 12 words");
    Staged st;
    uint32_t *dst, *dd;
    int32_t *dc;
    bool o;
    if ((rc = stage_in((const uint32_t *) state, (size_t) items*8, mem, &dst, &o)) < 0) return rc;
    st.keep(dst, o);
    if ((rc = stage_in(desc, (desc_stride  ?  (size_t) desc_stride*items  :  (size_t) 12), mem, &dd, &o)) < 0) return rc;
    st.keep(dd, o);
    if ((rc = stage_in((const int32_t *) correction, (size_t) items, mem, &dc, &o)) < 0) return rc;
    st.keep(dc, o);
    hipLaunchKernelGGL(godard_baud_kernel, dim3((items + 63)/64), dim3(64), 0, 0, dst, (const uint32_t *) dd, desc_stride, dc, items);
    PR_TRY(hipGetLastError());
    if (mem != SPANGPU_MEM_DEVICE)
    {
        PR_TRY(hipMemcpy(state, dst, (size_t) items*8*sizeof(uint32_t), hipMemcpyDeviceToHost));
        PR_TRY(hipMemcpy(correction, dc, (size_t) items*sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    else
    {
        PR_TRY(hipDeviceSynchronize());
    }
    return SPANGPU_OK;
}

}   // extern "C"
