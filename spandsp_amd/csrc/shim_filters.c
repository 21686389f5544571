/*
 * shim_filters.c -- the spandsp-named filter instance entry points of src/spandsp/complex_filters.h:62-68
 * (reference: src/complex_filters.c:39-118).  The reference's "filter" is a state block handed to a step function the
 * CALLER supplies through fspec_t: there is no arithmetic of the library's own on this path, so this is host code only
 * (a thin shim).  A filter's state is np + 1 delay elements, a running sum and, for moving-average
 * kinds, a ring position -- all zero at creation.
 */
#include <stdlib.h>

#include "spangpu_spandsp.h"

/* bytes of one instance for a spec: the fixed part and the delay line behind it */
static size_t instance_bytes(const fspec_t *fs)
{
    return sizeof(filter_t) + sizeof(float)*((size_t) fs->np + 1);
}

static int usable(const fspec_t *fs)
{
    return fs != NULL  &&  fs->np >= 0;
}

filter_t *filter_create(fspec_t *fs)
{
    filter_t *fi;

    if (!usable(fs))
        return NULL;
    /* zeroed storage is the cleared state: sum 0.0f, ptr 0, every delay element 0.0f */
    fi = (filter_t *) calloc(1, instance_bytes(fs));
    if (fi)
        fi->fs = fs;
    return fi;
}

void filter_delete(filter_t *fi)
{
    free(fi);
}

float filter_step(filter_t *fi, float x)
{
    filter_step_func_t step = fi->fs->fsf;

    return step(fi, x);
}

/* A complex filter is the same real filter run on the two parts separately: two independent instances. */
cfilter_t *cfilter_create(fspec_t *fs)
{
    cfilter_t *pair;
    filter_t *re;
    filter_t *im;

    if (!usable(fs))
        return NULL;
    pair = (cfilter_t *) malloc(sizeof(*pair));
    re = filter_create(fs);
    im = filter_create(fs);
    if (pair == NULL  ||  re == NULL  ||  im == NULL)
    {
        free(pair);
        free(re);
        free(im);
        return NULL;
    }
    pair->ref = re;
    pair->imf = im;
    return pair;
}

void cfilter_delete(cfilter_t *cfi)
{
    if (cfi == NULL)
        return;
    filter_delete(cfi->imf);
    filter_delete(cfi->ref);
    free(cfi);
}

complexf_t cfilter_step(cfilter_t *cfi, const complexf_t *z)
{
    complexf_t y;

    /* real part first: the caller's step function may have side effects of its own */
    y.re = filter_step(cfi->ref, z->re);
    y.im = filter_step(cfi->imf, z->im);
    return y;
}
