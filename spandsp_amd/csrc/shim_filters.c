/*
 * shim_filters.c -- the spandsp-named filter instance entry points of src/spandsp/complex_filters.h:62-68
 * (reference: src/complex_filters.c:39-118).  The reference's "filter" is a state block handed to a step function the
 * CALLER supplies through fspec_t: there is no arithmetic of the library's own on this path, so this is host code only
 * (a thin shim).  A filter's state is np + 1 delay elements, a running sum and, for moving-average
 * kinds, a ring position -- all cleared at creation.
 */
#include <stdlib.h>

#include "spangpu_spandsp.h"

filter_t *filter_create(fspec_t *fs)
{
    filter_t *fi;
    int k;

    if (fs == NULL  ||  fs->np < 0)
        return NULL;
    if ((fi = (filter_t *) malloc(sizeof(*fi) + sizeof(float)*((size_t) fs->np + 1))) == NULL)
        return NULL;
    fi->fs = fs;
    fi->sum = 0.0f;
    fi->ptr = 0;
    for (k = 0;  k <= fs->np;  k++)
        fi->v[k] = 0.0f;
    return fi;
}

void filter_delete(filter_t *fi)
{
    free(fi);
}

float filter_step(filter_t *fi, float x)
{
    return fi->fs->fsf(fi, x);
}

/* A complex filter is the same real filter run on the two parts separately. */
cfilter_t *cfilter_create(fspec_t *fs)
{
    cfilter_t *cfi;

    if ((cfi = (cfilter_t *) malloc(sizeof(*cfi))) == NULL)
        return NULL;
    cfi->ref = filter_create(fs);
    cfi->imf = (cfi->ref)  ?  filter_create(fs)  :  NULL;
    if (cfi->imf == NULL)
    {
        filter_delete(cfi->ref);
        free(cfi);
        return NULL;
    }
    return cfi;
}

void cfilter_delete(cfilter_t *cfi)
{
    if (cfi == NULL)
        return;
    filter_delete(cfi->ref);
    filter_delete(cfi->imf);
    free(cfi);
}

complexf_t cfilter_step(cfilter_t *cfi, const complexf_t *z)
{
    complexf_t out;

    out.re = filter_step(cfi->ref, z->re);
    out.im = filter_step(cfi->imf, z->im);
    return out;
}
