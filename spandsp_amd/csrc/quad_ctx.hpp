// quad_ctx.hpp -- the lane group ("quad") abstraction of the modem receiver kernels that give a channel FOUR LANES
// (v29_quad.hpp, ...), and the macros that let the same kernel body compile for gfx950 and -- for the tests -- for the
// host, where the four lanes of a channel run as four cooperatively scheduled fibers (tests/emul/quad_emul.cpp).
//
// On the device a channel's lanes are a DPP quad: lanes 4c .. 4c+3 of a wavefront.  They execute in lock step, so the
// exchange primitives are register moves:
//   bcast<S>(v)   the value lane S of the quad holds          (v_mov_b32 quad_perm:[S,S,S,S], usually folded into its user)
//   swap2(v)      the value of the lane two further on         (quad_perm:[2,3,0,1])
//   swap1(v)      the value of the neighbouring lane            (quad_perm:[1,0,3,2])
//   prev1(v)      the value of the lane before                  (quad_perm:[0,0,1,2])
//   any(b)        true if b holds on any lane of the WAVE (a uniform branch around rare work; loop control)
//   sync()        nothing: LDS operations of one wave are performed in program order
// Every exchange must sit in control flow that is uniform over the quad (the scalars of a channel are replicated in its
// four lanes, so conditions on them are); a DPP read from a lane that is switched off returns rubbish.
//
// On the host the body runs once per lane; bcast / swap2 / any / sync hand over to the next lane's fiber, and a
// generation is complete when all four lanes have arrived at the same call site (checked: a lane that takes another path
// than its quad is reported, which is exactly the error a DPP exchange under divergent control flow would be on the GPU).
// Between two hand-overs a lane runs alone, in an order the test chooses, so an LDS word written by one lane and read by
// another needs a sync() between the two in program order -- on the device the lock step provides that for free.
#pragma once

#include <stdint.h>

namespace spg {
// a compile-time integer as a function argument: a generic lambda takes QuadTag<k>{} where a function template would take <k>
template <int K> struct QuadTag
{
    static constexpr int value = K;
};
}   // namespace spg

#if defined(SPG_HOST_EMUL)

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SPG_FN              static inline
#define SPG_FN_NOINLINE     static inline
#define SPG_UNROLL          _Pragma("unroll")
#define SPG_SCHED_FENCE()   do { } while (0)
#define SPG_LOADS_DONE()    do { } while (0)

namespace spg {

struct float2
{
    float x;
    float y;
};
static inline float2 make_float2(float x, float y) { float2 r; r.x = x; r.y = y; return r; }
struct int4
{
    int x, y, z, w;
};
static inline uint32_t __float_as_uint(float v) { uint32_t u; memcpy(&u, &v, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float v; memcpy(&v, &u, 4); return v; }
static inline int min(int a, int b) { return (a < b)  ?  a  :  b; }
static inline int max(int a, int b) { return (a > b)  ?  a  :  b; }

// implemented by the emulator: hand over to the next lane of the quad; `site` identifies the call site
struct QuadHostState;
void quad_host_yield(QuadHostState *st, int lane, int site);

struct QuadHostState
{
    uint32_t slot[2][4];
    int gen[4];
    void *impl;
};

struct QuadHost
{
    QuadHostState *st;
    int lane;
    int role() const { return lane; }
    uint32_t exchange(uint32_t v, int src, int site)
    {
        const int g = st->gen[lane] & 1;
        st->slot[g][lane] = v;
        st->gen[lane]++;
        quad_host_yield(st, lane, site);
        return st->slot[g][src];
    }
    template <int S> float bcast(float v, int site = 0) { return __uint_as_float(exchange(__float_as_uint(v), S, 1000 + site)); }
    template <int S> int bcast(int v, int site = 0) { return (int) exchange((uint32_t) v, S, 2000 + site); }
    float swap2(float v, int site = 0) { return __uint_as_float(exchange(__float_as_uint(v), lane ^ 2, 3000 + site)); }
    int prev1(int v, int site = 0) { return (int) exchange((uint32_t) v, (lane > 0)  ?  (lane - 1)  :  0, 6000 + site); }
    uint32_t swap1(uint32_t v, int site = 0) { return exchange(v, lane ^ 1, 7000 + site); }
    float swap1f(float v, int site = 0) { return __uint_as_float(exchange(__float_as_uint(v), lane ^ 1, 7500 + site)); }
    uint32_t swap2(uint32_t v, int site = 0) { return exchange(v, lane ^ 2, 8000 + site); }
    bool any(bool b, int site = 0)
    {
        const int g = st->gen[lane] & 1;
        st->slot[g][lane] = b  ?  1u  :  0u;
        st->gen[lane]++;
        quad_host_yield(st, lane, 4000 + site);
        return (st->slot[g][0] | st->slot[g][1] | st->slot[g][2] | st->slot[g][3]) != 0;
    }
    void sync(int site = 0)
    {
        st->gen[lane]++;
        quad_host_yield(st, lane, 5000 + site);
    }
};

}   // namespace spg

#else

#include <hip/hip_runtime.h>

#define SPG_FN              __device__ __forceinline__
#define SPG_FN_NOINLINE     __device__ __noinline__
#define SPG_UNROLL          _Pragma("unroll")
// nothing is scheduled across this point: keeps the loads of one group of taps from being hoisted above the arithmetic of
// the group before (left alone the scheduler requests a whole inner product's operands at once and runs out of registers)
#define SPG_SCHED_FENCE()   __builtin_amdgcn_sched_barrier(0)
// s_waitcnt vmcnt(0), once, before a loop that only stores to global memory: the state words loaded ahead of the loop are
// first used inside it, the compiler therefore puts its vmcnt(0) there, and on gfx9 stores count in vmcnt too -- every
// iteration would sit out the round trip of the event bytes the one before it stored
#define SPG_LOADS_DONE()    __builtin_amdgcn_s_waitcnt(0x0F70)

namespace spg {

struct QuadDev
{
    int lane4;
    __device__ __forceinline__ int role() const { return lane4; }
    template <int S> __device__ __forceinline__ float bcast(float v, int = 0)
    {
        return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), S*0x55, 0xF, 0xF, true));
    }
    template <int S> __device__ __forceinline__ int bcast(int v, int = 0)
    {
        return __builtin_amdgcn_mov_dpp(v, S*0x55, 0xF, 0xF, true);
    }
    __device__ __forceinline__ float swap2(float v, int = 0)
    {
        return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));     // quad_perm:[2,3,0,1]
    }
    __device__ __forceinline__ uint32_t swap1(uint32_t v, int = 0)
    {
        return (uint32_t) __builtin_amdgcn_mov_dpp((int) v, 0xB1, 0xF, 0xF, true);                     // quad_perm:[1,0,3,2]
    }
    __device__ __forceinline__ float swap1f(float v, int = 0)
    {
        return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
    }
    __device__ __forceinline__ uint32_t swap2(uint32_t v, int = 0)
    {
        return (uint32_t) __builtin_amdgcn_mov_dpp((int) v, 0x4E, 0xF, 0xF, true);
    }
    // the value of the lane before (lane 0 of the quad: its own)
    __device__ __forceinline__ int prev1(int v, int = 0)
    {
        return __builtin_amdgcn_mov_dpp(v, 0x90, 0xF, 0xF, true);                                      // quad_perm:[0,0,1,2]
    }
    __device__ __forceinline__ bool any(bool b, int = 0) { return __any(b); }
    // The lanes of a quad hand data to each other through LDS at these points.  A wave's LDS operations complete in order, so
    // there is nothing for the hardware to do; the fence keeps the COMPILER from moving a lane's reads over stores that
    // lane may not execute itself (its neighbour's are what it is after).
    __device__ __forceinline__ void sync(int = 0)
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
};

}   // namespace spg

#endif
