#!/usr/bin/env python3
"""Writes spandsp_amd/csrc/tone_pairs_asm.inc: the sixteen sample pairs of one 64-byte row piece as ONE inline-asm body
that can be entered at any pair and left after any pair (tone_fast.hpp, the segment that holds a block end).

Why generated: every pair block differs from its neighbours only in constants (which LDS dword it converts, which dword is
requested three pairs ahead, how many LDS reads may still be on their way), and the body exists for every number of packed
bin pairs a detector has (NP) with and without the block energy.  The arithmetic of a pair is Bank<>::step2's: per bin pair
v_pk_mul_f32, v_pk_add_f32 (neg), v_pk_add_f32 with op_sel picking the sample -- (fac*v2 - v1) + x rounded three times, as
src/spandsp/tone_detect.h:172-192 does -- and energy += x*x per sample in sample order (src/dtmf.c:199).

Fixed registers (clobbered): T_i = v[100+2i:101+2i] (i < 4), X = v[108:109], SQ = v[110:111], D0..D3 = v112..v115
(as low as the kernel's own register use allows: the highest one sets the wave's allocation).
A packed result must not be read by the very next instruction (one wait state on gfx950): the order below keeps every
reader of a 64-bit result at least one instruction away.
"""
import os

OUT = os.path.join(os.path.dirname(__file__), "..", "spandsp_amd", "csrc", "tone_pairs_asm.inc")


def T(i):
    return "v[%d:%d]" % (100 + 2*i, 101 + 2*i)


X = "v[108:109]"
XL, XH = "v108", "v109"
SQ = "v[110:111]"
SQL, SQH = "v110", "v111"


def D(k):
    return "v%d" % (112 + (k % 4))


def read(k):
    return "ds_read_b32 %s, %%[ad%d] offset:%d" % (D(k), k//4, 4*(k % 4))


def body(np_, energy, fma=False):
    L = []
    a = lambda i: "%%[a%d]" % i
    b = lambda i: "%%[b%d]" % i
    f = lambda i: "%%[f%d]" % i
    L.append("s_waitcnt lgkmcnt(0)")                  # scalar loads of the compiler's may be out: they return out of order
    # entry: k0 == 0 first (every segment's first part), then a binary search
    L.append("s_cmp_eq_u32 %[k0], 0")
    L.append("s_cbranch_scc1 Le%=_0")

    def tree(lo, hi):
        if lo == hi:
            L.append("s_branch Le%%=_%d" % lo)
            return
        mid = (lo + hi + 1)//2
        L.append("s_cmp_lt_u32 %%[k0], %d" % mid)
        L.append("s_cbranch_scc1 Lt%%=_%d_%d" % (lo, mid - 1))
        tree(mid, hi)
        L.append("Lt%%=_%d_%d:" % (lo, mid - 1))
        tree(lo, mid - 1)
    tree(1, 15)
    # trampolines 1..15, then 0 falling into pair 0
    for k in list(range(1, 16)) + [0]:
        L.append("Le%%=_%d:" % k)
        for j in range(k, min(k + 3, 16)):
            L.append(read(j))
        if k != 0:
            L.append("s_branch Lp%%=_%d" % k)
    for k in range(16):
        L.append("Lp%%=_%d:" % k)
        if k + 3 < 16:
            L.append(read(k + 3))
        L.append("s_waitcnt lgkmcnt(%d)" % min(3, 15 - k))
        L.append("v_cvt_f32_i32_sdwa %s, sext(%s) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" % (XL, D(k)))
        L.append("v_cvt_f32_i32_sdwa %s, sext(%s) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" % (XH, D(k)))
        if fma:
            # the experiment of round 6 (-DSPG_TONE_FMA, NOT bit-exact): fac*v2 - v1 as ONE v_pk_fma_f32 -- 2 packed operations
            # per bin pair and sample in place of 3 -- then + x as before
            L += ["v_pk_fma_f32 %s, %s, %s, %s neg_lo:[0,0,1] neg_hi:[0,0,1]" % (T(i), f(i), b(i), a(i)) for i in range(np_)]
            if energy:
                L.append("v_pk_mul_f32 %s, %s, %s" % (SQ, X, X))
            if np_ == 1:
                L.append("s_nop 0")
            L += ["v_pk_add_f32 %s, %s, %s op_sel_hi:[1,0]" % (a(i), T(i), X) for i in range(np_)]
            if energy:
                L.append("v_add_f32 %%[en], %%[en], %s" % SQL)
            if np_ == 1:
                L.append("s_nop 0")
            L += ["v_pk_fma_f32 %s, %s, %s, %s neg_lo:[0,0,1] neg_hi:[0,0,1]" % (T(i), f(i), a(i), b(i)) for i in range(np_)]
            if np_ == 1:
                L.append("s_nop 0")
            L += ["v_pk_add_f32 %s, %s, %s op_sel:[0,1] op_sel_hi:[1,1]" % (b(i), T(i), X) for i in range(np_)]
            if energy:
                L.append("v_add_f32 %%[en], %%[en], %s" % SQH)
            if k < 15:
                L.append("s_cmp_eq_u32 %%[k1], %d" % (k + 1))
                L.append("s_cbranch_scc1 Lx%=")
            continue
        muls = ["v_pk_mul_f32 %s, %s, %s" % (T(i), f(i), b(i)) for i in range(np_)]
        if energy:
            # cvt, cvt, two products, the squares, the other products, first energy add
            head = muls[:2] + ["v_pk_mul_f32 %s, %s, %s" % (SQ, X, X)] + muls[2:]
            if np_ <= 2:
                head = muls + ["v_pk_mul_f32 %s, %s, %s" % (SQ, X, X)]
            L += head
            subs = ["v_pk_add_f32 %s, %s, %s neg_lo:[0,1] neg_hi:[0,1]" % (T(i), T(i), a(i)) for i in range(np_)]
            if np_ <= 2:
                # SQ was the last thing written: one subtraction in between
                L.append(subs[0])
                L.append("v_add_f32 %%[en], %%[en], %s" % SQL)
                L += subs[1:]
            else:
                L.append("v_add_f32 %%[en], %%[en], %s" % SQL)
                L += subs
        else:
            L += muls
            L += ["v_pk_add_f32 %s, %s, %s neg_lo:[0,1] neg_hi:[0,1]" % (T(i), T(i), a(i)) for i in range(np_)]
        if np_ == 1:
            L.append("s_nop 0")
        L += ["v_pk_add_f32 %s, %s, %s op_sel_hi:[1,0]" % (a(i), T(i), X) for i in range(np_)]
        if np_ == 1:
            L.append("s_nop 0")
        L += ["v_pk_mul_f32 %s, %s, %s" % (T(i), f(i), a(i)) for i in range(np_)]
        if np_ == 1:
            L.append("s_nop 0")
        L += ["v_pk_add_f32 %s, %s, %s neg_lo:[0,1] neg_hi:[0,1]" % (T(i), T(i), b(i)) for i in range(np_)]
        if np_ == 1:
            L.append("s_nop 0")
        L += ["v_pk_add_f32 %s, %s, %s op_sel:[0,1] op_sel_hi:[1,1]" % (b(i), T(i), X) for i in range(np_)]
        if energy:
            L.append("v_add_f32 %%[en], %%[en], %s" % SQH)
        if k < 15:
            L.append("s_cmp_eq_u32 %%[k1], %d" % (k + 1))
            L.append("s_cbranch_scc1 Lx%=")
    L.append("Lx%=:")
    L.append("s_waitcnt lgkmcnt(0)")                  # reads past the exit land in the clobbered registers: before they are anybody else's
    return L


def write(path, fma):
    out = []
    out.append("// %s -- GENERATED by tools/gen_pairs_asm.py; do not edit.  See that file and tone_fast.hpp (pairs_asm)." % os.path.basename(path))
    if fma:
        out.append("// The v_pk_fma_f32 form of the recurrence (fac*v2 - v1 fused: NOT the reference's roundings): only built with -DSPG_TONE_FMA.")
    for np_ in (2, 3, 4):
        for energy in (0, 1):
            out.append("#define SPG_PAIRS_ASM_NP%d_E%d \\" % (np_, energy))
            lines = body(np_, energy, fma)
            for i, l in enumerate(lines):
                out.append('    "%s\\n\\t"%s' % (l, " \\" if i + 1 < len(lines) else ""))
    out.append('#define SPG_PAIRS_ASM_CLOBBERS "memory", "scc", ' + ", ".join('"v%d"' % r for r in range(100, 116)))
    with open(path, "w") as fh:
        fh.write("\n".join(out) + "\n")
    print("wrote", os.path.normpath(path), len(out), "lines")


def main():
    write(OUT, False)
    write(OUT.replace("tone_pairs_asm.inc", "tone_pairs_asm_fma.inc"), True)


if __name__ == "__main__":
    main()
