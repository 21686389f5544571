"""Timing-only builds of the V.29 four-lane kernel: ONE phase of a round taken out (5, 7, 8: results are wrong on purpose), carried out
twice (11 .. 16: results unchanged) or the four waves of a workgroup started apart (21 .. 23).  What a phase costs on the launch's
critical path is the launch time without it, or the time a second copy of it adds.  Writes tools/experiments/libspangpu_ko<N>.so
(git-ignored); run with  MQ_LIBS="ko1 ko2 ..." MQ_W=v29 bash tools/gpu_round5.sh modem_quick  on the GPU box.
Usage: python3 tools/v29_knockouts.py [N ...]"""
import os, shutil, subprocess, sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "spandsp_amd", "csrc")
FLAGS = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function "
         "-mllvm -amdgpu-kernarg-preload-count=16 -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp").split()

KO = {
    # knock-outs that leave the receivers' behaviour alone (taking out the sums, the power estimate or the carrier's sine changes
    # what the receivers do afterwards -- they drop out of data mode -- and was measured with the doubled builds below instead)
    5: ("the slicer's two table look-ups", [("v29_quad.hpp", "                    nearest = T.space_map[re*20 + im];\n                    const bool full", "                    nearest = (re + im) & 15;\n                    const bool full"),
                                            ("v29_quad.hpp", "                const float tre = T.konst[2*nearest];\n                const float tim = T.konst[2*nearest + 1];\n                do_track = true;", "                const float tre = (float) nearest;\n                const float tim = (float) (nearest ^ 5);\n                do_track = true;")]),
    7: ("the LMS update (and the reload of the register taps behind it)", [("v29_quad.hpp", "        if (q.any(do_tune, 12))\n", "        if (q.any(do_tune  &&  pos < 0, 12))\n")]),
    8: ("the events' store", [("v29_quad.hpp", "                if (role == 0)\n                {\n                    if (n_ev + 4 <= L.ev_cap)", "                if (role == 0  &&  pos < 0)\n                {\n                    if (n_ev + 4 <= L.ev_cap)")]),
}

# Doubling: the phase is carried out TWICE (the second time on operands the optimiser cannot tell from the first, into values that
# are kept alive but unused), results unchanged -- unlike a knock-out this does not change what the receivers do afterwards.
OPQ = "int zz_ = 0; asm volatile(\"\" : \"+v\"(zz_));"
KEEP = lambda v: "asm volatile(\"\" :: \"v\"(%s));" % v
KO.update({
    11: ("(doubled) the equaliser's inner product", [("v29_quad.hpp", "                float z = acc + q.swap2(acc, 1);\n                if (q.any(!(fabsf(z) < __builtin_inff()), 5))", "                { " + OPQ + " const float2 *xd = x + zz_; float dd = 0.0f;\n                  SPG_UNROLL for (int i = 0;  i < kEqLen;  i++) { const f32x2v pd = (f32x2v) {xd[i].x, xd[i].y}*tc[i]; dd += pd.x - pd.y; }\n                  " + KEEP("dd") + " }\n                float z = acc + q.swap2(acc, 1);\n                if (q.any(!(fabsf(z) < __builtin_inff()), 5))")]),
    12: ("(doubled) the shaping sums", [("quad_round_front.inc", "            vre = are.x + are.y;\n", "            vre = are.x + are.y;\n            { " + OPQ + " const float2 *xd = xw + zz_; f32x2v dre_ = {0.0f, 0.0f}; f32x2v dim_ = {0.0f, 0.0f};\n              SPG_UNROLL for (int i = 0;  i < kRrcLen;  i++) { const float2 xv = xd[i]; const float2 cv = QF_RRC_COEF(i); dre_ += (f32x2v) {xv.x, xv.y}*(f32x2v) {cv.x, cv.x}; dim_ += (f32x2v) {xv.x, xv.y}*(f32x2v) {cv.y, cv.y}; }\n              " + KEEP("dre_.x + dre_.y") + KEEP("dim_.x + dim_.y") + " }\n")]),
    13: ("(doubled) the power estimate's chain", [("quad_round_front.inc", "            calm_sample(QuadTag<3>{}, pw3);\n", "            calm_sample(QuadTag<3>{}, pw3);\n            { " + OPQ + " int s_pr = power_reading + zz_, s_high = high_sample + zz_, s_low = low_samples + zz_, s_bad = 0;\n              auto dgo = [&](const int k, const int sq, const int ad, const int ad10) { if (k < m) { const int power = s_pr + ((sq - s_pr) >> 4); s_bad |= (power < off1)  ?  1  :  0; const bool low = (ad10 < s_high); const int low_inc = s_low + 1; const bool wipe = low  &&  (low_inc > 120); s_pr = wipe  ?  0  :  power; s_high = low  ?  (wipe  ?  0  :  s_high)  :  max(s_high, ad); s_low = low  ?  (wipe  ?  0  :  low_inc)  :  0; } };\n              dgo(0, q.template bcast<0>(my_sq, 5), q.template bcast<0>(my_ad, 9), q.template bcast<0>(my_ad10, 13)); dgo(1, q.template bcast<1>(my_sq, 6), q.template bcast<1>(my_ad, 10), q.template bcast<1>(my_ad10, 14)); dgo(2, q.template bcast<2>(my_sq, 7), q.template bcast<2>(my_ad, 11), q.template bcast<2>(my_ad10, 15)); dgo(3, q.template bcast<3>(my_sq, 8), q.template bcast<3>(my_ad, 12), q.template bcast<3>(my_ad10, 16));\n              " + KEEP("s_pr") + KEEP("s_high") + KEEP("s_low") + KEEP("s_bad") + " }\n")]),
    14: ("(doubled) the Godard filters per sample", [("quad_round_front.inc", "            godard_sample(QuadTag<0>{});\n", "            { " + OPQ + " float a0 = glow0 + (float) zz_, a1 = glow1, b0 = ghigh0, b1 = ghigh1;\n              auto dg = [&](const bool acc_k, const float sre) { if (acc_k) { const float tl = a0*g0 + a1*g1 + sre; a1 = a0; a0 = tl; const float th = b0*g3 + b1*g4 + sre; b1 = b0; b0 = th; } };\n              dg((flags & (F_ACC << 0)) != 0, q.template bcast<0>(my_sre, 11)); dg((flags & (F_ACC << 1)) != 0, q.template bcast<1>(my_sre, 12)); dg((flags & (F_ACC << 2)) != 0, q.template bcast<2>(my_sre, 13)); dg((flags & (F_ACC << 3)) != 0, q.template bcast<3>(my_sre, 14));\n              " + KEEP("a0") + KEEP("a1") + KEEP("b0") + KEEP("b1") + " }\n            godard_sample(QuadTag<0>{});\n")]),
    15: ("(doubled) the T/2 instants' sine look-ups and products", [("quad_round_front.inc", "            const float2 h = make_float2(hre, him);\n", "            const float2 h = make_float2(hre, him);\n            { " + OPQ + " const float d1 = T.sine[((uint32_t) (my_cp + (1u << 30)) >> 21) + zz_]; const float d2 = T.sine[(my_cp >> 21) + zz_]; " + KEEP("my_sre*d1 - sim*d2") + KEEP("-my_sre*d2 - sim*d1") + " }\n")]),
})

# Skew: the four waves of a workgroup (one per SIMD, sharing the CU's LDS pipe) start their channels a fraction of a round apart
def skew(unit):
    code = "    { const int wq = (int) (threadIdx.x >> 6); if (wq == 1) __builtin_amdgcn_s_sleep(%d); else if (wq == 2) __builtin_amdgcn_s_sleep(%d); else if (wq == 3) __builtin_amdgcn_s_sleep(%d); }\n" % (unit, 2*unit, 3*unit)
    return [("v29_quad.hpp", "    __syncthreads();\n    const int lane = threadIdx.x & 63;\n    const int wv = (int) (threadIdx.x >> 6);\n    const int cw = lane >> 2;\n    const int ch = (blockIdx.x*WPB + wv)*CPW + cw;\n    if (cw >= CPW  ||  ch >= L.n_ch)\n        return;\n    QuadDev q{lane & 3};\n    const int slot = wv*CPW + cw;\n    const V29QuadChan C = {s_pcm + slot*kQuadPcmStride, s_rrc + slot*kQuadRrcStride, s_u + slot*kQuadEqStride, s_taps + slot*kQuadTapStride};\n",
             "    __syncthreads();\n" + code + "    const int lane = threadIdx.x & 63;\n    const int wv = (int) (threadIdx.x >> 6);\n    const int cw = lane >> 2;\n    const int ch = (blockIdx.x*WPB + wv)*CPW + cw;\n    if (cw >= CPW  ||  ch >= L.n_ch)\n        return;\n    QuadDev q{lane & 3};\n    const int slot = wv*CPW + cw;\n    const V29QuadChan C = {s_pcm + slot*kQuadPcmStride, s_rrc + slot*kQuadRrcStride, s_u + slot*kQuadEqStride, s_taps + slot*kQuadTapStride};\n")]
KO.update({21: ("(skew) waves 1/4 round apart", skew(29)), 22: ("(skew) waves 1/8 round apart", skew(15)), 23: ("(skew) waves 1/16 round apart", skew(7))})


def build(n):
    what, edits = KO[n]
    d = "/tmp/v29_ko/%d" % n
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    for f in os.listdir(SRC):
        if f.endswith((".hpp", ".inc", ".h", ".hip")):
            shutil.copy(os.path.join(SRC, f), d)
    for f, old, new in edits:
        p = os.path.join(d, f)
        s = open(p).read()
        assert s.count(old) == 1, (n, f, old, s.count(old))
        open(p, "w").write(s.replace(old, new))
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-I", os.path.join(ROOT, "include"), "-c", os.path.join(d, "modem_v29q.hip"), "-o", os.path.join(d, "modem_v29q.o")],
                          stderr=subprocess.DEVNULL)
    objs = [os.path.join(SRC, f) for f in sorted(os.listdir(SRC)) if (f.endswith(".o") or f.endswith(".co")) and f != "modem_v29q.o"]
    out = os.path.join(ROOT, "tools", "experiments", "libspangpu_ko%d.so" % n)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic-functions", "-o", out, os.path.join(d, "modem_v29q.o")] + objs + ["-lm"])
    return "ko%d: %s%s" % (n, "" if what.startswith("(") else "without ", what)


if __name__ == "__main__":
    which = [int(a) for a in sys.argv[1:]] or sorted(KO)
    with ThreadPoolExecutor(4) as ex:
        for line in ex.map(build, which):
            print(line)
