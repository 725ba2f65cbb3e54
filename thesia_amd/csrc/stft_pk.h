// stft_pk.h — packed-f32 (v_pk_*_f32) complex arithmetic for the wave FFT plans (round 4).
//
// gfx950 issues a v_pk_fma_f32 (two FMAs per lane) in 5.2 cycles against 3.0 for a v_fma_f32 (scripts/ubench/valu_rate.hip):
// 13 % fewer issue cycles per flop and HALF the instructions.  Built as the lever VERDICT r3 named (the memory skeleton with
// the kernel's 672 FMAs as 336 packed ones had run 4.5 % faster); MEASURED: the n_fft 2048 kernel executes 417 VALU
// instructions per frame instead of 681 and takes the same time, box by box (profiles/r04_ab_packed.txt; the skeleton's
// gain was its dependent FMA chains getting shorter).  The scalar pipeline stays the default; this one is selector 9 of
// th_plan_set_kernel, parity-tested on the GPU and on the CPU lane emulator.  Two layouts are used, each where the data
// arrives in it for free:
//   AoS  v2f = (re, im) of ONE point        — the 8-byte global loads deliver x[2n], x[2n+1] = (re, im) of packed point n: the
//        window multiply and the whole un-twiddled first pass run on these.  A complex add / sub is one packed add, a
//        multiplication by -i / +i an operand swap + sign (op_sel / neg modifiers: free), a general complex multiply 2 packed.
//   SoA  v2f = the re (or im) parts of TWO points — what a ds_read_b128 of a plane exchange returns (four neighbouring lanes'
//        values of one component).  A twiddled radix-2 butterfly s = x + t y, d = 2 x - s on two butterflies at once is 6
//        packed FMAs (12 scalar) when the pair members are different butterflies ("clean" levels), and 4 packed FMAs per
//        butterfly (6 scalar) when the butterfly's two inputs ARE the pair ("cross" levels: the result pair is (s, d)).
// The LDS exchange between the passes converts AoS to SoA for nothing: ds_write_addtid_b32 stores single dwords (sub-registers
// of a pair), ds_read_b128 returns neighbouring points' components.
//
// Modifier semantics of VOP3P with 32-bit halves (pairs of VGPRs): op_sel[i] picks the half of source i that feeds the LOW
// result, op_sel_hi[i] the half that feeds the HIGH result (default 1), neg_lo / neg_hi negate source i for the low / high
// result.  Plain element-wise operations are left to the compiler (it selects v_pk_* for 2-vectors and folds whole-vector
// negations and broadcasts); operations with mixed signs or half swaps are inline asm, because the compiler materialises those
// with v_xor / v_mov.  No inline-asm operand is ever the direct result of a transcendental (gfx940 trans-forwarding hazard:
// the compiler's hazard recogniser does not look into asm).
//
// Like stft_wave.h this compiles for gfx950 (hipcc) and for the CPU lane emulator (g++, plain structs).
#pragma once
#include "stft_core.h"

namespace th {

#if defined(__HIP_DEVICE_COMPILE__)
typedef float v2f __attribute__((ext_vector_type(2)));
TH_HD v2f mk2(float a, float b) { return (v2f){a, b}; }
TH_HD v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
#else
struct v2f {
    float x, y;
};
TH_HD v2f mk2(float a, float b) { return {a, b}; }
TH_HD v2f operator+(v2f a, v2f b) { return {a.x + b.x, a.y + b.y}; }
TH_HD v2f operator-(v2f a, v2f b) { return {a.x - b.x, a.y - b.y}; }
TH_HD v2f operator*(v2f a, v2f b) { return {a.x * b.x, a.y * b.y}; }
TH_HD v2f operator-(v2f a) { return {-a.x, -a.y}; }
TH_HD v2f pk_fma(v2f a, v2f b, v2f c) { return {__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)}; }
#endif
TH_HD v2f pk_splat(float a) { return mk2(a, a); }
TH_HD v2f pk_swap(v2f a) { return mk2(a.y, a.x); }
// 2 x - s  (the second output of a butterfly whose first output is s = x + t y)
TH_HD v2f pk_2x_minus(v2f x, v2f s) { return pk_fma(x, pk_splat(2.0f), -s); }

#if defined(__HIP_DEVICE_COMPILE__)
#define TH_PK2(NAME, OP, MODS)                                                 \
    __device__ __forceinline__ v2f NAME(v2f a, v2f b) {                        \
        v2f r;                                                                 \
        asm(OP " %0, %1, %2 " MODS : "=v"(r) : "v"(a), "v"(b));               \
        return r;                                                              \
    }
#define TH_PK3(NAME, MODS)                                                     \
    __device__ __forceinline__ v2f NAME(v2f a, v2f b, v2f c) {                 \
        v2f r;                                                                 \
        asm("v_pk_fma_f32 %0, %1, %2, %3 " MODS : "=v"(r) : "v"(a), "v"(b), "v"(c)); \
        return r;                                                              \
    }
// the same with the second operand in an SGPR pair (wave-uniform constants: one constant-bus read, no VGPRs)
#define TH_PK2S(NAME, OP, MODS)                                                \
    __device__ __forceinline__ v2f NAME(v2f a, v2f b) {                        \
        v2f r;                                                                 \
        asm(OP " %0, %1, %2 " MODS : "=v"(r) : "v"(a), "s"(b));               \
        return r;                                                              \
    }
#define TH_PK3S(NAME, MODS)                                                    \
    __device__ __forceinline__ v2f NAME(v2f a, v2f b, v2f c) {                 \
        v2f r;                                                                 \
        asm("v_pk_fma_f32 %0, %1, %2, %3 " MODS : "=v"(r) : "v"(a), "s"(b), "v"(c)); \
        return r;                                                              \
    }
#else
#define TH_PK2(NAME, OP, MODS)
#define TH_PK3(NAME, MODS)
#define TH_PK2S(NAME, OP, MODS)
#define TH_PK3S(NAME, MODS)
#endif

// ---------------------------------------------------------------------------------------------- AoS: v2f = (re, im)
// a - i b = (a.re + b.im, a.im - b.re)
TH_PK2(c_add_mi, "v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")
// a + i b = (a.re - b.im, a.im + b.re)
TH_PK2(c_add_pi, "v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")
// (a.re w.re, a.im w.re)
TH_PK2(c_mul_re, "v_pk_mul_f32", "op_sel:[0,0] op_sel_hi:[1,0]")
// (c.re - a.im w.im, c.im + a.re w.im)
TH_PK3(c_fma_im, "op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]")
TH_PK2S(c_mul_re_k, "v_pk_mul_f32", "op_sel:[0,0] op_sel_hi:[1,0]")
TH_PK3S(c_fma_im_k, "op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]")
// (a.x + b.y, a.y + b.x)   and   (a.x - b.y, a.y - b.x): add / subtract with the second operand's halves swapped
TH_PK2(pk_add_sw, "v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0]")
TH_PK2(pk_sub_sw, "v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]")
#if !defined(__HIP_DEVICE_COMPILE__)
TH_HD v2f c_mul_re_k(v2f a, v2f w) { return {a.x * w.x, a.y * w.x}; }
TH_HD v2f c_fma_im_k(v2f a, v2f w, v2f c) { return {__builtin_fmaf(-a.y, w.y, c.x), __builtin_fmaf(a.x, w.y, c.y)}; }
TH_HD v2f pk_add_sw(v2f a, v2f b) { return {a.x + b.y, a.y + b.x}; }
TH_HD v2f pk_sub_sw(v2f a, v2f b) { return {a.x - b.y, a.y - b.x}; }
TH_HD v2f c_add_mi(v2f a, v2f b) { return {a.x + b.y, a.y - b.x}; }
TH_HD v2f c_add_pi(v2f a, v2f b) { return {a.x - b.y, a.y + b.x}; }
TH_HD v2f c_mul_re(v2f a, v2f w) { return {a.x * w.x, a.y * w.x}; }
TH_HD v2f c_fma_im(v2f a, v2f w, v2f c) { return {__builtin_fmaf(-a.y, w.y, c.x), __builtin_fmaf(a.x, w.y, c.y)}; }
#endif
// complex product a w (two packed instructions); _k: w is a wave-uniform constant (SGPR pair)
TH_HD v2f c_mul(v2f a, v2f w) { return c_fma_im(a, w, c_mul_re(a, w)); }
TH_HD v2f c_mul_k(v2f a, v2f w) { return c_fma_im_k(a, w, c_mul_re_k(a, w)); }

// un-twiddled radix-4 butterfly, forward (the same outputs as fft4, stft_core.h): 8 packed adds
TH_HD void pk_fft4(v2f &a0, v2f &a1, v2f &a2, v2f &a3) {
    const v2f b0 = a0 + a2, b1 = a0 - a2, b2 = a1 + a3, t = a1 - a3;  // b3 = -i t
    a0 = b0 + b2;
    a1 = c_add_mi(b1, t);
    a2 = b0 - b2;
    a3 = c_add_pi(b1, t);
}
// the same with input a2 standing for (-i a2): W16^4 of the 4 x 4 decomposition folded into the butterfly
TH_HD void pk_fft4_a2_negi(v2f &a0, v2f &a1, v2f &a2, v2f &a3) {
    const v2f b0 = c_add_mi(a0, a2), b1 = c_add_pi(a0, a2), b2 = a1 + a3, t = a1 - a3;
    a0 = b0 + b2;
    a1 = c_add_mi(b1, t);
    a2 = b0 - b2;
    a3 = c_add_pi(b1, t);
}

// DFT-16 as 4 x 4 on AoS points, the instruction-for-instruction packed form of dft16_head + dft16_tail<0..3> (stft_wave.h):
// natural output X[k], k = 4 k1 + k2, sits in slot k1 + 4 k2.  pk_dft16_head leaves v[10] UN-multiplied by W16^4 = -i: the
// tail of group 2 takes it in that form (pk_dft16_tail<2>).
TH_HD void pk_dft16_head(v2f (&v)[16]) {
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;  // cos, sin(pi/8)
    const float h = 0.70710678118654752440f;
    pk_fft4(v[0], v[4], v[8], v[12]);
    pk_fft4(v[1], v[5], v[9], v[13]);
    pk_fft4(v[2], v[6], v[10], v[14]);
    pk_fft4(v[3], v[7], v[11], v[15]);
    // twiddle W16^(n1 k2) on v[n1 + 4 k2]; W16^m = (cos(pi m / 8), -sin(pi m / 8))
    v[5] = c_mul_k(v[5], mk2(c1, -s1));                  // W^1
    v[9] = c_add_mi(v[9], v[9]) * pk_splat(h);         // W^2 = h (1 - i):  h (re + im, im - re)
    v[13] = c_mul_k(v[13], mk2(s1, -c1));                // W^3
    v[6] = c_add_mi(v[6], v[6]) * pk_splat(h);         // W^2
    /* v[10]: W^4 = -i, folded into pk_dft16_tail<2> */
    v[14] = c_add_pi(v[14], v[14]) * pk_splat(-h);     // W^6 = -h (1 + i): -h (re - im, re + im)
    v[7] = c_mul_k(v[7], mk2(s1, -c1));                  // W^3
    v[11] = c_add_pi(v[11], v[11]) * pk_splat(-h);     // W^6
    v[15] = c_mul_k(v[15], mk2(-c1, s1));                // W^9
}
template <int G>
TH_HD void pk_dft16_tail(v2f (&v)[16]) {
    if constexpr (G == 2) pk_fft4_a2_negi(v[8], v[9], v[10], v[11]);
    else pk_fft4(v[4 * G], v[4 * G + 1], v[4 * G + 2], v[4 * G + 3]);
}

// ---------------------------------------------------------------------------------------------- SoA: v2f = one component of TWO points
// Twiddle t = (t.re, t.im) in one v2f, the same for both points of a pair.
//   acc + u t.re          acc - u t.im          acc + u t.im
TH_PK3(s_fma_tre, "op_sel:[0,0,0] op_sel_hi:[1,0,1]")
TH_PK3(s_fms_tim, "op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0]")
TH_PK3(s_fma_tim, "op_sel:[0,1,0] op_sel_hi:[1,1,1]")
TH_PK3(s_fms_tre, "op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]")
#if !defined(__HIP_DEVICE_COMPILE__)
TH_HD v2f s_fma_tre(v2f u, v2f t, v2f acc) { return {__builtin_fmaf(u.x, t.x, acc.x), __builtin_fmaf(u.y, t.x, acc.y)}; }
TH_HD v2f s_fms_tim(v2f u, v2f t, v2f acc) { return {__builtin_fmaf(-u.x, t.y, acc.x), __builtin_fmaf(-u.y, t.y, acc.y)}; }
TH_HD v2f s_fma_tim(v2f u, v2f t, v2f acc) { return {__builtin_fmaf(u.x, t.y, acc.x), __builtin_fmaf(u.y, t.y, acc.y)}; }
TH_HD v2f s_fms_tre(v2f u, v2f t, v2f acc) { return {__builtin_fmaf(-u.x, t.x, acc.x), __builtin_fmaf(-u.y, t.x, acc.y)}; }
#endif
// "Clean" level: two butterflies (x, y) -> (x + t y, x - t y) at once; x, y are pairs of different butterflies' inputs.
// The packed form of bfly2_tw (stft_wave.h): 6 packed FMAs.
TH_HD void s_bfly2(v2f &xr, v2f &xi, v2f &yr, v2f &yi, v2f t) {
    const v2f sr = s_fms_tim(yi, t, s_fma_tre(yr, t, xr));
    const v2f si = s_fma_tim(yr, t, s_fma_tre(yi, t, xi));
    yr = pk_2x_minus(xr, sr);
    yi = pk_2x_minus(xi, si);
    xr = sr;
    xi = si;
}
// the same with the twiddle -i t: (t.im, -t.re)
TH_HD void s_bfly2_mi(v2f &xr, v2f &xi, v2f &yr, v2f &yi, v2f t) {
    const v2f sr = s_fma_tre(yi, t, s_fma_tim(yr, t, xr));   // x.re + t.im y.re + t.re y.im
    const v2f si = s_fms_tre(yr, t, s_fma_tim(yi, t, xi));   // x.im + t.im y.im - t.re y.re
    yr = pk_2x_minus(xr, sr);
    yi = pk_2x_minus(xi, si);
    xr = sr;
    xi = si;
}
// Radix-4 decimation-in-time butterfly with the input twiddles (1, t, t^2, t^3) on TWO butterflies at once (bfly4_tw of
// stft_wave.h, pair member by pair member): in (a, b, c, d) = inputs 0..3, out a = X0, c = X1, b = X2, d = X3.  48 -> 24.
TH_HD void s_bfly4(v2f &ar, v2f &ai, v2f &br, v2f &bi, v2f &cr, v2f &ci, v2f &dr, v2f &di, v2f t, v2f t2) {
    s_bfly2(ar, ai, cr, ci, t2);
    s_bfly2(br, bi, dr, di, t2);
    s_bfly2(ar, ai, br, bi, t);
    s_bfly2_mi(cr, ci, dr, di, t);
}

// "Cross" level: the butterfly's two inputs are the halves of ONE pair, p = (a, b): (p.re, p.im) -> ((s, d).re, (s, d).im)
// with s = a + t b, d = a - t b.  4 packed FMAs per butterfly (6 scalar).
//   (u.lo + u.hi t.re, u.lo - u.hi t.re)      acc + (-w.hi t.im, +w.hi t.im)      acc + (w.hi t.im, -w.hi t.im)
TH_PK3(x_fma_tre, "op_sel:[1,0,0] op_sel_hi:[1,0,0] neg_hi:[0,1,0]")
TH_PK3(x_fms_tim, "op_sel:[1,1,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0]")
TH_PK3(x_fma_tim, "op_sel:[1,1,0] op_sel_hi:[1,1,1] neg_hi:[0,1,0]")
//   the same for the twiddle -i t = (t.im, -t.re):
//   (u.lo + u.hi t.im, u.lo - u.hi t.im)      acc + (w.hi t.re, -w.hi t.re)       acc + (-w.hi t.re, +w.hi t.re)
TH_PK3(x_fma_tim0, "op_sel:[1,1,0] op_sel_hi:[1,1,0] neg_hi:[0,1,0]")
TH_PK3(x_fma_tre1, "op_sel:[1,0,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]")
TH_PK3(x_fms_tre1, "op_sel:[1,0,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]")
#if !defined(__HIP_DEVICE_COMPILE__)
TH_HD v2f x_fma_tre(v2f u, v2f t, v2f u2) { return {__builtin_fmaf(u.y, t.x, u2.x), __builtin_fmaf(-u.y, t.x, u2.x)}; }
TH_HD v2f x_fms_tim(v2f w, v2f t, v2f acc) { return {__builtin_fmaf(-w.y, t.y, acc.x), __builtin_fmaf(w.y, t.y, acc.y)}; }
TH_HD v2f x_fma_tim(v2f w, v2f t, v2f acc) { return {__builtin_fmaf(w.y, t.y, acc.x), __builtin_fmaf(-w.y, t.y, acc.y)}; }
TH_HD v2f x_fma_tim0(v2f u, v2f t, v2f u2) { return {__builtin_fmaf(u.y, t.y, u2.x), __builtin_fmaf(-u.y, t.y, u2.x)}; }
TH_HD v2f x_fma_tre1(v2f w, v2f t, v2f acc) { return {__builtin_fmaf(w.y, t.x, acc.x), __builtin_fmaf(-w.y, t.x, acc.y)}; }
TH_HD v2f x_fms_tre1(v2f w, v2f t, v2f acc) { return {__builtin_fmaf(-w.y, t.x, acc.x), __builtin_fmaf(w.y, t.x, acc.y)}; }
#endif
// p = (a, b) -> (a + t b, a - t b)
TH_HD void x_bfly2(v2f &pr, v2f &pi, v2f t) {
    const v2f sr = x_fms_tim(pi, t, x_fma_tre(pr, t, pr));  // a.re + t.re b.re - t.im b.im | a.re - t.re b.re + t.im b.im
    const v2f si = x_fma_tim(pr, t, x_fma_tre(pi, t, pi));  // a.im + t.re b.im + t.im b.re | a.im - t.re b.im - t.im b.re
    pr = sr;
    pi = si;
}
// p = (a, b) -> (a + (-i t) b, a - (-i t) b)
TH_HD void x_bfly2_mi(v2f &pr, v2f &pi, v2f t) {
    const v2f sr = x_fma_tre1(pi, t, x_fma_tim0(pr, t, pr));  // a.re + t.im b.re + t.re b.im | a.re - t.im b.re - t.re b.im
    const v2f si = x_fms_tre1(pr, t, x_fma_tim0(pi, t, pi));  // a.im + t.im b.im - t.re b.re | a.im - t.im b.im + t.re b.re
    pr = sr;
    pi = si;
}
// Radix-4 DIT butterfly with input twiddles (1, t, t^2, t^3) on ONE butterfly whose inputs are the pairs p = (in0, in1),
// q = (in2, in3):  out p = (X0, X2), q = (X1, X3).  6 + 8 = 14 packed FMAs (24 scalar).
TH_HD void x_bfly4(v2f &pr, v2f &pi, v2f &qr, v2f &qi, v2f t, v2f t2) {
    s_bfly2(pr, pi, qr, qi, t2);  // (in0, in1) +- t^2 (in2, in3)
    x_bfly2(pr, pi, t);           // X0, X2
    x_bfly2_mi(qr, qi, t);        // X1, X3
}

}  // namespace th
