// stft_wave.h — one-wavefront-per-frame FFT building blocks (the fast STFT kernel).
//
// A 64-lane wave owns one frame: the packed real FFT of n_fft samples is an Nc = n_fft/2 point
// complex FFT with P = Nc/64 points per lane, done as three register-resident radix passes
// (Stockham autosort, decimation in time) with two lane exchanges through a private LDS slab and a
// third exchange for the real-FFT split pass.  No workgroup barrier is needed anywhere: every
// exchange stays inside the wave.
//
// Invariant before every pass: lane j holds in[j + 64*m] in z[m], m = 0..P-1.
// A pass with sub-transform size Ns and radix R runs B = P/R butterflies per lane:
//   butterfly b: jj = j + 64*b, k = jj mod Ns,
//   v_r = z[b + B*r] * W_{Ns*R}^{r*k}                       (r = 0..R-1)
//   out[(jj - k)*R + k + r*Ns] = DFT_R(v)_r
// For the last pass Ns*R = Nc so out index = j + 64*(b + B*r): results stay in registers.
//
// Like stft_core.h this header compiles both for gfx950 (hipcc) and for the CPU lane emulator in
// tests/emu (g++), which runs the phases lane by lane to check the index arithmetic.
#pragma once
#include "stft_core.h"
#include "stft_pk.h"

namespace th {

// 8-byte LDS read that the backend may not fuse with a neighbour.  hipcc pairs adjacent 8-byte LDS
// loads into ds_read2_b64 / ds_read2st64_b64, which on gfx950 occupy the LDS pipe for 8 cycles per
// wave-instruction; two separate ds_read_b64 take 2 x 2.2 (scripts/ubench/lds_rate.hip).  A volatile
// access is left alone by the load/store optimiser and still gets exact s_waitcnt tracking.
TH_HD cf32 lds_ld(const cf32 *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    // LDS address space, explicit: address-space inference does not look through volatile accesses
    // (they would become flat loads).  Loaded as one 64-bit integer and split, not as a float2 vector:
    // vector-typed values make the backend pick half-rate v_pk_*_f32 for the arithmetic that follows.
    const uint64_t v = *(const volatile __attribute__((address_space(3))) uint64_t *)(p);
    return {__builtin_bit_cast(float, (uint32_t)v), __builtin_bit_cast(float, (uint32_t)(v >> 32))};
#else
    return *p;
#endif
}

// 8-byte LDS write (the slot-layout exchanges of the generic plans)
TH_HD void lds_st(cf32 *p, cf32 v) { *p = v; }

// 4-byte LDS read in program order (volatile, explicit LDS address space; see lds_ld)
TH_HD float lds_ldf(const float *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(float, *(const volatile __attribute__((address_space(3))) uint32_t *)(p));
#else
    return *p;
#endif
}

#if defined(__HIPCC__)
// the same by 32-bit LDS byte address (a loop that moves a generic pointer pays a 64-bit addition and a null check per read)
__device__ __forceinline__ uint32_t lds_addr(const float *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float *)(p);
#else
    return 0;
#endif
}
__device__ __forceinline__ float lds_ldf_at(uint32_t a) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(float, *(const volatile __attribute__((address_space(3))) uint32_t *)(uintptr_t)(a));
#else
    return 0.0f;
#endif
}
#endif

// 16-byte LDS read (ds_read_b128, 16-byte aligned address), as four scalars (no vector-typed arithmetic downstream)
struct f32x4 {
    float a, b, c, d;
};
TH_HD f32x4 lds_ld4(const float *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = *(const volatile __attribute__((address_space(3))) u32x4 *)(p);
    // (scalar copies first: __builtin_bit_cast applied to a vector-element lvalue reads element 0 for every element)
    const uint32_t x = v.x, y = v.y, z = v.z, w = v.w;
    return {__builtin_bit_cast(float, x), __builtin_bit_cast(float, y), __builtin_bit_cast(float, z),
            __builtin_bit_cast(float, w)};
#else
    return {p[0], p[1], p[2], p[3]};
#endif
}

// 16-byte LDS write (ds_write_b128, 16-byte aligned address)
TH_HD void lds_st4(float *p, const f32x4 &v) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    f32x4v q;
    q.x = v.a;
    q.y = v.b;
    q.z = v.c;
    q.w = v.d;
    *(__attribute__((address_space(3))) f32x4v *)(p) = q;
#else
    p[0] = v.a;
    p[1] = v.b;
    p[2] = v.c;
    p[3] = v.d;
#endif
}

// 16-byte LDS read as two packed pairs (sub-registers of the loaded quad: no moves), for the packed-f32 plan (stft_pk.h)
struct v2x2 {
    v2f lo, hi;
};
TH_HD v2x2 lds_ld4_pk(const float *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    const f32x4v v = *(const volatile __attribute__((address_space(3))) f32x4v *)(p);
    return {v.xy, v.zw};
#else
    return {{p[0], p[1]}, {p[2], p[3]}};
#endif
}
// 8-byte LDS read as one packed pair, in program order (see lds_ld)
TH_HD v2f lds_ld_pk(const void *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *(const volatile __attribute__((address_space(3))) v2f *)(p);
#else
    const float *f = static_cast<const float *>(p);
    return {f[0], f[1]};
#endif
}

// "Plane" stores: ds_write_addtid_b32 writes one dword per lane at LDS address M0 + offset + 4 * lane — no address
// VGPR, so the store moves one source dword per lane instead of three (ds_write_b64) or five (ds_write_b128): 2
// cycles per 256 bytes against 6 per 512 / 13 per 1024, and 64 consecutive dwords can never bank-conflict.  Four
// complex values (v0..v3) of every lane go to the dword planes at byte offsets RE0 + i * STEP (real parts) and
// IM0 + i * STEP (imaginary parts) of the wave's slab.  `slab` must be wave-uniform.  The compiler does not know these
// are LDS stores: wave_lds_sync() (a compiler fence) must separate them from the reads of the same data, and since
// LDS executes one wave's DS instructions in order and s_waitcnt lgkmcnt(N) waits for all but the N youngest, the
// untracked stores can only make a later compiler-generated wait longer, never too short.
template <int RE0, int IM0, int STEP>
TH_HD void lds_st_planes4(float *slab, uint32_t lane, cf32 v0, cf32 v1, cf32 v2, cf32 v3) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)slab);
    asm volatile(
        "s_mov_b32 m0, %8\n\t"
        "s_nop 0\n\t"  // an SALU write of M0 needs one wait state before an LDS "add-TID" instruction (no interlock)
        "ds_write_addtid_b32 %0 offset:%9\n\t"
        "ds_write_addtid_b32 %1 offset:%10\n\t"
        "ds_write_addtid_b32 %2 offset:%11\n\t"
        "ds_write_addtid_b32 %3 offset:%12\n\t"
        "ds_write_addtid_b32 %4 offset:%13\n\t"
        "ds_write_addtid_b32 %5 offset:%14\n\t"
        "ds_write_addtid_b32 %6 offset:%15\n\t"
        "ds_write_addtid_b32 %7 offset:%16"
        :
        : "v"(v0.re), "v"(v0.im), "v"(v1.re), "v"(v1.im), "v"(v2.re), "v"(v2.im), "v"(v3.re), "v"(v3.im), "s"(base),
          "n"(RE0), "n"(IM0), "n"(RE0 + STEP), "n"(IM0 + STEP), "n"(RE0 + 2 * STEP), "n"(IM0 + 2 * STEP),
          "n"(RE0 + 3 * STEP), "n"(IM0 + 3 * STEP)
        : "memory");
    (void)lane;
#else
    const cf32 v[4] = {v0, v1, v2, v3};
    for (int i = 0; i < 4; i++) {
        slab[(RE0 + i * STEP) / 4 + lane] = v[i].re;
        slab[(IM0 + i * STEP) / 4 + lane] = v[i].im;
    }
#endif
}

// the same from eight scalars (the packed plan stores sub-registers of its pairs)
template <int RE0, int IM0, int STEP>
TH_HD void lds_st_planes4f(float *slab, uint32_t lane, float r0, float i0, float r1, float i1, float r2, float i2, float r3, float i3) {
    lds_st_planes4<RE0, IM0, STEP>(slab, lane, cf32{r0, i0}, cf32{r1, i1}, cf32{r2, i2}, cf32{r3, i3});
}

// full unrolling is required everywhere below: register arrays must never be indexed dynamically
#if defined(__HIP_DEVICE_COMPILE__)
#define TH_UNROLL _Pragma("unroll")
#else
#define TH_UNROLL
#endif

// ---------------------------------------------------------------------------------------------
// register DFTs (forward).  Outputs are left in "digit-reversed slots"; OUT_SLOT maps natural
// output index -> register slot, so callers permute at compile time for free.
// ---------------------------------------------------------------------------------------------
TH_HD cf32 cmul_c(cf32 a, float wr, float wi) { return {a.re * wr - a.im * wi, a.re * wi + a.im * wr}; }

// Twiddled butterflies in fused multiply-adds.  A radix-2 butterfly whose second input carries a twiddle,
//   s = x + t y,   d = x - t y = 2 x - s,
// is six FMAs (two per component of s, one per component of d) where "multiply, then add and subtract" is eight
// operations; x and y are replaced by s and d.
TH_HD float th_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
TH_HD void bfly2_tw(cf32 &x, cf32 &y, float tr, float ti) {
    const float sr = th_fma(-ti, y.im, th_fma(tr, y.re, x.re));
    const float si = th_fma(ti, y.re, th_fma(tr, y.im, x.im));
    y = {th_fma(2.0f, x.re, -sr), th_fma(2.0f, x.im, -si)};
    x = {sr, si};
}
// Radix-4 decimation-in-time butterfly with the input twiddles (1, t, t^2, t^3), as two levels of bfly2_tw: the second
// level's twiddles relative to the first are t^2 again (t^3 / t), then t and -i t, so t^3 is never needed.  24 FMAs
// against 28 operations for three complex multiplies plus an untwiddled radix-4.  In: (a, b, c, d) = inputs 0..3.
// Out: a = X0, c = X1, b = X2, d = X3.
TH_HD void bfly4_tw(cf32 &a, cf32 &b, cf32 &c, cf32 &d, cf32 t, cf32 t2) {
    bfly2_tw(a, c, t2.re, t2.im);  // a = a + t^2 c, c = a - t^2 c
    bfly2_tw(b, d, t2.re, t2.im);  // b = b + t^2 d, d = b - t^2 d
    bfly2_tw(a, b, t.re, t.im);    // X0, X2
    bfly2_tw(c, d, t.im, -t.re);   // X1, X3  (twiddle -i t)
}

// DFT-8 as 2 x 4: n = n1 + 2*n2 (n1 in 0..1, n2 in 0..3), k = 4*k1 + k2
//   v[n1 + 2*k2] <- DFT4 over n2 of v[n1 + 2*n2];  *= W8^(n1*k2);  DFT2 over n1  -> X[4*k1+k2] in v[k1 + 2*k2]
TH_HD void dft8(cf32 (&v)[8]) {
    const float h = 0.70710678118654752440f;
    fft4(v[0], v[2], v[4], v[6]);
    fft4(v[1], v[3], v[5], v[7]);
    // W8^1 = (h, -h), W8^2 = -i, W8^3 = (-h, -h) on v[1 + 2*k2], k2 = 1..3
    v[3] = {h * (v[3].re + v[3].im), h * (v[3].im - v[3].re)};
    v[5] = cmul_negi(v[5]);
    v[7] = {h * (v[7].im - v[7].re), -h * (v[7].re + v[7].im)};
    fft2(v[0], v[1]);
    fft2(v[2], v[3]);
    fft2(v[4], v[5]);
    fft2(v[6], v[7]);
}
// natural output X[k], k = 4*k1 + k2, sits in slot k1 + 2*k2
TH_HD constexpr int dft8_slot(int k) { return (k >> 2) + 2 * (k & 3); }

// DFT-16 as 4 x 4: n = n1 + 4*n2, k = 4*k1 + k2
//   v[n1 + 4*k2] <- DFT4 over n2 of v[n1 + 4*n2];  *= W16^(n1*k2);  DFT4 over n1 -> X[4*k1+k2] in v[k1 + 4*k2]
// dft16 = dft16_head (first four radix-4 butterflies + the W16 twiddles) followed by dft16_tail<G>, G = 0..3: the
// butterfly that produces X[G], X[4 + G], X[8 + G], X[12 + G] in v[4 G .. 4 G + 3] — callers that store the outputs
// can do so group by group.
TH_HD void dft16_head(cf32 (&v)[16]) {
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;  // cos, sin(pi/8)
    const float h = 0.70710678118654752440f;
    fft4(v[0], v[4], v[8], v[12]);
    fft4(v[1], v[5], v[9], v[13]);
    fft4(v[2], v[6], v[10], v[14]);
    fft4(v[3], v[7], v[11], v[15]);
    // twiddle W16^(n1*k2) on v[n1 + 4*k2]; W16^m = (cos(pi m/8), -sin(pi m/8))
    v[5] = cmul_c(v[5], c1, -s1);                                   // n1=1,k2=1: W^1
    v[9] = {h * (v[9].re + v[9].im), h * (v[9].im - v[9].re)};      // n1=1,k2=2: W^2
    v[13] = cmul_c(v[13], s1, -c1);                                 // n1=1,k2=3: W^3
    v[6] = {h * (v[6].re + v[6].im), h * (v[6].im - v[6].re)};      // n1=2,k2=1: W^2
    v[10] = cmul_negi(v[10]);                                       // n1=2,k2=2: W^4 = -i
    v[14] = {h * (v[14].im - v[14].re), -h * (v[14].re + v[14].im)};  // n1=2,k2=3: W^6
    v[7] = cmul_c(v[7], s1, -c1);                                   // n1=3,k2=1: W^3
    v[11] = {h * (v[11].im - v[11].re), -h * (v[11].re + v[11].im)};  // n1=3,k2=2: W^6
    v[15] = cmul_c(v[15], -c1, s1);                                 // n1=3,k2=3: W^9
}
template <int G>
TH_HD void dft16_tail(cf32 (&v)[16]) {
    fft4(v[4 * G], v[4 * G + 1], v[4 * G + 2], v[4 * G + 3]);
}
TH_HD void dft16(cf32 (&v)[16]) {
    dft16_head(v);
    dft16_tail<0>(v);
    dft16_tail<1>(v);
    dft16_tail<2>(v);
    dft16_tail<3>(v);
}
// natural output X[k], k = 4*k1 + k2, sits in slot k1 + 4*k2
TH_HD constexpr int dft16_slot(int k) { return (k >> 2) + 4 * (k & 3); }

template <int R>
struct RegDft;
template <>
struct RegDft<2> {
    static TH_HD void run(cf32 (&v)[2]) { fft2(v[0], v[1]); }
    static TH_HD constexpr int slot(int k) { return k; }
};
template <>
struct RegDft<4> {
    static TH_HD void run(cf32 (&v)[4]) { fft4(v[0], v[1], v[2], v[3]); }
    static TH_HD constexpr int slot(int k) { return k; }
};
template <>
struct RegDft<8> {
    static TH_HD void run(cf32 (&v)[8]) { dft8(v); }
    static TH_HD constexpr int slot(int k) { return dft8_slot(k); }
};
template <>
struct RegDft<16> {
    static TH_HD void run(cf32 (&v)[16]) { dft16(v); }
    static TH_HD constexpr int slot(int k) { return dft16_slot(k); }
};

// ---------------------------------------------------------------------------------------------
// Static description of the wave FFT for one n_fft.
// ---------------------------------------------------------------------------------------------
template <int LOG2_NC>
struct WaveFftCfg;
template <>
struct WaveFftCfg<9> {  // n_fft = 1024: Nc = 512 = 8 * 8 * 8, P = 8
    static constexpr int R1 = 8, R2 = 8, R3 = 8;
};
template <>
struct WaveFftCfg<10> {  // n_fft = 2048: Nc = 1024 = 16 * 16 * 4, P = 16
    static constexpr int R1 = 16, R2 = 16, R3 = 4;
};
template <>
struct WaveFftCfg<11> {  // n_fft = 4096: Nc = 2048 = 16 * 16 * 8, P = 32
    static constexpr int R1 = 16, R2 = 16, R3 = 8;
};

// LDS layout of the first exchange (pass-1 output, Ns = 1): lane jj writes the R1 consecutive
// slots R1*jj + r.  Two pad slots per 32 (pad1(i) = i + 2*(i/32)) make both sides of the transpose
// conflict-free at the cheapest rate the LDS has (scripts/ubench/lds_rate.hip):
//   writes: hipcc pairs the stores of r, r+1 into ds_write2_b64 (16 B per lane, 16 lanes per LDS pass): lane jj
//          starts at dword 2*R1*jj + 4*(R1*jj/32), so the 16 lanes of a pass cover all 64 banks exactly once
//          (one pad slot per 32 is enough for 8-byte writes but would put the 16-byte pairs of lanes jj and jj + 2 on
//          overlapping banks);
//   reads (pass 2, i = lane + 64*m): a half wave reads 32 CONSECUTIVE slots = all 64 banks once, 2.2 cycles
//          per wave-instruction.  (A pad per R1 slots made a half wave span 33+ slots: 4 cycles.)
// Unlike an XOR swizzle every address is "per-lane base + compile-time immediate": writes
// base = pad1(R1*jj), imm = r (R1 divides 32, so R1*jj + r stays inside the 32-group); reads
// base = pad1(lane), imm = 68*m.  No address registers stay live across the frame loop.
TH_HD uint32_t pad1(uint32_t i) { return i + ((i >> 5) << 1); }

template <int LOG2_NC>
struct WaveFft {
    using Cfg = WaveFftCfg<LOG2_NC>;
    static constexpr int NC = 1 << LOG2_NC;
    static constexpr int P = NC / 64;
    static constexpr int R1 = Cfg::R1, R2 = Cfg::R2, R3 = Cfg::R3;
    static constexpr int NS2 = R1, NS3 = R1 * R2;
    static_assert(R1 * R2 * R3 == NC, "radix plan must multiply to Nc");
    static constexpr int B1 = P / R1, B2 = P / R2, B3 = P / R3;
    static_assert(B1 >= 1 && B2 >= 1 && B3 >= 1, "radix larger than points per lane");
    // Twiddle tables (LDS in the kernel): a lane reads W_{Ns*R}^{r*k} for its own k = jj mod Ns.
    //   t2[(r-1)*NS2 + k], k < NS2 (NS2 <= 64 so k does not depend on the butterfly index b)
    //   t3[(r-1)*NS3 + jj], jj < NS3 (last pass: k = jj)
    // The n_fft = 2048 plan (R1 = R2 = 16, R3 = 4, one radix-16 butterfly per lane and pass) runs its twiddled passes as
    // fused-multiply-add butterflies (bfly4_tw): pass 2 = 4 x 4 decimation in time with the pass twiddle w = W^k folded in,
    //     X[m' + 4 m''] = sum_b (w W16^m')^b W4^(b m'')  sum_a (w^4)^a W4^(a m') z[4 a + b],
    // i.e. four inner butterflies with t = w^4 and four outer ones with t = w W16^m' — 10 table entries per lane
    // (w^4, w^8, then t, t^2 for m' = 0..3) instead of 15, 192 FMAs instead of 60 + 160 operations.  Pass 3 needs (t, t^2)
    // per butterfly instead of (t, t^2, t^3).
    // n_fft = 4096 (R1 = R2 = 16, R3 = 8, two radix-16 butterflies per lane and pass) runs the same way; its radix-8 last
    // pass is two bfly4_tw (base t^2) and four bfly2_tw with t W8^m': table entries (t, t^2, t^4, t W8) instead of t .. t^7.
    // n_fft = 1024 (8 x 8 x 8, one radix-8 butterfly per lane and pass): both twiddled passes are that radix-8 FMA butterfly
    // (bfly8_tw), four constants each.
    static constexpr bool FMA8 = (R1 == 8 && R2 == 8 && R3 == 8 && P == 8);
    static constexpr bool FMA_TW = (R1 == 16 && R2 == 16 && ((P == 16 && R3 == 4) || (P == 32 && R3 == 8))) || FMA8;
    static constexpr int NT2 = FMA8 ? 4 : FMA_TW ? 10 : R2 - 1;       // pass-2 twiddles per lane
    static constexpr int NT3 = FMA_TW ? (R3 == 4 ? 2 : 4) : R3 - 1;   // pass-3 twiddles per butterfly
    static constexpr int T2_LEN = NT2 * NS2, T3_LEN = NT3 * NS3;
    static_assert(NS2 <= 64, "pass-2 twiddle index must be butterfly independent");
    // tw[i] = exp(-2 pi i * i / n_fft), n_fft = 2*NC  ->  W_{M}^{e} = tw[e * (2*NC / M)]
    // index into tw of pass-2 table entry e for twiddle index k
    static TH_HD uint32_t t2_index(uint32_t e, uint32_t k) {
        if constexpr (FMA8) {
            // t = W_{Ns2 R2}^k = tw[k S]: entries t, t^2, t^4, t W8
            constexpr uint32_t S = 2 * NC / (NS2 * R2), M = 2 * NC;
            return (e < 3 ? (1u << e) * k * S : k * S + M / 8) % M;
        } else if constexpr (FMA_TW) {
            // w = W_{Ns2 R2}^k = tw[k S], S = 2 Nc / (Ns2 R2);  W16 = tw[2 Nc / 16]
            // e = 0: w^4, 1: w^8, 2 + 2 m': t = w W16^m', 3 + 2 m': t^2
            constexpr uint32_t S = 2 * NC / (NS2 * R2), S16 = 2 * NC / 16, M = 2 * NC;
            const uint32_t mp = e >= 2 ? (e - 2) / 2 : 0, pw = e < 2 ? 4u * (e + 1u) : 1u + ((e - 2) & 1u);
            return (pw * k * S + (e >= 2 ? pw * mp * S16 : 0u)) % M;
        } else {
            return ((e + 1) * k) * (2 * NC / (NS2 * R2));
        }
    }
    // t2 == nullptr: the pass-2 constants are not kept in LDS (every lane holds its own in registers, load_t2_from_tw)
    static TH_HD void fill_tables(uint32_t tid, uint32_t nthr, const cf32 *tw, cf32 *t2, cf32 *t3) {
        if (t2 != nullptr)
            for (uint32_t i = tid; i < (uint32_t)T2_LEN; i += nthr) t2[i] = tw[t2_index(i / NS2, i % NS2)];
        for (uint32_t i = tid; i < (uint32_t)T3_LEN; i += nthr) {
            const uint32_t r = i / NS3 + 1, k = i % NS3;
            if constexpr (FMA_TW && R3 == 8) {
                // t = W_Nc^k = tw[2 k]: entries t, t^2, t^4, t W8
                const uint32_t e = i / NS3;
                t3[i] = tw[(e < 3 ? (2u << e) * k : 2u * k + (uint32_t)(2 * NC / 8)) % (uint32_t)(2 * NC)];
            } else {
                t3[i] = tw[(r * k) * (2 * NC / (NS3 * R3))];
            }
        }
    }

    // -----------------------------------------------------------------------------------------
    // Plane exchanges (n_fft = 2048: R1 = R2 = 16, one butterfly per lane and pass).  Both exchanges store with
    // ds_write_addtid_b32 (lds_st_planes4) into 32 dword planes — plane k / 16 + k = real / imaginary parts of
    // butterfly output k of every lane, in hardware-lane order — and read back with ds_read_b128, four consecutive
    // lanes' values of one plane per load.  For the reads to be consecutive the butterflies are dealt to the hardware
    // lanes in a permuted order:
    //   pass 1: lane l owns column lane_col(l) = 4 (l & 15) + (l >> 4) of the input (complex points col + 64 m): the
    //           global loads of a 16-lane group step by 32 bytes, the wave still covers whole 512-byte spans.
    //   pass 2: lane l owns butterfly j = 16 a + c, a = l & 3, c = l >> 2 (twiddle index k = c).  Its inputs
    //           in[j + 64 r] = pass-1 output c of column 4 r + a = plane c, lanes 16 a + r: 16 consecutive dwords.
    //   pass 3: natural (mirror-local pairs of lane l: butterflies A_q = 64 q + l, B_q = 256 - A_q).  Input
    //           in[jj + 256 r] = pass-2 output jj >> 4 of butterfly 16 r + (jj & 15) = plane jj >> 4, lanes
    //           4 (jj & 15) + r: one 16-byte load per butterfly and component.
    // Plane pitch: 68 dwords for exchange 1 (the 16-lane groups of a ds_read_b128 then cover all 64 banks: quad
    // index (c + 4 a + t) mod 16 is distinct over {0-3, 12-15, 20-27} etc.), 64 for exchange 2 (quad = l & 15 for the
    // A reads, -l & 15 for the B reads: distinct).  LDS-array cycles per frame and exchange: 64 (stores) + 32 (loads)
    // against 128 + 32 for the 16-byte stores of the slot layout below (which 2-way bank-conflict: writes are banked
    // mod 32 dwords), scripts/ubench/stft_skeleton.hip.
    // -----------------------------------------------------------------------------------------
    static constexpr bool PLANES = (R1 == 16 && R2 == 16 && P == 16);
    // -----------------------------------------------------------------------------------------
    // The same for n_fft = 4096 (P = 32: two radix-16 butterflies per lane in passes 1 and 2, radix 8 last): 32 + 32 planes
    // per exchange, plane 16 b + c = output c of a lane's butterfly b.
    //   pass 1: lane l owns column lane_col(l) = (l >> 3) + 8 (l & 7); butterfly b takes its points col + 64 (b + 2 r).
    //   pass 2: lane l owns butterflies j = 16 a + c, c = l >> 2 (twiddle index), a = 2 (l & 3) + b, b = 0, 1.  Inputs
    //           in[j + 128 r] = pass-1 output c of column a + 8 (r & 7), butterfly r >> 3 = plane 16 (r >> 3) + c, lanes
    //           8 a .. 8 a + 7: two 16-byte loads.  (a = 2 (l & 3) + b and not (l & 3) + 4 b: the quad index of a load is then
    //           17 c + 4 (l & 3) + const mod 16 — distinct over a 16-lane group, no bank conflicts at pitch 68.)
    //   pass 3: natural mirror-local pairs; in[jj + 256 r] = pass-2 output jj >> 4 of butterfly 16 r + (jj & 15), which
    //           lane 4 (jj & 15) + (r >> 1) computed as its butterfly r & 1 = plane 16 (r & 1) + (jj >> 4), lanes
    //           4 (jj & 15) .. + 3: one 16-byte load per butterfly, component and r parity; float offset 1024 (r & 1) + 4 jj.
    // -----------------------------------------------------------------------------------------
    static constexpr bool PLANES32 = (R1 == 16 && R2 == 16 && P == 32 && R3 == 8);
    // -----------------------------------------------------------------------------------------
    // And for n_fft = 1024 (P = 8: one radix-8 butterfly per lane and pass): 8 + 8 planes per exchange, pitch 68 for both.
    //   pass 1: lane l owns column lane_col(l) = (l >> 3) + 8 (l & 7) (points col + 64 m); output c -> plane c.
    //   pass 2: lane l owns butterfly j = 8 a + c, a = l & 7, c = l >> 3 (twiddle index k = c).  Inputs in[j + 64 r] =
    //           pass-1 output c of column a + 8 r = plane c, lanes 8 a .. 8 a + 7: two 16-byte loads per component (quad
    //           index 17 c + 2 a + t mod 16: c in {2 g, 2 g + 1} and a = 0..7 over a 16-lane group = all 16 quads).
    //           Output rr = out[64 a + c + 8 rr] -> plane rr, lane l.
    //   pass 3: natural, butterfly jj = lane (no mirror-local pairs with one butterfly per lane): in[jj + 64 r] = pass-2
    //           output jj >> 3 of butterfly a = r, c = jj & 7 = plane jj >> 3, lanes 8 (jj & 7) .. + 7: two 16-byte loads,
    //           quad index 17 (jj >> 3) + 2 (jj & 7) + t: conflict-free for the same reason.
    // The split exchange (Z through LDS for the mirror reads) keeps the slot layout.
    // -----------------------------------------------------------------------------------------
    static constexpr bool PLANES8 = FMA8;
    static constexpr bool ANYPLANES = PLANES || PLANES32;  // the radix-16 plane plans (twiddle index lane >> 2)
    static constexpr int NPL = PLANES32 ? 32 : PLANES8 ? 8 : 16;  // planes per component
    static constexpr int PITCH1 = 68, PITCH2 = PLANES8 ? 68 : 64;
    static TH_HD uint32_t lane_col(uint32_t lane) {
        return PLANES ? 4u * (lane & 15u) + (lane >> 4) : (PLANES32 || PLANES8) ? (lane >> 3) + 8u * (lane & 7u) : lane;
    }
    static TH_HD uint32_t t2_k(uint32_t lane) { return ANYPLANES ? lane >> 2 : PLANES8 ? lane >> 3 : lane & (NS2 - 1); }
    // natural outputs X0..X7 (in v[0..7]) of a lane's radix-8 butterfly -> planes 0..7 of either exchange
    template <int PITCH>
    static TH_HD void st_planes8(float *sf, uint32_t lane, const cf32 (&v)[8]) {
        lds_st_planes4<0, 8 * PITCH * 4, 4 * PITCH>(sf, lane, v[0], v[1], v[2], v[3]);
        lds_st_planes4<4 * PITCH * 4, 12 * PITCH * 4, 4 * PITCH>(sf, lane, v[4], v[5], v[6], v[7]);
    }
    // 8 consecutive dwords of plane `pl` (both components) starting at lane position 8 g -> z[0..7]
    template <int PITCH>
    static TH_HD void ld_planes8(const float *sf, uint32_t pl, uint32_t g, cf32 (&z)[8]) {
        const float *const p = sf + pl * (uint32_t)PITCH + 8u * g;
        const f32x4 r0 = lds_ld4(p), r1 = lds_ld4(p + 4), i0 = lds_ld4(p + 8 * PITCH), i1 = lds_ld4(p + 8 * PITCH + 4);
        z[0] = {r0.a, i0.a};
        z[1] = {r0.b, i0.b};
        z[2] = {r0.c, i0.c};
        z[3] = {r0.d, i0.d};
        z[4] = {r1.a, i1.a};
        z[5] = {r1.b, i1.b};
        z[6] = {r1.c, i1.c};
        z[7] = {r1.d, i1.d};
    }
    // twiddled radix-8 FMA butterfly, natural order in and out; w = (t, t^2, t^4, t W8):
    // X[m' + 4 m''] = Y0[m'] + (-1)^m'' (t W8^m') Y1[m'],  Yb[m'] = sum_a (t^2)^a W4^(a m') x[2 a + b]
    static TH_HD void bfly8_tw(cf32 (&v)[8], cf32 t, cf32 t2, cf32 t4, cf32 t8) {
        bfly4_tw(v[0], v[2], v[4], v[6], t2, t4);  // Y0[0..3] in v0, v4, v2, v6
        bfly4_tw(v[1], v[3], v[5], v[7], t2, t4);  // Y1[0..3] in v1, v5, v3, v7
        bfly2_tw(v[0], v[1], t.re, t.im);          // X0, X4
        bfly2_tw(v[4], v[5], t8.re, t8.im);        // X1, X5
        bfly2_tw(v[2], v[3], t.im, -t.re);         // X2, X6   (-i t)
        bfly2_tw(v[6], v[7], t8.im, -t8.re);       // X3, X7   (-i t W8)
        const cf32 x1 = v[4], x3 = v[6], x4 = v[1], x6 = v[3];
        v[1] = x1;
        v[3] = x3;
        v[4] = x4;
        v[6] = x6;
    }
    // outputs 4 i + G (i = 0..3) of a radix-16 butterfly (held in v0..v3) -> planes B0 + G + 4 i
    template <int PITCH, int B0, int G>
    static TH_HD void st_group4(float *slab, uint32_t lane, cf32 v0, cf32 v1, cf32 v2, cf32 v3) {
        lds_st_planes4<(B0 + G) * PITCH * 4, (NPL + B0 + G) * PITCH * 4, 16 * PITCH>(slab, lane, v0, v1, v2, v3);
    }
    template <int PITCH, int B0 = 0>
    static TH_HD void dft16_to_planes(uint32_t lane, cf32 (&v)[16], cf32 *slab) {
        float *const sf = reinterpret_cast<float *>(slab);
        dft16_head(v);
        dft16_tail<0>(v);
        st_group4<PITCH, B0, 0>(sf, lane, v[0], v[1], v[2], v[3]);
        dft16_tail<1>(v);
        st_group4<PITCH, B0, 1>(sf, lane, v[4], v[5], v[6], v[7]);
        dft16_tail<2>(v);
        st_group4<PITCH, B0, 2>(sf, lane, v[8], v[9], v[10], v[11]);
        dft16_tail<3>(v);
        st_group4<PITCH, B0, 3>(sf, lane, v[12], v[13], v[14], v[15]);
    }

    // pass 1 (Ns = 1, no twiddles): registers -> LDS slab
    static TH_HD void pass1(uint32_t lane, cf32 (&z)[P], cf32 *slab) {
        if constexpr (PLANES) {
            cf32 v[16];
            TH_UNROLL for (int r = 0; r < 16; r++) v[r] = z[r];
            dft16_to_planes<PITCH1>(lane, v, slab);
        } else if constexpr (PLANES32) {
            {
                cf32 v[16];
                TH_UNROLL for (int r = 0; r < 16; r++) v[r] = z[2 * r];
                dft16_to_planes<PITCH1, 0>(lane, v, slab);
            }
            {
                cf32 v[16];
                TH_UNROLL for (int r = 0; r < 16; r++) v[r] = z[2 * r + 1];
                dft16_to_planes<PITCH1, 16>(lane, v, slab);
            }
        } else if constexpr (PLANES8) {
            cf32 v[8], o[8];
            TH_UNROLL for (int r = 0; r < 8; r++) v[r] = z[r % P];
            dft8(v);
            TH_UNROLL for (int c = 0; c < 8; c++) o[c] = v[dft8_slot(c)];
            st_planes8<PITCH1>(reinterpret_cast<float *>(slab), lane, o);
        } else {
            TH_UNROLL for (int b = 0; b < B1; b++) {
                cf32 v[R1];
                TH_UNROLL for (int r = 0; r < R1; r++) v[r] = z[b + B1 * r];
                RegDft<R1>::run(v);
                const uint32_t jj = lane + 64u * b;
                TH_UNROLL for (int r = 0; r < R1; r++) lds_st(&slab[pad1(jj * R1) + r], v[RegDft<R1>::slot(r)]);  // = pad1(jj*R1 + r)
            }
        }
    }
    static TH_HD void read1(uint32_t lane, cf32 (&z)[P], const cf32 *slab) {
        if constexpr (PLANES) {
            const float *const sf = reinterpret_cast<const float *>(slab) + (lane >> 2) * PITCH1 + 16u * (lane & 3u);
            f32x4 re[4], im[4];
            TH_UNROLL for (int t = 0; t < 4; t++) {
                re[t] = lds_ld4(sf + 4 * t);
                im[t] = lds_ld4(sf + 16 * PITCH1 + 4 * t);
            }
            TH_UNROLL for (int t = 0; t < 4; t++) {
                z[4 * t] = {re[t].a, im[t].a};
                z[4 * t + 1] = {re[t].b, im[t].b};
                z[4 * t + 2] = {re[t].c, im[t].c};
                z[4 * t + 3] = {re[t].d, im[t].d};
            }
        } else if constexpr (PLANES32) {
            // input r of the lane's butterfly b -> z[b + 2 r]: plane 16 (r >> 3) + c, lane 16 (l & 3) + 8 b + (r & 7)
            const float *const sf = reinterpret_cast<const float *>(slab) + (lane >> 2) * PITCH1 + 16u * (lane & 3u);
            TH_UNROLL for (int b = 0; b < 2; b++) {
                f32x4 re[4], im[4];
                TH_UNROLL for (int h = 0; h < 4; h++) {  // h = 2 (r >> 3) + ((r & 7) >> 2)
                    re[h] = lds_ld4(sf + (16 * (h >> 1)) * PITCH1 + 8 * b + 4 * (h & 1));
                    im[h] = lds_ld4(sf + (NPL + 16 * (h >> 1)) * PITCH1 + 8 * b + 4 * (h & 1));
                }
                TH_UNROLL for (int h = 0; h < 4; h++) {
                    z[b + 2 * (4 * h)] = {re[h].a, im[h].a};
                    z[b + 2 * (4 * h + 1)] = {re[h].b, im[h].b};
                    z[b + 2 * (4 * h + 2)] = {re[h].c, im[h].c};
                    z[b + 2 * (4 * h + 3)] = {re[h].d, im[h].d};
                }
            }
        } else if constexpr (PLANES8) {
            cf32 v[8];
            ld_planes8<PITCH1>(reinterpret_cast<const float *>(slab), lane >> 3, lane & 7u, v);
            TH_UNROLL for (int r = 0; r < 8; r++) z[r % P] = v[r];
        } else {
            TH_UNROLL for (int m = 0; m < P; m++) z[m] = lds_ld(&slab[pad1(lane) + 68u * m]);  // = pad1(lane + 64*m)
        }
    }

    // pass 2 (Ns = R1): registers -> LDS slab (linear).  In three pieces so that the kernel can issue the
    // twiddle reads long before their use (LDS returns in order: a read issued next to its use exposes
    // the whole LDS latency): load_t2 -> pass2_twiddle -> pass2_dft.  pass2() is the composition.
    static TH_HD void load_t2_from_tw(uint32_t lane, cf32 (&w2)[NT2], const cf32 *tw) {  // straight from the global table
        const uint32_t k = t2_k(lane);
        TH_UNROLL for (int r = 0; r < NT2; r++) w2[r] = tw[t2_index((uint32_t)r, k)];
    }
    static TH_HD void load_t2(uint32_t lane, cf32 (&w2)[NT2], const cf32 *t2) {
        // NS2 <= 64: the same twiddles for every butterfly of the lane
        const uint32_t k = t2_k(lane);
        TH_UNROLL for (int r = 0; r < NT2; r++) w2[r] = lds_ld(&t2[r * NS2 + k]);
    }
    static TH_HD void pass2_twiddle(cf32 (&z)[P], const cf32 (&w2)[NT2]) {
        if constexpr (!FMA_TW) {  // (the FMA plan folds the twiddles into pass2_dft_tw)
            TH_UNROLL for (int b = 0; b < B2; b++)
                TH_UNROLL for (int r = 1; r < R2; r++) z[b + B2 * r] = cmul(z[b + B2 * r], w2[r - 1]);
        }
    }
    // FMA plan: twiddled radix-16 butterfly of pass 2 straight into the planes B0 .. B0 + 15 of exchange 2 (see FMA_TW above)
    template <int B0>
    static TH_HD void dft16_tw_to_planes(uint32_t lane, cf32 (&v)[16], const cf32 (&w)[NT2], cf32 *slab) {
        float *const sf = reinterpret_cast<float *>(slab);
        // inner butterflies over a (inputs v[b + 4 a]), t = w^4: y_b[m'] ends up in v[b + 4 pi(m')], pi = (0, 2, 1, 3)
        TH_UNROLL for (int b = 0; b < 4; b++) bfly4_tw(v[b], v[4 + b], v[8 + b], v[12 + b], w[0], w[1 % NT2]);
        // outer butterfly m' over b (inputs y_b[m']), t = w W16^m': X[m' + 4 m''], m'' = 0..3 -> planes m' + 4 m''
#define TH_OUTER(MP, PI)                                                                                          \
    bfly4_tw(v[4 * (PI)], v[4 * (PI) + 1], v[4 * (PI) + 2], v[4 * (PI) + 3], w[(2 + 2 * (MP)) % NT2], w[(3 + 2 * (MP)) % NT2]); \
    st_group4<PITCH2, B0, (MP)>(sf, lane, v[4 * (PI)], v[4 * (PI) + 2], v[4 * (PI) + 1], v[4 * (PI) + 3])
        TH_OUTER(0, 0);
        TH_OUTER(1, 2);
        TH_OUTER(2, 1);
        TH_OUTER(3, 3);
#undef TH_OUTER
    }
    static TH_HD void pass2_dft_tw(uint32_t lane, cf32 (&z)[P], const cf32 (&w)[NT2], cf32 *slab) {
        if constexpr (PLANES) {
            dft16_tw_to_planes<0>(lane, z, w, slab);
        } else if constexpr (PLANES32) {
            {
                cf32 v[16];
                TH_UNROLL for (int r = 0; r < 16; r++) v[r] = z[2 * r];
                dft16_tw_to_planes<0>(lane, v, w, slab);
            }
            {
                cf32 v[16];
                TH_UNROLL for (int r = 0; r < 16; r++) v[r] = z[2 * r + 1];
                dft16_tw_to_planes<16>(lane, v, w, slab);
            }
        } else if constexpr (PLANES8) {
            cf32 v[8];
            TH_UNROLL for (int r = 0; r < 8; r++) v[r] = z[r % P];
            bfly8_tw(v, w[0], w[1 % NT2], w[2 % NT2], w[3 % NT2]);
            st_planes8<PITCH2>(reinterpret_cast<float *>(slab), lane, v);
        }
    }
    static TH_HD void pass2_dft(uint32_t lane, cf32 (&z)[P], cf32 *slab) {  // (plans without FMA_TW)
        TH_UNROLL for (int b = 0; b < B2; b++) {
            const uint32_t jj = lane + 64u * b, k = jj & (NS2 - 1);
            cf32 v[R2];
            TH_UNROLL for (int r = 0; r < R2; r++) v[r] = z[b + B2 * r];
            RegDft<R2>::run(v);
            const uint32_t j0 = (jj - k) * R2 + k;
            TH_UNROLL for (int r = 0; r < R2; r++) lds_st(&slab[j0 + r * NS2], v[RegDft<R2>::slot(r)]);
        }
    }
    // twiddles + butterflies + stores of pass 2 (either plan)
    static TH_HD void pass2_w(uint32_t lane, cf32 (&z)[P], const cf32 (&w2)[NT2], cf32 *slab) {
        if constexpr (FMA_TW) {
            pass2_dft_tw(lane, z, w2, slab);
        } else {
            pass2_twiddle(z, w2);
            pass2_dft(lane, z, slab);
        }
    }
    static TH_HD void pass2(uint32_t lane, cf32 (&z)[P], const cf32 *t2, cf32 *slab) {
        cf32 w2[NT2];
        load_t2(lane, w2, t2);
        pass2_w(lane, z, w2, slab);
    }
    static_assert(32 % R1 == 0, "pad1 needs R1 | 32");
    static constexpr int SLAB_LEN = NC + NC / 16;  // padded pass-1 image is the largest (>= NC + 1)
    static_assert(!(ANYPLANES || PLANES8) || 2 * SLAB_LEN >= 2 * NPL * PITCH1, "slab holds the planes of exchange 1");

    // -----------------------------------------------------------------------------------------
    // n_fft = 1024 (one last-pass butterfly per lane: no mirror-local pairs): the split pass without a third exchange.
    // The wave's halves trade registers instead (v_permlane32_swap, no LDS).  Lane l = 1..31 runs last-pass butterfly l,
    // z[m] = Z[l + 64 m]; lane 32 + l runs butterfly 64 - l, whose outputs are the mirror partners, in the order
    // z[m] = conj Z[Nc - l - 64 m]:  W_Nc^((64 - l) s) = W8^s conj(t)^s with t = W_Nc^l, so
    //     Z[64 - l + 64 r] = Y[(r + 1) & 7],  Y[q] = sum_s x_s conj(t)^s W8^(s q) = conj( sum_s conj(x_s) t^s W8^(-s q) ),
    // i.e. the SAME butterfly with the same twiddles on the conjugated inputs yields conj Z[Nc - l - 64 m] in slot m.
    // Swapping (z[i] of the upper half) with (z[7 - i] of the lower half), i < 4, leaves every lane with four pairs
    // (z[i], z[7 - i]) = (Z[k], conj Z[Nc - k]):  k = l + 64 i in lane l,  k = l + 64 (7 - i) in lane 32 + l.
    // Lanes 0 and 32 run the self-mirrored butterflies 0 and 32 (plain) and sit the swap out (EXEC): lane 32 already holds
    // its pairs as (z[i], z[7 - i]), k = 32 + 64 i; lane 0 moves z[5], z[6], z[7], z[0] into z[4..7] — pairs k = 64 i, the
    // first one (Z[0], Z[0]) = bins 0 and Nc — and computes bin Nc/2 from Z[256] on top.
    // Stores: every pair yields X[k] and X[Nc - k].  Store S1_i takes X[k] from the lower half (and lane 32) and X[Nc - k]
    // from the upper half: bins jr + 64 i = the 64 consecutive bins from 64 i on; store S2_i takes the other output:
    // bins 448 - 64 i .. 511 - 64 i, where lane 0 supplies bin 448 - 64 i (X[Nc - k] of its NEXT pair; bin 256 for i = 3).
    // Both are whole 256-byte spans per instruction, "per-lane base + immediate"; bin Nc is lane 0's own ninth store.
    // (Each lane storing its own two outputs per pair touched 3-4 lines per instruction: 27 % more L2 requests and a
    // slower kernel than with the third exchange.)
    // -----------------------------------------------------------------------------------------
    static constexpr bool SWAP8 = PLANES8;
    struct Swap8Lane {
        uint32_t jr, jt;   // butterfly whose inputs the lane reads; twiddle index
        uint32_t cj, sp;   // sign-bit masks: conjugate the inputs (upper half) / the partner of the unswapped lanes 0, 32
        uint32_t kb;       // bin of pair i: kb + i ks
        int32_t ks;
        uint32_t b1, b2;   // store S1_i -> bin b1 + 64 i, store S2_i -> bin b2 + 64 (3 - i)
        bool l0, hi, swaps;
    };
    static TH_HD Swap8Lane swap8_lane(uint32_t lane) {
        const uint32_t l = lane & 63u;
        Swap8Lane s;
        s.hi = l > 32u;
        s.jr = s.hi ? 96u - l : l;
        s.jt = s.hi ? l - 32u : l;
        s.cj = s.hi ? 0x80000000u : 0u;
        s.swaps = (l & 31u) != 0u;
        s.sp = s.swaps ? 0u : 0x80000000u;
        s.l0 = l == 0u;
        s.kb = s.hi ? l + 416u : l;  // upper half: (l - 32) + 64 * 7
        s.ks = s.hi ? -64 : 64;
        s.b1 = s.jr;
        s.b2 = s.l0 ? 256u : (s.hi ? l + 224u : 320u - l);
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(s.b1), "+v"(s.b2));  // opaque: the stores stay "base + immediate" (see split_base)
#endif
        return s;
    }
    static TH_HD float flip_sign(float v, uint32_t mask) {
        uint32_t u;
        __builtin_memcpy(&u, &v, 4);
        u ^= mask;
        __builtin_memcpy(&v, &u, 4);
        return v;
    }
    static TH_HD void read2_sw(const Swap8Lane &s, cf32 (&z)[P], const cf32 *slab) {
        cf32 v[8];
        ld_planes8<PITCH2>(reinterpret_cast<const float *>(slab), s.jr >> 3, s.jr & 7u, v);
        TH_UNROLL for (int r = 0; r < 8; r++) z[r % P] = v[r];
    }
    static TH_HD void load_t3_sw(const Swap8Lane &s, cf32 (&w)[4], const cf32 *t3) {
        TH_UNROLL for (int e = 0; e < 4; e++) w[e] = lds_ld(&t3[(e % NT3) * NS3 + s.jt]);
    }
    // last pass; z256 <- Z[Nc/2] (lane 0)
    static TH_HD void pass3_sw(const Swap8Lane &s, cf32 (&z)[P], const cf32 (&w)[4], cf32 &z256) {
        cf32 v[8];
        TH_UNROLL for (int r = 0; r < 8; r++) v[r] = {z[r % P].re, flip_sign(z[r % P].im, s.cj)};
        bfly8_tw(v, w[0], w[1], w[2], w[3]);
        z256 = v[4];
        TH_UNROLL for (int m = 4; m < 8; m++) {  // lane 0: z[4..7] <- Z[320], Z[384], Z[448], Z[0]
            v[m].re = s.l0 ? v[(m + 1) & 7].re : v[m].re;
            v[m].im = s.l0 ? v[(m + 1) & 7].im : v[m].im;
        }
        TH_UNROLL for (int r = 0; r < 8; r++) z[r % P] = v[r];
    }
#if defined(__HIP_DEVICE_COMPILE__)
    // lanes 32..63 of a <-> lanes 0..31 of b
    static __device__ __forceinline__ void swap_halves(float &a, float &b) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        a = __uint_as_float(r[0]);
        b = __uint_as_float(r[1]);
    }
    static __device__ __forceinline__ void mirror_swap(const Swap8Lane &s, cf32 (&z)[P]) {
        if (s.swaps) {
            TH_UNROLL for (int i = 0; i < 4; i++) {
                swap_halves(z[i % P].re, z[(7 - i) % P].re);
                swap_halves(z[i % P].im, z[(7 - i) % P].im);
            }
        }
    }
#else
    // (host pass of the kernel source; the CPU emulator swaps at wave level, tests/emu/emu_stft.cpp)
    static void mirror_swap(const Swap8Lane &, cf32 (&)[P]) {}
#endif
    // emit(base, constant, |X[base + constant]|^2): every bin 0..Nc exactly once
    template <class Emit>
    static TH_HD void split_sw(const Swap8Lane &s, const cf32 (&z)[P], cf32 z256, const cf32 *stw, Emit emit) {
        cf32 w[4];
        TH_UNROLL for (int i = 0; i < 4; i++) w[i] = lds_ld(&stw[s.kb + (uint32_t)(i * s.ks)]);
        float px[4], py[4];
        TH_UNROLL for (int i = 0; i < 4; i++) {
            const cf32 zk = z[i % P], c = {z[(7 - i) % P].re, flip_sign(z[(7 - i) % P].im, s.sp)};  // c = conj Z[Nc - k]
            const float er = zk.re + c.re, ei = zk.im + c.im, dr = zk.re - c.re, di = zk.im - c.im;
            const float xr = th_fma(di, w[i].re, th_fma(dr, w[i].im, er)), xi = th_fma(di, w[i].im, th_fma(-dr, w[i].re, ei));
            const float yr = th_fma(2.0f, er, -xr), yi = th_fma(2.0f, ei, -xi);
            px[i] = xr * xr + xi * xi;
            py[i] = yr * yr + yi * yi;
        }
        const float p256 = 4.0f * (z256.re * z256.re + z256.im * z256.im);  // Z[Nc/2] is its own partner, W^(Nc/2) = -i
        TH_UNROLL for (int i = 0; i < 4; i++) {
            const float p1 = s.hi ? py[i] : px[i];
            const float p2 = s.l0 ? (i < 3 ? py[(i + 1) & 3] : p256) : (s.hi ? px[i] : py[i]);
            emit(s.b1, 64 * i, p1);
            emit(s.b2, 64 * (3 - i), p2);
        }
        if (s.l0) emit((uint32_t)NC, 0, py[0]);
    }

    // -----------------------------------------------------------------------------------------
    // Mirror-local last pass (used when B3 is even).  The real-FFT split pass needs Z[k] together
    // with Z[Nc - k].  Output k = jj + r*Ns3 of butterfly jj mirrors to output R3-1-r of butterfly
    // Ns3 - jj, so a lane that owns both butterflies of such a pair has every partner in its own
    // registers and the third LDS exchange disappears.  Lane l owns the pairs (A_q, B_q), q < B3/2:
    //     A_q = 64*q + l,   B_q = Ns3 - A_q
    // except lane 0, q = 0, which owns the two self-mirrored butterflies A = 0 and B = Ns3/2
    // (their outputs pair up inside each butterfly; k = 0 also yields the Nyquist bin Nc).
    // -----------------------------------------------------------------------------------------
    static constexpr bool PAIRED = (B3 % 2 == 0);
    static constexpr int NQ = PAIRED ? B3 / 2 : 1;
    static TH_HD uint32_t jj_a(uint32_t lane, int q) { return 64u * q + lane; }
    static TH_HD uint32_t jj_b(uint32_t lane, int q) {
        return (q == 0 && lane == 0) ? (uint32_t)NS3 / 2 : (uint32_t)NS3 - 64u * q - lane;
    }
    // The same butterfly indices as "per-lane base + compile-time constant" (every LDS address of the last pass is then
    // one shift of a base plus an immediate offset): A_q = a + 64 q;  B_q = b_last + 64 (NQ - 1 - q) for q > 0, B_0 = b0.
    struct PairBase {
        uint32_t a, b0, b_last;
    };
    static TH_HD PairBase pair_base(uint32_t lane) {
        const uint32_t l6 = lane & 63u;
        PairBase pb;
        pb.a = l6;
        pb.b_last = (l6 ^ 63u) + (uint32_t)(NS3 - 64 * (NQ - 1) - 63);  // Ns3 - 64 (NQ - 1) - lane
        pb.b0 = l6 == 0 ? (uint32_t)NS3 / 2 : pb.b_last + 64u * (NQ - 1);
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(pb.b0), "+v"(pb.b_last));  // opaque: keep "base + immediate" (see split_base)
#endif
        return pb;
    }
    static TH_HD uint32_t pb_b(const PairBase &pb, int q, uint32_t scale, uint32_t extra) {  // scale * B_q + extra
        return q == 0 ? pb.b0 * scale + extra : pb.b_last * scale + (scale * 64u * (uint32_t)(NQ - 1 - q) + extra);
    }
    // exchange-2 read in the paired layout: za[q][r] = in[A_q + r*Ns3], zb[q][r] = in[B_q + r*Ns3]
    static TH_HD void read2_paired(uint32_t lane, cf32 (&za)[NQ][R3], cf32 (&zb)[NQ][R3], const cf32 *slab) {
        read2_paired(lane, pair_base(lane), za, zb, slab);
    }
    static TH_HD void read2_paired(uint32_t lane, const PairBase &pbs, cf32 (&za)[NQ][R3], cf32 (&zb)[NQ][R3], const cf32 *slab) {
        if constexpr (PLANES) {
            static_assert(!PLANES || R3 == 4, "one 16-byte load per butterfly and component");
            static_assert(!PLANES || PITCH2 == 64, "plane (j >> 4) + quad (j & 15) of butterfly j = float offset 4 j");
            const float *const sf = reinterpret_cast<const float *>(slab);
            f32x4 ar[NQ], ai[NQ], br[NQ], bi[NQ];
            TH_UNROLL for (int q = 0; q < NQ; q++) {
                const float *const pa = sf + (pbs.a * 4u + 256u * (uint32_t)q), *const pb = sf + pb_b(pbs, q, 4u, 0u);
                ar[q] = lds_ld4(pa);
                ai[q] = lds_ld4(pa + 16 * PITCH2);
                br[q] = lds_ld4(pb);
                bi[q] = lds_ld4(pb + 16 * PITCH2);
            }
            TH_UNROLL for (int q = 0; q < NQ; q++) {
                za[q][0] = {ar[q].a, ai[q].a};
                za[q][1 % R3] = {ar[q].b, ai[q].b};
                za[q][2 % R3] = {ar[q].c, ai[q].c};
                za[q][3 % R3] = {ar[q].d, ai[q].d};
                zb[q][0] = {br[q].a, bi[q].a};
                zb[q][1 % R3] = {br[q].b, bi[q].b};
                zb[q][2 % R3] = {br[q].c, bi[q].c};
                zb[q][3 % R3] = {br[q].d, bi[q].d};
            }
        } else if constexpr (PLANES32) {
            // input r of butterfly jj: component planes 16 (r & 1) + (jj >> 4), quad r >> 1 of the 16 bytes at float 4 jj
            const float *const sf = reinterpret_cast<const float *>(slab);
            TH_UNROLL for (int q = 0; q < NQ; q++) {
                const float *const pa = sf + (pbs.a * 4u + 256u * (uint32_t)q), *const pb = sf + pb_b(pbs, q, 4u, 0u);
                f32x4 ar[2], ai[2], br[2], bi[2];
                TH_UNROLL for (int b = 0; b < 2; b++) {
                    ar[b] = lds_ld4(pa + 16 * b * PITCH2);
                    ai[b] = lds_ld4(pa + (NPL + 16 * b) * PITCH2);
                    br[b] = lds_ld4(pb + 16 * b * PITCH2);
                    bi[b] = lds_ld4(pb + (NPL + 16 * b) * PITCH2);
                }
                TH_UNROLL for (int b = 0; b < 2; b++) {
                    za[q][(0 + b) % R3] = {ar[b].a, ai[b].a};
                    za[q][(2 + b) % R3] = {ar[b].b, ai[b].b};
                    za[q][(4 + b) % R3] = {ar[b].c, ai[b].c};
                    za[q][(6 + b) % R3] = {ar[b].d, ai[b].d};
                    zb[q][(0 + b) % R3] = {br[b].a, bi[b].a};
                    zb[q][(2 + b) % R3] = {br[b].b, bi[b].b};
                    zb[q][(4 + b) % R3] = {br[b].c, bi[b].c};
                    zb[q][(6 + b) % R3] = {br[b].d, bi[b].d};
                }
            }
        } else {
            TH_UNROLL for (int q = 0; q < NQ; q++) {
                const uint32_t a = jj_a(lane, q), b = jj_b(lane, q);
                TH_UNROLL for (int r = 0; r < R3; r++) {
                    za[q][r] = lds_ld(&slab[a + (uint32_t)r * NS3]);
                    zb[q][r] = lds_ld(&slab[b + (uint32_t)r * NS3]);
                }
            }
        }
    }
    // last-pass twiddles of the lane's butterflies: wa[q][r-1] = W^(r*A_q), wb[q][r-1] = W^(r*B_q)
    static TH_HD void load_t3_paired(uint32_t lane, cf32 (&wa)[NQ][NT3], cf32 (&wb)[NQ][NT3], const cf32 *t3) {
        load_t3_paired(pair_base(lane), wa, wb, t3);
    }
    static TH_HD void load_t3_paired(const PairBase &pbs, cf32 (&wa)[NQ][NT3], cf32 (&wb)[NQ][NT3], const cf32 *t3) {
        TH_UNROLL for (int q = 0; q < NQ; q++) {
            TH_UNROLL for (int r = 1; r <= NT3; r++) {
                wa[q][r - 1] = lds_ld(&t3[pbs.a + (uint32_t)((r - 1) * NS3 + 64 * q)]);
                wb[q][r - 1] = lds_ld(&t3[pb_b(pbs, q, 1u, (uint32_t)((r - 1) * NS3))]);
            }
        }
    }
    static TH_HD void bfly3(cf32 (&v)[R3], const cf32 (&w3)[NT3]) {
        if constexpr (FMA_TW && R3 == 4) {  // (t, t^2): a = X0, c = X1, b = X2, d = X3
            bfly4_tw(v[0], v[1 % R3], v[2 % R3], v[3 % R3], w3[0], w3[1 % NT3]);
            const cf32 x1 = v[2 % R3];
            v[2 % R3] = v[1 % R3];
            v[1 % R3] = x1;
            return;
        }
        if constexpr (FMA_TW && R3 == 8) {
            // X[m' + 4 m''] = Y0[m'] + (-1)^m'' (t W8^m') Y1[m'],  Yb[m'] = sum_a (t^2)^a W4^(a m') x[2 a + b];  w3 = (t, t^2, t^4, t W8)
            const cf32 t = w3[0], t2 = w3[1 % NT3], t4 = w3[2 % NT3], t8 = w3[3 % NT3];
            bfly4_tw(v[0], v[2 % R3], v[4 % R3], v[6 % R3], t2, t4);  // Y0[0..3] in v0, v4, v2, v6
            bfly4_tw(v[1 % R3], v[3 % R3], v[5 % R3], v[7 % R3], t2, t4);  // Y1[0..3] in v1, v5, v3, v7
            bfly2_tw(v[0], v[1 % R3], t.re, t.im);           // X0, X4
            bfly2_tw(v[4 % R3], v[5 % R3], t8.re, t8.im);    // X1, X5
            bfly2_tw(v[2 % R3], v[3 % R3], t.im, -t.re);     // X2, X6   (-i t)
            bfly2_tw(v[6 % R3], v[7 % R3], t8.im, -t8.re);   // X3, X7   (-i t W8)
            const cf32 x1 = v[4 % R3], x3 = v[6 % R3], x4 = v[1 % R3], x6 = v[3 % R3];
            v[1 % R3] = x1;
            v[3 % R3] = x3;
            v[4 % R3] = x4;
            v[6 % R3] = x6;
            return;
        }
        cf32 w[R3];
        w[0] = v[0];
        TH_UNROLL for (int r = 1; r < R3; r++) w[r] = cmul(v[r], w3[(r - 1) % NT3]);
        RegDft<R3>::run(w);
        TH_UNROLL for (int r = 0; r < R3; r++) v[r] = w[RegDft<R3>::slot(r)];
    }
    // last pass: za[q][r] <- Z[A_q + r*Ns3], zb[q][r] <- Z[B_q + r*Ns3]
    static TH_HD void pass3_paired_w(cf32 (&za)[NQ][R3], cf32 (&zb)[NQ][R3], const cf32 (&wa)[NQ][NT3],
                                     const cf32 (&wb)[NQ][NT3]) {
        TH_UNROLL for (int q = 0; q < NQ; q++) {
            bfly3(za[q], wa[q]);
            bfly3(zb[q], wb[q]);
        }
    }
    static TH_HD void pass3_paired(uint32_t lane, cf32 (&za)[NQ][R3], cf32 (&zb)[NQ][R3], const cf32 *t3) {
        cf32 wa[NQ][NT3], wb[NQ][NT3];
        load_t3_paired(lane, wa, wb, t3);
        pass3_paired_w(za, zb, wa, wb);
    }
    // bin index of pair (q, s) of this lane.  Lane 0, q = 0 pairs inside butterfly 0 (s < R3/2: k = s*Ns3, as
    // for every other lane) and inside butterfly Ns3/2 (s >= R3/2: k = Ns3/2 + (s - R3/2)*Ns3, i.e. the
    // generic k minus (R3-1)*Ns3/2): one adjusted per-lane base keeps every k "base + immediate".
    static TH_HD int32_t split_k(uint32_t lane, int q, int s) {
        const int32_t lane_lo = (int32_t)lane, lane_hi = lane == 0 ? -(int32_t)((R3 - 1) * NS3 / 2) : lane_lo;
        return (q == 0 && s >= R3 / 2 ? lane_hi : lane_lo) + 64 * q + s * NS3;
    }
    // Output addressing of the split pass.  Every bin index a lane emits is "per-lane base + compile-time constant", and
    // the bases are built so that the compiler can prove their range (a 6-bit lane, xor / select / add of small
    // constants): row[base + C] then becomes one store with an immediate offset off a base computed once per frame.
    // (Written as row[k] / row[Nc - k] with k = lane + C, every mirrored store cost an or + sub + shift: 45 of the
    // kernel's 760 VALU instructions per frame.)
    //   pair (q, s), s <  R3/2 or q > 0:  k = lo + C,   Nc - k = mlo + (CMAX - C),   C = 64 q + s Ns3
    //   pair (0, s), s >= R3/2         :  k = hi + C',  Nc - k = mhi + (Ns3 (R3/2 - 1) - C'),  C' = (s - R3/2) Ns3
    // lo = lane, mlo = Nc - CMAX - lane; hi / mhi are lo / mlo shifted by R3/2 Ns3 for every lane but 0, whose second
    // self-mirrored butterfly Ns3/2 starts at bin Ns3/2.
    static constexpr int CMAX = 64 * (NQ - 1) + (R3 - 1) * NS3;
    struct SplitBase {
        uint32_t lo, hi, mlo, mhi;
    };
    static TH_HD SplitBase split_base(uint32_t lane) {
        const uint32_t l6 = lane & 63u, rl = l6 ^ 63u;  // rl = 63 - lane
        SplitBase b;
        b.lo = l6;
        b.mlo = rl + (uint32_t)(NC - CMAX - 63);
        b.hi = l6 == 0 ? (uint32_t)NS3 / 2 : l6 + (uint32_t)(R3 / 2) * NS3;
        b.mhi = l6 == 0 ? (uint32_t)(NC - NS3 / 2 - (R3 / 2 - 1) * NS3) : rl + (uint32_t)(NC - (R3 - 1) * NS3 - 63);
#if defined(__HIP_DEVICE_COMPILE__)
        // opaque: the optimiser would otherwise fold the constants back in ("0x400 - (lane | C)": one or + sub + shift per store)
        asm volatile("" : "+v"(b.hi), "+v"(b.mlo), "+v"(b.mhi));
#endif
        return b;
    }
    // split twiddles of the lane's pairs: ws[q][s] = stw[k(q, s)], ws_mid = stw[Nc/2] (only lane 0 uses it)
    static TH_HD void load_stw_paired(uint32_t lane, cf32 (&ws)[NQ][R3], const cf32 *stw) {
        const SplitBase sb = split_base(lane);  // = stw[split_k(lane, q, s)], as base + immediate
        TH_UNROLL for (int q = 0; q < NQ; q++)
            TH_UNROLL for (int s = 0; s < R3; s++)
                ws[q][s] = (q == 0 && s >= R3 / 2) ? lds_ld(&stw[sb.hi + (uint32_t)((s - R3 / 2) * NS3)])
                                                   : lds_ld(&stw[sb.lo + (uint32_t)(64 * q + s * NS3)]);
    }
    // w_mid = stw[Nc/2] (the twiddle of the self-mirrored bin; only lane 0 uses it)
    // emit(base, C, |X[base + C]|^2): once for every bin this lane owns
    template <class Emit>
    static TH_HD void split_paired_w(uint32_t lane, const cf32 (&za)[NQ][R3], const cf32 (&zb)[NQ][R3],
                                     const cf32 (&ws)[NQ][R3], cf32 w_mid, Emit emit) {
        const bool l0 = (lane & 63u) == 0;
        const SplitBase sb = split_base(lane);
        TH_UNROLL for (int q = 0; q < NQ; q++) {
            TH_UNROLL for (int s = 0; s < R3; s++) {
                cf32 zk = za[q][s], zm = zb[q][R3 - 1 - s];
                if (q == 0) {
                    const int rp = s - R3 / 2;
                    const cf32 zk0 = s < R3 / 2 ? za[0][s] : zb[0][rp];
                    const cf32 zm0 = s < R3 / 2 ? za[0][(R3 - s) % R3] : zb[0][R3 - 1 - rp];
                    if (s >= R3 / 2) {
                        zk.re = l0 ? zk0.re : zk.re;
                        zk.im = l0 ? zk0.im : zk.im;
                    }
                    zm.re = l0 ? zm0.re : zm.re;
                    zm.im = l0 ? zm0.im : zm.im;
                }
                const cf32 w = ws[q][s];
                // e = Z[k] + conj Z[Nc-k];  d = Z[k] - conj Z[Nc-k];  t = W^k * (-i d)
                // X[k] = e + t,  X[Nc-k] = conj(e - t)
                // (x = e + t in two FMAs per component, y = e - t = 2 e - x in one)
                const float er = zk.re + zm.re, ei = zk.im - zm.im;
                const float dr = zk.re - zm.re, di = zk.im + zm.im;
                const float xr = th_fma(di, w.re, th_fma(dr, w.im, er)), xi = th_fma(di, w.im, th_fma(-dr, w.re, ei));
                const float yr = th_fma(2.0f, er, -xr), yi = th_fma(2.0f, ei, -xi);
                if (q == 0 && s >= R3 / 2) {
                    emit(sb.hi, (s - R3 / 2) * NS3, xr * xr + xi * xi);
                    emit(sb.mhi, (R3 - 1 - s) * NS3, yr * yr + yi * yi);
                } else {
                    emit(sb.lo, 64 * q + s * NS3, xr * xr + xi * xi);
                    emit(sb.mlo, CMAX - (64 * q + s * NS3), yr * yr + yi * yi);
                }
            }
        }
        if (l0) {  // the self-mirrored bin Nc/2 = output R3/2 of butterfly 0
            const cf32 z = za[0][R3 / 2];
            const cf32 w = w_mid;
            const float er = 2.0f * z.re, di = 2.0f * z.im;  // zm = zk: e = (2 re, 0), d = (0, 2 im)
            const float xr = er + di * w.re, xi = di * w.im;
            emit((uint32_t)NC / 2, 0, xr * xr + xi * xi);
        }
    }
    // Split pass on lane-local pairs.  emit(base, C, |X[base + C]|^2) is called once for every bin this lane owns
    // (k in [0, Nc]; lane 0 owns 17 of the 1025 at Nc = 1024, every other lane 16).
    // stw[k] = W_{n_fft}^k = exp(-2 pi i k / (2 Nc)), k < Nc.
    template <class Emit>
    static TH_HD void split_paired(uint32_t lane, const cf32 (&za)[NQ][R3], const cf32 (&zb)[NQ][R3], const cf32 *stw,
                                   Emit emit) {
        cf32 ws[NQ][R3];
        load_stw_paired(lane, ws, stw);
        split_paired_w(lane, za, zb, ws, stw[NC / 2], emit);
    }
    // The bins of split_paired_w's emit calls, in its order: fn(base, C) once per call (the last one, bin Nc / 2, on lane 0 only).
    // The frame-pair mel epilogue keeps a frame's amplitudes in registers in emit order and lays them out as a row one frame later.
    // (n_fft 1024, split_sw: nine calls — S1_i and S2_i for i < 4, then bin Nc on lane 0)
    static constexpr int N_EMIT = PAIRED ? 2 * NQ * R3 + 1 : 9;
    template <class Fn>
    static TH_HD void split_enumerate(uint32_t lane, Fn fn) {
        if constexpr (!PAIRED) {
            const Swap8Lane s = swap8_lane(lane);
            TH_UNROLL for (int i = 0; i < 4; i++) {
                fn(s.b1, 64 * i);
                fn(s.b2, 64 * (3 - i));
            }
            if (s.l0) fn((uint32_t)NC, 0);
            return;
        }
        const bool l0 = (lane & 63u) == 0;
        const SplitBase sb = split_base(lane);
        TH_UNROLL for (int q = 0; q < NQ; q++) {
            TH_UNROLL for (int s = 0; s < R3; s++) {
                if (q == 0 && s >= R3 / 2) {
                    fn(sb.hi, (s - R3 / 2) * NS3);
                    fn(sb.mhi, (R3 - 1 - s) * NS3);
                } else {
                    fn(sb.lo, 64 * q + s * NS3);
                    fn(sb.mlo, CMAX - (64 * q + s * NS3));
                }
            }
        }
        if (l0) fn((uint32_t)NC / 2, 0);
    }

    // -----------------------------------------------------------------------------------------
    // Packed-f32 pipeline of the n_fft 2048 plane plan (round 4; arithmetic: stft_pk.h).  Same LDS layouts, same lane ->
    // butterfly maps and the same butterfly algebra as pass1 / read1 / pass2_w / read2_paired / pass3_paired_w /
    // split_paired_w above — the instruction stream is re-expressed on register PAIRS:
    //   pass 1   AoS (re, im): the windowed points as loaded (8-byte loads), un-twiddled 4 x 4 DFT-16  ->  planes
    //   pass 2   SoA: a ds_read_b128 returns one component of the points 4 t .. 4 t + 3 of the lane's butterfly = the pairs
    //            (4 t, 4 t + 1), (4 t + 2, 4 t + 3).  Inner butterflies (over a, one per b): b = 0, 1 and b = 2, 3 run as pairs
    //            (clean); outer butterfly m' takes y_b[m'], b = 0..3, = two pairs: first level clean, second level cross.
    //   pass 3   SoA: one ds_read_b128 = the four inputs of one radix-4 butterfly = two pairs: clean level, cross level;
    //            outputs as the pairs (X0, X2), (X1, X3).
    //   split    pairs of (Z[k], conj Z[Nc - k]) combinations: bins (s, s + 2) of butterfly pair q together; the split
    //            twiddles come from a table laid out for that (stwp: (w[k].re, w[k + 512].re), (w[k].im, w[k + 512].im)).
    // -----------------------------------------------------------------------------------------
    static constexpr bool PK = PLANES;
    // table of the split twiddles for the packed plan: entry e < 512: re2[e] = (w[e].re, w[e + 512].re), im2[e] likewise;
    // entries 512, 513: lane 0's two self-mirrored butterflies 0 and Ns3 / 2 = 128: (w[0], w[128]) and (w[256], w[384])
    static constexpr int STWP_N = 514;           // v2f entries per component
    static constexpr int STWP_LEN = 2 * STWP_N;  // in cf32-sized (8-byte) units: re2[514] then im2[514]
    static TH_HD void fill_stwp(uint32_t tid, uint32_t nthr, const cf32 *tw, cf32 *stwp) {
        for (uint32_t e = tid; e < (uint32_t)STWP_N; e += nthr) {
            const uint32_t k0 = e < 512u ? e : (e == 512u ? 0u : 256u), k1 = e < 512u ? e + 512u : (e == 512u ? 128u : 384u);
            stwp[e] = {tw[k0].re, tw[k1].re};
            stwp[STWP_N + e] = {tw[k0].im, tw[k1].im};
        }
    }
    template <int PITCH, int G>
    static TH_HD void st_group4_pk(float *sf, uint32_t lane, v2f v0, v2f v1, v2f v2, v2f v3) {  // AoS points -> planes G, G + 4, ..
        lds_st_planes4f<G * PITCH * 4, (NPL + G) * PITCH * 4, 16 * PITCH>(sf, lane, v0.x, v0.y, v1.x, v1.y, v2.x, v2.y, v3.x, v3.y);
    }
    static TH_HD void pass1_pk(uint32_t lane, v2f (&v)[16], cf32 *slab) {
        float *const sf = reinterpret_cast<float *>(slab);
        pk_dft16_head(v);
        pk_dft16_tail<0>(v);
        st_group4_pk<PITCH1, 0>(sf, lane, v[0], v[1], v[2], v[3]);
        pk_dft16_tail<1>(v);
        st_group4_pk<PITCH1, 1>(sf, lane, v[4], v[5], v[6], v[7]);
        pk_dft16_tail<2>(v);
        st_group4_pk<PITCH1, 2>(sf, lane, v[8], v[9], v[10], v[11]);
        pk_dft16_tail<3>(v);
        st_group4_pk<PITCH1, 3>(sf, lane, v[12], v[13], v[14], v[15]);
    }
    // zr[j], zi[j]: components of the points (2 j, 2 j + 1) of the lane's pass-2 butterfly
    static TH_HD void read1_pk(uint32_t lane, v2f (&zr)[8], v2f (&zi)[8], const cf32 *slab) {
        const float *const sf = reinterpret_cast<const float *>(slab) + (lane >> 2) * PITCH1 + 16u * (lane & 3u);
        v2x2 re[4], im[4];
        TH_UNROLL for (int t = 0; t < 4; t++) {
            re[t] = lds_ld4_pk(sf + 4 * t);
            im[t] = lds_ld4_pk(sf + 16 * PITCH1 + 4 * t);
        }
        TH_UNROLL for (int t = 0; t < 4; t++) {
            zr[2 * t] = re[t].lo;
            zr[2 * t + 1] = re[t].hi;
            zi[2 * t] = im[t].lo;
            zi[2 * t + 1] = im[t].hi;
        }
    }
    static TH_HD void load_t2_pk(uint32_t lane, v2f (&w2)[NT2], const cf32 *t2) {
        const uint32_t k = t2_k(lane);
        TH_UNROLL for (int r = 0; r < NT2; r++) w2[r] = lds_ld_pk(&t2[r * NS2 + k]);
    }
    // twiddled radix-16 butterfly of pass 2 (dft16_tw_to_planes) on pairs, into the planes of exchange 2
    static TH_HD void pass2_pk(uint32_t lane, v2f (&zr)[8], v2f (&zi)[8], const v2f (&w)[NT2], cf32 *slab) {
        float *const sf = reinterpret_cast<float *>(slab);
        // inner butterflies over a (inputs v[b + 4 a]), t = w^4: pairs b = (0, 1) -> j = 0, (2, 3) -> j = 1; v[b + 4 a] = pair 2 a + j.
        // y_b[m'] ends up in v[b + 4 pi(m')], pi = (0, 2, 1, 3)
        TH_UNROLL for (int j = 0; j < 2; j++)
            s_bfly4(zr[j], zi[j], zr[2 + j], zi[2 + j], zr[4 + j], zi[4 + j], zr[6 + j], zi[6 + j], w[0], w[1 % NT2]);
        // outer butterfly m' over b (inputs y_b[m'] = v[4 PI + b], PI = pi(m')): pairs p = 2 PI, q = 2 PI + 1; t = w W16^m'.
        // out p = (X[m'], X[m' + 8]), q = (X[m' + 4], X[m' + 12]) -> planes m' + 4 m''
#define TH_OUTER_PK(MP, PI)                                                                                               \
    x_bfly4(zr[2 * (PI)], zi[2 * (PI)], zr[2 * (PI) + 1], zi[2 * (PI) + 1], w[(2 + 2 * (MP)) % NT2], w[(3 + 2 * (MP)) % NT2]); \
    lds_st_planes4f<(MP) * PITCH2 * 4, (NPL + (MP)) * PITCH2 * 4, 16 * PITCH2>(                                           \
        sf, lane, zr[2 * (PI)].x, zi[2 * (PI)].x, zr[2 * (PI) + 1].x, zi[2 * (PI) + 1].x, zr[2 * (PI)].y, zi[2 * (PI)].y, \
        zr[2 * (PI) + 1].y, zi[2 * (PI) + 1].y)
        TH_OUTER_PK(0, 0);
        TH_OUTER_PK(1, 2);
        TH_OUTER_PK(2, 1);
        TH_OUTER_PK(3, 3);
#undef TH_OUTER_PK
    }
    // exchange-2 read, paired layout: butterfly A_q / B_q inputs (0, 1) -> p, (2, 3) -> q
    struct PkPairs {
        v2f pr, pi, qr, qi;
    };
    static TH_HD void read2_paired_pk(const PairBase &pbs, PkPairs (&za)[NQ], PkPairs (&zb)[NQ], const cf32 *slab) {
        static_assert(!PLANES || (R3 == 4 && PITCH2 == 64), "one 16-byte load per butterfly and component");
        const float *const sf = reinterpret_cast<const float *>(slab);
        v2x2 ar[NQ], ai[NQ], br[NQ], bi[NQ];
        TH_UNROLL for (int q = 0; q < NQ; q++) {
            const float *const pa = sf + (pbs.a * 4u + 256u * (uint32_t)q), *const pb = sf + pb_b(pbs, q, 4u, 0u);
            ar[q] = lds_ld4_pk(pa);
            ai[q] = lds_ld4_pk(pa + 16 * PITCH2);
            br[q] = lds_ld4_pk(pb);
            bi[q] = lds_ld4_pk(pb + 16 * PITCH2);
        }
        TH_UNROLL for (int q = 0; q < NQ; q++) {
            za[q] = {ar[q].lo, ai[q].lo, ar[q].hi, ai[q].hi};
            zb[q] = {br[q].lo, bi[q].lo, br[q].hi, bi[q].hi};
        }
    }
    static TH_HD void load_t3_paired_pk(const PairBase &pbs, v2f (&wa)[NQ][NT3], v2f (&wb)[NQ][NT3], const cf32 *t3) {
        TH_UNROLL for (int q = 0; q < NQ; q++) {
            TH_UNROLL for (int r = 1; r <= NT3; r++) {
                wa[q][r - 1] = lds_ld_pk(&t3[pbs.a + (uint32_t)((r - 1) * NS3 + 64 * q)]);
                wb[q][r - 1] = lds_ld_pk(&t3[pb_b(pbs, q, 1u, (uint32_t)((r - 1) * NS3))]);
            }
        }
    }
    // last pass: za[q] -> p = (Z[A_q], Z[A_q + 2 Ns3]), q = (Z[A_q + Ns3], Z[A_q + 3 Ns3]); zb likewise
    static TH_HD void pass3_paired_pk(PkPairs (&za)[NQ], PkPairs (&zb)[NQ], const v2f (&wa)[NQ][NT3], const v2f (&wb)[NQ][NT3]) {
        TH_UNROLL for (int q = 0; q < NQ; q++) {
            x_bfly4(za[q].pr, za[q].pi, za[q].qr, za[q].qi, wa[q][0], wa[q][1 % NT3]);
            x_bfly4(zb[q].pr, zb[q].pi, zb[q].qr, zb[q].qi, wb[q][0], wb[q][1 % NT3]);
        }
    }
    // split twiddle pairs of the lane: wr[q][h], wi[q][h] = (w[k(q, h)], w[k(q, h + 2)]) components, h = 0, 1
    static TH_HD void load_stw_paired_pk(uint32_t lane, v2f (&wr)[NQ][2], v2f (&wi)[NQ][2], const cf32 *stwp) {
        const uint32_t l6 = lane & 63u;
        uint32_t e00 = l6 == 0 ? 512u : l6, e01 = l6 == 0 ? 513u : l6 + (uint32_t)NS3;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(e00), "+v"(e01));  // opaque (see split_base)
#endif
        TH_UNROLL for (int q = 0; q < NQ; q++) {
            TH_UNROLL for (int h = 0; h < 2; h++) {
                const cf32 *const e = q == 0 ? stwp + (h == 0 ? e00 : e01) : stwp + (l6 + (uint32_t)(64 * q + h * NS3));
                wr[q][h] = lds_ld_pk(e);
                wi[q][h] = lds_ld_pk(e + STWP_N);
            }
        }
    }
    // Split pass on pairs (split_paired_w: the same bins, the same emit order of bases and constants).
    // Generic lane: pair h of butterfly pair q = bins s = h and s = h + 2:  zk = (za X_h, za X_{h+2}), zm = (zb X_{3-h}, zb X_{1-h}).
    // With za = (p, q) = ((X0, X2), (X1, X3)):  h = 0: zk = za.p, zm = swap(zb.q);  h = 1: zk = za.q, zm = swap(zb.p).
    // Lane 0, q = 0 pairs inside its two self-mirrored butterflies:  h = 0: zk = (A.X0, B.X0), zm = (A.X0, B.X3);
    //                                                               h = 1: zk = (A.X1, B.X1), zm = (A.X3, B.X2).
    template <class Emit>
    static TH_HD void split_paired_pk(uint32_t lane, const PkPairs (&za)[NQ], const PkPairs (&zb)[NQ], const v2f (&wr)[NQ][2],
                                      const v2f (&wi)[NQ][2], cf32 w_mid, Emit emit) {
        static_assert(R3 == 4, "pairs (s, s + 2)");
        const bool l0 = (lane & 63u) == 0;
        const SplitBase sb = split_base(lane);
        TH_UNROLL for (int q = 0; q < NQ; q++) {
            TH_UNROLL for (int h = 0; h < 2; h++) {
                const v2f zkr_g = h == 0 ? za[q].pr : za[q].qr, zki_g = h == 0 ? za[q].pi : za[q].qi;
                const v2f zmr_g = h == 0 ? zb[q].qr : zb[q].pr, zmi_g = h == 0 ? zb[q].qi : zb[q].pi;  // to be read swapped
                v2f er, ei, dr, di;
                if (q == 0) {
                    // lane 0: its own pairs (selects build the pairs in order, no operand swap)
                    const v2f ak = h == 0 ? za[0].pr : za[0].qr, aki = h == 0 ? za[0].pi : za[0].qi;   // (A.X_h, A.X_{h+2})
                    const v2f bk = h == 0 ? zb[0].pr : zb[0].qr, bki = h == 0 ? zb[0].pi : zb[0].qi;   // (B.X_h, B.X_{h+2})
                    const v2f bo = h == 0 ? zb[0].qr : zb[0].pr, boi = h == 0 ? zb[0].qi : zb[0].pi;   // (B.X_{1-h}', ..): (B.X1, B.X3) / (B.X0, B.X2)
                    // h = 0: zk = (A.X0, l0 ? B.X0 : A.X2), zm = (l0 ? A.X0 : B.X3, l0 ? B.X3 : B.X1)
                    // h = 1: zk = (A.X1, l0 ? B.X1 : A.X3), zm = (l0 ? A.X3 : B.X2, l0 ? B.X2 : B.X0)
                    const v2f zkr = mk2(ak.x, l0 ? bk.x : ak.y), zki = mk2(aki.x, l0 ? bki.x : aki.y);
                    const v2f zmr = h == 0 ? mk2(l0 ? ak.x : bo.y, l0 ? bo.y : bo.x) : mk2(l0 ? ak.y : bo.y, l0 ? bo.y : bo.x);
                    const v2f zmi = h == 0 ? mk2(l0 ? aki.x : boi.y, l0 ? boi.y : boi.x) : mk2(l0 ? aki.y : boi.y, l0 ? boi.y : boi.x);
                    er = zkr + zmr;
                    ei = zki - zmi;
                    dr = zkr - zmr;
                    di = zki + zmi;
                } else {
                    er = pk_add_sw(zkr_g, zmr_g);
                    ei = pk_sub_sw(zki_g, zmi_g);
                    dr = pk_sub_sw(zkr_g, zmr_g);
                    di = pk_add_sw(zki_g, zmi_g);
                }
                // e = Z[k] + conj Z[Nc-k];  d = Z[k] - conj Z[Nc-k];  t = W^k (-i d);  X[k] = e + t,  X[Nc-k] = conj(e - t)
                const v2f xr = pk_fma(di, wr[q][h], pk_fma(dr, wi[q][h], er));
                const v2f xi = pk_fma(di, wi[q][h], pk_fma(-dr, wr[q][h], ei));
                const v2f yr = pk_2x_minus(er, xr), yi = pk_2x_minus(ei, xi);
                const v2f px = pk_fma(xr, xr, xi * xi), py = pk_fma(yr, yr, yi * yi);
                TH_UNROLL for (int u = 0; u < 2; u++) {
                    const int s = h + 2 * u;
                    const float pxs = u == 0 ? px.x : px.y, pys = u == 0 ? py.x : py.y;
                    if (q == 0 && s >= R3 / 2) {
                        emit(sb.hi, (s - R3 / 2) * NS3, pxs);
                        emit(sb.mhi, (R3 - 1 - s) * NS3, pys);
                    } else {
                        emit(sb.lo, 64 * q + s * NS3, pxs);
                        emit(sb.mlo, CMAX - (64 * q + s * NS3), pys);
                    }
                }
            }
        }
        if (l0) {  // the self-mirrored bin Nc/2 = output R3/2 = X2 of butterfly 0
            const float zre = za[0].pr.y, zim = za[0].pi.y;
            const float er = 2.0f * zre, di = 2.0f * zim;
            const float xr = er + di * w_mid.re, xi = di * w_mid.im;
            emit((uint32_t)NC / 2, 0, xr * xr + xi * xi);
        }
    }
};

// |X[k]|^2 from the half-scaled packed spectrum (window pre-multiplied by 1/2):
//   e = Z[k] + conj Z[Nc-k],  o = -i (Z[k] - conj Z[Nc-k]),  X[k] = e + W^k o
TH_HD float split_power(cf32 zk, cf32 zm, cf32 w) {
    const float er = zk.re + zm.re, ei = zk.im - zm.im;
    const float dr = zk.re - zm.re, di = zk.im + zm.im;  // d = Z[k] - conj Z[Nc-k];  o = (di, -dr)
    const float xr = er + (di * w.re + dr * w.im);
    const float xi = ei + (di * w.im - dr * w.re);
    return xr * xr + xi * xi;
}

}  // namespace th

namespace th {

// ---------------------------------------------------------------------------------------------
// Frame load: lane j fetches x[m] = (fr[2n], fr[2n+1]), n = j + 64*m, of the frame's n_fft-sample
// span starting at signal position e0 = k*hop - win/2 - pad_left (stft.rs:137-146).  The zero
// padding of a window shorter than n_fft is applied by the window table (zeros there), so the
// load is the same for every window length: plain coalesced 8-byte loads, no predication.
// The wave kernel only takes INTERIOR frames (the whole n_fft span inside [0, n_samples)).  Frames
// that touch the signal boundaries (the first and last few per channel, and every frame of inputs
// shorter than n_fft) go to the generic kernel, which implements the reflect padding.
// M0: first slot to (re)load — 0 loads the whole frame, P - S only the S slots that are new after the
// previous frame's registers were moved down by S slots.
// ---------------------------------------------------------------------------------------------
template <int P, int M0, class WavPtr>
TH_HD void wave_fetch(uint32_t lane, cf32 (&x)[P], WavPtr wav, int64_t e0) {
    TH_UNROLL for (int m = M0; m < P; m++) {
        const WavPtr p = wav + (e0 + 2 * (int64_t)(lane + 64u * m));  // two adjacent dwords: one 8-byte load
#if defined(__HIP_DEVICE_COMPILE__) && defined(TH_STFT_NT) && (TH_STFT_NT & 2)
        x[m] = {__builtin_nontemporal_load(p), __builtin_nontemporal_load(p + 1)};
#else
        x[m] = {p[0], p[1]};
#endif
    }
}

// Boundary frames (part of the span outside [0, n_samples)): numpy-'reflect' indexing per sample (stft.rs:77-95 via
// utils.rs:111-138); the wave kernel takes them as one-frame chunks of channels with n_samples >= n_fft.  With at
// least n_fft samples a position is at most one reflection away (-(n-1) <= i <= 2(n-1)), so the general index with its
// modulo (reflect_index: 64-bit divisions, ~3000 instructions for the 32 samples of a lane) is not needed here.
TH_HD uint32_t reflect_once(int32_t i, int32_t n) {
    i = i < 0 ? -i : i;
    return (uint32_t)(i >= n ? 2 * (n - 1) - i : i);
}
template <int P, class WavPtr>
TH_HD void wave_fetch_reflect(uint32_t lane, cf32 (&x)[P], WavPtr wav, int64_t e0, uint32_t n_samples) {
    TH_UNROLL for (int m = 0; m < P; m++) {
        const int32_t i = (int32_t)e0 + 2 * (int32_t)(lane + 64u * m);
        x[m] = {wav[reflect_once(i, (int32_t)n_samples)], wav[reflect_once(i + 1, (int32_t)n_samples)]};
    }
}

// Rotated variants for the register-rotation frame loop: logical slot m lives in physical x[(m + OFF) % P].
// wave_fetch_rot loads the S newest logical slots P-S..P-1 of the NEXT frame (rotation OFF + S), which are the
// physical slots (OFF + i) % P that held the current frame's oldest S slots.
template <int P, int S, int OFF, class WavPtr>
TH_HD void wave_fetch_rot(uint32_t lane, cf32 (&x)[P], WavPtr wav, int64_t e0_next) {
    TH_UNROLL for (int i = 0; i < S; i++) {
        const WavPtr p = wav + (e0_next + 2 * (int64_t)(lane + 64u * (P - S + i)));
#if defined(__HIP_DEVICE_COMPILE__) && defined(TH_STFT_NT) && (TH_STFT_NT & 2)
        x[(OFF + i) % P] = {__builtin_nontemporal_load(p), __builtin_nontemporal_load(p + 1)};
#else
        x[(OFF + i) % P] = {p[0], p[1]};
#endif
    }
}
template <int P, int OFF>
TH_HD void wave_window_rot(uint32_t lane, cf32 (&z)[P], const cf32 (&x)[P], const cf32 *wtab) {
    TH_UNROLL for (int m = 0; m < P; m++) {
        const cf32 w = lds_ld(&wtab[lane + 64u * m]);
        const cf32 v = x[(m + OFF) % P];
        z[m] = {v.re * w.re, v.im * w.im};
    }
}

template <int P>
TH_HD void wave_window(uint32_t lane, cf32 (&z)[P], const cf32 (&x)[P], const cf32 *wtab) {
    TH_UNROLL for (int m = 0; m < P; m++) {
        const cf32 w = lds_ld(&wtab[lane + 64u * m]);
        z[m] = {x[m].re * w.re, x[m].im * w.im};
    }
}

// The wave kernel takes dB from the POWER |X|^2 (one v_log_f32, no square root).  Squaring halves the exponent range:
// with the plain window |X| < ~1e-19 would square to zero (-inf) where the reference's hypot -> log10
// (spectrogram.rs:200, decibel.rs:186-194) is still finite.  The kernel's window table therefore carries an exact factor
// 2^32 (host: WAVE_PRESCALE), every spectrum value is 2^32 too large, and the 20 log10(2^32) comes off in the same
// multiply-add that converts log2 to dB: finite down to |X| ~ 2.5e-29, overflow only beyond |X| ~ 4e9 (samples are <= 1).
// (WAVE_PRESCALE = 2^32 and WAVE_PRESCALE_DB = 20 log10(2^32) live in stft_core.h: the host builds the table with them)
TH_HD float fma_rn(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fmaf(a, b, c);
#else
    return __builtin_fmaf(a, b, c);
#endif
}
// 10*log10(p / 2^64) with p = |2^32 X|^2  ==  20*log10(|X|)  (decibel.rs:170-214 with amin = 0: p = +0 -> -inf)
TH_HD float power_to_dB(float p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fma_rn(__builtin_amdgcn_logf(p), 3.01029995663981195f, -WAVE_PRESCALE_DB);  // v_log_f32 (log2), <= 1 ulp
#else
    return fma_rn(__builtin_log2f(p), 3.01029995663981195f, -WAVE_PRESCALE_DB);
#endif
}

// 2^32 |X| from |2^32 X|^2 for the fused mel epilogue (the filterbank is linear: the factor comes off in amp_to_dB_fast)
TH_HD float power_to_amp_scaled(float p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(p);  // v_sqrt_f32, <= 1 ulp
#else
    return __builtin_sqrtf(p);
#endif
}
// |X| itself: the amplitude rows of the matrix-core mel path (the mel filterbank is applied to amplitudes, spectrogram.rs:200-207)
TH_HD float power_to_amp(float p) { return power_to_amp_scaled(p) * (1.0f / WAVE_PRESCALE); }

// 20*log10(a / 2^32) for a pre-scaled amplitude (the fused mel epilogue): a = +0 -> -inf like decibel.rs:189-193
TH_HD float amp_to_dB_fast(float a) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fma_rn(__builtin_amdgcn_logf(a), 6.02059991327962390f, -WAVE_PRESCALE_DB);
#else
    return fma_rn(__builtin_log2f(a), 6.02059991327962390f, -WAVE_PRESCALE_DB);
#endif
}

// ---------------------------------------------------------------------------------------------
// Fused mel epilogue (tables: mel_fuse.h).  The wave has written the frame's amplitudes |X[k]| to amp[0 .. n_freq)
// (floats, in its own exchange slab, which is free after the second exchange).  mel_pieces: lane + 64 slot owns
// one piece = 4 consecutive bins against (rise, fall) weight pairs -> partial sums (r, f) into prf[].  mel_gather:
// mel m = lane + 64 g adds the r of segment m's pieces and the f of segment m+1's (consecutive pieces) and hands
// the filter output to emit(m, value).  The caller synchronises the wave's LDS traffic between the two.
// ---------------------------------------------------------------------------------------------
struct MelFuseTab {
    const uint32_t *k0;    // [64 S]
    const cf32 *w;         // [256 S]  (rise, fall) of bin k0 + i at (slot * 4 + i) * 64 + lane
    const uint32_t *gat;   // [64 G]
    const uint32_t *gmax;  // [G]
    uint32_t S, G;
};
TH_HD MelFuseTab mel_fuse_view(const uint32_t *t, uint32_t S, uint32_t G) {
    MelFuseTab v;
    v.k0 = t;
    v.w = reinterpret_cast<const cf32 *>(t + 64u * S);
    v.gat = t + 64u * S + 512u * S;
    v.gmax = v.gat + 64u * G;
    v.S = S;
    v.G = G;
    return v;
}
TH_HD constexpr uint32_t mel_fuse_words(uint32_t S, uint32_t G) { return 64u * S + 512u * S + 64u * G + G + (G & 1u); }

// LDS returns in order and a read that is waited for right after its issue exposes the whole LDS latency, so both
// functions issue their reads in batches: all piece origins first, then two slots' weights and amplitudes at a time;
// the gather reads four (r, f) pairs per step.  S <= MEL_MAX_SLOTS and G <= MEL_MAX_GROUPS (the host checks).
constexpr uint32_t MEL_MAX_SLOTS = 8, MEL_MAX_GROUPS = 8;
TH_HD void mel_pieces(uint32_t lane, const float *amp, cf32 *prf, const MelFuseTab &t) {
    uint32_t k0[MEL_MAX_SLOTS];
    TH_UNROLL for (uint32_t s = 0; s < MEL_MAX_SLOTS; s++) k0[s] = t.k0[64u * (s < t.S ? s : t.S - 1u) + lane];
    TH_UNROLL for (uint32_t s2 = 0; s2 < MEL_MAX_SLOTS; s2 += 2) {
        if (s2 < t.S) {  // wave-uniform
            cf32 w[2][4];
            float a[2][4];
            uint32_t sj[2];
            TH_UNROLL for (uint32_t j = 0; j < 2; j++) {
                sj[j] = s2 + j < t.S ? s2 + j : t.S - 1u;  // odd S: the last slot is simply done twice
                TH_UNROLL for (uint32_t i = 0; i < 4; i++) w[j][i] = lds_ld(&t.w[(4u * sj[j] + i) * 64u + lane]);
            }
            TH_UNROLL for (uint32_t j = 0; j < 2; j++)
                TH_UNROLL for (uint32_t i = 0; i < 4; i++) a[j][i] = amp[k0[s2 + j] + i];
            TH_UNROLL for (uint32_t j = 0; j < 2; j++) {
                cf32 o;
                o.re = a[j][0] * w[j][0].re + a[j][1] * w[j][1].re + a[j][2] * w[j][2].re + a[j][3] * w[j][3].re;
                o.im = a[j][0] * w[j][0].im + a[j][1] * w[j][1].im + a[j][2] * w[j][2].im + a[j][3] * w[j][3].im;
                lds_st(&prf[64u * sj[j] + lane], o);
            }
        }
    }
}

template <class Emit>
TH_HD void mel_gather(uint32_t lane, const cf32 *prf, const MelFuseTab &t, Emit emit) {
    const uint32_t last = 64u * t.S - 1u;
    uint32_t pk[MEL_MAX_GROUPS];
    TH_UNROLL for (uint32_t g = 0; g < MEL_MAX_GROUPS; g++) pk[g] = t.gat[64u * (g < t.G ? g : t.G - 1u) + lane];
    TH_UNROLL for (uint32_t g = 0; g < MEL_MAX_GROUPS; g++) {
        if (g < t.G) {  // wave-uniform
            const uint32_t pb = pk[g] & 0xffffu, nr = (pk[g] >> 16) & 0xffu, nrf = nr + (pk[g] >> 24);
#if defined(__HIP_DEVICE_COMPILE__)
            const uint32_t n = __builtin_amdgcn_readfirstlane(t.gmax[g]);
#else
            const uint32_t n = t.gmax[g];
#endif
            float acc = 0.0f;
            for (uint32_t i = 0; i < n; i += 4) {
                cf32 v[4];
                TH_UNROLL for (uint32_t u = 0; u < 4; u++) {
                    const uint32_t at = pb + i + u;
                    v[u] = lds_ld(&prf[at < last ? at : last]);
                }
                TH_UNROLL for (uint32_t u = 0; u < 4; u++) acc += i + u < nr ? v[u].re : (i + u < nrf ? v[u].im : 0.0f);
            }
            emit(64u * g + lane, acc);
        }
    }
}

// Banded-sum mel epilogue (tables: build_mel_band, mel_fuse.h): lane = mel.  amp[0 .. n_freq) holds the frame's amplitudes and
// at least MEL_BAND_MAX_TAPS finite floats behind them.  Four taps per step: the reads of a step are issued together.
// (LDS pointers carry their address space explicitly here: the reads then are `ds_read … offset:imm` from one running base per
// array — written as amp[lo + t + u] the compiler formed every address with its own add, 8 of the 12 vector instructions per
// four taps — and eight taps are read before the first of their FMAs, so a step exposes one LDS latency, not two)
#if defined(__HIP_DEVICE_COMPILE__)
#define TH_LDS_F32 const __attribute__((address_space(3))) float
#define TH_LDS_F32_PTR(p) ((TH_LDS_F32 *)(p))
#else
#define TH_LDS_F32 const float
#define TH_LDS_F32_PTR(p) (p)
#endif
// off[g], n[g]: block offset and taps of group g (the table's header; the kernel gets them as scalar arguments)
// PAIRED: the table's paired layout (build_mel_band): even first bins, weights [tap / 4][lane][4] — four taps are one
// ds_read_b128 of weights and two ds_read_b64 of amplitudes (amp and the table 16-byte aligned in LDS).
template <bool PAIRED = false, class Emit>
TH_HD void mel_banded(uint32_t lane, const float *amp, const uint32_t *tab, uint32_t n_groups, const uint32_t (&off)[8],
                      const uint32_t (&n)[8], Emit emit) {
    uint32_t lo[8];  // every group's first bin up front: one LDS round trip for all of them
    TH_UNROLL for (uint32_t g = 0; g < 8; g++) lo[g] = tab[off[g < n_groups ? g : 0] + lane];
    TH_UNROLL for (uint32_t g = 0; g < 8; g++) {
        if (g < n_groups) {  // wave-uniform
            if constexpr (PAIRED) {
                auto ld2 = [](const float *q) -> cf32 {  // two amplitudes, one ds_read_b64
#if defined(__HIP_DEVICE_COMPILE__)
                    return lds_ld(reinterpret_cast<const cf32 *>(q));
#else
                    return {q[0], q[1]};
#endif
                };
                const float *ap = amp + lo[g];
                const float *wp = reinterpret_cast<const float *>(tab) + (off[g] + 64u + 4u * lane);
                float acc[4];
                {
                    const f32x4 w = lds_ld4(wp);
                    const cf32 a01 = ld2(ap), a23 = ld2(ap + 2);
                    acc[0] = a01.re * w.a;
                    acc[1] = a01.im * w.b;
                    acc[2] = a23.re * w.c;
                    acc[3] = a23.im * w.d;
                }
                ap += 4;
                wp += 256;
                uint32_t t = 4;
                for (; t + 8 <= n[g]; t += 8, ap += 8, wp += 512) {
                    const f32x4 w0 = lds_ld4(wp), w1 = lds_ld4(wp + 256);
                    cf32 a[4];
                    TH_UNROLL for (uint32_t u = 0; u < 4; u++) a[u] = ld2(ap + 2u * u);
                    acc[0] = fma_rn(a[0].re, w0.a, acc[0]);
                    acc[1] = fma_rn(a[0].im, w0.b, acc[1]);
                    acc[2] = fma_rn(a[1].re, w0.c, acc[2]);
                    acc[3] = fma_rn(a[1].im, w0.d, acc[3]);
                    acc[0] = fma_rn(a[2].re, w1.a, acc[0]);
                    acc[1] = fma_rn(a[2].im, w1.b, acc[1]);
                    acc[2] = fma_rn(a[3].re, w1.c, acc[2]);
                    acc[3] = fma_rn(a[3].im, w1.d, acc[3]);
                }
                if (t < n[g]) {  // (n is a multiple of 4)
                    const f32x4 w = lds_ld4(wp);
                    const cf32 a01 = ld2(ap), a23 = ld2(ap + 2);
                    acc[0] = fma_rn(a01.re, w.a, acc[0]);
                    acc[1] = fma_rn(a01.im, w.b, acc[1]);
                    acc[2] = fma_rn(a23.re, w.c, acc[2]);
                    acc[3] = fma_rn(a23.im, w.d, acc[3]);
                }
                emit(64u * g + lane, (acc[0] + acc[1]) + (acc[2] + acc[3]));
            } else {
            // (cast first, then index: 32-bit LDS address arithmetic instead of a 64-bit generic pointer that is truncated afterwards)
            TH_LDS_F32 *ap = TH_LDS_F32_PTR(amp) + lo[g];
            TH_LDS_F32 *wp = TH_LDS_F32_PTR(reinterpret_cast<const float *>(tab)) + (off[g] + 64u + lane);
            // four partial sums (taps t = u mod 4): a wide group's 64 taps are then four chains of 16 dependent FMAs, not one of 64;
            // the first four taps start them (n >= 4)
            float acc[4];
            {
                float a[4], w[4];
                TH_UNROLL for (uint32_t u = 0; u < 4; u++) {
                    a[u] = ap[u];
                    w[u] = wp[64u * u];
                }
                TH_UNROLL for (uint32_t u = 0; u < 4; u++) acc[u] = a[u] * w[u];
            }
            ap += 4;
            wp += 4 * 64;
            uint32_t t = 4;
            for (; t + 8 <= n[g]; t += 8, ap += 8, wp += 8 * 64) {
                float a[8], w[8];
                TH_UNROLL for (uint32_t u = 0; u < 8; u++) {
                    a[u] = ap[u];
                    w[u] = wp[64u * u];
                }
                TH_UNROLL for (uint32_t u = 0; u < 8; u++) acc[u & 3u] = fma_rn(a[u], w[u], acc[u & 3u]);
            }
            if (t < n[g]) {  // (n is a multiple of 4)
                float a[4], w[4];
                TH_UNROLL for (uint32_t u = 0; u < 4; u++) {
                    a[u] = ap[u];
                    w[u] = wp[64u * u];
                }
                TH_UNROLL for (uint32_t u = 0; u < 4; u++) acc[u] = fma_rn(a[u], w[u], acc[u]);
            }
            emit(64u * g + lane, (acc[0] + acc[1]) + (acc[2] + acc[3]));
            }
        }
    }
}

// Two frames at once (round 5, the frame-pair epilogue of stft_wave_kernel<.., OUT = 3>): amp[] holds the first frame's amplitude
// row, amp[RB ..] the second one's.  A group's first bins, its weight quads and the loop around them are read / run once for both
// frames — at the 48 kHz default (347 mels, 72 taps) the sums are 72 of the single-frame epilogue's ~250 vector instructions per
// frame, the rest is per group.  Paired table layout only; the sums of each frame are formed in mel_banded's order (bit-identical).
// All LDS addresses are 32-bit (cast first, then index) and every read keeps program order (volatile).
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(3))) float *lds_cfp;
#define TH_TO_LDS_CFP(p) ((lds_cfp)(p))
TH_HD f32x4 ldsp_ld4(lds_cfp p) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = *(const volatile __attribute__((address_space(3))) u32x4 *)(p);
    const uint32_t x = v.x, y = v.y, z = v.z, w = v.w;
    return {__builtin_bit_cast(float, x), __builtin_bit_cast(float, y), __builtin_bit_cast(float, z), __builtin_bit_cast(float, w)};
}
TH_HD cf32 ldsp_ld2(lds_cfp p) {
    const uint64_t v = *(const volatile __attribute__((address_space(3))) uint64_t *)(p);
    return {__builtin_bit_cast(float, (uint32_t)v), __builtin_bit_cast(float, (uint32_t)(v >> 32))};
}
#else
typedef const float *lds_cfp;
#define TH_TO_LDS_CFP(p) (p)
TH_HD f32x4 ldsp_ld4(lds_cfp p) { return {p[0], p[1], p[2], p[3]}; }
TH_HD cf32 ldsp_ld2(lds_cfp p) { return {p[0], p[1]}; }
#endif
// WIDE = false: four taps per loop trip instead of eight (24 registers fewer in flight: the n_fft 1024 kernel lives on 80 VGPRs)
template <int RB, bool WIDE = true, class EmitA, class EmitB>
TH_HD void mel_banded_pair(uint32_t lane, const float *amp, const uint32_t *tab, uint32_t n_groups, const uint32_t (&off)[8],
                           const uint32_t (&n)[8], EmitA emitA, EmitB emitB) {
    // every group's first bin up front: one LDS round trip for all of them (WIDE); the narrow form keeps one group ahead instead
    // (two registers, not eight)
    uint32_t lo[8];
    TH_UNROLL for (uint32_t g = 0; g < (WIDE ? 8u : 1u); g++) lo[g] = tab[off[g < n_groups ? g : 0] + lane];
    const lds_cfp amp3 = TH_TO_LDS_CFP(amp), tab3 = TH_TO_LDS_CFP(reinterpret_cast<const float *>(tab)) + 4u * lane;
    TH_UNROLL for (uint32_t g = 0; g < 8; g++) {
        if (g < n_groups) {  // wave-uniform
            if constexpr (!WIDE)
                if (g + 1 < 8) lo[(g + 1) % 8] = tab[off[g + 1 < n_groups ? g + 1 : 0] + lane];
            lds_cfp ap = amp3 + lo[g];
            lds_cfp wp = tab3 + (off[g] + 64u);
            float sa[4], sb[4];
            {
                const f32x4 w = ldsp_ld4(wp);
                const cf32 a01 = ldsp_ld2(ap), a23 = ldsp_ld2(ap + 2), b01 = ldsp_ld2(ap + RB), b23 = ldsp_ld2(ap + RB + 2);
                sa[0] = a01.re * w.a;
                sa[1] = a01.im * w.b;
                sa[2] = a23.re * w.c;
                sa[3] = a23.im * w.d;
                sb[0] = b01.re * w.a;
                sb[1] = b01.im * w.b;
                sb[2] = b23.re * w.c;
                sb[3] = b23.im * w.d;
            }
            ap += 4;
            wp += 256;
            uint32_t t = 4;
            for (; WIDE && t + 8 <= n[g]; t += 8, ap += 8, wp += 512) {
                const f32x4 w0 = ldsp_ld4(wp), w1 = ldsp_ld4(wp + 256);
                cf32 a[4], b[4];
                TH_UNROLL for (uint32_t u = 0; u < 4; u++) a[u] = ldsp_ld2(ap + 2u * u);
                TH_UNROLL for (uint32_t u = 0; u < 4; u++) b[u] = ldsp_ld2(ap + RB + 2u * u);
                sa[0] = fma_rn(a[0].re, w0.a, sa[0]);
                sa[1] = fma_rn(a[0].im, w0.b, sa[1]);
                sa[2] = fma_rn(a[1].re, w0.c, sa[2]);
                sa[3] = fma_rn(a[1].im, w0.d, sa[3]);
                sb[0] = fma_rn(b[0].re, w0.a, sb[0]);
                sb[1] = fma_rn(b[0].im, w0.b, sb[1]);
                sb[2] = fma_rn(b[1].re, w0.c, sb[2]);
                sb[3] = fma_rn(b[1].im, w0.d, sb[3]);
                sa[0] = fma_rn(a[2].re, w1.a, sa[0]);
                sa[1] = fma_rn(a[2].im, w1.b, sa[1]);
                sa[2] = fma_rn(a[3].re, w1.c, sa[2]);
                sa[3] = fma_rn(a[3].im, w1.d, sa[3]);
                sb[0] = fma_rn(b[2].re, w1.a, sb[0]);
                sb[1] = fma_rn(b[2].im, w1.b, sb[1]);
                sb[2] = fma_rn(b[3].re, w1.c, sb[2]);
                sb[3] = fma_rn(b[3].im, w1.d, sb[3]);
            }
            for (; t < n[g]; t += 4, ap += 4, wp += 256) {  // (n is a multiple of 4; WIDE: at most one trip)
                const f32x4 w = ldsp_ld4(wp);
                const cf32 a01 = ldsp_ld2(ap), a23 = ldsp_ld2(ap + 2), b01 = ldsp_ld2(ap + RB), b23 = ldsp_ld2(ap + RB + 2);
                sa[0] = fma_rn(a01.re, w.a, sa[0]);
                sa[1] = fma_rn(a01.im, w.b, sa[1]);
                sa[2] = fma_rn(a23.re, w.c, sa[2]);
                sa[3] = fma_rn(a23.im, w.d, sa[3]);
                sb[0] = fma_rn(b01.re, w.a, sb[0]);
                sb[1] = fma_rn(b01.im, w.b, sb[1]);
                sb[2] = fma_rn(b23.re, w.c, sb[2]);
                sb[3] = fma_rn(b23.im, w.d, sb[3]);
            }
            emitA(64u * g + lane, (sa[0] + sa[1]) + (sa[2] + sa[3]));
            emitB(64u * g + lane, (sb[0] + sb[1]) + (sb[2] + sb[3]));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Moment form of the mel filterbank (round 6; tables and the algebra: build_mel_moments, mel_fuse.h): lane = SEGMENT of the
// triangle points, no weight table.  amp[] = the frame's amplitude row in the wave's slab.  A lane walks the taps of its group
// downwards, P += a_t (a_t = 0 where the wave's mask for tap t does not name the lane), S1' += P: S0 = P at the end and
// S1' = sum_t a_t (t + 1).  Then R = alpha S1' + beta' S0 (the rising filter's share of the segment), F = S0 - R (the falling
// one's), and mel m = (R of segment m + F of segment m + 1) / d_m.
// ---------------------------------------------------------------------------------------------
struct MelMomLane {
    float R, F;
};
// a where the wave-uniform mask names this lane, else +0 (a select, not a product: whatever sits behind the amplitude row —
// stale exchange data, NaN — never enters a sum)
TH_HD float mel_mom_sel(float a, uint64_t mask, uint32_t lane) {
#if defined(__HIP_DEVICE_COMPILE__)
    (void)lane;
    return __builtin_amdgcn_inverse_ballot_w64(mask) ? a : 0.0f;  // v_cndmask_b32 with the mask's SGPR pair (a scalar load's result) as its condition
#else
    return ((mask >> lane) & 1u) ? a : 0.0f;
#endif
}
struct MelMomMasks4 {
    uint64_t m0, m1, m2, m3;
};
// the masks of taps t .. t + 3 (device: `masks` points into the CONSTANT address space and t is wave-uniform — one s_load_dwordx8)
template <class MaskPtr>
TH_HD MelMomMasks4 mel_mom_masks4(MaskPtr masks, uint32_t t) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint64_t u64x4v __attribute__((ext_vector_type(4)));
    const u64x4v v = *(const __attribute__((address_space(4))) u64x4v *)(masks + t);
    return {v.x, v.y, v.z, v.w};
#else
    return {masks[t], masks[t + 1], masks[t + 2], masks[t + 3]};
#endif
}
// one lane, one group: first = the lane's first bin, masks[t] = lanes whose segment holds bin first + t, n = the group's taps
// (a multiple of MEL_MOM_UNROLL = 4, >= 4).  Four taps per trip; the next trip's amplitudes and masks are requested before this
// trip's additions (two waves per SIMD hide little).
template <class MaskPtr>
TH_HD MelMomLane mel_mom_lane(uint32_t lane, const float *amp, uint32_t first, float alpha, float beta, MaskPtr masks, uint32_t n) {
    const float *ap = amp + first + n - 4;
    float P = 0.0f, S1 = 0.0f;
    float a0 = lds_ldf(ap), a1 = lds_ldf(ap + 1), a2 = lds_ldf(ap + 2), a3 = lds_ldf(ap + 3);
    MelMomMasks4 m = mel_mom_masks4(masks, n - 4);
    for (uint32_t t0 = n - 4;; t0 -= 4) {  // taps t0 + 3 .. t0
        const float c0 = a0, c1 = a1, c2 = a2, c3 = a3;
        const MelMomMasks4 cm = m;
        const bool more = t0 != 0;  // wave-uniform
        if (more) {
            ap -= 4;
            a0 = lds_ldf(ap);
            a1 = lds_ldf(ap + 1);
            a2 = lds_ldf(ap + 2);
            a3 = lds_ldf(ap + 3);
            m = mel_mom_masks4(masks, t0 - 4);
        }
        P += mel_mom_sel(c3, cm.m3, lane);
        S1 += P;
        P += mel_mom_sel(c2, cm.m2, lane);
        S1 += P;
        P += mel_mom_sel(c1, cm.m1, lane);
        S1 += P;
        P += mel_mom_sel(c0, cm.m0, lane);
        S1 += P;
        if (!more) break;
    }
    const float R = fma_rn(alpha, S1, beta * P);
    return {R, P - R};
}
// The same sums for a group of exactly 4 K taps as straight-line code (the kernel dispatches on the group's trip count): all
// amplitude reads up front, no loop registers to rotate, tap offsets as immediates.  Same additions in the same order as
// mel_mom_lane: bit-identical.
template <int K, class MaskPtr>
TH_HD MelMomLane mel_mom_lane_k(uint32_t lane, const float *amp, uint32_t first, float alpha, float beta, MaskPtr masks) {
    const float *ap = amp + first;
    float a[4 * K];
    TH_UNROLL for (int i = 4 * K - 1; i >= 0; i--) a[i] = lds_ldf(ap + i);
    float P = 0.0f, S1 = 0.0f;
    TH_UNROLL for (int c = K - 1; c >= 0; c--) {
        const MelMomMasks4 m = mel_mom_masks4(masks, 4u * (uint32_t)c);
        P += mel_mom_sel(a[4 * c + 3], m.m3, lane);
        S1 += P;
        P += mel_mom_sel(a[4 * c + 2], m.m2, lane);
        S1 += P;
        P += mel_mom_sel(a[4 * c + 1], m.m1, lane);
        S1 += P;
        P += mel_mom_sel(a[4 * c], m.m0, lane);
        S1 += P;
    }
    const float R = fma_rn(alpha, S1, beta * P);
    return {R, P - R};
}
// dispatch on the trip count (wave-uniform): straight-line code up to 32 taps, the loop beyond
template <class MaskPtr>
TH_HD MelMomLane mel_mom_lane_any(uint32_t lane, const float *amp, uint32_t first, float alpha, float beta, MaskPtr masks, uint32_t n) {
    switch (n >> 2) {
        case 1: return mel_mom_lane_k<1>(lane, amp, first, alpha, beta, masks);
        case 2: return mel_mom_lane_k<2>(lane, amp, first, alpha, beta, masks);
        case 3: return mel_mom_lane_k<3>(lane, amp, first, alpha, beta, masks);
        case 4: return mel_mom_lane_k<4>(lane, amp, first, alpha, beta, masks);
        case 5: return mel_mom_lane_k<5>(lane, amp, first, alpha, beta, masks);
        case 6: return mel_mom_lane_k<6>(lane, amp, first, alpha, beta, masks);
        case 7: return mel_mom_lane_k<7>(lane, amp, first, alpha, beta, masks);
        case 8: return mel_mom_lane_k<8>(lane, amp, first, alpha, beta, masks);
        default: return mel_mom_lane(lane, amp, first, alpha, beta, masks, n);
    }
}
// The select by the lane's own WINDOW instead of a wave-uniform mask (win = first tap | taps << 16, the table's window words):
// three vector operations instead of one, but no scalar load in the walk — what the workgroup-per-frame kernels want, whose
// waves all sit in the epilogue at once and hide nothing.
TH_HD float mel_mom_win_sel(float a, uint32_t t, uint32_t win) { return (t - (win & 0xffffu)) < (win >> 16) ? a : 0.0f; }
// one lane, one group, walked from tap n_run - 1 (n_run >= the group's taps, a multiple of 4: the taps of the widest group of
// the batch walked in lockstep).  Taps above the group's own add +0 to P = +0: bit-identical to mel_mom_lane.
TH_HD MelMomLane mel_mom_lane_win(const float *amp, uint32_t first, float alpha, float beta, uint32_t win, uint32_t n_run) {
    float P = 0.0f, S1 = 0.0f;
    for (uint32_t t = n_run; t-- != 0;) {
        P += mel_mom_win_sel(lds_ldf(amp + first + t), t, win);
        S1 += P;
    }
    const float R = fma_rn(alpha, S1, beta * P);
    return {R, P - R};
}
// W form (groups whose segments hold one or two bins): the (u, 1 - u) pairs themselves, exact products
TH_HD MelMomLane mel_mom_w_lane(const float *amp, uint32_t first, uint32_t n, float u0, float v0, float u1, float v1) {
    const float a0 = lds_ldf(amp + first);
    MelMomLane s = {a0 * u0, a0 * v0};
    if (n > 1) {  // wave-uniform
        const float a1 = lds_ldf(amp + first + 1);
        s.R = fma_rn(a1, u1, s.R);
        s.F = fma_rn(a1, v1, s.F);
    }
    return s;
}
// the filter output from its two shares (a sum of non-negative terms up to rounding: never below zero)
TH_HD float mel_mom_combine(float inv_d, float R, float F_next) {
    const float v = inv_d * (R + F_next);
    return v > 0.0f ? v : 0.0f;
}

#if defined(__HIPCC__)  // (both passes of hipcc; not the CPU lane emulator, which drives the lane functions itself)
// The whole epilogue of one frame — or, in the workgroup-per-frame kernels, one wave's share of it: the groups [g_lo, g_hi) from
// the top down (mel m needs F of segment m + 1: the next lane, or across the group border lane 0 of the group above — carried in
// a scalar; `carry` on entry = F of lane 0 of group g_hi, 0 above the last group), in BATCHES of MEL_MOM_BATCH groups: a batch's
// header words are one scalar load and its per-lane words are requested together — and a batch AHEAD of their use, i.e. before
// the rows of the batch in front of it are stored: vmcnt counts loads and stores in one order, so per-lane words requested behind
// a row store would wait for that store's round trip (two batches of 8 groups, each fetched when its turn came: 695 mels 0.88 ms
// against 0.78 for the group-ahead fetch this replaces; one group's own work is only a few dozen instructions —
// scripts/ubench/mom_probe.hip).  The table's group header carries MEL_MOM_BATCH zero entries behind the last group (taps 0:
// skipped; offset 0: the fetch reads the header itself).
template <class Emit>
__device__ __forceinline__ void mel_moments_range(uint32_t lane, const float *amp, gptr<const uint32_t> tab, uint32_t g_lo, uint32_t g_hi, float carry, Emit emit) {
    typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x8v __attribute__((ext_vector_type(8)));
    // the header words and the masks through the CONSTANT address space: the table is not written while the kernel runs, and
    // wave-uniform loads from there are scalar loads (the rows this kernel stores keep the compiler from proving that of a
    // global pointer)
    typedef const __attribute__((address_space(4))) uint64_t *cptr64;
    typedef const __attribute__((address_space(4))) u32x8v *cptr8v;
    constexpr uint32_t GB = MEL_MOM_BATCH;
    static_assert(GB == 4, "a batch's header is one s_load_dwordx8");
    if (g_hi <= g_lo) return;
    struct Batch {
        uint32_t nw[GB], off[GB];
        u32x4v prm[GB];
        u32x2v w1[GB];
    };
    const uint32_t lane16 = lane << 4, lane8 = lane << 3;
    auto fetch = [&](Batch &bt, uint32_t g0) {
        const u32x8v hv = *(cptr8v)(uintptr_t)(tab + MEL_MOM_HDR0 + 2u * g0);
        // (scalar copies: vector elements are not indexed dynamically, and __builtin_bit_cast on a vector-element lvalue reads element 0)
        const uint32_t nw[GB] = {hv.s0, hv.s2, hv.s4, hv.s6}, off[GB] = {hv.s1, hv.s3, hv.s5, hv.s7};
        TH_UNROLL for (uint32_t j = 0; j < GB; j++) {
            bt.nw[j] = g0 + j < g_hi ? nw[j] : 0u;  // (groups of the wave above, or the padding behind the last group: skipped)
            bt.off[j] = off[j];
            // address = block (scalar: table + offset) + zext(lane's byte offset): global_load's saddr + voffset form, no 64-bit vector adds
            const gptr<const char> blk = reinterpret_cast<gptr<const char>>(tab) + ((uint64_t)off[j] << 2);
            bt.prm[j] = *reinterpret_cast<gptr<const u32x4v>>(blk + (uint64_t)lane16);  // (padding groups: offset 0, the header itself)
            bt.w1[j] = *reinterpret_cast<gptr<const u32x2v>>(blk + 1024 + (uint64_t)lane8);  // (W form: the second pair; M form: two mask words, unused)
        }
    };
    // batches [b, b + GB) with b = g_lo + GB i: the top one first
    uint32_t b = g_lo + (g_hi - g_lo - 1u) / GB * GB;
    Batch nxt;
    fetch(nxt, b);
    for (;;) {
        const uint32_t g0 = b;
        const Batch cur = nxt;
        const bool more = b != g_lo;  // wave-uniform
        if (more) {
            b -= GB;
            fetch(nxt, b);
        }
        TH_UNROLL for (uint32_t jj = GB; jj-- != 0;) {
            const uint32_t n = cur.nw[jj] & 0xffffu;
            if (n != 0) {  // wave-uniform (0: not this wave's, or padding)
                const uint32_t first = cur.prm[jj].x, w_y = cur.prm[jj].y, w_z = cur.prm[jj].z, w_w = cur.prm[jj].w;
                MelMomLane s;
                if (cur.nw[jj] & 0x10000u) {  // W form
                    const uint32_t u1 = cur.w1[jj].x, v1 = cur.w1[jj].y;
                    s = mel_mom_w_lane(amp, first, n, __builtin_bit_cast(float, w_y), __builtin_bit_cast(float, w_z), __builtin_bit_cast(float, u1),
                                       __builtin_bit_cast(float, v1));
                } else {
                    const cptr64 masks = (cptr64)(uintptr_t)(tab + cur.off[jj] + 256u);
                    s = mel_mom_lane_any(lane, amp, first, __builtin_bit_cast(float, w_y), __builtin_bit_cast(float, w_z), masks, n);
                }
                // F of the next lane; lane 63 takes the group above's lane 0
                const float fn_in = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)(4u * ((lane + 1u) & 63u)), __builtin_bit_cast(int, s.F)));
                const float fn = lane == 63u ? carry : fn_in;
                carry = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s.F)));
                emit(64u * (g0 + jj) + lane, mel_mom_combine(__builtin_bit_cast(float, w_w), s.R, fn));
            }
        }
        if (!more) break;
    }
}
template <class Emit>
__device__ __forceinline__ void mel_moments_global(uint32_t lane, const float *amp, gptr<const uint32_t> tab, uint32_t n_groups, Emit emit) {
    mel_moments_range(lane, amp, tab, 0u, n_groups, 0.0f, emit);
}

// One wave's share of a frame in the workgroup-per-frame kernels: the groups [g_lo, g_hi), and the group g_hi above them walked
// along without being emitted (g_top = g_hi + 1, or g_hi at the top of the filterbank) for the F its lane 0 owes to mel
// 64 g_hi - 1.  All waves of the workgroup are in this epilogue together, so a wave's own latency is the frame's: the M groups of
// a batch are walked in LOCKSTEP (one loop over the taps of the widest, four independent P / S1' chains, sixteen amplitude reads
// in flight) and select by the per-lane window words (mel_mom_win_sel: no scalar load, i.e. no lgkmcnt(0) drain, inside the
// walk).  mel_moments_range's group-after-group walk cost this kernel ~2400 cycles per group (n_fft 16384, 2785 mels: 1.22 ms
// against 0.74 for linear rows).  Batches as there: the next one's header and per-lane words are requested before this one's
// rows are stored.  Reads above a narrower group's own taps stay inside the exchange buffer (build_mel_moments checks first +
// the table's widest group against max_index when asked to) and are never selected.
#if defined(TH_BLK_MEL_PROF)
#define TH_PROF_Q_PARAM , uint64_t *th_prof_q
#else
#define TH_PROF_Q_PARAM
#endif
// a batch's per-lane words: the two planes of the lane table (stft_core.h).  No load depends on another: the blocks sit at
// fixed addresses.  (Requested in front of the barrier before the epilogue, the first batch's words stay live across it and
// the kernels leave their register budget: 274 VGPRs at n_fft 8192, scratch at 16384.)
struct MelLaneBatch {
    typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
    u32x4v p0[MEL_MOM_BATCH], p1[MEL_MOM_BATCH];
};
__device__ __forceinline__ void mel_lane_fetch(MelLaneBatch &bt, gptr<const uint32_t> tab, uint32_t lane, uint32_t g0) {
    typedef MelLaneBatch::u32x4v u32x4v;
    TH_UNROLL for (uint32_t j = 0; j < MEL_MOM_BATCH; j++) {
        const gptr<const char> blk = reinterpret_cast<gptr<const char>>(tab) + 4u * (MEL_LANE_BLK0 + MEL_LANE_STRIDE * (g0 + j));
        bt.p0[j] = *reinterpret_cast<gptr<const u32x4v>>(blk + (uint64_t)(lane << 4));
        bt.p1[j] = *reinterpret_cast<gptr<const u32x4v>>(blk + 1024 + (uint64_t)(lane << 4));
    }
}
// the first (topmost) batch of the share [g_lo, g_top)
__device__ __forceinline__ uint32_t mel_lane_first_batch(uint32_t g_lo, uint32_t g_top) { return g_lo + (g_top - g_lo - 1u) / MEL_MOM_BATCH * MEL_MOM_BATCH; }
template <class Emit>
__device__ __forceinline__ void mel_moments_range_lockstep(uint32_t lane, const float *amp, gptr<const uint32_t> tab, uint32_t g_lo, uint32_t g_hi, uint32_t g_top, Emit emit TH_PROF_Q_PARAM) {
    typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
    constexpr uint32_t GB = MEL_MOM_BATCH;
    constexpr uint32_t TAPS = 4;  // per group and trip of the walk (the groups' taps are multiples of MEL_MOM_UNROLL = 4)
    if (g_hi <= g_lo) return;
    typedef MelLaneBatch Batch;
    auto fetch = [&](Batch &bt, uint32_t g0) { mel_lane_fetch(bt, tab, lane, g0); };
    uint32_t b = mel_lane_first_batch(g_lo, g_top);
    float carry = 0.0f;
    const uint32_t amp_b = lds_addr(amp);
    Batch nxt;
    fetch(nxt, b);
    for (;;) {
        const uint32_t g0 = b;
        const Batch got = nxt;
        const bool more = b != g_lo;  // wave-uniform
        if (more) {  // (requested before the batch in hand is looked at: both of a share's first two batches are one round trip)
            b -= GB;
            fetch(nxt, b);
        }
#if defined(TH_BLK_MEL_PROF)
        const uint64_t q0 = __builtin_readcyclecounter();
        __builtin_amdgcn_s_waitcnt(0);
        const uint64_t q1 = __builtin_readcyclecounter();
#endif
        struct {
            uint32_t nw[GB], win[GB];
            u32x4v prm[GB];
            u32x2v w1[GB];
        } cur;
        TH_UNROLL for (uint32_t j = 0; j < GB; j++) {
            const uint32_t w_nw = got.p1[j].z, w_win = got.p1[j].x, w_v1 = got.p1[j].y;  // (scalar copies, see mel_moments_range)
            cur.nw[j] = g0 + j < g_top ? (uint32_t)__builtin_amdgcn_readfirstlane((int)w_nw) : 0u;  // (the same word in every lane; groups above g_top and the padding: skipped)
            cur.win[j] = w_win;
            cur.w1[j] = u32x2v{w_win, w_v1};
            cur.prm[j] = got.p0[j];
        }
        uint32_t n_run = 0;  // taps of the batch's widest M group
        TH_UNROLL for (uint32_t j = 0; j < GB; j++)
            if (!(cur.nw[j] & 0x10000u)) n_run = max(n_run, cur.nw[j] & 0xffffu);
        float P[GB], S1[GB];
        TH_UNROLL for (uint32_t j = 0; j < GB; j++) P[j] = S1[j] = 0.0f;
        uint32_t ao[GB];  // LDS byte address of the group's first bin (32-bit: no flat-pointer arithmetic in the walk)
        TH_UNROLL for (uint32_t j = 0; j < GB; j++) {
            const uint32_t first = cur.prm[j].x;
            ao[j] = amp_b + 4u * first;
        }
        if (n_run != 0) {  // wave-uniform
            uint32_t wn[GB];  // the lane's taps: its window is [0, wn) (the lane table is built without shifted first bins)
            TH_UNROLL for (uint32_t j = 0; j < GB; j++) wn[j] = cur.win[j] >> 16;
            // taps t0 + TAPS - 1 .. t0 of every group per trip (not software-pipelined: the registers of a second set of
            // amplitudes cost the n_fft 8192 kernel its second workgroup per CU).  Per tap and group: one compare with the
            // wave-uniform tap number, one select, two additions — the epilogue is bound by the vector unit (the workgroup's
            // waves all walk at once, two to a SIMD).
            for (uint32_t t0 = n_run - TAPS;; t0 -= TAPS) {
                float c[GB][TAPS];
                TH_UNROLL for (uint32_t j = 0; j < GB; j++)
                    TH_UNROLL for (uint32_t i = 0; i < TAPS; i++) c[j][i] = lds_ldf_at(ao[j] + 4u * (t0 + i));
                TH_UNROLL for (uint32_t i = TAPS; i-- != 0;) {
                    TH_UNROLL for (uint32_t j = 0; j < GB; j++) {
                        P[j] += (t0 + i) < wn[j] ? c[j][i] : 0.0f;
                        S1[j] += P[j];
                    }
                }
                if (t0 == 0) break;  // wave-uniform
            }
        }
#if defined(TH_BLK_MEL_PROF)
        __builtin_amdgcn_s_waitcnt(0);
        const uint64_t q2 = __builtin_readcyclecounter();
#endif
        // The batch's four groups side by side, without a branch (group after group, every step waited for the one before:
        // ~400 cycles per group): the W groups' amplitudes, then the four exchanges of F, then the rows.
        float a0[GB], a1[GB];
        TH_UNROLL for (uint32_t j = 0; j < GB; j++) {
            a0[j] = lds_ldf_at(ao[j]);
            a1[j] = lds_ldf_at(ao[j] + 4u);
        }
        MelMomLane sj[GB];
        float fn_in[GB];
        TH_UNROLL for (uint32_t j = 0; j < GB; j++) {
            const uint32_t w_y = cur.prm[j].y, w_z = cur.prm[j].z, u1 = cur.w1[j].x, v1 = cur.w1[j].y;
            const float py = __builtin_bit_cast(float, w_y), pz = __builtin_bit_cast(float, w_z);
            const bool wform = (cur.nw[j] & 0x10000u) != 0, two = (cur.nw[j] & 0xffffu) > 1u;  // wave-uniform
            // W form: mel_mom_w_lane's operations
            float Rw = a0[j] * py, Fw = a0[j] * pz;
            const float Rw2 = fma_rn(a1[j], __builtin_bit_cast(float, u1), Rw), Fw2 = fma_rn(a1[j], __builtin_bit_cast(float, v1), Fw);
            Rw = two ? Rw2 : Rw;
            Fw = two ? Fw2 : Fw;
            const float Rm = fma_rn(py, S1[j], pz * P[j]), Fm = P[j] - Rm;
            sj[j].R = wform ? Rw : Rm;
            sj[j].F = wform ? Fw : Fm;
            fn_in[j] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)(4u * ((lane + 1u) & 63u)), __builtin_bit_cast(int, sj[j].F)));
        }
        TH_UNROLL for (uint32_t jj = GB; jj-- != 0;) {
            const bool part = (cur.nw[jj] & 0xffffu) != 0;  // wave-uniform (not: above g_top, or padding — the top of the top batch only)
            const uint32_t w_w = cur.prm[jj].w;
            const float fn = lane == 63u ? carry : fn_in[jj];
            const float f0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sj[jj].F)));
            carry = part ? f0 : carry;
            const uint32_t m = (part && g0 + jj < g_hi) ? 64u * (g0 + jj) + lane : 0xffffffffu;  // (emit drops mel numbers past the last)
            emit(m, mel_mom_combine(__builtin_bit_cast(float, w_w), sj[jj].R, fn));
        }
#if defined(TH_BLK_MEL_PROF)
        {
            const uint64_t q3 = __builtin_readcyclecounter();
            th_prof_q[0] += q1 - q0;
            th_prof_q[1] += q2 - q1;
            th_prof_q[2] += q3 - q2;
            th_prof_q[3] += 1;
        }
#endif
        if (!more) break;
    }
}

#endif

}  // namespace th
