// frontend_core.hpp -- the receiver front-end in front of rrc_filter: what
// examples/dmr-decoder.sh:13-17 does with `rtl_fm -M fm -s 48000 | csdr convert -i s16 -o float | csdr dcblock`.
// Neither rtl_fm nor csdr is part of the reference tree, so there is no source to follow: the arithmetic below is THIS
// project's specification ("front-end"), stated operation by operation so that the numpy / C
// oracle (oracle/frontend.c) reproduces every float bit.  PARITY UNPINNED by construction.
//
//   mode DH_FE_AUDIO_S16   x[n] = (float) s16[n] * 2^-15                                     (csdr convert -i s16 -o float)
//   mode DH_FE_IQ_S16      (re, im) = z[n] * conj(z[n-1]) in exact int32 arithmetic on the int16 I / Q pairs,
//                          x[n] = atan2(im, re) / pi with the polynomial below                (rtl_fm -M fm: polar discriminator)
//   dcblock                y[n] = (x[n] - x[n-1]) + 0.995f * y[n-1]   every operation rounded to float, in this order
//
// atan2(im, re): r = min(|re|, |im|) / max(|re|, |im|) (IEEE float division), s = r * r,
//   p = a16; p = p * s + a14; ... ; p = p * s + a2; p = p * s + 1; a = p * r   (separate multiply and add, each rounded:
//   the library is built with -ffp-contract=off; coefficients: Abramowitz & Stegun 4.4.49, |error| <= 2e-8),
//   |im| > |re| -> a = pi/2 - a;  re < 0 -> a = pi - a;  im < 0 -> a = -a;  re == im == 0 -> 0;  result a * (1 / pi).
// One channel per lane: the DC blocker is a strictly sequential recurrence.
#pragma once

#include "dh_portable.hpp"

#include "../../include/digiham_amd.h"      // DH_FE_AUDIO_S16, DH_FE_IQ_S16
#define DH_FE_STATE_WORDS 4            // per channel: x[n-1], y[n-1], I[n-1], Q[n-1] (the last two as floats holding int16 values)

DH_HD float dh_fe_atan2_over_pi(int32_t im, int32_t re) {
    if (re == 0 && im == 0) return 0.0f;
    const float fre = (float) re, fim = (float) im;              // |values| < 2^31: one rounding each
    const float are = fre < 0.0f ? -fre : fre, aim = fim < 0.0f ? -fim : fim;
    const bool swap = aim > are;
    const float r = (swap ? are : aim) / (swap ? aim : are);
    const float s = r * r;
    float p = 0.0028662257f;
    p = p * s; p = p + -0.0161657367f;
    p = p * s; p = p + 0.0429096138f;
    p = p * s; p = p + -0.0752896400f;
    p = p * s; p = p + 0.1065626393f;
    p = p * s; p = p + -0.1420889944f;
    p = p * s; p = p + 0.1999355085f;
    p = p * s; p = p + -0.3333314528f;
    p = p * s; p = p + 1.0f;
    float a = p * r;
    if (swap) a = 1.57079632679489661923f - a;
    if (fre < 0.0f) a = 3.14159265358979323846f - a;
    if (fim < 0.0f) a = -a;
    return a * 0.31830988618379067154f;
}

// sample t of a channel before the DC blocker; (ip, qp) = the I / Q pair before it (from the state for t = 0)
DH_HD float dh_fe_convert(const int16_t* in, size_t t, int mode, int32_t ip0, int32_t qp0) {
    if (mode == DH_FE_AUDIO_S16) return (float) in[t] * 0.000030517578125f;
    const int32_t i = in[2 * t], q = in[2 * t + 1];
    const int32_t ip = t ? (int32_t) in[2 * t - 2] : ip0, qp = t ? (int32_t) in[2 * t - 1] : qp0;
    return dh_fe_atan2_over_pi(q * ip - i * qp, i * ip + q * qp);
}

// n new samples of one channel; `in` points at its first new sample (int16 audio, or interleaved I / Q pairs)
DH_HD void dh_frontend_channel(const int16_t* in, float* out, float* st, size_t n, int mode, int dcblock) {
    float xp = st[0], yp = st[1];
    int32_t ip = (int32_t) st[2], qp = (int32_t) st[3];
    for (size_t t = 0; t < n; t++) {
        float x;
        if (mode == DH_FE_AUDIO_S16) x = (float) in[t] * 0.000030517578125f;
        else {
            const int32_t i = in[2 * t], q = in[2 * t + 1];
            x = dh_fe_atan2_over_pi(q * ip - i * qp, i * ip + q * qp);
            ip = i; qp = q;
        }
        float y = x;
        if (dcblock) { const float d = x - xp; const float f = 0.995f * yp; y = d + f; }
        xp = x; yp = y;
        out[t] = y;
    }
    st[0] = xp; st[1] = yp; st[2] = (float) ip; st[3] = (float) qp;
}
