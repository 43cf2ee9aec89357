/*
 * frontend.c -- oracle of the receiver front-end (TEST INFRASTRUCTURE ONLY): `rtl_fm -M fm | csdr convert -i s16 -o float |
 * csdr dcblock` of examples/dmr-decoder.sh:13-17.
 *
 * PARITY UNPINNED: neither rtl_fm nor csdr is in the reference tree (third-party tools of the example script), so there
 * is no reference source and no reference output for this stage.  What is restated here, independently of the product's
 * kernel, is the project's own specification of it ("front-end"):
 *   audio:  x[n] = s16[n] / 32768
 *   IQ:     d = z[n] conj(z[n-1]) (exact integers), x[n] = atan2(Im d, Re d) / pi through the A&S 4.4.49 polynomial
 *   dc:     y[n] = (x[n] - x[n-1]) + 0.995 y[n-1], float, in that order
 * Built with -ffp-contract=off (oracle/Makefile): every multiply and add below is one IEEE single operation.
 */
#include "dh_oracle.h"
#include <math.h>

static const float fe_coef[8] = { 0.0028662257f, -0.0161657367f, 0.0429096138f, -0.0752896400f,
                                  0.1065626393f, -0.1420889944f, 0.1999355085f, -0.3333314528f };

float orc_fe_atan2_over_pi(int32_t im, int32_t re) {
    if (re == 0 && im == 0) return 0.0f;
    const float fre = (float) re, fim = (float) im;
    const float are = fabsf(fre), aim = fabsf(fim);
    const float lo = aim > are ? are : aim, hi = aim > are ? aim : are;
    const float r = lo / hi;
    const float s = r * r;
    float p = fe_coef[0];
    for (int k = 1; k < 8; k++) { p = p * s; p = p + fe_coef[k]; }
    p = p * s; p = p + 1.0f;
    float a = p * r;
    if (aim > are) a = 1.57079632679489661923f - a;
    if (fre < 0.0f) a = 3.14159265358979323846f - a;
    if (fim < 0.0f) a = -a;
    return a * 0.31830988618379067154f;
}

/* state[4] = x[n-1], y[n-1], I[n-1], Q[n-1]; mode 1 = int16 audio, 2 = interleaved int16 I / Q */
void orc_frontend_process(float* state, const int16_t* in, size_t n, float* out, int mode, int dcblock) {
    float xp = state[0], yp = state[1];
    int32_t ip = (int32_t) state[2], qp = (int32_t) state[3];
    for (size_t t = 0; t < n; t++) {
        float x;
        if (mode == 1) x = (float) in[t] / 32768.0f;
        else {
            const int32_t i = in[2 * t], q = in[2 * t + 1];
            x = orc_fe_atan2_over_pi(q * ip - i * qp, i * ip + q * qp);
            ip = i; qp = q;
        }
        float y = x;
        if (dcblock) {
            const float d = x - xp;
            const float f = 0.995f * yp;
            y = d + f;
        }
        xp = x; yp = y;
        out[t] = y;
    }
    state[0] = xp; state[1] = yp; state[2] = (float) ip; state[3] = (float) qp;
}
