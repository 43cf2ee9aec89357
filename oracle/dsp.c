/*
 * dsp.c -- oracle restatement of the reference's DSP stages (TEST INFRASTRUCTURE ONLY).
 *
 * PARITY UNPINNED: the reference TUs (src/rrc_filter/rrc_filter.cpp,
 * src/gfsk_demodulator/gfsk_demodulator.cpp, src/fsk_demodulator/fsk_demodulator.cpp,
 * src/digitalvoice_filter/digitalvoice_filter.cpp) need <csdr/module.hpp>
 * (csdr 0.18, not in this image) and cannot be built here.  The arithmetic is
 * restated operation by operation; canonical evaluation = x86-64 SSE2 scalar,
 * no FMA contraction (build with -ffp-contract=off -O2, no -march), heap state
 * zero-initialised (the reference leaves `delay`, `variance_rb`
 * and `volume_rb` uninitialised; zero is the value a fresh calloc'd heap gives).
 */
#include "dh_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

/* ------------------------------------------------------------ RRC filter */
/* Filter coefficients: mkfilter/mkshape output quoted by the reference
 * (rrc_filter.cpp:92-112 wide, 81 taps, gain 8.337797030; :42-82 narrow, 161
 * taps, gain 16.67711971).  Both are symmetric; only the first half + centre
 * is stored.  These are interoperability constants (data), not code. */
#include "../digiham_amd/csrc/rrc_taps.h"

struct orc_rrc {
    unsigned n_zeros;
    double gain;
    float coeffs[DH_RRC_MAX_TAPS];
    float delay[DH_RRC_MAX_TAPS];
};

const float* orc_rrc_taps(int narrow, unsigned* n_zeros, double* gain) {
    static float wide[DH_RRC_MAX_TAPS], nar[DH_RRC_MAX_TAPS];
    static int init = 0;
    if (!init) {
        dh_rrc_expand_taps(0, wide);
        dh_rrc_expand_taps(1, nar);
        init = 1;
    }
    if (n_zeros) *n_zeros = narrow ? DH_RRC_NARROW_NZEROS : DH_RRC_WIDE_NZEROS;
    if (gain) *gain = narrow ? DH_RRC_NARROW_GAIN : DH_RRC_WIDE_GAIN;
    return narrow ? nar : wide;
}

/* rrc_filter.cpp:5-10 (ctor; delay is malloc'd, canonical = zero) */
orc_rrc* orc_rrc_new(int narrow) {
    orc_rrc* f = (orc_rrc*) calloc(1, sizeof(orc_rrc));
    const float* t = orc_rrc_taps(narrow, &f->n_zeros, &f->gain);
    memcpy(f->coeffs, t, sizeof(float) * (f->n_zeros + 1));
    return f;
}

/* RrcFilter(nZeros, gain, coeffs[]) with any table (include/rrc_filter.hpp:12, rrc_filter.cpp:6-14) */
orc_rrc* orc_rrc_new_custom(unsigned n_zeros, double gain, const float* coeffs) {
    if (n_zeros + 1 > DH_RRC_MAX_TAPS) return NULL;
    orc_rrc* f = (orc_rrc*) calloc(1, sizeof(orc_rrc));
    f->n_zeros = n_zeros; f->gain = gain;
    memcpy(f->coeffs, coeffs, sizeof(float) * (n_zeros + 1));
    return f;
}

void orc_rrc_free(orc_rrc* f) { free(f); }

/* rrc_filter.cpp:22-34 */
static float rrc_filter(orc_rrc* f, float sample) {
    float sum = 0.0f;
    for (unsigned i = 0; i < f->n_zeros; i++) f->delay[i] = f->delay[i + 1];
    f->delay[f->n_zeros] = sample;
    for (unsigned i = 0; i <= f->n_zeros; i++) sum += (f->coeffs[i] * f->delay[i]);
    return (float) (sum / f->gain);   /* float / double -> double divide, then narrowed */
}

/* rrc_filter.cpp:16-20 */
void orc_rrc_process(orc_rrc* f, const float* in, float* out, size_t n) {
    for (size_t i = 0; i < n; i++) out[i] = rrc_filter(f, in[i]);
}

/* ------------------------------------------------- GFSK / FSK demodulator */
#define VARIANCE_SYMBOLS 100   /* include/gfsk_demodulator.hpp:5 */
#define VOLUME_RB_SIZE 100     /* include/gfsk_demodulator.hpp:6 */

struct orc_demod {
    unsigned sps, lowest_eval, highest_eval, variance_rb_size, variance_rb_pos;
    int levels, invert;
    float* variance_rb;
    int variance_offset;
    float volume_rb[VOLUME_RB_SIZE];
    unsigned volume_rb_pos;
    float min, max, center, umid, lmid;
};

/* gfsk_demodulator.cpp:6-12 / fsk_demodulator.cpp:6-13 */
orc_demod* orc_demod_new(unsigned sps, int levels, int invert) {
    orc_demod* d = (orc_demod*) calloc(1, sizeof(orc_demod));
    d->sps = sps;
    d->levels = levels;
    d->invert = invert ? 1 : 0;
    d->lowest_eval = (unsigned) (int) roundf((float) sps / 3);
    d->highest_eval = (unsigned) (int) roundf((float) sps * 2 / 3);
    d->variance_rb_size = VARIANCE_SYMBOLS * sps;
    d->variance_rb = (float*) calloc(d->variance_rb_size, sizeof(float));
    return d;
}

void orc_demod_free(orc_demod* d) { if (d) { free(d->variance_rb); free(d); } }

/* gfsk_demodulator.cpp:109-122 / fsk_demodulator.cpp:102-112 */
static void calibrate_audio(orc_demod* d) {
    d->min = FLT_MAX; d->max = FLT_MIN;    /* sic: FLT_MIN is the smallest positive float */
    for (int i = 0; i < VOLUME_RB_SIZE; i++) {
        if (d->volume_rb[i] < d->min) d->min = d->volume_rb[i];
        if (d->volume_rb[i] > d->max) d->max = d->volume_rb[i];
    }
    d->center = (d->max + d->min) / 2;
    if (d->levels == 4) {
        d->umid = (float) ((d->max - d->center) * 0.625 + d->center);
        d->lmid = (float) ((d->min - d->center) * 0.625 + d->center);
    }
}

/* one call of gfsk_demodulator.cpp:24-107 (fsk_demodulator.cpp:25-100) */
static uint8_t demod_symbol(orc_demod* d, const float* input, size_t* advance) {
    float sum = 0.0f, volume_sum = 0.0f;
    for (size_t i = 0; i < d->sps; i++) {
        float value = input[i];
        if (i >= d->lowest_eval && i < d->highest_eval) sum += value;
        volume_sum += value;
        d->variance_rb[d->variance_rb_pos + i] = value;
    }
    *advance = (size_t) ((long) d->sps + d->variance_offset);
    d->variance_offset = 0;

    d->variance_rb_pos += d->sps;
    if (d->variance_rb_pos >= d->variance_rb_size) {
        double vmin = 0; size_t vmin_pos = 0;
        for (size_t i = 0; i < d->sps; i++) {
            float total = 0;
            for (int k = 0; k < VARIANCE_SYMBOLS; k++) total += d->variance_rb[k * d->sps + i];
            double mean = total / VARIANCE_SYMBOLS;          /* float division, then widened */
            double dsum = 0;
            for (int k = 0; k < VARIANCE_SYMBOLS; k++) {
                double diff = mean - d->variance_rb[k * d->sps + i];
                dsum += diff * diff;                         /* pow(x, 2) */
            }
            double variance = dsum / VARIANCE_SYMBOLS;
            if (i == 0 || variance < vmin) { vmin = variance; vmin_pos = i; }
        }
        if (vmin <= 0 || vmin > 5000000) {
        } else if (vmin_pos > 0 && vmin_pos < d->sps / 2) {
            d->variance_offset = +1;
        } else if (vmin_pos >= d->sps / 2 && vmin_pos < d->sps - 1) {
            d->variance_offset = -1;
        }
        d->variance_rb_pos %= d->variance_rb_size;
    }

    float volume_average = volume_sum / d->sps;
    d->volume_rb[d->volume_rb_pos] = volume_average;
    d->volume_rb_pos += 1;
    if (d->volume_rb_pos >= VOLUME_RB_SIZE) d->volume_rb_pos = 0;

    calibrate_audio(d);

    float average = sum / (d->highest_eval - d->lowest_eval);
    if (d->levels == 4) {
        if (average > d->center) return average > d->umid ? 1 : 0;
        return average < d->lmid ? 3 : 2;
    }
    if (average > d->center) return (uint8_t) !d->invert;
    return (uint8_t) d->invert;
}

/* the caller loop `while (canProcess()) process()` (src/lib/cli.cpp:29-33) with
 * canProcess = available > sps + 1 && writeable > 0 (gfsk_demodulator.cpp:18-22) */
size_t orc_demod_process(orc_demod* d, const float* in, size_t n, uint8_t* out, size_t cap, size_t* n_out) {
    size_t pos = 0, w = 0;
    while (n - pos > (size_t) d->sps + 1 && cap - w > 0) {
        size_t adv;
        out[w++] = demod_symbol(d, in + pos, &adv);
        pos += adv;
    }
    *n_out = w;
    return pos;
}

/* -------------------------------------------------- digital voice filter */
struct orc_dvfilter { float xv[11]; float yv[11]; };

orc_dvfilter* orc_dvfilter_new(void) { return (orc_dvfilter*) calloc(1, sizeof(orc_dvfilter)); }
void orc_dvfilter_free(orc_dvfilter* f) { free(f); }

/* digitalvoice_filter.cpp:34-45; GAIN 5 (:32).  The feed-forward part is float
 * arithmetic, the feedback products are double (double literals), the running
 * sum is double from the first feedback term on, the result is stored to float. */
static float dv_filter(orc_dvfilter* f, float sample) {
    float* xv = f->xv; float* yv = f->yv;
    for (int i = 0; i < 10; i++) xv[i] = xv[i + 1];
    xv[10] = sample / 5;
    for (int i = 0; i < 10; i++) yv[i] = yv[i + 1];
    yv[10] = (float) ((xv[10] - xv[0]) + 5 * (xv[2] - xv[8]) + 10 * (xv[6] - xv[4])
               + (  0.1254306222 * yv[0]) + (  0.1285714097 * yv[1])
               + ( -0.8106454980 * yv[2]) + ( -0.7664515771 * yv[3])
               + (  2.1846187758 * yv[4]) + (  1.8106678608 * yv[5])
               + ( -3.1465011600 * yv[6]) + ( -2.0391991609 * yv[7])
               + (  2.4873968618 * yv[8]) + (  1.0249072542 * yv[9]));
    return yv[10];
}

/* digitalvoice_filter.cpp:6-10.  (short) of an out-of-range float is undefined in C;
 * canonical = what x86-64 does: cvttss2si to int32 (0x80000000 when out of int32
 * range or NaN), then truncation to the low 16 bits. */
void orc_dvfilter_process(orc_dvfilter* f, const int16_t* in, int16_t* out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        float v = dv_filter(f, (float) in[i] / SHRT_MAX) * SHRT_MAX;
        int32_t iv;
        if (!(v > -2147483904.0f && v < 2147483648.0f)) iv = INT32_MIN;
        else iv = (int32_t) v;
        out[i] = (int16_t) (uint16_t) (uint32_t) iv;
    }
}
