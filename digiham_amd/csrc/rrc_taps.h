/*
 * rrc_taps.h -- root-raised-cosine coefficient tables (interoperability constants).
 *
 * mkfilter/mkshape designs quoted by the reference: wide = 81 taps for 12.5 kHz
 * channels (DMR, YSF; src/rrc_filter/rrc_filter.cpp:86-113, gain 8.337797030),
 * narrow = 161 taps for 6.25 kHz channels (NXDN; rrc_filter.cpp:36-84, gain
 * 16.67711971).  Both impulse responses are exactly symmetric, so only taps
 * 0..N/2 are stored and the table is mirrored when expanded.
 */
#ifndef DH_RRC_TAPS_H
#define DH_RRC_TAPS_H

#define DH_RRC_WIDE_NZEROS 80
#define DH_RRC_NARROW_NZEROS 160
#define DH_RRC_MAX_TAPS 161
#define DH_RRC_WIDE_GAIN 8.337797030e+00
#define DH_RRC_NARROW_GAIN 1.667711971e+01

static const float dh_rrc_wide_half[41] = {
    -0.0008938217f, -0.0002609230f, +0.0005898982f, +0.0016095188f, +0.0026805019f, +0.0035892828f,
    +0.0040255371f, +0.0036242975f, +0.0020553299f, -0.0008516117f, -0.0049736668f, -0.0097942071f,
    -0.0143781385f, -0.0174576799f, -0.0176417629f, -0.0137316693f, -0.0050921107f, +0.0080011038f,
    +0.0241300735f, +0.0407081846f, +0.0542175970f, +0.0607228306f, +0.0566126484f, +0.0394623171f,
    +0.0088613798f, -0.0329693214f, -0.0809351463f, -0.1273151201f, -0.1625361486f, -0.1764143887f,
    -0.1597076656f, -0.1057455528f, -0.0118628528f, +0.1196309860f, +0.2811569136f, +0.4603559944f,
    +0.6413467573f, +0.8066010425f, +0.9391765221f, +1.0249723677f, +1.0546584365f,
};

static const float dh_rrc_narrow_half[81] = {
    -0.0008965127f, -0.0006084266f, -0.0002629259f, +0.0001376901f, +0.0005891423f, +0.0010840181f,
    +0.0016105739f, +0.0021516457f, +0.0026838327f, +0.0031771176f, +0.0035950725f, +0.0038957679f,
    +0.0040334554f, +0.0039610403f, +0.0036332901f, +0.0030106572f, +0.0020635228f, +0.0007766025f,
    -0.0008467956f, -0.0027810092f, -0.0049751193f, -0.0073512625f, -0.0098044779f, -0.0122043473f,
    -0.0143986008f, -0.0162187503f, -0.0174876896f, -0.0180290597f, -0.0176780431f, -0.0162931143f,
    -0.0137681562f, -0.0100442577f, -0.0051204456f, +0.0009374242f, +0.0079903670f, +0.0158232514f,
    +0.0241456376f, +0.0325968938f, +0.0407558163f, +0.0481547523f, +0.0542979823f, +0.0586838603f,
    +0.0608299644f, +0.0603002781f, +0.0567332283f, +0.0498692532f, +0.0395764841f, +0.0258730951f,
    +0.0089449258f, -0.0108429006f, -0.0329414440f, -0.0566213193f, -0.0809844704f, -0.1049844817f,
    -0.1274551627f, -0.1471467396f, -0.1627685874f, -0.1730370678f, -0.1767267207f, -0.1727227994f,
    -0.1600729711f, -0.1380359261f, -0.1061246612f, -0.0641423317f, -0.0122087987f, +0.0492236806f,
    +0.1193667582f, +0.1971049660f, +0.2810174958f, +0.3694123940f, +0.4603722307f, +0.5518097911f,
    +0.6415318736f, +0.7273088884f, +0.8069476569f, +0.8783646253f, +0.9396566353f, +0.9891664557f,
    +1.0255404526f, +1.0477760738f, +1.0552572221f,
};

/* expand the half table into the full symmetric impulse response */
static inline void dh_rrc_expand_taps(int narrow, float* out) {
    const float* half = narrow ? dh_rrc_narrow_half : dh_rrc_wide_half;
    int nz = narrow ? DH_RRC_NARROW_NZEROS : DH_RRC_WIDE_NZEROS;
    for (int i = 0; i <= nz / 2; i++) {
        out[i] = half[i];
        out[nz - i] = half[i];
    }
}

#endif
