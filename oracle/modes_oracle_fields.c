/* modes_oracle_fields.c — plain-C restatement of the per-message field decode behind the CRC stage.
 * TEST INFRASTRUCTURE ONLY (see oracle_io.h): the checker the GPU field decoder is compared with.
 *
 * Follows, for a frame that was ACCEPTED (decodeResult 0):
 *   decodeModesMessage  mode_s.c:598-760   AA AC CA CC CF DR FS ID KE ND RI SL UM VS, altitude, squawk
 *   decodeExtendedSquitter and the ME decoders  mode_s.c:806-1555
 *   decodeAC13Field/decodeAC12Field/decodeID13Field/decodeMovementFieldV0/V2  mode_s.c:82-241
 *   modeAToModeC / internalModeAToModeC  mode_ac.c:80-170, modeAToIndex track.h:724
 *   decodeModeAMessage  mode_ac.c:171-200 (msgbits == 16)
 *   decodeCommB and its nine register hypotheses  comm_b.c:52-961 (DF20/21)
 * Pinned against the reference's own decodeModesMessage (oracle/_ref, ref_decode_fields) by tests/test_oracle_fields.py.
 *
 * Written around one 56-bit integer per field group (the first 32 bits of the frame, the ME block) and a
 * "bits first..last, 1-based, MSB first" extractor — the numbering the reference's comments and Annex 10 use. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "modes_oracle.h"

#define NON_ICAO (1u << 24)                 /* MODES_NON_ICAO_ADDRESS, readsb.h:295 */
#define BAD_ALT (-9999)                     /* INVALID_ALTITUDE, readsb.h:148 */

enum { AG_INVALID_ = 0, AG_GROUND_ = 1, AG_AIRBORNE_ = 2, AG_UNCERTAIN_ = 3 };                      /* readsb.h:216 */
enum { SRC_MODE_AC = 2, SRC_MODE_S = 5, SRC_MODE_S_CHECKED = 7, SRC_TISB = 8, SRC_ADSR = 9, SRC_ADSB = 10 };   /* readsb.h:159 */
enum { AT_ADSB_ICAO = 0, AT_ADSB_ICAO_NT = 1, AT_ADSR_ICAO = 2, AT_TISB_ICAO = 3, AT_MODE_S = 7, AT_ADSB_OTHER = 8,
       AT_ADSR_OTHER = 9, AT_TISB_TRACKFILE = 10, AT_TISB_OTHER = 11, AT_MODE_A = 12, AT_UNKNOWN = 15 };      /* readsb.h:178 */
enum { HD_GROUND_TRACK = 1, HD_TRUE = 2, HD_MAGNETIC = 3, HD_MAG_OR_TRUE = 4, HD_TRACK_OR_HEADING = 5 };   /* readsb.h:239 */
enum { SILT_UNKNOWN = 1, SILT_PER_SAMPLE = 2, SILT_PER_HOUR = 3 };                                   /* readsb.h:224 */
enum { CPRT_SURFACE = 1, CPRT_AIRBORNE = 2 };                                                        /* readsb.h:229 */
enum { NM_AUTOPILOT = 1, NM_VNAV = 2, NM_ALT_HOLD = 4, NM_APPROACH = 8, NM_LNAV = 16, NM_TCAS = 32 }; /* readsb.h:263 */
enum { NA_AIRCRAFT = 2, NA_MCP = 3, NA_FMS = 4 };                                                    /* readsb.h:287 */

/* bits first..last (1-based, bit 1 = MSB) of a `width`-bit big-endian integer */
static inline unsigned bits_of(uint64_t v, int width, int first, int last) {
    return (unsigned) ((v >> (width - last)) & ((1ull << (last - first + 1)) - 1));
}

/* Gillham: ID13 bit order C1 A1 C2 A2 C4 A4 X B1 D1 B2 D2 B4 D4 (bit 12 .. bit 0) -> hex digits A B C D (mode_s.c:82-100) */
static unsigned id13_to_hex(unsigned id13) {
    static const uint16_t place[13] = { /* bit 0..12 */ 0x0004, 0x0400, 0x0002, 0x0200, 0x0001, 0x0100, 0, 0x4000, 0x0040, 0x2000, 0x0020, 0x1000, 0x0010 };
    unsigned hex = 0;
    for (int b = 0; b < 13; ++b)
        if (id13 & (1u << b)) hex |= place[b];
    return hex;
}

/* Mode A (hex) -> Mode C in 100 ft units, or BAD_ALT (mode_ac.c:80-86 via the table built from :98-170; the table
 * index drops the zero bits, so D1 and the bits outside 0x7777 never reach the conversion) */
static int mode_a_to_c(unsigned a) {
    /* modeAToIndex/indexToModeA round trip (track.h:724-733): keep the 12 code bits only */
    a &= 0x7777;
    if ((a & 0x8889) != 0 || (a & 0x00f0) == 0) return BAD_ALT;       /* D1 set, or no C bits: not an altitude */
    unsigned hundreds = 0, fives = 0;
    if (a & 0x0010) hundreds ^= 7;
    if (a & 0x0020) hundreds ^= 3;
    if (a & 0x0040) hundreds ^= 1;
    if ((hundreds & 5) == 5) hundreds ^= 2;
    if (hundreds > 5) return BAD_ALT;
    if (a & 0x0002) fives ^= 0x0ff;
    if (a & 0x0004) fives ^= 0x07f;
    if (a & 0x1000) fives ^= 0x03f;
    if (a & 0x2000) fives ^= 0x01f;
    if (a & 0x4000) fives ^= 0x00f;
    if (a & 0x0100) fives ^= 0x007;
    if (a & 0x0200) fives ^= 0x003;
    if (a & 0x0400) fives ^= 0x001;
    if (fives & 1) hundreds = 6 - hundreds;
    return (int) (fives * 5 + hundreds) - 13;
}

static unsigned squawk_dec(unsigned hex) {   /* squawkHex2Dec, track.h:737 */
    return ((hex >> 12) & 15) * 1000 + ((hex >> 8) & 15) * 100 + ((hex >> 4) & 15) * 10 + (hex & 15);
}

static void set_squawk(struct oracle_fields *f, unsigned id13) {   /* setSquawkFromID13, mode_s.c:303 */
    f->squawkHex = (uint16_t) id13_to_hex(id13);
    f->squawkDec = (uint16_t) squawk_dec(f->squawkHex);
    f->flags |= ORACLE_F_SQUAWK_VALID;
}

/* 13-bit altitude code: returns feet or BAD_ALT; *unit 0 feet / 1 metres; *q the Q bit (mode_s.c:109-139) */
static int ac13_altitude(unsigned ac13, uint8_t *unit, int *q) {
    const int metres = (ac13 & 0x40) != 0;
    *q = (ac13 & 0x10) != 0;
    *unit = (uint8_t) metres;
    if (metres) return BAD_ALT;
    if (*q) {
        const int n = (int) (((ac13 & 0x1f80) >> 2) | ((ac13 & 0x20) >> 1) | (ac13 & 0x0f));
        return n * 25 - 1000;
    }
    const int c = mode_a_to_c(id13_to_hex(ac13));
    return c < -12 ? BAD_ALT : 100 * c;
}

/* 12-bit altitude code of the ES position (mode_s.c:147-173) */
static int ac12_altitude(unsigned ac12, uint8_t *unit, int *q) {
    *q = (ac12 & 0x10) != 0;
    *unit = 0;
    if (*q) {
        const int n = (int) (((ac12 & 0x0fe0) >> 1) | (ac12 & 0x0f));
        return n * 25 - 1000;
    }
    const unsigned with_m = ((ac12 & 0x0fc0) << 1) | (ac12 & 0x3f);
    const int c = mode_a_to_c(id13_to_hex(with_m));
    return c < -12 ? BAD_ALT : 100 * c;
}

/* surface movement code -> knots, midpoint of the code's range; the two ADS-B versions differ below code 9 (mode_s.c:181-224) */
static float movement_knots(unsigned m, int v2) {
    if (m >= 125) return 0;
    if (m == 124) return 180;
    if (m >= 109) return (float) (100 + (m - 109 + 0.5) * 5);
    if (m >= 94) return (float) (70 + (m - 94 + 0.5) * 2);
    if (m >= 39) return (float) (15 + (m - 39 + 0.5) * 1);
    if (m >= 13) return (float) (2 + (m - 13 + 0.5) * 0.50);
    if (m >= 9) return (float) (1 + (m - 9 + 0.5) * 0.25);
    if (v2) {
        if (m >= 3) return (float) (0.125 + (m - 3 + 0.5) * 0.875 / 6);
        if (m >= 2) return (float) (0.125 / 2);
        return 0;
    }
    if (m >= 2) return (float) (0.125 + (m - 2 + 0.5) * 0.125);
    return 0;
}

static void set_imf(struct oracle_fields *f) {   /* setIMF, mode_s.c:847-869 */
    f->addr |= NON_ICAO;
    if (f->addrtype == AT_ADSB_ICAO || f->addrtype == AT_ADSB_ICAO_NT) f->addrtype = AT_ADSB_OTHER;
    else if (f->addrtype == AT_TISB_ICAO) f->addrtype = AT_TISB_TRACKFILE;
    else if (f->addrtype == AT_ADSR_ICAO) f->addrtype = AT_ADSR_OTHER;
}

static const char kAis[65] = "@ABCDEFGHIJKLMNOPQRSTUVWXYZ[\\]^_ !\"#$%&'()*+,-./0123456789:;<=>?";   /* ais_charset.c:26 */

#define ME(a, b) bits_of(me, 56, (a), (b))

static void es_ident(struct oracle_fields *f, uint64_t me) {   /* mode_s.c:806-843 */
    f->mesub = (uint8_t) ME(6, 8);
    int ok = 1;
    for (int k = 0; k < 8; ++k) {
        const char c = kAis[ME(9 + 6 * k, 14 + 6 * k)];
        f->callsign[k] = c;
        if (!((c >= 'A' && c <= 'Z') || (c >= '-' && c <= '9') || c == ' ' || c == '@')) ok = 0;
    }
    if (ok) f->flags |= ORACLE_F_CALLSIGN_VALID;
    f->category = (uint8_t) (((0x0e - f->metype) << 4) | f->mesub);
    f->flags |= ORACLE_F_CATEGORY_VALID;
}

static void es_velocity(struct oracle_fields *f, uint64_t me, int check_imf) {   /* mode_s.c:871-977 */
    f->mesub = (uint8_t) ME(6, 8);
    if (f->mesub < 1 || f->mesub > 4) return;
    if (check_imf && ME(9, 9)) set_imf(f);
    f->acc_flags |= ORACLE_ACC_NAC_V_VALID;
    f->nac_v = (uint8_t) ME(11, 13);
    if (f->mesub <= 2) {
        const unsigned ew_raw = ME(15, 24), ns_raw = ME(26, 35);
        if (ew_raw && ns_raw) {
            const int scale = f->mesub == 2 ? 4 : 1;
            const int ew = (int) (ew_raw - 1) * (ME(14, 14) ? -1 : 1) * scale;
            const int ns = (int) (ns_raw - 1) * (ME(25, 25) ? -1 : 1) * scale;
            const float gs = sqrtf((float) ((ns * ns) + (ew * ew) + 0.5));
            f->gs_v0 = f->gs_v2 = f->gs_selected = gs;
            f->flags |= ORACLE_F_GS_VALID;
            if (gs > 0) {
                float track = (float) (atan2((double) ew, (double) ns) * 180.0 / M_PI);
                if (track < 0) track += 360;
                f->heading = track;
                f->heading_type = HD_GROUND_TRACK;
                f->flags |= ORACLE_F_HEADING_VALID;
            }
        }
    } else {
        if (ME(14, 14)) {
            f->flags |= ORACLE_F_HEADING_VALID;
            f->heading = (float) (ME(15, 24) * 360.0 / 1024.0);
            f->heading_type = HD_MAG_OR_TRUE;
        }
        const unsigned airspeed = ME(26, 35);
        if (airspeed) {
            const unsigned kt = (airspeed - 1) * (f->mesub == 4 ? 4 : 1);
            if (ME(25, 25)) { f->flags |= ORACLE_F_TAS_VALID; f->tas = (uint16_t) kt; }
            else { f->flags |= ORACLE_F_IAS_VALID; f->ias = (uint16_t) kt; }
        }
    }
    const unsigned vr = ME(38, 46);
    if (vr) {
        const int rate = (int) (vr - 1) * (ME(37, 37) ? -64 : 64);
        if (ME(36, 36)) { f->baro_rate = rate; f->flags |= ORACLE_F_BARO_RATE_VALID; }
        else { f->geom_rate = rate; f->flags |= ORACLE_F_GEOM_RATE_VALID; }
    }
    const unsigned delta = ME(50, 56);
    if (delta) {
        f->flags |= ORACLE_F_GEOM_DELTA_VALID;
        f->geom_delta = (int) (delta - 1) * (ME(49, 49) ? -25 : 25);
    }
}

static void es_surface(struct oracle_fields *f, uint64_t me, int check_imf) {   /* mode_s.c:979-1016 */
    f->airground = AG_GROUND_;
    f->flags |= ORACLE_F_CPR_VALID;
    f->cpr_type = CPRT_SURFACE;
    const unsigned movement = ME(6, 12);
    if (movement > 0 && movement < 125) {
        f->flags |= ORACLE_F_GS_VALID;
        f->gs_selected = f->gs_v0 = movement_knots(movement, 0);
        f->gs_v2 = movement_knots(movement, 1);
    }
    if (ME(13, 13)) {
        f->flags |= ORACLE_F_HEADING_VALID;
        f->heading = (float) (ME(14, 20) * 360.0 / 128.0);
        f->heading_type = HD_TRACK_OR_HEADING;
    }
    if (check_imf && ME(21, 21)) set_imf(f);
    if (ME(22, 22)) f->flags |= ORACLE_F_CPR_ODD;
    f->cpr_lat = ME(23, 39);
    f->cpr_lon = ME(40, 56);
}

static void es_airborne(struct oracle_fields *f, uint64_t me, int check_imf) {   /* mode_s.c:1018-1101 */
    switch (ME(6, 7)) {
        case 0: f->flags |= ORACLE_F_ALERT_VALID | ORACLE_F_SPI_VALID; f->flags &= ~(ORACLE_F_ALERT | ORACLE_F_SPI); break;
        case 1: case 2: f->flags |= ORACLE_F_ALERT_VALID | ORACLE_F_ALERT; break;
        case 3: f->flags |= ORACLE_F_ALERT_VALID | ORACLE_F_SPI_VALID | ORACLE_F_SPI; f->flags &= ~ORACLE_F_ALERT; break;
    }
    if (check_imf) {
        if (ME(8, 8)) set_imf(f);
    } else {
        f->acc_flags |= ORACLE_ACC_NIC_B_VALID | (ME(8, 8) ? ORACLE_ACC_NIC_B : 0);
    }
    const unsigned ac12 = ME(9, 20);
    if (f->metype != 0) {
        f->cpr_lat = ME(23, 39);
        f->cpr_lon = ME(40, 56);
        const int bogus = ac12 == 0 && f->cpr_lon == 0 && (f->cpr_lat & 0x0fff) == 0 && f->metype == 15;
        if (!bogus) {
            f->flags |= ORACLE_F_CPR_VALID | (ME(22, 22) ? ORACLE_F_CPR_ODD : 0);
            f->cpr_type = CPRT_AIRBORNE;
        }
    }
    if (ac12 && f->airground != AG_GROUND_) {
        uint8_t unit; int q;
        const int alt = ac12_altitude(ac12, &unit, &q);
        if (alt != BAD_ALT) {
            if (q) f->flags |= ORACLE_F_ALT_Q_BIT;
            if (f->metype >= 20 && f->metype <= 22) { f->geom_alt = alt; f->geom_alt_unit = unit; f->flags |= ORACLE_F_GEOM_ALT_VALID; }
            else { f->baro_alt = alt; f->baro_alt_unit = unit; f->flags |= ORACLE_F_BARO_ALT_VALID; }
        }
    }
}

static void es_test(struct oracle_fields *f, uint64_t me) {   /* mode_s.c:1103-1114 */
    f->mesub = (uint8_t) ME(6, 8);
    if (f->mesub == 7 && ME(9, 21)) set_squawk(f, ME(9, 21));
}

static void es_status(struct oracle_fields *f, uint64_t me, int check_imf) {   /* mode_s.c:1116-1138 */
    f->mesub = (uint8_t) ME(6, 8);
    if (f->mesub == 1) {
        f->flags |= ORACLE_F_EMERGENCY_VALID;
        f->emergency = (uint8_t) ME(9, 11);
        if (ME(12, 24)) set_squawk(f, ME(12, 24));
        if (check_imf && ME(56, 56)) set_imf(f);
    }
    if (f->mesub == 2) f->flags |= ORACLE_F_ACAS_RA_VALID;
}

static void es_target_state(struct oracle_fields *f, uint64_t me, int check_imf) {   /* mode_s.c:1140-1342 */
    f->mesub = (uint8_t) ME(6, 7);
    if (check_imf && ME(51, 51)) set_imf(f);
    if (f->mesub == 0 && ME(11, 11) == 0) {                      /* version 1 layout */
        static const uint8_t src_of[4] = {0, NA_MCP, NA_AIRCRAFT, NA_FMS};
        if (ME(8, 9)) f->nav_altitude_source = src_of[ME(8, 9)];
        const unsigned vmode = ME(14, 15);
        if (vmode == 1 || vmode == 2) {
            f->nav_flags |= ORACLE_NAV_MODES_VALID;
            if (f->nav_altitude_source == NA_FMS) f->nav_modes |= NM_VNAV;
            else if (vmode == 2 && f->nav_altitude_source == NA_AIRCRAFT) f->nav_modes |= NM_ALT_HOLD;
            else f->nav_modes |= NM_AUTOPILOT;
        }
        const int alt = -1000 + 100 * (int) ME(16, 25);
        if (f->nav_altitude_source == NA_MCP) { f->nav_flags |= ORACLE_NAV_MCP_ALT_VALID; f->nav_mcp_altitude = (uint32_t) alt; }
        else if (f->nav_altitude_source == NA_FMS) { f->nav_flags |= ORACLE_NAV_FMS_ALT_VALID; f->nav_fms_altitude = (uint32_t) alt; }
        const unsigned hsrc = ME(26, 27);
        if (hsrc) {
            f->nav_flags |= ORACLE_NAV_HEADING_VALID;
            f->nav_heading = (float) ME(28, 36);
            f->nav_heading_type = ME(37, 37) ? HD_GROUND_TRACK : HD_MAG_OR_TRUE;
        }
        const unsigned hmode = ME(38, 39);
        if (hmode == 1 || hmode == 2) {
            f->nav_flags |= ORACLE_NAV_MODES_VALID;
            f->nav_modes |= hsrc == 3 ? NM_LNAV : NM_AUTOPILOT;
        }
        f->acc_flags |= ORACLE_ACC_NAC_P_VALID | ORACLE_ACC_NIC_BARO_VALID | (ME(44, 44) ? ORACLE_ACC_NIC_BARO : 0);
        f->nac_p = (uint8_t) ME(40, 43);
        f->sil = (uint8_t) ME(45, 46);
        f->sil_type = SILT_UNKNOWN;
        const unsigned tcas = ME(52, 53);
        if (tcas) f->nav_flags |= ORACLE_NAV_MODES_VALID;
        if (tcas != 1) f->nav_modes |= NM_TCAS;
        f->flags |= ORACLE_F_EMERGENCY_VALID;
        f->emergency = (uint8_t) ME(54, 56);
    } else if (f->mesub == 1) {                                  /* version 2 layout */
        const unsigned alt_bits = ME(10, 20);
        if (alt_bits) {
            if (ME(9, 9)) { f->nav_flags |= ORACLE_NAV_FMS_ALT_VALID; f->nav_fms_altitude = (alt_bits - 1) * 32; }
            else { f->nav_flags |= ORACLE_NAV_MCP_ALT_VALID; f->nav_mcp_altitude = (alt_bits - 1) * 32; }
        }
        const unsigned baro = ME(21, 29);
        if (baro) { f->nav_flags |= ORACLE_NAV_QNH_VALID; f->nav_qnh = (float) (800.0 + (baro - 1) * 0.8); }
        if (ME(30, 30)) {
            f->nav_flags |= ORACLE_NAV_HEADING_VALID;
            f->nav_heading = (float) (ME(31, 39) * 180.0 / 256.0);
            f->nav_heading_type = HD_MAG_OR_TRUE;
        }
        f->acc_flags |= ORACLE_ACC_NAC_P_VALID | ORACLE_ACC_NIC_BARO_VALID | (ME(44, 44) ? ORACLE_ACC_NIC_BARO : 0);
        f->nac_p = (uint8_t) ME(40, 43);
        f->sil = (uint8_t) ME(45, 46);
        f->sil_type = SILT_UNKNOWN;
        if (ME(47, 47)) {
            f->nav_flags |= ORACLE_NAV_MODES_VALID;
            f->nav_modes = (uint8_t) ((ME(48, 48) ? NM_AUTOPILOT : 0) | (ME(49, 49) ? NM_VNAV : 0) | (ME(50, 50) ? NM_ALT_HOLD : 0)
                                      | (ME(52, 52) ? NM_APPROACH : 0) | (ME(53, 53) ? NM_TCAS : 0) | (ME(54, 54) ? NM_LNAV : 0));
        }
    }
}

static void op_set(struct oracle_fields *f, unsigned flag, unsigned on) { if (on) f->op_flags |= (uint16_t) flag; }

static void es_opstatus(struct oracle_fields *f, uint64_t me, int check_imf) {   /* mode_s.c:1344-1455 */
    f->mesub = (uint8_t) ME(6, 8);
    if (check_imf && ME(56, 56)) set_imf(f);
    if (f->mesub > 1) return;
    const int airborne = f->mesub == 0;
    f->op_flags |= ORACLE_OP_VALID;
    f->op_version = (uint8_t) ME(41, 43);
    const unsigned cc_hdr = ME(9, 10);
    if (f->op_version == 0) {
        if (airborne && cc_hdr == 0) { op_set(f, ORACLE_OP_CC_ACAS, !ME(12, 12)); op_set(f, ORACLE_OP_CC_CDTI, ME(13, 13)); }
        return;
    }
    if (f->op_version > 2) return;
    const int v2 = f->op_version == 2;
    if (ME(25, 26) == 0) {
        op_set(f, ORACLE_OP_OM_ACAS_RA, ME(27, 27)); op_set(f, ORACLE_OP_OM_IDENT, ME(28, 28)); op_set(f, ORACLE_OP_OM_ATC, ME(29, 29));
        if (v2) {
            op_set(f, ORACLE_OP_OM_SAF, ME(30, 30));
            f->acc_flags |= ORACLE_ACC_SDA_VALID;
            f->sda = (uint8_t) ME(31, 32);
        }
    }
    if (cc_hdr == 0 && (v2 || ME(13, 14) == 0)) {
        if (airborne) {
            op_set(f, ORACLE_OP_CC_ACAS, v2 ? ME(11, 11) : !ME(11, 11));
            op_set(f, v2 ? ORACLE_OP_CC_1090_IN : ORACLE_OP_CC_CDTI, ME(12, 12));
            op_set(f, ORACLE_OP_CC_ARV, ME(15, 15)); op_set(f, ORACLE_OP_CC_TS, ME(16, 16));
            f->op_cc_tc = (uint8_t) ME(17, 18);
            if (v2) op_set(f, ORACLE_OP_CC_UAT_IN, ME(19, 19));
        } else {
            op_set(f, ORACLE_OP_CC_POA, ME(11, 11));
            op_set(f, v2 ? ORACLE_OP_CC_1090_IN : ORACLE_OP_CC_CDTI, ME(12, 12));
            op_set(f, ORACLE_OP_CC_B2_LOW, ME(15, 15));
            if (v2) {
                op_set(f, ORACLE_OP_CC_UAT_IN, ME(16, 16));
                f->acc_flags |= ORACLE_ACC_NAC_V_VALID | ORACLE_ACC_NIC_C_VALID | (ME(20, 20) ? ORACLE_ACC_NIC_C : 0);
                f->nac_v = (uint8_t) ME(17, 19);
                f->op_cc_antenna_offset = (uint8_t) ME(33, 40);
            }
            f->op_flags |= ORACLE_OP_CC_LW_VALID;
            f->op_cc_lw = (uint8_t) ME(21, 24);
        }
    }
    f->acc_flags |= ORACLE_ACC_NIC_A_VALID | (ME(44, 44) ? ORACLE_ACC_NIC_A : 0) | ORACLE_ACC_NAC_P_VALID;
    f->nac_p = (uint8_t) ME(45, 48);
    f->sil = (uint8_t) ME(51, 52);
    f->sil_type = v2 ? (ME(55, 55) ? SILT_PER_SAMPLE : SILT_PER_HOUR) : SILT_UNKNOWN;
    f->op_hrd = ME(54, 54) ? HD_MAGNETIC : HD_TRUE;
    if (airborne) {
        if (v2) { f->acc_flags |= ORACLE_ACC_GVA_VALID; f->gva = (uint8_t) ME(49, 50); }
        f->acc_flags |= ORACLE_ACC_NIC_BARO_VALID | (ME(53, 53) ? ORACLE_ACC_NIC_BARO : 0);
    } else {
        f->op_tah = ME(53, 53) ? f->op_hrd : HD_GROUND_TRACK;
    }
}

static void extended_squitter(struct oracle_fields *f, uint64_t me) {   /* decodeExtendedSquitter, mode_s.c:1457-1555 */
    f->metype = (uint8_t) ME(1, 5);
    int check_imf = 0;
    if (f->msgtype == 18) {
        switch (f->CF) {
            case 0: f->addrtype = AT_ADSB_ICAO_NT; break;
            case 1: f->addrtype = AT_ADSB_OTHER; f->addr |= NON_ICAO; break;
            case 2: f->source = SRC_TISB; f->addrtype = AT_TISB_ICAO; check_imf = 1; break;
            case 3: f->source = SRC_TISB; f->addrtype = AT_TISB_ICAO; if (ME(1, 1)) set_imf(f); return;
            case 5: f->addrtype = AT_TISB_OTHER; f->source = SRC_TISB; f->addr |= NON_ICAO; break;
            case 6: f->addrtype = AT_ADSR_ICAO; f->source = SRC_ADSR; check_imf = 1; break;
            default: f->addrtype = AT_UNKNOWN; f->addr |= NON_ICAO; return;
        }
    }
    const unsigned t = f->metype;
    if (t >= 1 && t <= 4) es_ident(f, me);
    else if (t == 19) es_velocity(f, me, check_imf);
    else if (t >= 5 && t <= 8) es_surface(f, me, check_imf);
    else if (t == 0 || (t >= 9 && t <= 18) || (t >= 20 && t <= 22)) es_airborne(f, me, check_imf);
    else if (t == 23) es_test(f, me);
    else if (t == 28) es_status(f, me, check_imf);
    else if (t == 29) es_target_state(f, me, check_imf);
    else if (t == 31) es_opstatus(f, me, check_imf);
}

/* ------------------------------------------------------------------------------------------------ comm_b.c
 * decodeCommB (comm_b.c:52-86): the requested register is not known, so every register hypothesis is scored on the
 * 56-bit MB field and the single best one (score > 0) is decoded; two hypotheses sharing the best score -> AMBIGUOUS.
 * One function per hypothesis, `f` NULL = score only.  Formats: commb_format_t, readsb.h:249. */
enum { CB_UNKNOWN, CB_AMBIGUOUS, CB_EMPTY, CB_DATALINK_CAPS, CB_GICB_CAPS, CB_IDENT, CB_ACAS_RA, CB_VERTICAL_INTENT, CB_TRACK_TURN,
       CB_HEADING_SPEED, CB_MET_ROUTINE };

#define MBF(a, b) bits_of(mb, 56, (a), (b))

/* "status bit + value" subfield rule used all over BDS4,0/5,0/6,0: a set status bit needs a non-zero value (checked by the
 * caller), a clear one needs a zero value — anything else disqualifies the hypothesis.  Returns 1 (+1 score) or -1 (reject). */
static int idle_field(unsigned valid, unsigned raw) { return (!valid && raw == 0) ? 1 : -1; }

static int cb_empty(uint64_t mb, struct oracle_fields *f) {            /* comm_b.c:88-100 */
    if (mb) return 0;
    if (f) f->commb_format = CB_EMPTY;
    return 56;
}

static int cb_bds10(uint64_t mb, struct oracle_fields *f) {            /* comm_b.c:104-124 */
    if (MBF(1, 8) != 0x10 || MBF(10, 14)) return 0;
    if (f) f->commb_format = CB_DATALINK_CAPS;
    return 56;
}

static int cb_bds17(uint64_t mb, struct oracle_fields *f) {            /* comm_b.c:128-205 */
    if (MBF(25, 56)) return 0;
    int score = MBF(7, 7) ? 1 : -2;                                     /* BDS2,0 is on almost everything */
    static const int unlikely[8] = {10, 11, 12, 13, 14, 20, 21, 22};
    for (int k = 0; k < 8; ++k) if (MBF(unlikely[k], unlikely[k])) score -= 2;
    const unsigned es = MBF(1, 6);
    if ((es >> 1) == 0x1f) score += 5 + (int) (es & 1);                 /* ES capable (+ EDI) */
    else if (es == 0) score += 1;
    else score -= 12;
    const unsigned tt = MBF(16, 16), hs = MBF(24, 24), vi = MBF(9, 9);
    if (tt && hs) score += 2 + (int) vi;
    else if (!tt && !hs && !vi) score += 1;
    else score -= 6;
    if (f) f->commb_format = CB_GICB_CAPS;
    return score;
}

static int cb_bds20(uint64_t mb, struct oracle_fields *f) {            /* comm_b.c:209-259 */
    if (MBF(1, 8) != 0x20) return 0;
    char cs[8];
    for (int k = 0; k < 8; ++k) {
        const char c = kAis[MBF(9 + 6 * k, 14 + 6 * k)];
        if (!((c >= 'A' && c <= 'Z') || (c >= '-' && c <= '9') || c == ' ' || c == '@')) return 0;
        cs[k] = c;
    }
    if (f) { f->commb_format = CB_IDENT; memcpy(f->callsign, cs, 8); f->flags |= ORACLE_F_CALLSIGN_VALID; }
    return 8 + 8 * 6;
}

static int cb_bds30(uint64_t mb, struct oracle_fields *f) {            /* comm_b.c:333-348 */
    if (MBF(1, 8) != 0x30) return 0;
    if (f) { f->commb_format = CB_ACAS_RA; f->flags |= ORACLE_F_ACAS_RA_VALID; }
    return 56;
}

static int cb_bds40(uint64_t mb, struct oracle_fields *f) {            /* comm_b.c:352-514 */
    const unsigned mcp_v = MBF(1, 1), mcp = MBF(2, 13), fms_v = MBF(14, 14), fms = MBF(15, 26), baro_v = MBF(27, 27), baro = MBF(28, 39);
    const unsigned mode_v = MBF(48, 48), mode = MBF(49, 51), src_v = MBF(54, 54), src = MBF(55, 56);
    if (!(mcp_v | fms_v | baro_v | mode_v | src_v)) return 0;
    int score = 0, r;
    unsigned alt[2] = {0, 0};
    const unsigned av[2] = {mcp_v, fms_v}, araw[2] = {mcp, fms};
    for (int k = 0; k < 2; ++k) {
        if (av[k] && araw[k]) {
            alt[k] = araw[k] * 16;
            if (alt[k] < 1000 || alt[k] > 50000) return 0;
            score += 13;
        } else if ((r = idle_field(av[k], araw[k])) < 0) return 0;
        else score += r;
    }
    float qnh = 0;
    if (baro_v && baro) {
        qnh = (float) (800 + baro * 0.1);
        if (!(qnh >= 900 && qnh <= 1100)) return 0;
        score += 13;
    } else if ((r = idle_field(baro_v, baro)) < 0) return 0;
    else score += r;
    if (MBF(40, 47)) return 0;
    if (mode_v) score += 4; else if (mode == 0) score += 1; else return 0;
    if (MBF(52, 53)) return 0;
    if (src_v) score += 3; else if (src == 0) score += 1; else return 0;
    if (mcp_v && fms_v && alt[0] != alt[1]) score -= 4;
    for (int k = 0; k < 2; ++k) {
        const unsigned rem = alt[k] % 500;
        if (av[k] && !(rem < 16 || rem > 484)) score -= 4;             /* selected altitudes are multiples of 500 ft */
    }
    if (f) {
        f->commb_format = CB_VERTICAL_INTENT;
        if (mcp_v) { f->nav_flags |= ORACLE_NAV_MCP_ALT_VALID; f->nav_mcp_altitude = alt[0]; }
        if (fms_v) { f->nav_flags |= ORACLE_NAV_FMS_ALT_VALID; f->nav_fms_altitude = alt[1]; }
        if (baro_v) { f->nav_flags |= ORACLE_NAV_QNH_VALID; f->nav_qnh = qnh; }
        if (mode_v) {
            f->nav_flags |= ORACLE_NAV_MODES_VALID;
            f->nav_modes = (uint8_t) (((mode & 4) ? NM_VNAV : 0) | ((mode & 2) ? NM_ALT_HOLD : 0) | ((mode & 1) ? NM_APPROACH : 0));
        }
        f->nav_altitude_source = (uint8_t) (src_v ? src + 1 : 0);     /* UNKNOWN AIRCRAFT MCP FMS = 1..4, readsb.h:287 */
    }
    return score;
}

static int cb_bds50(uint64_t mb, struct oracle_fields *f) {            /* comm_b.c:518-672 */
    const unsigned roll_s = MBF(2, 2), roll_raw = MBF(3, 11), trk_s = MBF(13, 13), trk_raw = MBF(14, 23), gs_raw = MBF(25, 34);
    const unsigned rate_v = MBF(35, 35), rate_s = MBF(36, 36), rate_raw = MBF(37, 45), tas_raw = MBF(47, 56);
    if (!MBF(1, 1) || !MBF(12, 12) || !MBF(24, 24) || !MBF(46, 46)) return 0;    /* roll, track, gs, tas must all be present */
    float roll = (float) (roll_raw * 45.0 / 256.0);
    if (roll_s) roll = (float) (roll - 90.0);
    if (!(roll >= -40 && roll < 40)) return 0;
    float track = (float) (trk_raw * 90.0 / 512.0);
    if (trk_s) track = (float) (track + 180.0);
    const unsigned gs = gs_raw * 2, tas = tas_raw * 2;
    if (gs < 50 || gs > 700 || tas < 50 || tas > 700) return 0;          /* (a zero raw value fails the same test) */
    int score = 11 + 12 + 11 + 11;
    float rate = 0;
    if (rate_v) {
        rate = (float) (rate_raw * 8.0 / 256.0);
        if (rate_s) rate = (float) (rate - 16);
        if (!(rate >= -10.0 && rate <= 10.0)) return 0;
        score += 11;
        /* coordinated-turn rate for this roll and airspeed against the reported one (comm_b.c:631-637) */
        const double turn = 68625 * tan(roll * M_PI / 180.0) / (tas * 20 * M_PI);
        if (fabs(turn - rate) > 2.0) score -= 6;
    } else if (rate_raw == 0 && !rate_s) score += 1;
    else return 0;
    if (f) {
        f->commb_format = CB_TRACK_TURN;
        f->flags |= ORACLE_F_ROLL_VALID | ORACLE_F_HEADING_VALID | ORACLE_F_GS_VALID | ORACLE_F_TAS_VALID;
        f->roll = roll;
        f->heading = track; f->heading_type = HD_GROUND_TRACK;
        f->gs_v0 = f->gs_v2 = f->gs_selected = (float) gs;
        if (rate_v) { f->flags |= ORACLE_F_TRACK_RATE_VALID; f->track_rate = rate; }
        f->tas = (uint16_t) tas;
    }
    return score;
}

static int cb_bds60(uint64_t mb, struct oracle_fields *f) {            /* comm_b.c:676-824 */
    const unsigned hdg_s = MBF(2, 2), hdg_raw = MBF(3, 12), ias = MBF(14, 23), mach_raw = MBF(25, 34);
    const unsigned rv[2] = {MBF(35, 35), MBF(46, 46)}, rs[2] = {MBF(36, 36), MBF(47, 47)}, rr[2] = {MBF(37, 45), MBF(48, 56)};
    if (!MBF(1, 1) || !MBF(13, 13) || !MBF(24, 24) || !(rv[0] | rv[1])) return 0;
    float heading = (float) (hdg_raw * 90.0 / 512.0);
    if (hdg_s) heading = (float) (heading + 180.0);
    if (ias < 50 || ias > 700 || mach_raw == 0) return 0;
    const float mach = (float) (mach_raw * 2.048 / 512);
    if (!(mach >= 0.1 && mach <= 0.9)) return 0;
    int score = 12 + 11 + 11, rate[2] = {0, 0};
    for (int k = 0; k < 2; ++k) {                                       /* barometric, then inertial vertical rate */
        if (rv[k]) {
            rate[k] = (int) rr[k] * 32 - (rs[k] ? 16384 : 0);
            if (rate[k] < -6000 || rate[k] > 6000) return 0;
            score += 11;
        } else if (rr[k] == 0) score += 1;
        else return 0;
    }
    if (rv[0] && rv[1] && abs(rate[0] - rate[1]) > 2000) score -= 12;
    if (f) {
        f->commb_format = CB_HEADING_SPEED;
        f->flags |= ORACLE_F_HEADING_VALID | ORACLE_F_IAS_VALID | ORACLE_F_MACH_VALID;
        f->heading = heading; f->heading_type = HD_MAGNETIC;
        f->ias = (uint16_t) ias;
        f->mach = mach;
        if (rv[0]) { f->flags |= ORACLE_F_BARO_RATE_VALID; f->baro_rate = rate[0]; }
        if (rv[1]) { f->flags |= ORACLE_F_GEOM_RATE_VALID; f->geom_rate = rate[1]; }
    }
    return score;
}

/* BDS4,4 as the reference scores it (comm_b.c:828-961), including what its arithmetic really does: the wind direction
 * factor 180/256 is an integer division (direction is always 0), and a set static-pressure status bit always ends in
 * `return 0` (the 11-bit value cannot exceed 2048), so only reports without pressure can win. */
static int cb_bds44(uint64_t mb, struct oracle_fields *f) {
    const unsigned source = MBF(1, 4), wind_v = MBF(5, 5), wind = MBF(6, 14), t_raw = MBF(25, 34);
    const unsigned turb_v = MBF(47, 47), hum_v = MBF(50, 50);
    if (source > 6) return 0;
    if (MBF(35, 35)) return 0;
    const float oat = (float) (MBF(24, 24) ? (t_raw - 1024.0) * 0.25 : t_raw * 0.25);
    if (!(oat >= -128 && oat <= 128)) return 0;
    const int score = 4 + (wind_v ? 18 : 2) + 10 + 1 + (turb_v ? 2 : 1) + (hum_v ? 6 : 1);
    if (f) {
        f->commb_format = CB_MET_ROUTINE;
        f->flags |= ORACLE_F_MET_SOURCE_VALID | ORACLE_F_OAT_VALID;
        f->met_source = (uint8_t) source;
        if (wind_v) { f->flags |= ORACLE_F_WIND_VALID; f->wind_speed = (uint16_t) wind; f->wind_direction = 0; }
        f->oat = oat;
        if (turb_v) { f->flags |= ORACLE_F_TURBULENCE_VALID; f->turbulence = (uint8_t) MBF(48, 49); }
        if (hum_v) { f->flags |= ORACLE_F_HUMIDITY_VALID; f->humidity = MBF(51, 56) * (100.0f / 64); }
    }
    return score;
}

static void comm_b(struct oracle_fields *f, uint64_t mb) {
    typedef int (*hyp_fn)(uint64_t, struct oracle_fields *);
    static const hyp_fn hyp[9] = {cb_empty, cb_bds10, cb_bds20, cb_bds30, cb_bds17, cb_bds40, cb_bds50, cb_bds60, cb_bds44};
    f->commb_format = CB_UNKNOWN;
    /* comm_b.c:58 also tests mm->UM and mm->correctedbits, but decodeModesMessage extracts UM only AFTER it has called
     * decodeCommB (mode_s.c:713-716 vs :755-757), so UM is still 0 there, and DF20/21 never carry corrected bits */
    if (f->DR) return;
    int best = 0, winner = -1, ties = 0;
    for (int k = 0; k < 9; ++k) {
        const int sc = hyp[k](mb, NULL);
        if (sc > best) { best = sc; winner = k; ties = 0; }
        else if (sc == best) ++ties;
    }
    if (winner < 0) return;
    if (ties) f->commb_format = CB_AMBIGUOUS;
    else hyp[winner](mb, f);
}

void modes_oracle_decode_fields(const uint8_t *msg, int msgbits, struct oracle_fields *f) {
    memset(f, 0, sizeof(*f));
    if (msgbits == 16) {                                         /* decodeModeAMessage, mode_ac.c:171-200 */
        const unsigned a = ((unsigned) msg[0] << 8) | msg[1];
        f->source = SRC_MODE_AC; f->addrtype = AT_MODE_A; f->msgtype = 77;
        f->addr = (a & 0xff7f) | NON_ICAO;
        f->squawkHex = (uint16_t) (a & 0x7777);
        f->squawkDec = (uint16_t) squawk_dec(f->squawkHex);
        f->flags |= ORACLE_F_SQUAWK_VALID | ORACLE_F_SPI_VALID | ((a & 0x80) ? ORACLE_F_SPI : 0);
        if (!(a & 0x80)) {
            /* modeAToModeC indexes its table with the 12 code bits; the table entry carries the zero-bit check (mode_ac.c:80-86,102) */
            const int c = mode_a_to_c(a);
            if (c != BAD_ALT) { f->baro_alt = c * 100; f->baro_alt_unit = 0; f->flags |= ORACLE_F_BARO_ALT_VALID; }
        }
        return;
    }
    const unsigned df = msg[0] >> 3;
    const uint32_t head = ((uint32_t) msg[0] << 24) | ((uint32_t) msg[1] << 16) | ((uint32_t) msg[2] << 8) | msg[3];
#define HD(a, b) bits_of(head, 32, (a), (b))
    f->msgtype = (uint8_t) df;
    const int ap = !(df == 11 || df == 17 || df == 18);          /* Address/Parity: the syndrome is the address (mode_s.c:464-482) */
    const uint32_t crc = modes_oracle_checksum(msg, msgbits);
    if (ap) { f->addr = crc; f->addrtype = AT_MODE_S; f->source = SRC_MODE_S; }
    else {
        f->AA = HD(9, 32);
        f->addr = f->AA;
        if (df == 11) { f->IID = (uint8_t) (crc & 0x7f); f->source = SRC_MODE_S_CHECKED; f->addrtype = AT_MODE_S; }
        else { f->addrtype = AT_ADSB_ICAO; f->source = SRC_ADSB; }
    }
    if (df == 0 || df == 4 || df == 16 || df == 20) {
        f->AC = (uint16_t) HD(20, 32);
        if (f->AC) {
            int q;
            const int alt = ac13_altitude(f->AC, &f->baro_alt_unit, &q);
            f->baro_alt = alt;
            if (alt != BAD_ALT) { f->flags |= ORACLE_F_BARO_ALT_VALID | (q ? ORACLE_F_ALT_Q_BIT : 0); }
        }
    }
    if (df == 11 || df == 17) {
        static const uint8_t ag_of_ca[8] = {AG_UNCERTAIN_, 0, 0, 0, AG_GROUND_, AG_AIRBORNE_, AG_UNCERTAIN_, AG_UNCERTAIN_};
        f->CA = (uint8_t) HD(6, 8);
        f->airground = ag_of_ca[f->CA];
    }
    if (df == 0) f->CC = (uint8_t) HD(7, 7);
    if (df == 18) f->CF = (uint8_t) HD(6, 8);
    if (df == 4 || df == 5 || df == 20 || df == 21) {
        f->DR = (uint8_t) HD(9, 13);
        f->UM = (uint8_t) HD(14, 19);
        f->FS = (uint8_t) HD(6, 8);
        /* flight status -> air/ground, alert, SPI (mode_s.c:664-697) */
        static const uint8_t ag[6] = {AG_UNCERTAIN_, AG_GROUND_, AG_UNCERTAIN_, AG_GROUND_, AG_UNCERTAIN_, AG_UNCERTAIN_};
        if (f->FS <= 5) {
            f->flags |= ORACLE_F_ALERT_VALID | ORACLE_F_SPI_VALID;
            f->airground = ag[f->FS];
            if (f->FS >= 2 && f->FS <= 4) f->flags |= ORACLE_F_ALERT;
            if (f->FS >= 4) f->flags |= ORACLE_F_SPI;
        }
    }
    if (df == 5 || df == 21) {
        f->ID = (uint16_t) HD(20, 32);
        if (f->ID) set_squawk(f, f->ID);
    }
    if (df >= 24) { f->KE = (uint8_t) HD(4, 4); f->ND = (uint8_t) HD(5, 8); }
    if (df == 20 || df == 21) {
        uint64_t mb = 0;
        for (int k = 4; k < 11; ++k) mb = (mb << 8) | msg[k];
        comm_b(f, mb);
    }
    if (df == 17 || df == 18) {
        uint64_t me = 0;
        for (int k = 4; k < 11; ++k) me = (me << 8) | msg[k];
        extended_squitter(f, me);
    }
    if (df == 16 && msg[4] == 0x30) f->flags |= ORACLE_F_ACAS_RA_VALID;
    if (df == 0 || df == 16) {
        f->RI = (uint8_t) HD(14, 17);
        f->SL = (uint8_t) HD(9, 11);
        f->VS = (uint8_t) HD(6, 6);
        f->airground = f->VS ? AG_GROUND_ : AG_UNCERTAIN_;
    }
#undef HD
}

void modes_oracle_decode_fields_batch(const uint8_t *msgs, const int32_t *msgbits, uint64_t n, struct oracle_fields *out) {
    for (uint64_t i = 0; i < n; ++i) modes_oracle_decode_fields(msgs + 14 * i, msgbits[i], &out[i]);
}
