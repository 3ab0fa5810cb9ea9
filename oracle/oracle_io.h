/* oracle_io.h — record formats shared by the two CPU checkers under oracle/:
 *   (1) oracle/_ref      : the reference's own C files compiled where they lie
 *   (2) oracle/modes_oracle.c : our plain-C restatement of the same algorithm
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under readsb_amd/ (the product) may
 * include, link or execute anything under oracle/.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 */
#ifndef ORACLE_IO_H
#define ORACLE_IO_H
#include <stdint.h>

#define ORACLE_FMT_UC8     0
#define ORACLE_FMT_SC16    1
#define ORACLE_FMT_SC16Q11 2

/* fixed synthetic wall-clock the ifile grid is anchored to */
#define ORACLE_STARTUP_MS  1000000LL

/* One accepted message as seen at netUseMessage() (demod_2400.c:471). 72 bytes. */
struct oracle_msg {
    int64_t  timestamp;      /* mm->timestamp, 12 MHz ticks (demod_2400.c:406) */
    int64_t  sys_rel_ms;     /* mm->sysTimestamp - startup_time (demod_2400.c:409) */
    int32_t  score;          /* mm->score */
    int32_t  correctedbits;  /* mm->correctedbits */
    int32_t  msgbits;        /* mm->msgbits (after DF fix) */
    int32_t  msgtype;        /* mm->msgtype (after DF fix) */
    uint32_t addr;           /* mm->addr */
    uint8_t  msg[14];        /* mm->msg, i.e. after modesChecksumFix / DF fix */
    uint8_t  raw[14];        /* the bytes as sliced, before any correction */
    double   signalLevel;    /* mm->signalLevel (demod_2400.c:448) */
};

/* struct stats demod counters (stats.h:62-82) accumulated over the whole run. */
struct oracle_stats {
    uint64_t demod_preambles;
    uint64_t demod_rejected_bad;
    uint64_t demod_rejected_unknown_icao;
    uint64_t demod_accepted[3];
    uint64_t demod_preamblePhase[5];
    uint64_t demod_bestPhase[5];
    uint64_t strong_signal_count;
    uint64_t signal_power_count;
    uint64_t noise_power_count;
    uint64_t samples_processed;
    uint64_t samples_lost;
    uint64_t nbuffers;
    uint64_t nflips;         /* number of icaoFilterExpire() calls */
    double   signal_power_sum;
    double   noise_power_sum;
    double   peak_signal_power;
    double   t_convert_s;    /* CLOCK_MONOTONIC around the converter calls */
    double   t_demod_s;      /* CLOCK_MONOTONIC around demodulate2400() calls */
    uint64_t demod_modeac;   /* Mode A/C replies accepted by demodulate2400AC (stats.h:72), when enabled */
};

/* Per-message fields decodeModesMessage() leaves in struct modesMessage behind the CRC stage
 * (mode_s.c:598-803: AA AC CA CC CF DR FS ID KE ND RI SL UM VS; decodeExtendedSquitter and its eight
 * ME decoders, mode_s.c:806-1555), and what decodeModeAMessage (mode_ac.c:171-200) sets for a
 * Mode A/C reply, and what decodeCommB (comm_b.c:52-961) infers from the MB field of DF20/21.
 * Everything a memset-0 modesMessage would hold stays 0.  176 bytes, same layout as struct mgpu_fields. */
struct oracle_fields {
    uint32_t addr;              /* mm->addr (with MODES_NON_ICAO_ADDRESS = 1<<24 where the ME decode says so) */
    uint32_t AA;
    uint32_t flags;             /* ORACLE_F_* */
    uint16_t acc_flags;         /* ORACLE_ACC_* */
    uint8_t  nav_flags;         /* ORACLE_NAV_* */
    uint8_t  msgtype;           /* DF, 77 = Mode A/C */
    uint8_t  addrtype, source, airground, metype;
    uint8_t  mesub, CA, CC, CF;
    uint8_t  DR, FS, KE, ND;
    uint8_t  RI, SL, UM, VS;
    uint8_t  IID, category, emergency, cpr_type;
    uint16_t AC, ID;
    uint16_t squawkHex, squawkDec;
    int32_t  baro_alt, geom_alt;
    int32_t  geom_delta, baro_rate, geom_rate;
    uint16_t ias, tas;
    float    heading, gs_v0, gs_v2, gs_selected;
    uint32_t cpr_lat, cpr_lon;
    char     callsign[8];
    uint8_t  baro_alt_unit, geom_alt_unit, heading_type, sil_type;
    uint8_t  nac_p, nac_v, sil, gva;
    uint8_t  sda, op_version, op_hrd, op_tah;
    uint16_t op_flags;          /* ORACLE_OP_* */
    uint8_t  op_cc_lw, op_cc_antenna_offset;
    uint8_t  op_cc_tc, nav_heading_type, nav_altitude_source, nav_modes;
    uint32_t nav_fms_altitude, nav_mcp_altitude;
    float    nav_qnh, nav_heading;
    float    roll, track_rate, mach;          /* Comm-B BDS5,0 / 6,0 (mm->mach is a double holding this float) */
    float    oat, humidity, wind_direction;   /* Comm-B BDS4,4 */
    uint16_t wind_speed, static_pressure;
    uint8_t  commb_format, met_source, turbulence, pad0;
    uint8_t  reserved[8];
};

/* flags: the bools of struct modesMessage (readsb.h:954-993) */
#define ORACLE_F_BARO_ALT_VALID   (1u << 0)
#define ORACLE_F_GEOM_ALT_VALID   (1u << 1)
#define ORACLE_F_HEADING_VALID    (1u << 2)
#define ORACLE_F_GS_VALID         (1u << 3)
#define ORACLE_F_IAS_VALID        (1u << 4)
#define ORACLE_F_TAS_VALID        (1u << 5)
#define ORACLE_F_BARO_RATE_VALID  (1u << 6)
#define ORACLE_F_GEOM_RATE_VALID  (1u << 7)
#define ORACLE_F_SQUAWK_VALID     (1u << 8)
#define ORACLE_F_CALLSIGN_VALID   (1u << 9)
#define ORACLE_F_CPR_VALID        (1u << 10)
#define ORACLE_F_CPR_ODD          (1u << 11)
#define ORACLE_F_CATEGORY_VALID   (1u << 12)
#define ORACLE_F_GEOM_DELTA_VALID (1u << 13)
#define ORACLE_F_SPI_VALID        (1u << 14)
#define ORACLE_F_SPI              (1u << 15)
#define ORACLE_F_ALERT_VALID      (1u << 16)
#define ORACLE_F_ALERT            (1u << 17)
#define ORACLE_F_EMERGENCY_VALID  (1u << 18)
#define ORACLE_F_ALT_Q_BIT        (1u << 19)
#define ORACLE_F_ACAS_RA_VALID    (1u << 20)
#define ORACLE_F_ROLL_VALID       (1u << 21)
#define ORACLE_F_TRACK_RATE_VALID (1u << 22)
#define ORACLE_F_MACH_VALID       (1u << 23)
#define ORACLE_F_WIND_VALID       (1u << 24)
#define ORACLE_F_OAT_VALID        (1u << 25)
#define ORACLE_F_STATIC_PRESSURE_VALID (1u << 26)
#define ORACLE_F_TURBULENCE_VALID (1u << 27)
#define ORACLE_F_HUMIDITY_VALID   (1u << 28)
#define ORACLE_F_MET_SOURCE_VALID (1u << 29)
/* acc_flags: mm->accuracy (readsb.h:1061-1086) */
#define ORACLE_ACC_NIC_A_VALID    (1u << 0)
#define ORACLE_ACC_NIC_B_VALID    (1u << 1)
#define ORACLE_ACC_NIC_C_VALID    (1u << 2)
#define ORACLE_ACC_NIC_BARO_VALID (1u << 3)
#define ORACLE_ACC_NAC_P_VALID    (1u << 4)
#define ORACLE_ACC_NAC_V_VALID    (1u << 5)
#define ORACLE_ACC_GVA_VALID      (1u << 6)
#define ORACLE_ACC_SDA_VALID      (1u << 7)
#define ORACLE_ACC_NIC_A          (1u << 8)
#define ORACLE_ACC_NIC_B          (1u << 9)
#define ORACLE_ACC_NIC_C          (1u << 10)
#define ORACLE_ACC_NIC_BARO       (1u << 11)
/* nav_flags: mm->nav (readsb.h:1127-1142) */
#define ORACLE_NAV_HEADING_VALID  (1u << 0)
#define ORACLE_NAV_FMS_ALT_VALID  (1u << 1)
#define ORACLE_NAV_MCP_ALT_VALID  (1u << 2)
#define ORACLE_NAV_QNH_VALID      (1u << 3)
#define ORACLE_NAV_MODES_VALID    (1u << 4)
/* op_flags: the one-bit members of mm->opstatus (readsb.h:1103-1120) */
#define ORACLE_OP_VALID       (1u << 0)
#define ORACLE_OP_OM_ACAS_RA  (1u << 1)
#define ORACLE_OP_OM_IDENT    (1u << 2)
#define ORACLE_OP_OM_ATC      (1u << 3)
#define ORACLE_OP_OM_SAF      (1u << 4)
#define ORACLE_OP_CC_ACAS     (1u << 5)
#define ORACLE_OP_CC_CDTI     (1u << 6)
#define ORACLE_OP_CC_1090_IN  (1u << 7)
#define ORACLE_OP_CC_ARV      (1u << 8)
#define ORACLE_OP_CC_TS       (1u << 9)
#define ORACLE_OP_CC_UAT_IN   (1u << 10)
#define ORACLE_OP_CC_POA      (1u << 11)
#define ORACLE_OP_CC_B2_LOW   (1u << 12)
#define ORACLE_OP_CC_LW_VALID (1u << 13)

#endif
