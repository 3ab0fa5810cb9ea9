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

#endif
