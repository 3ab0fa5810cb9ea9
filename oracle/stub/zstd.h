/* Type-only stand-in for <zstd.h> so the reference's readsb.h parses in this
 * container (libzstd-dev headers are absent).  Nothing on the demodulator hot
 * path calls zstd; the oracle objects never reference a ZSTD_* symbol.
 * TEST INFRASTRUCTURE ONLY. */
#ifndef ORACLE_STUB_ZSTD_H
#define ORACLE_STUB_ZSTD_H
#include <stddef.h>
typedef struct ZSTD_CCtx_s ZSTD_CCtx;
typedef struct ZSTD_DCtx_s ZSTD_DCtx;
typedef ZSTD_CCtx ZSTD_CStream;
typedef struct ZSTD_inBuffer_s { const void *src; size_t size; size_t pos; } ZSTD_inBuffer;
typedef struct ZSTD_outBuffer_s { void *dst; size_t size; size_t pos; } ZSTD_outBuffer;
#endif
