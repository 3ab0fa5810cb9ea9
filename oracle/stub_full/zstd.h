/* Declarations-only stand-in for <zstd.h> (libzstd-dev headers are absent in this container; the runtime
 * libzstd.so.1 is present): just what the reference tree uses, with the signatures of zstd 1.4.x, so that the
 * WHOLE reference program can be built as a checker (oracle/Makefile: `make full`).  TEST INFRASTRUCTURE ONLY. */
#ifndef ORACLE_STUB_FULL_ZSTD_H
#define ORACLE_STUB_FULL_ZSTD_H
#include <stddef.h>
typedef struct ZSTD_CCtx_s ZSTD_CCtx;
typedef struct ZSTD_DCtx_s ZSTD_DCtx;
typedef ZSTD_CCtx ZSTD_CStream;
typedef struct ZSTD_inBuffer_s { const void *src; size_t size; size_t pos; } ZSTD_inBuffer;
typedef struct ZSTD_outBuffer_s { void *dst; size_t size; size_t pos; } ZSTD_outBuffer;
typedef enum { ZSTD_reset_session_only = 1, ZSTD_reset_parameters = 2, ZSTD_reset_session_and_parameters = 3 } ZSTD_ResetDirective;
typedef enum { ZSTD_c_compressionLevel = 100 } ZSTD_cParameter;
typedef enum { ZSTD_e_continue = 0, ZSTD_e_flush = 1, ZSTD_e_end = 2 } ZSTD_EndDirective;
size_t ZSTD_compressBound(size_t srcSize);
unsigned ZSTD_isError(size_t code);
const char *ZSTD_getErrorName(size_t code);
ZSTD_CCtx *ZSTD_createCCtx(void);
size_t ZSTD_freeCCtx(ZSTD_CCtx *cctx);
size_t ZSTD_compressCCtx(ZSTD_CCtx *cctx, void *dst, size_t dstCapacity, const void *src, size_t srcSize, int compressionLevel);
ZSTD_DCtx *ZSTD_createDCtx(void);
size_t ZSTD_freeDCtx(ZSTD_DCtx *dctx);
size_t ZSTD_decompressDCtx(ZSTD_DCtx *dctx, void *dst, size_t dstCapacity, const void *src, size_t srcSize);
size_t ZSTD_CCtx_reset(ZSTD_CCtx *cctx, ZSTD_ResetDirective reset);
size_t ZSTD_CCtx_setParameter(ZSTD_CCtx *cctx, ZSTD_cParameter param, int value);
ZSTD_CStream *ZSTD_createCStream(void);
size_t ZSTD_freeCStream(ZSTD_CStream *zcs);
size_t ZSTD_initCStream(ZSTD_CStream *zcs, int compressionLevel);
size_t ZSTD_compressStream(ZSTD_CStream *zcs, ZSTD_outBuffer *output, ZSTD_inBuffer *input);
size_t ZSTD_compressStream2(ZSTD_CCtx *cctx, ZSTD_outBuffer *output, ZSTD_inBuffer *input, ZSTD_EndDirective endOp);
size_t ZSTD_flushStream(ZSTD_CStream *zcs, ZSTD_outBuffer *output);
size_t ZSTD_endStream(ZSTD_CStream *zcs, ZSTD_outBuffer *output);
#endif
