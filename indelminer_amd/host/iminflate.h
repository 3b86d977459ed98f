/* iminflate.h -- raw DEFLATE decoder for BGZF blocks (see iminflate.c) */
#ifndef IM_INFLATE_H
#define IM_INFLATE_H

#include <stddef.h>
#include <stdint.h>

#define IM_INFLATE_SLACK 16     /* bytes the caller keeps readable behind the input and writable behind the output: refills and match copies go word-wise */

/* One complete raw deflate stream in[0..in_len) -> out.  Returns the decoded size, or -1 when the stream is corrupt, runs
 * past its input or does not fit out_cap.  Up to IM_INFLATE_SLACK bytes behind in[in_len) are READ (never used), and as many
 * behind out[out_cap) may be written. */
int64_t im_inflate(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap);

#endif
