// gs_inflate.hpp — device-side gzip (gs_inflate.hip) as seen by gs_files.hip
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <vector>
#include <hip/hip_runtime.h>
struct gs_ctx;
namespace gs {
enum { INF_OK = 0, INF_E_BTYPE = 1, INF_E_STORED = 2, INF_E_OVERSUB = 3, INF_E_CODE = 4, INF_E_DIST = 5, INF_E_OUTPUT = 6, INF_E_INPUT = 7, INF_E_REPEAT = 8 };
// one deflate stream: bytes [in_off, in_off + in_len) of the compressed buffer (the member after its gzip header, trailer included) ->
// text at out_off (64-byte aligned) of the text buffer, at most out_cap bytes (the slot is padded to a multiple of 64)
struct InflateStream { uint64_t in_off, in_len, out_off, out_cap; };
struct InflateResult { uint32_t status, blocks; uint64_t in_used, out_len; };
int inflate_streams_dev(gs_ctx *c, const void *comp_dev, const InflateStream *streams, uint32_t n, void *out_dev, InflateResult *results);
int inflate_streams_launch(gs_ctx *c, hipStream_t stream, const void *comp_dev, const InflateStream *streams_pinned, uint32_t n, void *out_dev, void *ds_dev, void *dr_dev,
                           InflateResult *results_pinned);
int crc32_texts_dev(gs_ctx *c, const void *text_dev, const uint64_t *text_off, const uint64_t *text_len, uint32_t n, uint32_t *crc_out);
int fasta_scan_dev(gs_ctx *c, const void *text_dev, const uint64_t *off, const uint64_t *len, uint32_t n, std::vector<std::vector<uint64_t>> &sb,
                   std::vector<std::vector<uint64_t>> &se);
size_t gzip_header_len(const uint8_t *p, size_t n);
}  // namespace gs
