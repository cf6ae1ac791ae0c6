/* hbs_hdrwin.h -- host-visible launchers of the header windows (hbs_hdrwin.hip, hbs_index_parse). */
#ifndef HBS_HDRWIN_H
#define HBS_HDRWIN_H

#include <hip/hip_runtime_api.h>
#include "hbs_common.h"

namespace hbs {

/* hbs_parsed_nal as the kernels see it */
struct ParsedWin { int32_t rc, nal_unit_type, nal_layer_id, nal_temporal_id_plus1; uint64_t struct_off; int32_t slice_data_size; uint32_t slice_data_off; };
static_assert(sizeof(ParsedWin) == sizeof(hbs_parsed_nal), "hbs_parsed_nal");

struct HdrWinArgs {
    const uint8_t* stream;
    const hbs_nal_entry* index;      /* the scan's index (rbsp_len = the NAL's real RBSP length) */
    uint64_t nals, index_cap;
    uint32_t window;                 /* RBSP bytes stripped per slice segment */
    uint8_t* arena;                  /* hdrwin_arena_bytes(): index_cap slots of `window` bytes, then the bump area of the parameter sets */
    uint64_t arena_bytes;
    hbs_nal_entry* idx2;             /* index_cap entries: the index K4 parses, pointing into the arena */
    unsigned long long* bump;        /* 16 bytes: the bump counter, then flags */
    void* notes;                     /* index_cap x 16 bytes: where each window's first emulation prevention bytes were */
};

uint64_t hdrwin_arena_bytes(uint64_t index_cap, uint32_t window, uint64_t stream_bytes);
hipError_t launch_hdr_strip(const HdrWinArgs& a, hipStream_t st);
hipError_t launch_hdr_fix(const HdrWinArgs& a, void* parsed, hbs_summary* summary, unsigned long long* payload_off, hipStream_t st, int compact = 0);

} // namespace hbs
#endif
