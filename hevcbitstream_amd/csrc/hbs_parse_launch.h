/* hbs_parse_launch.h -- host-visible launcher of K4 (header parse). */
#ifndef HBS_PARSE_LAUNCH_H
#define HBS_PARSE_LAUNCH_H

#include <hip/hip_runtime_api.h>
#include "hbs_common.h"

namespace hbs {

struct ParsedNal;
struct TraceRec;
struct RpsRow;
struct SliceCompact;
constexpr unsigned kParseMaxBlocks = 2048;

struct ParseArgs {
    const uint8_t* rbsp;
    const hbs_nal_entry* index;
    uint64_t n;
    ParsedNal* parsed;               /* n records (device)                        */
    uint8_t* structs;                /* struct arena or nullptr (plan only)       */
    uint64_t structs_cap;
    hbs_summary* summary;
    /* workspace */
    unsigned long long* slot_size;   /* n */
    long long* ctx_sps;              /* n */
    long long* ctx_pps;              /* n */
    const uint8_t* zeros;            /* >= sizeof(hevc_sps_t) zero bytes */
    const uint8_t* initial_sps_slot; /* optional: SPS slot (struct + RPS tables) in force before NAL 0 */
    const uint8_t* initial_pps;      /* optional: hevc_pps_t in force before NAL 0 */
    unsigned long long* total;
    uint32_t* err;
    uint32_t* div_flag;              /* set by a slice whose answer depends on NALs in front of its SPS: the batch is then parsed again in order */
    void* scan_tmp;                  /* 1024 x 24 bytes */
    TraceRec* trace;                 /* optional: trace_cap records per NAL (device) */
    uint32_t trace_cap;
    uint32_t* trace_count;           /* optional: records each NAL produced (may exceed trace_cap) */
    RpsRow* own_rows;                /* parse_own_rows_bytes(n): one row per lane of the parse grid */
    /* the exact re-walk of the few slices a flagged batch needs (hbs_parse_fix.h) */
    uint32_t* deps;                  /* n: what every slice's walk did with the derived tables */
    uint32_t* wmask;                 /* n: rows each NAL writes */
    uint32_t* bsum;                  /* n / 256 + 1 */
    uint32_t* fix_list;              /* n: ordinals of the slices to walk again; fix_count[0] how many, fix_count[1] "walk the whole batch in order after all" */
    uint32_t* fix_count;
    RpsRow* fix_temps;               /* parse_fix_temps_bytes() */
    /* optional: the parameter sets in force BEHIND the last NAL of the batch and the derived tables as the reference would
     * hold them there (hbs_parse_headers_state): an SPS slot (struct + tables) and a hevc_pps_t */
    uint8_t* state_sps_slot_out;
    uint8_t* state_pps_out;
    /* hbs_parse_headers_compact / hbs_parse_materialize: when `compact` is set, slices are walked WITHOUT a struct (a 64-byte
     * record each in compact[k]) -- except the want_n slices listed in want_list, which get their slot in the struct arena as
     * in the full parse.  No trace, no state output, not sequential. */
    SliceCompact* compact;           /* n records (device) or nullptr */
    const uint64_t* want_list;       /* want_n NAL numbers (device) or nullptr */
    uint64_t want_n;
    uint8_t* fix_structs;            /* parse_fix_structs_bytes(): a slice slot per lane of the exact re-walk, for slices without one */
    int sequential;                  /* n == 1 only: the RPS tables behind initial_sps_slot are read AND written, as the
                                        reference's file-static tables are (what the legacy single-NAL symbols need) */
};

unsigned parse_grid_blocks(uint64_t n);
uint64_t parse_own_rows_bytes(uint64_t n);
uint64_t parse_fix_temps_bytes();
uint64_t parse_fix_structs_bytes();
hipError_t launch_parse_headers(const ParseArgs& a, hipStream_t st);
hipError_t launch_parse_extended(const uint8_t* rbsp, const hbs_nal_entry* index, uint64_t n, ParsedNal* parsed, hbs_ext_nal* ext, hipStream_t st);

struct WrittenNal;

struct WriteArgs {
    const ParsedNal* parsed;         /* n records: type, layer, temporal id, struct_off            */
    uint64_t n;
    uint8_t* structs;                /* struct arena (an SPS's derived tables are refreshed)        */
    uint8_t* rbsp_out;               /* n x rbsp_cap bytes                                          */
    uint32_t rbsp_cap;
    WrittenNal* written;             /* n records                                                   */
    /* workspace */
    unsigned long long* slot_size;   /* n */
    long long* ctx_sps;              /* n */
    long long* ctx_pps;              /* n */
    const uint8_t* zeros;
    const uint8_t* initial_sps_slot;
    const uint8_t* initial_pps;
    unsigned long long* total;
    void* scan_tmp;
    RpsRow* own_rows;                /* parse_own_rows_bytes(n) */
};

hipError_t launch_write_headers(const WriteArgs& a, hipStream_t st);

} // namespace hbs
#endif
