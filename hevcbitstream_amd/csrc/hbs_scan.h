/* hbs_scan.h -- host-visible launcher of the fused scan/extract kernel (K12). */
#ifndef HBS_SCAN_H
#define HBS_SCAN_H

#include <hip/hip_runtime_api.h>
#include "hbs_common.h"

namespace hbs {

struct ScanArgs {
    const uint8_t* stream;        /* device, 16-byte aligned                      */
    uint64_t n;                   /* stream bytes                                 */
    hbs_nal_entry* index;         /* device, index_cap entries                    */
    uint64_t index_cap;
    uint8_t* rbsp;                /* device arena or nullptr (index only)         */
    uint64_t rbsp_cap;
    unsigned long long* desc;     /* workspace: 2 words per 16 KiB tile           */
    RunHeader* hdr;               /* workspace                                    */
    uint8_t* tail;                /* workspace of scan4_tail_bytes(): padded copy of the last tile (variant 4) */
    void* ws5;                    /* workspace of scan5_workspace_bytes(n): tile aggregates and recorded elements of the index-only
                                     kernels (hbs_scan5.hip); needed only when rbsp == nullptr */
    hbs_summary* summary;         /* device                                       */
    unsigned long long* ahead_cand;  /* workspace: a word per 192 KiB tile -- stamp | 1: the prologue's sample says "dense", | 2: counted (event-sparse kernel with count-ahead), or nullptr */
    uint32_t* ahead_list;         /* workspace: the tiles so marked, a word per tile of the stream at most                                */
    AheadCtl* ahead_ctl;          /* workspace: the call's number (-> its stamp) and the entries of the list (hbs_common.h)               */
    void* ahead_tab;              /* workspace: scan4_ahead_entry_bytes() per tile: the aggregates of the tiles counted ahead           */
              /* this call's number on its context (never 0): stamps the entries                                  */
    int grid_blocks;              /* persistent workgroups (<= resident capacity) of the LDS-image kernel */
    int grid_blocks4;                 /* ... of the event-sparse kernel                                        */
    int grid_blocks4r24;              /* ... of its 24-row geometry (hbs_scan4_r24.hip)                        */
    int first_static;                 /* hbs_ctx_set_device_exclusive: persistent kernels take their first tile by workgroup number (else by ticket) */
    int spare_wgs;                    /* hbs_ctx_reserve_workgroups: slots (of 256 threads) every scan kernel leaves free     */
    hipEvent_t ev_begin, ev_end;  /* when non-null: recorded around the main kernel only */
    int sched;                    /* tile schedule of the LDS-image kernel: 0 striped, 1 ticket at loop top, 2 ticket after prefix */
    int variant;                  /* 0: automatic (density probe, then event-sparse or LDS-image kernel, decided on the device),
                                     2: LDS-image kernel (hbs_scan.hip), 4: event-sparse kernel (hbs_scan4.hip),
                                     5: index only (rbsp == nullptr), streaming kernel (hbs_scan5.hip); with an arena it means 4 */
};

/* persistent grid size for `device` (CUs x co-resident workgroups per CU) */
int scan_grid_blocks(int device, int* blocks_per_cu_out);
hipError_t launch_scan_extract(const ScanArgs& a, hipStream_t st);
/* an arena-less call that goes to the streaming index-only kernel (5): pinned, or automatic from 1 GiB up (below, the
 * event-sparse kernel without an arena is quicker: scripts/experiments/index_only_by_size.py).  ONE rule for the launcher and
 * for what hbs_ctx_last_kernel reports. */
inline bool scan_uses_index_only(uint64_t n, int variant, const void* rbsp)
{
    return rbsp == nullptr && (variant == 5 || (variant == 0 && n >= (1ull << 30)));
}
/* automatic mode, at most one 64 KiB tile and a small index: one launch of one workgroup does the whole call */
bool scan_takes_small_path(uint64_t n, uint64_t index_cap, int variant);

/* event-sparse variant (hbs_scan4.hip) */
int scan4_grid_blocks(int device, int* blocks_per_cu_out);
int scan4_tile_bytes();
int scan4_tail_bytes();
/* header, probe, padded last tile, cleared index and look-back words: one launch in front of the main kernel */
void launch_scan_prologue(const ScanArgs& a, uint64_t desc_words, bool probe, int tail_tile_bytes /* 0: no padded copy */, hipStream_t st);
void scan4_launch_kernel(const ScanArgs& a, uint64_t num_tiles, int gate, hipStream_t st);
/* ... its 24-row geometry (hbs_scan4_r24.hip: 96 KiB tiles, 1024 elements a tile; variant 6) */
int scan4r24_grid_blocks(int device, int* blocks_per_cu_out);
int scan4r24_tile_bytes();
void scan4r24_launch_kernel(const ScanArgs& a, uint64_t num_tiles, int gate, hipStream_t st);
/* dense tiles counted ahead (round 5): bytes per tile of the table, the stream size from which a call uses it, the launch */
uint64_t scan4_ahead_entry_bytes();
bool scan4_counts_ahead(uint64_t n);
void launch_scan_ahead4(const ScanArgs& a, uint64_t num_tiles, int gate, hipStream_t st);

/* index-only streaming kernel (hbs_scan5.hip) */
uint64_t scan5_workspace_bytes(uint64_t stream_bytes);     /* for any tile height the launcher may pick */
struct Geo5 { int rows; int strided; uint64_t tiles, grid; };   /* KiB per tile; tiles dealt in whole rounds (no ticket); tiles; one-wavefront workgroups */
Geo5 scan5_geometry(uint64_t n, uint64_t waves);            /* for a stream of n bytes on at most `waves` resident wavefronts */
int scan5_tile_rows(uint64_t n, uint64_t waves);
void launch_scan_index5(const ScanArgs& a, int gate, hipStream_t st);

} // namespace hbs
#endif
